cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --hip-trace --stats --output-format csv -d gpurun_out/blocktrace -o bt -- python tools/probe_block.py > gpurun_out/blocktrace.log 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/blocktrace/**/*hip_api_trace.csv',recursive=True)
rows=list(csv.DictReader(open(f[0])))
t0=min(int(r['Start_Timestamp']) for r in rows)
rows.sort(key=lambda r:int(r['End_Timestamp'])-int(r['Start_Timestamp']),reverse=True)
for r in rows[:40]:
    print(r['Function'], round((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6,1),'ms', 'tid',r.get('Thread_Id'), 'start', round((int(r['Start_Timestamp'])-t0)/1e6,1))
PY
rm -rf gpurun_out/blocktrace
