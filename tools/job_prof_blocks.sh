K=${1:-48}
mkdir -p gpurun_out/r05
export PYTHONPATH=/root/repo
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r05/blk$K -o blk -- python /root/repo/tools/probe_block_concurrency.py $K 2 2>&1 | grep "K=" 
cd /root/repo
python - <<PY
import csv,glob
f=glob.glob("gpurun_out/r05/blk$K/**/*kernel_stats.csv", recursive=True)
print(f)
rows=list(csv.DictReader(open(f[0])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot/1e6)
for r in rows[:30]: print(r["Name"][:70].ljust(70), r["Calls"].rjust(7), ("%.1f"%(float(r["TotalDurationNs"])/1e6)).rjust(9), ("%.3f"%(float(r["AverageNs"])/1e6)).rjust(9), r["Percentage"])
PY
find gpurun_out/r05/blk$K -name "*kernel_trace.csv" -size +20M -delete
