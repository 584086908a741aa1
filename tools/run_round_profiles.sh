#!/bin/bash
# Reproduces the artefacts under profiles/rNN/ on a GPU box (run from the repo root):
#   bash tools/run_round_profiles.sh gpurun_out/r02
# then copy the directory's contents into profiles/r02/ and run `python tools/make_profile_readme.py r02`.
set -u
OUT=${1:-gpurun_out/profiles}
ROOT=$(pwd)
mkdir -p "$OUT"
OUT=$(cd "$OUT" && pwd)
export TMPDIR=/tmp
export ZKW_ROOT="$ROOT"
# 1. the default bench (throughput leg + full-block leg + CPU legs), without a profiler
S0=$(date +%s)
timeout -s KILL 1800 python bench.py --steps 20 > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
echo "python bench.py --steps 20: $(( $(date +%s) - S0 )) s of wall time" > "$OUT/bench_wall_time.txt"
timeout -s KILL 900 python bench.py --pipelines 1 --no-cpu-baseline --no-full-block --no-h2d --no-sensitivity > "$OUT/bench_sequential.json" 2> "$OUT/bench_sequential.err"
# 2. the same command under rocprofv3 --kernel-trace --stats, the batched full-block leg with its 512 blocks in flight included (round 5: that leg
#    died inside the HIP runtime under the profiler with ~500 host threads; round 6 has one thread per batch)
cd /tmp && rm -rf /tmp/prof_stats && timeout -s KILL 1800 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- \
    python "$ROOT/bench.py" --steps 5 --no-cpu-baseline --no-sensitivity > "$OUT/bench_default_under_rocprofv3.json" 2> "$OUT/rocprof_stats.err"
f=$(ls /tmp/prof_stats/*/*kernel_stats.csv 2>/dev/null | head -1)
echo "rocprofv3 --kernel-trace --stats over bench.py with 512 blocks in flight in the batched leg: $([ -n "$f" ] && [ -s "$OUT/bench_default_under_rocprofv3.json" ] && echo ok || echo FAILED)" > "$OUT/rocprof_stats_runs.txt"
[ -n "$f" ] && cp "$f" "$OUT/bench_default_kernel_stats.csv"
# 2b. the block leg alone (tools/probe_blocks_pipeline.py: 512 blocks in flight, batch after batch) under the profiler: what the builders' merged
#     launches and the synthesis cost per block
rm -rf /tmp/prof_blocks && timeout -s KILL 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_blocks -- \
    python "$ROOT/tools/probe_blocks_pipeline.py" 512 2 seq device > "$OUT/blocks_under_rocprofv3.txt" 2>&1
f=$(ls /tmp/prof_blocks/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp "$f" "$OUT/blocks_kernel_stats.csv"
# 3. HBM counters AT THE BENCHMARKED BATCH, one pass each (never combined with other trace domains): one timed step of the
#    sequential form (counter collection serialises the dispatches anyway)
for C in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU SQ_BUSY_CYCLES; do
    rm -rf /tmp/prof_pmc && timeout -s KILL 1500 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/prof_pmc -- \
        python "$ROOT/bench.py" --pipelines 1 --steps 1 --warmup 0 --no-cpu-baseline --no-full-block --no-h2d --no-sensitivity --no-validate > "$OUT/bench_pmc_$C.json" 2> "$OUT/rocprof_pmc_$C.err"
    python3 - "$(ls /tmp/prof_pmc/*/*counter_collection.csv | head -1)" "$OUT/bench_pmc_$C.summary.csv" <<'PY'
import collections, csv, os, sys
sys.path.insert(0, os.path.join(os.environ.get("ZKW_ROOT", "."), "tools"))
from kernel_names import short  # the k_single / k_multi forms of a kernel body by the name the kernel had (tools/kernel_names.py)
tot, disp = collections.defaultdict(float), collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = short(r["Kernel_Name"])
    tot[k] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
with open(sys.argv[2], "w") as f:
    f.write("kernel,dispatches,counter_total,counter_per_dispatch\n")
    for k in sorted(tot, key=lambda k: -tot[k]):
        f.write(f"\"{k}\",{len(disp[k])},{tot[k]:.0f},{tot[k]/len(disp[k]):.1f}\n")
PY
done
cd "$ROOT"
# 4. one production-capacity block: spans of zkw_block_run, per-kernel times of every branch, bit-exactness vs the oracle
ZKW_BLOCK_PROFILE=1 timeout -s KILL 600 python tools/probe_block.py 2>&1 | grep -v amdgpu.ids > "$OUT/full_block_probe.txt"
# 5. synthesis throughput of the other circuit types at production geometry
for P in ds es ld ss; do
    echo "== tools/probe_${P}_synth.py" >> "$OUT/synthesis_probes.txt"
    timeout -s KILL 300 python tools/probe_${P}_synth.py 2>&1 | grep -v amdgpu.ids >> "$OUT/synthesis_probes.txt"
done
echo "== tools/probe_ecrecover_synth.py (ECRecover, 7 requests per instance, 2^20 rows; 8 and 32 instances per call)" >> "$OUT/synthesis_probes.txt"
timeout -s KILL 300 python tools/probe_ecrecover_synth.py 2>&1 | grep -v amdgpu.ids >> "$OUT/synthesis_probes.txt"
rm -rf /tmp/pk_ec && timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk_ec -- python tools/probe_ecrecover_synth.py > /dev/null 2>&1
f=$(ls /tmp/pk_ec/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" "$OUT/ecrecover_kernel_stats.csv"  # (k_ec_segments / k_ec_leaves / k_ec_stream run on the side stream: only the trace times them)
echo "== tools/probe_netlist_perf.py (Keccak256RoundFunction 293 / Sha256RoundFunction 2206 cycles, 2^20 rows, 8 instances; L1MessagesHasher)" >> "$OUT/synthesis_probes.txt"
timeout -s KILL 300 python tools/probe_netlist_perf.py 2>&1 | grep -v amdgpu.ids >> "$OUT/synthesis_probes.txt"
echo "== tools/probe_setup_commit.py (setup side as field elements: NTT / LDE / Merkle tree of 131 columns x 2^20, zkw_setup_commit of three layouts)" >> "$OUT/synthesis_probes.txt"
timeout -s KILL 300 python tools/probe_setup_commit.py 2>&1 | grep -v amdgpu.ids >> "$OUT/synthesis_probes.txt"
rm -rf /tmp/pk_sc && timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk_sc -- python tools/probe_setup_commit.py > /dev/null 2>&1
cp "$(ls /tmp/pk_sc/*/*kernel_stats.csv | head -1)" "$OUT/setup_commit_kernel_stats.csv"
rm -rf /tmp/pk_sv && timeout -s KILL 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU --output-format csv -d /tmp/pk_sv -- python tools/probe_setup_commit.py > /dev/null 2>&1
python3 - "$(ls /tmp/pk_sv/*/*counter_collection.csv | head -1)" > "$OUT/setup_commit_valu.txt" <<'PY'
import collections, csv, os, sys
sys.path.insert(0, os.path.join(os.environ.get("ZKW_ROOT", "."), "tools"))
from kernel_names import short  # the k_single / k_multi forms of a kernel body by the name the kernel had (tools/kernel_names.py)
tot, disp = collections.defaultdict(float), collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = short(r["Kernel_Name"])
    tot[k] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
print("SQ_INSTS_VALU (wave-instructions) per dispatch; 131 columns x 2^20 points per NTT pass, 2^21 leaves x 17 permutations for k_merkle_leaves")
for k in sorted(tot, key=lambda k: -tot[k])[:8]:
    print(f"{k} {len(disp[k])} dispatches {tot[k] / len(disp[k]):.4g} per dispatch")
PY
rm -rf /tmp/pk_nl && timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk_nl -- python tools/probe_netlist_perf.py > /dev/null 2>&1
cp "$(ls /tmp/pk_nl/*/*kernel_stats.csv | head -1)" "$OUT/netlist_kernel_stats.csv"
# 5b. HBM counters of the netlist probe (separate passes per counter): bytes per dispatch of 8 instances
: > "$OUT/netlist_pmc.txt"
for C in WRITE_SIZE FETCH_SIZE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR; do
    rm -rf /tmp/pn && timeout -s KILL 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pn -- python tools/probe_netlist_perf.py > /dev/null 2>&1
    python3 - "$(ls /tmp/pn/*/*counter_collection.csv | head -1)" $C >> "$OUT/netlist_pmc.txt" <<'PY'
import collections, csv, os, sys
sys.path.insert(0, os.path.join(os.environ.get("ZKW_ROOT", "."), "tools"))
from kernel_names import short  # the k_single / k_multi forms of a kernel body by the name the kernel had (tools/kernel_names.py)
tot, disp = collections.defaultdict(float), collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = short(r["Kernel_Name"])
    tot[k] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
# the counters are in KiB; FETCH_SIZE x 2 on gfx950 (MI355X_MICROARCH.md, HBM section: the same corrections as traffic.json)
scale = 1024 * (2 if sys.argv[2] == "FETCH_SIZE" else 1)
for k in sorted(tot, key=lambda k: -tot[k])[:8]:
    if sys.argv[2].startswith("SQ_"):  # instruction counts (wave-instructions), all XCDs
        print(f"{sys.argv[2]} {k} {len(disp[k])} dispatches {tot[k] / len(disp[k]):.4g} wave-instructions per dispatch")
    else:
        print(f"{sys.argv[2]}{' x2' if scale > 1024 else ''} {k} {len(disp[k])} dispatches {tot[k] / len(disp[k]) * scale / 1e9:.3g} GB per dispatch")
PY
done
# 5c. the VALU ceiling: instruction classes at 1-8 waves per SIMD, the Goldilocks mixes counted by SQ_INSTS_VALU (class
#     indices 32 gl::mul, 33 lane-form Poseidon2, 34 quad-form Poseidon2 of tools/ubench_valu_ceiling.hip), the multiplication variants
if [ -x tools/ubench_valu_ceiling ]; then
    timeout -s KILL 600 tools/ubench_valu_ceiling > "$OUT/valu_ceiling_raw.json" 2> /dev/null
    for m in 32 33 34; do
        rm -rf /tmp/pvc_$m && (cd /tmp && timeout -s KILL 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU --output-format csv -d /tmp/pvc_$m -- "$ROOT/tools/ubench_valu_ceiling" $m > /dev/null 2>&1)
        cp "$(ls /tmp/pvc_$m/*/*counter_collection.csv | head -1)" "$OUT/valu_ceiling_pmc_$m.csv"
    done
fi
[ -x tools/ubench_glmul ] && timeout -s KILL 300 tools/ubench_glmul > "$OUT/glmul_variants.json" 2> "$OUT/glmul_variants_checks.txt"
# 6. the hardware probes behind DESIGN.md 3.2
(cd tools && for b in probe_wave_placement probe_clock_regime ubench_perm; do [ -x ./$b ] && { echo "== $b"; timeout -s KILL 300 ./$b; }; done) > "$OUT/hardware_probes.txt" 2>&1
(cd tools && for b in probe_hw_queues2; do [ -x ./$b ] && { echo "== $b (default environment)"; timeout -s KILL 120 ./$b; echo "== $b (GPU_MAX_HW_QUEUES=8)"; GPU_MAX_HW_QUEUES=8 timeout -s KILL 120 ./$b; }; done) > "$OUT/hw_queue_probes.txt" 2>&1
# 7. K production-capacity blocks in flight at once (zkw_blocks_run + zkw_blocks_synthesize + zkw_blocks_free), batch after batch; the builders'
#    timeline of 512 blocks (one line per flush of the batch); round 5's schedule (a thread per block, the chain service) on the same box
for K in 1 64 128 256 512; do timeout -s KILL 600 python tools/probe_blocks_pipeline.py $K 3 seq device 2>&1 | grep "^K=" | tail -1; done > "$OUT/blocks_in_flight.txt"
echo "two batches in flight (the builders of one under the synthesis of the other), 6 batches:" >> "$OUT/blocks_in_flight.txt"
timeout -s KILL 600 python tools/probe_blocks_pipeline.py 256 6 overlap device 2>&1 | grep "^K=" | tail -1 >> "$OUT/blocks_in_flight.txt"
echo "round 5's schedule (ZKW_BLOCKS_THREADS=1: a host thread per block and branch, chains through the chain service), 96 in flight:" >> "$OUT/blocks_in_flight.txt"
ZKW_BLOCKS_THREADS=1 timeout -s KILL 600 python tools/probe_blocks_pipeline.py 96 3 seq host 2>&1 | grep "^K=" | tail -1 >> "$OUT/blocks_in_flight.txt"
ZKW_BATCH_LOG=2 timeout -s KILL 300 python tools/probe_blocks_builders_trace.py 512 2>&1 | grep -E "zkw batch|builders of" > "$OUT/builders_timeline_512.txt"
# 8. the order of the legs: the full-block legs BEFORE the timed region against the default order, same box
for ORD in default first; do
    if [ $ORD = first ]; then export ZKW_FULL_BLOCK_FIRST=1; else unset ZKW_FULL_BLOCK_FIRST; fi
    timeout -s KILL 900 python bench.py --steps 10 --no-cpu-baseline --no-sensitivity --no-h2d --no-hash-circuits 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('full-block legs $ORD: value %.0f circuits/s, %.0f ms per step, full_block.batched %.1f blocks/s' % (d['value'], d['ms_per_step'], d['full_block']['batched']['blocks_per_s']))"
done > "$OUT/order_of_legs.txt" 2>&1
unset ZKW_FULL_BLOCK_FIRST
python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3 > "$OUT/gpu_tests_tail.txt"
ls -la "$OUT"
