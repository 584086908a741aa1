"""Regenerates profiles/<round>/README.md and traffic.json from the CSV / JSON / text files next to them (numbers are never
typed by hand). Usage: python tools/make_profile_readme.py r02"""
import csv
import re
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_names import short  # k_single / k_multi forms of a kernel body by the name the kernel had before round 6

ROUND = sys.argv[1] if len(sys.argv) > 1 else "r02"
D = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles", ROUND)


def last_json(name):
    return json.loads(open(os.path.join(D, name)).read().strip().splitlines()[-1])


def pmc(name):
    out = {}
    for r in csv.DictReader(open(os.path.join(D, name))):
        out[r["kernel"]] = (int(r["dispatches"]), float(r["counter_per_dispatch"]) * 1024)  # the counters are in KiB
    return out


rows = list(csv.DictReader(open(os.path.join(D, "bench_default_kernel_stats.csv"))))
for r in rows:
    r["Name"] = short(r["Name"])
bench = last_json("bench_default.json")
under = last_json("bench_default_under_rocprofv3.json")
seq = last_json("bench_sequential.json")
pmc_run = last_json("bench_pmc_WRITE_SIZE.json")
W, F = pmc("bench_pmc_WRITE_SIZE.summary.csv"), pmc("bench_pmc_FETCH_SIZE.summary.csv")

# algorithmic bytes per launch at the benchmarked batch (sequential form: one builder launch over all blocks, `instances_per_synthesis_launch`
# instances per synthesis launch), DESIGN.md 3.1
B = pmc_run["config"]["blocks_per_gpu"]
n = pmc_run["config"]["queries_per_block"]
stride, n_rows, inst = (n + 63) // 64 * 64, 1 << 20, pmc_run["config"].get("instances_per_synthesis_launch", 16)
# cells a region's fill stores per cycle in a slot that already holds the layout (all but the first launches of a ring slot): the slots
# its row type uses + its lookup cells, from the generated spec; the cells that are zero in every trace are not rewritten
_spec = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "include", "zkw_ram_circuit_spec.h")).read()
def _spec_list(name):
    return [int(x) for x in re.search(r"#define %s \{([^}]*)\}" % name, _spec).group(1).split(",")]
_used = [a + b for a, b in zip(_spec_list("RC_ROW_NUM_SLOTS_INIT"), _spec_list("RC_ROW_NUM_LOOKUPS_INIT"))]
alg_w = {"zkw::k_ram_fill_poseidon<0>": _used[0], "zkw::k_ram_fill_poseidon<1>": _used[1], "zkw::k_ram_fill_A": _used[2],
         "zkw::k_ram_fill_B": _used[3], "zkw::k_ram_fill_C": _used[4], "zkw::k_ram_fill_D": _used[5]}
alg_w = {k: v * n * 8 * inst for k, v in alg_w.items()}
# rows 0..255 of the multiplicity column; its other rows and the zero padding below the boundary rows are only written into a slot that held
# another layout (slot layout tags): in this run that is the FIRST launch (16 fresh slots), averaged in over the run's launches
_tail_warm = 256 * 8 * inst
_tail_cold = ((148 * (n_rows - (6 * stride + 40)) + n_rows) * 8) * inst
_tail_n = W.get("zkw::k_ram_fill_tail", (1, 0))[0] or 1
alg_w["zkw::k_ram_fill_tail"] = (_tail_warm * (_tail_n - 1) + _tail_cold) / _tail_n
chain = next((k for k in W if k.startswith("zkw::k_chain_full")), None)
alg_r = {}
if chain:
    alg_w[chain] = 2 * B * n * 32   # 4 capacity words per item, both queues of every block
    alg_r[chain] = 2 * B * n * 48   # the 48-byte query per item (the sorted side through the permutation: + 4 B index)

traffic = {}
for k in alg_w:
    if k not in W:
        continue
    w_b, f_raw = W[k][1], F.get(k, (0, 0.0))[1]
    traffic[k] = {"dispatches": W[k][0], "write_bytes_per_launch": w_b, "fetch_bytes_per_launch_raw": f_raw,
                  # MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE tallies 128-byte requests of wide coalesced reads at 64 B
                  "fetch_bytes_per_launch_corrected": 2 * f_raw,
                  "algorithmic_write_bytes": alg_w[k], "algorithmic_read_bytes": alg_r.get(k),
                  "traffic_bytes_per_launch": w_b + 2 * f_raw}
json.dump({"source": f"rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE, separate passes, python bench.py --pipelines 1 --steps 1 --warmup 0 "
                     f"({B} blocks: the benchmarked batch)", "blocks": B, "kernels": traffic},
          open(os.path.join(D, "traffic.json"), "w"), indent=1)

# VALU instruction volume (SQ_INSTS_VALU: wave-instructions) per unit of work, for bench.py's roofline_valu
valu = {}
if os.path.exists(os.path.join(D, "bench_pmc_SQ_INSTS_VALU.summary.csv")):
    V = {r["kernel"]: (int(r["dispatches"]), float(r["counter_per_dispatch"])) for r in csv.DictReader(open(os.path.join(D, "bench_pmc_SQ_INSTS_VALU.summary.csv")))}
    busy = {}
    if os.path.exists(os.path.join(D, "bench_pmc_SQ_BUSY_CYCLES.summary.csv")):
        busy = {r["kernel"]: float(r["counter_per_dispatch"]) for r in csv.DictReader(open(os.path.join(D, "bench_pmc_SQ_BUSY_CYCLES.summary.csv")))}
    units = {}
    if chain:
        units[chain] = 2 * B * n            # queue items of the one builder launch
    for k in V:
        if k.startswith("zkw::k_ram_fill"):
            units[k] = inst                 # instances per synthesis launch
    merged = {}
    for k, u in units.items():
        if k not in V:
            continue
        base = k.split("<")[0]
        e = merged.setdefault(base, {"dispatches": 0, "wave_insts_per_unit": 0.0, "sq_busy_cycles_per_dispatch": None})
        e["dispatches"] += V[k][0]
        e["wave_insts_per_unit"] += V[k][1] / u   # (the two template instances of k_ram_fill_poseidon add up to one instance's rows)
        if k in busy:
            e["sq_busy_cycles_per_dispatch"] = busy[k]
    valu = merged
    json.dump({"source": f"rocprofv3 --pmc SQ_INSTS_VALU (and SQ_BUSY_CYCLES, separate pass), python bench.py --pipelines 1 --steps 1 --warmup 0 ({B} blocks)",
               "unit": "wave-instructions per queue item (chains) / per instance (fills)", "blocks": B, "kernels": merged}, open(os.path.join(D, "valu.json"), "w"), indent=1)

o = []
w = o.append
w(f"# rocprofv3 summaries, round {ROUND[1:].lstrip('0')} (MI355X, gfx950, ROCm 7.2) — generated by tools/make_profile_readme.py {ROUND}\n")
w("Commands (`tools/run_round_profiles.sh` runs all of them on the GPU box; `cd /tmp && export TMPDIR=/tmp` first):\n")
w("```")
w("python bench.py                                                                                   # bench_default.json")
w("python bench.py --pipelines 1 --no-cpu-baseline --no-full-block                                   # bench_sequential.json")
w("rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --no-cpu-baseline         # bench_default_kernel_stats.csv, bench_default_under_rocprofv3.json")
w("rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -- python bench.py --pipelines 1 --steps 1 --warmup 0 --no-cpu-baseline --no-full-block   # bench_pmc_WRITE_SIZE.*")
w("rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -- python bench.py --pipelines 1 --steps 1 --warmup 0 --no-cpu-baseline --no-full-block   # bench_pmc_FETCH_SIZE.*")
w("ZKW_BLOCK_PROFILE=1 python tools/probe_block.py                                                   # full_block_probe.txt")
w("```\n")
fb = bench.get("full_block", {})
w(f"## Headline numbers (bench_default.json)\n")
w(f"* throughput leg: **{bench['value']:.1f} circuits/s**, {bench['ms_per_step']:.0f} ms per step, {bench['config']['blocks_per_gpu']} blocks "
  f"in {bench['config']['pipelines_per_gpu']} pipelines ({bench['hbm_used_GB']:.0f} GB of HBM); sequential form {seq['value']:.1f} circuits/s, "
  f"{seq['ms_per_step']:.0f} ms per step; under the profiler {under['value']:.1f} circuits/s.")
if fb:
    w(f"* full-block leg: **{fb['wall_ms']:.1f} ms** for one production-capacity block (builders {fb['builders_ms']:.1f} ms + synthesis of "
      f"{fb['instances_synthesized']} instances {fb['synthesis_ms']:.1f} ms); the oracle on the same host: {fb['cpu']['wall_ms']:.0f} ms "
      f"(builders {fb['cpu']['builders_ms']:.0f} ms on 1 thread + synthesis {fb['cpu']['synthesis_ms']:.0f} ms on {fb['cpu']['cores']} threads) = "
      f"{fb['speedup_vs_cpu']:.2f}x.")
    if fb.get("batched"):
        bt = fb["batched"]
        if "builders_ms_per_batch" in bt:
            w(f"* batched full-block leg: **{bt['blocks_per_s']:.1f} blocks/s** = {bt['synthesized_circuits_per_s']:.0f} circuits/s over {bt['batches']} batches of "
              f"{bt['blocks_per_gpu_in_flight']} blocks in flight (batch after batch): builders {bt['builders_ms_per_batch']} ms, synthesis {bt['synthesis_ms_per_batch']} ms, "
              f"release {bt['release_ms_per_batch']} ms per batch.")
        else:
            w(f"* batched full-block leg: **{bt['blocks_per_s']:.1f} blocks/s** with {bt['blocks']} blocks in flight (builders {bt['builders_ms']:.0f} ms, "
              f"synthesis of {bt['instances_synthesized']} instances {bt['synthesis_ms']:.0f} ms, release {bt['release_ms']:.0f} ms).")
    if bench.get("hash_circuits"):
        w("* netlist circuits at the reference geometry (2^20 rows; circuits/s into slots that already hold the layout; cold rates in the JSON): " + ", ".join(
            f"{k} {v['circuits_per_s']:.0f} (capacity {v['capacity']})" for k, v in bench["hash_circuits"].items()) + "."
          + ("".join(f" ECRecover at 32 instances per call: {e['at_32_instances_per_call']['circuits_per_s']:.0f}" + (f", at 64: {e['at_64_instances_per_call']['circuits_per_s']:.0f}" if e.get("at_64_instances_per_call") else "")
                     + (", with " + ", ".join(f"{n_} calls in flight {x['circuits_per_s']:.0f}" for n_, x in e["calls_in_flight"].items() if isinstance(x, dict)) if e.get("calls_in_flight") else "") + "."
                     for e in [bench["hash_circuits"].get("ecrecover", {})] if e.get("at_32_instances_per_call"))))
    w("* spans of the builders (ms): " + ", ".join(f"{k} {v:.1f}" for k, v in sorted(fb["spans_ms"].items(), key=lambda kv: -kv[1])[:8]) + ".")
cb = bench.get("cpu_baseline")
if cb:
    w(f"* cpu_baseline (throughput workload): {cb['value']:.2f} circuits/s on {cb['cores']} threads, {cb['single_thread_value']:.2f} on one ({cb['sample']}).")
w(f"\n## Kernel time, default bench ({bench['config']['blocks_per_gpu']} blocks, {bench['config'].get('pipelines_per_gpu', 1)} pipelines; "
  f"{under['warmup']} warm-up + {under['steps']} timed steps + the full-block leg)\n")
w("| kernel | calls | total ms | avg us | % |")
w("|---|---|---|---|---|")
for r in rows[:22]:
    w(f"| `{r['Name'][:72]}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.1f} | {float(r['AverageNs'])/1e3:.1f} | {r['Percentage']} |")
ch = next(r for r in rows if "k_chain_full_q4" in r["Name"])
w(f"\nbench.py's own HIP-event average of the dominant kernel against the trace: k_chain_full_q4 {bench['roofline']['avg_launch_ms']:.1f} ms per "
  f"launch without the profiler, {under['roofline']['avg_launch_ms']:.1f} ms in the profiled run, {float(ch['AverageNs'])/1e6:.1f} ms average in the "
  f"trace (which also holds the short launches of the full-block leg). In the default run the two pipelines' kernels overlap, so "
  f"per-kernel times sum to more than the wall time; `bench_sequential.json` adds up to its step (k_chain_full_q4 "
  f"{seq['kernels_ms_per_step'].get('k_chain_full_q4', 0):.0f} ms of {seq['ms_per_step']:.0f}).\n")
w(f"## HBM traffic per launch at the benchmarked batch ({B} blocks, sequential form; counter value x 1024 B) — traffic.json\n")
w("| kernel | launches | WRITE_SIZE GB | algorithmic write GB | ratio | FETCH_SIZE GB raw | x2 (gfx950 correction) | algorithmic read GB |")
w("|---|---|---|---|---|---|---|---|")
for k, t in traffic.items():
    ar = f"{t['algorithmic_read_bytes']/1e9:.3f}" if t["algorithmic_read_bytes"] else ""
    w(f"| `{k}` | {t['dispatches']} | {t['write_bytes_per_launch']/1e9:.3f} | {t['algorithmic_write_bytes']/1e9:.3f} | "
      f"{t['write_bytes_per_launch']/t['algorithmic_write_bytes']:.3f} | {t['fetch_bytes_per_launch_raw']/1e9:.3f} | "
      f"{t['fetch_bytes_per_launch_corrected']/1e9:.3f} | {ar} |")
w("\nWRITE_SIZE x 1024 B equals the algorithmic bytes of every fill kernel to about 1 % at the benchmarked batch as well (round 1 measured it at 32 "
  "blocks): no wasted partial lines, no re-writes. The chain kernel's reads are the one place where traffic exceeds the algorithmic "
  "bytes: half of its 48-byte reads are random gathers through the sorting permutation (a 128-byte line fetched per 48 bytes used) "
  "and its ~29 k sequential streams advance 48 bytes per ~22 us step, too slowly for a line to survive in L2 until its next use; at "
  "64 chains (round 1) raw FETCH_SIZE x 2 equalled the algorithmic bytes exactly, so the x2 calibration of MI355X_MICROARCH.md holds "
  "for this access width and the excess here is real re-fetching. It costs nothing: the kernel is latency-bound at ~1 % of the HBM "
  "rate (DESIGN.md 3.2). The fills read 8-48 B per lane from arrays the preceding kernels wrote.")
if valu:
    w("\n## VALU instruction volume (SQ_INSTS_VALU, wave-instructions) per unit of work — valu.json\n")
    w("| kernel | dispatches | wave-instructions per unit | unit |")
    w("|---|---|---|---|")
    for k, e in valu.items():
        w(f"| `{k}` | {e['dispatches']} | {e['wave_insts_per_unit']:.1f} | {'queue item' if 'chain' in k else 'instance (2^20 rows)'} |")
    w("\nPeak issue: 256 CUs x 4 SIMDs x 2.4 GHz / 4 cycles per wave-instruction = 6.14e11 wave-instructions/s; bench.py's `roofline_valu` divides.")
for name, title in (("full_block_probe.txt", "One production-capacity block (tools/probe_block.py, ZKW_BLOCK_PROFILE=1: per-kernel HIP-event times of every branch's context, then the spans)"),
                    ("synthesis_probes.txt", "Synthesis of the other circuit types at production geometry (tools/probe_{ds,es,ld,ss}_synth.py, 16 instances per pass; tools/probe_netlist_perf.py)"),
                    ("ecrecover_steps.txt", "ECRecover synthesis at 32 instances per call through round 6's steps (tools/probe_ecrecover_synth.py on the box after each step; per-kernel HIP-event times in ms; docs/KERNELS.md 3.19)"),
                    ("ecrecover_kernel_stats.csv", "rocprofv3 --kernel-trace --stats of tools/probe_ecrecover_synth.py (ECRecover at production geometry: 4 calls of 8 and 4 calls of 32 instances + the builder and two checks)"),
                    ("blocks_in_flight.txt", "K production-capacity blocks in flight at once (tools/probe_blocks_pipeline.py K 3 seq device: zkw_blocks_run + zkw_blocks_synthesize + zkw_blocks_free, batch after batch; the summary line of 3 batches)"),
                    ("blocks_kernel_stats.csv", "rocprofv3 --kernel-trace --stats of the block leg alone (tools/probe_blocks_pipeline.py 512 2 seq device: warm-up batch + 2 batches of 512 blocks; ` [merged]` = the k_multi form)"),
                    ("builders_timeline_512.txt", "The builders of 512 blocks, one line per flush of the batch (ZKW_BATCH_LOG=2 tools/probe_blocks_builders_trace.py 512; second run)"),
                    ("order_of_legs.txt", "The order of the legs (ZKW_FULL_BLOCK_FIRST=1 against the default), 10 timed steps each on the same box"),
                    ("rocprof_stats_runs.txt", "Did rocprofv3 --kernel-trace --stats run through the whole bench, 512 blocks in flight in the batched leg included?"),
                    ("hardware_probes.txt", "Hardware probes behind DESIGN.md 3.2 (tools/probe_wave_placement, probe_clock_regime, ubench_perm)"),
                    ("hw_queue_probes.txt", "Concurrent long kernels per priority class against GPU_MAX_HW_QUEUES (tools/probe_hw_queues2, DESIGN.md 3.14)"),
                    ("netlist_kc_kernel_stats.csv", "rocprofv3 --kernel-trace --stats of tools/probe_kc_synth.py (Keccak256RoundFunction, 3 x 8 instances + the builder)"),
                    ("netlist_sc_kernel_stats.csv", "rocprofv3 --kernel-trace --stats of tools/probe_sc_synth.py (Sha256RoundFunction, 3 x 8 instances + the builder)"),
                    ("netlist_kernel_stats.csv", "rocprofv3 --kernel-trace --stats of tools/probe_netlist_perf.py (netlist circuits at production geometry: Keccak256RoundFunction and Sha256RoundFunction 4 x 8 instances, L1MessagesHasher 2 x 1 + 2 x 8 queues, + the builders)"),
                    ("netlist_pmc.txt", "rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE / SQ_INSTS_VALU / SQ_INSTS_LDS / SQ_INSTS_VMEM_WR (separate passes) of the netlist probe: HBM bytes and wave-instructions per dispatch (k_nl_fill: 8 instances per dispatch, mixed over the circuits)"),
                    ("netlist_fill_split.txt", "Where the netlist fill's time went (builds with one phase compiled out) and the steps of moving the multiplicity counting out of it"),
                    ("gpu_tests_tail.txt", "pytest tests -m gpu on the same box")):
    path = os.path.join(D, name)
    if os.path.exists(path):
        w(f"\n## {title}\n")
        w("```")
        txt = open(path).read().rstrip().splitlines()
        if name.endswith("_kernel_stats.csv"):
            hdr, body = txt[0], txt[1:26 if name.startswith("blocks") else 16 if name.startswith("ecrecover") else 12]
            txt = [hdr] + ['"%s",%s' % (short(l.split('",')[0].lstrip('"')), l.split('",', 1)[1][:60]) if '",' in l else l[:170] for l in body]
        if name == "builders_timeline_512.txt":
            txt = txt[-80:]
        if name == "full_block_probe.txt":  # keep the last repetition's kernel table and the spans
            keep = [l for l in txt if not l.startswith("[zkw_block]")]
            tab = [l for l in txt if l.startswith("[zkw_block]")]
            per = len(tab) // 4 if len(tab) >= 4 else len(tab)
            txt = tab[-per:] + keep
        w("\n".join(txt))
        w("```")
open(os.path.join(D, "README.md"), "w").write("\n".join(o) + "\n")
print("\n".join(o)[:2500])
