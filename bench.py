#!/usr/bin/env python
"""bench.py — base-layer circuits/s for the RAM-permutation hot path on MI355X.

One "step" = one pass of the hot path over one batch: B independent blocks' memory queues (each at the
production RAMPermutation capacity, 136 714 queries = one 2^20-row instance) go through witness
generation (encode, sort, both Poseidon2 queue chains, Fiat-Shamir challenges, grand products, instance
records) and synthesis (every instance materialised into a full 149-column x 2^20-row trace, written
into a ring of trace buffers). Inputs are
resident in HBM before the timed region. N > 1: one process per GPU, blocks sharded with no data-path
collective; the per-instance closed-form records are gathered to rank 0 (RCCL) inside the timed region.

Prints ONE JSON line (rank 0). See DESIGN.md "Measurement" for the roofline/cpu_baseline definitions.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from era_zkevm_test_harness_amd import native, parallel, synthetic  # noqa: E402

CAPACITY = 136714  # cycles_per_ram_permutation, circuit_sequencer_api/src/geometry_config.rs:12
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def make_inputs(blocks, n, rank, dev):
    """B valid memory traces in HBM: one seeded base trace, each block's values XOR-ed with its own
    256-bit mask (stays a valid memory: a read still returns the last written value of its cell)."""
    base = synthetic.ram_trace(n, seed=2 + rank)
    qb = torch.from_numpy(base.view(np.int32).reshape(n, 12).copy()).to(dev)
    q = qb.unsqueeze(0).repeat(blocks, 1, 1)
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    masks = torch.randint(-2**31, 2**31 - 1, (blocks, 1, 8), generator=g, dtype=torch.int64).to(torch.int32).to(dev)
    q[:, :, 4:12] ^= masks
    return base, q.contiguous()


def cpu_baseline(base, sample_blocks):
    """The oracle (CPU restatement of the reference algorithm) on a bounded sample of the same workload: independent
    blocks on up to 8 host threads (the reference's Worker::new_with_num_threads(8), complex_tests/mod.rs:303; the
    builders and synthesis of one instance are single-threaded there too), plus the single-thread rate."""
    from concurrent.futures import ThreadPoolExecutor

    from oracle import pyoracle

    pyoracle.build()
    pyoracle.ram_build_instances(base[:2048], 2048, 0)  # warm

    def one(_):
        o = pyoracle.ram_build_instances(base, CAPACITY, 0)  # ctypes releases the GIL inside the C calls
        pyoracle.ram_synthesize(o, 0, CAPACITY, 1 << 20)

    t0 = time.perf_counter()
    one(0)
    one(0)
    dt1 = time.perf_counter() - t0
    threads = max(1, min(os.cpu_count() or 1, 8))
    per_thread = max(1, sample_blocks // 2)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(one, range(threads * per_thread)))
    dt = time.perf_counter() - t0
    n_done = threads * per_thread
    return {"value": n_done / dt, "unit": "circuits/s", "cores": threads, "kind": "port",
            "single_thread_value": 2 / dt1,
            "sample": f"{n_done} RAMPermutation instances of {CAPACITY} queries on {threads} threads ({dt:.1f} s) after 2 on one "
                      f"thread ({dt1:.1f} s): witness generation + synthesis of the 2^20-row trace, oracle/liboracle.so"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--blocks", type=int, default=0,
                    help="independent memory queues per GPU per step (0 = size the batch to the free HBM)")
    ap.add_argument("--queries", type=int, default=CAPACITY)
    ap.add_argument("--ring", type=int, default=16, help="trace buffers (1.25 GB each) in the output ring")
    ap.add_argument("--cpu-sample", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank, local_rank, world = parallel.init_from_env()
    assert world == args.gpus or world == 1, f"WORLD_SIZE={world} but --gpus {args.gpus}"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    ctx = native.Context(local_rank)  # raises if libzkw / the GPU is missing: no fallback
    stream = torch.cuda.Stream(device=dev)  # one stream for torch ops, RCCL and libzkw kernels
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)
    ctx.set_pointer_mode(native.PTR_DEVICE)
    if os.environ.get("ZKW_CHAIN_FORM"):
        ctx.set_chain_form(int(os.environ["ZKW_CHAIN_FORM"]))

    n = args.queries
    B = args.blocks
    if B <= 0:
        # one block of n queries keeps ~74 B/query... measured 69 MB per 136 714-query block (inputs 6.6 + witness
        # 54.7 + sort scratch 5.5 + chain/descriptor scratch); the chains want as many concurrent queues as fit
        free, _total = torch.cuda.mem_get_info(dev)
        # resident per query: input 48 + sorted copy 48 + two encodings 2 x 64 + capacity words of the tails 2 x 32 +
        # grand products 2 x 16 = 320 bytes (sort scratch aliases arrays that are filled later). Cap: 4096 blocks =
        # 8192 queue chains. Inside the builder the chain kernel runs at 14.7 us per step up to ~8 400 chains and
        # at 22.3 us beyond (measured cliff), so more blocks per step only pay above ~6 000 blocks,
        # which do not fit the HBM.
        per_block = int(n * 330)
        B = int(max(16, min(4096, (0.90 * free - args.ring * 1.25e9) // per_block)))
        B = parallel.min_over_ranks(B, dev)  # every rank runs the same batch (weak scaling, equal record counts)
    n_rows = 1 << 20  # TARGET_CIRCUIT_TRACE_LENGTH, base_layer/mod.rs:17
    ring = native.Trace(ctx, n_rows, args.ring)  # trace buffers a prover would consume and hand back
    base, q = make_inputs(B, n, rank, dev)
    offs = np.arange(B + 1, dtype=np.uint64) * n
    w = native.RamWitness(ctx)
    # what rank 0 needs from every instance to replay the recursion queue and build the scheduler witness
    # (SURVEY 8(e)): compact closed-form input (2 flags + 4 commitments = 18 words) + public input (4 words) = 176 B
    inst_bytes = (18 + 4) * 8
    n_inst_local = B * (-(-n // CAPACITY))
    records = torch.empty((n_inst_local, inst_bytes), dtype=torch.uint8, device=dev)
    compact = torch.empty((n_inst_local, 18), dtype=torch.int64, device=dev)
    pis = torch.empty((n_inst_local, 4), dtype=torch.int64, device=dev)
    counts = [n_inst_local] * world

    def step():
        ctx.compute_ram_circuit_snapshots((q.data_ptr(), B * n), CAPACITY, 0, block_offsets=offs, witness=w)
        for first in range(0, n_inst_local, args.ring):  # synthesis: every instance -> a full 2^20-row trace
            ctx.synthesize_ram(w, ring, first, min(args.ring, n_inst_local - first), 0)
        lib = native.load()
        native._check(lib.zkw_ram_witness_get(w.handle, native.RAM_COMPACT_FORMS, compact.data_ptr(), compact.numel() * 8))
        native._check(lib.zkw_ram_witness_get(w.handle, native.RAM_PUBLIC_INPUTS, pis.data_ptr(), pis.numel() * 8))
        records.view(torch.int64).view(n_inst_local, 22)[:, :18] = compact
        records.view(torch.int64).view(n_inst_local, 22)[:, 18:] = pis
        return parallel.gather_records(records, counts, dst=0)

    for _ in range(args.warmup):
        step()
    ctx.profile_enable(True)
    ctx.profile_reset()
    parallel.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        gathered = step()
    torch.cuda.synchronize()
    parallel.barrier()
    dt = parallel.max_over_ranks(time.perf_counter() - t0, dev)
    prof = ctx.profile()
    ctx.profile_enable(False)

    if rank == 0:
        assert gathered.shape[0] == n_inst_local * world
        circuits = n_inst_local * world * args.steps
        # dominant kernel by time, with its algorithmic HBM bytes per launch (DESIGN.md "Measurement")
        items = B * n
        # algorithmic HBM bytes PER LAUNCH of each kernel (DESIGN.md "Measurement"): cells written once +
        # the instance's witness read once. A synthesis launch covers `args.ring` instances.
        per_launch_inst = min(args.ring, n_inst_local)
        stride = (CAPACITY + 63) // 64 * 64            # rows per region incl. the alignment gap (zkw trace v2)
        cell = 8 * stride * per_launch_inst            # one column of one region, all instances of a launch
        wit = per_launch_inst * n                      # witness items read by a launch
        alg_bytes = {
            "k_chain_full": 2 * items * (64 + 32),            # per chain item: 8 words in, 4 capacity words out
            "k_chain_full_q4": 2 * items * (64 + 32),
            "k_gp_local": 2 * items * (64 + 16),              # rows read once, both repetitions written
            "k_gp_apply": 2 * items * 32,
            "k_encode_mem": items * (48 + 64),
            "k_gather_encode": items * (48 + 4 + 48 + 64),
            "k_ram_fill_poseidon": 148 * cell + wit * (64 + 48 + 32),   # one Poseidon2 region per launch
            "k_ram_fill_A": 148 * cell + wit * (64 + 48 + 32),
            "k_ram_fill_B": 148 * cell + wit * 96,
            "k_ram_fill_C": 148 * cell + wit * 96,
            "k_ram_fill_D": 148 * cell + 48 * cell,           # reads the queue tails back from the Poseidon2 rows
            "k_ram_fill_tail": per_launch_inst * 8 * (148 * (n_rows - 6 * stride) + n_rows),
        }
        name, (ms, cnt) = max(prof.items(), key=lambda kv: kv[1][0])
        avg_ms = ms / max(cnt, 1)
        ab = alg_bytes.get(name)
        achieved = (ab / (avg_ms * 1e-3) / 1e9) if ab else None
        hbm_kernels = {}
        for k, (kms, kcnt) in prof.items():
            if k.startswith("k_ram_fill") and k in alg_bytes and kcnt:
                gbs = alg_bytes[k] / (kms / kcnt * 1e-3) / 1e9
                hbm_kernels[k] = {"achieved_GBps": gbs, "frac": gbs / HBM_PEAK_GBS, "avg_launch_ms": kms / kcnt}
        synth_ms = sum(v[0] for k, v in prof.items() if k.startswith("k_ram_fill") or k.startswith("k_ram_nd"))
        synth_gbs = (149 * n_rows * 8 * n_inst_local * args.steps) / (synth_ms * 1e-3) / 1e9 if synth_ms else None
        chain_ms, chain_cnt = prof.get("k_chain_full", prof.get("k_chain_full_q4", (0.0, 1)))
        free_after, total_mem = torch.cuda.mem_get_info(dev)
        out = {
            "metric": "base-layer circuits/sec (2^20 rows)",
            "value": circuits / dt,
            "unit": "circuits/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64 (Goldilocks, p = 2^64 - 2^32 + 1)",
            "data": "synthetic",
            "config": {"workload": f"RAMPermutation base circuit, capacity {CAPACITY} (2^20-row geometry), "
                                   f"{B} independent memory queues per GPU per step, witness generation + synthesis "
                                   f"of every instance into a 149-column x 2^20-row trace",
                       "blocks_per_gpu": B, "queries_per_block": n, "parallelism": f"instances sharded x{world}"},
            "roofline": {"bound": "hbm", "kernel": name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": None,
                         "avg_launch_ms": avg_ms,
                         "note": "k_chain_full is a serial Poseidon2 chain per queue: latency/VALU-bound, "
                                 "permutations/s is its meaningful rate"},
            "synthesis": {"trace_bytes_per_circuit": 149 * n_rows * 8, "kernels_ms_per_step": synth_ms / args.steps,
                          "achieved_GBps": synth_gbs, "frac_of_hbm_peak": synth_gbs / HBM_PEAK_GBS if synth_gbs else None,
                          "per_kernel": hbm_kernels},
            "hbm_used_GB": (total_mem - free_after) / 1e9,
            "poseidon2_perm_per_s": 2 * items * chain_cnt / (chain_ms * 1e-3) if chain_ms else None,
            "kernels_ms_per_step": {k: v[0] / args.steps for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])},
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(base, args.cpu_sample)
        print(json.dumps(out), flush=True)
    parallel.barrier()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
