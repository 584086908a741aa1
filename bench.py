#!/usr/bin/env python
"""bench.py — base-layer circuits/s for the RAM-permutation hot path on MI355X.

One "step" = one pass of the hot path over one batch: B independent blocks' memory queues (each at the
production RAMPermutation capacity, 136 714 queries = one 2^20-row instance) go through witness
generation (encode, sort, both Poseidon2 queue chains, Fiat-Shamir challenges, grand products, instance
records) and synthesis (every instance materialised into a full 149-column x 2^20-row trace, written
into a ring of trace buffers). Inputs are
resident in HBM before the timed region. By default the B blocks run as two pipelines (two contexts, HIP streams and host
threads over one half of the blocks each) whose synthesis phases take turns, so that one half's synthesis overlaps the
other half's queue chains; every timed step is still one pass of every block through the whole path. N > 1: one process per GPU, blocks sharded with no data-path
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


def make_comm(ctx, rank, world):
    """zkw_comm over RCCL for the C-ABI gather (include/zkw.h); the 128-byte id travels over torch.distributed, the way a
    Rust host would use its own control channel. Returns None (every rank alike) when RCCL cannot be set up inside libzkw:
    the gather then goes through torch.distributed (the same RCCL, torch's copy)."""
    if world == 1:
        return native.Comm(ctx, 0, 1)
    import torch.distributed as dist

    box = [None]
    if rank == 0:
        try:
            box[0] = native.Comm.unique_id()
        except native.ZkwError as e:
            print(f"[bench] zkw_comm_unique_id failed ({e}); gathering through torch.distributed", file=sys.stderr)
    dist.broadcast_object_list(box, src=0)
    if box[0] is None:
        return None
    ok = torch.ones(1, dtype=torch.int32, device=f"cuda:{torch.cuda.current_device()}")
    comm = None
    try:
        comm = native.Comm(ctx, rank, world, box[0])
    except native.ZkwError as e:
        print(f"[bench] rank {rank}: zkw_comm_init failed ({e})", file=sys.stderr)
        ok.zero_()
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok.item()) == 0:
        if comm is not None:
            comm.destroy()
        return None
    return comm


def full_block_gpu(local_rank, reps=3, rank=0, world=1, comm=None):
    """BASELINE.json's second figure: wall time of ONE block — the instance multiset of the reference's basic_test with
    every builder at production capacity (synthetic.block_production) — through zkw_block_run (all witness builders as a
    dependency graph inside libzkw) + zkw_block_synthesize (every instance of the synthesizable types into a 2^20-row
    trace). Inputs are host arrays, as `external_calls::run` hands them over; the storage tree answers are prepared
    before the timed region. Returns the report and the block dict (for the CPU leg)."""
    blk = synthetic.block_production(seed=1)
    first = native.Block(local_rank, blk)  # untimed: warms the library and yields the deduplicated storage queries
    dedup = first.witness_get(9, native.STO_RESULT_QUERIES, np.uint8).view(native.LOG_QUERY)
    tree, answers = synthetic.storage_tree_for(dedup, seed=1)
    root0, next0 = tree.root, tree.next_enumeration_index
    first.synthesize(1 << 20, ring_slots=2)
    first.free()
    best = None
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        B = native.Block(local_rank, blk, storage_tree=answers, storage_initial_root=root0, storage_next_enumeration_index=next0)
        t1 = time.perf_counter()
        n_synth = B.synthesize(1 << 20, ring_slots=2, rank=rank, world=world)  # this rank's LPT share of the instances
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        if comm is not None:  # the one collective: closed-form records to rank 0
            B.gather_closed_form_inputs(comm, rank, world, 0)
        t3 = time.perf_counter()
        wall = parallel.max_over_ranks(t3 - t0, torch.device("cuda", local_rank))
        rep = {"wall_ms": wall * 1e3, "builders_ms": (t1 - t0) * 1e3, "synthesis_ms": (t2 - t1) * 1e3, "gather_ms": (t3 - t2) * 1e3,
               "instances_synthesized": n_synth, "instances_in_block": 3 + sum(B.num_instances(t) for t in range(2, 14)),  # + basic_test's 3 MainVM instances (need the VM)
               "n_gpus": world, "instances": {str(t): B.num_instances(t) for t in range(2, 14)},
               "spans_ms": {name: round(e - s_, 2) for name, s_, e in B.timings() if name != "builders"},
               "memory_queue_items": B.memory_queue_length}
        B.free()
        if best is None or rep["wall_ms"] < best["wall_ms"]:
            best = rep
    best["batched"] = full_blocks_batched(local_rank, blk, rank=rank, world=world, comm=comm)
    best["scaling_note"] = ("`batched` is the leg that scales with N (whole blocks sharded over the ranks, nothing replicated); the single block above "
                            "does not: its builders are one serial chain replicated on every rank, only its ~35 ms of synthesis are shared")
    best["note"] = ("one block on %d GPU(s): builders = zkw_block_run (every builder of the post-VM half of "
                    "create_artifacts_from_tracer; replicated on every rank: they are bounded by the block's longest serial "
                    "Poseidon2 queue chain, memory queue = %d items x ~10.3 us, which more GPUs cannot shorten); synthesis = this "
                    "rank's LPT share of the synthesizable instances (six queue circuits x 1.25 GB, keccak256 round function and L1-messages hasher x 1.15 GB); gather = the closed-form records to rank 0"
                    % (world, best["memory_queue_items"]))
    return best, blk


def full_blocks_batched(local_rank, blk, K=512, rounds=3, rank=0, world=1, comm=None):
    """Throughput of WHOLE blocks: K production-capacity blocks PER GPU in flight at once, batch after batch. zkw_blocks_run runs a batch's
    builder graphs as fibers of ONE host thread and merges their launches per kernel and stage (csrc/zkw_batch.h: no thread and no stream per
    block; the chains of a stage as one launch, longest chains first); zkw_blocks_synthesize synthesizes every instance of every block — groups
    of slot-owning fibers going through their blocks type by type, the ECRecover instances of all blocks in joint calls on two priority
    streams —; zkw_blocks_free releases the batch. The figure is the blocks of the timed batches over their wall time; a first batch, which fills
    the library's buffer caches, is untimed. `two_batches_in_flight` is the same work as two batches of K / 2 with the builders of one under
    the synthesis of the other. The blocks' four queues are resident in HBM when the clock starts
    (zkw_block_inputs.queues_on_device: the contract's "inputs already resident"); `host_inputs` repeats one batch with host arrays.
    With N GPUs the blocks of every batch are sharded over the ranks by zkw_blocks_run_sharded (round-robin, nothing replicated) and every
    block's closed-form records are gathered to rank 0 (zkw_blocks_gather_closed_form_inputs), batch after batch."""
    import threading

    K = int(os.environ.get("ZKW_BATCHED_BLOCKS", K))
    Kb = max(1, K // 2)  # blocks per batch and GPU in the overlapped schedule
    distinct_host = [blk] + [synthetic.block_production(seed=2 + k) for k in range(3)]
    distinct = [native.Block.queues_to_device(b, local_rank) for b in distinct_host]
    pick = lambda pool, n: [pool[(k // world) % len(pool)] for k in range(n * world)]  # noqa: E731
    dev = torch.device("cuda", local_rank)

    def build(tpl):
        return native.Block.run_prepared(local_rank, tpl) if world == 1 else native.Block.run_sharded_prepared(local_rank, tpl, rank, world)

    def synth(bs, rep):
        t1 = time.perf_counter()
        mine = [b for b in bs if b is not None]
        rep["instances"] += native.Block.synthesize_many(mine, 1 << 20, ring_slots=1)
        t2 = time.perf_counter()
        if world > 1 and comm is not None:
            got = native.Block.gather_sharded(bs, comm, rank, world, root=0)
            rep["records"] = (rep["records"] or 0) + (sum(len(g) for g in got) if got is not None else 0)
        rep["synthesis_ms"].append((t2 - t1) * 1e3); rep["gather_ms"].append((time.perf_counter() - t2) * 1e3)
        return mine

    def release(mine, rep):
        t = time.perf_counter()
        native.Block.free_many(mine)
        rep["release_ms"].append((time.perf_counter() - t) * 1e3)

    def new_rep():
        return {"instances": 0, "records": None, "synthesis_ms": [], "gather_ms": [], "release_ms": [], "builders_ms": []}

    def in_turn(tpl, n_rounds):
        rep = new_rep()
        parallel.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _r in range(n_rounds):
            tb = time.perf_counter()
            cur = build(tpl)
            rep["builders_ms"].append((time.perf_counter() - tb) * 1e3)
            release(synth(cur, rep), rep)
        torch.cuda.synchronize()
        parallel.barrier()
        rep["wall"] = parallel.max_over_ranks(time.perf_counter() - t0, dev)
        rep["n_all"] = int(parallel.sum_over_ranks(rep["instances"], dev)) if world > 1 else rep["instances"]
        return rep

    def overlapped(tpl, n_rounds):
        rep = new_rep()
        box = {}

        def t_build():
            tb = time.perf_counter()
            box["next"] = build(tpl)
            rep["builders_ms"].append((time.perf_counter() - tb) * 1e3)

        parallel.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        t_build()
        cur, old = box.pop("next"), None
        for r in range(n_rounds):
            th = []
            if r + 1 < n_rounds:
                th.append(threading.Thread(target=t_build))
            if old is not None:
                th.append(threading.Thread(target=release, args=(old, rep)))
            for t in th:
                t.start()
            old = synth(cur, rep)  # (this thread: the collective of the N > 1 leg stays on the thread that owns the process group)
            for t in th:
                t.join()
            cur = box.pop("next", None)
        release(old, rep)
        torch.cuda.synchronize()
        parallel.barrier()
        rep["wall"] = parallel.max_over_ranks(time.perf_counter() - t0, dev)
        rep["n_all"] = int(parallel.sum_over_ranks(rep["instances"], dev)) if world > 1 else rep["instances"]
        return rep

    # the input structs, once: a service builds them as its blocks arrive
    tpl_full = native.Block.prepare_many(local_rank, pick(distinct, 2 * Kb))
    in_turn(tpl_full, 1)  # fills the caches
    it = in_turn(tpl_full, rounds)
    free_b, total_b = torch.cuda.mem_get_info(dev)
    r3 = lambda v: [round(x, 1) for x in v]  # noqa: E731
    out = {"blocks": 2 * Kb * world * rounds, "blocks_per_gpu_in_flight": 2 * Kb, "batches": rounds, "n_gpus": world,
           "blocks_per_s": 2 * Kb * world * rounds / it["wall"], "per_rank_blocks_per_s": 2 * Kb * rounds / it["wall"],
           "synthesized_circuits_per_s": it["n_all"] / it["wall"], "wall_ms": it["wall"] * 1e3,
           "builders_ms_per_batch": r3(it["builders_ms"]), "synthesis_ms_per_batch": r3(it["synthesis_ms"]), "gather_ms_per_batch": r3(it["gather_ms"]),
           "release_ms_per_batch": r3(it["release_ms"]), "instances_synthesized": it["n_all"], "records_gathered": it["records"],
           "inputs": "the blocks' four queues resident in HBM (zkw_block_inputs.queues_on_device); bytecodes and input structs on the host",
           "host_threads_per_block": 0, "streams_per_block": 0,
           "schedule": "batch after batch: builders (zkw_blocks_run: fibers of one thread, launches merged per kernel and stage, a stage's chains as one launch, longest "
                       "first), synthesis (zkw_blocks_synthesize: slot-owning fibers type by type, ECRecover in joint calls on priority streams), release (zkw_blocks_free)",
           "sharding": "one GPU" if world == 1 else "zkw_blocks_run_sharded (round-robin over ranks) + zkw_blocks_gather_closed_form_inputs",
           "rccl_ranks": world if (world > 1 and comm is not None) else 0, "hbm_in_use_GB": (total_b - free_b) / 1e9}
    try:
        # the same with the queues in host memory (PCIe inside the builders): one batch
        host_tpl = native.Block.prepare_many(local_rank, pick(distinct_host, 2 * Kb))
        h = in_turn(host_tpl, 1)
        out["host_inputs"] = {"blocks_per_s": 2 * Kb * world / h["wall"], "builders_ms": round(h["builders_ms"][0], 1), "synthesis_ms": round(h["synthesis_ms"][0], 1),
                              "note": "one batch, the four queues of every block as host arrays (~20 MB per block over PCIe inside zkw_blocks_run)"}
        del host_tpl, tpl_full
        # two batches of K / 2 in flight: the builders of batch k + 1 under the synthesis of batch k, the release of batch k - 1 on a third thread.
        # Faster on average and less even (round 6, eight runs of 8 batches: 113 - 150 blocks/s against 136 - 138 for one batch of K at a time —
        # the chains' priority waves and the fills share SIMDs, and a step is as long as the slower of the two): reported next to the headline figure
        tpl_half = native.Block.prepare_many(local_rank, pick(distinct, Kb))
        ov = overlapped(tpl_half, 8)
        out["two_batches_in_flight"] = {"blocks_per_batch_and_gpu": Kb, "batches": 8, "blocks_per_s": Kb * world * 8 / ov["wall"],
                                        "builders_ms_per_batch": r3(ov["builders_ms"]), "synthesis_ms_per_batch": r3(ov["synthesis_ms"]),
                                        "release_ms_per_batch": r3(ov["release_ms"]),
                                        "note": "start-up (the first batch's builders have nothing to hide under) and drain inside the timed region"}
    except Exception as e:  # noqa: BLE001 - side figures
        out["side_legs_error"] = repr(e)
    return out



def setup_commit_gpu(local_rank):
    """The setup side as field elements (create_base_layer_setup_data, src/prover_utils.rs:48-197) at production size:
    131 columns x 2^20 rows (the events sorter's setup columns' shape), LDE x 2, Merkle tree with cap 16, on device-resident columns.
    Per-kernel times from the library's own events; NTT passes as GB/s over their read + write, leaf hashing as permutations/s. Not part
    of `value`."""
    import ctypes as C_
    ctx = native.Context(local_rank)
    ctx.set_pointer_mode(native.PTR_DEVICE)
    lib = native.load()
    log_n, n_cols, lde = 20, 131, 2
    n = 1 << log_n
    g = torch.Generator(device="cuda").manual_seed(1)
    vals = torch.randint(0, 2**62, (n_cols, n), dtype=torch.int64, device="cuda", generator=g)
    ext = torch.empty((lde, n_cols, n), dtype=torch.int64, device="cuda")
    cap = torch.empty((16, 4), dtype=torch.int64, device="cuda")

    def run():
        native._check(lib.zkw_lde(ctx.handle, vals.data_ptr(), log_n, n_cols, lde, ext.data_ptr()))
        native._check(lib.zkw_merkle_tree_with_cap(ctx.handle, ext.data_ptr(), lde, n_cols, n, 16, cap.data_ptr(), None))
    run(); ctx.synchronize()
    ctx.profile_enable(True); ctx.profile_reset()
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        run()
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / reps
    prof = {k: (v[0] / reps, v[1] // reps) for k, v in ctx.profile().items()}
    ctx.profile_enable(False)
    arr = n_cols * n * 8
    out = {"columns": n_cols, "log_n": log_n, "lde_factor": lde, "cap_size": 16, "ms_per_commit": dt * 1e3,
           "kernels_ms": {k: ms for k, (ms, c) in prof.items()},
           "ntt_pass_GBps_read_plus_write": {k: c * 2 * arr / ms / 1e6 for k, (ms, c) in prof.items() if k.startswith("k_ntt")},
           "leaf_permutations_per_s": lde * n * ((n_cols + 7) // 8) / (prof["k_merkle_leaves"][0] * 1e-3) if "k_merkle_leaves" in prof else None,
           "note": "NTT passes are VALU-bound (5 multiplications + 10 additions of a 64-bit prime field per point and pass on 32-bit ALUs: "
                   "0.86 of VALU issue peak by SQ_INSTS_VALU, profiles/r04/setup_commit_valu.txt), not HBM-bound"}
    del vals, ext, cap
    ctx.close()
    torch.cuda.empty_cache()
    return out

def hash_circuits_gpu(local_rank, blk):
    """Synthesis rates of the netlist circuits at the reference's geometry — 2^20 rows, capacities of
    geometry_config.rs — on synthetic precompile calls / the block's bytecodes, 8 (4) traces per call. Two rates per circuit:
    `circuits_per_s` with the slots re-used (a slot that already holds the circuit's layout keeps its zeros: the call writes
    `write_bytes_per_circuit`, NOT `trace_bytes`) and `cold` with the slots' layout forgotten before every call (every cell of the slot
    is written: cleared or filled). `written_GBps` = circuits_per_s x write_bytes_per_circuit, the figure to hold against the HBM peak.
    Not part of `value`."""
    ctx = native.Context(local_rank)
    n_rows = 1 << 20
    out = {}

    def timed(n, fn, reps=3, trace=None, ctype=None, cap=0):
        def run(cold):
            best = None
            for _ in range(reps):
                if cold and trace is not None:
                    for k in range(n):
                        trace.device_ptr(k)  # taking the pointer forgets the slot's layout: the next call clears it
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                fn()
                ctx.synchronize()
                dt = time.perf_counter() - t0
                best = dt if best is None or dt < best else best
            return best
        best = run(False)
        rep = {"circuits_per_s": n / best, "ms_per_call": best * 1e3, "instances_per_call": n}
        if ctype is not None:
            warm_b, cold_b = native.circuit_fill_bytes(ctype, cap, n_rows)
            c = run(True)
            rep.update(write_bytes_per_circuit=warm_b, written_GBps=n / best * warm_b / 1e9, frac_of_hbm_peak_on_bytes_written=n / best * warm_b / 8e12,
                       cold={"circuits_per_s": n / c, "ms_per_call": c * 1e3, "write_bytes_per_circuit": cold_b, "written_GBps": n / c * cold_b / 1e9})
        return rep

    mem_in = np.zeros(1, native.QUEUE_STATE12)
    for name, kind, n_req, cap, cols, synth, ctype in (("keccak256_round_function", 0, 1400, 293, native.KC_COLS, ctx.synthesize_keccak_round_function, 5),
                                                       ("sha256_round_function", 1, 6000, 2206, native.SC_COLS, ctx.synthesize_sha256_round_function, 6),
                                                       ("ecrecover", 2, 7 * 64, 7, native.EK_COLS, ctx.synthesize_ecrecover, 7)):
        req, mq = synthetic.precompile_trace(kind, n_req, seed=5, max_rounds=6)
        tails = ctx.queue_push_chain_log(ctx.encode_log_queries(req))[1]
        w = ctx._precompile(kind, req, tails, mq, cap, mem_in)
        n = min(8, w.num_instances)
        t = native.Trace(ctx, n_rows, n, n_cols=cols)
        out[name] = dict(timed(n, lambda: synth(w, t, 0, n, 0), trace=t, ctype=ctype, cap=cap), capacity=cap, columns=cols, trace_bytes=cols * n_rows * 8)
        t.free()
        if ctype == 7:  # the accumulator chain of a request is serial (a wave per request, ~2.5 ms per call whatever the batch): a second figure at 32 instances per call
            n32 = min(32, w.num_instances)
            t = native.Trace(ctx, n_rows, n32, n_cols=cols)
            out[name]["at_32_instances_per_call"] = timed(n32, lambda: synth(w, t, 0, n32, 0))
            n64 = min(64, w.num_instances)
            t64 = native.Trace(ctx, n_rows, n64, n_cols=cols)
            out[name]["at_64_instances_per_call"] = timed(n64, lambda: synth(w, t64, 0, n64, 0))
            t64.free()
            out[name]["note"] = ("7 requests per instance (geometry_config.rs): the accumulator chain of a request is serial — k_ec_chain, a wave per request that owns its "
                                 "SIMD, 256-bit arithmetic with a limb per lane: ~2.5 ms per call whatever the batch (round 5: one lane per request, 13 ms) —, the segments "
                                 "are item lists (MAIN / MULS / LEAVES) side by side, the EC rows stream beside the netlist's fill: docs/KERNELS.md 3.19, profiles/r06/README.md")
            # ... and that chain is a wave per request on a SIMD of its own: MORE CALLS IN FLIGHT (a context = a stream, its own witness and slots, one
            # host thread each) run their chains under the other calls' segment / stream kernels.
            from concurrent.futures import ThreadPoolExecutor
            extra = []
            for _k in range(3):
                c_ = native.Context(local_rank)
                w_ = c_._precompile(kind, req, c_.queue_push_chain_log(c_.encode_log_queries(req))[1], mq, cap, mem_in)
                t_ = native.Trace(c_, n_rows, n32, n_cols=cols)
                c_.synthesize_ecrecover(w_, t_, 0, n32, 0)  # (slots claimed, kernels loaded)
                c_.synchronize()
                extra.append((c_, w_, t_))
            lanes = [(ctx, w, t)] + extra

            def calls(c_, w_, t_, reps_=6):
                for _ in range(reps_):
                    c_.synthesize_ecrecover(w_, t_, 0, n32, 0)
                c_.synchronize()

            in_flight = {}
            for nt in (2, 4):
                best_ = None
                with ThreadPoolExecutor(nt) as ex:
                    for _ in range(3):
                        torch.cuda.synchronize()
                        t0 = time.perf_counter()
                        for f in [ex.submit(calls, *lanes[k]) for k in range(nt)]:
                            f.result()
                        dt = time.perf_counter() - t0
                        best_ = dt if best_ is None or dt < best_ else best_
                in_flight[str(nt)] = {"circuits_per_s": nt * 6 * n32 / best_, "ms": best_ * 1e3}
            out[name]["calls_in_flight"] = dict(in_flight, instances_per_call=n32, calls_per_thread=6,
                                                note="N contexts, one host thread each, 6 calls of 32 instances per thread: a call's serial chains hide under the other calls' kernels")
            out[name]["two_calls_in_flight"] = in_flight["2"]
            for c_, w_, t_ in extra:
                t_.free()
                w_.free()
                c_.close()
            t.free()
        w.free()
    dec = ctx.compute_decommitts_sorter_circuit_snapshots(blk["decommit_queries"], 117500)
    dq, dt_ = dec.get(native.DEC_DEDUP_QUERIES), dec.get(native.DEC_DEDUP_TAILS)
    codes = [blk["bytecodes"][h.tobytes()] for h in dq["hash"]]
    woff = np.concatenate([[0], np.cumsum([c.shape[0] for c in codes])]).astype(np.uint64)
    w = ctx.compute_decommitter_circuit_snapshots(dq, dt_, np.concatenate(codes), woff, 2845, mem_in)
    n = min(4, w.num_instances)
    t = native.Trace(ctx, n_rows, n, n_cols=native.DC_COLS)
    out["code_decommitter"] = dict(timed(n, lambda: ctx.synthesize_code_decommitter(w, t, 0, n, 0), trace=t, ctype=3, cap=2845), capacity=2845, columns=native.DC_COLS,
                                   trace_bytes=native.DC_COLS * n_rows * 8)
    t.free()
    w.free()
    dec.free()
    queues = [synthetic.mixed_log_queue(4000, seed=3 + k)[:700] for k in range(8)]
    t = native.Trace(ctx, n_rows, 8, n_cols=native.LH_COLS)
    states = np.zeros(8, native.QUEUE_STATE4)
    qtails = [ctx.queue_push_chain_log(ctx.encode_log_queries(q))[1] for q in queues]  # the queues' states: the sorter that builds a queue holds them
    for k_, qt_ in enumerate(qtails):  # (a queue state names its tail: the circuit's closed-form section ties the last pop to it)
        states["tail"][k_] = qt_[-1]
        states["length"][k_] = len(queues[k_])
    out["linear_hasher"] = dict(timed(8, lambda: ctx.synthesize_linear_hasher_batch(queues, states, 774, t, 0, tails=qtails), trace=t, ctype=13, cap=774), capacity=774,
                                columns=native.LH_COLS, trace_bytes=native.LH_COLS * n_rows * 8,
                                note="the L1-messages queues of 8 blocks per call (zkw_linear_hasher_synthesize_batch_with_tails: the queues' "
                                     "states come from the sorter that built them, as in zkw_block_synthesize); the sponge of a "
                                     "queue is serial, ~4.5 ms whatever the batch; single_queue_ms_per_call hashes the queue's states inside the call",
                                single_queue_ms_per_call=timed(1, lambda: ctx.synthesize_linear_hasher(queues[0], states[:1], 774, t, 0))["ms_per_call"])
    t.free()
    # StorageApplication (type 10): Blake2s Merkle walks, 33 tree queries (8 481 cycles) per instance
    sq, _existing = synthetic.storage_application_trace(200, seed=4, write_fraction=0.6)
    stails = ctx.queue_push_chain_log(ctx.encode_log_queries(sq))[1]
    tree, answers = synthetic.storage_tree_for(sq, seed=1)
    idx, paths = answers(sq)
    w = ctx.decompose_into_storage_application_witnesses(sq, stails, idx, paths, tree.root, tree.next_enumeration_index, 33)
    n = min(8, w.num_instances)
    t = native.Trace(ctx, n_rows, n, n_cols=native.SA_COLS)
    out["storage_application"] = dict(timed(n, lambda: ctx.synthesize_storage_application(w, t, 0, n, 0), trace=t, ctype=10, cap=33), capacity=33, columns=native.SA_COLS,
                                      trace_bytes=native.SA_COLS * n_rows * 8, cycles_per_instance=33 * native.SA_CYCLES_PER_WALK)
    t.free()
    w.free()
    ctx.close()
    return out


def full_block_cpu(blk, threads):
    """The oracle (oracle/block.py: the builders one after the other in the reference's order on one thread, the way
    create_artifacts_from_tracer runs them, then every instance synthesized on up to `threads` threads — synthesis of
    distinct instances is independent in the reference too) on the SAME block."""
    from concurrent.futures import ThreadPoolExecutor

    from oracle import block as ob, pyoracle

    pyoracle.build()
    # untimed, like the GPU leg's first block: the deduplicated storage queries, for the tree that holds the pre-block state
    # (the same leaves as synthetic.storage_tree_for(dedup, seed=1), in the oracle's tree)
    dedup = ob.create_artifacts_after_vm(blk)["witnesses"]["storage_sorter"]["result_q"]
    tree = pyoracle.Tree()
    rng = np.random.default_rng(1)
    for _ in range(10):
        tree.insert_leaf(rng.bytes(32), rng.bytes(32))
    for q in dedup:
        if q["read_value"].any():
            tree.insert_leaf(pyoracle.derive_final_address(q), b"".join(int(x).to_bytes(4, "big") for x in q["read_value"][::-1]))
    timings = {}
    t0 = time.perf_counter()
    a = ob.create_artifacts_after_vm(blk, storage_tree=tree, timings=timings)
    t1 = time.perf_counter()
    jobs = [(ct, i) for ct in ob.EMISSION_ORDER if ob.SYNTH[ct][0] in a["witnesses"] for i in range(a["witnesses"][ob.SYNTH[ct][0]]["instances"].size)]

    def synth(job):
        ct, i = job
        key, fn = ob.SYNTH[ct]
        fn(a["witnesses"][key], i, a["capacities"][ct], 1 << 20)

    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(synth, jobs))
    t2 = time.perf_counter()
    conc, conc_s = ob.create_artifacts_after_vm_concurrent(blk, threads=threads)  # the same builders as a concurrent graph
    assert conc["ram_permutation"]["instances"].tobytes() == a["witnesses"]["ram_permutation"]["instances"].tobytes()
    return {"wall_ms": (t2 - t0) * 1e3, "builders_ms": (t1 - t0) * 1e3, "synthesis_ms": (t2 - t1) * 1e3, "cores": threads,
            "builders_sequential_ms": (t1 - t0) * 1e3, "builders_concurrent_ms": conc_s * 1e3,
            "wall_concurrent_ms": (conc_s + (t2 - t1)) * 1e3,
            "kind": "port", "instances_synthesized": len(jobs),
            "instances_in_block": 3 + sum(w_["instances"].size for w_ in a["witnesses"].values()),  # + 3 MainVM
            "builders_s": {k: round(v, 3) for k, v in timings.items()},
            "sample": "the same block: builders sequential on 1 thread (reference order, incl. the storage application over the "
                      "oracle's tree), %d instances synthesized on %d threads" % (len(jobs), threads)}


def launch_ranks(args):
    """`python bench.py --gpus N` without a torchrun environment: start N ranks of this same command line (one process per GPU,
    torch.distributed.run with a 127.0.0.1 rendezvous on a free port), let rank 0 print the JSON line on the inherited stdout,
    return the job's exit code. Fails loudly when the node has fewer than N GPUs: N ranks on fewer devices would be a different
    measurement (one process per GPU)."""
    import socket
    import subprocess

    n = args.gpus
    if not args.launcher_self_test:
        have = torch.cuda.device_count()
        if have < n:
            print(f"[bench] --gpus {n} but this node has {have} GPU(s): refusing to run {n} ranks on fewer devices", file=sys.stderr)
            return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    print(f"[bench] starting {n} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def launcher_self_test(args):
    """The N-rank plumbing of this file without a GPU (tests/test_bench_launcher.py): rendezvous from the torchrun variables over
    gloo, barrier + max-over-ranks timing, the per-instance closed-form records (176 B each, deterministic in rank and index)
    gathered to rank 0 in rank order through torch.distributed AND through libzkw's zkw_gather_closed_form_inputs over its TCP
    transport (the RCCL branch's code path minus the transport). No circuit is built: the line says so and carries no rate."""
    rank, _local_rank, world = parallel.init_from_env(backend="gloo")
    assert world == args.gpus, f"WORLD_SIZE={world} but --gpus {args.gpus}"
    inst_local, words = 5, 22
    rec = (np.arange(inst_local * words, dtype=np.uint64).reshape(inst_local, words) + np.uint64(1000 * rank)) * np.uint64(0x9E3779B97F4A7C15)
    counts = [inst_local] * world
    comm = native.Comm.tcp(None, "127.0.0.1", int(os.environ["MASTER_PORT"]) + 1, rank, world, 30000) if world > 1 else None
    recv = np.zeros((inst_local * world, words), np.uint64) if rank == 0 else None
    parallel.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = parallel.gather_records(torch.from_numpy(rec.view(np.uint8).reshape(inst_local, -1).copy()), counts, dst=0)
        if comm is not None:
            cnt = np.array(counts, np.uint64)
            native._check(native.load().zkw_gather_closed_form_inputs(comm.handle, rec.ctypes.data, cnt.ctypes.data, words * 8, 0,
                                                                      recv.ctypes.data if rank == 0 else None))
            comm.synchronize()
    parallel.barrier()
    dt = parallel.max_over_ranks(time.perf_counter() - t0, "cpu")
    if rank == 0:
        exp = np.concatenate([(np.arange(inst_local * words, dtype=np.uint64).reshape(inst_local, words) + np.uint64(1000 * r)) * np.uint64(0x9E3779B97F4A7C15)
                              for r in range(world)])
        got = out.numpy().reshape(-1).view(np.uint64).reshape(-1, words)
        ok = bool(np.array_equal(got, exp)) and (comm is None or bool(np.array_equal(recv, exp)))
        # the block-sharded leg's plan and record format (full_block.batched at N > 1): block k belongs to rank zkw_blocks_owner(k, N), its
        # closed-form records travel as ONE fixed-size record of 1 + 24 x max_per_block words through zkw_gather_records
        lib = native.load()
        n_blocks, max_per_block = 2 * world + 1, 3
        owners = [int(lib.zkw_blocks_owner(k, world)) for k in range(n_blocks)]
        print(json.dumps({"metric": "LAUNCHER SELF-TEST (no circuit work, not a measurement)", "value": None, "value_cold_slots": None, "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": dt / max(args.steps, 1) * 1e3, "records_gathered": int(got.shape[0]),
                          "records_match": ok, "config": {"workload": "none", "gather": "torch.distributed (gloo) + libzkw zkw_gather_closed_form_inputs (TCP)"},
                          "roofline": {"bytes_basis": None}, "synthesis": {"bytes_basis": None, "frac_of_hbm_peak_on_trace_bytes": None},
                          "full_block": {"batched": {"sharding": "zkw_blocks_run_sharded (round-robin over ranks) + zkw_blocks_gather_closed_form_inputs",
                                                     "n_gpus": world, "rccl_ranks": 0, "blocks_per_s": None, "per_rank_blocks_per_s": None,
                                                     "block_owners": owners, "record_words": 1 + 24 * max_per_block},
                                         "scaling_note": "the single block does not scale with N (replicated builders); `batched` does"}}),
              flush=True)
    if comm is not None:
        comm.destroy()
    parallel.barrier()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--blocks", type=int, default=0,
                    help="independent memory queues per GPU per step (0 = size the batch to the free HBM)")
    ap.add_argument("--queries", type=int, default=CAPACITY)
    ap.add_argument("--ring", type=int, default=20, help="trace buffers (1.25 GB each) in the output ring, all pipelines together (20: a synthesis phase of 10 instances per launch is as long as the other pipeline's builder phase, DESIGN.md 3.2)")
    ap.add_argument("--pipelines", type=int, default=int(os.environ.get("ZKW_PIPELINES", "2")),
                    help="the step's blocks are split into P sub-batches that run as independent pipelines (own context, HIP "
                         "stream, host thread), started a fraction of a chain pass apart so that one pipeline's synthesis "
                         "overlaps the other's queue chains (DESIGN.md 3.2: +10..14 %% at P = 2 on the HBM-sized batch; "
                         "P = 1 is the plain sequential step)")
    ap.add_argument("--stagger-ms", type=float, default=-1.0, help="start offset between pipelines (default 500 ms)")
    ap.add_argument("--cpu-sample", type=int, default=4)
    ap.add_argument("--no-h2d", action="store_true", help="skip the inputs-from-host leg (2 extra steps with a concurrent pinned H2D feed)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-hash-circuits", action="store_true", help="skip the synthesis-rate leg of the netlist circuits")
    ap.add_argument("--no-full-block", action="store_true", help="skip the full-block wall-time leg")
    ap.add_argument("--no-sensitivity", action="store_true", help="skip the sensitivity legs (cold slots, per-block address patterns, wide sort keys: 3 steps each)")
    ap.add_argument("--cold-steps", type=int, default=0, help="timed steps of the cold-slots figure `value_cold_slots` (0 = as many as --steps)")
    ap.add_argument("--no-validate", action="store_true", help="skip the oracle comparison of one ring slot per pipeline after the timed region")
    ap.add_argument("--launcher-self-test", action="store_true",
                    help="CPU only, no circuit work: the N ranks exchange synthetic closed-form records over gloo and over the C ABI's "
                         "TCP transport and rank 0 prints a line marked as a self-test (tests/test_bench_launcher.py)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))  # `python bench.py --gpus N` on its own: one process per GPU, this process only waits for them
    if args.launcher_self_test:
        return launcher_self_test(args)

    rank, local_rank, world = parallel.init_from_env()
    assert world == args.gpus, f"WORLD_SIZE={world} but --gpus {args.gpus}: start the ranks with torchrun --nproc-per-node {args.gpus}, or let bench.py do it (unset WORLD_SIZE)"
    assert torch.cuda.device_count() > local_rank, f"rank {rank}: no GPU {local_rank} (torch sees {torch.cuda.device_count()})"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    n = args.queries
    B = args.blocks
    P = max(1, args.pipelines)
    full_block = blk_inputs = hash_circuits = setup_commit = None
    comm_ctx = native.Context(local_rank)
    comm = make_comm(comm_ctx, rank, world)
    gather_backend = "libzkw zkw_gather_closed_form_inputs (RCCL)" if comm is not None else "torch.distributed (RCCL)"
    if not args.no_full_block and os.environ.get("ZKW_FULL_BLOCK_FIRST"):
        # experiment: the full-block legs BEFORE the timed region, as rounds 2-5 had them. They leave the process in one of two
        # states (round 5, six default runs: 2 896 - 3 039 circuits/s four times, 2 516 - 2 586 twice — every HBM- or latency-bound kernel of
        # the step 30 - 100 % slower, k_ram_fill_tail 12 -> 70 ms): hundreds of streams and the chain service's high-priority queues stay mapped
        # and the hardware scheduler time-slices them. The timed region now runs FIRST, in a process that has created nothing else.
        full_block, blk_inputs = full_block_gpu(local_rank, rank=rank, world=world, comm=comm)
        native.trim_caches()  # the batch below is sized by the free HBM
        torch.cuda.empty_cache()
    if B <= 0:
        free, _total = torch.cuda.mem_get_info(dev)
        # resident per query: input 48 + sorting permutation 4 + capacity words of the tails 2 x 32 = 116 bytes (no
        # encodings and no sorted copy are kept: every kernel re-encodes the 48-byte query, the sorted side through the
        # permutation; the grand-product chains and the sorted queries of the blocks being synthesized live in 2.3 GB of
        # windows per context, recomputed per synthesis launch; sort scratch aliases arrays that are filled later). The
        # chain kernel is serial per queue, so a step wants as many concurrent queues as fit.
        per_block = int(n * 128)
        cap = int(os.environ.get("ZKW_MAX_BLOCKS", "16384"))
        B = int(max(16, min(cap, (0.90 * free - max(2, args.ring // P) * P * 1.25e9 - P * 2.5e9) // per_block)))
        B = parallel.min_over_ranks(B, dev)  # every rank runs the same batch (weak scaling, equal record counts)
    if P > 1 and 2 * (B // P) <= 8400 and "--pipelines" not in sys.argv and "ZKW_PIPELINES" not in os.environ:
        P = 1  # small batches keep their chains below the cliff of DESIGN.md 3.2, where overlapping pipelines lose
    ring_p = max(2, args.ring // P)
    B = max(P, B // P * P)
    Bp = B // P
    ctxs, streams, chain_streams = [], [], []
    for _p in range(P):  # raises if libzkw / the GPU is missing: no fallback
        c = native.Context(local_rank)
        cu_split = os.environ.get("ZKW_CU_SPLIT")  # experiment: "x/y" hex words, e.g. 55555555/aaaaaaaa
        if cu_split:
            import ctypes as C
            hip = C.CDLL("libamdhip64.so")
            mx, my = (int(v, 16) for v in cu_split.split("/"))
            def masked(word):
                h = C.c_void_p()
                arr = (C.c_uint32 * 8)(*([word] * 8))
                assert hip.hipExtStreamCreateWithCUMask(C.byref(h), 8, arr) == 0
                return h.value
            st = torch.cuda.ExternalStream(masked(my), device=dev)   # everything but the chains
            c.set_stream(st.cuda_stream)
            c.set_chain_stream(masked(mx))
        else:
            # one stream per pipeline for torch ops and libzkw kernels. The HIP runtime multiplexes streams over a few
            # hardware queues per PRIORITY class (in order per queue): two pipelines whose streams land on one queue do
            # not overlap at all (measured: 1390 instead of 1790 circuits/s after the full-block legs had shuffled the
            # queue assignment). Alternating priorities puts neighbouring pipelines in different classes by construction.
            st = torch.cuda.Stream(device=dev, priority=-(_p % 2))
            c.set_stream(st.cuda_stream)
            if os.environ.get("ZKW_CHAIN_PRIO", "0") != "0":
                # the latency-bound queue chains on a HIGH-PRIORITY stream of their own: their waves issue first whenever
                # they are ready, the other pipeline's fills take the slots (and the HBM) they leave
                cst = torch.cuda.Stream(device=dev, priority=-1)
                chain_streams.append(cst)
                c.set_chain_stream(cst.cuda_stream)
        c.set_pointer_mode(native.PTR_DEVICE)
        if os.environ.get("ZKW_CHAIN_FORM"):
            c.set_chain_form(int(os.environ["ZKW_CHAIN_FORM"]))
        ctxs.append(c)
        streams.append(st)
    ctx = ctxs[0]
    torch.cuda.set_stream(streams[0])
    n_rows = 1 << 20  # TARGET_CIRCUIT_TRACE_LENGTH, base_layer/mod.rs:17
    rings = [native.Trace(ctxs[p], n_rows, ring_p) for p in range(P)]  # trace buffers a prover would consume and hand back
    base, q = make_inputs(B, n, rank, dev)
    offs = np.arange(Bp + 1, dtype=np.uint64) * n
    ws = [native.RamWitness(ctxs[p]) for p in range(P)]
    # what rank 0 needs from every instance to replay the recursion queue and build the scheduler witness
    # (SURVEY 8(e)): compact closed-form input (2 flags + 4 commitments = 18 words) + public input (4 words) = 176 B
    inst_bytes = (18 + 4) * 8
    inst_per_block = -(-n // CAPACITY)
    n_inst_p = Bp * inst_per_block
    n_inst_local = B * inst_per_block
    records = torch.empty((n_inst_local, inst_bytes), dtype=torch.uint8, device=dev)
    compact = torch.empty((n_inst_local, 18), dtype=torch.int64, device=dev)
    pis = torch.empty((n_inst_local, 4), dtype=torch.int64, device=dev)
    counts = [n_inst_local] * world
    lib = native.load()
    q_item = q.element_size() * q.shape[2]  # 48 bytes per query

    def pipeline_pass(p, last=False):
        """witness generation + synthesis of sub-batch p (blocks p*Bp .. (p+1)*Bp) on pipeline p's stream"""
        c, w, ring = ctxs[p], ws[p], rings[p]
        span = [p, time.perf_counter()]  # host clock of the pass: start | builders done | synthesis turn taken | synthesis done | end
        with torch.cuda.stream(streams[p]):
            c.compute_ram_circuit_snapshots((q.data_ptr() + p * Bp * n * q_item, Bp * n), CAPACITY, 0, block_offsets=offs, witness=w)
            if phased:
                # two pipelines in lockstep: a phase = one pipeline's builders next to the other's synthesis; the next phase
                # starts when both are done. Two chain passes never overlap (1 804 chain waves on 1 024 SIMDs: two on one SIMD take turns
                # and both chains stretch from 2.0 to 3.3 s), two synthesis phases never overlap either.
                streams[p].synchronize()
                span.append(time.perf_counter())
                phase_barrier.wait()
                span.append(time.perf_counter())
            elif P > 1 and synth_turns:
                # synthesis phases take turns: a pipeline fills at full speed while the others are in their chain pass,
                # instead of two synthesis phases slowing each other down
                streams[p].synchronize()
                span.append(time.perf_counter())
                synth_lock.acquire()
                span.append(time.perf_counter())
            try:
                for first in range(0, n_inst_p, ring_p):  # synthesis: every instance -> a full 2^20-row trace
                    if cold_slots[0]:  # sensitivity leg: the consumer took every slot's pointer, the layout tags are forgotten, every cell is written
                        for k_ in range(min(ring_p, n_inst_p - first)):
                            ring.device_ptr(k_)
                    c.synthesize_ram(w, ring, first, min(ring_p, n_inst_p - first), 0)
                if P > 1 and (synth_turns or phased):
                    streams[p].synchronize()
                    span.append(time.perf_counter())
                if phased and not (last and p == P - 1):
                    phase_barrier.wait()  # (the last synthesis of the last pipeline has nothing next to it)
            finally:
                if P > 1 and synth_turns and not phased:
                    synth_lock.release()
            lo, hi = p * n_inst_p, (p + 1) * n_inst_p
            cp, pp = compact[lo:hi], pis[lo:hi]
            native._check(lib.zkw_ram_witness_get(w.handle, native.RAM_COMPACT_FORMS, cp.data_ptr(), cp.numel() * 8))
            native._check(lib.zkw_ram_witness_get(w.handle, native.RAM_PUBLIC_INPUTS, pp.data_ptr(), pp.numel() * 8))
            rec = records.view(torch.int64).view(n_inst_local, 22)
            rec[lo:hi, :18] = cp
            rec[lo:hi, 18:] = pp
            streams[p].synchronize()
        span.append(time.perf_counter())
        pass_spans.append(span)

    from concurrent.futures import ThreadPoolExecutor
    import threading
    pool = ThreadPoolExecutor(P)
    synth_lock = threading.Lock()
    pass_spans = []
    synth_turns = os.environ.get("ZKW_SYNTH_TURNS", "1") != "0"
    cold_slots = [False]
    phased = P == 2 and os.environ.get("ZKW_PHASED", "0") != "0"  # experiment: with one chain workgroup per CU the turn-taking default does as well
    phase_barrier = threading.Barrier(2)

    def pipeline_run(p, passes, delay_s):
        torch.cuda.set_device(local_rank)
        try:
            if phased:
                if p == 1:
                    phase_barrier.wait()  # phase 0 is pipeline 0's builders alone
            elif delay_s > 0:
                time.sleep(delay_s)
            for k in range(passes):
                pipeline_pass(p, last=k == passes - 1)
        except BaseException:
            phase_barrier.abort()  # the other pipeline must not wait for a phase that will not come
            raise

    recv_all = torch.empty((n_inst_local * world, inst_bytes), dtype=torch.uint8, device=dev) if rank == 0 else None

    def gather_step():
        """the closed-form records of this step to rank 0: through the C ABI (RCCL inside libzkw) when available"""
        if comm is None:
            return parallel.gather_records(records, counts, dst=0)
        torch.cuda.synchronize()  # the records were written on the pipelines' streams; the gather runs on the comm context's
        comm.gather(records.data_ptr(), counts, inst_bytes, 0, recv_all.data_ptr() if rank == 0 else 0)
        comm_ctx.synchronize()
        return recv_all

    def run_steps(passes, stagger_s):
        """`passes` steps; after each step the closed-form records go to rank 0. P == 1: pass, gather, pass, gather ...
        P > 1: every pipeline makes `passes` passes over its sub-batch back to back (= `passes` passes over all B
        blocks), pipeline p starting p * stagger_s after pipeline 0, and the `passes` gathers follow."""
        out = None
        if P == 1:
            for _ in range(passes):
                pipeline_pass(0)
                out = gather_step()
            return out
        futs = [pool.submit(pipeline_run, p, passes, p * stagger_s) for p in range(P)]
        for f in futs:
            f.result()
        for _ in range(passes):
            out = gather_step()
        return out

    # Start offset between pipelines. A pipeline's pass = its queue chains (one wave per SIMD at most, latency-bound: T_chain
    # whatever the sub-batch size between ~9 k and ~19 k chains) followed by everything else (HBM / VALU-bound). Started
    # apart, one pipeline's synthesis runs inside the other's chain pass, whose waves leave issue slots free in the
    # throttled regime of DESIGN.md 3.2.
    # Offset = 0.72 x the chain pass of one sub-batch (2.1 s at P = 2: best of 2.1 / 2.4 / 2.8 s measured), from the
    # per-item latency of the regime the sub-batch is in rather than from a noisy warm-up measurement.
    per_item_s = 21.5e-6 if 2 * Bp > 8400 else 14.4e-6
    stagger_s = 0.72 * n * per_item_s * 2 / P if P > 1 else 0.0
    if P > 1 and os.environ.get("ZKW_SYNTH_TURNS", "1") != "0":
        stagger_s = 0.5  # with the synthesis phases taking turns the pipelines order themselves; a short offset starts that
    for k in range(args.warmup):
        run_steps(1, 0.0)
    if args.stagger_ms >= 0:
        stagger_s = args.stagger_ms * 1e-3
    for c in ctxs:
        c.profile_enable(True)
        c.profile_reset()
    parallel.barrier()
    torch.cuda.synchronize()
    pass_spans.clear()
    t0 = time.perf_counter()
    gathered = run_steps(args.steps, stagger_s)
    torch.cuda.synchronize()
    parallel.barrier()
    dt_rank = time.perf_counter() - t0
    dt = parallel.max_over_ranks(dt_rank, dev)
    dt_ranks = parallel.all_ranks(dt_rank, dev)
    timed_spans = [[s_[0]] + [round((x - t0) * 1e3, 1) for x in s_[1:]] for s_ in sorted(pass_spans, key=lambda s_: s_[1])]
    prof = {}
    for c in ctxs:  # kernel time per name, summed over the pipelines (HIP events on each pipeline's own stream)
        for k, (ms, cnt) in c.profile().items():
            a = prof.get(k, (0.0, 0))
            prof[k] = (a[0] + ms, a[1] + cnt)
        c.profile_enable(False)

    # ---- inputs arriving from host memory. The throughput leg's queries are resident in HBM; a host hands them over
    # from its own memory (external_calls::run's witness vectors). A true double buffer of the step's inputs needs twice their HBM
    # (2 x 95 GB next to the batch), so this leg measures the two things that decide whether the transfer matters: the pinned
    # host -> device rate of this box, and the step time while a copy stream moves one step's worth of input bytes per step (1 GiB chunks
    # from a pinned buffer into two alternating staging buffers) next to the unchanged compute. The consumed queries stay the resident ones.
    h2d = None
    if world == 1 and not args.no_h2d:
        try:
            chunk = 1 << 30
            host = torch.empty(chunk, dtype=torch.uint8, pin_memory=True)
            host.random_(0, 256)
            stage = [torch.empty(chunk, dtype=torch.uint8, device=dev) for _ in range(2)]
            cstream = torch.cuda.Stream(device=dev)
            step_bytes = B * n * q_item

            def feed(total_bytes, done):
                torch.cuda.set_device(local_rank)
                t_a = time.perf_counter()
                with torch.cuda.stream(cstream):
                    for k in range(-(-total_bytes // chunk)):
                        stage[k % 2].copy_(host, non_blocking=True)
                        if k % 8 == 7:
                            cstream.synchronize()  # (bounds the queue of pending copies)
                cstream.synchronize()
                done.append(time.perf_counter() - t_a)

            alone = []
            feed(16 * chunk, alone)
            rate_alone = 16 * chunk / alone[0] / 1e9
            done = []
            th = threading.Thread(target=feed, args=(2 * step_bytes, done))
            torch.cuda.synchronize()
            t_h = time.perf_counter()
            th.start()
            run_steps(2, stagger_s)
            torch.cuda.synchronize()
            dt_h = time.perf_counter() - t_h
            th.join()
            h2d = {"pinned_h2d_GBps_alone": rate_alone, "input_bytes_per_step": step_bytes,
                   "ms_per_step_with_concurrent_feed": dt_h / 2 * 1e3, "feed_ms_per_step": done[0] / 2 * 1e3,
                   "pinned_h2d_GBps_during_compute": 2 * step_bytes / done[0] / 1e9,
                   "circuits_per_s_with_concurrent_feed": 2 * n_inst_local / dt_h,
                   "transfer_hidden": done[0] <= dt_h,
                   "note": "2 steps of the same work while a copy stream moves one step's input bytes per step from pinned host memory into two "
                           "alternating 1 GiB staging buffers; the queries the kernels consume stay the resident ones (a full double buffer of the "
                           "inputs needs 2 x their HBM)"}
            del host, stage
        except Exception as e:  # noqa: BLE001 — a box without enough pinned memory: report, do not fail the bench
            h2d = {"error": repr(e)}

    # ---- the timed region's own output against the oracle: one ring slot of every pipeline, as the last pass left it, is
    # compared cell for cell with the CPU restatement's trace of the same block (untimed; the oracle is the checker here, never the product)
    validation = None
    if rank == 0 and not args.no_validate:
        from oracle import pyoracle

        pyoracle.build()
        checked = []
        last_first = (n_inst_p - 1) // ring_p * ring_p  # the last synthesis launch of a pass wrote instances last_first.. into slots 0..
        for p in range(P):
            slot = (n_inst_p - 1 - last_first) if p == 0 else 0  # pipeline 0: the pass's very last instance; the others: the launch's first
            inst = last_first + slot
            blk = p * Bp + inst // inst_per_block
            streams[p].synchronize()
            qh = q[blk].cpu().numpy().reshape(-1).view(native.MEM_QUERY)
            o = pyoracle.ram_build_instances(qh, CAPACITY, 0)
            exp = pyoracle.ram_synthesize(o, inst % inst_per_block, CAPACITY, n_rows)
            got = rings[p].get(slot)
            n_diff = int(np.count_nonzero(got != exp))
            pi_ok = bool(np.array_equal(pis[p * n_inst_p + inst].cpu().numpy().view(np.uint64), pyoracle.ram_public_inputs(o["instances"])[1][inst % inst_per_block]))
            checked.append({"pipeline": p, "block": int(blk), "instance": int(inst), "ring_slot": int(slot), "cells": int(got.size), "cells_differing": n_diff, "public_input_equal": pi_ok})
            del got, exp
        validation = {"checker": "oracle/liboracle.so (orc_ram_build_instances + orc_ram_synthesize)", "slots": checked,
                      "ok": all(c["cells_differing"] == 0 and c["public_input_equal"] for c in checked)}
        if not validation["ok"]:
            print(f"[bench] VALIDATION FAILED: {checked}", file=sys.stderr)

    # ---- how much the headline depends on three things the synthetic workload is kind about; `value` is unchanged.
    sensitivity = None
    if world == 1 and not args.no_sensitivity:
        def timed_steps(k):
            torch.cuda.synchronize()
            t_ = time.perf_counter()
            run_steps(k, stagger_s)
            torch.cuda.synchronize()
            return k * n_inst_local / (time.perf_counter() - t_)
        sensitivity = {"steps_per_leg": 3}
        try:
            run_steps(1, stagger_s)  # (the validation and host-feed legs left the pipelines idle: one untimed step first)
            sensitivity["baseline_circuits_per_s"] = timed_steps(3)  # the unchanged workload over the same 3 steps: what the legs compare with
            # (1) cold slots: the prover took every slot's pointer (zkw_trace_device_ptr), so a synthesis writes all 1 250 MB of a trace
            #     instead of the 578 MB that differ between two traces of the layout. Over as many steps as `value`
            cold_slots[0] = True
            run_steps(1, stagger_s)
            sensitivity["cold_slots_steps"] = args.cold_steps if args.cold_steps > 0 else args.steps
            sensitivity["cold_slots_circuits_per_s"] = timed_steps(sensitivity["cold_slots_steps"])
            cold_slots[0] = False
            run_steps(1, stagger_s)  # (re-warm the ring)
            # (2) every block its own address pattern: page ^= m_b, index ^= n_b (bijections, the trace stays a valid memory): 14 142 distinct
            #     sort permutations instead of one base trace's
            g2 = torch.Generator(device="cpu").manual_seed(77 + rank)
            pm = torch.randint(0, 64, (B, 1), generator=g2, dtype=torch.int32).to(dev)
            im = torch.randint(0, 256, (B, 1), generator=g2, dtype=torch.int32).to(dev)
            q[:, :, 1] ^= pm
            q[:, :, 2] ^= im
            sensitivity["per_block_address_patterns_circuits_per_s"] = timed_steps(3)
            # (3) wide keys: pages over 20 bits, indices over 16, timestamps over 32 (odd multipliers mod 2^k are bijections, the timestamp
            #     shift keeps the order): block + page + index + timestamp no longer fit one 64-bit key, the sort takes route two
            q[:, :, 1] = (q[:, :, 1] * 0x9E375) & 0xFFFFF
            q[:, :, 2] = (q[:, :, 2] * 0x9E37) & 0xFFFF
            q[:, :, 0] <<= 14  # 136 714 << 14 < 2^32 (the int32 tensor holds the same 32 bits)
            sensitivity["wide_keys_20_16_32_circuits_per_s"] = timed_steps(3)
            sensitivity["note"] = ("3 timed steps each after the timed region: every figure carries the pipelines' start-up, which `value` amortises over "
                                   f"{args.steps} steps — compare the legs with baseline_circuits_per_s, not with `value`; legs are cumulative in the order listed: (3) runs on (2)'s inputs")
        except Exception as e:  # noqa: BLE001 — a side leg
            sensitivity["error"] = repr(e)
            cold_slots[0] = False

    free_after, total_mem = torch.cuda.mem_get_info(dev)  # (HBM in use by the batch: read before it is released)
    batch_released = False
    if not args.no_full_block and full_block is None:
        # ---- the full-block legs, AFTER the timed region and with the batch released (every rank: N > 1 runs the deterministic, chain-bound
        # builders on every rank, synthesizes its LPT share of the block's instances, one gather to rank 0)
        for w_ in ws:
            w_.free()
        for r_ in rings:
            r_.free()
        del q, records, compact, pis
        torch.cuda.empty_cache()
        native.trim_caches()
        batch_released = True
        if rank == 0 and not args.no_hash_circuits:  # the hash-circuit rates before the full-block legs, for the same reason the timed region is first
            hash_circuits = hash_circuits_gpu(local_rank, synthetic.block_production(seed=1))
            try:
                setup_commit = setup_commit_gpu(local_rank)
            except Exception as e:  # noqa: BLE001 — a side leg: never the reason the contract's line is missing
                setup_commit = {"error": repr(e)}
            native.trim_caches()
            torch.cuda.empty_cache()
        full_block, blk_inputs = full_block_gpu(local_rank, rank=rank, world=world, comm=comm)
        native.trim_caches()
        torch.cuda.empty_cache()

    if rank == 0:
        assert gathered.shape[0] == n_inst_local * world
        circuits = n_inst_local * world * args.steps
        # dominant kernel by time, with its algorithmic HBM bytes per launch (DESIGN.md "Measurement")
        items = Bp * n  # queue items per builder launch (one sub-batch)
        # algorithmic HBM bytes PER LAUNCH of each kernel (DESIGN.md "Measurement"): cells written once +
        # the instance's witness read once. A synthesis launch covers `args.ring` instances.
        per_launch_inst = min(ring_p, n_inst_p)
        stride = (CAPACITY + 63) // 64 * 64            # rows per region incl. the alignment gap (zkw trace v2)
        cell = 8 * stride * per_launch_inst            # one column of one region, all instances of a launch
        cell_used = 8 * CAPACITY * per_launch_inst     # the rows of it a warm slot gets (the alignment gap keeps its zeros)
        wit = per_launch_inst * n                      # witness items read by a launch
        # cells a region's fill stores per cycle: the slots its row type uses + its lookup cells (generated spec); the cells that are zero
        # in every trace stay untouched in a slot that already holds the layout (slot layout tag)
        with open(os.path.join(ROOT, "include", "zkw_ram_circuit_spec.h")) as f_:
            spec_txt = f_.read()
        def spec_list(name):
            import re
            return [int(x) for x in re.search(r"#define %s \{([^}]*)\}" % name, spec_txt).group(1).split(",")]
        used = [a_ + b_ for a_, b_ in zip(spec_list("RC_ROW_NUM_SLOTS_INIT"), spec_list("RC_ROW_NUM_LOOKUPS_INIT"))][:6]  # PU PS A B C D
        alg_bytes = {
            "k_chain_full": 2 * items * (48 + 32),            # per chain item: the 48-byte query in, 4 capacity words out
            "k_chain_full_q4": 2 * items * (48 + 32),
            "k_gp_local": 2 * items * (48 + 16),              # queries read once, both repetitions written
            "k_gp_apply": 2 * items * 32,
            "k_encode_mem": items * (48 + 64),
            "k_gather_encode": items * (48 + 4 + 48),
            "k_ram_fill_poseidon": (used[0] + used[1]) / 2 * cell_used + wit * (48 + 32),  # one Poseidon2 region per launch
            "k_ram_fill_A": used[2] * cell_used + wit * (48 + 48 + 32),
            "k_ram_fill_B": used[3] * cell_used + wit * 96,
            "k_ram_fill_C": used[4] * cell_used + wit * 96,
            "k_ram_fill_D": used[5] * cell_used + 48 * cell_used,  # reads the queue tails back from the Poseidon2 rows
            # rows 0..255 of the multiplicity column; its other rows and the zero padding below the boundary rows are written only into a slot that held another layout
            # (slot layout tag): never inside the timed region, whose ring slots were filled by the warm-up step
            "k_ram_fill_tail": per_launch_inst * 8 * 256,
        }
        def pmc_traffic(kernel, launches_items):
            """HBM bytes per launch of `kernel` from the committed PMC passes (rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE, separate
            runs at the benchmarked batch, gfx950 FETCH correction applied: the newest profiles/rNN/traffic.json; counters cannot be read
            inside this process). Scaled by items per launch when this run's launch differs from the profiled one."""
            path = next((p_ for p_ in (os.path.join(ROOT, "profiles", r_, "traffic.json") for r_ in ("r06", "r05", "r04", "r03", "r02")) if os.path.exists(p_)), None)
            if path is None:
                return None
            t = json.load(open(path))
            k = t["kernels"].get("zkw::" + kernel)
            if not k:
                return None
            profiled_items = 2 * t["blocks"] * n if kernel.startswith("k_chain") else None
            scale = (launches_items / profiled_items) if profiled_items else 1.0
            return k["traffic_bytes_per_launch"] * scale

        def valu_roofline():
            """The bound that matters for this step: VALU issue, against a MEASURED ceiling. tools/ubench_valu_ceiling.hip
            (profiles/rNN/valu_ceiling.json) gives the chip's wave-instructions/s per instruction class at 1-8 waves per SIMD: full-rate ops
            (32-bit add / sub / logic / mov) 1.1e12, half-rate ops (v_mad_u64_u32, shifts, carries, 64-bit ops, DPP, v_cndmask on SGPR masks)
            6.0e11 — and the rate of the two instruction MIXES the Goldilocks kernels are made of, run with no memory traffic at the best
            occupancy: Poseidon2 one state per quad (the chain kernels' code) and one state per lane (the fills' code), 6.7-6.8e11. A kernel's
            ceiling is the rate of its own mix; the step's floor is the sum of its kernels' instruction volumes over their ceilings.
            Instruction counts per unit of work: SQ_INSTS_VALU of the committed PMC pass (profiles/rNN/valu.json, rocprofv3 --pmc
            SQ_INSTS_VALU at the benchmarked batch) when there is one, else the ISA counts of DESIGN.md."""
            cpath = next((p_ for p_ in (os.path.join(ROOT, "profiles", r_, "valu_ceiling.json") for r_ in ("r06", "r05")) if os.path.exists(p_)), None)
            if cpath is None:
                return None
            ceil_ = json.load(open(cpath))["summary"]
            mix = ceil_["goldilocks_mix_wave_insts_per_s"]
            per_unit = {"k_chain_full_q4": 507.0, "k_chain_full": 1480.0, "k_ram_fill_poseidon": 2 * 228.0 * n}  # per item / per instance
            src = "ISA instruction counts (DESIGN.md)"
            path = next((p_ for p_ in (os.path.join(ROOT, "profiles", r_, "valu.json") for r_ in ("r06", "r05", "r04")) if os.path.exists(p_)), None)
            if path is not None:
                v = json.load(open(path))
                per_unit.update({k.replace("zkw::", ""): x["wave_insts_per_unit"] for k, x in v["kernels"].items()})
                src = "SQ_INSTS_VALU, " + os.path.relpath(path, ROOT)
            units = {"k_chain_full_q4": 2 * items * P, "k_chain_full": 2 * items * P}  # queue items per step (all pipelines)
            per_kernel, total, floor_s = {}, 0.0, 0.0
            for k, (kms, kcnt) in prof.items():
                base_name = k.split("<")[0]
                if base_name not in per_unit or not kcnt:
                    continue
                u = units.get(base_name, n_inst_local)  # the fills: instances per step
                insts = per_unit[base_name] * u
                if base_name == "k_ram_fill_poseidon" and "<" in k:
                    insts /= 2  # two template instances share the per-instance count
                ceiling = mix["p2_quad_form"] if base_name.startswith("k_chain") else mix["p2_lane_form"]
                total += insts
                floor_s += insts / ceiling
                a = insts / (kms / args.steps * 1e-3)
                per_kernel[k] = {"wave_insts_per_step": insts, "ms_per_step": kms / args.steps, "achieved": a, "ceiling": ceiling, "frac": a / ceiling,
                                 "mix": "Poseidon2, one state per quad" if base_name.startswith("k_chain") else "Poseidon2 / field arithmetic, one state per lane"}
                if base_name.startswith("k_chain"):
                    # the chain launch runs ONE wave per SIMD by design (a serial chain per quad): what that occupancy allows for this code
                    waves = -(-2 * items // 16)
                    alone = ceil_["one_wave_per_simd"]["p2_quad_form_perm_per_s"] * min(1.0, waves / 1024.0)
                    per_kernel[k]["perm_per_s_inside_launches"] = 2 * items * kcnt / (kms * 1e-3)
                    per_kernel[k]["perm_per_s_one_wave_per_simd_alone"] = alone
                    per_kernel[k]["frac_of_one_wave_per_simd"] = per_kernel[k]["perm_per_s_inside_launches"] / alone
            floor_ms = floor_s * 1e3
            return {"bound": "valu", "peak_measured": {"full_rate": ceil_["full_rate_wave_insts_per_s"], "half_rate": ceil_["half_rate_wave_insts_per_s"], "goldilocks_mix": mix},
                    "peak": mix["p2_lane_form"], "unit": "wave-instructions/s", "ceiling_source": os.path.relpath(cpath, ROOT), "source": src, "per_kernel": per_kernel,
                    "step": {"wave_insts": total, "floor_ms": floor_ms, "ms_per_step": dt / args.steps * 1e3, "frac": floor_ms / (dt / args.steps * 1e3),
                             "note": "kernels with a counted instruction volume only (the chains, the Poseidon2 rows and rows A-D: ~95 % of the step's VALU work); "
                                     "frac = the step's VALU-issue floor at the measured ceiling of each kernel's instruction mix over its wall time. With P pipelines a "
                                     "kernel overlaps the other pipeline's kernels, so a per-kernel frac is its share of a SIMD it does not have to itself"}}

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
        synth_gbs = (native.circuit_fill_bytes(8, CAPACITY, n_rows)[0] * n_inst_local * args.steps) / (synth_ms * 1e-3) / 1e9 if synth_ms else None  # bytes actually written (slot reuse)
        chain_ms, chain_cnt = prof.get("k_chain_full", prof.get("k_chain_full_q4", (0.0, 1)))
        out = {
            "metric": "base-layer circuits/sec (2^20 rows); full-block synth wall-time 1/8 GPU",
            "value": circuits / dt,
            # the same workload when the consumer has taken every slot's pointer before each synthesis (all 1 250 MB of a trace written instead of
            # the 578 MB a slot that still holds the layout needs): the figure a prover that scribbles on its slots would see
            "value_cold_slots": (sensitivity or {}).get("cold_slots_circuits_per_s"),
            "value_cold_slots_steps": (sensitivity or {}).get("cold_slots_steps"),
            "unit": "circuits/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "per_rank_circuits_per_s": [n_inst_local * args.steps / d_ for d_ in dt_ranks],  # each rank's own clock over the same timed region
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64 (Goldilocks, p = 2^64 - 2^32 + 1)",
            "data": "synthetic",
            "config": {"workload": f"RAMPermutation base circuit, capacity {CAPACITY} (2^20-row geometry), "
                                   f"{B} independent memory queues per GPU per step, witness generation + synthesis "
                                   f"of every instance into a 149-column x 2^20-row trace",
                       "trace_layout": "zkw trace v2 (own gate placement, same geometry as the reference wrapper; not interoperable "
                                       "with the reference's vk_8 / finalization_hint_8: DESIGN.md section 4)",
                       "blocks_per_gpu": B, "queries_per_block": n, "parallelism": f"instances sharded x{world}",
                       "pipelines_per_gpu": P, "pipeline_stagger_ms": 0.0 if phased else stagger_s * 1e3,
                       "pipeline_schedule": ("lockstep: a phase = one pipeline's builders next to the other's synthesis, the next phase starts when both are done" if phased else
                                             "synthesis phases take turns" if P > 1 and synth_turns else "free-running" if P > 1 else "one pipeline"), "gather": gather_backend,
                       "ring_slots": args.ring, "instances_per_synthesis_launch": per_launch_inst,
                       "trace_slots": f"ring of {args.ring} slots; a slot that already holds this layout keeps every cell that is zero in all of its traces (layout tag): "
                                      "write_bytes_per_circuit of trace_bytes_per_circuit are written per synthesis",
                       "write_bytes_per_circuit": native.circuit_fill_bytes(8, CAPACITY, n_rows)[0],
                       "trace_bytes_per_circuit": native.circuit_fill_bytes(8, CAPACITY, n_rows)[1]},
            "roofline": {"bound": "hbm", "kernel": name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": pmc_traffic(name, 2 * items),
                         "traffic_unit": "bytes per launch (PMC, newest profiles/rNN/traffic.json)", "algorithmic_bytes_per_launch": ab,
                         "bytes_basis": "SURVEY 8(d) algorithmic bytes of the kernel: 48 B query read + 32 B capacity words written per queue item (nothing else moves)",
                         "avg_launch_ms": avg_ms,
                         "note": "k_chain_full is a serial Poseidon2 chain per queue: latency/VALU-bound, permutations/s is "
                                 "its meaningful rate; with P pipelines its launches overlap the other pipelines' synthesis, "
                                 "so per-kernel times sum to more than the wall time"},
            "roofline_valu": valu_roofline(),
            "inputs_from_host": h2d,
            "sensitivity": sensitivity,
            "synthesis": {"trace_bytes_per_circuit": 149 * n_rows * 8, "kernels_ms_per_step": synth_ms / args.steps,
                          "achieved_GBps": synth_gbs, "frac_of_hbm_peak": synth_gbs / HBM_PEAK_GBS if synth_gbs else None,
                          "bytes_basis": "written: write_bytes_per_circuit (a slot that holds the layout keeps the cells that are zero in every trace)",
                          "achieved_GBps_on_trace_bytes": (synth_gbs * 149 * n_rows * 8 / native.circuit_fill_bytes(8, CAPACITY, n_rows)[0]) if synth_gbs else None,
                          "frac_of_hbm_peak_on_trace_bytes": (synth_gbs * 149 * n_rows * 8 / native.circuit_fill_bytes(8, CAPACITY, n_rows)[0] / HBM_PEAK_GBS) if synth_gbs else None,
                          "bytes_basis_trace": "SURVEY 8(d): 149 x 2^20 x 8 B per circuit, what a cold slot gets; the kernels' time is the warm slots'",
                          "per_kernel": hbm_kernels},
            "hbm_used_GB": (total_mem - free_after) / 1e9,
            "poseidon2_perm_per_s": 2 * items * chain_cnt / (chain_ms * 1e-3) if chain_ms else None,  # inside the chain launches
            "kernels_ms_per_step": {k: v[0] / args.steps for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])},
            # host clock of every pipeline pass of the timed region, ms since its start: [pipeline, start, builders done, synthesis turn taken,
            # synthesis done, end] (P = 1: [pipeline, start, end]) - where the step's wall time goes between the two pipelines
            "pass_spans_ms": timed_spans,
            "validation": validation,
        }
        if full_block is not None:
            out["full_block"] = full_block
            if hash_circuits is not None:
                out["hash_circuits"] = hash_circuits
                if setup_commit is not None:
                    out["setup_commit"] = setup_commit
            elif not args.no_hash_circuits:
                # AFTER the timed region and with the batch released. Round 2 measured an 11 % loss of the throughput leg when this
                # leg ran first (1604 against 1800 circuits/s): the leg's context and streams shifted the round-robin assignment of
                # the pipelines' streams to hardware queues, and two pipelines on one queue do not overlap — the mechanism fixed at
                # the pipelines' stream creation above (alternating priorities). With that fix the order no longer matters:
                # ZKW_HASH_CIRCUITS_FIRST=1 gives 1761 against 1797 circuits/s (3-step runs, round 3, gpurun_out/r03l)
                if not batch_released:
                    for w_ in ws:
                        w_.free()
                    for r_ in rings:
                        r_.free()
                    del q, records, compact, pis
                    torch.cuda.empty_cache()
                    native.trim_caches()
                out["hash_circuits"] = hash_circuits_gpu(local_rank, blk_inputs)
                try:
                    out["setup_commit"] = setup_commit_gpu(local_rank)
                except Exception as e:  # a side leg: never the reason the contract's line is missing
                    out["setup_commit"] = {"error": repr(e)}
        if not args.no_cpu_baseline and world == 1:  # the CPU leg is reported at N = 1 only
            out["cpu_baseline"] = cpu_baseline(base, args.cpu_sample)
            if full_block is not None:
                cpu_fb = full_block_cpu(blk_inputs, max(1, min(os.cpu_count() or 1, 8)))
                out["full_block"]["cpu"] = cpu_fb
                # two ratios, not one: against the reference-order CPU side (builders sequential) and against a CPU side
                # whose builders are also a concurrent graph; the GPU's advantage on one block is the synthesis, its
                # builders are bound by one serial Poseidon2 chain (a host core runs a chain faster than one GPU wave)
                out["full_block"]["speedup_vs_cpu"] = cpu_fb["wall_ms"] / full_block["wall_ms"]
                out["full_block"]["speedup_vs_cpu_concurrent_builders"] = cpu_fb["wall_concurrent_ms"] / full_block["wall_ms"]
                out["full_block"]["builders_gpu_over_cpu_concurrent"] = full_block["builders_ms"] / cpu_fb["builders_concurrent_ms"]
        print(json.dumps(out), flush=True)
    parallel.barrier()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
