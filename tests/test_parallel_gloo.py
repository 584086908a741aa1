"""N > 1 path on CPU: two gloo ranks shard blocks, build their instances (with the oracle standing in for
the GPU builder — the collective and the shard plan are what is under test), gather the closed-form
records to rank 0 in instance order."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp

from era_zkevm_test_harness_amd import parallel, synthetic


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_blocks, q_out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from oracle import pyoracle

    r, _, w = parallel.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    bounds = parallel.shard_contiguous(n_blocks, world)
    recs = []
    for b in range(bounds[rank], bounds[rank + 1]):
        q = synthetic.ram_trace(300 + b, seed=100 + b)
        recs.append(pyoracle.ram_build_instances(q, 128)["instances"])
    local = np.concatenate(recs) if recs else np.zeros(0, pyoracle.RAM_INSTANCE)
    counts = [sum(-(-(300 + b) // 128) for b in range(bounds[k], bounds[k + 1])) for k in range(world)]
    t = torch.from_numpy(local.view(np.uint8).reshape(local.size, -1).copy())
    parallel.barrier()
    out = parallel.gather_records(t, counts, dst=0)
    assert parallel.min_over_ranks(100 + rank, "cpu") == 100  # the batch size every rank agrees on
    m = parallel.max_over_ranks(float(rank + 1), "cpu")
    assert m == float(world)
    if rank == 0:
        q_out.put(out.numpy().tobytes())
    else:
        assert out is None
    torch.distributed.destroy_process_group()


def test_two_rank_shard_and_gather():
    from oracle import pyoracle

    pyoracle.build()
    world, n_blocks = 2, 5
    ctx = mp.get_context("spawn")
    q_out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_blocks, q_out)) for r in range(world)]
    for p in procs:
        p.start()
    got = q_out.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    exp = np.concatenate([pyoracle.ram_build_instances(synthetic.ram_trace(300 + b, seed=100 + b), 128)["instances"]
                          for b in range(n_blocks)])
    assert got == exp.tobytes()


def test_shard_plans():
    assert parallel.shard_contiguous(17, 8) == [0, 3, 5, 7, 9, 11, 13, 15, 17]
    # the basic_test block: 17 instances over 8 GPUs -> every rank gets 2 or 3
    types = [1, 1, 1, 2, 3, 4, 5, 6, 7, 7, 8, 9, 10, 10, 11, 12, 13]
    owner = parallel.shard_lpt(types, 8)
    per_rank = [owner.count(r) for r in range(8)]
    assert sorted(per_rank) == [2, 2, 2, 2, 2, 2, 2, 3]
    assert parallel.shard_lpt(types, 1) == [0] * 17
    loads = [sum(parallel.ROWS_USED[t] for t, o in zip(types, owner) if o == r) for r in range(8)]
    assert max(loads) <= 3 * 1046318


def test_c_abi_shard_plan_matches_python():
    """zkw_shard_lpt (the plan a Rust host reaches through the C ABI; pure host code, no GPU) == parallel.shard_lpt"""
    import numpy as np

    from era_zkevm_test_harness_amd import native, parallel

    rng = np.random.default_rng(3)
    basic_test = [1, 1, 1, 2, 3, 4, 5, 6, 7, 7, 8, 9, 10, 10, 11, 12, 13]
    for world in (1, 2, 3, 8):
        assert native.shard_lpt(basic_test, world) == parallel.shard_lpt(basic_test, world)
        for _ in range(5):
            types = [int(x) for x in rng.integers(1, 14, int(rng.integers(1, 60)))]
            assert native.shard_lpt(types, world) == parallel.shard_lpt(types, world)
    loads = [0] * 8
    for t, r in zip(basic_test, native.shard_lpt(basic_test, 8)):
        loads[r] += 1
    assert sorted(loads) == [2, 2, 2, 2, 2, 2, 2, 3]  # 17 instances -> {3,2,2,2,2,2,2,2}
