"""CPU, multi-process: the C-ABI collective itself (csrc/zkw_comm.hip: zkw_comm_init_tcp, zkw_gather_closed_form_inputs,
zkw_gather_records, zkw_shard_lpt) between real processes over the TCP transport with host memory — the code path RCCL runs
with device pointers, minus the transport. Covers what a one-rank GPU test cannot: unequal counts per rank, a rank that owns
nothing, root != 0, rank order -> emission order (the order the reference replays RecursionQueueSimulator pushes in,
src/witness/postprocessing/mod.rs:396-402)."""
import multiprocessing as mp
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port_block(n):
    for _ in range(50):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        p = s.getsockname()[1]
        s.close()
        if p + n < 65000:
            return p
    raise RuntimeError("no port")


BASIC_TEST_TYPES = [4, 8, 10, 10, 1, 1, 1, 2, 3, 5, 6, 7, 7, 9, 11, 12, 13]  # emission order of the reference's basic_test block


def _records(types):
    """record k = [type, index within type, 18 compact-form words, 4 public-input words], deterministic in (k, type)"""
    rec = np.zeros((len(types), 24), np.uint64)
    seen = {}
    for k, t in enumerate(int(x) for x in types):
        rec[k, 0], rec[k, 1] = t, seen.get(t, 0)
        seen[t] = seen.get(t, 0) + 1
        rec[k, 2:] = (np.arange(22, dtype=np.uint64) + np.uint64(1000 * k + t)) * np.uint64(0x9E3779B97F4A7C15)
    return rec


def _rank_main(rank, world, port, root, types, q):
    try:
        sys.path.insert(0, ROOT)
        from era_zkevm_test_harness_amd import native as nv

        comm = nv.Comm.tcp(None, "127.0.0.1", port, rank, world, 20000)
        owner = np.array(nv.shard_lpt(types, world), np.uint32)
        rec = _records(types)
        mine = rec[owner == rank]
        out = comm.gather_records(owner, mine, root)
        # second call on the same communicator (buffers are reused), this time through the counts-based entry point
        counts = np.bincount(owner, minlength=world).astype(np.uint64)
        recv = np.zeros((len(types), 24), np.uint64) if rank == root else None
        import ctypes as C
        nv._check(nv.load().zkw_gather_closed_form_inputs(comm.handle, mine.ctypes.data if mine.size else None, counts.ctypes.data, 192, root,
                                                         recv.ctypes.data if recv is not None else None))
        comm.synchronize()
        comm.destroy()
        q.put((rank, None if out is None else out.tolist(), None if recv is None else recv.tolist(), owner.tolist()))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, "error", traceback.format_exc(), None))


@pytest.mark.parametrize("world,root,types", [
    (2, 0, BASIC_TEST_TYPES),
    (2, 1, BASIC_TEST_TYPES),
    (3, 2, BASIC_TEST_TYPES),
    (3, 0, [8, 8]),           # rank 2 owns nothing
    (4, 1, [5]),              # three ranks own nothing
    (2, 0, []),               # an empty block
])
def test_gather_over_tcp_between_processes(world, root, types):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port_block(world)
    types = np.array(types, np.uint8)
    procs = [ctx.Process(target=_rank_main, args=(r, world, port, root, types, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    want = _records(types)
    for rank, out, recv, owner in got:
        assert out != "error", recv
        if rank == root:
            assert np.array_equal(np.array(out, np.uint64).reshape(-1, 24), want)  # emission order
            order = np.argsort(np.array(owner), kind="stable")  # rank order = what the raw counts-based gather delivers
            assert np.array_equal(np.array(recv, np.uint64).reshape(-1, 24), want[order])
        else:
            assert out is None and recv is None
    if len(types) >= world:
        assert len(set(got[0][3])) == world  # LPT uses every rank


def _late_rank_main(rank, world, port, delay_s, q):
    try:
        sys.path.insert(0, ROOT)
        import time
        from era_zkevm_test_harness_amd import native as nv

        if rank == 1:
            time.sleep(delay_s[0])  # connects late: the root's accept() has used up most of its init timeout by then
        comm = nv.Comm.tcp(None, "127.0.0.1", port, rank, world, 3000)
        if rank == 1:
            time.sleep(delay_s[1])  # ... and needs longer than what was left of it to finish its shard
        owner = np.array([0, 1, 1], np.uint32)
        rec = _records(np.array([8, 9, 9], np.uint8))
        out = comm.gather_records(owner, rec[owner == rank], 0)
        comm.destroy()
        q.put((rank, None if out is None else out.tolist()))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, "error: " + traceback.format_exc()))


def test_gather_waits_for_a_slow_peer():
    """a receive timeout on the listening socket was inherited by the accepted data sockets, so a peer that
    connected late and then computed for longer than the leftover init time failed the root's gather."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port_block(2)
    procs = [ctx.Process(target=_late_rank_main, args=(r, 2, port, (2.0, 2.5), q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    assert got[1] is None
    assert np.array_equal(np.array(got[0], np.uint64).reshape(-1, 24), _records(np.array([8, 9, 9], np.uint8)))


def _job_rank_main(rank, port, job, q):
    try:
        sys.path.insert(0, ROOT)
        os.environ["ZKW_COMM_JOB_ID"] = job
        from era_zkevm_test_harness_amd import native as nv

        try:
            comm = nv.Comm.tcp(None, "127.0.0.1", port, rank, 2, 1500)
            comm.destroy()
            q.put((rank, "connected"))
        except nv.ZkwError as e:
            q.put((rank, "refused: " + str(e)))
    except Exception:  # noqa: BLE001
        import traceback
        q.put((rank, "error: " + traceback.format_exc()))


def test_tcp_hello_turns_away_another_job():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port_block(2)
    procs = [ctx.Process(target=_job_rank_main, args=(r, port, "0x1234" if r == 0 else "0x9999", q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=60) for _ in procs)
    for p in procs:
        p.join(30)
    assert got[0].startswith("refused") and "timed out" in got[0], got  # rank 0 never saw a rank of ITS job
    # ... and the connector learns it at init (it used to store the descriptor as connected and fail at the first collective)
    assert got[1].startswith("refused") and "turned away" in got[1], got


def test_tcp_comm_rejects_bad_arguments():
    from era_zkevm_test_harness_amd import native as nv

    with pytest.raises(nv.ZkwError):
        nv.Comm.tcp(None, "not-an-address", 20000, 0, 1)
    with pytest.raises(nv.ZkwError):
        nv.Comm.tcp(None, "127.0.0.1", 20000, 2, 2)
    c = nv.Comm.tcp(None, "127.0.0.1", _free_port_block(1), 0, 1)  # one rank: no peers, the gather is a copy
    rec = _records(np.array([8, 8, 3], np.uint8))
    assert np.array_equal(c.gather_records(np.zeros(3, np.uint32), rec, 0), rec)
    with pytest.raises(nv.ZkwError):
        c.gather_records(np.array([0, 1, 0], np.uint32), rec[:2], 0)  # owner 1 of 1 ranks
    c.destroy()
