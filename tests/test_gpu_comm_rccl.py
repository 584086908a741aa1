"""GPU: the RCCL transport of csrc/zkw_comm.hip on the hardware that exists — ONE rank. zkw_comm_init_rccl forces the RCCL
branch at world == 1 (ncclCommInitRank over one rank), so that the dlopen, the symbol table, grouped ncclSend / ncclRecv on
the context's stream, the staging buffers of zkw_gather_records and the error paths have run on a real device before a
multi-GPU node sees them (the N > 1 gather logic itself runs between processes in tests/test_comm_tcp_multiprocess.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from era_zkevm_test_harness_amd import native

    c = native.Context(0)
    yield c
    c.close()


def test_rccl_single_rank_send_recv_to_self_and_gather(ctx):
    import torch
    from era_zkevm_test_harness_amd import native as nv

    comm = nv.Comm.rccl(ctx, 0, 1)  # unique id from ncclGetUniqueId, ncclCommInitRank(world = 1)
    # grouped send-to-self + receive-from-self through the transport table, stream-ordered after a kernel that writes src
    for nbytes in (192, 24 * 8 * 1000, 1 << 22):
        src = torch.randint(0, 256, (nbytes,), dtype=torch.uint8, device="cuda:0")
        dst = torch.zeros(nbytes, dtype=torch.uint8, device="cuda:0")
        torch.cuda.synchronize()
        comm.exchange(src.data_ptr(), dst.data_ptr(), nbytes, 0)
        comm.synchronize()
        assert torch.equal(src, dst), nbytes
    # the gather as the sequencer uses it, over the RCCL transport (one rank owns every record: the root's own share)
    rec = (np.arange(17 * 24, dtype=np.uint64).reshape(17, 24) * np.uint64(0x9E3779B97F4A7C15))
    out = comm.gather_records(np.zeros(17, np.uint32), rec, 0)
    assert np.array_equal(out, rec)
    out = comm.gather_records(np.zeros(0, np.uint32), rec[:0], 0)  # an empty block
    assert out.shape == (0, 24)
    # error paths: a peer that does not exist must come back as an error with the group closed, and the communicator stays usable
    src = torch.ones(64, dtype=torch.uint8, device="cuda:0")
    dst = torch.zeros(64, dtype=torch.uint8, device="cuda:0")
    with pytest.raises(nv.ZkwError):
        comm.exchange(src.data_ptr(), dst.data_ptr(), 64, 1)
    with pytest.raises(nv.ZkwError):
        comm.gather_records(np.array([0, 1], np.uint32), rec[:1], 0)  # owner 1 of 1 ranks
    comm.exchange(src.data_ptr(), dst.data_ptr(), 64, 0)
    comm.synchronize()
    assert torch.equal(src, dst)
    comm.destroy()


def test_rccl_and_local_transports_agree(ctx):
    from era_zkevm_test_harness_amd import native as nv

    rec = (np.arange(5 * 24, dtype=np.uint64).reshape(5, 24) + np.uint64(7)) * np.uint64(0xD1342543DE82EF95)
    a = nv.Comm(ctx, 0, 1)
    b = nv.Comm.rccl(ctx, 0, 1)
    ra = a.gather_records(np.zeros(5, np.uint32), rec, 0)
    rb = b.gather_records(np.zeros(5, np.uint32), rec, 0)
    assert np.array_equal(ra, rec) and np.array_equal(rb, rec)
    a.destroy()
    b.destroy()
