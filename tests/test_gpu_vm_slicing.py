"""GPU: zkw_vm_slice_instances (one lane per instance: binary searches over the cycle stamps, one stable read / write
partition of the memory stream for all instances) against the oracle's sequential restatement of
src/witness/oracle.rs:1229-1469 — every record bit-exact, incl. a production-sized stream."""
import numpy as np
import pytest

from era_zkevm_test_harness_amd import synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from era_zkevm_test_harness_amd import native

    c = native.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("kw", [dict(seed=1), dict(seed=2, first_snapshot_cycle=500), dict(seed=3, n_memory=0, sparse=2),
                                dict(seed=4, n_cycles=20000, cycles_per_snapshot=5585, n_memory=136714, sparse=3000),
                                dict(seed=5, n_cycles=50, cycles_per_snapshot=5, n_memory=9000, sparse=40)])
def test_vm_slice_instances(ctx, oracle, kw):
    from era_zkevm_test_harness_amd import native

    t = synthetic.vm_tracer_streams(**kw)
    gi, gri, gwi = native.vm_slice_instances(ctx, t)
    oi, ori, owi = oracle.vm_slice_instances(t)
    assert gi.size == oi.size
    for name in gi.dtype.names:
        if name in ("first_memory_read", "first_memory_write"):
            continue  # the library partitions the whole memory stream, the oracle only what falls into a window: compare the slices
        assert gi[name].tobytes() == oi[name].tobytes(), name
    for g, o in zip(gi, oi):
        for f, n, ga, oa in (("first_memory_read", "num_memory_reads", gri, ori), ("first_memory_write", "num_memory_writes", gwi, owi)):
            a, b, k = int(g[f]), int(o[f]), int(g[n])
            assert np.array_equal(ga[a:a + k], oa[b:b + k])
    # the partition itself: reads and writes of the memory stream, each in order
    rw = t["vm_memory_queries"]["rw_flag"][:t["stream_cycles"][0].size]
    assert np.array_equal(gri, np.nonzero(rw == 0)[0]) and np.array_equal(gwi, np.nonzero(rw != 0)[0])
