"""GPU: two contexts on one device, each on its own HIP stream and host thread, building and synthesizing at the same
time (the bench's pipelines): results stay bit-exact vs the oracle — no state is shared between contexts."""
import threading

import numpy as np
import pytest

from era_zkevm_test_harness_amd import synthetic

pytestmark = pytest.mark.gpu


def test_two_contexts_in_parallel(oracle):
    import torch

    from era_zkevm_test_harness_amd import native

    capacity, n_rows, n = 700, 1 << 13, 2000
    qs = [synthetic.ram_trace(n, seed=11 + p) for p in range(2)]
    exp = []
    for q in qs:
        o = oracle.ram_build_instances(q, capacity, 0)
        exp.append((o, [oracle.ram_synthesize(o, i, capacity, n_rows) for i in range(o["instances"].size)]))
    errors = []

    def work(p):
        try:
            torch.cuda.set_device(0)
            ctx = native.Context(0)
            st = torch.cuda.Stream()
            ctx.set_stream(st.cuda_stream)
            with torch.cuda.stream(st):
                for _ in range(6):
                    w = ctx.compute_ram_circuit_snapshots(qs[p], capacity, 0)
                    o, traces = exp[p]
                    assert w.get(native.RAM_INSTANCES).tobytes() == o["instances"].tobytes()
                    t = native.Trace(ctx, n_rows, w.num_instances)
                    ctx.synthesize_ram(w, t)
                    for i, want in enumerate(traces):
                        assert np.array_equal(t.get(i), want), (p, i)
                        assert ctx.check_if_satisfied_ram(t, i, capacity)[0] == 0
                    t.free()
                    w.free()
            ctx.close()
        except BaseException as e:  # noqa: BLE001 - reported to the main thread
            errors.append((p, repr(e)))

    threads = [threading.Thread(target=work, args=(p,)) for p in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors


def test_chain_kernels_on_their_own_stream(oracle):
    """zkw_set_chain_stream: the queue chains run on a second stream (here a CU-masked one), ordered against the context's
    stream by events; results are unchanged"""
    import ctypes as C

    import torch

    from era_zkevm_test_harness_amd import native

    hip = C.CDLL("libamdhip64.so")
    h = C.c_void_p()
    mask = (C.c_uint32 * 8)(*([0x55555555] * 8))
    assert hip.hipExtStreamCreateWithCUMask(C.byref(h), 8, mask) == 0
    ctx = native.Context(0)
    st = torch.cuda.Stream()
    ctx.set_stream(st.cuda_stream)
    ctx.set_chain_stream(h.value)
    q = synthetic.ram_trace(3000, seed=5)
    with torch.cuda.stream(st):
        for _ in range(3):
            w = ctx.compute_ram_circuit_snapshots(q, 700, 0)
            o = oracle.ram_build_instances(q, 700, 0)
            assert w.get(native.RAM_INSTANCES).tobytes() == o["instances"].tobytes()
            assert np.array_equal(w.get(native.RAM_UNSORTED_TAILS), o["unsorted_tails"])
            w.free()
    ctx.set_chain_stream(None)
    ctx.close()
    assert hip.hipStreamDestroy(h) == 0
