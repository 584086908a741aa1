"""GPU parity: every entry point of the RAM-permutation path through the C ABI vs the CPU oracle,
bit-exact (integer arithmetic mod p), on seeded inputs, including ragged / edge shapes."""
import numpy as np
import pytest

from era_zkevm_test_harness_amd import synthetic

pytestmark = pytest.mark.gpu

P = 0xFFFFFFFF00000001


@pytest.fixture(scope="module")
def ctx():
    from era_zkevm_test_harness_amd import native

    c = native.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("n", [1, 63, 64, 65, 1000, 70001])
def test_encode(ctx, oracle, n):
    q = synthetic.ram_trace(n, seed=n)
    q["value"][0] = 0xFFFFFFFF
    assert np.array_equal(ctx.encode_memory_queries(q), oracle.encode_memory_queries(q))


def test_poseidon2_cooperative_and_lane_forms_match_oracle(ctx, oracle):
    # chain of length 1 from the zero tail == one permutation of (enc || 0000)
    enc = synthetic.random_field_elements(5, (7, 8))
    enc[0] = 0
    enc[1] = P - 1
    offsets = np.arange(8, dtype=np.uint64)  # 7 queues of one item: exercises all 4 DPP rows + a ragged wave
    got = ctx.queue_push_chain_full_batch(enc, offsets)
    for k in range(7):
        s = np.zeros(12, np.uint64)
        s[:8] = enc[k]
        assert np.array_equal(got[k], oracle.poseidon2(s)), k


@pytest.mark.parametrize("form", [1, 2, 4, 16])
def test_queue_chain_forms_agree(ctx, oracle, form):
    """the three layouts of the chain kernel (one lane / quad / row of 16 per state) against the oracle, ragged batch"""
    lens = [3, 0, 40, 1, 9, 17, 2, 5, 33, 8, 1, 1, 64, 7, 12, 3, 100, 6, 2] + [4, 9, 1] * 20  # 79 queues: > one wave in every form
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    enc = synthetic.random_field_elements(177, (int(offsets[-1]), 8))
    tins = synthetic.random_field_elements(178, (len(lens), 12))
    ctx.set_chain_form(form)
    try:
        got = ctx.queue_push_chain_full_batch(enc, offsets, tins)
    finally:
        ctx.set_chain_form(0)
    for k, ln in enumerate(lens):
        lo = int(offsets[k])
        if ln:
            assert np.array_equal(got[lo:lo + ln], oracle.queue_push_chain_full(enc[lo:lo + ln], tins[k])), k


def test_queue_chain_thousands_of_queues_take_the_quad_form(ctx, oracle):
    """4 100 short queues in one call: dev_chains picks the quad form by itself (>= 4 096 chains) and launches it as 4-wave workgroups,
    one per CU (launch_chain_q4: the path of the throughput benchmark); the last workgroup is partly empty. Every queue against the oracle."""
    rng = np.random.default_rng(5)
    lens = rng.integers(0, 4, 4100).tolist()
    lens[17], lens[4099] = 9, 3
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    enc = synthetic.random_field_elements(277, (int(offsets[-1]), 8))
    tins = synthetic.random_field_elements(278, (len(lens), 12))
    got = ctx.queue_push_chain_full_batch(enc, offsets, tins)
    for k in list(range(0, 4100, 37)) + [17, 4095, 4096, 4099]:
        lo, ln = int(offsets[k]), lens[k]
        if ln:
            assert np.array_equal(got[lo:lo + ln], oracle.queue_push_chain_full(enc[lo:lo + ln], tins[k])), k


@pytest.mark.parametrize("form", [1, 2, 4])
def test_ram_builder_chain_forms(ctx, oracle, form):
    """the RAM builder's chain path (queries encoded on the fly, the sorted side through the permutation, capacity words +
    instance-end tails as outputs) in the lane and quad forms: a ragged batch of blocks against the oracle"""
    from era_zkevm_test_harness_amd import native

    sizes = [2500, 700, 1301, 64, 999]
    qs = [synthetic.ram_trace(n, seed=300 + k) for k, n in enumerate(sizes)]
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    ctx.set_chain_form(form)
    try:
        w = ctx.compute_ram_circuit_snapshots(np.concatenate(qs), 512, 0, block_offsets=offs)
    finally:
        ctx.set_chain_form(0)
    ut, st, inst = w.get(native.RAM_UNSORTED_TAILS), w.get(native.RAM_SORTED_TAILS), w.get(native.RAM_INSTANCES)
    i0 = 0
    for k, q in enumerate(qs):
        o = oracle.ram_build_instances(q, 512, 0)
        lo, hi = int(offs[k]), int(offs[k + 1])
        assert np.array_equal(ut[lo:hi], o["unsorted_tails"]) and np.array_equal(st[lo:hi], o["sorted_tails"]), k
        ni = o["instances"].size
        assert inst[i0:i0 + ni].tobytes() == o["instances"].tobytes(), k
        i0 += ni
    w.free()


@pytest.mark.parametrize("n", [1, 2, 17, 500])
def test_queue_chain(ctx, oracle, n):
    enc = synthetic.random_field_elements(100 + n, (n, 8))
    assert np.array_equal(ctx.queue_push_chain_full(enc), oracle.queue_push_chain_full(enc))
    tin = synthetic.random_field_elements(7, (12,))
    assert np.array_equal(ctx.queue_push_chain_full(enc, tin), oracle.queue_push_chain_full(enc, tin))


def test_queue_chain_batch_ragged(ctx, oracle):
    lens = [5, 0, 33, 1, 64, 7, 12, 3, 100]  # 9 queues: two full waves + a partial one, one empty queue
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    enc = synthetic.random_field_elements(77, (int(offsets[-1]), 8))
    tins = synthetic.random_field_elements(78, (len(lens), 12))
    got = ctx.queue_push_chain_full_batch(enc, offsets, tins)
    for k, ln in enumerate(lens):
        lo = int(offsets[k])
        if ln:
            assert np.array_equal(got[lo:lo + ln], oracle.queue_push_chain_full(enc[lo:lo + ln], tins[k])), k


@pytest.mark.parametrize("state_w,n_chal", [(12, 9), (4, 21)])
def test_fs_challenges(ctx, oracle, state_w, n_chal):
    tu = synthetic.random_field_elements(21, (state_w,))
    ts = synthetic.random_field_elements(22, (state_w,))
    got = ctx.produce_fs_challenges(tu, 136714, ts, 136714, state_w, n_chal)
    assert np.array_equal(got, oracle.fs_challenges(tu, 136714, ts, 136714, state_w, n_chal))


@pytest.mark.parametrize("n,width", [(1, 8), (255, 8), (1024, 8), (1025, 8), (136714, 8), (46921, 20)])
def test_grand_product(ctx, oracle, n, width):
    lhs = synthetic.random_field_elements(31 + n, (n, width))
    rhs = lhs[np.random.default_rng(5).permutation(n)]
    ch = synthetic.random_field_elements(32, (2, width + 1))
    lz, rz = ctx.compute_grand_product_chains(lhs, rhs, ch)
    for r in range(2):
        rc, olz, orz = oracle.grand_product_chains(lhs, rhs, ch[r])
        assert rc == 0
        assert np.array_equal(lz[r], olz) and np.array_equal(rz[r], orz)
    l1, r1 = ctx.compute_grand_product_chains(lhs, rhs, ch[1])
    assert np.array_equal(l1, lz[1]) and np.array_equal(r1, rz[1])


def test_grand_product_detects_non_permutation(ctx):
    from era_zkevm_test_harness_amd import native

    lhs = synthetic.random_field_elements(1, (100, 8))
    rhs = lhs.copy()
    rhs[3, 2] = (int(rhs[3, 2]) + 1) % P
    with pytest.raises(native.ZkwError) as ei:
        ctx.compute_grand_product_chains(lhs, rhs, synthetic.random_field_elements(2, (9,)))
    assert ei.value.code == native.ERR_CHECK_FAILED


def _assert_witness_equal(w, o, native):
    assert np.array_equal(w.get(native.RAM_SORTED_QUERIES), o["sorted_q"])
    assert np.array_equal(w.get(native.RAM_UNSORTED_ENC), o["unsorted_enc"])
    assert np.array_equal(w.get(native.RAM_SORTED_ENC), o["sorted_enc"])
    assert np.array_equal(w.get(native.RAM_UNSORTED_TAILS), o["unsorted_tails"])
    assert np.array_equal(w.get(native.RAM_SORTED_TAILS), o["sorted_tails"])
    assert np.array_equal(w.get(native.RAM_CHALLENGES)[0], o["challenges"])
    n = o["sorted_q"].size
    assert np.array_equal(w.get(native.RAM_LHS_Z).reshape(2, n), o["lhs_z"])
    assert np.array_equal(w.get(native.RAM_RHS_Z).reshape(2, n), o["rhs_z"])
    gi, oi = w.get(native.RAM_INSTANCES), o["instances"]
    assert gi.size == oi.size
    assert gi.tobytes() == oi.tobytes()
    from oracle import pyoracle

    compact, pi = pyoracle.ram_public_inputs(oi)
    assert np.array_equal(w.get(native.RAM_COMPACT_FORMS), compact)
    assert np.array_equal(w.get(native.RAM_PUBLIC_INPUTS), pi)


@pytest.mark.parametrize("n,capacity", [(1, 4), (64, 64), (1000, 128), (8192, 2048), (8192, 8192), (5000, 136714)])
def test_ram_builder(ctx, oracle, n, capacity):
    from era_zkevm_test_harness_amd import native

    q = synthetic.ram_trace(n, seed=1000 + n)
    k = min(n, 3)
    q["page"][:k] = 10
    q["timestamp"][:k] = 0
    q["rw_flag"][:k] = 1
    w = ctx.compute_ram_circuit_snapshots(q, capacity, 3)
    o = oracle.ram_build_instances(q, capacity, 3)
    assert w.num_instances == o["instances"].size
    _assert_witness_equal(w, o, native)
    w.free()


def test_ram_builder_duplicate_keys_is_stable(ctx, oracle):
    from era_zkevm_test_harness_amd import native

    q = synthetic.ram_trace(3000, seed=4, pages=2, indices=4)
    q["timestamp"] = q["timestamp"] // 8  # many (cell, ts) collisions: order must follow the stable sort
    w = ctx.compute_ram_circuit_snapshots(q, 512, 0)
    _assert_witness_equal(w, oracle.ram_build_instances(q, 512, 0), native)
    w.free()


def test_ram_builder_batch_of_blocks(ctx, oracle):
    from era_zkevm_test_harness_amd import native

    lens = [700, 1, 4096, 333, 2500]
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    q = np.concatenate([synthetic.ram_trace(ln, seed=50 + i) for i, ln in enumerate(lens)])
    w = ctx.compute_ram_circuit_snapshots(q, 1024, [0, 1, 2, 3, 4], block_offsets=offs)
    sq, ue, se = w.get(native.RAM_SORTED_QUERIES), w.get(native.RAM_UNSORTED_ENC), w.get(native.RAM_SORTED_ENC)
    ut, st = w.get(native.RAM_UNSORTED_TAILS), w.get(native.RAM_SORTED_TAILS)
    ch, lz, rz, inst = (w.get(native.RAM_CHALLENGES), w.get(native.RAM_LHS_Z), w.get(native.RAM_RHS_Z),
                        w.get(native.RAM_INSTANCES))
    i0 = 0
    for b, ln in enumerate(lens):
        lo = int(offs[b])
        o = oracle.ram_build_instances(q[lo:lo + ln], 1024, b)
        assert np.array_equal(sq[lo:lo + ln], o["sorted_q"]), b
        assert np.array_equal(ue[lo:lo + ln], o["unsorted_enc"]) and np.array_equal(se[lo:lo + ln], o["sorted_enc"])
        assert np.array_equal(ut[lo:lo + ln], o["unsorted_tails"]) and np.array_equal(st[lo:lo + ln], o["sorted_tails"])
        assert np.array_equal(ch[b], o["challenges"])
        assert np.array_equal(lz[2 * lo:2 * (lo + ln)].reshape(2, ln), o["lhs_z"])
        assert np.array_equal(rz[2 * lo:2 * (lo + ln)].reshape(2, ln), o["rhs_z"])
        k = o["instances"].size
        assert inst[i0:i0 + k].tobytes() == o["instances"].tobytes()
        i0 += k
    assert i0 == w.num_instances
    w.free()


@pytest.mark.parametrize("route,pages,indices,ts_start,n_blocks", [
    ("one", 64, 256, 1, 3),                 # every field narrow: a single packed sort
    ("one", 1, 1, 0, 1),                    # one cell, one block: a key of (almost) no bits
    ("two", 2**32 - 9, 2**16, 2**31, 3),    # 2 + 32 + 16 + 32 bits: timestamps first, then the packed cell word
    ("three", 2**32 - 9, 2**32, 2**31, 3),  # 2 + 32 + 32 bits of cell alone: timestamp, cell, block id
    ("two", 2**32 - 9, 2**32, 2**31, 1),    # one block: full-width cell word, no block pass
])
def test_ram_sort_routes_by_key_width(ctx, oracle, route, pages, indices, ts_start, n_blocks):
    """The sort packs (block, page, index, timestamp) into as few radix passes as the batch's key widths allow; each
    route has to give the reference's stable order (W/ram_permutation.rs:48-53), including (cell, ts) ties."""
    from era_zkevm_test_harness_amd import native

    lens = [1500, 1, 2200][:n_blocks]
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    blocks = []
    for i, ln in enumerate(lens):
        b = synthetic.ram_trace(ln, seed=900 + i, pages=pages, indices=indices, first_page=8 if pages > 1 else 0,
                                ts_start=ts_start)
        if pages > 64:  # wide keys are sparse: fold most of them onto a few cells so that ties and runs exist
            fold = np.arange(ln) % 3 != 0
            b["page"][fold] = b["page"][0]
            b["index"][fold] = b["index"][fold] % 5
        b["timestamp"] = ts_start + (b["timestamp"] - ts_start) // 4
        blocks.append(b)
    q = np.concatenate(blocks)
    w = ctx.compute_ram_circuit_snapshots(q, 512, list(range(n_blocks)), block_offsets=offs)
    sq, inst = w.get(native.RAM_SORTED_QUERIES), w.get(native.RAM_INSTANCES)
    i0 = 0
    for b, ln in enumerate(lens):
        lo = int(offs[b])
        o = oracle.ram_build_instances(q[lo:lo + ln], 512, b)
        assert np.array_equal(sq[lo:lo + ln], o["sorted_q"]), (route, b)
        k = o["instances"].size
        assert inst[i0:i0 + k].tobytes() == o["instances"].tobytes(), (route, b)
        i0 += k
    w.free()


def test_full_size_properties(ctx):
    """Production capacity (136 714 queries, BASELINE config geometry): size-independent properties."""
    from era_zkevm_test_harness_amd import native

    n = 136714
    q = synthetic.ram_trace(2 * n + 11, seed=2)
    w = ctx.compute_ram_circuit_snapshots(q, n, 0)
    assert w.num_instances == 3
    inst = w.get(native.RAM_INSTANCES)
    sq = w.get(native.RAM_SORTED_QUERIES)
    key = (sq["page"].astype(np.uint64) << np.uint64(32)) | sq["index"]
    assert np.all(key[1:] >= key[:-1])
    fo = inst[-1]["hidden_fsm_output"]
    assert np.array_equal(fo["lhs_accumulator"], fo["rhs_accumulator"])  # the permutation argument closes
    assert np.array_equal(inst[1]["hidden_fsm_input"]["lhs_accumulator"], inst[0]["hidden_fsm_output"]["lhs_accumulator"])
    assert not np.array_equal(inst[0]["hidden_fsm_output"]["lhs_accumulator"], inst[0]["hidden_fsm_output"]["rhs_accumulator"])
    ut = w.get(native.RAM_UNSORTED_TAILS)
    assert np.array_equal(inst[0]["hidden_fsm_output"]["current_unsorted_queue_state"]["head"], ut[n - 1])
    assert int(ut.max()) < P
    w.free()


@pytest.mark.parametrize("n", [1, 255, 3000])
def test_log_and_decommit_encodings(ctx, oracle, n):
    q = synthetic.random_log_queries(n, seed=n)
    q["key"][0] = 0xFFFFFFFF
    q["address"][0] = 0xFFFFFFFF
    assert np.array_equal(ctx.encode_log_queries(q), oracle.encode_log_queries(q))
    ext = (np.arange(n, dtype=np.uint32) * np.uint32(2654435761)) | np.uint32(0x80000000)
    assert np.array_equal(ctx.encode_log_queries(q, ext), oracle.encode_log_queries(q, ext))
    d = synthetic.random_decommit_queries(n, seed=n + 1)
    assert np.array_equal(ctx.encode_decommit_queries(d), oracle.encode_decommit_queries(d))


def test_log_queue_chain_batch(ctx, oracle):
    lens = [5, 0, 33, 1, 64, 7, 130]
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    enc = oracle.encode_log_queries(synthetic.random_log_queries(int(offsets[-1]), seed=9))
    tins = synthetic.random_field_elements(78, (len(lens), 4))
    old_t, new_t = ctx.queue_push_chain_log(enc, offsets, tins)
    for k, ln in enumerate(lens):
        lo = int(offsets[k])
        if ln:
            o_old, o_new = oracle.queue_push_chain_log(enc[lo:lo + ln], tins[k])
            assert np.array_equal(old_t[lo:lo + ln], o_old) and np.array_equal(new_t[lo:lo + ln], o_new), k
    o2, n2 = ctx.queue_push_chain_log(enc[:40])
    e_old, e_new = oracle.queue_push_chain_log(enc[:40])
    assert np.array_equal(o2, e_old) and np.array_equal(n2, e_new)


@pytest.mark.parametrize("n,hashes,capacity", [(1, 1, 4), (64, 5, 16), (100, 30, 16), (3000, 700, 256), (5000, 40, 117500),
                                               (2048, 2048, 1024)])
def test_decommit_sorter(ctx, oracle, n, hashes, capacity):
    from era_zkevm_test_harness_amd import native as nv

    q = synthetic.decommit_trace(n, hashes, seed=n + hashes)
    din = np.zeros(1, oracle.QUEUE_STATE12)
    din["tail"] = synthetic.random_field_elements(3, (12,))
    din["head"] = synthetic.random_field_elements(4, (12,))
    din["length"] = 9
    for dedup_in in (None, din):
        w = ctx.compute_decommitts_sorter_circuit_snapshots(q, capacity, dedup_in)
        o = oracle.decommit_sorter_build(q, capacity, dedup_in)
        assert w.num_dedup == o["dedup_q"].size
        for what, key in ((nv.DEC_SORTED_QUERIES, "sorted_q"), (nv.DEC_UNSORTED_ENC, "unsorted_enc"),
                          (nv.DEC_SORTED_ENC, "sorted_enc"), (nv.DEC_UNSORTED_TAILS, "unsorted_tails"),
                          (nv.DEC_SORTED_TAILS, "sorted_tails"), (nv.DEC_DEDUP_QUERIES, "dedup_q"),
                          (nv.DEC_DEDUP_TAILS, "dedup_tails"), (nv.DEC_CHALLENGES, "challenges"),
                          (nv.DEC_LHS_Z, "lhs_z"), (nv.DEC_RHS_Z, "rhs_z")):
            assert np.array_equal(w.get(what), o[key]), key
        gi = w.get(nv.DEC_INSTANCES)
        assert gi.size == o["instances"].size
        for a, b in zip(gi, o["instances"]):
            for f in a.dtype.names:
                assert a[f].tobytes() == b[f].tobytes(), f
        w.free()


def test_decommit_sorter_self_check(ctx):
    from era_zkevm_test_harness_amd import native as nv

    q = synthetic.decommit_trace(40, 3, seed=2)
    q["memory_page"][-1] += 1
    with pytest.raises(nv.ZkwError) as ei:
        ctx.compute_decommitts_sorter_circuit_snapshots(q, 16)
    assert ei.value.code == nv.ERR_CHECK_FAILED


@pytest.mark.parametrize("nf,frac,capacity", [(1, 0.0, 4), (1, 1.0, 4), (40, 0.3, 16), (2000, 0.5, 256), (64, 0.0, 16),
                                              (30, 1.0, 7), (5000, 0.2, 31287)])
def test_events_sorter(ctx, oracle, nf, frac, capacity):
    from era_zkevm_test_harness_amd import native as nv

    q = synthetic.events_trace(nf, frac, seed=nf + 1)
    rin = np.zeros(1, oracle.QUEUE_STATE4)
    rin["tail"] = synthetic.random_field_elements(3, (4,))
    rin["head"] = synthetic.random_field_elements(4, (4,))
    rin["length"] = 3
    for result_in in (None, rin):
        w = ctx.compute_events_dedup_and_sort(q, capacity, result_in)
        o = oracle.events_sorter_build(q, capacity, result_in)
        assert w.num_results == o["result_q"].size
        for what, key in ((nv.EVT_SORTED_QUERIES, "sorted_q"), (nv.EVT_UNSORTED_ENC, "unsorted_enc"), (nv.EVT_SORTED_ENC, "sorted_enc"),
                          (nv.EVT_UNSORTED_OLD_TAILS, "unsorted_old_tails"), (nv.EVT_UNSORTED_NEW_TAILS, "unsorted_new_tails"),
                          (nv.EVT_SORTED_OLD_TAILS, "sorted_old_tails"), (nv.EVT_SORTED_NEW_TAILS, "sorted_new_tails"),
                          (nv.EVT_RESULT_QUERIES, "result_q"), (nv.EVT_RESULT_NEW_TAILS, "result_new_tails"),
                          (nv.EVT_CHALLENGES, "challenges"), (nv.EVT_LHS_Z, "lhs_z"), (nv.EVT_RHS_Z, "rhs_z")):
            assert np.array_equal(w.get(what), o[key]), key
        gi = w.get(nv.EVT_INSTANCES)
        assert gi.size == o["instances"].size
        for a, b in zip(gi, o["instances"]):
            for f in a.dtype.names:
                assert a[f].tobytes() == b[f].tobytes(), f
        w.free()


def test_events_sorter_empty_and_malformed(ctx, oracle):
    from era_zkevm_test_harness_amd import native as nv

    w = ctx.compute_events_dedup_and_sort(np.zeros(0, nv.LOG_QUERY), 16)
    o = oracle.events_sorter_build(np.zeros(0, oracle.LOG_QUERY), 16)
    assert w.get(nv.EVT_INSTANCES).tobytes() == o["instances"].tobytes()
    w.free()
    q = synthetic.events_trace(10, 0.0, seed=3)
    q["rollback"][4] = 1
    with pytest.raises(nv.ZkwError) as ei:
        ctx.compute_events_dedup_and_sort(q, 16)
    assert ei.value.code == nv.ERR_CHECK_FAILED


@pytest.mark.parametrize("n,capacity", [(1, 4), (100, 16), (3000, 128), (77, 1000), (5000, 58750), (2048, 1024)])
def test_log_demux(ctx, oracle, n, capacity):
    from era_zkevm_test_harness_amd import native as nv

    q = synthetic.mixed_log_queue(n, seed=n + 3)
    w = ctx.compute_logs_demux(q, capacity)
    o = oracle.log_demux_build(q, capacity)
    for what, key in ((nv.DMX_IN_ENC, "in_enc"), (nv.DMX_IN_OLD_TAILS, "in_old_tails"), (nv.DMX_IN_NEW_TAILS, "in_new_tails"),
                      (nv.DMX_OUT_QUERIES, "out_q"), (nv.DMX_OUT_ENC, "out_enc"), (nv.DMX_OUT_OLD_TAILS, "out_old_tails"),
                      (nv.DMX_OUT_NEW_TAILS, "out_new_tails"), (nv.DMX_OUT_OFFSETS, "out_offsets")):
        assert np.array_equal(w.get(what), o[key]), key
    gi = w.get(nv.DMX_INSTANCES)
    assert gi.size == o["instances"].size
    for a, b in zip(gi, o["instances"]):
        for f in a.dtype.names:
            assert a[f].tobytes() == b[f].tobytes(), f
    w.free()


def test_log_demux_empty_and_unreachable(ctx, oracle):
    from era_zkevm_test_harness_amd import native as nv

    w = ctx.compute_logs_demux(np.zeros(0, nv.LOG_QUERY), 16)
    assert w.get(nv.DMX_INSTANCES).tobytes() == oracle.log_demux_build(np.zeros(0, oracle.LOG_QUERY), 16)["instances"].tobytes()
    w.free()
    q = synthetic.mixed_log_queue(50, seed=1)
    q["aux_byte"][7] = 9
    with pytest.raises(nv.ZkwError) as ei:
        ctx.compute_logs_demux(q, 16)
    assert ei.value.code == nv.ERR_CHECK_FAILED


@pytest.mark.parametrize("n,cells,capacity", [(1, 1, 4), (60, 4, 16), (3000, 300, 256), (300, 300, 64), (128, 9, 32),
                                              (5000, 50, 46921), (2048, 700, 1024)])
def test_storage_sorter(ctx, oracle, n, cells, capacity):
    from era_zkevm_test_harness_amd import native as nv

    q = synthetic.storage_trace(n, cells, seed=n + cells + 1)
    w = ctx.compute_storage_dedup_and_sort(q, capacity)
    o = oracle.storage_sorter_build(q, capacity)
    assert w.num_results == o["result_q"].size
    for what, key in ((nv.STO_SORTED_QUERIES, "sorted_q"), (nv.STO_SORTED_EXT_TS, "sorted_ext_ts"), (nv.STO_UNSORTED_ENC, "unsorted_enc"),
                      (nv.STO_LHS_ENC, "lhs_enc"), (nv.STO_SORTED_ENC, "sorted_enc"), (nv.STO_UNSORTED_OLD_TAILS, "unsorted_old_tails"),
                      (nv.STO_UNSORTED_NEW_TAILS, "unsorted_new_tails"), (nv.STO_SORTED_OLD_TAILS, "sorted_old_tails"),
                      (nv.STO_SORTED_NEW_TAILS, "sorted_new_tails"), (nv.STO_RESULT_QUERIES, "result_q"),
                      (nv.STO_RESULT_NEW_TAILS, "result_new_tails"), (nv.STO_CHALLENGES, "challenges"), (nv.STO_LHS_Z, "lhs_z"),
                      (nv.STO_RHS_Z, "rhs_z")):
        assert np.array_equal(w.get(what), o[key]), key
    gi = w.get(nv.STO_INSTANCES)
    assert gi.size == o["instances"].size
    for a, b in zip(gi, o["instances"]):
        for f in a.dtype.names:
            if a[f].dtype.names:
                for g in a[f].dtype.names:
                    assert a[f][g].tobytes() == b[f][g].tobytes(), (f, g)
            else:
                assert a[f].tobytes() == b[f].tobytes(), f
    w.free()


def test_storage_sorter_empty_and_inconsistent(ctx, oracle):
    from era_zkevm_test_harness_amd import native as nv

    w = ctx.compute_storage_dedup_and_sort(np.zeros(0, nv.LOG_QUERY), 16)
    assert w.get(nv.STO_INSTANCES).tobytes() == oracle.storage_sorter_build(np.zeros(0, oracle.LOG_QUERY), 16)["instances"].tobytes()
    w.free()
    q = synthetic.storage_trace(40, 3, seed=9, p_rollback=0.0)
    q["rw_flag"][0], q["rollback"][0] = 1, 1  # a rollback with nothing to roll back
    with pytest.raises(nv.ZkwError) as ei:
        ctx.compute_storage_dedup_and_sort(q, 16)
    assert ei.value.code == nv.ERR_CHECK_FAILED


@pytest.mark.parametrize("n_req,capacity", [(1, 4), (5, 7), (40, 16), (3, 2845), (9, 1), (300, 64)])
def test_decommitter(ctx, oracle, n_req, capacity):
    from era_zkevm_test_harness_amd import native as nv
    from tests.test_oracle_ram import _bytecodes

    req, tails, words, woff = _bytecodes(oracle, n_req, seed=n_req + 5)
    mem_in = np.zeros(1, oracle.QUEUE_STATE12)
    mem_in["tail"] = synthetic.random_field_elements(9, (12,))
    mem_in["length"] = 77
    w = ctx.compute_decommitter_circuit_snapshots(req, tails, words, woff, capacity, mem_in)
    o = oracle.decommitter_build(req, tails, words, woff, capacity, mem_in)
    for what, key in ((nv.DCM_MEM_QUERIES, "mem_q"), (nv.DCM_MEM_ENC, "mem_enc"), (nv.DCM_MEM_TAILS, "mem_tails"),
                      (nv.DCM_ROUND_STATES, "round_states")):
        assert np.array_equal(w.get(what), o[key]), key
    gi = w.get(nv.DCM_INSTANCES)
    assert gi.size == o["instances"].size
    for a, b in zip(gi, o["instances"]):
        for f in a.dtype.names:
            if a[f].dtype.names:
                for g in a[f].dtype.names:
                    assert a[f][g].tobytes() == b[f][g].tobytes(), (f, g)
            else:
                assert a[f].tobytes() == b[f].tobytes(), f
    w.free()
    bad = words.copy()
    bad[-1, 3] ^= 0x10
    with pytest.raises(nv.ZkwError) as ei:
        ctx.compute_decommitter_circuit_snapshots(req, tails, bad, woff, capacity, mem_in)
    assert ei.value.code == nv.ERR_CHECK_FAILED


@pytest.mark.parametrize("n", [0, 1, 2, 17, 31, 774])
def test_linear_hasher(ctx, oracle, n):
    # n = 17: 17 * 88 = 1496 = 11 * 136 bytes -> a pure padding block; n = 774: production capacity
    q = synthetic.random_log_queries(max(n, 1), seed=n + 2)[:n]
    assert ctx.compute_linear_keccak256(q) == oracle.linear_keccak256(q)


@pytest.mark.parametrize("item_len", [0, 1, 7, 8, 9, 18, 51, 69])
def test_commit_encodings(ctx, oracle, item_len):
    enc = synthetic.random_field_elements(item_len + 5, (37, item_len))
    got = ctx.commit_variable_length_encodable_items(enc)
    for i in range(enc.shape[0]):
        assert np.array_equal(got[i], oracle.commit_var_length(enc[i]))


def test_recursion_queue(ctx, oracle):
    pi = synthetic.random_field_elements(77, (29, 4))
    tail_in = synthetic.random_field_elements(78, (12,))
    for tin in (None, tail_in):
        enc, tails = ctx.recursion_queue_push(8, pi, tin)
        oenc, otails = oracle.recursion_queue(8, pi, tin)
        assert np.array_equal(enc, oenc) and np.array_equal(tails, otails)


@pytest.mark.parametrize("n_ops,max_depth", [(1, 4), (50, 3), (3000, 40), (20000, 200)])
def test_callstack_simulator(ctx, oracle, n_ops, max_depth):
    ops, e = synthetic.callstack_trace(n_ops, seed=n_ops, max_depth=max_depth, final_unwind=n_ops != 50)
    assert np.array_equal(ctx.encode_callstack_entries(e), oracle.encode_callstack_entries(e))
    g, o = ctx.callstack_simulate(ops, e), oracle.callstack_simulate(ops, e)
    for k in o:
        assert np.array_equal(g[k], o[k]), k


def test_callstack_pop_from_empty(ctx):
    from era_zkevm_test_harness_amd import native

    ops, e = synthetic.callstack_trace(10, seed=1)
    bad = np.concatenate([ops, np.array([0], np.uint8)])
    with pytest.raises(native.ZkwError) as ei:
        ctx.callstack_simulate(bad, e)
    assert ei.value.code == native.ERR_INVALID


@pytest.mark.parametrize("kind", [0, 1, 2])
@pytest.mark.parametrize("n_req,capacity", [(0, 5), (1, 1), (40, 3), (40, 7), (300, 293), (300, 100000), (2500, 400)])  # (2500 requests: three tiles of the prefix sums)
def test_precompile_builders(ctx, oracle, kind, n_req, capacity):
    from era_zkevm_test_harness_amd import native

    req, mq = synthetic.precompile_trace(kind, n_req, seed=7 * n_req + kind, max_rounds=6)
    new = oracle.queue_push_chain_log(oracle.encode_log_queries(req))[1] if n_req else np.zeros((0, 4), np.uint64)
    mem_in = np.zeros(1, native.QUEUE_STATE12)
    mem_in["tail"] = synthetic.random_field_elements(n_req + 3, (12,))
    mem_in["length"] = 12345
    o = oracle.precompile_build(kind, req, new, mq, capacity, mem_in)
    w = ctx._precompile(kind, req, new, mq, capacity, mem_in)
    assert w.num_instances == o["instances"].size
    assert np.array_equal(w.get(native.PRC_MEM_ENC), o["mem_enc"])
    assert np.array_equal(w.get(native.PRC_MEM_TAILS), o["mem_tails"])
    gi = w.get(native.PRC_INSTANCES)
    for name in gi.dtype.names:
        assert gi[name].tobytes() == o["instances"][name].tobytes(), name
    if kind == 0:  # the cycles of the Keccak256RoundFunction circuit: padded block, reset flag, sponge state after the call
        assert w.get(native.PRC_KECCAK_ROUNDS).tobytes() == o["keccak_rounds"].tobytes()


def test_precompile_builder_rejects_inconsistent_queries(ctx, oracle):
    from era_zkevm_test_harness_amd import native

    req, mq = synthetic.precompile_trace(0, 10, seed=5)
    new = oracle.queue_push_chain_log(oracle.encode_log_queries(req))[1]
    mem_in = np.zeros(1, native.QUEUE_STATE12)
    with pytest.raises(native.ZkwError) as ei:
        ctx._precompile(0, req, new, mq[:-1], 3, mem_in)
    assert ei.value.code == native.ERR_INVALID
    bad = mq.copy()
    bad["rw_flag"][-1] = 0
    with pytest.raises(native.ZkwError) as ei:
        ctx._precompile(0, req, new, bad, 3, mem_in)
    assert ei.value.code == native.ERR_CHECK_FAILED


def _be32(limbs):
    return b"".join(int(x).to_bytes(4, "big") for x in limbs[::-1])


def _storage_application_case(oracle, n, seed):
    q, existing = synthetic.storage_application_trace(n, seed=seed)
    tails = oracle.queue_push_chain_log(oracle.encode_log_queries(q))[1] if n else np.zeros((0, 4), np.uint64)
    tree = oracle.Tree()
    rng = np.random.default_rng(seed)
    for _ in range(30):
        tree.insert_leaf(rng.bytes(32), rng.bytes(32))
    keys = [oracle.derive_final_address(q[i]) for i in range(n)]
    for i in np.nonzero(existing)[0]:
        tree.insert_leaf(keys[i], _be32(q["read_value"][i]))
    # what the tree answers before the block
    idx = np.zeros(n, np.uint64)
    paths = np.zeros((n, 256, 32), np.uint8)
    for i in range(n):
        idx[i], _, paths[i] = tree.get_leaf(keys[i])
    return q, tails, tree, idx, paths


@pytest.mark.parametrize("n,capacity", [(0, 33), (1, 33), (40, 5), (300, 33), (1500, 33)])
def test_storage_application(ctx, oracle, n, capacity):
    from era_zkevm_test_harness_amd import native

    q, tails, tree, idx, paths = _storage_application_case(oracle, n, seed=n + 11)
    root0, next0 = tree.root, tree.next_enumeration_index
    w = ctx.decompose_into_storage_application_witnesses(q, tails, idx, paths, root0, next0, capacity)
    o = oracle.storage_application_build(tree, q, tails, capacity)  # mutates the tree
    assert np.array_equal(w.get(native.SAP_DERIVED_KEYS), o["derived_keys"])
    assert np.array_equal(w.get(native.SAP_LEAF_INDEXES), o["leaf_indexes"])
    assert np.array_equal(w.get(native.SAP_ROOTS), o["roots"])
    assert np.array_equal(w.get(native.SAP_MERKLE_PATHS), o["merkle_paths"])
    gi = w.get(native.SAP_INSTANCES)
    assert gi.size == o["instances"].size
    for name in gi.dtype.names:
        assert gi[name].tobytes() == o["instances"][name].tobytes(), name


def test_storage_application_rejects_bad_proofs(ctx, oracle):
    from era_zkevm_test_harness_amd import native

    q, tails, tree, idx, paths = _storage_application_case(oracle, 50, seed=3)
    bad = paths.copy()
    bad[7, 100, 5] ^= 1
    with pytest.raises(native.ZkwError) as ei:
        ctx.decompose_into_storage_application_witnesses(q, tails, idx, bad, tree.root, tree.next_enumeration_index, 33)
    assert ei.value.code == native.ERR_CHECK_FAILED
    badq = q.copy()
    badq["read_value"][9][0] ^= 1
    with pytest.raises(native.ZkwError) as ei:
        ctx.decompose_into_storage_application_witnesses(badq, tails, idx, paths, tree.root, tree.next_enumeration_index, 33)
    assert ei.value.code == native.ERR_CHECK_FAILED
