"""Oracle (CPU restatement) of the RAM-permutation builder vs independent python restatements and the
structural invariants the reference asserts on itself (utils.rs:654-696, lib.rs:733-786)."""
import numpy as np
import pytest

from era_zkevm_test_harness_amd import synthetic
from tests.test_oracle_field_hash import _py_poseidon2, _read_constants

P = 0xFFFFFFFF00000001


def py_encode(q):
    v = [int(x) for x in q["value"]]
    b = lambda w, i: (w >> (8 * i)) & 0xFF
    return [
        int(q["timestamp"]), int(q["page"]),
        int(q["index"]) + (int(q["rw_flag"]) << 32) + (int(q["value_is_pointer"]) << 33),
        v[0] + (b(v[5], 0) << 32) + (b(v[5], 1) << 40) + (b(v[5], 2) << 48),
        v[1] + (b(v[5], 3) << 32) + (b(v[6], 0) << 40) + (b(v[6], 1) << 48),
        v[2] + (b(v[6], 2) << 32) + (b(v[6], 3) << 40) + (b(v[7], 0) << 48),
        v[3] + (b(v[7], 1) << 32) + (b(v[7], 2) << 40) + (b(v[7], 3) << 48),
        v[4],
    ]


def test_memory_query_encoding(oracle):
    q = synthetic.ram_trace(300, seed=3)
    q["value"][0] = 0xFFFFFFFF  # all-ones value: maximal limbs
    q["value"][1] = 0
    enc = oracle.encode_memory_queries(q)
    for i in range(q.size):
        assert [int(x) for x in enc[i]] == py_encode(q[i])
    assert int(enc.max()) < P


def test_full_width_queue_chain(oracle):
    rc, sh = _read_constants()
    enc = synthetic.random_field_elements(11, (6, 8))
    tails = oracle.queue_push_chain_full(enc)
    state = [0] * 12
    for i in range(6):
        state = _py_poseidon2([int(x) for x in enc[i]] + state[8:], rc, sh)
        assert [int(x) for x in tails[i]] == state
    # continuing from a tail == one long chain (split/merge invariant, lib.rs:365-377)
    t2 = oracle.queue_push_chain_full(enc[3:], tails[2])
    assert np.array_equal(t2, tails[3:])


def test_log_queue_chain(oracle):
    rc, sh = _read_constants()
    enc = synthetic.random_field_elements(12, (4, 20))
    old_t, new_t = oracle.queue_push_chain_log(enc)
    tail = [0] * 4
    for i in range(4):
        assert [int(x) for x in old_t[i]] == tail
        to_hash = [int(x) for x in enc[i]] + tail
        state = [0] * 12
        for r in range(3):
            state = _py_poseidon2(to_hash[8 * r:8 * r + 8] + state[8:], rc, sh)
        tail = state[:4]
        assert [int(x) for x in new_t[i]] == tail


def py_fs_challenges(tail_u, len_u, tail_s, len_s, n_chal, rc, sh):
    fs = [int(x) for x in tail_u] + [len_u] + [int(x) for x in tail_s] + [len_s]
    state = [0] * 11 + [len(fs)]
    while len(fs) % 8:
        fs.append(0)
    for i in range(0, len(fs), 8):
        state = _py_poseidon2(fs[i:i + 8] + state[8:], rc, sh)
    out, can_take = [], 8
    for rep in range(2):
        row = [1]
        for _ in range(n_chal - 1):
            if can_take == 0:
                state = _py_poseidon2(state, rc, sh)
                can_take = 8
            row.append(state[8 - can_take])
            can_take -= 1
        out.append(row)
    return out


@pytest.mark.parametrize("state_w,n_chal", [(12, 9), (4, 21)])
def test_fs_challenges(oracle, state_w, n_chal):
    rc, sh = _read_constants()
    tu = synthetic.random_field_elements(21, (state_w,))
    ts = synthetic.random_field_elements(22, (state_w,))
    got = oracle.fs_challenges(tu, 1234, ts, 1234, state_w, n_chal)
    exp = py_fs_challenges(tu, 1234, ts, 1234, n_chal, rc, sh)
    assert [[int(x) for x in r] for r in got] == exp
    assert got[0][0] == 1 and got[1][0] == 1


@pytest.mark.parametrize("n,width", [(1, 8), (1000, 8), (70000, 8), (66000, 20)])
def test_grand_product_chains(oracle, n, width):
    lhs = synthetic.random_field_elements(31, (n, width))
    perm = np.random.default_rng(5).permutation(n)
    rhs = lhs[perm]
    ch = synthetic.random_field_elements(32, (width + 1,))
    rc, lz, rz = oracle.grand_product_chains(lhs, rhs, ch)
    assert rc == 0 and lz[-1] == rz[-1]
    # naive python check on a prefix and the chunk boundary (2^16)
    acc = 1
    for i in range(min(n, 300)):
        term = (int(ch[width]) + sum(int(lhs[i, j]) * int(ch[j]) for j in range(width))) % P
        acc = acc * term % P
        assert int(lz[i]) == acc
    if n > 65537:
        term = (int(ch[width]) + sum(int(lhs[65536, j]) * int(ch[j]) for j in range(width))) % P
        assert int(lz[65536]) == int(lz[65535]) * term % P
    rc2, lz2, rz2 = oracle.grand_product_chains(lhs, rhs, ch, threads=4)
    assert rc2 == 0 and np.array_equal(lz, lz2) and np.array_equal(rz, rz2)
    # a non-permutation must fail the final check (utils.rs:685-696)
    bad = rhs.copy()
    bad[0, 0] = (int(bad[0, 0]) + 1) % P
    assert oracle.grand_product_chains(lhs, bad, ch)[0] == -1


@pytest.mark.parametrize("n,capacity", [(1, 4), (64, 64), (1000, 128), (1000, 1000), (8192, 2048), (5000, 1 << 20)])
def test_ram_builder_invariants(oracle, n, capacity):
    q = synthetic.ram_trace(n, seed=n)
    q["page"][: min(n, 3)] = 10  # a few bootloader-heap writes at timestamp 0
    q["timestamp"][: min(n, 3)] = 0
    q["rw_flag"][: min(n, 3)] = 1
    out = oracle.ram_build_instances(q, capacity, num_nondet=3)
    sq = out["sorted_q"]
    key = (sq["page"].astype(np.uint64) << np.uint64(32)) | sq["index"]
    assert np.all(key[1:] >= key[:-1])
    same = key[1:] == key[:-1]
    assert np.all(sq["timestamp"][1:][same] >= sq["timestamp"][:-1][same])
    inst = out["instances"]
    k = inst.size
    assert k == -(-n // capacity)
    assert inst[0]["start_flag"] == 1 and inst[-1]["completion_flag"] == 1
    assert np.all(inst["hidden_fsm_input"]["lhs_accumulator"][0] == 1)
    fo_last = inst[-1]["hidden_fsm_output"]
    assert np.array_equal(fo_last["lhs_accumulator"], fo_last["rhs_accumulator"])
    assert fo_last["current_unsorted_queue_state"]["length"] == 0
    assert np.array_equal(fo_last["current_unsorted_queue_state"]["head"], fo_last["current_unsorted_queue_state"]["tail"])
    assert np.array_equal(fo_last["current_sorted_queue_state"]["head"], out["sorted_tails"][-1])
    assert int(fo_last["num_nondeterministic_writes"]) == min(n, 3)
    for i in range(k - 1):
        a, b = inst[i]["hidden_fsm_output"], inst[i + 1]["hidden_fsm_input"]
        for f in ("lhs_accumulator", "rhs_accumulator", "current_unsorted_queue_state", "current_sorted_queue_state",
                  "previous_sorting_key", "previous_full_key", "previous_value", "num_nondeterministic_writes"):
            assert np.array_equal(a[f], b[f]), f
        assert int(inst[i]["first_item"]) == i * capacity and int(inst[i]["num_items"]) == capacity
    if n % capacity:
        assert not np.any(fo_last["previous_sorting_key"]) and not np.any(fo_last["previous_value"])
    assert np.all(inst["unsorted_queue_initial_state"]["length"] == n)
    assert np.array_equal(inst[0]["unsorted_queue_initial_state"]["tail"], out["unsorted_tails"][-1])


def test_log_and_decommit_encodings_vs_python(oracle):
    q = synthetic.random_log_queries(50, seed=4)
    enc = oracle.encode_log_queries(q)
    ext = np.arange(50, dtype=np.uint32) + np.uint32(7)
    enc_x = oracle.encode_log_queries(q, ext)
    for i in range(50):
        r = q[i]
        kb = b"".join(int(x).to_bytes(4, "little") for x in r["key"])
        ab = b"".join(int(x).to_bytes(4, "little") for x in r["address"])
        riders = kb + ab
        exp = []
        for k in range(17):
            base = int(r["read_value"][k]) if k < 8 else (int(r["written_value"][k - 8]) if k < 16 else int(r["timestamp"]))
            exp.append(base + (riders[3 * k] << 32) + (riders[3 * k + 1] << 40) + (riders[3 * k + 2] << 48))
        exp.append(int(r["tx_number_in_block"]) + (ab[19] << 32) + (int(r["aux_byte"]) << 40) + (int(r["shard_id"]) << 48))
        exp.append(int(r["rw_flag"]) + 2 * int(r["is_service"]))
        exp.append(int(r["rollback"]))
        assert [int(x) for x in enc[i]] == exp
        exp[19] += int(ext[i]) << 8
        assert [int(x) for x in enc_x[i]] == exp
    d = synthetic.random_decommit_queries(20, seed=5)
    e = oracle.encode_decommit_queries(d)
    for i in range(20):
        r = d[i]
        pb, tb = int(r["memory_page"]).to_bytes(4, "little"), int(r["timestamp"]).to_bytes(4, "little")
        h = [int(x) for x in r["hash"]]
        exp = [h[0] + (pb[0] << 32) + (pb[1] << 40) + (pb[2] << 48), h[1] + (pb[3] << 32) + (tb[0] << 40) + (tb[1] << 48),
               h[2] + (tb[2] << 32) + (tb[3] << 40) + (int(r["is_fresh"]) << 48)] + h[3:]
        assert [int(x) for x in e[i]] == exp


@pytest.mark.parametrize("n,hashes,capacity", [(1, 1, 4), (64, 5, 16), (100, 30, 16), (300, 300, 64), (50, 7, 1000)])
def test_decommit_sorter_oracle_invariants(oracle, n, hashes, capacity):
    q = synthetic.decommit_trace(n, hashes, seed=n)
    o = oracle.decommit_sorter_build(q, capacity)
    sq = o["sorted_q"]
    keys = [tuple(int(x) for x in r["hash"][::-1]) + (int(r["timestamp"]),) for r in sq]
    assert keys == sorted(keys)
    assert o["dedup_q"].size == int(q["is_fresh"].sum()) == len({tuple(r["hash"]) for r in q})
    assert np.array_equal(o["dedup_q"], sq[sq["is_fresh"] == 1])
    inst = o["instances"]
    k = inst.size
    assert k == -(-n // capacity) and inst[0]["start_flag"] == 1 and inst[-1]["completion_flag"] == 1
    assert not inst[0]["hidden_fsm_input"].tobytes().strip(b"\0")  # placeholder
    for i in range(k - 1):
        assert inst[i + 1]["hidden_fsm_input"].tobytes() == inst[i]["hidden_fsm_output"].tobytes()
    fo = inst[-1]["hidden_fsm_output"]
    assert np.array_equal(fo["lhs_accumulator"], fo["rhs_accumulator"])
    assert fo["initial_queue_state"]["length"] == 0 and fo["sorted_queue_state"]["length"] == 0
    assert np.array_equal(fo["final_queue_state"]["tail"], o["dedup_tails"][-1])
    assert int(fo["final_queue_state"]["length"]) == o["dedup_q"].size
    assert inst[-1]["final_queue_state"].tobytes() == fo["final_queue_state"].tobytes()
    if n % capacity:
        assert not fo["previous_packed_key"].any() and fo["first_encountered_timestamp"] == 0
    # appending to a non-empty deduplicated queue continues its chain
    din = np.zeros(1, oracle.QUEUE_STATE12)
    din["tail"] = synthetic.random_field_elements(3, (12,))
    din["length"] = 5
    o2 = oracle.decommit_sorter_build(q, capacity, din)
    assert np.array_equal(o2["dedup_tails"][0], oracle.queue_push_chain_full(o2["dedup_enc"][:1], din["tail"][0])[0])
    assert int(o2["instances"][-1]["final_queue_state"]["length"]) == 5 + o["dedup_q"].size


def test_decommit_sorter_rejects_inconsistent_pages(oracle):
    q = synthetic.decommit_trace(40, 3, seed=2)
    q["memory_page"][-1] += 1
    with pytest.raises(RuntimeError):
        oracle.decommit_sorter_build(q, 16)


@pytest.mark.parametrize("nf,frac,capacity", [(1, 0.0, 4), (1, 1.0, 4), (40, 0.3, 16), (200, 0.5, 64), (64, 0.0, 16), (30, 1.0, 7)])
def test_events_sorter_oracle_invariants(oracle, nf, frac, capacity):
    q = synthetic.events_trace(nf, frac, seed=nf)
    n = q.size
    o = oracle.events_sorter_build(q, capacity)
    sq = o["sorted_q"]
    assert np.all(np.diff(sq["timestamp"].astype(np.int64)) >= 0)
    n_rb = int(q["rollback"].sum())
    assert o["result_q"].size == nf - n_rb
    # net events are exactly the forwards that were never rolled back, normalised
    rolled_ts = set(int(x) for x in q["timestamp"][q["rollback"] == 1])
    keep = [r for r in q if not r["rollback"] and int(r["timestamp"]) not in rolled_ts]
    assert len(keep) == o["result_q"].size
    for a, b in zip(o["result_q"], keep):
        assert a["timestamp"] == 0 and a["rw_flag"] == 0 and a["aux_byte"] == 0 and not a["read_value"].any()
        assert np.array_equal(a["key"], b["key"]) and np.array_equal(a["written_value"], b["written_value"])
    inst = o["instances"]
    fo = inst[-1]["hidden_fsm_output"]
    assert np.array_equal(fo["lhs_accumulator"], fo["rhs_accumulator"])
    assert int(fo["final_result_queue_state"]["length"]) == o["result_q"].size
    assert inst[-1]["final_queue_state"].tobytes() == fo["final_result_queue_state"].tobytes()
    for i in range(inst.size - 1):
        a, b = inst[i]["hidden_fsm_output"], inst[i + 1]["hidden_fsm_input"]
        for f in ("lhs_accumulator", "rhs_accumulator", "initial_unsorted_queue_state", "intermediate_sorted_queue_state",
                  "final_result_queue_state", "previous_key", "previous_item"):
            assert a[f].tobytes() == b[f].tobytes(), f
    assert np.array_equal(o["unsorted_old_tails"][1:], o["unsorted_new_tails"][:-1])


def test_events_sorter_empty_and_malformed(oracle):
    o = oracle.events_sorter_build(np.zeros(0, oracle.LOG_QUERY), 16)
    assert o["instances"].size == 1 and o["instances"][0]["start_flag"] == 1 and o["instances"][0]["completion_flag"] == 1
    assert np.all(o["instances"][0]["hidden_fsm_output"]["lhs_accumulator"] == 1)
    q = synthetic.events_trace(10, 0.0, seed=3)
    q["rollback"][4] = 1  # a rollback without a forward twin
    with pytest.raises(RuntimeError):
        oracle.events_sorter_build(q, 16)


@pytest.mark.parametrize("n,capacity", [(1, 4), (100, 16), (1000, 128), (77, 1000)])
def test_log_demux_oracle(oracle, n, capacity):
    q = synthetic.mixed_log_queue(n, seed=n)
    o = oracle.log_demux_build(q, capacity)
    offs = [int(x) for x in o["out_offsets"]]
    pre = q["aux_byte"] == 3
    exp = [q[(q["aux_byte"] == 0)], q[(q["aux_byte"] == 1)], q[(q["aux_byte"] == 2)],
           q[pre & (q["address"][:, 0] == 0x8010)], q[pre & (q["address"][:, 0] == 2)], q[pre & (q["address"][:, 0] == 1)]]
    for k in range(6):
        assert np.array_equal(o["out_q"][offs[k]:offs[k + 1]], exp[k]), k
        if offs[k + 1] > offs[k]:
            old_t, new_t = oracle.queue_push_chain_log(oracle.encode_log_queries(exp[k]))
            assert np.array_equal(o["out_new_tails"][offs[k]:offs[k + 1]], new_t)
            assert np.array_equal(o["out_old_tails"][offs[k]:offs[k + 1]], old_t)
    inst = o["instances"]
    assert inst.size == -(-n // capacity)
    for i in range(inst.size - 1):
        assert inst[i + 1]["hidden_fsm_input"].tobytes() == inst[i]["hidden_fsm_output"].tobytes()
    fo = inst[-1]["hidden_fsm_output"]
    assert fo["initial_log_queue_state"]["length"] == 0
    assert [int(x) for x in fo["queue_state"]["length"]] == [offs[k + 1] - offs[k] for k in range(6)]
    assert inst[-1]["output_queue_state"].tobytes() == fo["queue_state"].tobytes()
    assert oracle.log_demux_build(np.zeros(0, oracle.LOG_QUERY), 8)["instances"].size == 1


@pytest.mark.parametrize("n,cells,capacity", [(1, 1, 4), (60, 4, 16), (400, 40, 64), (300, 300, 64), (128, 9, 32), (50, 3, 1000)])
def test_storage_sorter_oracle(oracle, n, cells, capacity):
    q = synthetic.storage_trace(n, cells, seed=n + cells)
    o = oracle.storage_sorter_build(q, capacity)
    sq, ext = o["sorted_q"], o["sorted_ext_ts"]
    assert sorted(int(x) for x in ext) == list(range(n))
    keyf = lambda r, e: tuple(int(x) for x in r["address"][::-1]) + tuple(int(x) for x in r["key"][::-1]) + (int(e),)
    ks = [keyf(sq[i], ext[i]) for i in range(n)]
    assert ks == sorted(ks)
    # replay the memory model: net effect per cell
    hist = {}
    for r in q:
        cell = (r["address"].tobytes(), r["key"].tobytes())
        h = hist.setdefault(cell, dict(initial=r["read_value"].copy(), cur=r["read_value"].copy(), depth=0, read0=False))
        if not r["rw_flag"]:
            h["read0"] |= h["depth"] == 0
        elif not r["rollback"]:
            h["depth"] += 1; h["cur"] = r["written_value"].copy()
        else:
            h["depth"] -= 1; h["cur"] = r["read_value"].copy()
    exp = []
    for cell in sorted(hist, key=lambda c: (c[0][::-1], c[1][::-1])):
        h = hist[cell]
        if h["depth"] > 0 or h["read0"]:
            exp.append((cell, h["initial"], h["cur"], not np.array_equal(h["initial"], h["cur"])))
    assert o["result_q"].size == len(exp)
    for rq, (cell, ini, cur, rw) in zip(o["result_q"], exp):
        assert rq["address"].tobytes() == cell[0] and rq["key"].tobytes() == cell[1]
        assert np.array_equal(rq["read_value"], ini) and np.array_equal(rq["written_value"], cur) and bool(rq["rw_flag"]) == rw
    inst = o["instances"]
    fo = inst[-1]["hidden_fsm_output"]
    assert np.array_equal(fo["lhs_accumulator"], fo["rhs_accumulator"])
    assert int(fo["current_final_sorted_queue_state"]["length"]) == len(exp)
    assert int(fo["cycle_idx"]) == inst.size * capacity
    for i in range(inst.size - 1):
        a, b = inst[i]["hidden_fsm_output"], inst[i + 1]["hidden_fsm_input"]
        for f in a.dtype.names:
            assert a[f].tobytes() == b[f].tobytes(), f
    dummy = oracle.storage_sorter_build(np.zeros(0, oracle.LOG_QUERY), 8)["instances"]
    assert dummy.size == 1 and dummy[0]["hidden_fsm_output"]["cycle_idx"] == 4


def _bytecodes(oracle, n_req, seed, max_words=41):
    """n_req fresh decommit requests with random bytecodes of odd length, hashes computed the decommitter's way,
    and the deduplicated queue's tails (queue started empty)."""
    r = synthetic.splitmix64(seed, n_req)
    lens = [1 + 2 * int(x % np.uint64((max_words + 1) // 2)) for x in r]
    woff = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    words = (synthetic.splitmix64(seed + 1, 4 * int(woff[-1])).view(np.uint32)).reshape(-1, 8).copy()
    req = np.zeros(n_req, oracle.DECOMMIT_QUERY)
    for k in range(n_req):
        req["hash"][k] = oracle.bytecode_hash(words[int(woff[k]):int(woff[k + 1])])
    req["timestamp"] = 5 + 2 * np.arange(n_req)
    req["memory_page"] = 16 + 8 * np.arange(n_req)
    req["is_fresh"] = 1
    req["decommitted_length"] = np.array(lens, np.uint16)
    tails = oracle.queue_push_chain_full(oracle.encode_decommit_queries(req))
    return req, tails, words, woff


@pytest.mark.parametrize("n_req,capacity", [(1, 4), (5, 7), (12, 16), (3, 1000), (9, 1)])
def test_decommitter_oracle(oracle, n_req, capacity):
    import hashlib

    req, tails, words, woff = _bytecodes(oracle, n_req, seed=n_req)
    # the versioned hash really is SHA-256 of the big-endian words (public algorithm check)
    k = n_req - 1
    code = b"".join(int(x).to_bytes(4, "big") for w in words[int(woff[k]):int(woff[k + 1])] for x in w[::-1])
    dig = hashlib.sha256(code).digest()
    assert b"".join(int(x).to_bytes(4, "big") for x in req["hash"][k][::-1])[4:] == dig[4:]
    mem_in = np.zeros(1, oracle.QUEUE_STATE12)
    mem_in["tail"] = synthetic.random_field_elements(9, (12,))
    mem_in["length"] = 1234
    o = oracle.decommitter_build(req, tails, words, woff, capacity, mem_in)
    total_rounds = sum((int(woff[i + 1] - woff[i]) + 1) // 2 for i in range(n_req))
    inst = o["instances"]
    assert inst.size == -(-total_rounds // capacity)
    assert inst[0]["start_flag"] == 1 and inst[-1]["completion_flag"] == 1 and inst["completion_flag"].sum() == 1
    assert int(inst["num_rounds"].sum()) == total_rounds and int(inst["num_words"].sum()) == int(woff[-1])
    assert int(inst["num_requests"].sum()) == n_req
    fo = inst[-1]["hidden_fsm_output"]
    assert fo["finished"] == 1 and fo["num_rounds_left"] == 0
    assert int(fo["memory_queue_state"]["length"]) == 1234 + int(woff[-1])
    assert np.array_equal(fo["memory_queue_state"]["tail"], o["mem_tails"][-1])
    assert inst[-1]["memory_queue_final_state"].tobytes() == fo["memory_queue_state"].tobytes()
    assert fo["decommittment_requests_queue_state"]["length"] == 0
    assert np.array_equal(fo["decommittment_requests_queue_state"]["head"], fo["decommittment_requests_queue_state"]["tail"])
    for i in range(inst.size - 1):
        a, b = inst[i]["hidden_fsm_output"], inst[i + 1]["hidden_fsm_input"]
        assert a.tobytes() == b.tobytes()
    assert np.array_equal(o["mem_q"]["page"][: int(woff[1])], np.full(int(woff[1]), 16))
    assert np.array_equal(o["mem_q"]["index"][: int(woff[1])], np.arange(int(woff[1])))
    bad = words.copy()
    bad[0, 0] ^= 1
    with pytest.raises(RuntimeError):
        oracle.decommitter_build(req, tails, bad, woff, capacity, mem_in)


def test_linear_hasher_oracle(oracle):
    q = synthetic.random_log_queries(5, seed=2)
    ser = oracle.serialize_l1_message(q[0])
    r = q[0]
    exp = bytes([int(r["shard_id"]), int(r["is_service"])]) + int(r["tx_number_in_block"]).to_bytes(2, "big")
    exp += b"".join(int(x).to_bytes(4, "big") for x in r["address"][::-1])
    exp += b"".join(int(x).to_bytes(4, "big") for x in r["key"][::-1])
    exp += b"".join(int(x).to_bytes(4, "big") for x in r["written_value"][::-1])
    assert ser == exp and len(ser) == 88
    full = b"".join(oracle.serialize_l1_message(x) for x in q)
    assert oracle.linear_keccak256(q) == oracle.keccak256(full)
    assert oracle.linear_keccak256(q[:0]) == oracle.keccak256(b"")


def test_public_input_commitment_oracle(oracle):
    """a20: structure of the compact form and the sponge (length specialisation, zero padding, empty item)."""
    q = synthetic.ram_trace(300, seed=5)
    o = oracle.ram_build_instances(q, 128, 0)
    inst = o["instances"]
    compact, pi = oracle.ram_public_inputs(inst)
    assert compact.shape == (3, 18) and pi.shape == (3, 4)
    assert list(compact[:, 0]) == [1, 0, 0] and list(compact[:, 1]) == [0, 0, 1]
    # the observable input is shared by every instance of the block; the observable output is () -> zero
    assert (compact[:, 2:6] == compact[0, 2:6]).all() and (compact[:, 6:10] == 0).all()
    # FSM chaining: output commitment of instance i == input commitment of instance i + 1
    assert np.array_equal(compact[:-1, 14:18], compact[1:, 10:14])
    for i in range(3):
        assert np.array_equal(pi[i], oracle.commit_var_length(compact[i]))
        assert np.array_equal(compact[i, 14:18], oracle.commit_var_length(oracle.ram_encode_fsm(inst[i]["hidden_fsm_output"])))
    # the sponge: length in state[11], overwrite absorption, zero padded tail chunk
    enc = synthetic.random_field_elements(9, (11,))
    s = np.zeros(12, np.uint64)
    s[11] = 11
    s[:8] = enc[:8]
    s = oracle.poseidon2(s)
    s[:8] = 0
    s[:3] = enc[8:]
    s = oracle.poseidon2(s)
    assert np.array_equal(oracle.commit_var_length(enc), s[:4])
    assert not oracle.commit_var_length(enc[:0]).any()
    # recursion request layout
    enc8, tails = oracle.recursion_queue(8, pi)
    assert list(enc8[1]) == [8, *pi[1], 0, 0, 0]
    assert np.array_equal(tails, oracle.queue_push_chain_full(enc8))


def test_callstack_oracle(oracle):
    ops, e = synthetic.callstack_trace(200, seed=3)
    enc = oracle.encode_callstack_entries(e)
    r = e[0]
    assert int(enc[0, 27]) == int(r["code_page"]) | int(r["pc"]) << 32 | int(r["this_shard_id"]) << 48 | int(r["is_static"]) << 56
    kernel = int(not r["this_address"][1:].any() and r["this_address"][0] < 65536)
    assert int(enc[0, 28]) >> 56 == kernel
    ln = int(r["rollback_queue_segment_length"])
    assert int(enc[0, 30]) == int(r["heap_bound"]) | (ln & 0xFFFF) << 32
    assert int(enc[0, 31]) == int(r["aux_heap_bound"]) | (ln >> 16) << 32
    o = oracle.callstack_simulate(ops, e)
    # first operation pushes onto the empty stack: 4 overwrite rounds from the zero state
    s = np.zeros(12, np.uint64)
    for rnd in range(4):
        s[:8] = enc[0, 8 * rnd:8 * rnd + 8]
        s = oracle.poseidon2(s)
        assert np.array_equal(o["round_states"][0, rnd], s)
    assert np.array_equal(o["new_state"][0], s) and not o["previous_state"][0].any() and o["depth"][0] == 1
    # every pop restores the state its push saw, and the walk ends on the empty stack
    assert o["depth"][-1] == 0 and not o["new_state"][-1].any()
    assert np.array_equal(o["previous_state"][1:], o["new_state"][:-1])
    with pytest.raises(RuntimeError):
        oracle.callstack_simulate(np.array([1, 0, 0], np.uint8), e[:1])


def _word_be(value):
    return b"".join(int(x).to_bytes(4, "big") for x in value[::-1])


@pytest.mark.parametrize("kind", [0, 1, 2])
def test_precompile_builders_oracle(oracle, kind):
    req, mq = synthetic.precompile_trace(kind, 25, seed=40 + kind)
    _, new = oracle.queue_push_chain_log(oracle.encode_log_queries(req))
    mem_in = np.zeros(1, oracle.QUEUE_STATE12)
    mem_in["tail"] = synthetic.random_field_elements(5, (12,))
    mem_in["length"] = 1000
    one = oracle.precompile_build(kind, req, new, mq, 1, mem_in)["instances"]  # an instance per round
    for cap in (3, 1000):
        inst = oracle.precompile_build(kind, req, new, mq, cap, mem_in)["instances"]
        assert inst["num_rounds"].sum() == one.size and inst["num_requests"].sum() == req.size
        assert inst["num_reads"].sum() == int((mq["rw_flag"] == 0).sum())
        assert inst[0]["start_flag"] == 1 and inst[-1]["completion_flag"] == 1 and inst["start_flag"].sum() == 1
        # FSM chaining
        assert inst["hidden_fsm_output"][:-1].tobytes() == inst["hidden_fsm_input"][1:].tobytes()
        last = inst[-1]["hidden_fsm_output"]
        assert last["memory_queue_state"]["length"] == 1000 + mq.size and last["log_queue_state"]["length"] == 0
        if kind != 2:
            assert last["completed"] == 1
    if kind == 0:
        # the sponge state after a request's last round is Keccak-256 of its input: pins the buffer / padding walk
        # against the public hash
        qpos, g = 0, 0
        for k in range(req.size):
            off, length = int(req["key"][k][0]), int(req["key"][k][1])
            n_words = (off + length - 1) // 32 - off // 32 + 1 if length else 0
            data = b"".join(_word_be(mq["value"][qpos + i]) for i in range(n_words))[off % 32:off % 32 + length]
            qpos += n_words + 1
            g += (length + 135) // 136 + (1 if length % 136 == 0 else 0)
            st = one[g - 1]["hidden_fsm_output"]["keccak_internal_state"].reshape(5, 5, 8)
            if k + 1 < req.size:  # the very last instance shows the state over an empty block instead (early termination n/a: cap 1)
                digest = b"".join(st[x, 0].tobytes() for x in range(4))
                assert digest == oracle.keccak256(data), k
    if kind == 1:
        qpos, g = 0, 0
        for k in range(req.size):
            rounds = int(req["key"][k][6])
            data = b"".join(_word_be(mq["value"][qpos + i]) for i in range(2 * rounds))
            qpos += 2 * rounds + 1
            g += rounds
            st = one[g - 1]["hidden_fsm_output"]["sha256_inner_state"]
            assert np.array_equal(st, oracle.sha256_compress_chain(data)), k
    # no requests: one dummy instance
    d = oracle.precompile_build(kind, req[:0], new[:0], mq[:0], 7, mem_in)["instances"]
    assert d.size == 1 and d[0]["start_flag"] == 1 and d[0]["completion_flag"] == 1
    assert d[0]["final_memory_state"].tobytes() == mem_in[0].tobytes()
    # a read where a write is expected
    bad = mq.copy()
    bad["rw_flag"][-1] = 0
    with pytest.raises(RuntimeError):
        oracle.precompile_build(kind, req, new, bad, 3, mem_in)


def _be32(limbs):
    return b"".join(int(x).to_bytes(4, "big") for x in limbs[::-1])


def _populated_tree(oracle, q, existing, extra=20, seed=1):
    import hashlib

    tree = oracle.Tree()
    rng = np.random.default_rng(seed)
    for _ in range(extra):  # unrelated leaves
        tree.insert_leaf(rng.bytes(32), rng.bytes(32))
    for i in np.nonzero(existing)[0]:
        tree.insert_leaf(oracle.derive_final_address(q[i]), _be32(q["read_value"][i]))
    return tree


def test_storage_application_oracle(oracle):
    import hashlib

    # the tree against hashlib: empty root and one insertion
    t0 = oracle.Tree()
    h = hashlib.blake2s(bytes(40)).digest()
    for _ in range(256):
        h = hashlib.blake2s(h + h).digest()
    assert t0.root == h and t0.next_enumeration_index == 1
    key, val = bytes(range(32)), bytes(range(100, 132))
    assert t0.insert_leaf(key, val) == 1 and t0.next_enumeration_index == 2
    idx, value, path = t0.get_leaf(key)
    assert (idx, value) == (1, val)
    cur = hashlib.blake2s((1).to_bytes(8, "big") + val).digest()
    for level in range(256):
        right = (key[level // 8] >> (level % 8)) & 1
        sib = path[level].tobytes()
        cur = hashlib.blake2s(sib + cur if right else cur + sib).digest()
    assert cur == t0.root and t0.verify_inclusion(t0.root, key, 1, val, path)

    q, existing = synthetic.storage_application_trace(60, seed=4)
    a, k = q[0]["address"], q[0]["key"]
    msg = bytes(12) + b"".join(int(x).to_bytes(4, "big") for x in a[::-1]) + _be32(k)
    assert oracle.derive_final_address(q[0]) == hashlib.blake2s(msg).digest()
    enc = oracle.state_diff_encode(q[0], b"\x07" * 32, 0x0102030405060708)
    assert enc == msg[12:32] + msg[32:] + b"\x07" * 32 + bytes([1, 2, 3, 4, 5, 6, 7, 8]) + _be32(q[0]["read_value"]) + _be32(q[0]["written_value"])

    _, tails = oracle.queue_push_chain_log(oracle.encode_log_queries(q))
    tree = _populated_tree(oracle, q, existing)
    root0, next0 = tree.root, tree.next_enumeration_index
    o = oracle.storage_application_build(tree, q, tails, 33)
    inst = o["instances"]
    # chunking: at most capacity - 1 tree operations (+1 for a trailing write) per instance
    ops = np.where(q["rw_flag"] == 1, 2, 1)
    for w in inst:
        s = ops[int(w["first_item"]):int(w["first_item"] + w["num_items"])].sum()
        assert 32 <= s <= 33 or w["completion_flag"]
    assert inst["num_items"].sum() == 60 and inst[0]["initial_root_hash"].tobytes() == root0
    assert int(inst[0]["initial_next_enumeration_counter"][0]) == next0
    first_writes = int(((q["rw_flag"] == 1) & ~existing).sum())
    assert int(inst[-1]["new_next_enumeration_counter"][0]) == next0 + first_writes == tree.next_enumeration_index
    assert inst[-1]["new_root_hash"].tobytes() == tree.root == o["roots"][-1].tobytes()
    assert inst["hidden_fsm_output"][:-1].tobytes() == inst["hidden_fsm_input"][1:].tobytes()
    # the pubdata hash is Keccak-256 over the zero-extended state diffs of the writes, in order
    data = b""
    for i in range(60):
        if q["rw_flag"][i]:
            data += oracle.state_diff_encode(q[i], o["derived_keys"][i].tobytes(), int(o["leaf_indexes"][i])).ljust(272, b"\0")
    assert inst[-1]["state_diffs_keccak256_hash"].tobytes() == oracle.keccak256(data)
    # a read of a value the tree does not hold is rejected
    bad = q.copy()
    bad["read_value"][5][0] ^= 1
    with pytest.raises(RuntimeError):
        oracle.storage_application_build(_populated_tree(oracle, q, existing), bad, tails, 33)
    # no queries: one dummy instance carrying the tree's state and the hash of nothing
    d = oracle.storage_application_build(tree, q[:0], tails[:0], 33)["instances"]
    assert d.size == 1 and d[0]["new_root_hash"].tobytes() == tree.root
    assert d[0]["state_diffs_keccak256_hash"].tobytes() == oracle.keccak256(b"")
