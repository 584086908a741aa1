"""pyoracle — TEST INFRASTRUCTURE. ctypes/numpy front-end of oracle/liboracle.so (the CPU restatement
of the reference hot path, see oracle/oracle.h). Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; the product package never does.
"""
import ctypes as C
import re
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("ZKW_ORACLE_LIB") or os.path.join(_HERE, "liboracle.so")  # override: sanitizer builds

P = 0xFFFFFFFF00000001

# numpy mirrors of include/zkw_types.h
MEM_QUERY = np.dtype(
    [("timestamp", "<u4"), ("page", "<u4"), ("index", "<u4"), ("rw_flag", "u1"), ("value_is_pointer", "u1"),
     ("_pad", "u1", (2,)), ("value", "<u4", (8,))], align=False)
assert MEM_QUERY.itemsize == 48
QUEUE_STATE12 = np.dtype([("head", "<u8", (12,)), ("tail", "<u8", (12,)), ("length", "<u4"), ("_pad", "<u4")])
assert QUEUE_STATE12.itemsize == 200
RAM_FSM = np.dtype(
    [("lhs_accumulator", "<u8", (2,)), ("rhs_accumulator", "<u8", (2,)),
     ("current_unsorted_queue_state", QUEUE_STATE12), ("current_sorted_queue_state", QUEUE_STATE12),
     ("previous_sorting_key", "<u4", (3,)), ("previous_full_key", "<u4", (2,)), ("previous_value", "<u4", (8,)),
     ("previous_is_ptr", "<u4"), ("num_nondeterministic_writes", "<u4"), ("_pad", "<u4")])
assert RAM_FSM.itemsize == 32 + 400 + 64
RAM_INSTANCE = np.dtype(
    [("start_flag", "<u4"), ("completion_flag", "<u4"), ("unsorted_queue_initial_state", QUEUE_STATE12),
     ("sorted_queue_initial_state", QUEUE_STATE12),
     ("non_deterministic_bootloader_memory_snapshot_length", "<u4"), ("_pad", "<u4"),
     ("hidden_fsm_input", RAM_FSM), ("hidden_fsm_output", RAM_FSM), ("first_item", "<u8"), ("num_items", "<u8")])
assert RAM_INSTANCE.itemsize == 8 + 400 + 8 + 2 * 496 + 16


LOG_QUERY = np.dtype(
    [("timestamp", "<u4"), ("tx_number_in_block", "<u2"), ("aux_byte", "u1"), ("shard_id", "u1"),
     ("address", "<u4", (5,)), ("key", "<u4", (8,)), ("read_value", "<u4", (8,)), ("written_value", "<u4", (8,)),
     ("rw_flag", "u1"), ("rollback", "u1"), ("is_service", "u1"), ("_pad", "u1")])
assert LOG_QUERY.itemsize == 128
DECOMMIT_QUERY = np.dtype([("hash", "<u4", (8,)), ("timestamp", "<u4"), ("memory_page", "<u4"),
                           ("decommitted_length", "<u2"), ("is_fresh", "u1"), ("_pad", "u1", (5,))])
assert DECOMMIT_QUERY.itemsize == 48
CALLSTACK_ENTRY = np.dtype(
    [("rollback_queue_head", "<u8", (4,)), ("rollback_queue_tail", "<u8", (4,)), ("rollback_queue_segment_length", "<u4"),
     ("code_address", "<u4", (5,)), ("this_address", "<u4", (5,)), ("msg_sender", "<u4", (5,)),
     ("context_u128_value", "<u4", (4,)), ("code_page", "<u4"), ("base_memory_page", "<u4"), ("ergs_remaining", "<u4"),
     ("heap_bound", "<u4"), ("aux_heap_bound", "<u4"), ("pc", "<u2"), ("sp", "<u2"), ("exception_handler_location", "<u2"),
     ("this_shard_id", "u1"), ("caller_shard_id", "u1"), ("code_shard_id", "u1"), ("is_static", "u1"),
     ("is_local_frame", "u1"), ("_pad", "u1", (1,))])
assert CALLSTACK_ENTRY.itemsize == 176


DECOMMIT_FSM = np.dtype(
    [("initial_queue_state", QUEUE_STATE12), ("sorted_queue_state", QUEUE_STATE12), ("final_queue_state", QUEUE_STATE12),
     ("lhs_accumulator", "<u8", (2,)), ("rhs_accumulator", "<u8", (2,)), ("previous_packed_key", "<u4", (9,)),
     ("first_encountered_timestamp", "<u4"), ("_pad", "<u4", (2,)), ("previous_record", DECOMMIT_QUERY)])
DECOMMIT_INSTANCE = np.dtype(
    [("start_flag", "<u4"), ("completion_flag", "<u4"), ("initial_queue_state", QUEUE_STATE12),
     ("sorted_queue_initial_state", QUEUE_STATE12), ("final_queue_state", QUEUE_STATE12),
     ("hidden_fsm_input", DECOMMIT_FSM), ("hidden_fsm_output", DECOMMIT_FSM), ("first_item", "<u8"), ("num_items", "<u8")])


QUEUE_STATE4 = np.dtype([("head", "<u8", (4,)), ("tail", "<u8", (4,)), ("length", "<u4"), ("_pad", "<u4")])
EVENTS_FSM = np.dtype(
    [("lhs_accumulator", "<u8", (2,)), ("rhs_accumulator", "<u8", (2,)), ("initial_unsorted_queue_state", QUEUE_STATE4),
     ("intermediate_sorted_queue_state", QUEUE_STATE4), ("final_result_queue_state", QUEUE_STATE4),
     ("previous_key", "<u4"), ("_pad", "<u4"), ("previous_item", LOG_QUERY)])
EVENTS_INSTANCE = np.dtype(
    [("start_flag", "<u4"), ("completion_flag", "<u4"), ("initial_log_queue_state", QUEUE_STATE4),
     ("intermediate_sorted_queue_state", QUEUE_STATE4), ("final_queue_state", QUEUE_STATE4),
     ("hidden_fsm_input", EVENTS_FSM), ("hidden_fsm_output", EVENTS_FSM), ("first_item", "<u8"), ("num_items", "<u8")])


DEMUX_FSM = np.dtype([("initial_log_queue_state", QUEUE_STATE4), ("queue_state", QUEUE_STATE4, (6,))])
DEMUX_INSTANCE = np.dtype(
    [("start_flag", "<u4"), ("completion_flag", "<u4"), ("initial_log_queue_state", QUEUE_STATE4),
     ("output_queue_state", QUEUE_STATE4, (6,)), ("hidden_fsm_input", DEMUX_FSM), ("hidden_fsm_output", DEMUX_FSM),
     ("first_item", "<u8"), ("num_items", "<u8")])


STORAGE_FSM = np.dtype(
    [("lhs_accumulator", "<u8", (2,)), ("rhs_accumulator", "<u8", (2,)), ("current_unsorted_queue_state", QUEUE_STATE4),
     ("current_intermediate_sorted_queue_state", QUEUE_STATE4), ("current_final_sorted_queue_state", QUEUE_STATE4),
     ("cycle_idx", "<u4"), ("previous_packed_key", "<u4", (13,)), ("previous_key", "<u4", (8,)), ("previous_address", "<u4", (5,)),
     ("previous_timestamp", "<u4"), ("this_cell_has_explicit_read_and_rollback_depth_zero", "<u4"),
     ("this_cell_base_value", "<u4", (8,)), ("this_cell_current_value", "<u4", (8,)), ("this_cell_current_depth", "<u4"),
     ("_pad", "<u4", (2,))])
STORAGE_INSTANCE = np.dtype(
    [("start_flag", "<u4"), ("completion_flag", "<u4"), ("shard_id_to_process", "<u4"), ("_pad", "<u4"),
     ("unsorted_log_queue_state", QUEUE_STATE4), ("intermediate_sorted_queue_state", QUEUE_STATE4),
     ("final_sorted_queue_state", QUEUE_STATE4), ("hidden_fsm_input", STORAGE_FSM), ("hidden_fsm_output", STORAGE_FSM),
     ("first_item", "<u8"), ("num_items", "<u8")])


DECOMMITTER_FSM = np.dtype(
    [("decommittment_requests_queue_state", QUEUE_STATE12), ("memory_queue_state", QUEUE_STATE12),
     ("sha256_inner_state", "<u4", (8,)), ("hash_to_compare_against", "<u4", (8,)), ("current_index", "<u4"),
     ("current_page", "<u4"), ("timestamp", "<u4"), ("num_rounds_left", "<u4"), ("length_in_bits", "<u4"),
     ("state_get_from_queue", "u1"), ("state_decommit", "u1"), ("finished", "u1"), ("_pad", "u1")])
DECOMMITTER_INSTANCE = np.dtype(
    [("start_flag", "<u4"), ("completion_flag", "<u4"), ("sorted_requests_queue_initial_state", QUEUE_STATE12),
     ("memory_queue_initial_state", QUEUE_STATE12), ("memory_queue_final_state", QUEUE_STATE12),
     ("hidden_fsm_input", DECOMMITTER_FSM), ("hidden_fsm_output", DECOMMITTER_FSM), ("first_round", "<u8"),
     ("num_rounds", "<u8"), ("first_request", "<u8"), ("num_requests", "<u8"), ("first_word", "<u8"), ("num_words", "<u8")])


def _sources_digest():
    """sha256 over everything liboracle.so is made of: oracle/*.c, *.h, the Makefile and the shared format headers in include/ (the generated
    circuit specs among them — a spec regenerated by tools/gen_*.py must rebuild the checker too; file times do not survive a snapshot)"""
    import hashlib
    h = hashlib.sha256()
    inc = os.path.join(os.path.dirname(_HERE), "include")
    files = [os.path.join(_HERE, f) for f in sorted(os.listdir(_HERE)) if f.endswith((".c", ".h")) or f == "Makefile"]
    files += [os.path.join(inc, f) for f in sorted(os.listdir(inc)) if f.endswith(".h")]
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def build(force=False):
    """Compile liboracle.so with gcc (building the checker is not using it). Keyed on the content of its sources."""
    if os.environ.get("ZKW_ORACLE_LIB"):
        return _LIB_PATH
    stamp, want = _LIB_PATH + ".sha256", _sources_digest()
    have = open(stamp).read().strip() if os.path.exists(stamp) else ""
    if force or not os.path.exists(_LIB_PATH) or have != want:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "liboracle.so"])
        with open(stamp, "w") as f:
            f.write(want + "\n")
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        u64, u64p, sz, vp = C.c_uint64, C.POINTER(C.c_uint64), C.c_size_t, C.c_void_p
        for name in ("orc_gl_add", "orc_gl_sub", "orc_gl_mul", "orc_gl_pow", "orc_gl_add_ref", "orc_gl_sub_ref", "orc_gl_mul_ref"):
            getattr(_lib, name).restype = u64
            getattr(_lib, name).argtypes = [u64, u64]
        _lib.orc_gl_inv.restype = u64
        _lib.orc_gl_inv.argtypes = [u64]
        _lib.orc_ram_build_instances.restype = C.c_int64
        _lib.orc_grand_product_chains.restype = C.c_int
        _lib.orc_grand_product_chains_mt.restype = C.c_int
    return _lib


PRECOMPILE_KECCAK256, PRECOMPILE_SHA256, PRECOMPILE_ECRECOVER = range(3)
PRECOMPILE_FSM = np.dtype(
    [("log_queue_state", QUEUE_STATE4), ("memory_queue_state", QUEUE_STATE12), ("read_precompile_call", "u1"),
     ("read_words_for_round", "u1"), ("padding_round", "u1"), ("completed", "u1"), ("timestamp_to_use_for_read", "<u4"),
     ("timestamp_to_use_for_write", "<u4"), ("input_page", "<u4"), ("input_offset", "<u4"), ("input_length", "<u4"),
     ("output_page", "<u4"), ("output_offset", "<u4"), ("num_rounds", "<u4"), ("needs_full_padding_round", "<u4"),
     ("buffer_filled", "<u4"), ("sha256_inner_state", "<u4", (8,)), ("keccak_internal_state", "u1", (200,)),
     ("buffer_bytes", "u1", (192,)), ("_pad", "<u4")])
SHA256_ROUND_RECORD = np.dtype([("block", "u1", (64,)), ("reset", "<u4"), ("state_after", "<u4", (8,)), ("_pad", "<u4")])
assert SHA256_ROUND_RECORD.itemsize == 104
KECCAK_ROUND_RECORD = np.dtype([("block", "u1", (136,)), ("reset", "u1"), ("_pad", "u1", (7,)), ("state_after", "u1", (200,))])
assert KECCAK_ROUND_RECORD.itemsize == 344
PRECOMPILE_INSTANCE = np.dtype(
    [("start_flag", "<u4"), ("completion_flag", "<u4"), ("initial_log_queue_state", QUEUE_STATE4),
     ("initial_memory_queue_state", QUEUE_STATE12), ("final_memory_state", QUEUE_STATE12),
     ("hidden_fsm_input", PRECOMPILE_FSM), ("hidden_fsm_output", PRECOMPILE_FSM), ("first_request", "<u8"),
     ("num_requests", "<u8"), ("first_read", "<u8"), ("num_reads", "<u8"), ("first_round", "<u8"), ("num_rounds", "<u8")])
assert PRECOMPILE_FSM.itemsize == 744 and PRECOMPILE_INSTANCE.itemsize == 2016


STORAGE_APPLICATION_FSM = np.dtype(
    [("next_enumeration_counter", "<u4", (2,)), ("current_root_hash", "u1", (32,)),
     ("current_storage_application_log_state", QUEUE_STATE4), ("current_diffs_keccak_accumulator_state", "u1", (200,))])
STORAGE_APPLICATION_INSTANCE = np.dtype(
    [("start_flag", "<u4"), ("completion_flag", "<u4"), ("initial_next_enumeration_counter", "<u4", (2,)),
     ("initial_root_hash", "u1", (32,)), ("shard", "<u4"), ("_pad0", "<u4"), ("storage_application_log_state", QUEUE_STATE4),
     ("new_next_enumeration_counter", "<u4", (2,)), ("new_root_hash", "u1", (32,)), ("state_diffs_keccak256_hash", "u1", (32,)),
     ("hidden_fsm_input", STORAGE_APPLICATION_FSM), ("hidden_fsm_output", STORAGE_APPLICATION_FSM), ("first_item", "<u8"),
     ("num_items", "<u8")])
assert STORAGE_APPLICATION_FSM.itemsize == 312 and STORAGE_APPLICATION_INSTANCE.itemsize == 840


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def poseidon2(state):
    s = _u64(state).copy()
    assert s.shape == (12,)
    lib().orc_poseidon2_permutation(_p(s))
    return s


def poseidon2_ref(state):
    s = _u64(state).copy()
    assert s.shape == (12,)
    lib().orc_poseidon2_permutation_ref(_p(s))
    return s


def hash_node(left, right):
    out = np.zeros(4, np.uint64)
    lib().orc_poseidon2_hash_node(_p(_u64(left)), _p(_u64(right)), _p(out))
    return out


def hash_leaf(elems):
    e = _u64(elems)
    out = np.zeros(4, np.uint64)
    lib().orc_poseidon2_hash_leaf(_p(e), C.c_size_t(e.size), _p(out))
    return out


def encode_memory_queries(q):
    q = np.ascontiguousarray(q, dtype=MEM_QUERY)
    out = np.zeros((q.size, 8), np.uint64)
    lib().orc_encode_memory_queries(_p(q), C.c_size_t(q.size), _p(out))
    return out


def encode_log_queries(q, ext_ts=None):
    q = np.ascontiguousarray(q, dtype=LOG_QUERY)
    out = np.zeros((q.size, 20), np.uint64)
    e = None if ext_ts is None else np.ascontiguousarray(ext_ts, dtype=np.uint32)
    lib().orc_encode_log_queries(_p(q), C.c_size_t(q.size), None if e is None else _p(e), _p(out))
    return out


def encode_decommit_queries(q):
    q = np.ascontiguousarray(q, dtype=DECOMMIT_QUERY)
    out = np.zeros((q.size, 8), np.uint64)
    lib().orc_encode_decommit_queries(_p(q), C.c_size_t(q.size), _p(out))
    return out


def queue_push_chain_full(enc, tail_in=None):
    enc = _u64(enc)
    n = enc.shape[0]
    tail_in = np.zeros(12, np.uint64) if tail_in is None else _u64(tail_in)
    tails = np.zeros((n, 12), np.uint64)
    lib().orc_queue_push_chain_full(_p(enc), C.c_size_t(n), _p(tail_in), _p(tails))
    return tails


def queue_push_chain_log(enc, tail_in=None):
    enc = _u64(enc)
    n = enc.shape[0]
    tail_in = np.zeros(4, np.uint64) if tail_in is None else _u64(tail_in)
    old_t = np.zeros((n, 4), np.uint64)
    new_t = np.zeros((n, 4), np.uint64)
    lib().orc_queue_push_chain_log(_p(enc), C.c_size_t(n), _p(tail_in), _p(old_t), _p(new_t))
    return old_t, new_t


def fs_challenges(tail_u, len_u, tail_s, len_s, state_w, n_chal):
    out = np.zeros((2, n_chal), np.uint64)
    lib().orc_fs_challenges(_p(_u64(tail_u)), C.c_uint32(len_u), _p(_u64(tail_s)), C.c_uint32(len_s),
                            C.c_int(state_w), C.c_int(n_chal), _p(out))
    return out


def grand_product_chains(lhs, rhs, challenges, threads=1):
    lhs, rhs, ch = _u64(lhs), _u64(rhs), _u64(challenges)
    n, w = lhs.shape
    assert rhs.shape == (n, w) and ch.shape == (w + 1,)
    lz, rz = np.zeros(n, np.uint64), np.zeros(n, np.uint64)
    if threads == 1:
        rc = lib().orc_grand_product_chains(_p(lhs), _p(rhs), C.c_size_t(n), C.c_int(w), _p(ch), _p(lz), _p(rz))
    else:
        rc = lib().orc_grand_product_chains_mt(_p(lhs), _p(rhs), C.c_size_t(n), C.c_int(w), _p(ch), _p(lz),
                                               _p(rz), C.c_int(threads))
    return rc, lz, rz


def ram_build_instances(q, capacity, num_nondet=0):
    q = np.ascontiguousarray(q, dtype=MEM_QUERY)
    n = q.size
    n_inst = (n + capacity - 1) // capacity
    out = dict(
        sorted_q=np.zeros(n, MEM_QUERY), unsorted_enc=np.zeros((n, 8), np.uint64),
        sorted_enc=np.zeros((n, 8), np.uint64), unsorted_tails=np.zeros((n, 12), np.uint64),
        sorted_tails=np.zeros((n, 12), np.uint64), challenges=np.zeros((2, 9), np.uint64),
        lhs_z=np.zeros((2, n), np.uint64), rhs_z=np.zeros((2, n), np.uint64),
        instances=np.zeros(n_inst, RAM_INSTANCE))
    rc = lib().orc_ram_build_instances(
        _p(q), C.c_size_t(n), C.c_uint32(capacity), C.c_uint32(num_nondet), _p(out["sorted_q"]),
        _p(out["unsorted_enc"]), _p(out["sorted_enc"]), _p(out["unsorted_tails"]), _p(out["sorted_tails"]),
        _p(out["challenges"]), _p(out["lhs_z"]), _p(out["rhs_z"]), _p(out["instances"]))
    if rc < 0:
        raise RuntimeError(f"orc_ram_build_instances failed: {rc}")
    assert rc == n_inst
    return out


def sha256(msg: bytes) -> bytes:
    out = C.create_string_buffer(32)
    lib().orc_sha256(msg, C.c_size_t(len(msg)), out)
    return out.raw


def keccak256(msg: bytes) -> bytes:
    out = C.create_string_buffer(32)
    lib().orc_keccak256(msg, C.c_size_t(len(msg)), out)
    return out.raw


def blake2s256(msg: bytes) -> bytes:
    out = C.create_string_buffer(32)
    lib().orc_blake2s256(msg, C.c_size_t(len(msg)), out)
    return out.raw


# ---- RAMPermutation synthesis ("zkw trace v1", include/zkw_ram_circuit_spec.h)
RC_COLS = 149
RC_ROWS_PER_CYCLE = 6


def ram_synthesize(build_out, instance_index, capacity, n_rows):
    """Fill the trace of one instance from the outputs of ram_build_instances. Returns [RC_COLS][n_rows]."""
    o = build_out
    n_total = o["sorted_q"].size
    trace = np.zeros((RC_COLS, n_rows), np.uint64)
    inst = o["instances"][instance_index:instance_index + 1]
    f = lib().orc_ram_synthesize
    f.restype = C.c_int
    rc = f(_p(inst), _p(o["sorted_q"]), _p(o["unsorted_enc"]), _p(o["sorted_enc"]), _p(o["unsorted_tails"]),
           _p(o["sorted_tails"]), _p(o["challenges"]), _p(o["lhs_z"]), _p(o["rhs_z"]), C.c_size_t(n_total),
           C.c_uint32(capacity), C.c_size_t(n_rows), _p(trace))
    if rc != 0:
        raise RuntimeError(f"orc_ram_synthesize failed: {rc}")
    first = o["instances"][0:1]  # build_out is one block: its first instance carries the shared observable input
    lib().orc_ram_fill_public_input(_p(first), _p(inst), C.c_uint32(capacity), C.c_size_t(n_rows), _p(trace))
    return trace


def ram_check(trace, capacity):
    """Satisfiability of a filled trace: (number of violated relations, code of the first one)."""
    trace = np.ascontiguousarray(trace, dtype=np.uint64)
    first_bad = C.c_uint64(0)
    f = lib().orc_ram_check
    f.restype = C.c_uint64
    bad = f(_p(trace), C.c_uint32(capacity), C.c_size_t(trace.shape[1]), C.byref(first_bad))
    v = first_bad.value
    return bad, (v >> 56, (v >> 32) & 0xFFFFFF, v & 0xFFFFFFFF)


def poseidon2_flattened(state):
    s = _u64(state)
    out = np.zeros(130, np.uint64)
    lib().orc_poseidon2_flattened(_p(s), _p(out))
    return out


def decommit_sorter_build(q, capacity, dedup_in=None):
    q = np.ascontiguousarray(q, dtype=DECOMMIT_QUERY)
    n = q.size
    n_inst = (n + capacity - 1) // capacity
    o = dict(sorted_q=np.zeros(n, DECOMMIT_QUERY), unsorted_enc=np.zeros((n, 8), np.uint64),
             sorted_enc=np.zeros((n, 8), np.uint64), unsorted_tails=np.zeros((n, 12), np.uint64),
             sorted_tails=np.zeros((n, 12), np.uint64), dedup_q=np.zeros(n, DECOMMIT_QUERY),
             dedup_enc=np.zeros((n, 8), np.uint64), dedup_tails=np.zeros((n, 12), np.uint64),
             challenges=np.zeros((2, 9), np.uint64), lhs_z=np.zeros((2, n), np.uint64), rhs_z=np.zeros((2, n), np.uint64),
             instances=np.zeros(n_inst, DECOMMIT_INSTANCE))
    nd = C.c_uint64(0)
    din = None if dedup_in is None else _p(np.ascontiguousarray(dedup_in, dtype=QUEUE_STATE12))
    f = lib().orc_decommit_sorter_build
    f.restype = C.c_int64
    rc = f(_p(q), C.c_size_t(n), C.c_uint32(capacity), din, _p(o["sorted_q"]), _p(o["unsorted_enc"]), _p(o["sorted_enc"]),
           _p(o["unsorted_tails"]), _p(o["sorted_tails"]), _p(o["dedup_q"]), _p(o["dedup_enc"]), _p(o["dedup_tails"]),
           C.byref(nd), _p(o["challenges"]), _p(o["lhs_z"]), _p(o["rhs_z"]), _p(o["instances"]))
    if rc < 0:
        raise RuntimeError(f"orc_decommit_sorter_build failed: {rc}")
    k = nd.value
    o["dedup_q"], o["dedup_enc"], o["dedup_tails"] = o["dedup_q"][:k], o["dedup_enc"][:k], o["dedup_tails"][:k]
    return o


def events_sorter_build(q, capacity, result_in=None):
    q = np.ascontiguousarray(q, dtype=LOG_QUERY)
    n = q.size
    m = max(n, 1)
    n_inst = max(1, (n + capacity - 1) // capacity)
    o = dict(sorted_q=np.zeros(m, LOG_QUERY), unsorted_enc=np.zeros((m, 20), np.uint64), sorted_enc=np.zeros((m, 20), np.uint64),
             unsorted_old_tails=np.zeros((m, 4), np.uint64), unsorted_new_tails=np.zeros((m, 4), np.uint64),
             sorted_old_tails=np.zeros((m, 4), np.uint64), sorted_new_tails=np.zeros((m, 4), np.uint64),
             result_q=np.zeros(m, LOG_QUERY), result_enc=np.zeros((m, 20), np.uint64),
             result_new_tails=np.zeros((m, 4), np.uint64), challenges=np.zeros((2, 21), np.uint64),
             lhs_z=np.zeros((2, n), np.uint64), rhs_z=np.zeros((2, n), np.uint64), instances=np.zeros(n_inst, EVENTS_INSTANCE))
    nr = C.c_uint64(0)
    rin = None if result_in is None else _p(np.ascontiguousarray(result_in, dtype=QUEUE_STATE4))
    f = lib().orc_events_sorter_build
    f.restype = C.c_int64
    rc = f(_p(q), C.c_size_t(n), C.c_uint32(capacity), rin, _p(o["sorted_q"]), _p(o["unsorted_enc"]), _p(o["sorted_enc"]),
           _p(o["unsorted_old_tails"]), _p(o["unsorted_new_tails"]), _p(o["sorted_old_tails"]), _p(o["sorted_new_tails"]),
           _p(o["result_q"]), _p(o["result_enc"]), _p(o["result_new_tails"]), C.byref(nr), _p(o["challenges"]),
           _p(o["lhs_z"]), _p(o["rhs_z"]), _p(o["instances"]))
    if rc < 0:
        raise RuntimeError(f"orc_events_sorter_build failed: {rc}")
    k = nr.value
    for key in ("sorted_q", "unsorted_enc", "sorted_enc", "unsorted_old_tails", "unsorted_new_tails", "sorted_old_tails",
                "sorted_new_tails"):
        o[key] = o[key][:n]
    for key in ("result_q", "result_enc", "result_new_tails"):
        o[key] = o[key][:k]
    return o


def log_demux_build(q, capacity):
    q = np.ascontiguousarray(q, dtype=LOG_QUERY)
    n = q.size
    m = max(n, 1)
    n_inst = max(1, (n + capacity - 1) // capacity)
    o = dict(in_enc=np.zeros((m, 20), np.uint64), in_old_tails=np.zeros((m, 4), np.uint64), in_new_tails=np.zeros((m, 4), np.uint64),
             out_q=np.zeros(m, LOG_QUERY), out_enc=np.zeros((m, 20), np.uint64), out_old_tails=np.zeros((m, 4), np.uint64),
             out_new_tails=np.zeros((m, 4), np.uint64), out_offsets=np.zeros(7, np.uint64), instances=np.zeros(n_inst, DEMUX_INSTANCE))
    params = (C.c_uint8 * 4)(0, 1, 2, 3)
    pbuf = np.zeros(1, np.dtype([("b", "u1", (4,)), ("k", "<u4"), ("s", "<u4"), ("e", "<u4")]))
    pbuf["b"] = [0, 1, 2, 3]
    pbuf["k"], pbuf["s"], pbuf["e"] = 0x8010, 2, 1
    f = lib().orc_log_demux_build
    f.restype = C.c_int64
    rc = f(_p(q), C.c_size_t(n), C.c_uint32(capacity), _p(pbuf), _p(o["in_enc"]), _p(o["in_old_tails"]), _p(o["in_new_tails"]),
           _p(o["out_q"]), _p(o["out_enc"]), _p(o["out_old_tails"]), _p(o["out_new_tails"]), _p(o["out_offsets"]), _p(o["instances"]))
    if rc < 0:
        raise RuntimeError(f"orc_log_demux_build failed: {rc}")
    r = int(o["out_offsets"][6])
    for key in ("in_enc", "in_old_tails", "in_new_tails"):
        o[key] = o[key][:n]
    for key in ("out_q", "out_enc", "out_old_tails", "out_new_tails"):
        o[key] = o[key][:r]
    return o


def storage_sorter_build(q, capacity):
    q = np.ascontiguousarray(q, dtype=LOG_QUERY)
    n = q.size
    m = max(n, 1)
    n_inst = max(1, (n + capacity - 1) // capacity)
    o = dict(sorted_q=np.zeros(m, LOG_QUERY), sorted_ext_ts=np.zeros(m, np.uint32), unsorted_enc=np.zeros((m, 20), np.uint64),
             lhs_enc=np.zeros((m, 20), np.uint64), sorted_enc=np.zeros((m, 20), np.uint64),
             unsorted_old_tails=np.zeros((m, 4), np.uint64), unsorted_new_tails=np.zeros((m, 4), np.uint64),
             sorted_old_tails=np.zeros((m, 4), np.uint64), sorted_new_tails=np.zeros((m, 4), np.uint64),
             result_q=np.zeros(m, LOG_QUERY), result_enc=np.zeros((m, 20), np.uint64), result_new_tails=np.zeros((m, 4), np.uint64),
             challenges=np.zeros((2, 21), np.uint64), lhs_z=np.zeros((2, n), np.uint64), rhs_z=np.zeros((2, n), np.uint64),
             instances=np.zeros(n_inst, STORAGE_INSTANCE))
    nr = C.c_uint64(0)
    f = lib().orc_storage_sorter_build
    f.restype = C.c_int64
    rc = f(_p(q), C.c_size_t(n), C.c_uint32(capacity), _p(o["sorted_q"]), _p(o["sorted_ext_ts"]), _p(o["unsorted_enc"]),
           _p(o["lhs_enc"]), _p(o["sorted_enc"]), _p(o["unsorted_old_tails"]), _p(o["unsorted_new_tails"]),
           _p(o["sorted_old_tails"]), _p(o["sorted_new_tails"]), _p(o["result_q"]), _p(o["result_enc"]),
           _p(o["result_new_tails"]), C.byref(nr), _p(o["challenges"]), _p(o["lhs_z"]), _p(o["rhs_z"]), _p(o["instances"]))
    if rc < 0:
        raise RuntimeError(f"orc_storage_sorter_build failed: {rc}")
    k = nr.value
    for key in ("sorted_q", "sorted_ext_ts", "unsorted_enc", "lhs_enc", "sorted_enc", "unsorted_old_tails", "unsorted_new_tails",
                "sorted_old_tails", "sorted_new_tails"):
        o[key] = o[key][:n]
    for key in ("result_q", "result_enc", "result_new_tails"):
        o[key] = o[key][:k]
    return o


def bytecode_hash(words):
    """versioned bytecode hash the decommitter checks: SHA-256 of the big-endian words, top limb = 0x0100<<16 | n_words"""
    w = np.ascontiguousarray(words, dtype=np.uint32).reshape(-1, 8)
    out = np.zeros(8, np.uint32)
    lib().orc_bytecode_hash(_p(w), C.c_size_t(w.shape[0]), C.c_uint32(0x01000000 | w.shape[0]), _p(out))
    return out


def decommitter_build(requests, dedup_tails, words, word_offsets, capacity, mem_in):
    requests = np.ascontiguousarray(requests, dtype=DECOMMIT_QUERY)
    dedup_tails = _u64(dedup_tails)
    words = np.ascontiguousarray(words, dtype=np.uint32).reshape(-1, 8)
    woff = _u64(word_offsets)
    mem_in = np.ascontiguousarray(mem_in, dtype=QUEUE_STATE12)
    total_words = int(woff[-1] - woff[0])
    total_rounds = int(sum((int(woff[k + 1] - woff[k]) + 1) // 2 for k in range(requests.size)))
    n_inst = -(-total_rounds // capacity)
    o = dict(mem_q=np.zeros(total_words, MEM_QUERY), mem_enc=np.zeros((total_words, 8), np.uint64),
             mem_tails=np.zeros((total_words, 12), np.uint64), round_states=np.zeros((total_rounds, 8), np.uint32),
             instances=np.zeros(n_inst, DECOMMITTER_INSTANCE))
    o["sha256_rounds"] = np.zeros(total_rounds, SHA256_ROUND_RECORD)  # the cycles of the CodeDecommitter circuit
    lib().orc_decommitter_set_sha256_rounds(_p(o["sha256_rounds"]))
    f = lib().orc_decommitter_build
    f.restype = C.c_int64
    rc = f(_p(requests), _p(dedup_tails), C.c_size_t(requests.size), _p(words), _p(woff), C.c_uint32(capacity), _p(mem_in),
           _p(o["mem_q"]), _p(o["mem_enc"]), _p(o["mem_tails"]), _p(o["round_states"]), _p(o["instances"]))
    lib().orc_decommitter_set_sha256_rounds(None)
    if rc < 0:
        raise RuntimeError(f"orc_decommitter_build failed: {rc}")
    o["instances"] = o["instances"][:rc]
    o.update(requests=requests, dedup_tails=dedup_tails.reshape(-1, 12), word_offsets=woff - woff[0], mem_in=mem_in)  # the queues (queue section)
    return o


def linear_keccak256(q) -> bytes:
    q = np.ascontiguousarray(q, dtype=LOG_QUERY)
    out = C.create_string_buffer(32)
    lib().orc_linear_keccak256(_p(q), C.c_size_t(q.size), out)
    return out.raw


def serialize_l1_message(q) -> bytes:
    q = np.ascontiguousarray(q, dtype=LOG_QUERY).reshape(1)
    out = C.create_string_buffer(88)
    lib().orc_serialize_l1_message(_p(q), out)
    return out.raw


def commit_var_length(enc) -> np.ndarray:
    enc = np.ascontiguousarray(enc, dtype=np.uint64)
    out = np.zeros(4, np.uint64)
    lib().orc_commit_var_length(_p(enc), C.c_size_t(enc.size), _p(out))
    return out


def ram_public_inputs(instances):
    """(compact forms [n][18], public inputs [n][4]) of consecutive RAM instances (blocks delimited by start_flag)."""
    inst = np.ascontiguousarray(instances, dtype=RAM_INSTANCE)
    compact = np.zeros((inst.size, 18), np.uint64)
    pi = np.zeros((inst.size, 4), np.uint64)
    lib().orc_ram_public_inputs(_p(inst), C.c_size_t(inst.size), _p(compact), _p(pi))
    return compact, pi


def ram_encode_fsm(fsm) -> np.ndarray:
    fsm = np.ascontiguousarray(fsm, dtype=RAM_FSM).reshape(1)
    out = np.zeros(69, np.uint64)
    f = lib().orc_ram_encode_fsm
    f.restype = C.c_size_t
    m = f(_p(fsm), _p(out))
    return out[:m]


def recursion_queue(circuit_type, pi, tail_in=None):
    pi = np.ascontiguousarray(pi, dtype=np.uint64).reshape(-1, 4)
    n = pi.shape[0]
    tin = np.zeros(12, np.uint64) if tail_in is None else np.ascontiguousarray(tail_in, dtype=np.uint64)
    enc = np.zeros((n, 8), np.uint64)
    tails = np.zeros((n, 12), np.uint64)
    lib().orc_recursion_queue(C.c_uint64(circuit_type), _p(pi), C.c_size_t(n), _p(tin), _p(enc), _p(tails))
    return enc, tails


LEAF_PARAMS = np.dtype([("circuit_type", "<u8"), ("basic_circuit_vk_commitment", "<u8", 4), ("leaf_layer_vk_commitment", "<u8", 4)])
QUEUE_TAIL12 = np.dtype([("tail", "<u8", 12), ("length", "<u4"), ("_pad", "<u4")])


def vk_commitment(cap):
    cap = np.ascontiguousarray(cap, dtype=np.uint64).reshape(-1, 4)
    out = np.zeros(4, np.uint64)
    lib().orc_vk_commitment(_p(cap), C.c_size_t(cap.shape[0]), _p(out))
    return out


def leaf_params(circuit_type, base_cap, leaf_cap):
    b = np.ascontiguousarray(base_cap, dtype=np.uint64).reshape(-1, 4)
    l = np.ascontiguousarray(leaf_cap, dtype=np.uint64).reshape(-1, 4)
    out = np.zeros(1, LEAF_PARAMS)
    lib().orc_leaf_params(C.c_uint8(circuit_type), _p(b), _p(l), C.c_size_t(b.shape[0]), _p(out))
    return out


def leaf_vks_and_params_commitment(params):
    p = np.ascontiguousarray(params, dtype=LEAF_PARAMS).reshape(13)
    out = np.zeros(4, np.uint64)
    lib().orc_leaf_vks_and_params_commitment(_p(p), _p(out))
    return out


def leaf_public_input(params, queue_state):
    p = np.ascontiguousarray(params, dtype=LEAF_PARAMS).reshape(1)
    q = np.ascontiguousarray(queue_state, dtype=QUEUE_STATE12).reshape(1)
    out = np.zeros(4, np.uint64)
    lib().orc_leaf_public_input(_p(p), _p(q), _p(out))
    return out


def node_witness(branch_circuit_type, leaf_layer_params, node_vk_commitment, chunks):
    p = np.ascontiguousarray(leaf_layer_params, dtype=LEAF_PARAMS).reshape(13)
    ch = np.ascontiguousarray(chunks, dtype=QUEUE_STATE12)
    nvk = np.ascontiguousarray(node_vk_commitment, dtype=np.uint64).reshape(4)
    st, sp, pi = np.zeros(1, QUEUE_STATE12), np.zeros(31, QUEUE_TAIL12), np.zeros(4, np.uint64)
    f = lib().orc_node_witness
    f.restype = C.c_int
    rc = f(C.c_uint8(branch_circuit_type), _p(p), _p(nvk), _p(ch), C.c_size_t(ch.size), _p(st), _p(sp), _p(pi))
    if rc != 0:
        raise ValueError("orc_node_witness: empty or non-chaining chunks")
    return st[0], sp, pi


def encode_callstack_entries(e) -> np.ndarray:
    e = np.ascontiguousarray(e, dtype=CALLSTACK_ENTRY)
    out = np.zeros((e.size, 32), np.uint64)
    lib().orc_encode_callstack_entries(_p(e), C.c_size_t(e.size), _p(out))
    return out


def callstack_simulate(is_push, pushed):
    ops = np.ascontiguousarray(is_push, dtype=np.uint8)
    e = np.ascontiguousarray(pushed, dtype=CALLSTACK_ENTRY)
    n = ops.size
    o = {"previous_state": np.zeros((n, 12), np.uint64), "new_state": np.zeros((n, 12), np.uint64),
         "depth": np.zeros(n, np.uint32), "round_states": np.zeros((n, 4, 12), np.uint64),
         "entry_index": np.zeros(n, np.uint32)}
    f = lib().orc_callstack_simulate
    f.restype = C.c_int
    rc = f(_p(ops), C.c_size_t(n), _p(e), C.c_size_t(e.size), _p(o["previous_state"]), _p(o["new_state"]),
           _p(o["depth"]), _p(o["round_states"]), _p(o["entry_index"]))
    if rc != 0:
        raise RuntimeError(f"orc_callstack_simulate failed: {rc}")
    return o


def precompile_build(kind, requests, request_tails, mem_queries, capacity, mem_in, max_instances=None):
    req = np.ascontiguousarray(requests, dtype=LOG_QUERY)
    rt = _u64(request_tails)
    mq = np.ascontiguousarray(mem_queries, dtype=MEM_QUERY)
    mem_in = np.ascontiguousarray(mem_in, dtype=QUEUE_STATE12)
    n_inst = max_instances if max_instances is not None else mq.size + req.size + 1  # upper bound on rounds
    o = dict(mem_enc=np.zeros((mq.size, 8), np.uint64), mem_tails=np.zeros((mq.size, 12), np.uint64),
             instances=np.zeros(n_inst, PRECOMPILE_INSTANCE))
    rounds = np.zeros(mq.size + req.size + 1, KECCAK_ROUND_RECORD) if kind == 0 else None
    sha_rounds = np.zeros(mq.size + req.size + 1, SHA256_ROUND_RECORD) if kind == 1 else None
    lib().orc_precompile_set_sha256_rounds(_p(sha_rounds) if sha_rounds is not None else None)
    f = lib().orc_precompile_build_ex
    f.restype = C.c_int64
    rc = f(C.c_int(kind), _p(req), _p(rt), C.c_size_t(req.size), _p(mq), C.c_size_t(mq.size), C.c_uint32(capacity),
           _p(mem_in), _p(o["mem_enc"]), _p(o["mem_tails"]), _p(o["instances"]), _p(rounds) if rounds is not None else None)
    lib().orc_precompile_set_sha256_rounds(None)
    if rc < 0:
        raise RuntimeError(f"orc_precompile_build failed: {rc}")
    o["instances"] = o["instances"][:rc]
    if sha_rounds is not None:
        o["sha256_rounds"] = sha_rounds[:int(o["instances"]["num_rounds"].sum()) if req.size else 0]
    if rounds is not None:  # one record per Keccak-f call, in the global round order
        o["keccak_rounds"] = rounds[:int(o["instances"]["num_rounds"].sum()) if req.size else 0]
    o.update(requests=req, request_tails=rt.reshape(-1, 4), mem_queries=mq, mem_in=mem_in)  # the queues (queue section)
    return o




def keccak_round_synthesize(build_out, instance_index, capacity, n_rows, public_input=None):
    """Fill the Keccak256RoundFunction trace ("zkw trace v3") of one instance from the outputs of precompile_build(0, ...):
    cycles = the instance's Keccak-f calls, then idle cycles up to `capacity`."""
    inst = build_out["instances"][instance_index]
    first, n = int(inst["first_round"]), int(inst["num_rounds"])
    recs = np.ascontiguousarray(build_out["keccak_rounds"][first:first + n])
    state_in = np.ascontiguousarray(build_out["keccak_rounds"][first - 1]["state_after"]) if first else np.zeros(200, np.uint8)
    pi = np.ascontiguousarray(public_input if public_input is not None else
                              closed_form_public_inputs(5, build_out["instances"])[1][instance_index], dtype=np.uint64)
    trace = np.zeros((nl_geometry(5)["cols"], n_rows), np.uint64)
    f = lib().orc_keccak_round_synthesize
    f.restype = C.c_int
    rc = f(_p(state_in), _p(recs) if n else None, C.c_uint32(n), C.c_uint32(capacity), _p(pi), C.c_size_t(n_rows), _p(trace))
    if rc != 0:
        raise RuntimeError(f"orc_keccak_round_synthesize failed: {rc}")
    if "requests" not in build_out:
        return trace  # (the sections the bare records imply: orc_nlq_standalone_keccak, orc_nlcf_standalone)
    return _nlcf_overlay(5, _nlq_overlay(5, trace, build_out, instance_index, capacity), build_out["instances"], instance_index, capacity, PRECOMPILE_INSTANCE)


def nl_geometry(circuit_type):
    """{cols, general, lookup width, lookups per row, total table rows, rows per cycle} of a netlist circuit's layout (3, 5, 6, 13)"""
    out = np.zeros(6, np.uint32)
    lib().orc_nl_geometry(C.c_int(circuit_type), _p(out))
    return dict(zip(("cols", "general", "width", "lookups_per_row", "table_rows", "rows_per_cycle"), (int(x) for x in out)))


NLQ_FEED = np.dtype([("en", np.uint32), ("idx", np.uint32), ("aux", np.uint32)])


class _NlqQueue(C.Structure):
    _fields_ = [("items", C.c_void_p), ("states", C.c_void_p), ("init", C.c_void_p), ("n_items", C.c_size_t)]


def nlq_geometry(circuit_type, capacity):
    """the queue section of a netlist circuit (include/zkw_netlist_queue.h): first row, rows per cycle, rows used by netlist + section,
    operations per cycle, the largest capacity that fits 2^20 rows"""
    out = np.zeros(8, np.uint64)
    lib().orc_nlq_geometry(C.c_int(circuit_type), C.c_uint32(capacity), _p(out))
    return dict(zip(("has", "first_row", "rows_per_cycle", "rows_used", "ops", "max_capacity", "queues"), (int(x) for x in out)))


def nlq_cell(circuit_type, capacity, cycle, op, block=-1, region=0, k=0):
    """(column, row) of a cell of the queue section: block -1 = the ENC block (region 0 components, 1 enc, 2 old, 3 new), block p = P2
    block p; op == ops per cycle: the QBND row (k = column)"""
    out = np.zeros(2, np.uint64)
    f = lib().orc_nlq_cell
    f.restype = C.c_int
    assert f(C.c_int(circuit_type), C.c_uint32(capacity), C.c_uint32(cycle), C.c_uint32(op), C.c_int(block), C.c_int(region), C.c_uint32(k), _p(out)) == 0
    return int(out[0]), int(out[1])


def nlcf_geometry(circuit_type, cycles):
    """the closed-form section of a netlist circuit (include/zkw_netlist_closed_form.h): first row, rows, rows used by the whole trace,
    header cells, permutations, words of the observable input / output and the hidden FSM input / output"""
    out = np.zeros(9, np.uint64)
    lib().orc_nlcf_geometry(C.c_int(circuit_type), C.c_uint32(cycles), _p(out))
    return dict(zip(("first_row", "rows", "rows_used", "header_cells", "perms", "n_oi", "n_oo", "n_fi", "n_fo"), (int(x) for x in out)))


def nlcf_cell(circuit_type, cycles, what, k=0, group=None, tie=0):
    """(column, row) of a cell of the closed-form section: what = "flag" (k = 0 start, 1 completion), "oi" / "oo" / "fi" / "fo" word k,
    "p2" (k = 130 * permutation + variable), "tie" (cell k of tie `tie` of group `group`: 0 a, 1 b, 2.. register digits), "pi" """
    code = {"flag": 0, "oi": 1, "oo": 2, "fi": 3, "fo": 4, "p2": 5, "tie": 6, "pi": 7}[what]
    if what == "tie":
        assert group < 64 and tie < 4096 and k < 16384
        k = (group << 26) | (tie << 14) | k
    out = np.zeros(2, np.uint64)
    f = lib().orc_nlcf_cell
    f.restype = C.c_int
    rc = f(C.c_int(circuit_type), C.c_uint32(cycles), C.c_int(code), C.c_uint32(k), _p(out))
    assert rc == 0, rc
    return int(out[0]), int(out[1])


def _nlcf_overlay(circuit_type, trace, instances, instance_index, cycles, dtype):
    """write the instance's closed-form section (words of its record, commitments, compact form; the PI row becomes the last
    permutation's output) below the other sections"""
    inst = np.ascontiguousarray(instances, dtype=dtype)
    f = lib().orc_nlcf_fill
    f.restype = C.c_int
    rc = f(C.c_int(circuit_type), _p(inst), C.c_size_t(instance_index), C.c_uint32(cycles), C.c_size_t(trace.shape[1]), _p(trace))
    if rc != 0:
        raise RuntimeError(f"orc_nlcf_fill failed: {rc}")
    return trace


def nlq_check(circuit_type, trace, capacity):
    """orc_nlq_check: the queue section's own checker alone (the whole-trace checkers add the netlist's and the closed-form section's)"""
    trace = np.ascontiguousarray(trace, dtype=np.uint64)
    first_bad = C.c_uint64(0)
    f = lib().orc_nlq_check
    f.restype = C.c_uint64
    bad = f(C.c_int(circuit_type), _p(trace), C.c_uint32(capacity), C.c_size_t(trace.shape[1]), C.byref(first_bad))
    v = first_bad.value if bad else 0
    return int(bad), (v >> 56, (v >> 32) & 0xFFFFFF, v & 0xFFFFFFFF)


def _nlq_overlay(circuit_type, trace, build_out, instance_index, capacity):
    """write the instance's REAL queue section (the block's request / memory queues) over the one the bare records imply"""
    inst = build_out["instances"][instance_index]
    first, n = int(inst["first_round"]), int(inst["num_rounds"])
    rounds = np.ascontiguousarray(build_out["sha256_rounds"]) if circuit_type != 5 else None
    g = nlq_geometry(circuit_type, capacity)
    feed = np.zeros((capacity, g["ops"]), NLQ_FEED)
    mem_in = np.ascontiguousarray(build_out["mem_in"], dtype=QUEUE_STATE12)
    init_tail = np.ascontiguousarray(mem_in["tail"][0], dtype=np.uint64)
    mt = np.ascontiguousarray(build_out["mem_tails"], dtype=np.uint64)
    if circuit_type == 5:
        req = np.ascontiguousarray(build_out["requests"], dtype=LOG_QUERY)
        lib().orc_keccak_queue_feed(_p(req), C.c_size_t(req.size), C.c_size_t(first), C.c_uint32(n), C.c_uint32(capacity), _p(feed))
        items0, states0 = req, np.ascontiguousarray(build_out["request_tails"], dtype=np.uint64)
        items1 = np.ascontiguousarray(build_out["mem_queries"], dtype=MEM_QUERY)
    elif circuit_type == 6:
        lib().orc_sha256_queue_feed(_p(rounds), C.c_size_t(rounds.size), C.c_size_t(first), C.c_uint32(n), C.c_uint32(capacity), _p(feed))
        items0, states0 = np.ascontiguousarray(build_out["requests"], dtype=LOG_QUERY), np.ascontiguousarray(build_out["request_tails"], dtype=np.uint64)
        items1 = np.ascontiguousarray(build_out["mem_queries"], dtype=MEM_QUERY)
    else:
        woff = np.ascontiguousarray(build_out["word_offsets"], dtype=np.uint64)
        lib().orc_code_decommitter_queue_feed(_p(rounds), C.c_size_t(rounds.size), _p(woff), C.c_size_t(first), C.c_uint32(n), C.c_uint32(capacity), _p(feed))
        items0, states0 = np.ascontiguousarray(build_out["requests"], dtype=DECOMMIT_QUERY), np.ascontiguousarray(build_out["dedup_tails"], dtype=np.uint64)
        items1 = np.ascontiguousarray(build_out["mem_q"], dtype=MEM_QUERY)
    queues = (_NlqQueue * 2)(_NlqQueue(items0.ctypes.data, states0.ctypes.data, None, items0.size),
                             _NlqQueue(items1.ctypes.data, mt.ctypes.data, init_tail.ctypes.data, items1.size))
    f = lib().orc_nlq_synthesize
    f.restype = C.c_int
    rc = f(C.c_int(circuit_type), C.c_uint32(capacity), _p(feed), queues, C.c_size_t(trace.shape[1]), _p(trace))
    if rc != 0:
        raise RuntimeError(f"orc_nlq_synthesize failed: {rc}")
    return trace


# ---- ECRecover circuit (type 7): oracle/ecrecover_circuit.c ------------------------------------------------------------------
def ec_geometry(capacity):
    """the EC section of an ECRecover trace: first row, rows per cycle, rows used by the whole trace, tape values per cycle"""
    out = np.zeros(8, np.uint64)
    lib().orc_ec_geometry(C.c_uint32(capacity), _p(out))
    return dict(zip(("first_row", "rows_per_cycle", "rows_used", "tape_per_cycle", "types", "runs"), (int(x) for x in out)))


def ec_inputs(h, v, r, s):
    """the 128 input bytes of a cycle: value bytes (little end first) of the reads hash, v, r, s (256-bit integers)"""
    return np.frombuffer(b"".join(int(x).to_bytes(32, "little") for x in (h, v, r, s)), np.uint8).copy()


def ec_eval_cycle(inputs):
    """the value tape of one cycle (oracle/ecrecover_eval.c: the oracle's own evaluator); raises when the inputs have no witness"""
    inp = np.ascontiguousarray(inputs, dtype=np.uint8)
    assert inp.size == 128
    tape = np.zeros(ec_geometry(1)["tape_per_cycle"], np.uint64)
    f = lib().orc_ec_eval_cycle
    f.restype = C.c_uint32
    rc = f(_p(inp), _p(tape))
    if rc:
        raise RuntimeError(f"ec_eval_cycle: no witness (run {rc >> 24}, instance {(rc >> 12) & 0xFFF}, item {(rc & 0xFFF) - 1})")
    return tape


def ec_outputs(tape):
    """(ok, mask, the 64 key bytes Q.x || Q.y big end first) of an evaluated tape"""
    out = np.zeros(66, np.uint8)
    lib().orc_ec_outputs(_p(np.ascontiguousarray(tape, dtype=np.uint64)), _p(out))
    return int(out[0]), int(out[1]), out[2:].tobytes()


def ec_cell(capacity, cycle, what, k):
    """(column, trace row) of a cell of the EC section: what = "glob" (home of global k), "in" (home of input byte k), "key" (home of key
    byte k), "mul" (column 0 of the k-th MUL row of the first double-and-add segment)"""
    out = np.zeros(2, np.uint32)
    f = lib().orc_ec_cell
    f.restype = C.c_int
    assert f(C.c_int({"glob": 0, "in": 1, "key": 2, "mul": 3}[what]), C.c_uint32(k), _p(out)) == 0
    g = ec_geometry(capacity)
    return int(out[1]), g["first_row"] + cycle * g["rows_per_cycle"] + int(out[0])


def ecrecover_inputs_of(build_out, instance_index, capacity):
    """[capacity][128]: the value bytes of the four reads of every cycle of an instance (zeros for the idle cycles)"""
    inst = build_out["instances"][instance_index]
    first, n = int(inst["first_round"]), int(inst["num_rounds"])
    mq = build_out["mem_queries"]
    inp = np.zeros((capacity, 128), np.uint8)
    for c in range(n):
        q = mq[6 * (first + c):6 * (first + c) + 4]
        inp[c] = np.ascontiguousarray(q["value"]).view(np.uint8).reshape(-1)  # limb 0 first, little endian = value byte x
    return inp


def ecrecover_synthesize(build_out, instance_index, capacity, n_rows, public_input=None):
    """Fill the ECRecover trace of one instance from the outputs of precompile_build(2, ...): the Keccak-f netlist over the recovered keys,
    the queue section, the EC section"""
    inst = build_out["instances"][instance_index]
    first, n = int(inst["first_round"]), int(inst["num_rounds"])
    inp = ecrecover_inputs_of(build_out, instance_index, capacity)
    pi = np.ascontiguousarray(public_input if public_input is not None else
                              closed_form_public_inputs(7, build_out["instances"])[1][instance_index], dtype=np.uint64)
    trace = np.zeros((nl_geometry(7)["cols"], n_rows), np.uint64)
    f = lib().orc_ecrecover_synthesize
    f.restype = C.c_int
    rc = f(_p(inp), C.c_uint32(n), C.c_uint32(capacity), _p(pi), C.c_size_t(n_rows), _p(trace))
    if rc != 0:
        raise RuntimeError(f"orc_ecrecover_synthesize failed: {rc}")
    feed = np.zeros((capacity, 7), NLQ_FEED)
    lib().orc_ecrecover_queue_feed(C.c_size_t(first), C.c_uint32(n), C.c_uint32(capacity), _p(feed))
    req = np.ascontiguousarray(build_out["requests"], dtype=LOG_QUERY)
    states0 = np.ascontiguousarray(build_out["request_tails"], dtype=np.uint64)
    mq = np.ascontiguousarray(build_out["mem_queries"], dtype=MEM_QUERY)
    mt = np.ascontiguousarray(build_out["mem_tails"], dtype=np.uint64)
    init_tail = np.ascontiguousarray(np.ascontiguousarray(build_out["mem_in"], dtype=QUEUE_STATE12)["tail"][0], dtype=np.uint64)
    queues = (_NlqQueue * 2)(_NlqQueue(req.ctypes.data, states0.ctypes.data, None, req.size), _NlqQueue(mq.ctypes.data, mt.ctypes.data, init_tail.ctypes.data, mq.size))
    g = lib().orc_nlq_synthesize
    g.restype = C.c_int
    rc = g(C.c_int(7), C.c_uint32(capacity), _p(feed), queues, C.c_size_t(n_rows), _p(trace))
    if rc != 0:
        raise RuntimeError(f"orc_nlq_synthesize failed: {rc}")
    return _nlcf_overlay(7, trace, build_out["instances"], instance_index, capacity, PRECOMPILE_INSTANCE)


def ecrecover_check(trace, capacity):
    t = np.ascontiguousarray(trace, dtype=np.uint64)
    first = C.c_uint64(0)
    f = lib().orc_ecrecover_check
    f.restype = C.c_uint64
    bad = f(_p(t), C.c_uint32(capacity), C.c_size_t(t.shape[1]), C.byref(first))
    v = first.value
    return int(bad), (v >> 56, (v >> 32) & 0xFFFFFF, v & 0xFFFFFFFF)


def nl_spec_state(circuit_type):
    """elements of a netlist circuit's cycle state (what the boundary rows hold): 256 for the SHA-256 circuits (the chaining value's 64
    nibbles, then zeros: the steps of a compression pass more between them), 200 bytes for the Keccak circuits"""
    f = lib().orc_nl_state
    f.restype = C.c_uint32
    return int(f(C.c_int(circuit_type)))


def nl_slots_per_cycle(circuit_type):
    """lookup slots of one cycle (padding included: every slot is counted by the multiplicity column)"""
    out = np.zeros(1, np.uint32)
    lib().orc_nl_slots_per_cycle(C.c_int(circuit_type), _p(out))
    return int(out[0])


def __getattr__(name):  # SC_COLS, KC_ROWS_PER_CYCLE, ...: read from the compiled specs, not restated here
    m = re.fullmatch(r"(SC|DC|KC|LH|SA)_(COLS|ROWS_PER_CYCLE)", name)
    if not m:
        raise AttributeError(name)
    g = nl_geometry({"SC": 6, "DC": 3, "KC": 5, "LH": 13, "SA": 10}[m.group(1)])
    return g["cols"] if m.group(2) == "COLS" else g["rows_per_cycle"]


def sha256_round_synthesize_raw(state_in, records, capacity, n_rows, public_input):
    """orc_sha256_round_synthesize: cycles = the given records, then idle cycles up to `capacity`; raises when the netlist's
    chaining state differs from a record's state_after"""
    recs = np.ascontiguousarray(records, dtype=SHA256_ROUND_RECORD)
    st = np.ascontiguousarray(state_in, dtype=np.uint8)
    pi = np.ascontiguousarray(public_input, dtype=np.uint64)
    trace = np.zeros((nl_geometry(6)["cols"], n_rows), np.uint64)
    f = lib().orc_sha256_round_synthesize
    f.restype = C.c_int
    rc = f(_p(st), _p(recs) if recs.size else None, C.c_uint32(recs.size), C.c_uint32(capacity), _p(pi), C.c_size_t(n_rows), _p(trace))
    if rc != 0:
        raise RuntimeError(f"orc_sha256_round_synthesize failed: {rc}")
    return trace


def sha256_round_synthesize(build_out, instance_index, capacity, n_rows, public_input=None):
    """Fill the Sha256RoundFunction trace of one instance from the outputs of precompile_build(1, ...)"""
    inst = build_out["instances"][instance_index]
    first, n = int(inst["first_round"]), int(inst["num_rounds"])
    state_in = (np.ascontiguousarray(build_out["sha256_rounds"][first - 1]["state_after"]).view(np.uint8) if first
                else np.zeros(32, np.uint8))
    pi = (public_input if public_input is not None else closed_form_public_inputs(6, build_out["instances"])[1][instance_index])
    trace = sha256_round_synthesize_raw(state_in, build_out["sha256_rounds"][first:first + n], capacity, n_rows, pi)
    if "requests" not in build_out:
        return trace
    return _nlcf_overlay(6, _nlq_overlay(6, trace, build_out, instance_index, capacity), build_out["instances"], instance_index, capacity, PRECOMPILE_INSTANCE)




def code_decommitter_synthesize(build_out, instance_index, capacity, n_rows, public_input=None):
    """Fill the CodeDecommitter trace (the SHA-256 netlist at 18 lookups per row) of one instance of decommitter_build(...)"""
    inst = build_out["instances"][instance_index]
    first, n = int(inst["first_round"]), int(inst["num_rounds"])
    recs = np.ascontiguousarray(build_out["sha256_rounds"][first:first + n])
    state_in = (np.ascontiguousarray(build_out["sha256_rounds"][first - 1]["state_after"]).view(np.uint8) if first
                else np.zeros(32, np.uint8))
    pi = np.ascontiguousarray(public_input if public_input is not None else
                              closed_form_public_inputs(3, build_out["instances"])[1][instance_index], dtype=np.uint64)
    trace = np.zeros((nl_geometry(3)["cols"], n_rows), np.uint64)
    f = lib().orc_code_decommitter_round_synthesize
    f.restype = C.c_int
    rc = f(_p(np.ascontiguousarray(state_in)), _p(recs) if n else None, C.c_uint32(n), C.c_uint32(capacity), _p(pi), C.c_size_t(n_rows), _p(trace))
    if rc != 0:
        raise RuntimeError(f"orc_code_decommitter_round_synthesize failed: {rc}")
    if "dedup_tails" not in build_out:
        return trace
    return _nlcf_overlay(3, _nlq_overlay(3, trace, build_out, instance_index, capacity), build_out["instances"], instance_index, capacity, DECOMMITTER_INSTANCE)


def code_decommitter_check(trace, capacity):
    trace = np.ascontiguousarray(trace, dtype=np.uint64)
    first_bad = C.c_uint64(0)
    f = lib().orc_code_decommitter_round_check
    f.restype = C.c_uint64
    bad = f(_p(trace), C.c_uint32(capacity), C.c_size_t(trace.shape[1]), C.byref(first_bad))
    v = first_bad.value
    return bad, (v >> 56, (v >> 32) & 0xFFFFFF, v & 0xFFFFFFFF)


def sha256_round_check(trace, capacity):
    trace = np.ascontiguousarray(trace, dtype=np.uint64)
    first_bad = C.c_uint64(0)
    f = lib().orc_sha256_round_check
    f.restype = C.c_uint64
    bad = f(_p(trace), C.c_uint32(capacity), C.c_size_t(trace.shape[1]), C.byref(first_bad))
    v = first_bad.value
    return bad, (v >> 56, (v >> 32) & 0xFFFFFF, v & 0xFFFFFFFF)


def linear_hasher_cycles(capacity):
    """cycles (Keccak-f calls) of the LinearHasher circuit for `capacity` 88-byte messages: ZKW_LINEAR_HASHER_CYCLES"""
    return capacity * 88 // 136 + 1


def linear_hasher_queue_state(messages, head=None):
    """the QueueState4 of a queue that holds exactly `messages` pushed over `head` (zeros): tail = the state after the last push (= head
    when there is none), length = their number — what the circuit's closed-form section ties its pops to"""
    q = np.ascontiguousarray(messages, dtype=LOG_QUERY)
    st = np.zeros(1, QUEUE_STATE4)
    h = np.zeros(4, np.uint64) if head is None else np.asarray(head, dtype=np.uint64)
    st["head"][0] = h
    st["tail"][0] = queue_push_chain_log(encode_log_queries(q), h)[1][-1] if q.size else h
    st["length"] = q.size
    return st


def linear_hasher_synthesize(messages, queue_state, capacity, n_rows):
    """LinearHasher (type 13): (trace, instance record, public input) for the net L2 -> L1 messages of a block"""
    q = np.ascontiguousarray(messages, dtype=LOG_QUERY)
    assert q.size <= capacity
    recs = np.zeros(q.size * 88 // 136 + 1, KECCAK_ROUND_RECORD)
    f = lib().orc_linear_hasher_rounds
    f.restype = C.c_size_t
    n = f(_p(q), C.c_size_t(q.size), _p(recs))
    assert n == recs.size
    inst = np.zeros(1, LINEAR_HASHER_INSTANCE)
    inst["start_flag"] = inst["completion_flag"] = 1
    inst["queue_state"] = queue_state
    inst["keccak256_hash"] = recs["state_after"][-1][:32]
    pi = closed_form_public_inputs(13, inst)[1][0]
    cycles = linear_hasher_cycles(capacity)
    trace = np.zeros((nl_geometry(13)["cols"], n_rows), np.uint64)
    g = lib().orc_linear_hasher_round_synthesize
    g.restype = C.c_int
    rc = g(_p(np.zeros(200, np.uint8)), _p(recs), C.c_uint32(n), C.c_uint32(cycles), _p(pi), C.c_size_t(n_rows), _p(trace))
    if rc != 0:
        raise RuntimeError(f"orc_linear_hasher_round_synthesize failed: {rc}")
    h = lib().orc_linear_hasher_queue_section  # the pops of the messages as Poseidon2 rows (the queue section)
    h.restype = C.c_int
    head = np.ascontiguousarray(inst["queue_state"]["head"][0], dtype=np.uint64)
    rc = h(_p(q) if q.size else None, C.c_size_t(q.size), _p(head), C.c_uint32(cycles), C.c_size_t(n_rows), _p(trace))
    if rc != 0:
        raise RuntimeError(f"orc_linear_hasher_queue_section failed: {rc}")
    return _nlcf_overlay(13, trace, inst, 0, cycles, LINEAR_HASHER_INSTANCE), inst, pi


def linear_hasher_check(trace, cycles):
    trace = np.ascontiguousarray(trace, dtype=np.uint64)
    first_bad = C.c_uint64(0)
    f = lib().orc_linear_hasher_round_check
    f.restype = C.c_uint64
    bad = f(_p(trace), C.c_uint32(cycles), C.c_size_t(trace.shape[1]), C.byref(first_bad))
    v = first_bad.value
    return bad, (v >> 56, (v >> 32) & 0xFFFFFF, v & 0xFFFFFFFF)


def keccak_round_check(trace, capacity):
    trace = np.ascontiguousarray(trace, dtype=np.uint64)
    first_bad = C.c_uint64(0)
    f = lib().orc_keccak_round_check
    f.restype = C.c_uint64
    bad = f(_p(trace), C.c_uint32(capacity), C.c_size_t(trace.shape[1]), C.byref(first_bad))
    v = first_bad.value
    return bad, (v >> 56, (v >> 32) & 0xFFFFFF, v & 0xFFFFFFFF)


def sha256_compress_chain(data: bytes) -> np.ndarray:
    """SHA-256 state after compressing the 64-byte blocks of `data` (no padding)."""
    st = (C.c_uint32 * 8).in_dll(lib(), "ORC_SHA256_IV")
    state = np.array(list(st), np.uint32)
    for i in range(0, len(data), 64):
        lib().orc_sha256_compress(_p(state), C.c_char_p(data[i:i + 64]))
    return state


class Tree:
    """InMemoryStorageTree<256, 32, 8, Blake2s256, ZkSyncStorageLeaf> (src/witness/tree/mod.rs:113-384)."""

    def __init__(self):
        f = lib().orc_tree_new
        f.restype = C.c_void_p
        self.handle = C.c_void_p(f())

    def __del__(self):
        try:
            lib().orc_tree_free(self.handle)
        except Exception:
            pass

    @property
    def root(self) -> bytes:
        out = C.create_string_buffer(32)
        lib().orc_tree_root(self.handle, out)
        return out.raw

    @property
    def next_enumeration_index(self) -> int:
        f = lib().orc_tree_next_enumeration_index
        f.restype = C.c_uint64
        return f(self.handle)

    def get_leaf(self, key: bytes):
        idx = C.c_uint64(0)
        value = C.create_string_buffer(32)
        path = np.zeros((256, 32), np.uint8)
        lib().orc_tree_get_leaf(self.handle, C.c_char_p(key), C.byref(idx), value, _p(path))
        return idx.value, value.raw, path

    def insert_leaf(self, key: bytes, value: bytes) -> int:
        f = lib().orc_tree_insert_leaf
        f.restype = C.c_uint64
        return f(self.handle, C.c_char_p(key), C.c_char_p(value), None)

    def verify_inclusion(self, root: bytes, key: bytes, index: int, value: bytes, path) -> bool:
        path = np.ascontiguousarray(path, dtype=np.uint8)
        return bool(lib().orc_tree_verify_inclusion(C.c_char_p(root), C.c_char_p(key), C.c_uint64(index), C.c_char_p(value), _p(path)))


def derive_final_address(q) -> bytes:
    q = np.ascontiguousarray(q, dtype=LOG_QUERY).reshape(1)
    out = C.create_string_buffer(32)
    lib().orc_derive_final_address(_p(q), out)
    return out.raw


def state_diff_encode(q, derived_key: bytes, enumeration_index: int) -> bytes:
    q = np.ascontiguousarray(q, dtype=LOG_QUERY).reshape(1)
    out = C.create_string_buffer(156)
    lib().orc_state_diff_encode(_p(q), C.c_char_p(derived_key), C.c_uint64(enumeration_index), out)
    return out.raw


def storage_application_build(tree, queries, query_tails, capacity):
    """decompose_into_storage_application_witnesses on `tree` (mutated)."""
    q = np.ascontiguousarray(queries, dtype=LOG_QUERY)
    qt = _u64(query_tails)
    n = q.size
    o = dict(derived_keys=np.zeros((n, 32), np.uint8), merkle_paths=np.zeros((n, 256, 32), np.uint8),
             leaf_indexes=np.zeros(n, np.uint64), roots=np.zeros((n, 32), np.uint8),
             instances=np.zeros(n + 1, STORAGE_APPLICATION_INSTANCE))
    f = lib().orc_storage_application_build
    f.restype = C.c_int64
    rc = f(tree.handle, _p(q), _p(qt), C.c_size_t(n), C.c_uint32(capacity), _p(o["derived_keys"]), _p(o["merkle_paths"]),
           _p(o["leaf_indexes"]), _p(o["roots"]), _p(o["instances"]))
    if rc < 0:
        raise RuntimeError(f"orc_storage_application_build failed: {rc}")
    o["instances"] = o["instances"][:rc]
    return o


SA_CYCLES_PER_WALK = 257


def storage_application_synthesize(build_out, queries, instance_index, capacity, n_rows, public_input=None):
    """Fill the StorageApplication trace ("zkw trace v4": Blake2s Merkle walks, tools/gen_storage_application_circuit.py) of one
    instance from the outputs of storage_application_build: one walk per read, two per write, idle cycles up to `capacity` walks."""
    inst = build_out["instances"][instance_index]
    first, n = int(inst["first_item"]), int(inst["num_items"])
    q = np.ascontiguousarray(np.asarray(queries, dtype=LOG_QUERY)[first:first + n])
    keys = np.ascontiguousarray(build_out["derived_keys"][first:first + n])
    paths = np.ascontiguousarray(build_out["merkle_paths"][first:first + n])
    idx = np.ascontiguousarray(build_out["leaf_indexes"][first:first + n])
    ctr = inst["initial_next_enumeration_counter"] if inst["start_flag"] else inst["hidden_fsm_input"]["next_enumeration_counter"]
    next_index = int(ctr[0]) | (int(ctr[1]) << 32)
    pi = np.ascontiguousarray(public_input if public_input is not None else
                              closed_form_public_inputs(10, build_out["instances"])[1][instance_index], dtype=np.uint64)
    trace = np.zeros((nl_geometry(10)["cols"], n_rows), np.uint64)
    f = lib().orc_storage_application_synthesize
    f.restype = C.c_int
    rc = f(_p(q) if n else None, C.c_size_t(n), _p(keys) if n else None, _p(paths) if n else None, _p(idx) if n else None,
           C.c_uint64(next_index), C.c_uint32(capacity), _p(pi), C.c_size_t(n_rows), _p(trace),
           _p(np.ascontiguousarray(inst["hidden_fsm_output"]["current_root_hash"], dtype=np.uint8)))  # (an instance without walks carries its root)
    if rc != 0:
        raise RuntimeError(f"orc_storage_application_synthesize failed: {rc}")
    return _nlcf_overlay(10, trace, build_out["instances"], instance_index, capacity * SA_CYCLES_PER_WALK, STORAGE_APPLICATION_INSTANCE)


def storage_application_check(trace, capacity):
    trace = np.ascontiguousarray(trace, dtype=np.uint64)
    first_bad = C.c_uint64(0)
    f = lib().orc_storage_application_check
    f.restype = C.c_uint64
    bad = f(_p(trace), C.c_uint32(capacity), C.c_size_t(trace.shape[1]), C.byref(first_bad))
    v = first_bad.value
    return bad, (v >> 56, (v >> 32) & 0xFFFFFF, v & 0xFFFFFFFF)


DS_COLS = 149


def decommit_sorter_synthesize(build_out, instance_index, capacity, n_rows):
    """Fill the CodeDecommittmentsSorter trace of one instance from the outputs of decommit_sorter_build."""
    o = build_out
    trace = np.zeros((DS_COLS, n_rows), np.uint64)
    inst = o["instances"][instance_index:instance_index + 1]
    f = lib().orc_decommit_sorter_synthesize
    f.restype = C.c_int
    rc = f(_p(inst), _p(o["sorted_q"]), _p(o["unsorted_enc"]), _p(o["sorted_enc"]), _p(o["challenges"]), None, C.c_uint32(0),
           None, C.c_uint32(capacity), C.c_size_t(n_rows), _p(trace))
    if rc != 0:
        raise RuntimeError(f"orc_decommit_sorter_synthesize failed: {rc}")
    first = o["instances"][0:1]  # one block: its first instance carries the shared observable input
    lib().orc_ds_fill_closed_form(_p(first), _p(inst), C.c_uint32(capacity), C.c_size_t(n_rows), _p(trace))
    return trace


def decommit_sorter_check(trace, capacity):
    trace = np.ascontiguousarray(trace, dtype=np.uint64)
    first_bad = C.c_uint64(0)
    f = lib().orc_decommit_sorter_check
    f.restype = C.c_uint64
    bad = f(_p(trace), C.c_uint32(capacity), C.c_size_t(trace.shape[1]), C.byref(first_bad))
    v = first_bad.value
    return bad, (v >> 56, (v >> 32) & 0xFFFFFF, v & 0xFFFFFFFF)


def decommit_sorter_public_inputs(instances):
    """(compact forms [n][18], public inputs [n][4]) of the instances of one decommit-sorter block."""
    inst = np.ascontiguousarray(instances, dtype=DECOMMIT_INSTANCE)
    compact = np.zeros((inst.size, 18), np.uint64)
    pi = np.zeros((inst.size, 4), np.uint64)
    lib().orc_ds_public_inputs(_p(inst), C.c_size_t(inst.size), _p(compact), _p(pi))
    return compact, pi


ES_COLS = 139


def events_sorter_synthesize(build_out, instance_index, capacity, n_rows):
    """Fill the EventsSorter / L1MessagesSorter trace of one instance from the outputs of events_sorter_build."""
    o = build_out
    trace = np.zeros((ES_COLS, n_rows), np.uint64)
    inst = o["instances"][instance_index:instance_index + 1]
    f = lib().orc_events_sorter_synthesize
    f.restype = C.c_int
    rc = f(_p(inst), _p(o["sorted_q"]), _p(o["unsorted_enc"]), _p(o["sorted_enc"]), _p(o["challenges"]), None, C.c_uint32(0),
           None, C.c_uint32(capacity), C.c_size_t(n_rows), _p(trace))
    if rc != 0:
        raise RuntimeError(f"orc_events_sorter_synthesize failed: {rc}")
    first = o["instances"][0:1]  # one block: its first instance carries the shared observable input
    lib().orc_es_fill_closed_form(_p(first), _p(inst), C.c_uint32(capacity), C.c_size_t(n_rows), _p(trace))
    return trace


def events_sorter_check(trace, capacity):
    trace = np.ascontiguousarray(trace, dtype=np.uint64)
    first_bad = C.c_uint64(0)
    f = lib().orc_events_sorter_check
    f.restype = C.c_uint64
    bad = f(_p(trace), C.c_uint32(capacity), C.c_size_t(trace.shape[1]), C.byref(first_bad))
    v = first_bad.value
    return bad, (v >> 56, (v >> 32) & 0xFFFFFF, v & 0xFFFFFFFF)


LD_COLS = 151


def log_demux_synthesize(build_out, instance_index, capacity, n_rows):
    """Fill the LogDemuxer trace of one instance from the outputs of log_demux_build."""
    o = build_out
    trace = np.zeros((LD_COLS, n_rows), np.uint64)
    inst = o["instances"][instance_index:instance_index + 1]
    enc = o["in_enc"] if o["in_enc"].size else np.zeros((1, 20), np.uint64)
    f = lib().orc_log_demux_synthesize
    f.restype = C.c_int
    rc = f(_p(inst), _p(enc), None, C.c_uint32(capacity), C.c_size_t(n_rows), _p(trace))
    if rc != 0:
        raise RuntimeError(f"orc_log_demux_synthesize failed: {rc}")
    first = o["instances"][0:1]  # one block: its first instance carries the shared observable input
    lib().orc_ld_fill_closed_form(_p(first), _p(inst), C.c_uint32(capacity), C.c_size_t(n_rows), _p(trace))
    return trace


def log_demux_check(trace, capacity):
    trace = np.ascontiguousarray(trace, dtype=np.uint64)
    first_bad = C.c_uint64(0)
    f = lib().orc_log_demux_check
    f.restype = C.c_uint64
    bad = f(_p(trace), C.c_uint32(capacity), C.c_size_t(trace.shape[1]), C.byref(first_bad))
    v = first_bad.value
    return bad, (v >> 56, (v >> 32) & 0xFFFFFF, v & 0xFFFFFFFF)


SS_COLS = 149


def storage_sorter_synthesize(build_out, instance_index, capacity, n_rows):
    """Fill the StorageSorter trace of one instance from the outputs of storage_sorter_build."""
    o = build_out
    trace = np.zeros((SS_COLS, n_rows), np.uint64)
    inst = o["instances"][instance_index:instance_index + 1]
    f = lib().orc_storage_sorter_synthesize
    f.restype = C.c_int
    rc = f(_p(inst), _p(o["unsorted_enc"]), _p(o["sorted_enc"]), _p(o["challenges"]), None, C.c_uint32(capacity), C.c_size_t(n_rows),
           _p(trace))
    if rc != 0:
        raise RuntimeError(f"orc_storage_sorter_synthesize failed: {rc}")
    first = o["instances"][0:1]  # one block: its first instance carries the shared observable input
    lib().orc_ss_fill_closed_form(_p(first), _p(inst), C.c_uint32(capacity), C.c_size_t(n_rows), _p(trace))
    return trace


def storage_sorter_check(trace, capacity):
    trace = np.ascontiguousarray(trace, dtype=np.uint64)
    first_bad = C.c_uint64(0)
    f = lib().orc_storage_sorter_check
    f.restype = C.c_uint64
    bad = f(_p(trace), C.c_uint32(capacity), C.c_size_t(trace.shape[1]), C.byref(first_bad))
    v = first_bad.value
    return bad, (v >> 56, (v >> 32) & 0xFFFFFF, v & 0xFFFFFFFF)


def _public_inputs(fn, instances):
    instances = np.ascontiguousarray(instances)
    n = instances.size
    compact, pi = np.zeros((n, 18), np.uint64), np.zeros((n, 4), np.uint64)
    getattr(lib(), fn)(_p(instances), C.c_size_t(n), _p(compact), _p(pi))
    return compact, pi


def log_demux_public_inputs(instances):
    """(compact closed-form inputs [n][18], public inputs [n][4]) of a block's LogDemuxer instances"""
    return _public_inputs("orc_log_demux_public_inputs", instances)


def events_sorter_public_inputs(instances):
    return _public_inputs("orc_events_sorter_public_inputs", instances)


def storage_sorter_public_inputs(instances):
    return _public_inputs("orc_storage_sorter_public_inputs", instances)


LINEAR_HASHER_INSTANCE = np.dtype([("start_flag", "<u4"), ("completion_flag", "<u4"), ("queue_state", QUEUE_STATE4),
                                   ("keccak256_hash", "u1", (32,))])
CLOSED_FORM_RECORD = {3: DECOMMITTER_INSTANCE, 5: PRECOMPILE_INSTANCE, 6: PRECOMPILE_INSTANCE, 7: PRECOMPILE_INSTANCE,
                      10: STORAGE_APPLICATION_INSTANCE, 13: LINEAR_HASHER_INSTANCE}


def closed_form_public_inputs(circuit_type, instances):
    """orc_closed_form_public_inputs: (compact [n][18], public inputs [n][4]) of the instance records of circuit type
    3, 5, 6, 7, 10 or 13"""
    instances = np.ascontiguousarray(instances, dtype=CLOSED_FORM_RECORD[circuit_type])
    n = instances.size
    compact, pi = np.zeros((n, 18), np.uint64), np.zeros((n, 4), np.uint64)
    f = lib().orc_closed_form_public_inputs
    f.restype = C.c_int
    rc = f(C.c_int(circuit_type), _p(instances), C.c_size_t(n), _p(compact), _p(pi))
    if rc != 0:
        raise ValueError(f"orc_closed_form_public_inputs: unknown circuit type {circuit_type}")
    return compact, pi


# ---- MainVM instance slicing (include/zkw_types.h: zkw_vm_instance & co) ------------------------------------------------
VM_NUM_STREAMS = 8
(VMS_MEMORY, VMS_STORAGE_QUERIES, VMS_REFUNDS, VMS_DECOMMIT_REQUESTS, VMS_ROLLBACK_TAILS_FOR_NEW_FRAMES, VMS_CALLSTACK_VALUES,
 VMS_ROLLBACK_HEAD_SEGMENTS, VMS_NEW_FRAMES) = range(8)
STORAGE_LOG_DETAILED_STATE = np.dtype([("forward_tail", "<u8", (4,)), ("rollback_head", "<u8", (4,)), ("rollback_tail", "<u8", (4,)),
                                       ("forward_length", "<u4"), ("rollback_length", "<u4")])
VM_AUX_PARAMETERS = np.dtype(
    [("callstack_state", "<u8", (12,)), ("decommittment_queue_state", QUEUE_STATE12), ("memory_queue_state", QUEUE_STATE12),
     ("storage_log_queue_state", QUEUE_STATE4), ("current_frame_rollback_queue_tail", "<u8", (4,)),
     ("current_frame_rollback_queue_head", "<u8", (4,)), ("current_frame_rollback_queue_segment_length", "<u4"), ("_pad", "<u4")])
VM_INSTANCE = np.dtype(
    [("start_flag", "<u4"), ("completion_flag", "<u4"), ("cycle_from", "<u4"), ("cycle_to", "<u4"), ("snapshot_initial", "<u4"),
     ("snapshot_final", "<u4"), ("range", "<u8", (8, 2)), ("first_memory_read", "<u8"), ("num_memory_reads", "<u8"),
     ("first_memory_write", "<u8"), ("num_memory_writes", "<u8"), ("auxilary_initial_parameters", VM_AUX_PARAMETERS),
     ("auxilary_final_parameters", VM_AUX_PARAMETERS), ("rollback_queue_tail_for_block", "<u8", (4,)),
     ("memory_queue_initial_tail", "<u8", (12,)), ("memory_queue_initial_length", "<u4"), ("decommitment_queue_initial_length", "<u4"),
     ("decommitment_queue_initial_tail", "<u8", (12,)), ("memory_queue_final_state", QUEUE_STATE12),
     ("decommitment_queue_final_state", QUEUE_STATE12), ("log_queue_final_state", QUEUE_STATE4)])


class VmTracerStreams(C.Structure):
    """zkw_vm_tracer_streams"""
    _fields_ = [("snapshot_cycles", C.c_void_p), ("n_snapshots", C.c_size_t), ("stream_cycles", C.c_void_p * 8), ("stream_len", C.c_size_t * 8),
                ("vm_memory_queries", C.c_void_p), ("memory_queue_tails", C.c_void_p), ("decommit_state_cycles", C.c_void_p),
                ("decommit_queue_tails", C.c_void_p), ("n_decommit_states", C.c_size_t), ("callstack_sponge_cycles", C.c_void_p),
                ("callstack_sponge_states", C.c_void_p), ("n_callstack_sponges", C.c_size_t), ("storage_log_state_cycles", C.c_void_p),
                ("storage_log_states", C.c_void_p), ("n_storage_log_states", C.c_size_t), ("global_end_of_storage_log", C.c_uint64 * 4)]


def _vm_streams_struct(t, keep):
    """t: dict with snapshot_cycles, stream_cycles (list of 8 arrays), vm_memory_queries, memory_queue_tails, decommit_state_cycles,
    decommit_queue_tails, callstack_sponge_cycles, callstack_sponge_states, storage_log_state_cycles, storage_log_states,
    global_end_of_storage_log"""
    def arr(a, dtype):
        a = np.ascontiguousarray(a, dtype=dtype)
        keep.append(a)
        return a
    s = VmTracerStreams()
    sc = arr(t["snapshot_cycles"], np.uint32)
    s.snapshot_cycles, s.n_snapshots = sc.ctypes.data, sc.size
    for k in range(8):
        c = arr(t["stream_cycles"][k], np.uint32)
        s.stream_cycles[k] = c.ctypes.data if c.size else None
        s.stream_len[k] = c.size
    for name, dtype, cnt in (("vm_memory_queries", MEM_QUERY, None), ("memory_queue_tails", np.uint64, None),
                             ("decommit_state_cycles", np.uint32, "n_decommit_states"), ("decommit_queue_tails", np.uint64, None),
                             ("callstack_sponge_cycles", np.uint32, "n_callstack_sponges"), ("callstack_sponge_states", np.uint64, None),
                             ("storage_log_state_cycles", np.uint32, "n_storage_log_states"), ("storage_log_states", STORAGE_LOG_DETAILED_STATE, None)):
        a = arr(t[name], dtype)
        setattr(s, name, a.ctypes.data if a.size else None)
        if cnt:
            setattr(s, cnt, a.size)
    for j in range(4):
        s.global_end_of_storage_log[j] = int(t["global_end_of_storage_log"][j])
    return s


def vm_slice_instances(tracer):
    """orc_vm_slice_instances: (instances, memory read indices, memory write indices)"""
    keep = []
    st = _vm_streams_struct(tracer, keep)
    n_inst, n_mem = st.n_snapshots - 1, st.stream_len[0]
    inst = np.zeros(n_inst, VM_INSTANCE)
    ri, wi = np.zeros(max(n_mem, 1), np.uint32), np.zeros(max(n_mem, 1), np.uint32)
    nr, nw = C.c_uint64(0), C.c_uint64(0)
    rc = lib().orc_vm_slice_instances(C.byref(st), _p(inst), _p(ri), _p(wi), C.byref(nr), C.byref(nw))
    if rc != 0:
        raise RuntimeError(f"orc_vm_slice_instances failed: {rc}")
    return inst, ri[:nr.value], wi[:nw.value]


# ---- the setup side as field elements (commit.c)
def root_of_unity(log_n):
    f = lib().orc_root_of_unity
    f.restype = C.c_uint64
    return int(f(C.c_uint32(log_n)))


def gl_powers(base, n):
    out = np.zeros(n, np.uint64)
    lib().orc_gl_powers(C.c_uint64(base), C.c_size_t(n), _p(out))
    return out


def ntt(values, inverse=False):
    """orc_ntt over the last axis (natural order in and out)"""
    v = np.ascontiguousarray(values, dtype=np.uint64).copy()
    v2 = v.reshape(-1, v.shape[-1])
    log_n = int(v2.shape[1]).bit_length() - 1
    for r in range(v2.shape[0]):
        row = np.ascontiguousarray(v2[r])
        lib().orc_ntt(_p(row), C.c_uint32(log_n), C.c_int(1 if inverse else 0))
        v2[r] = row
    return v


def lde(values, lde_factor=2):
    v = np.ascontiguousarray(values, dtype=np.uint64)
    log_n = int(v.shape[1]).bit_length() - 1
    out = np.zeros((lde_factor,) + v.shape, np.uint64)
    lib().orc_lde(_p(v), C.c_uint32(log_n), C.c_size_t(v.shape[0]), C.c_uint32(lde_factor), _p(out))
    return out


def poly_eval(coeffs, x):
    c = np.ascontiguousarray(coeffs, dtype=np.uint64)
    f = lib().orc_poly_eval
    f.restype = C.c_uint64
    return int(f(_p(c), C.c_size_t(c.size), C.c_uint64(x)))


def merkle_tree_with_cap(leaf_cols, cap_size=16):
    """every level, leaves first: [nodes][4]; the cap is the last cap_size nodes"""
    v = np.ascontiguousarray(leaf_cols, dtype=np.uint64)
    n_sets, n_cols, n = v.shape
    tree = np.zeros((2 * n_sets * n - cap_size, 4), np.uint64)
    lib().orc_merkle_tree_with_cap(_p(v), C.c_size_t(n_sets), C.c_size_t(n_cols), C.c_size_t(n), C.c_uint32(cap_size), _p(tree))
    return tree
