"""ctypes binding of libzkw.so — the C ABI declared in include/zkw.h.

This is the only way Python reaches the engine; there is no Python/CPU fallback: importing works
anywhere (so that the symbol/ABI checks can run without a GPU), but `Context()` raises when the
library is missing or no gfx950 device is usable.
"""
import ctypes as C
import os

# zkw_block_run overlaps the builders of a block on ~14 HIP streams; HIP's default of 4 hardware queues per priority class
# would serialise them, more than 8 oversubscribes the chip's queue slots (csrc/zkw_block.hip). Read by the HIP runtime
# at device initialisation, i.e. before the first torch.cuda call.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np  # noqa: E402

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ZKW_LIB", os.path.join(_HERE, "libzkw.so"))  # ZKW_LIB: an alternative build (A/B measurements)

OK, ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_OOM, ERR_CHECK_FAILED = 0, -1, -2, -3, -4, -5
PTR_HOST, PTR_DEVICE = 0, 1
(RAM_SORTED_QUERIES, RAM_UNSORTED_ENC, RAM_SORTED_ENC, RAM_UNSORTED_TAILS, RAM_SORTED_TAILS, RAM_CHALLENGES,
 RAM_LHS_Z, RAM_RHS_Z, RAM_INSTANCES, RAM_COMPACT_FORMS, RAM_PUBLIC_INPUTS) = range(11)

# every symbol include/zkw.h declares: (name, restype, argtypes)
_vp, _sz, _u32, _u64p, _int = C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p, C.c_int
SYMBOLS = [
    ("zkw_process_init", _int, []),
    ("zkw_create", _vp, [_int]),
    ("zkw_destroy", None, [_vp]),
    ("zkw_last_error", C.c_char_p, []),
    ("zkw_set_stream", _int, [_vp, _vp]),
    ("zkw_set_chain_stream", _int, [_vp, _vp]),
    ("zkw_set_pointer_mode", _int, [_vp, _int]),
    ("zkw_synchronize", _int, [_vp]),
    ("zkw_set_chain_form", _int, [_vp, _int]),
    ("zkw_set_netlist_fill_form", _int, [_vp, _int]),
    ("zkw_set_chain_service", _int, [_vp, _int]),
    ("zkw_set_chain_tag", _int, [_vp, _int]),
    ("zkw_chain_service_expect", _int, [_int, _int]),
    ("zkw_buffer_alloc", _int, [_vp, _int, C.c_size_t, C.POINTER(C.c_void_p)]),
    ("zkw_buffer_free", None, [_int, _vp]),
    ("zkw_stream_acquire", _int, [_vp, C.POINTER(C.c_void_p)]),
    ("zkw_stream_release", None, [_vp, _vp]),
    ("zkw_trim_caches", None, []),
    ("zkw_block_linear_hasher_instance", _int, [_vp, _vp]),
    ("zkw_setup_copy_permutation", _int, [C.c_uint8, C.c_uint32, _sz, _vp, _vp]),
    ("zkw_check_copy_permutation", _int, [_vp, _vp, _sz, _vp, C.c_uint32, _vp, _vp]),
    ("zkw_ntt", _int, [_vp, _vp, _vp, _u32, _sz, _int]),
    ("zkw_lde", _int, [_vp, _vp, _u32, _sz, _u32, _vp]),
    ("zkw_merkle_tree_words", _sz, [_sz, _u32]),
    ("zkw_merkle_tree_with_cap", _int, [_vp, _vp, _sz, _sz, _sz, _u32, _vp, _vp]),
    ("zkw_setup_lookup_tables", _int, [C.c_uint8, _sz, _vp, _vp]),
    ("zkw_setup_num_columns", _int, [C.c_uint8, _vp]),
    ("zkw_setup_columns", _int, [_vp, C.c_uint8, _u32, _u32, _vp]),
    ("zkw_setup_commit", _int, [_vp, C.c_uint8, _u32, _u32, _u32, _u32, _vp]),
    ("zkw_setup_row_selectors", _int, [C.c_uint8, C.c_uint32, _sz, _vp]),
    ("zkw_vm_trace_build", _int, [_vp, _vp, _sz, _vp, _sz, _vp, _sz, C.POINTER(_vp)]),
    ("zkw_vm_trace_count", _sz, [_vp, _int]),
    ("zkw_vm_trace_ptr", _vp, [_vp, _int]),
    ("zkw_vm_trace_get", _int, [_vp, _int, _vp, _sz]),
    ("zkw_vm_trace_info", _int, [_vp, _vp]),
    ("zkw_vm_trace_streams", _int, [_vp, _vp]),
    ("zkw_vm_trace_free", None, [_vp]),
    ("zkw_recursion_queue_split", _int, [_vp, _sz, C.c_uint32, _vp, _sz, _vp]),
    ("zkw_vk_commitment", _int, [_vp, _vp, _sz, _vp]),
    ("zkw_compute_leaf_params", _int, [_vp, C.c_uint8, _vp, _vp, _sz, _vp]),
    ("zkw_leaf_vks_and_params_commitment", _int, [_vp, _vp, _vp]),
    ("zkw_create_leaf_witnesses", _int, [_vp, _vp, _vp, _sz, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    ("zkw_create_node_witnesses", _int, [_vp, C.c_uint8, _vp, _vp, _vp, _sz, _vp, _vp, _vp, _sz, _vp]),
    ("zkw_closed_form_public_inputs", _int, [_vp, C.c_uint8, _vp, C.c_size_t, _vp, _vp]),
    ("zkw_version", C.c_char_p, []),
    ("zkw_circuit_geometry_of", _int, [C.c_uint8, _vp]),
    ("zkw_circuit_layout_of", _int, [C.c_uint8, _u32, _vp]),
    ("zkw_circuit_fill_bytes", _int, [C.c_uint8, _u32, _sz, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    ("zkw_profile_enable", _int, [_vp, _int]),
    ("zkw_profile_reset", _int, [_vp]),
    ("zkw_profile_get", _int, [_vp, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    ("zkw_profile_names", _int, [_vp, C.c_char_p, _sz]),
    ("zkw_encode_memory_queries", _int, [_vp, _vp, _sz, _u64p]),
    ("zkw_queue_push_chain_full", _int, [_vp, _u64p, _sz, _u64p, _u64p]),
    ("zkw_queue_push_chain_full_batch", _int, [_vp, _u64p, _u64p, _sz, _u64p, _u64p]),
    ("zkw_encode_log_queries", _int, [_vp, _vp, _sz, _vp, _u64p]),
    ("zkw_encode_decommit_queries", _int, [_vp, _vp, _sz, _u64p]),
    ("zkw_queue_push_chain_log_batch", _int, [_vp, _u64p, _u64p, _sz, _u64p, _u64p, _u64p]),
    ("zkw_queue_push_chain_log", _int, [_vp, _u64p, _sz, _u64p, _u64p, _u64p]),
    ("zkw_fs_challenges", _int, [_vp, _u64p, _u32, _u64p, _u32, _int, _int, _u64p]),
    ("zkw_grand_product_chains", _int, [_vp, _u64p, _u64p, _sz, _int, _u64p, _int, _u64p, _u64p]),
    ("zkw_ram_build_instances", _int, [_vp, _vp, _sz, _u32, _u32, C.POINTER(_vp)]),
    ("zkw_ram_build_instances_batch", _int, [_vp, _vp, _u64p, _sz, _u32, _vp, C.POINTER(_vp)]),
    ("zkw_ram_witness_num_instances", _sz, [_vp]),
    ("zkw_ram_witness_num_items", _sz, [_vp]),
    ("zkw_ram_witness_bytes", _sz, [_vp, _int]),
    ("zkw_ram_witness_device_ptr", _vp, [_vp, _int]),
    ("zkw_ram_witness_get", _int, [_vp, _int, _vp, _sz]),
    ("zkw_ram_witness_free", None, [_vp]),
    ("zkw_decommit_sorter_build", _int, [_vp, _vp, _sz, _u32, _vp, C.POINTER(_vp)]),
    ("zkw_decommit_sorter_prepare", _int, [_vp, _vp, _sz, _u32, _vp, C.POINTER(_vp)]),
    ("zkw_decommit_sorter_finish", _int, [_vp, _vp]),
    ("zkw_decommit_witness_num_instances", _sz, [_vp]),
    ("zkw_decommit_witness_num_dedup", _sz, [_vp]),
    ("zkw_decommit_witness_bytes", _sz, [_vp, _int]),
    ("zkw_decommit_witness_device_ptr", _vp, [_vp, _int]),
    ("zkw_decommit_witness_get", _int, [_vp, _int, _vp, _sz]),
    ("zkw_decommit_witness_free", None, [_vp]),
    ("zkw_events_sorter_build", _int, [_vp, _vp, _sz, _u32, _vp, C.POINTER(_vp)]),
    ("zkw_events_witness_num_instances", _sz, [_vp]),
    ("zkw_events_witness_num_results", _sz, [_vp]),
    ("zkw_events_witness_bytes", _sz, [_vp, _int]),
    ("zkw_events_witness_device_ptr", _vp, [_vp, _int]),
    ("zkw_events_witness_get", _int, [_vp, _int, _vp, _sz]),
    ("zkw_events_witness_free", None, [_vp]),
    ("zkw_log_demux_build", _int, [_vp, _vp, _sz, _u32, _vp, C.POINTER(_vp)]),
    ("zkw_demux_witness_num_instances", _sz, [_vp]),
    ("zkw_demux_witness_bytes", _sz, [_vp, _int]),
    ("zkw_demux_witness_device_ptr", _vp, [_vp, _int]),
    ("zkw_demux_witness_get", _int, [_vp, _int, _vp, _sz]),
    ("zkw_demux_witness_free", None, [_vp]),
    ("zkw_storage_sorter_build", _int, [_vp, _vp, _sz, _u32, C.POINTER(_vp)]),
    ("zkw_storage_witness_num_instances", _sz, [_vp]),
    ("zkw_storage_witness_num_results", _sz, [_vp]),
    ("zkw_storage_witness_bytes", _sz, [_vp, _int]),
    ("zkw_storage_witness_device_ptr", _vp, [_vp, _int]),
    ("zkw_storage_witness_get", _int, [_vp, _int, _vp, _sz]),
    ("zkw_storage_witness_free", None, [_vp]),
    ("zkw_decommitter_build", _int, [_vp, _vp, _u64p, _sz, _vp, _u64p, _u32, _vp, C.POINTER(_vp)]),
    ("zkw_decommitter_build_with_tails", _int, [_vp, _vp, _u64p, _sz, _vp, _u64p, _u32, _vp, _vp, C.POINTER(_vp)]),
    ("zkw_decommitter_memory_queries", _int, [_vp, _vp, _sz, _vp, _u64p, _vp]),
    ("zkw_decommitter_witness_num_instances", _sz, [_vp]),
    ("zkw_decommitter_witness_bytes", _sz, [_vp, _int]),
    ("zkw_decommitter_witness_device_ptr", _vp, [_vp, _int]),
    ("zkw_decommitter_witness_get", _int, [_vp, _int, _vp, _sz]),
    ("zkw_decommitter_witness_free", None, [_vp]),
    ("zkw_linear_keccak256", _int, [_vp, _vp, _sz, _vp]),
    ("zkw_decommit_sorter_synthesize", _int, [_vp, _vp, _sz, _sz, _vp, _sz]),
    ("zkw_decommit_sorter_check_satisfied", _int, [_vp, _vp, _sz, C.c_uint32, _vp, _vp]),
    ("zkw_events_sorter_synthesize", _int, [_vp, _vp, _sz, _sz, _vp, _sz]),
    ("zkw_events_sorter_check_satisfied", _int, [_vp, _vp, _sz, C.c_uint32, _vp, _vp]),
    ("zkw_log_demux_synthesize", _int, [_vp, _vp, _sz, _sz, _vp, _sz]),
    ("zkw_storage_sorter_synthesize", _int, [_vp, _vp, _sz, _sz, _vp, _sz]),
    ("zkw_storage_sorter_check_satisfied", _int, [_vp, _vp, _sz, C.c_uint32, _vp, _vp]),
    ("zkw_log_demux_check_satisfied", _int, [_vp, _vp, _sz, C.c_uint32, _vp, _vp]),
    ("zkw_precompile_closed_forms", _int, [_vp, _vp, _vp, _vp]),
    ("zkw_keccak_round_synthesize", _int, [_vp, _vp, _sz, _sz, _vp, _sz]),
    ("zkw_sha256_round_synthesize", _int, [_vp, _vp, _sz, _sz, _vp, _sz]),
    ("zkw_code_decommitter_synthesize", _int, [_vp, _vp, _sz, _sz, _vp, _sz]),
    ("zkw_code_decommitter_check_satisfied", _int, [_vp, _vp, _sz, C.c_uint32, _vp, _vp]),
    ("zkw_sha256_round_check_satisfied", _int, [_vp, _vp, _sz, C.c_uint32, _vp, _vp]),
    ("zkw_linear_hasher_synthesize", _int, [_vp, _vp, _sz, _vp, C.c_uint32, _vp, _sz, _vp, _vp]),
    ("zkw_linear_hasher_synthesize_batch", _int, [_vp, _vp, _vp, _sz, _vp, C.c_uint32, _vp, _sz, _vp, _vp]),
    ("zkw_linear_hasher_synthesize_batch_with_tails", _int, [_vp, _vp, _vp, _sz, _vp, _vp, C.c_uint32, _vp, _sz, _vp, _vp]),
    ("zkw_linear_hasher_check_satisfied", _int, [_vp, _vp, _sz, C.c_uint32, _vp, _vp]),
    ("zkw_keccak_round_check_satisfied", _int, [_vp, _vp, _sz, C.c_uint32, _vp, _vp]),
    ("zkw_ecrecover_synthesize", _int, [_vp, _vp, _sz, _sz, _vp, _sz]),
    ("zkw_ecrecover_synthesize_multi", _int, [_vp, _vp, _sz, _vp, _sz]),
    ("zkw_ecrecover_check_satisfied", _int, [_vp, _vp, _sz, C.c_uint32, _vp, _vp]),
    ("zkw_storage_application_build", _int, [_vp, _vp, _vp, _sz, _vp, _vp, _vp, C.c_uint64, C.c_uint32, _vp]),
    ("zkw_storage_application_witness_num_instances", _sz, [_vp]),
    ("zkw_storage_application_witness_bytes", _sz, [_vp, _int]),
    ("zkw_storage_application_witness_device_ptr", _vp, [_vp, _int]),
    ("zkw_storage_application_witness_get", _int, [_vp, _int, _vp, _sz]),
    ("zkw_storage_application_witness_free", None, [_vp]),
    ("zkw_storage_application_synthesize", _int, [_vp, _vp, _sz, _sz, _vp, _sz]),
    ("zkw_storage_application_check_satisfied", _int, [_vp, _vp, _sz, C.c_uint32, _vp, _vp]),
    ("zkw_precompile_build", _int, [_vp, _int, _vp, _vp, _sz, _vp, _sz, C.c_uint32, _vp, _vp]),
    ("zkw_precompile_build_with_tails", _int, [_vp, _int, _vp, _vp, _sz, _vp, _sz, C.c_uint32, _vp, _vp, _vp]),
    ("zkw_precompile_witness_num_instances", _sz, [_vp]),
    ("zkw_precompile_witness_num_rounds", _sz, [_vp]),
    ("zkw_precompile_witness_bytes", _sz, [_vp, _int]),
    ("zkw_precompile_witness_device_ptr", _vp, [_vp, _int]),
    ("zkw_precompile_witness_get", _int, [_vp, _int, _vp, _sz]),
    ("zkw_precompile_witness_free", None, [_vp]),
    ("zkw_encode_callstack_entries", _int, [_vp, _vp, _sz, _vp]),
    ("zkw_callstack_simulate", _int, [_vp, _vp, _sz, _vp, _sz, _vp, _vp, _vp, _vp, _vp]),
    ("zkw_commit_encodings", _int, [_vp, _vp, _sz, C.c_uint32, _vp]),
    ("zkw_encode_recursion_requests", _int, [_vp, C.c_uint64, _vp, _sz, _vp]),
    ("zkw_trace_create", _int, [_vp, _sz, _sz, C.POINTER(_vp)]),
    ("zkw_trace_create_with_columns", _int, [_vp, _sz, _sz, _sz, C.POINTER(_vp)]),
    ("zkw_trace_free", None, [_vp]),
    ("zkw_trace_num_rows", _sz, [_vp]),
    ("zkw_trace_num_cols", _sz, [_vp]),
    ("zkw_trace_num_slots", _sz, [_vp]),
    ("zkw_trace_device_ptr", _vp, [_vp, _sz]),
    ("zkw_trace_get", _int, [_vp, _sz, _u32, _u32, _vp]),
    ("zkw_ram_synthesize", _int, [_vp, _vp, _sz, _sz, _vp, _sz]),
    ("zkw_ram_check_satisfied", _int, [_vp, _vp, _sz, _u32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    ("zkw_synthesize", _int, [_vp, C.c_uint8, _vp, _sz, _sz, _vp, _sz]),
    ("zkw_check_satisfied", _int, [_vp, C.c_uint8, _vp, _sz, _u32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    ("zkw_vm_slice_instances", _int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("zkw_shard_lpt", _int, [_vp, _sz, _int, _vp]),
    ("zkw_comm_unique_id", _int, [_vp]),
    ("zkw_comm_init", _int, [_vp, _vp, _int, _int, C.POINTER(_vp)]),
    ("zkw_comm_init_tcp", _int, [_vp, C.c_char_p, _int, _int, _int, _int, C.POINTER(_vp)]),
    ("zkw_comm_init_rccl", _int, [_vp, _vp, _int, _int, C.POINTER(_vp)]),
    ("zkw_comm_exchange", _int, [_vp, _vp, _vp, C.c_size_t, _int]),
    ("zkw_comm_destroy", None, [_vp]),
    ("zkw_comm_synchronize", _int, [_vp]),
    ("zkw_gather_records", _int, [_vp, _vp, _sz, _vp, _sz, _int, _vp]),
    ("zkw_gather_closed_form_inputs", _int, [_vp, _vp, _vp, _sz, _int, _vp]),
    ("zkw_block_run", _int, [_int, _vp, C.POINTER(_vp)]),
    ("zkw_blocks_run", _int, [_int, _vp, _sz, _vp]),
    ("zkw_blocks_owner", _int, [_sz, _int]),
    ("zkw_blocks_run_sharded", _int, [_int, _vp, _sz, _int, _int, _vp]),
    ("zkw_blocks_gather_closed_form_inputs", _int, [_vp, _sz, _vp, _int, _int, _int, _sz, _vp]),
    ("zkw_block_last_error", C.c_char_p, []),
    ("zkw_block_free", None, [_vp]),
    ("zkw_blocks_free", None, [_vp, _sz]),
    ("zkw_block_witness", _vp, [_vp, C.c_uint8]),
    ("zkw_block_context", _vp, [_vp, C.c_uint8]),
    ("zkw_block_num_instances", _sz, [_vp, C.c_uint8]),
    ("zkw_block_public_inputs", _vp, [_vp, C.c_uint8]),
    ("zkw_block_recursion_encodings", _vp, [_vp, C.c_uint8]),
    ("zkw_block_recursion_states", _vp, [_vp, C.c_uint8]),
    ("zkw_block_vm_instances", _vp, [_vp]),
    ("zkw_block_memory_queue_length", _sz, [_vp]),
    ("zkw_block_memory_queue_device_ptr", _vp, [_vp]),
    ("zkw_block_memory_queue_state", _int, [_vp, _vp]),
    ("zkw_block_demuxed_offsets", _int, [_vp, _vp]),
    ("zkw_block_l1_messages_hash", _int, [_vp, _vp]),
    ("zkw_block_timings", _int, [_vp, C.c_char_p, _sz, _vp, _vp, _sz, C.POINTER(_sz)]),
    ("zkw_block_synthesize", _int, [_vp, _sz, _sz, _vp, _vp, C.POINTER(_sz)]),
    ("zkw_block_synthesize_sharded", _int, [_vp, _sz, _sz, _int, _int, _vp, _vp, C.POINTER(_sz)]),
    ("zkw_blocks_synthesize", _int, [_vp, _sz, _sz, _sz, _sz, _vp, _vp, C.POINTER(_sz)]),
    ("zkw_block_gather_closed_form_inputs", _int, [_vp, _vp, _int, _int, _int, _vp, _sz, C.POINTER(_sz)]),
]

_lib = None


class ZkwError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libzkw error {code}: {msg}")
        self.code = code


def load():
    """dlopen libzkw.so and type every exported symbol. Fails loudly when the extension is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ZkwError(ERR_NO_DEVICE, f"{LIB_PATH} not built: run `python -m era_zkevm_test_harness_amd.build`")
        # torch bundles its own libamdhip64; loading it first makes libzkw bind to the SAME HIP runtime, so
        # that device pointers and streams are interchangeable (two runtimes in one process cannot both
        # open the device). A C/Rust host without torch simply uses the system runtime.
        import torch  # noqa: F401

        lib = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(lib, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def _check(rc):
    if rc != OK:
        raise ZkwError(rc, load().zkw_last_error().decode())


def _np_ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


from .synthetic import MEM_QUERY  # noqa: E402

LOG_QUERY = np.dtype(
    [("timestamp", "<u4"), ("tx_number_in_block", "<u2"), ("aux_byte", "u1"), ("shard_id", "u1"),
     ("address", "<u4", (5,)), ("key", "<u4", (8,)), ("read_value", "<u4", (8,)), ("written_value", "<u4", (8,)),
     ("rw_flag", "u1"), ("rollback", "u1"), ("is_service", "u1"), ("_pad", "u1")])
DECOMMIT_QUERY = np.dtype([("hash", "<u4", (8,)), ("timestamp", "<u4"), ("memory_page", "<u4"),
                           ("decommitted_length", "<u2"), ("is_fresh", "u1"), ("_pad", "u1", (5,))])
assert LOG_QUERY.itemsize == 128 and DECOMMIT_QUERY.itemsize == 48
CALLSTACK_ENTRY = np.dtype(
    [("rollback_queue_head", "<u8", (4,)), ("rollback_queue_tail", "<u8", (4,)), ("rollback_queue_segment_length", "<u4"),
     ("code_address", "<u4", (5,)), ("this_address", "<u4", (5,)), ("msg_sender", "<u4", (5,)),
     ("context_u128_value", "<u4", (4,)), ("code_page", "<u4"), ("base_memory_page", "<u4"), ("ergs_remaining", "<u4"),
     ("heap_bound", "<u4"), ("aux_heap_bound", "<u4"), ("pc", "<u2"), ("sp", "<u2"), ("exception_handler_location", "<u2"),
     ("this_shard_id", "u1"), ("caller_shard_id", "u1"), ("code_shard_id", "u1"), ("is_static", "u1"),
     ("is_local_frame", "u1"), ("_pad", "u1", (1,))])
assert CALLSTACK_ENTRY.itemsize == 176
QUEUE_STATE12 = np.dtype([("head", "<u8", (12,)), ("tail", "<u8", (12,)), ("length", "<u4"), ("_pad", "<u4")])
RAM_FSM = np.dtype(
    [("lhs_accumulator", "<u8", (2,)), ("rhs_accumulator", "<u8", (2,)),
     ("current_unsorted_queue_state", QUEUE_STATE12), ("current_sorted_queue_state", QUEUE_STATE12),
     ("previous_sorting_key", "<u4", (3,)), ("previous_full_key", "<u4", (2,)), ("previous_value", "<u4", (8,)),
     ("previous_is_ptr", "<u4"), ("num_nondeterministic_writes", "<u4"), ("_pad", "<u4")])
RAM_INSTANCE = np.dtype(
    [("start_flag", "<u4"), ("completion_flag", "<u4"), ("unsorted_queue_initial_state", QUEUE_STATE12),
     ("sorted_queue_initial_state", QUEUE_STATE12),
     ("non_deterministic_bootloader_memory_snapshot_length", "<u4"), ("_pad", "<u4"),
     ("hidden_fsm_input", RAM_FSM), ("hidden_fsm_output", RAM_FSM), ("first_item", "<u8"), ("num_items", "<u8")])
assert QUEUE_STATE12.itemsize == 200 and RAM_FSM.itemsize == 496 and RAM_INSTANCE.itemsize == 1424


DECOMMIT_FSM = np.dtype(
    [("initial_queue_state", QUEUE_STATE12), ("sorted_queue_state", QUEUE_STATE12), ("final_queue_state", QUEUE_STATE12),
     ("lhs_accumulator", "<u8", (2,)), ("rhs_accumulator", "<u8", (2,)), ("previous_packed_key", "<u4", (9,)),
     ("first_encountered_timestamp", "<u4"), ("_pad", "<u4", (2,)), ("previous_record", DECOMMIT_QUERY)])
DECOMMIT_INSTANCE = np.dtype(
    [("start_flag", "<u4"), ("completion_flag", "<u4"), ("initial_queue_state", QUEUE_STATE12),
     ("sorted_queue_initial_state", QUEUE_STATE12), ("final_queue_state", QUEUE_STATE12),
     ("hidden_fsm_input", DECOMMIT_FSM), ("hidden_fsm_output", DECOMMIT_FSM), ("first_item", "<u8"), ("num_items", "<u8")])
(DEC_SORTED_QUERIES, DEC_UNSORTED_ENC, DEC_SORTED_ENC, DEC_UNSORTED_TAILS, DEC_SORTED_TAILS, DEC_DEDUP_QUERIES,
 DEC_DEDUP_TAILS, DEC_CHALLENGES, DEC_LHS_Z, DEC_RHS_Z, DEC_INSTANCES, DEC_COMPACT_FORMS, DEC_PUBLIC_INPUTS) = range(13)


QUEUE_STATE4 = np.dtype([("head", "<u8", (4,)), ("tail", "<u8", (4,)), ("length", "<u4"), ("_pad", "<u4")])
EVENTS_FSM = np.dtype(
    [("lhs_accumulator", "<u8", (2,)), ("rhs_accumulator", "<u8", (2,)), ("initial_unsorted_queue_state", QUEUE_STATE4),
     ("intermediate_sorted_queue_state", QUEUE_STATE4), ("final_result_queue_state", QUEUE_STATE4),
     ("previous_key", "<u4"), ("_pad", "<u4"), ("previous_item", LOG_QUERY)])
EVENTS_INSTANCE = np.dtype(
    [("start_flag", "<u4"), ("completion_flag", "<u4"), ("initial_log_queue_state", QUEUE_STATE4),
     ("intermediate_sorted_queue_state", QUEUE_STATE4), ("final_queue_state", QUEUE_STATE4),
     ("hidden_fsm_input", EVENTS_FSM), ("hidden_fsm_output", EVENTS_FSM), ("first_item", "<u8"), ("num_items", "<u8")])
(EVT_SORTED_QUERIES, EVT_UNSORTED_ENC, EVT_SORTED_ENC, EVT_UNSORTED_OLD_TAILS, EVT_UNSORTED_NEW_TAILS, EVT_SORTED_OLD_TAILS,
 EVT_SORTED_NEW_TAILS, EVT_RESULT_QUERIES, EVT_RESULT_NEW_TAILS, EVT_CHALLENGES, EVT_LHS_Z, EVT_RHS_Z, EVT_INSTANCES,
 EVT_COMPACT_FORMS, EVT_PUBLIC_INPUTS) = range(15)


class EventsWitness:
    """Owner of a zkw_events_witness handle."""

    _DTYPES = {EVT_SORTED_QUERIES: LOG_QUERY, EVT_RESULT_QUERIES: LOG_QUERY, EVT_INSTANCES: EVENTS_INSTANCE}
    _SHAPES = {EVT_UNSORTED_ENC: (-1, 20), EVT_SORTED_ENC: (-1, 20), EVT_UNSORTED_OLD_TAILS: (-1, 4),
               EVT_UNSORTED_NEW_TAILS: (-1, 4), EVT_SORTED_OLD_TAILS: (-1, 4), EVT_SORTED_NEW_TAILS: (-1, 4),
               EVT_RESULT_NEW_TAILS: (-1, 4), EVT_CHALLENGES: (2, 21), EVT_LHS_Z: (2, -1), EVT_RHS_Z: (2, -1),
               EVT_COMPACT_FORMS: (-1, 18), EVT_PUBLIC_INPUTS: (-1, 4)}

    def __init__(self, ctx):
        self.ctx = ctx
        self.handle = C.c_void_p(None)

    @property
    def num_instances(self):
        return load().zkw_events_witness_num_instances(self.handle)

    @property
    def num_results(self):
        return load().zkw_events_witness_num_results(self.handle)

    def get(self, what):
        lib = load()
        nbytes = lib.zkw_events_witness_bytes(self.handle, what)
        dt = self._DTYPES.get(what, np.dtype("<u8"))
        out = np.zeros(nbytes // dt.itemsize, dt)
        if nbytes:
            _check(lib.zkw_events_witness_get(self.handle, what, _np_ptr(out), nbytes))
        shape = self._SHAPES.get(what)
        return out.reshape(shape) if shape else out

    def free(self):
        if self.handle:
            load().zkw_events_witness_free(self.handle)
            self.handle = C.c_void_p(None)

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


DEMUX_FSM = np.dtype([("initial_log_queue_state", QUEUE_STATE4), ("queue_state", QUEUE_STATE4, (6,))])
DEMUX_INSTANCE = np.dtype(
    [("start_flag", "<u4"), ("completion_flag", "<u4"), ("initial_log_queue_state", QUEUE_STATE4),
     ("output_queue_state", QUEUE_STATE4, (6,)), ("hidden_fsm_input", DEMUX_FSM), ("hidden_fsm_output", DEMUX_FSM),
     ("first_item", "<u8"), ("num_items", "<u8")])
(DMX_IN_ENC, DMX_IN_OLD_TAILS, DMX_IN_NEW_TAILS, DMX_OUT_QUERIES, DMX_OUT_ENC, DMX_OUT_OLD_TAILS, DMX_OUT_NEW_TAILS,
 DMX_OUT_OFFSETS, DMX_INSTANCES, DMX_COMPACT_FORMS, DMX_PUBLIC_INPUTS) = range(11)


class DemuxWitness:
    """Owner of a zkw_demux_witness handle."""

    _DTYPES = {DMX_OUT_QUERIES: LOG_QUERY, DMX_INSTANCES: DEMUX_INSTANCE}
    _SHAPES = {DMX_IN_ENC: (-1, 20), DMX_OUT_ENC: (-1, 20), DMX_IN_OLD_TAILS: (-1, 4), DMX_IN_NEW_TAILS: (-1, 4),
               DMX_OUT_OLD_TAILS: (-1, 4), DMX_OUT_NEW_TAILS: (-1, 4), DMX_COMPACT_FORMS: (-1, 18), DMX_PUBLIC_INPUTS: (-1, 4)}

    def __init__(self, ctx):
        self.ctx = ctx
        self.handle = C.c_void_p(None)

    @property
    def num_instances(self):
        return load().zkw_demux_witness_num_instances(self.handle)

    def get(self, what):
        lib = load()
        nbytes = lib.zkw_demux_witness_bytes(self.handle, what)
        dt = self._DTYPES.get(what, np.dtype("<u8"))
        out = np.zeros(nbytes // dt.itemsize, dt)
        if nbytes:
            _check(lib.zkw_demux_witness_get(self.handle, what, _np_ptr(out), nbytes))
        shape = self._SHAPES.get(what)
        return out.reshape(shape) if shape else out

    def free(self):
        if self.handle:
            load().zkw_demux_witness_free(self.handle)
            self.handle = C.c_void_p(None)

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


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
(STO_SORTED_QUERIES, STO_SORTED_EXT_TS, STO_UNSORTED_ENC, STO_LHS_ENC, STO_SORTED_ENC, STO_UNSORTED_OLD_TAILS,
 STO_UNSORTED_NEW_TAILS, STO_SORTED_OLD_TAILS, STO_SORTED_NEW_TAILS, STO_RESULT_QUERIES, STO_RESULT_NEW_TAILS, STO_CHALLENGES,
 STO_LHS_Z, STO_RHS_Z, STO_INSTANCES, STO_COMPACT_FORMS, STO_PUBLIC_INPUTS) = range(17)


class StorageWitness:
    """Owner of a zkw_storage_witness handle."""

    _DTYPES = {STO_SORTED_QUERIES: LOG_QUERY, STO_RESULT_QUERIES: LOG_QUERY, STO_INSTANCES: STORAGE_INSTANCE,
               STO_SORTED_EXT_TS: np.dtype("<u4")}
    _SHAPES = {STO_UNSORTED_ENC: (-1, 20), STO_LHS_ENC: (-1, 20), STO_SORTED_ENC: (-1, 20), STO_UNSORTED_OLD_TAILS: (-1, 4),
               STO_UNSORTED_NEW_TAILS: (-1, 4), STO_SORTED_OLD_TAILS: (-1, 4), STO_SORTED_NEW_TAILS: (-1, 4),
               STO_RESULT_NEW_TAILS: (-1, 4), STO_CHALLENGES: (2, 21), STO_LHS_Z: (2, -1), STO_RHS_Z: (2, -1),
               STO_COMPACT_FORMS: (-1, 18), STO_PUBLIC_INPUTS: (-1, 4)}

    def __init__(self, ctx):
        self.ctx = ctx
        self.handle = C.c_void_p(None)

    @property
    def num_instances(self):
        return load().zkw_storage_witness_num_instances(self.handle)

    @property
    def num_results(self):
        return load().zkw_storage_witness_num_results(self.handle)

    def get(self, what):
        lib = load()
        nbytes = lib.zkw_storage_witness_bytes(self.handle, what)
        dt = self._DTYPES.get(what, np.dtype("<u8"))
        out = np.zeros(nbytes // dt.itemsize, dt)
        if nbytes:
            _check(lib.zkw_storage_witness_get(self.handle, what, _np_ptr(out), nbytes))
        shape = self._SHAPES.get(what)
        return out.reshape(shape) if shape else out

    def free(self):
        if self.handle:
            load().zkw_storage_witness_free(self.handle)
            self.handle = C.c_void_p(None)

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


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
DCM_MEM_QUERIES, DCM_MEM_ENC, DCM_MEM_TAILS, DCM_ROUND_STATES, DCM_INSTANCES, DCM_SHA256_ROUNDS = range(6)
PRECOMPILE_KECCAK256, PRECOMPILE_SHA256, PRECOMPILE_ECRECOVER = range(3)
PRECOMPILE_FSM = np.dtype(
    [("log_queue_state", QUEUE_STATE4), ("memory_queue_state", QUEUE_STATE12), ("read_precompile_call", "u1"),
     ("read_words_for_round", "u1"), ("padding_round", "u1"), ("completed", "u1"), ("timestamp_to_use_for_read", "<u4"),
     ("timestamp_to_use_for_write", "<u4"), ("input_page", "<u4"), ("input_offset", "<u4"), ("input_length", "<u4"),
     ("output_page", "<u4"), ("output_offset", "<u4"), ("num_rounds", "<u4"), ("needs_full_padding_round", "<u4"),
     ("buffer_filled", "<u4"), ("sha256_inner_state", "<u4", (8,)), ("keccak_internal_state", "u1", (200,)),
     ("buffer_bytes", "u1", (192,)), ("_pad", "<u4")])
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
LINEAR_HASHER_INSTANCE = np.dtype([("start_flag", "<u4"), ("completion_flag", "<u4"), ("queue_state", QUEUE_STATE4),
                                   ("keccak256_hash", "u1", (32,))])
assert LINEAR_HASHER_INSTANCE.itemsize == 112
# instance record of the circuits zkw_closed_form_public_inputs serves, by numeric circuit type
CLOSED_FORM_RECORD = {3: DECOMMITTER_INSTANCE, 5: PRECOMPILE_INSTANCE, 6: PRECOMPILE_INSTANCE, 7: PRECOMPILE_INSTANCE,
                      10: STORAGE_APPLICATION_INSTANCE, 13: LINEAR_HASHER_INSTANCE}
PRC_MEM_ENC, PRC_MEM_TAILS, PRC_INSTANCES, PRC_KECCAK_ROUNDS, PRC_SHA256_ROUNDS = range(5)
SHA256_ROUND_RECORD = np.dtype([("block", "u1", (64,)), ("reset", "<u4"), ("state_after", "<u4", (8,)), ("_pad", "<u4")])
assert SHA256_ROUND_RECORD.itemsize == 104
KECCAK_ROUND_RECORD = np.dtype([("block", "u1", (136,)), ("reset", "u1"), ("_pad", "u1", (7,)), ("state_after", "u1", (200,))])
assert KECCAK_ROUND_RECORD.itemsize == 344
SAP_DERIVED_KEYS, SAP_MERKLE_PATHS, SAP_LEAF_INDEXES, SAP_ROOTS, SAP_INSTANCES = range(5)


class StorageApplicationWitness:
    """Owner of a zkw_storage_application_witness handle."""

    _DTYPES = {SAP_DERIVED_KEYS: np.dtype("u1"), SAP_MERKLE_PATHS: np.dtype("u1"), SAP_ROOTS: np.dtype("u1"),
               SAP_INSTANCES: STORAGE_APPLICATION_INSTANCE}
    _SHAPES = {SAP_DERIVED_KEYS: (-1, 32), SAP_MERKLE_PATHS: (-1, 256, 32), SAP_ROOTS: (-1, 32)}

    def __init__(self, ctx):
        self.ctx = ctx
        self.handle = C.c_void_p(None)

    @property
    def num_instances(self):
        return load().zkw_storage_application_witness_num_instances(self.handle)

    def get(self, what):
        lib = load()
        nbytes = lib.zkw_storage_application_witness_bytes(self.handle, what)
        dt = self._DTYPES.get(what, np.dtype("<u8"))
        out = np.zeros(nbytes // dt.itemsize, dt)
        if nbytes:
            _check(lib.zkw_storage_application_witness_get(self.handle, what, _np_ptr(out), nbytes))
        shape = self._SHAPES.get(what)
        return out.reshape(shape) if shape else out

    def free(self):
        if self.handle:
            load().zkw_storage_application_witness_free(self.handle)
            self.handle = C.c_void_p(None)

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass



class PrecompileWitness:
    """Owner of a zkw_precompile_witness handle (keccak256 / sha256 / ecrecover round-function instances)."""

    _DTYPES = {PRC_INSTANCES: PRECOMPILE_INSTANCE, PRC_KECCAK_ROUNDS: KECCAK_ROUND_RECORD, PRC_SHA256_ROUNDS: SHA256_ROUND_RECORD}
    _SHAPES = {PRC_MEM_ENC: (-1, 8), PRC_MEM_TAILS: (-1, 12)}

    def __init__(self, ctx):
        self.ctx = ctx
        self.handle = C.c_void_p(None)

    @property
    def num_instances(self):
        return load().zkw_precompile_witness_num_instances(self.handle)

    @property
    def num_rounds(self):
        return load().zkw_precompile_witness_num_rounds(self.handle)

    def get(self, what):
        lib = load()
        nbytes = lib.zkw_precompile_witness_bytes(self.handle, what)
        dt = self._DTYPES.get(what, np.dtype("<u8"))
        out = np.zeros(nbytes // dt.itemsize, dt)
        if nbytes:
            _check(lib.zkw_precompile_witness_get(self.handle, what, _np_ptr(out), nbytes))
        shape = self._SHAPES.get(what)
        return out.reshape(shape) if shape else out

    def free(self):
        if self.handle:
            load().zkw_precompile_witness_free(self.handle)
            self.handle = C.c_void_p(None)

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass



class DecommitterWitness:
    """Owner of a zkw_decommitter_witness handle."""

    _DTYPES = {DCM_MEM_QUERIES: MEM_QUERY, DCM_INSTANCES: DECOMMITTER_INSTANCE, DCM_ROUND_STATES: np.dtype("<u4"),
               DCM_SHA256_ROUNDS: SHA256_ROUND_RECORD}
    _SHAPES = {DCM_MEM_ENC: (-1, 8), DCM_MEM_TAILS: (-1, 12), DCM_ROUND_STATES: (-1, 8)}

    def __init__(self, ctx):
        self.ctx = ctx
        self.handle = C.c_void_p(None)

    @property
    def num_instances(self):
        return load().zkw_decommitter_witness_num_instances(self.handle)

    def get(self, what):
        lib = load()
        nbytes = lib.zkw_decommitter_witness_bytes(self.handle, what)
        dt = self._DTYPES.get(what, np.dtype("<u8"))
        out = np.zeros(nbytes // dt.itemsize, dt)
        if nbytes:
            _check(lib.zkw_decommitter_witness_get(self.handle, what, _np_ptr(out), nbytes))
        shape = self._SHAPES.get(what)
        return out.reshape(shape) if shape else out

    def free(self):
        if self.handle:
            load().zkw_decommitter_witness_free(self.handle)
            self.handle = C.c_void_p(None)

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DecommitWitness:
    """Owner of a zkw_decommit_witness handle."""

    _DTYPES = {DEC_SORTED_QUERIES: DECOMMIT_QUERY, DEC_DEDUP_QUERIES: DECOMMIT_QUERY, DEC_INSTANCES: DECOMMIT_INSTANCE}
    _SHAPES = {DEC_UNSORTED_ENC: (-1, 8), DEC_SORTED_ENC: (-1, 8), DEC_UNSORTED_TAILS: (-1, 12), DEC_SORTED_TAILS: (-1, 12),
               DEC_DEDUP_TAILS: (-1, 12), DEC_CHALLENGES: (2, 9), DEC_LHS_Z: (2, -1), DEC_RHS_Z: (2, -1),
               DEC_COMPACT_FORMS: (-1, 18), DEC_PUBLIC_INPUTS: (-1, 4)}

    def __init__(self, ctx):
        self.ctx = ctx
        self.handle = C.c_void_p(None)

    @property
    def num_instances(self):
        return load().zkw_decommit_witness_num_instances(self.handle)

    @property
    def num_dedup(self):
        return load().zkw_decommit_witness_num_dedup(self.handle)

    def get(self, what):
        lib = load()
        nbytes = lib.zkw_decommit_witness_bytes(self.handle, what)
        dt = self._DTYPES.get(what, np.dtype("<u8"))
        out = np.zeros(nbytes // dt.itemsize, dt)
        if nbytes:
            _check(lib.zkw_decommit_witness_get(self.handle, what, _np_ptr(out), nbytes))
        shape = self._SHAPES.get(what)
        return out.reshape(shape) if shape else out

    def free(self):
        if self.handle:
            load().zkw_decommit_witness_free(self.handle)
            self.handle = C.c_void_p(None)

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class RamWitness:
    """Owner of a zkw_ram_witness handle (block-wide arrays + per-instance records, all in HBM)."""

    _DTYPES = {RAM_SORTED_QUERIES: MEM_QUERY, RAM_INSTANCES: RAM_INSTANCE}

    def __init__(self, ctx):
        self.ctx = ctx
        self.handle = C.c_void_p(None)

    @property
    def num_instances(self):
        return load().zkw_ram_witness_num_instances(self.handle)

    @property
    def num_items(self):
        return load().zkw_ram_witness_num_items(self.handle)

    def device_ptr(self, what):
        return load().zkw_ram_witness_device_ptr(self.handle, what)

    def nbytes(self, what):
        return load().zkw_ram_witness_bytes(self.handle, what)

    def get(self, what):
        """Copy one array to host memory as a numpy array."""
        lib = load()
        nbytes = lib.zkw_ram_witness_bytes(self.handle, what)
        dt = self._DTYPES.get(what, np.dtype("<u8"))
        out = np.zeros(nbytes // dt.itemsize, dt)
        mode = self.ctx.pointer_mode
        self.ctx.set_pointer_mode(PTR_HOST)
        try:
            _check(lib.zkw_ram_witness_get(self.handle, what, _np_ptr(out), nbytes))
        finally:
            self.ctx.set_pointer_mode(mode)
        t = self.num_items
        shape = {RAM_UNSORTED_ENC: (t, 8), RAM_SORTED_ENC: (t, 8), RAM_UNSORTED_TAILS: (t, 12),
                 RAM_SORTED_TAILS: (t, 12), RAM_CHALLENGES: (-1, 2, 9), RAM_COMPACT_FORMS: (-1, 18),
                 RAM_PUBLIC_INPUTS: (-1, 4)}.get(what)
        return out.reshape(shape) if shape else out

    def free(self):
        if self.handle:
            load().zkw_ram_witness_free(self.handle)
            self.handle = C.c_void_p(None)

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Trace:
    """zkw_trace: a ring of filled-trace buffers in HBM, each column-major u64[n_cols][n_rows]."""

    def __init__(self, ctx, n_rows, n_slots=1, n_cols=None):
        self.ctx = ctx
        self.handle = C.c_void_p(None)
        if n_cols is None:  # the RAMPermutation geometry, 149 columns
            _check(load().zkw_trace_create(ctx.handle, n_rows, n_slots, C.byref(self.handle)))
        else:
            _check(load().zkw_trace_create_with_columns(ctx.handle, n_rows, n_cols, n_slots, C.byref(self.handle)))
        self.n_rows, self.n_slots = n_rows, n_slots
        self.n_cols = load().zkw_trace_num_cols(self.handle)

    def device_ptr(self, slot=0):
        return load().zkw_trace_device_ptr(self.handle, slot)

    def get(self, slot=0, first_col=0, n_cols=None):
        n_cols = self.n_cols - first_col if n_cols is None else n_cols
        out = np.zeros((n_cols, self.n_rows), np.uint64)
        mode = self.ctx.pointer_mode
        self.ctx.set_pointer_mode(PTR_HOST)
        try:
            _check(load().zkw_trace_get(self.handle, slot, first_col, n_cols, _np_ptr(out)))
        finally:
            self.ctx.set_pointer_mode(mode)
        return out

    def free(self):
        if self.handle:
            load().zkw_trace_free(self.handle)
            self.handle = C.c_void_p(None)

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Context:
    """zkw_ctx: one per process / GPU."""

    def __init__(self, device_id=0):
        lib = load()
        self.handle = lib.zkw_create(device_id)
        if not self.handle:
            raise ZkwError(ERR_NO_DEVICE, lib.zkw_last_error().decode())
        self.pointer_mode = PTR_HOST

    def close(self):
        if self.handle:
            load().zkw_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_chain_stream(self, hip_stream):
        _check(load().zkw_set_chain_stream(self.handle, C.c_void_p(hip_stream)))

    def set_pointer_mode(self, mode):
        _check(load().zkw_set_pointer_mode(self.handle, mode))
        self.pointer_mode = mode

    def set_stream(self, stream_handle):
        """stream_handle: a hipStream_t as int (0 = the null stream = torch's default stream), or None for
        the context's own stream."""
        h = C.c_void_p(-1) if stream_handle is None else C.c_void_p(stream_handle)
        _check(load().zkw_set_stream(self.handle, h))

    def set_chain_form(self, lanes_per_state):
        _check(load().zkw_set_chain_form(self.handle, lanes_per_state))

    def set_netlist_fill_form(self, form):
        """0: a wave per cycle (default), 1: a lane per cycle (zkw.h)"""
        _check(load().zkw_set_netlist_fill_form(self.handle, form))

    def synchronize(self):
        _check(load().zkw_synchronize(self.handle))

    def profile_enable(self, on=True):
        _check(load().zkw_profile_enable(self.handle, 1 if on else 0))

    def profile_reset(self):
        _check(load().zkw_profile_reset(self.handle))

    def profile(self):
        """{kernel name: (total ms, launches)} measured with HIP events on the context's stream."""
        lib = load()
        buf = C.create_string_buffer(4096)
        _check(lib.zkw_profile_names(self.handle, buf, 4096))
        out = {}
        for name in filter(None, buf.value.decode().split(",")):
            ms, cnt = C.c_double(0), C.c_uint64(0)
            _check(lib.zkw_profile_get(self.handle, name.encode(), C.byref(ms), C.byref(cnt)))
            out[name] = (ms.value, cnt.value)
        return out

    # ---- host-pointer conveniences (numpy in, numpy out) mirroring the reference's function names
    def encode_memory_queries(self, q):
        """MemoryQuery::encoding_witness over an array of queries (memory_query.rs:24-118)."""
        q = np.ascontiguousarray(q, dtype=MEM_QUERY)
        enc = np.zeros((q.size, 8), np.uint64)
        _check(load().zkw_encode_memory_queries(self.handle, _np_ptr(q), q.size, _np_ptr(enc)))
        return enc

    def queue_push_chain_full(self, enc, tail_in=None):
        """FullWidthQueueSimulator pushes (lib.rs:391-429): returns tails [n][12]."""
        enc = _u64(enc)
        n = enc.shape[0]
        tails = np.zeros((n, 12), np.uint64)
        tin = None if tail_in is None else _np_ptr(_u64(tail_in))
        _check(load().zkw_queue_push_chain_full(self.handle, _np_ptr(enc), n, tin, _np_ptr(tails)))
        return tails

    def queue_push_chain_full_batch(self, enc, offsets, tails_in=None):
        enc = _u64(enc)
        offsets = _u64(offsets)
        tails = np.zeros((enc.shape[0], 12), np.uint64)
        tin = None if tails_in is None else _np_ptr(_u64(tails_in))
        _check(load().zkw_queue_push_chain_full_batch(self.handle, _np_ptr(enc), _np_ptr(offsets), offsets.size - 1,
                                                      tin, _np_ptr(tails)))
        return tails

    def encode_log_queries(self, q, ext_ts=None):
        """LogQuery::encoding_witness (log_query.rs:102-396); ext_ts -> the extended-enumeration variant."""
        q = np.ascontiguousarray(q, dtype=LOG_QUERY)
        enc = np.zeros((q.size, 20), np.uint64)
        e = None if ext_ts is None else np.ascontiguousarray(ext_ts, dtype=np.uint32)
        _check(load().zkw_encode_log_queries(self.handle, _np_ptr(q), q.size, None if e is None else _np_ptr(e),
                                             _np_ptr(enc)))
        return enc

    def encode_decommit_queries(self, q):
        q = np.ascontiguousarray(q, dtype=DECOMMIT_QUERY)
        enc = np.zeros((q.size, 8), np.uint64)
        _check(load().zkw_encode_decommit_queries(self.handle, _np_ptr(q), q.size, _np_ptr(enc)))
        return enc

    def queue_push_chain_log(self, enc, offsets=None, tails_in=None):
        """QueueSimulator pushes (lib.rs:179-221): returns (old_tails, new_tails), each [n][4]."""
        enc = _u64(enc)
        n = enc.shape[0]
        offsets = np.array([0, n], np.uint64) if offsets is None else _u64(offsets)
        old_t, new_t = np.zeros((n, 4), np.uint64), np.zeros((n, 4), np.uint64)
        tin = None if tails_in is None else _np_ptr(_u64(tails_in))
        _check(load().zkw_queue_push_chain_log_batch(self.handle, _np_ptr(enc), _np_ptr(offsets), offsets.size - 1,
                                                     tin, _np_ptr(old_t), _np_ptr(new_t)))
        return old_t, new_t

    def produce_fs_challenges(self, tail_u, len_u, tail_s, len_s, state_w, n_chal):
        """produce_fs_challenges (utils.rs:498-550): [2][n_chal]."""
        out = np.zeros((2, n_chal), np.uint64)
        tu, ts = _u64(tail_u), _u64(tail_s)
        _check(load().zkw_fs_challenges(self.handle, _np_ptr(tu), len_u, _np_ptr(ts), len_s, state_w, n_chal,
                                        _np_ptr(out)))
        return out

    def compute_grand_product_chains(self, lhs, rhs, challenges):
        """compute_grand_product_chains (utils.rs:554-697). challenges: [W+1] or [n_reps][W+1]."""
        lhs, rhs, ch = _u64(lhs), _u64(rhs), _u64(challenges)
        n, w = lhs.shape
        single = ch.ndim == 1
        ch2 = ch.reshape(1, -1) if single else ch
        reps = ch2.shape[0]
        assert ch2.shape[1] == w + 1 and rhs.shape == (n, w)
        lz, rz = np.zeros((reps, n), np.uint64), np.zeros((reps, n), np.uint64)
        _check(load().zkw_grand_product_chains(self.handle, _np_ptr(lhs), _np_ptr(rhs), n, w, _np_ptr(ch2), reps,
                                               _np_ptr(lz), _np_ptr(rz)))
        return (lz[0], rz[0]) if single else (lz, rz)

    def compute_ram_circuit_snapshots(self, queries, per_circuit_capacity, num_non_deterministic_heap_queries=0,
                                      block_offsets=None, witness=None):
        """compute_ram_circuit_snapshots (W/ram_permutation.rs:26-470) for one block, or for several
        independent blocks when block_offsets is given. Returns a RamWitness."""
        lib = load()
        w = witness or RamWitness(self)
        if self.pointer_mode == PTR_HOST:
            queries = np.ascontiguousarray(queries, dtype=MEM_QUERY)
            qptr, n = _np_ptr(queries), queries.size
        else:
            qptr, n = queries  # (device address, count)
        if block_offsets is None:
            nd = num_non_deterministic_heap_queries
            _check(lib.zkw_ram_build_instances(self.handle, qptr, n, per_circuit_capacity, nd, C.byref(w.handle)))
        else:
            offs = _u64(block_offsets)
            nb = offs.size - 1
            nd = np.ascontiguousarray(np.broadcast_to(np.asarray(num_non_deterministic_heap_queries, np.uint32), (nb,)))
            _check(lib.zkw_ram_build_instances_batch(self.handle, qptr, _np_ptr(offs), nb, per_circuit_capacity,
                                                     _np_ptr(nd), C.byref(w.handle)))
        return w

    def synthesize_ram(self, witness, trace, first_instance=0, n_instances=None, first_slot=0):
        """ZkSyncBaseLayerCircuit::synthesis for RAMPermutation instances (base_layer/mod.rs:286-323):
        fills trace slots (first_slot + k) % n_slots with instances first_instance + k."""
        n = witness.num_instances - first_instance if n_instances is None else n_instances
        _check(load().zkw_ram_synthesize(self.handle, witness.handle, first_instance, n, trace.handle, first_slot))

    def synthesize(self, circuit_type, witness, trace, first_instance=0, n_instances=1, first_slot=0):
        """zkw_synthesize: ZkSyncBaseLayerCircuit::synthesis as ONE entry point over the circuit types (base_layer/mod.rs:286-323);
        `witness` = the witness object of the type (its .handle is passed) or a raw handle"""
        h = getattr(witness, "handle", witness)
        _check(load().zkw_synthesize(self.handle, circuit_type, h, first_instance, n_instances, trace.handle, first_slot))

    def check_if_satisfied(self, circuit_type, trace, slot, capacity):
        """zkw_check_satisfied: (violations, (kind, index, row) of the first) for a slot holding `circuit_type`"""
        bad, first = C.c_uint64(0), C.c_uint64(0)
        _check(load().zkw_check_satisfied(self.handle, circuit_type, trace.handle, slot, capacity, C.byref(bad), C.byref(first)))
        v = first.value
        return bad.value, (v >> 56, (v >> 32) & 0xFFFFFF, v & 0xFFFFFFFF)

    def check_if_satisfied_ram(self, trace, slot, capacity):
        """check_if_satisfied (src/tests/mod.rs:130-259): (violations, (kind, index, row) of the first)."""
        bad, first = C.c_uint64(0), C.c_uint64(0)
        _check(load().zkw_ram_check_satisfied(self.handle, trace.handle, slot, capacity, C.byref(bad), C.byref(first)))
        v = first.value
        return bad.value, (v >> 56, (v >> 32) & 0xFFFFFF, v & 0xFFFFFFFF)

    def compute_decommitts_sorter_circuit_snapshots(self, queries, deduplicator_circuit_capacity, dedup_in=None):
        """compute_decommitts_sorter_circuit_snapshots (sort_decommit_requests.rs:20-420) -> DecommitWitness."""
        q = np.ascontiguousarray(queries, dtype=DECOMMIT_QUERY)
        w = DecommitWitness(self)
        din = None if dedup_in is None else _np_ptr(np.ascontiguousarray(dedup_in, dtype=QUEUE_STATE12))
        _check(load().zkw_decommit_sorter_build(self.handle, _np_ptr(q), q.size, deduplicator_circuit_capacity, din,
                                                C.byref(w.handle)))
        return w

    def compute_events_dedup_and_sort(self, unsorted_queries, per_circuit_capacity, result_in=None):
        """compute_events_dedup_and_sort (events_sort_dedup.rs:16-506) -> EventsWitness."""
        q = np.ascontiguousarray(unsorted_queries, dtype=LOG_QUERY)
        w = EventsWitness(self)
        rin = None if result_in is None else _np_ptr(np.ascontiguousarray(result_in, dtype=QUEUE_STATE4))
        _check(load().zkw_events_sorter_build(self.handle, _np_ptr(q) if q.size else None, q.size, per_circuit_capacity, rin,
                                              C.byref(w.handle)))
        return w

    def compute_logs_demux(self, log_queries, per_circuit_capacity):
        """compute_logs_demux (log_demux.rs:20-388) -> DemuxWitness (six demuxed queues + per-instance records)."""
        q = np.ascontiguousarray(log_queries, dtype=LOG_QUERY)
        w = DemuxWitness(self)
        _check(load().zkw_log_demux_build(self.handle, _np_ptr(q) if q.size else None, q.size, per_circuit_capacity, None,
                                          C.byref(w.handle)))
        return w

    def compute_storage_dedup_and_sort(self, demuxed_rollup_storage_queries, per_circuit_capacity):
        """sort_storage_access_queries + compute_storage_dedup_and_sort (storage_sort_dedup.rs:12-703) -> StorageWitness."""
        q = np.ascontiguousarray(demuxed_rollup_storage_queries, dtype=LOG_QUERY)
        w = StorageWitness(self)
        _check(load().zkw_storage_sorter_build(self.handle, _np_ptr(q) if q.size else None, q.size, per_circuit_capacity,
                                               C.byref(w.handle)))
        return w

    def compute_decommitter_circuit_snapshots(self, requests, dedup_tails, words, word_offsets, decommiter_circuit_capacity,
                                              mem_in):
        """compute_decommitter_circuit_snapshots (decommit_code.rs:20-439) -> DecommitterWitness."""
        req = np.ascontiguousarray(requests, dtype=DECOMMIT_QUERY)
        dt = _u64(dedup_tails)
        wd = np.ascontiguousarray(words, dtype=np.uint32).reshape(-1, 8)
        woff = _u64(word_offsets)
        mi = np.ascontiguousarray(mem_in, dtype=QUEUE_STATE12)
        w = DecommitterWitness(self)
        _check(load().zkw_decommitter_build(self.handle, _np_ptr(req), _np_ptr(dt), req.size, _np_ptr(wd), _np_ptr(woff),
                                            decommiter_circuit_capacity, _np_ptr(mi), C.byref(w.handle)))
        return w

    def compute_linear_keccak256(self, messages) -> bytes:
        """compute_linear_keccak256 (data_hasher_and_merklizer.rs:8-67): the pubdata hash of the L2->L1 messages."""
        q = np.ascontiguousarray(messages, dtype=LOG_QUERY)
        out = np.zeros(32, np.uint8)
        _check(load().zkw_linear_keccak256(self.handle, _np_ptr(q) if q.size else None, q.size, _np_ptr(out)))
        return out.tobytes()

    def commit_variable_length_encodable_items(self, enc) -> np.ndarray:
        """commit_variable_length_encodable_item for a batch of equal-length flat encodings [n][len] -> [n][4]
        (the step simulate_public_input_value_from_witness applies four times + once, utils.rs:269-306)."""
        enc = np.ascontiguousarray(enc, dtype=np.uint64)
        assert enc.ndim == 2
        out = np.zeros((enc.shape[0], 4), np.uint64)
        # one spare element keeps the pointer valid for zero-length items
        buf = np.concatenate([enc.reshape(-1), np.zeros(1, np.uint64)])
        _check(load().zkw_commit_encodings(self.handle, _np_ptr(buf), enc.shape[0], enc.shape[1], _np_ptr(out)))
        return out

    # ---- the setup side as field elements (include/zkw.h: NTT, LDE, Merkle tree with a cap)
    def ntt(self, values, inverse=False) -> np.ndarray:
        """zkw_ntt over the last axis of [n_cols][2^k] (natural order in and out)"""
        v = np.ascontiguousarray(values, dtype=np.uint64)
        v2 = v.reshape(-1, v.shape[-1])
        log_n = int(v2.shape[1]).bit_length() - 1
        assert 1 << log_n == v2.shape[1]
        out = np.zeros_like(v2)
        _check(load().zkw_ntt(self.handle, _np_ptr(v2), _np_ptr(out), C.c_uint32(log_n), C.c_size_t(v2.shape[0]), C.c_int(1 if inverse else 0)))
        return out.reshape(v.shape)

    def lde(self, values, lde_factor=2) -> np.ndarray:
        """zkw_lde: [n_cols][n] values on the domain -> [lde_factor][n_cols][n] on the cosets 7 * w_(lde n)^c * <w_n>"""
        v = np.ascontiguousarray(values, dtype=np.uint64)
        assert v.ndim == 2
        log_n = int(v.shape[1]).bit_length() - 1
        out = np.zeros((lde_factor,) + v.shape, np.uint64)
        _check(load().zkw_lde(self.handle, _np_ptr(v), C.c_uint32(log_n), C.c_size_t(v.shape[0]), C.c_uint32(lde_factor), _np_ptr(out)))
        return out

    def merkle_tree_with_cap(self, leaf_cols, cap_size=16, want_tree=False):
        """zkw_merkle_tree_with_cap: leaf_cols [n_sets][n_cols][n] -> cap [cap_size][4] (and every level, leaves first, [nodes][4])"""
        v = np.ascontiguousarray(leaf_cols, dtype=np.uint64)
        assert v.ndim == 3
        n_sets, n_cols, n = v.shape
        cap = np.zeros((cap_size, 4), np.uint64)
        lib = load()
        tree = np.zeros((lib.zkw_merkle_tree_words(C.c_size_t(n_sets * n), C.c_uint32(cap_size)) // 4, 4), np.uint64) if want_tree else None
        _check(lib.zkw_merkle_tree_with_cap(self.handle, _np_ptr(v), C.c_size_t(n_sets), C.c_size_t(n_cols), C.c_size_t(n), C.c_uint32(cap_size),
                                            _np_ptr(cap), _np_ptr(tree) if want_tree else None))
        return (cap, tree) if want_tree else cap

    def setup_columns(self, circuit_type, capacity, log_n) -> np.ndarray:
        """zkw_setup_columns: the sigma columns as field elements, then the selector column: [n_columns][2^log_n]"""
        nc = C.c_uint32(0)
        _check(load().zkw_setup_num_columns(C.c_uint8(circuit_type), C.byref(nc)))
        out = np.zeros((nc.value, 1 << log_n), np.uint64)
        _check(load().zkw_setup_columns(self.handle, C.c_uint8(circuit_type), C.c_uint32(capacity), C.c_uint32(log_n), _np_ptr(out)))
        return out

    def setup_commit(self, circuit_type, capacity, log_n, lde_factor=2, cap_size=16) -> np.ndarray:
        """zkw_setup_commit: setup columns -> monomial form -> LDE -> Merkle tree -> cap [cap_size][4], all on the device"""
        cap = np.zeros((cap_size, 4), np.uint64)
        _check(load().zkw_setup_commit(self.handle, C.c_uint8(circuit_type), C.c_uint32(capacity), C.c_uint32(log_n), C.c_uint32(lde_factor),
                                       C.c_uint32(cap_size), _np_ptr(cap)))
        return cap

    def closed_form_public_inputs(self, circuit_type, instances):
        """zkw_closed_form_public_inputs: compact closed-form inputs [n][18] and public inputs [n][4] of the instance
        records of circuit type 3, 5, 6, 7, 10 or 13 (postprocessing/mod.rs:353-369)."""
        inst = np.ascontiguousarray(instances, dtype=CLOSED_FORM_RECORD[circuit_type])
        compact = np.zeros((inst.size, 18), np.uint64)
        pi = np.zeros((inst.size, 4), np.uint64)
        _check(load().zkw_closed_form_public_inputs(self.handle, circuit_type, _np_ptr(inst) if inst.size else None, inst.size,
                                                    _np_ptr(compact), _np_ptr(pi)))
        return compact, pi

    def recursion_queue_push(self, circuit_type, public_inputs, tail_in=None):
        """RecursionQueueSimulator::push for every instance of one circuit type (postprocessing/mod.rs:393-400):
        returns (encodings [n][8], queue states [n][12])."""
        pi = np.ascontiguousarray(public_inputs, dtype=np.uint64).reshape(-1, 4)
        enc = np.zeros((pi.shape[0], 8), np.uint64)
        _check(load().zkw_encode_recursion_requests(self.handle, circuit_type, _np_ptr(pi), pi.shape[0], _np_ptr(enc)))
        return enc, self.queue_push_chain_full(enc, tail_in)

    def encode_callstack_entries(self, entries) -> np.ndarray:
        """ExtendedCallstackEntry::encoding_witness (callstack_entry.rs:36-179): [n][32]."""
        e = np.ascontiguousarray(entries, dtype=CALLSTACK_ENTRY)
        enc = np.zeros((e.size, 32), np.uint64)
        _check(load().zkw_encode_callstack_entries(self.handle, _np_ptr(e), e.size, _np_ptr(enc)))
        return enc

    def callstack_simulate(self, is_push, pushed):
        """CallstackSimulator pushes / pops replayed on an empty stack (lib.rs:558-644)."""
        ops = np.ascontiguousarray(is_push, dtype=np.uint8)
        e = np.ascontiguousarray(pushed, dtype=CALLSTACK_ENTRY)
        n = ops.size
        o = {"previous_state": np.zeros((n, 12), np.uint64), "new_state": np.zeros((n, 12), np.uint64),
             "depth": np.zeros(n, np.uint32), "round_states": np.zeros((n, 4, 12), np.uint64),
             "entry_index": np.zeros(n, np.uint32)}
        _check(load().zkw_callstack_simulate(self.handle, _np_ptr(ops), n, _np_ptr(e) if e.size else None, e.size,
                                             _np_ptr(o["previous_state"]), _np_ptr(o["new_state"]), _np_ptr(o["depth"]),
                                             _np_ptr(o["round_states"]), _np_ptr(o["entry_index"])))
        return o

    def _precompile(self, kind, requests, request_tails, mem_queries, num_rounds_per_circuit, mem_in):
        req = np.ascontiguousarray(requests, dtype=LOG_QUERY)
        rt = _u64(request_tails)
        mq = np.ascontiguousarray(mem_queries, dtype=MEM_QUERY)
        mi = np.ascontiguousarray(mem_in, dtype=QUEUE_STATE12)
        w = PrecompileWitness(self)
        _check(load().zkw_precompile_build(self.handle, kind, _np_ptr(req) if req.size else None,
                                           _np_ptr(rt) if req.size else None, req.size, _np_ptr(mq) if mq.size else None,
                                           mq.size, num_rounds_per_circuit, _np_ptr(mi), C.byref(w.handle)))
        return w

    def keccak256_decompose_into_per_circuit_witness(self, requests, request_tails, mem_queries, num_rounds_per_circuit, mem_in):
        """keccak256_round_function.rs:23-528 -> PrecompileWitness."""
        return self._precompile(PRECOMPILE_KECCAK256, requests, request_tails, mem_queries, num_rounds_per_circuit, mem_in)

    def sha256_decompose_into_per_circuit_witness(self, requests, request_tails, mem_queries, num_rounds_per_circuit, mem_in):
        """sha256_round_function.rs:23-406 -> PrecompileWitness."""
        return self._precompile(PRECOMPILE_SHA256, requests, request_tails, mem_queries, num_rounds_per_circuit, mem_in)

    def ecrecover_decompose_into_per_circuit_witness(self, requests, request_tails, mem_queries, num_rounds_per_circuit, mem_in):
        """ecrecover.rs:12-262 -> PrecompileWitness."""
        return self._precompile(PRECOMPILE_ECRECOVER, requests, request_tails, mem_queries, num_rounds_per_circuit, mem_in)

    def decompose_into_storage_application_witnesses(self, queries, query_tails, init_leaf_indexes, init_merkle_paths,
                                                     initial_root, initial_next_enumeration_index, num_rounds_per_circuit):
        """storage_application.rs:31-361 over the pre-block answers of the storage tree -> StorageApplicationWitness."""
        q = np.ascontiguousarray(queries, dtype=LOG_QUERY)
        qt = _u64(query_tails)
        ii = _u64(init_leaf_indexes)
        ip = np.ascontiguousarray(init_merkle_paths, dtype=np.uint8)
        root = np.frombuffer(bytes(initial_root), np.uint8).copy()
        assert root.size == 32 and ip.size == q.size * 256 * 32
        w = StorageApplicationWitness(self)
        some = q.size > 0
        _check(load().zkw_storage_application_build(self.handle, _np_ptr(q) if some else None, _np_ptr(qt) if some else None,
                                                    q.size, _np_ptr(ii) if some else None, _np_ptr(ip) if some else None,
                                                    _np_ptr(root), initial_next_enumeration_index, num_rounds_per_circuit,
                                                    C.byref(w.handle)))
        return w


CIRCUIT_GEOMETRY = np.dtype(
    [("num_columns_under_copy_permutation", "<u4"), ("num_witness_columns", "<u4"), ("num_constant_columns", "<u4"),
     ("max_allowed_constraint_degree", "<u4"), ("lookup_width", "<u4"), ("lookup_repetitions", "<u4"), ("capacity", "<u4"),
     ("trace_len_log2", "<u4"), ("size_hint_variables", "<u8")])


CIRCUIT_LAYOUT = np.dtype(
    [("synthesizable", "<u4"), ("fits", "<u4"), ("capacity", "<u4"), ("num_columns", "<u4"), ("rows_per_cycle", "<u4"), ("total_table_rows", "<u4"),
     ("region_stride", "<u8"), ("rows_used", "<u8"), ("nop_rows", "<u8"), ("trace_len", "<u8"), ("public_input_column", "<u4", (4,)),
     ("public_input_row", "<u8", (4,)), ("queue_first_row", "<u8"), ("queue_rows_per_cycle", "<u4"), ("ec_rows_per_cycle", "<u4"), ("ec_first_row", "<u8"),
     ("closed_form_first_row", "<u8"), ("closed_form_rows", "<u4"), ("closed_form_header_rows", "<u4")])


def circuit_fill_bytes(circuit_type: int, capacity: int = 0, n_rows: int = 1 << 20):
    """zkw_circuit_fill_bytes: (warm, cold) bytes one synthesis call writes into one slot"""
    warm, cold = C.c_uint64(0), C.c_uint64(0)
    _check(load().zkw_circuit_fill_bytes(circuit_type, capacity, n_rows, C.byref(warm), C.byref(cold)))
    return warm.value, cold.value


def circuit_layout(circuit_type: int, capacity: int = 0):
    """zkw_circuit_layout_of: rows used, padding and public-input cells of this library's trace layout (no GPU needed)"""
    g = np.zeros(1, CIRCUIT_LAYOUT)
    _check(load().zkw_circuit_layout_of(circuit_type, capacity, _np_ptr(g)))
    return g[0]


def circuit_geometry(circuit_type: int):
    """ZkSyncBaseLayerCircuit::geometry / size_hint + the GeometryConfig capacity of one circuit type (no GPU needed)."""
    g = np.zeros(1, CIRCUIT_GEOMETRY)
    _check(load().zkw_circuit_geometry_of(circuit_type, _np_ptr(g)))
    return g[0]


def _ctx_synthesize_decommit_sorter(self, witness, trace, first_instance=0, n_instances=None, first_slot=0):
    """ZkSyncBaseLayerCircuit::CodeDecommittmentsSorter synthesis for instances of a DecommitWitness."""
    n = witness.num_instances - first_instance if n_instances is None else n_instances
    _check(load().zkw_decommit_sorter_synthesize(self.handle, witness.handle, first_instance, n, trace.handle, first_slot))


def _ctx_check_if_satisfied_decommit_sorter(self, trace, slot, capacity):
    bad, first = C.c_uint64(0), C.c_uint64(0)
    _check(load().zkw_decommit_sorter_check_satisfied(self.handle, trace.handle, slot, capacity, C.byref(bad), C.byref(first)))
    v = first.value
    return bad.value, (v >> 56, (v >> 32) & 0xFFFFFF, v & 0xFFFFFFFF)


Context.synthesize_decommit_sorter = _ctx_synthesize_decommit_sorter
Context.check_if_satisfied_decommit_sorter = _ctx_check_if_satisfied_decommit_sorter


def _ctx_synthesize_events_sorter(self, witness, trace, first_instance=0, n_instances=None, first_slot=0):
    """ZkSyncBaseLayerCircuit::{EventsSorter, L1MessagesSorter} synthesis for instances of an EventsWitness."""
    n = witness.num_instances - first_instance if n_instances is None else n_instances
    _check(load().zkw_events_sorter_synthesize(self.handle, witness.handle, first_instance, n, trace.handle, first_slot))


def _ctx_check_if_satisfied_events_sorter(self, trace, slot, capacity):
    bad, first = C.c_uint64(0), C.c_uint64(0)
    _check(load().zkw_events_sorter_check_satisfied(self.handle, trace.handle, slot, capacity, C.byref(bad), C.byref(first)))
    v = first.value
    return bad.value, (v >> 56, (v >> 32) & 0xFFFFFF, v & 0xFFFFFFFF)


Context.synthesize_events_sorter = _ctx_synthesize_events_sorter


def _ctx_synthesize_log_demux(self, witness, trace, first_instance=0, n_instances=None, first_slot=0):
    """ZkSyncBaseLayerCircuit::LogDemuxer synthesis for instances of a DemuxWitness (the trace needs 151 columns)."""
    n = witness.num_instances - first_instance if n_instances is None else n_instances
    _check(load().zkw_log_demux_synthesize(self.handle, witness.handle, first_instance, n, trace.handle, first_slot))


def _ctx_check_if_satisfied_log_demux(self, trace, slot, capacity):
    bad, first = C.c_uint64(0), C.c_uint64(0)
    _check(load().zkw_log_demux_check_satisfied(self.handle, trace.handle, slot, capacity, C.byref(bad), C.byref(first)))
    v = first.value
    return bad.value, (v >> 56, (v >> 32) & 0xFFFFFF, v & 0xFFFFFFFF)


Context.synthesize_log_demux = _ctx_synthesize_log_demux


KC_COLS, LH_COLS = 129, 145  # 86 + 3 x 14 + 1, 66 + 3 x 26 + 1 (include/zkw_keccak_circuit_spec.h, zkw_linear_hasher_circuit_spec.h)


def _ctx_synthesize_keccak_round_function(self, witness, trace, first_instance=0, n_instances=None, first_slot=0):
    """ZkSyncBaseLayerCircuit::Keccak256RoundFunction synthesis ("zkw trace v4") for instances of a keccak256
    PrecompileWitness (the trace needs KC_COLS columns and at least 132 096 rows: the stacked tables)."""
    n = witness.num_instances - first_instance if n_instances is None else n_instances
    _check(load().zkw_keccak_round_synthesize(self.handle, witness.handle, first_instance, n, trace.handle, first_slot))


def _ctx_check_if_satisfied_keccak_round_function(self, trace, slot, capacity):
    bad, first = C.c_uint64(0), C.c_uint64(0)
    _check(load().zkw_keccak_round_check_satisfied(self.handle, trace.handle, slot, capacity, C.byref(bad), C.byref(first)))
    v = first.value
    return bad.value, (v >> 56, (v >> 32) & 0xFFFFFF, v & 0xFFFFFFFF)


EK_COLS = 129  # 80 + 3 x 16 + 1 (include/zkw_ecrecover_circuit_spec.h)


def _ctx_synthesize_ecrecover(self, witness, trace, first_instance=0, n_instances=None, first_slot=0):
    """ZkSyncBaseLayerCircuit::ECRecover synthesis for instances of an ecrecover PrecompileWitness (the trace needs EK_COLS columns and
    at least 197 632 rows: the stacked tables)"""
    n = witness.num_instances - first_instance if n_instances is None else n_instances
    _check(load().zkw_ecrecover_synthesize(self.handle, witness.handle, first_instance, n, trace.handle, first_slot))


def _ctx_check_if_satisfied_ecrecover(self, trace, slot, capacity):
    bad, first = C.c_uint64(0), C.c_uint64(0)
    _check(load().zkw_ecrecover_check_satisfied(self.handle, trace.handle, slot, capacity, C.byref(bad), C.byref(first)))
    v = first.value
    return bad.value, (v >> 56, (v >> 32) & 0xFFFFFF, v & 0xFFFFFFFF)


def linear_hasher_cycles(capacity):
    return capacity * 88 // 136 + 1  # ZKW_LINEAR_HASHER_CYCLES


def _ctx_synthesize_linear_hasher(self, messages, queue_state, capacity, trace, slot=0):
    """ZkSyncBaseLayerCircuit::LinearHasher synthesis (type 13) over the net L2 -> L1 messages of a block: returns the
    instance record and its public input (the trace needs LH_COLS columns). Check with check_if_satisfied_linear_hasher(trace,
    slot, capacity)."""
    q = np.ascontiguousarray(messages, dtype=LOG_QUERY)
    qs = np.ascontiguousarray(queue_state, dtype=QUEUE_STATE4).reshape(1)
    rec = np.zeros(1, LINEAR_HASHER_INSTANCE)
    pi = np.zeros(4, np.uint64)
    _check(load().zkw_linear_hasher_synthesize(self.handle, _np_ptr(q) if q.size else None, q.size, _np_ptr(qs), capacity, trace.handle,
                                               slot, _np_ptr(rec), _np_ptr(pi)))
    return rec, pi


def _ctx_synthesize_linear_hasher_batch(self, queues, queue_states, capacity, trace, first_slot=0, tails=None):
    """zkw_linear_hasher_synthesize_batch[_with_tails]: the L1-messages queues of several blocks in one call (queue b -> slot
    first_slot + b). tails: per queue the states [n, 4] after each message's push (what the sorter that built the queue holds; the
    pops of the trace's queue section run through them) — None: hashed inside the call, one serial chain per queue.
    Returns (records, public inputs [n, 4])."""
    qs = [np.ascontiguousarray(q, dtype=LOG_QUERY) for q in queues]
    off = np.zeros(len(qs) + 1, np.uint64)
    off[1:] = np.cumsum([q.size for q in qs])
    flat = np.concatenate(qs) if qs else np.zeros(0, LOG_QUERY)
    st = np.ascontiguousarray(queue_states, dtype=QUEUE_STATE4).reshape(len(qs))
    rec = np.zeros(len(qs), LINEAR_HASHER_INSTANCE)
    pi = np.zeros((len(qs), 4), np.uint64)
    if tails is not None:
        ft = (np.concatenate([np.ascontiguousarray(t_, dtype=np.uint64).reshape(-1, 4) for t_ in tails]) if flat.size else np.zeros((0, 4), np.uint64))
        assert ft.shape[0] == flat.size
        _check(load().zkw_linear_hasher_synthesize_batch_with_tails(self.handle, _np_ptr(flat) if flat.size else None, _np_ptr(off), len(qs), _np_ptr(st),
                                                                    _np_ptr(ft) if flat.size else None, capacity, trace.handle, first_slot, _np_ptr(rec), _np_ptr(pi)))
        return rec, pi
    _check(load().zkw_linear_hasher_synthesize_batch(self.handle, _np_ptr(flat) if flat.size else None, _np_ptr(off), len(qs), _np_ptr(st),
                                                     capacity, trace.handle, first_slot, _np_ptr(rec), _np_ptr(pi)))
    return rec, pi


SC_COLS = 153  # 116 + 4 x 9 + 1 (include/zkw_sha256_circuit_spec.h)


def _ctx_synthesize_sha256_round_function(self, witness, trace, first_instance=0, n_instances=None, first_slot=0):
    """ZkSyncBaseLayerCircuit::Sha256RoundFunction synthesis ("zkw trace v4") for instances of a sha256 PrecompileWitness
    (the trace needs SC_COLS columns)."""
    n = witness.num_instances - first_instance if n_instances is None else n_instances
    _check(load().zkw_sha256_round_synthesize(self.handle, witness.handle, first_instance, n, trace.handle, first_slot))


def _ctx_check_if_satisfied_sha256_round_function(self, trace, slot, capacity):
    bad, first = C.c_uint64(0), C.c_uint64(0)
    _check(load().zkw_sha256_round_check_satisfied(self.handle, trace.handle, slot, capacity, C.byref(bad), C.byref(first)))
    v = first.value
    return bad.value, (v >> 56, (v >> 32) & 0xFFFFFF, v & 0xFFFFFFFF)


DC_COLS = 153  # 108 + 4 x 11 + 1 (include/zkw_code_decommitter_circuit_spec.h)


def _ctx_synthesize_code_decommitter(self, witness, trace, first_instance=0, n_instances=None, first_slot=0):
    """ZkSyncBaseLayerCircuit::CodeDecommitter synthesis ("zkw trace v4": the SHA-256 netlist on 108 + 4 x 11 columns) for
    instances of a DecommitterWitness (the trace needs DC_COLS columns)."""
    n = witness.num_instances - first_instance if n_instances is None else n_instances
    _check(load().zkw_code_decommitter_synthesize(self.handle, witness.handle, first_instance, n, trace.handle, first_slot))


def _ctx_check_if_satisfied_code_decommitter(self, trace, slot, capacity):
    bad, first = C.c_uint64(0), C.c_uint64(0)
    _check(load().zkw_code_decommitter_check_satisfied(self.handle, trace.handle, slot, capacity, C.byref(bad), C.byref(first)))
    v = first.value
    return bad.value, (v >> 56, (v >> 32) & 0xFFFFFF, v & 0xFFFFFFFF)


Context.synthesize_code_decommitter = _ctx_synthesize_code_decommitter

SA_COLS = 139  # 60 + 3 x 26 + 1 (include/zkw_storage_application_circuit_spec.h)
SA_CYCLES_PER_WALK = 257


def _ctx_synthesize_storage_application(self, witness, trace, first_instance=0, n_instances=None, first_slot=0):
    """ZkSyncBaseLayerCircuit::StorageApplication synthesis ("zkw trace v4": the Merkle walks of the instance's tree queries as
    Blake2s compressions on 60 + 3 x 26 columns) for instances of a StorageApplicationWitness (the trace needs SA_COLS columns)."""
    n = witness.num_instances - first_instance if n_instances is None else n_instances
    _check(load().zkw_storage_application_synthesize(self.handle, witness.handle, first_instance, n, trace.handle, first_slot))


def _ctx_check_if_satisfied_storage_application(self, trace, slot, capacity):
    bad, first = C.c_uint64(0), C.c_uint64(0)
    _check(load().zkw_storage_application_check_satisfied(self.handle, trace.handle, slot, capacity, C.byref(bad), C.byref(first)))
    v = first.value
    return bad.value, (v >> 56, (v >> 32) & 0xFFFFFF, v & 0xFFFFFFFF)


Context.synthesize_storage_application = _ctx_synthesize_storage_application
Context.check_if_satisfied_storage_application = _ctx_check_if_satisfied_storage_application
Context.check_if_satisfied_code_decommitter = _ctx_check_if_satisfied_code_decommitter
Context.synthesize_sha256_round_function = _ctx_synthesize_sha256_round_function
Context.check_if_satisfied_sha256_round_function = _ctx_check_if_satisfied_sha256_round_function
def _ctx_check_copy_permutation(self, trace, slot, sigma):
    """zkw_check_copy_permutation: (violations, (kind, column, row)) of trace[cell] == trace[sigma[cell]]"""
    sg = np.ascontiguousarray(sigma, dtype=np.uint64)
    bad, first = C.c_uint64(0), C.c_uint64(0)
    _check(load().zkw_check_copy_permutation(self.handle, trace.handle, slot, _np_ptr(sg), sg.shape[0], C.byref(bad), C.byref(first)))
    v = first.value
    return bad.value, (v >> 56, (v >> 32) & 0xFFFFFF, v & 0xFFFFFFFF)


def _ctx_check_if_satisfied_linear_hasher(self, trace, slot, capacity):
    bad, first = C.c_uint64(0), C.c_uint64(0)
    _check(load().zkw_linear_hasher_check_satisfied(self.handle, trace.handle, slot, capacity, C.byref(bad), C.byref(first)))
    v = first.value
    return bad.value, (v >> 56, (v >> 32) & 0xFFFFFF, v & 0xFFFFFFFF)


Context.check_if_satisfied_linear_hasher = _ctx_check_if_satisfied_linear_hasher
Context.check_copy_permutation = _ctx_check_copy_permutation
Context.synthesize_linear_hasher = _ctx_synthesize_linear_hasher
Context.synthesize_ecrecover = _ctx_synthesize_ecrecover
Context.check_if_satisfied_ecrecover = _ctx_check_if_satisfied_ecrecover
Context.synthesize_linear_hasher_batch = _ctx_synthesize_linear_hasher_batch
Context.synthesize_keccak_round_function = _ctx_synthesize_keccak_round_function
Context.check_if_satisfied_keccak_round_function = _ctx_check_if_satisfied_keccak_round_function


def _ctx_synthesize_storage_sorter(self, witness, trace, first_instance=0, n_instances=None, first_slot=0):
    """ZkSyncBaseLayerCircuit::StorageSorter synthesis for instances of a StorageWitness."""
    n = witness.num_instances - first_instance if n_instances is None else n_instances
    _check(load().zkw_storage_sorter_synthesize(self.handle, witness.handle, first_instance, n, trace.handle, first_slot))


def _ctx_check_if_satisfied_storage_sorter(self, trace, slot, capacity):
    bad, first = C.c_uint64(0), C.c_uint64(0)
    _check(load().zkw_storage_sorter_check_satisfied(self.handle, trace.handle, slot, capacity, C.byref(bad), C.byref(first)))
    v = first.value
    return bad.value, (v >> 56, (v >> 32) & 0xFFFFFF, v & 0xFFFFFFFF)


Context.synthesize_storage_sorter = _ctx_synthesize_storage_sorter
Context.check_if_satisfied_storage_sorter = _ctx_check_if_satisfied_storage_sorter
Context.check_if_satisfied_log_demux = _ctx_check_if_satisfied_log_demux
Context.check_if_satisfied_events_sorter = _ctx_check_if_satisfied_events_sorter



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


def vm_slice_instances(ctx, tracer):
    """zkw_vm_slice_instances (host pointer mode): (instances [n_snapshots - 1], memory read indices, memory write indices)"""
    keep = []
    st = _vm_streams_struct(tracer, keep)
    n_inst, n_mem = st.n_snapshots - 1, st.stream_len[0]
    inst = np.zeros(n_inst, VM_INSTANCE)
    ri, wi = np.zeros(max(n_mem, 1), np.uint32), np.zeros(max(n_mem, 1), np.uint32)
    nr, nw = C.c_uint64(0), C.c_uint64(0)
    _check(load().zkw_vm_slice_instances(ctx.handle, C.byref(st), _np_ptr(inst), _np_ptr(ri), _np_ptr(wi), C.byref(nr), C.byref(nw)))
    return inst, ri[:nr.value], wi[:nw.value]

VM_EVENT = np.dtype([("kind", "<u4"), ("cycle", "<u4"), ("panicked", "<u4"), ("index", "<u4")])
VM_TRACE_SUMMARY = np.dtype([("n_flat", "<u8"), ("original_log_queue_length", "<u8"), ("n_frames", "<u8"),
                             ("global_end_of_storage_log", "<u8", 4), ("original_log_queue_tail", "<u8", 4)])
_VMT = {"flat_queries": (0, None), "flat_cycles": (1, np.uint32), "flat_frames": (2, np.uint32), "flat_old_tails": (3, (np.uint64, 4)),
        "flat_new_tails": (4, (np.uint64, 4)), "new_frame_tail_cycles": (5, np.uint32), "new_frame_tails": (6, (np.uint64, 4)),
        "head_segment_cycles": (7, np.uint32), "head_segments": (8, (np.uint64, 4)), "storage_log_state_cycles": (9, np.uint32),
        "storage_log_state_frames": (10, np.uint32), "storage_log_states": (11, None), "callstack_witness_cycles": (12, np.uint32),
        "callstack_witness_is_push": (13, np.uint8), "callstack_witness_entries": (14, None),
        "callstack_witness_previous_states": (15, (np.uint64, 12)), "callstack_witness_new_states": (16, (np.uint64, 12)),
        "callstack_witness_depths": (17, np.uint32), "callstack_witness_round_states": (18, (np.uint64, 48)),
        "callstack_sponge_cycles": (19, np.uint32), "callstack_sponge_states": (20, (np.uint64, 12)), "new_frame_cycles": (21, np.uint32),
        "new_frame_entries": (22, None)}


class VmTrace:
    """zkw_vm_trace: the pre-builder half of create_artifacts_from_tracer (callstack_handler.rs:174-460, oracle.rs:233-843)
    over the tracer's raw record (synthetic.vm_events shapes)."""

    def __init__(self, ctx, events, log_queries, entries):
        ev = np.ascontiguousarray(events, dtype=VM_EVENT)
        q = np.ascontiguousarray(log_queries, dtype=LOG_QUERY)
        e = np.ascontiguousarray(entries, dtype=CALLSTACK_ENTRY)
        self.handle = C.c_void_p(None)
        _check(load().zkw_vm_trace_build(ctx.handle, _np_ptr(ev) if ev.size else None, ev.size, _np_ptr(q) if q.size else None, q.size,
                                         _np_ptr(e) if e.size else None, e.size, C.byref(self.handle)))

    def get(self, name):
        what, kind = _VMT[name]
        n = load().zkw_vm_trace_count(self.handle, what)
        if kind is None:
            dt = {0: LOG_QUERY, 11: STORAGE_LOG_DETAILED_STATE, 14: CALLSTACK_ENTRY, 22: CALLSTACK_ENTRY}[what]
            out = np.zeros(n, dt)
        elif isinstance(kind, tuple):
            out = np.zeros((n, kind[1]), kind[0])
        else:
            out = np.zeros(n, kind)
        _check(load().zkw_vm_trace_get(self.handle, what, _np_ptr(out) if out.size else None, out.nbytes))
        return out

    def info(self):
        out = np.zeros(1, VM_TRACE_SUMMARY)
        _check(load().zkw_vm_trace_info(self.handle, _np_ptr(out)))
        return out[0]

    def tracer_streams(self, vm_streams):
        """the dict zkw_vm_slice_instances takes: `vm_streams` (the VM's own streams, synthetic.vm_tracer_streams keys) with the
        four FIFOs and the histories this trace produced put in"""
        t = dict(vm_streams)
        sc = list(t["stream_cycles"])
        sc[4], sc[5], sc[6], sc[7] = (self.get("new_frame_tail_cycles"), self.get("callstack_witness_cycles"), self.get("head_segment_cycles"),
                                      self.get("new_frame_cycles"))
        t["stream_cycles"] = sc
        t["callstack_sponge_cycles"], t["callstack_sponge_states"] = self.get("callstack_sponge_cycles"), self.get("callstack_sponge_states")
        t["storage_log_state_cycles"], t["storage_log_states"] = self.get("storage_log_state_cycles"), self.get("storage_log_states")
        t["global_end_of_storage_log"] = self.info()["global_end_of_storage_log"].copy()
        return t

    def free(self):
        if self.handle:
            load().zkw_vm_trace_free(self.handle)
            self.handle = C.c_void_p(None)

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


# ---- one block (zkw_block_run): the post-VM half of create_artifacts_from_tracer inside the library ------------------
STORAGE_TREE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p)
CIRCUIT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint8, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64))
BLOCKS_CIRCUIT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.c_uint8, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64))


class BlockInputs(C.Structure):
    """zkw_block_inputs (include/zkw.h)"""
    _fields_ = [("vm_memory_queries", C.c_void_p), ("n_vm_memory_queries", C.c_size_t),
                ("decommit_queries", C.c_void_p), ("n_decommit_queries", C.c_size_t),
                ("bytecode_hashes", C.c_void_p), ("bytecode_words", C.c_void_p), ("bytecode_word_offsets", C.c_void_p),
                ("n_bytecodes", C.c_size_t),
                ("log_queries", C.c_void_p), ("n_log_queries", C.c_size_t),
                ("precompile_memory_queries", C.c_void_p * 3), ("n_precompile_memory_queries", C.c_size_t * 3),
                ("num_non_deterministic_heap_queries", C.c_uint32),
                ("storage_tree", STORAGE_TREE_FN), ("storage_tree_user", C.c_void_p),
                ("storage_initial_root", C.c_uint8 * 32), ("storage_initial_next_enumeration_index", C.c_uint64),
                ("capacities", C.c_uint32 * 14), ("vm_tracer", C.c_void_p), ("queues_on_device", C.c_uint32)]


class Block:
    """zkw_block: every witness builder of one block scheduled as a dependency graph inside libzkw (csrc/zkw_block.hip).
    `block`: the dict of synthetic.block_after_vm (or the same arrays from a real VM run); `capacities`: circuit type ->
    capacity (default geometry_config.rs); `storage_tree`: callable(dedup_queries) -> (leaf_indexes, merkle_paths,
    initial_root, next_enumeration_index is given separately) or None."""

    WITNESS_GETTERS = {2: "zkw_decommit_witness", 3: "zkw_decommitter_witness", 4: "zkw_demux_witness", 5: "zkw_precompile_witness",
                       6: "zkw_precompile_witness", 7: "zkw_precompile_witness", 8: "zkw_ram_witness", 9: "zkw_storage_witness",
                       10: "zkw_storage_application_witness", 11: "zkw_events_witness", 12: "zkw_events_witness"}

    @staticmethod
    def queues_to_device(block, device_id=0):
        """The block dict with its four queues (VM memory queries, decommit requests, log queries, the precompiles' memory queries) copied
        to the device ONCE, as a VM running next to the library would leave them: Block(..) of the result passes device pointers
        (zkw_block_inputs.queues_on_device) and the builders read them in place. The host arrays stay in the dict for callers that
        want them."""
        import torch

        def up(a, dtype):
            a = np.ascontiguousarray(a, dtype=dtype)
            t = torch.from_numpy(a.view(np.uint8).reshape(-1).copy()).to(torch.device("cuda", device_id))
            return (t, a.size)

        d = dict(block)
        d["_device_queues"] = {"vm": up(block["vm_memory_queries"], MEM_QUERY), "dq": up(block["decommit_queries"], DECOMMIT_QUERY),
                               "lq": up(block["log_queries"], LOG_QUERY),
                               "pm": [up(block["precompile_memory_queries"][k], MEM_QUERY) for k in range(3)]}
        torch.cuda.synchronize(device_id)
        return d

    def __init__(self, device_id, block, capacities=None, storage_tree=None, storage_initial_root=None,
                 storage_next_enumeration_index=0, num_non_deterministic_heap_queries=0, vm_tracer=None, _run=True):
        lib = load()
        inp = BlockInputs()
        keep = []
        self._inp = inp
        dev_q = block.get("_device_queues")

        class _Dev:  # a queue that lives on the device: what the input struct needs of it
            def __init__(self, pair):
                self.tensor, self.size = pair
                self.ctypes = type("p", (), {"data": self.tensor.data_ptr() if self.size else None})

        def arr(a, dtype, dev=None):
            if dev is not None:
                keep.append(dev[0])
                return _Dev(dev)
            a = np.ascontiguousarray(a, dtype=dtype)
            keep.append(a)
            return a

        inp.queues_on_device = 1 if dev_q else 0
        vm = arr(block["vm_memory_queries"], MEM_QUERY, dev_q and dev_q["vm"])
        inp.vm_memory_queries, inp.n_vm_memory_queries = vm.ctypes.data, vm.size
        dq = arr(block["decommit_queries"], DECOMMIT_QUERY, dev_q and dev_q["dq"])
        inp.decommit_queries, inp.n_decommit_queries = dq.ctypes.data, dq.size
        hashes = list(block["bytecodes"].keys())
        codes = [np.ascontiguousarray(block["bytecodes"][h], dtype=np.uint32).reshape(-1, 8) for h in hashes]
        hh = arr(np.frombuffer(b"".join(hashes), np.uint32).reshape(-1, 8) if hashes else np.zeros((0, 8), np.uint32), np.uint32)
        ww = arr(np.concatenate(codes) if codes else np.zeros((0, 8), np.uint32), np.uint32)
        wo = arr(np.concatenate([[0], np.cumsum([c.shape[0] for c in codes])]), np.uint64)
        inp.bytecode_hashes, inp.bytecode_words, inp.bytecode_word_offsets, inp.n_bytecodes = hh.ctypes.data, ww.ctypes.data, wo.ctypes.data, len(hashes)
        lq = arr(block["log_queries"], LOG_QUERY, dev_q and dev_q["lq"])
        inp.log_queries, inp.n_log_queries = lq.ctypes.data, lq.size
        for k in range(3):
            mq = arr(block["precompile_memory_queries"][k], MEM_QUERY, dev_q and dev_q["pm"][k])
            inp.precompile_memory_queries[k] = mq.ctypes.data if mq.size else None
            inp.n_precompile_memory_queries[k] = mq.size
        inp.num_non_deterministic_heap_queries = num_non_deterministic_heap_queries
        self._tree_error = None
        if storage_tree is not None:
            def _cb(_user, q_ptr, n, idx_ptr, paths_ptr):
                try:
                    q = np.ctypeslib.as_array(C.cast(q_ptr, C.POINTER(C.c_uint8)), (n * LOG_QUERY.itemsize,)).view(LOG_QUERY).copy()
                    idx, paths = storage_tree(q)
                    np.ctypeslib.as_array(C.cast(idx_ptr, C.POINTER(C.c_uint64)), (n,))[:] = np.asarray(idx, np.uint64)
                    np.ctypeslib.as_array(C.cast(paths_ptr, C.POINTER(C.c_uint8)), (n * 256 * 32,))[:] = np.asarray(paths, np.uint8).reshape(-1)
                    return 0
                except Exception as e:  # noqa: BLE001 - reported through the return code
                    self._tree_error = e
                    return 1
            self._cb = STORAGE_TREE_FN(_cb)
            inp.storage_tree = self._cb
            root = np.frombuffer(bytes(storage_initial_root), np.uint8)
            assert root.size == 32
            for i in range(32):
                inp.storage_initial_root[i] = int(root[i])
            inp.storage_initial_next_enumeration_index = storage_next_enumeration_index
        if vm_tracer is not None:  # the tracer's cycle-stamped vectors: MainVM instance records come back with the block
            t = dict(vm_tracer)
            t.setdefault("vm_memory_queries", np.ascontiguousarray(block["vm_memory_queries"], dtype=MEM_QUERY))  # ignored by the block (it uses its own memory queue / states)
            t.setdefault("memory_queue_tails", np.zeros((vm.size, 12), np.uint64))
            t.setdefault("decommit_queue_tails", np.zeros((dq.size, 12), np.uint64))
            self._vm_struct = _vm_streams_struct(t, keep)
            inp.vm_tracer = C.addressof(self._vm_struct)
        self.capacities = {t: int(circuit_geometry(t)["capacity"]) for t in range(1, 14)}
        for t, c in (capacities or {}).items():
            inp.capacities[t] = c
            self.capacities[t] = c
        self.handle = C.c_void_p(None)
        self._keep = keep
        if not _run:
            return
        rc = lib.zkw_block_run(device_id, C.byref(inp), C.byref(self.handle))
        if rc != OK:
            if self._tree_error is not None:
                raise self._tree_error
            raise ZkwError(rc, (lib.zkw_block_last_error() or b"").decode() or lib.zkw_last_error().decode())

    @staticmethod
    def run_many(device_id, blocks, capacities=None):
        """zkw_blocks_run: the given blocks (dicts as for Block) in flight together, their queue chains merged into shared
        launches by the chain service. Returns the list of Block objects."""
        lib = load()
        objs = [Block(device_id, b, capacities, _run=False) for b in blocks]
        ptrs = (C.c_void_p * len(objs))(*[C.addressof(o._inp) for o in objs])
        outs = (C.c_void_p * len(objs))()
        rc = lib.zkw_blocks_run(device_id, ptrs, len(objs), outs)
        if rc != OK:
            raise ZkwError(rc, (lib.zkw_block_last_error() or b"").decode() or lib.zkw_last_error().decode())
        for o, h in zip(objs, outs):
            o.handle = C.c_void_p(h)
        return objs

    @staticmethod
    def prepare_many(device_id, blocks, capacities=None):
        """The input structs of `blocks` built once (a service that receives its blocks as arrays does this as they arrive): the list
        run_prepared takes, any number of times."""
        return [Block(device_id, b, capacities, _run=False) for b in blocks]

    @staticmethod
    def run_prepared(device_id, templates):
        """zkw_blocks_run over inputs prepared by prepare_many: nothing but the call. Returns new Block objects (the templates stay
        reusable; they share the input arrays)."""
        import copy

        lib = load()
        ptrs = (C.c_void_p * len(templates))(*[C.addressof(o._inp) for o in templates])
        outs = (C.c_void_p * len(templates))()
        rc = lib.zkw_blocks_run(device_id, ptrs, len(templates), outs)
        if rc != OK:
            raise ZkwError(rc, (lib.zkw_block_last_error() or b"").decode() or lib.zkw_last_error().decode())
        objs = []
        for t, h in zip(templates, outs):
            o = copy.copy(t)
            o.handle = C.c_void_p(h)
            objs.append(o)
        return objs

    @staticmethod
    def run_sharded_prepared(device_id, templates, rank, world):
        """zkw_blocks_run_sharded over inputs prepared by prepare_many: a Block for every owned index, None elsewhere"""
        import copy

        lib = load()
        ptrs = (C.c_void_p * len(templates))(*[C.addressof(o._inp) for o in templates])
        outs = (C.c_void_p * len(templates))()
        rc = lib.zkw_blocks_run_sharded(device_id, ptrs, len(templates), rank, world, outs)
        if rc != OK:
            raise ZkwError(rc, (lib.zkw_block_last_error() or b"").decode() or lib.zkw_last_error().decode())
        res = []
        for t, h in zip(templates, outs):
            if h is None:
                res.append(None)
            else:
                o = copy.copy(t)
                o.handle = C.c_void_p(h)
                res.append(o)
        return res

    @staticmethod
    def run_sharded(device_id, blocks, rank, world, capacities=None):
        """zkw_blocks_run_sharded: every rank passes the same list of blocks; rank r builds the blocks k with k % world == r.
        Returns a list with a Block for every owned index and None elsewhere."""
        lib = load()
        objs = [Block(device_id, b, capacities, _run=False) for b in blocks]
        ptrs = (C.c_void_p * len(objs))(*[C.addressof(o._inp) for o in objs])
        outs = (C.c_void_p * len(objs))()
        rc = lib.zkw_blocks_run_sharded(device_id, ptrs, len(objs), rank, world, outs)
        if rc != OK:
            raise ZkwError(rc, (lib.zkw_block_last_error() or b"").decode() or lib.zkw_last_error().decode())
        res = []
        for k, (o, h) in enumerate(zip(objs, outs)):
            assert (h is not None) == (lib.zkw_blocks_owner(k, world) == rank)
            if h is None:
                res.append(None)
            else:
                o.handle = C.c_void_p(h)
                res.append(o)
        return res

    @staticmethod
    def gather_sharded(blocks, comm, rank, world, root=0, max_per_block=64):
        """zkw_blocks_gather_closed_form_inputs (collective): on the root a list, per block, of its records [n, 24] (type,
        instance, compact form, public input) in emission order; None on the other ranks."""
        n = len(blocks)
        handles = (C.c_void_p * max(n, 1))(*[b.handle if b is not None else None for b in blocks])
        words = 1 + 24 * max_per_block
        out = np.zeros((n, words), np.uint64)
        rc = load().zkw_blocks_gather_closed_form_inputs(handles, n, comm.handle, rank, world, root, max_per_block, _np_ptr(out))
        if rc != OK:
            raise ZkwError(rc, (load().zkw_block_last_error() or b"").decode() or load().zkw_last_error().decode())
        if rank != root:
            return None
        return [out[k, 1:1 + 24 * int(out[k, 0])].reshape(-1, 24).copy() for k in range(n)]

    def num_instances(self, circuit_type):
        return load().zkw_block_num_instances(self.handle, circuit_type)

    def witness_get(self, circuit_type, what, dtype=np.uint64):
        """one array of the witness of `circuit_type` (the ZKW_* enum of its witness type), copied to the host"""
        lib = load()
        prefix = self.WITNESS_GETTERS[circuit_type]
        w = lib.zkw_block_witness(self.handle, circuit_type)
        if not w:
            return None
        nbytes = getattr(lib, prefix + "_bytes")(w, what)
        out = np.zeros(nbytes, np.uint8)
        if nbytes:
            ctx = lib.zkw_block_context(self.handle, circuit_type)
            _check(lib.zkw_set_pointer_mode(ctx, PTR_HOST))  # the block's contexts run in device-pointer mode
            try:
                _check(getattr(lib, prefix + "_get")(w, what, _np_ptr(out), nbytes))
            finally:
                _check(lib.zkw_set_pointer_mode(ctx, PTR_DEVICE))
        return out.view(dtype)

    def _host_u64(self, fn, circuit_type, width):
        p = fn(self.handle, circuit_type)
        n = self.num_instances(circuit_type)
        if not p:
            return None
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint64)), (n, width)).copy()

    def vm_instances(self):
        """MainVM instance records (VM_INSTANCE) when the block was given the tracer's vectors, else None"""
        p = load().zkw_block_vm_instances(self.handle)
        n = self.num_instances(1)
        if not p or not n:
            return None
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), (n * VM_INSTANCE.itemsize,)).view(VM_INSTANCE).copy()

    def public_inputs(self, t):
        return self._host_u64(load().zkw_block_public_inputs, t, 4)

    def recursion_queue(self, t):
        return (self._host_u64(load().zkw_block_recursion_encodings, t, 8), self._host_u64(load().zkw_block_recursion_states, t, 12))

    @property
    def memory_queue_length(self):
        return load().zkw_block_memory_queue_length(self.handle)

    def memory_queue_state(self):
        s = np.zeros(1, QUEUE_STATE12)
        _check(load().zkw_block_memory_queue_state(self.handle, _np_ptr(s)))
        return s

    def demuxed_offsets(self):
        o = np.zeros(7, np.uint64)
        _check(load().zkw_block_demuxed_offsets(self.handle, _np_ptr(o)))
        return o

    def linear_hasher_instance(self):
        rec = np.zeros(1, LINEAR_HASHER_INSTANCE)
        _check(load().zkw_block_linear_hasher_instance(self.handle, _np_ptr(rec)))
        return rec

    def l1_messages_hash(self):
        h = np.zeros(32, np.uint8)
        _check(load().zkw_block_l1_messages_hash(self.handle, _np_ptr(h)))
        return h.tobytes()

    def timings(self):
        """[(name, start_ms, end_ms)] of the builders' (and the last synthesis call's) wall-clock spans"""
        names = C.create_string_buffer(4096)
        a, b = np.zeros(128), np.zeros(128)
        n = C.c_size_t(0)
        _check(load().zkw_block_timings(self.handle, names, 4096, _np_ptr(a), _np_ptr(b), 128, C.byref(n)))
        nm = names.value.decode().split(",") if n.value else []
        return [(nm[i], float(a[i]), float(b[i])) for i in range(n.value)]

    def synthesize(self, n_rows, ring_slots=4, callback=None, rank=0, world=1):
        """ZkSyncBaseLayerCircuit::synthesis of every instance (of this rank's LPT share when world > 1) in emission order;
        callback(circuit_type, instance, trace_handle, slot, public_input[4]) -> None (raise to stop). Returns the number of
        instances synthesized."""
        err = []

        def _cb(_user, ctype, inst, trace, slot, pi):
            try:
                if callback is not None:
                    callback(int(ctype), int(inst), trace, int(slot), [int(pi[i]) for i in range(4)])
                return 0
            except Exception as e:  # noqa: BLE001
                err.append(e)
                return 1
        cb = CIRCUIT_FN(_cb)
        n = C.c_size_t(0)
        rc = load().zkw_block_synthesize_sharded(self.handle, n_rows, ring_slots, rank, world, C.cast(cb, C.c_void_p), None, C.byref(n))
        if err:
            raise err[0]
        _check(rc)
        return n.value

    @staticmethod
    def synthesize_many(blocks, n_rows, ring_slots=1, ec_chunk=0, callback=None):
        """zkw_blocks_synthesize: every instance of every block (None entries skipped); the ECRecover instances of all blocks in joint
        calls of at most ec_chunk instances. callback(block_index, circuit_type, instance, trace_handle, slot, public_input[4]) may be
        called from several library threads. Returns the number of instances synthesized."""
        mine = [b for b in blocks if b is not None]
        if not mine:
            return 0
        err = []

        def _cb(_user, blk, ctype, inst, trace, slot, pi):
            try:
                if callback is not None:
                    callback(int(blk), int(ctype), int(inst), trace, int(slot), [int(pi[i]) for i in range(4)])
                return 0
            except Exception as e:  # noqa: BLE001
                err.append(e)
                return 1
        cb = BLOCKS_CIRCUIT_FN(_cb) if callback is not None else None
        ptrs = (C.c_void_p * len(mine))(*[b.handle for b in mine])
        n = C.c_size_t(0)
        rc = load().zkw_blocks_synthesize(ptrs, len(mine), n_rows, ring_slots, ec_chunk, C.cast(cb, C.c_void_p) if cb is not None else None, None, C.byref(n))
        if err:
            raise err[0]
        _check(rc)
        return n.value

    def gather_closed_form_inputs(self, comm, rank=0, world=1, root=0):
        """the multi-GPU path's one collective: records [n][24] = [circuit type, instance, compact form (18), public input (4)]
        in emission order on the root, None elsewhere"""
        total = sum(self.num_instances(t) for t in (4, 8, 10, 2, 3, 5, 6, 7, 9, 11, 12, 13))  # the synthesized types, zkw_block.hip kOrder
        out = np.zeros((total, 24), np.uint64)
        n = C.c_size_t(0)
        _check(load().zkw_block_gather_closed_form_inputs(self.handle, comm.handle, rank, world, root, _np_ptr(out), total, C.byref(n)))
        assert n.value == total
        return out if rank == root else None

    def check_satisfied(self, circuit_type, trace_handle, slot, ctx=None):
        """check_if_satisfied (src/tests/mod.rs:130-259) on a slot handed to a synthesize callback, through the type-dispatching
        entry point zkw_check_satisfied: (n_violations, first_bad). `ctx`: the Context the check runs on — REQUIRED inside a callback of
        synthesize_many (one per calling thread): the block's own contexts are busy there (a block's ECRecover instances arrive from another
        thread than its other types, and both would use the block's precompile context). Default: the block's context of that type, which is
        fine for Block.synthesize (one thread, one instance at a time)."""
        lib = load()
        bad, first = C.c_uint64(0), C.c_uint64(0)
        cap = self.capacities[circuit_type]
        h = ctx.handle if ctx is not None else lib.zkw_block_context(self.handle, circuit_type)
        _check(lib.zkw_check_satisfied(h, circuit_type, trace_handle, slot, cap, C.byref(bad), C.byref(first)))
        return bad.value, first.value

    def free(self):
        if self.handle:
            load().zkw_block_free(self.handle)
            self.handle = C.c_void_p(None)

    @staticmethod
    def free_many(blocks):
        """zkw_blocks_free: the blocks released on a few threads of the library"""
        live = [b for b in blocks if b is not None and b.handle]
        if not live:
            return
        hs = (C.c_void_p * len(live))(*[b.handle for b in live])
        load().zkw_blocks_free(hs, len(live))
        for b in live:
            b.handle = C.c_void_p(None)

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


ROW_HAS_GATES, ROW_HEADER, ROW_BOUNDARY, ROW_PADDING = 0x40, 0x80, 0xC0, 0xFF


def setup_row_selectors(circuit_type, capacity=0, n_rows=1 << 20):
    """zkw_setup_row_selectors: the selector (row type / lookup table id) of every row of this library's layout; no GPU needed"""
    out = np.zeros(n_rows, np.uint8)
    _check(load().zkw_setup_row_selectors(circuit_type, capacity, n_rows, _np_ptr(out)))
    return out


def setup_lookup_tables(circuit_type, n_rows):
    """zkw_setup_lookup_tables: [width + 1][n_rows]: the stacked lookup table's cells, then the table-id column; no GPU needed"""
    nc = C.c_uint32(0)
    _check(load().zkw_setup_lookup_tables(circuit_type, 0, None, C.byref(nc)))
    out = np.zeros((nc.value, n_rows), np.uint64)
    _check(load().zkw_setup_lookup_tables(circuit_type, n_rows, _np_ptr(out), C.byref(nc)))
    return out


def setup_copy_permutation(circuit_type, capacity, n_rows):
    """zkw_setup_copy_permutation: sigma [n_columns][n_rows] (cell ids) of a queue circuit's layout; no GPU needed"""
    ncol = C.c_uint32(0)
    _check(load().zkw_setup_copy_permutation(circuit_type, capacity, n_rows, None, C.byref(ncol)))
    sigma = np.zeros((ncol.value, n_rows), np.uint64)
    _check(load().zkw_setup_copy_permutation(circuit_type, capacity, n_rows, _np_ptr(sigma), C.byref(ncol)))
    return sigma


def recursion_queue_split(states, arity=32):
    """zkw_recursion_queue_split: the RecursionLeafInput queue state of every leaf (split_by(RECURSION_ARITY)); no GPU needed"""
    st = np.ascontiguousarray(states, dtype=np.uint64).reshape(-1, 12)
    n_leaves = (st.shape[0] + arity - 1) // arity
    out = np.zeros(n_leaves, QUEUE_STATE12)
    n = C.c_size_t(0)
    _check(load().zkw_recursion_queue_split(_np_ptr(st) if st.size else None, st.shape[0], arity, _np_ptr(out) if n_leaves else None, n_leaves, C.byref(n)))
    assert n.value == n_leaves
    return out


LEAF_PARAMS = np.dtype([("circuit_type", "<u8"), ("basic_circuit_vk_commitment", "<u8", 4), ("leaf_layer_vk_commitment", "<u8", 4)])
QUEUE_TAIL12 = np.dtype([("tail", "<u8", 12), ("length", "<u4"), ("_pad", "<u4")])


def vk_commitment(ctx, cap):
    """zkw_vk_commitment: commitment of a verification key = of its setup_merkle_tree_cap [cap_size][4]"""
    cap = np.ascontiguousarray(cap, dtype=np.uint64).reshape(-1, 4)
    out = np.zeros(4, np.uint64)
    _check(load().zkw_vk_commitment(ctx.handle, _np_ptr(cap), cap.shape[0], _np_ptr(out)))
    return out


def compute_leaf_params(ctx, circuit_type, base_layer_cap, leaf_layer_cap):
    """compute_leaf_params (recursive_aggregation.rs:163-216)"""
    b = np.ascontiguousarray(base_layer_cap, dtype=np.uint64).reshape(-1, 4)
    l = np.ascontiguousarray(leaf_layer_cap, dtype=np.uint64).reshape(-1, 4)
    assert b.shape == l.shape
    out = np.zeros(1, LEAF_PARAMS)
    _check(load().zkw_compute_leaf_params(ctx.handle, circuit_type, _np_ptr(b), _np_ptr(l), b.shape[0], _np_ptr(out)))
    return out


def leaf_vks_and_params_commitment(ctx, leaf_params):
    p = np.ascontiguousarray(leaf_params, dtype=LEAF_PARAMS).reshape(13)
    out = np.zeros(4, np.uint64)
    _check(load().zkw_leaf_vks_and_params_commitment(ctx.handle, _np_ptr(p), _np_ptr(out)))
    return out


def create_leaf_witnesses(ctx, params, public_inputs, queue_tail_in=None):
    """create_leaf_witnesses (recursive_aggregation.rs:71-161) -> dict(enc, states, leaf_states, leaf_public_inputs)"""
    p = np.ascontiguousarray(params, dtype=LEAF_PARAMS).reshape(1)
    pi = np.ascontiguousarray(public_inputs, dtype=np.uint64).reshape(-1, 4)
    n = pi.shape[0]
    n_leaves = (n + 31) // 32
    enc, states = np.zeros((n, 8), np.uint64), np.zeros((n, 12), np.uint64)
    leaf_states, leaf_pi = np.zeros(n_leaves, QUEUE_STATE12), np.zeros((n_leaves, 4), np.uint64)
    tin = None if queue_tail_in is None else _np_ptr(_u64(queue_tail_in))
    got = C.c_size_t(0)
    opt = lambda a: _np_ptr(a) if a.size else None
    _check(load().zkw_create_leaf_witnesses(ctx.handle, _np_ptr(p), opt(pi), n, tin, opt(enc), opt(states), opt(leaf_states), opt(leaf_pi),
                                            n_leaves, C.byref(got)))
    assert got.value == n_leaves
    return {"enc": enc, "states": states, "leaf_states": leaf_states, "leaf_public_inputs": leaf_pi}


def create_node_witnesses(ctx, branch_circuit_type, leaf_layer_params, node_layer_vk_commitment, chunks):
    """create_node_witnesses (recursive_aggregation.rs:270-421) -> dict(node_states, split_points [n][31], node_public_inputs)"""
    p = np.ascontiguousarray(leaf_layer_params, dtype=LEAF_PARAMS).reshape(13)
    ch = np.ascontiguousarray(chunks, dtype=QUEUE_STATE12)
    nvk = np.ascontiguousarray(node_layer_vk_commitment, dtype=np.uint64).reshape(4)
    n_nodes = (ch.size + 31) // 32
    st, sp, pi = np.zeros(n_nodes, QUEUE_STATE12), np.zeros((n_nodes, 31), QUEUE_TAIL12), np.zeros((n_nodes, 4), np.uint64)
    got = C.c_size_t(0)
    _check(load().zkw_create_node_witnesses(ctx.handle, branch_circuit_type, _np_ptr(p), _np_ptr(nvk), _np_ptr(ch) if ch.size else None, ch.size,
                                            _np_ptr(st) if n_nodes else None, _np_ptr(sp) if n_nodes else None, _np_ptr(pi) if n_nodes else None,
                                            n_nodes, C.byref(got)))
    return {"node_states": st, "split_points": sp, "node_public_inputs": pi}


def trim_caches():
    """zkw_trim_caches: hand every idle buffer / stream of the library's caches back to the HIP runtime"""
    load().zkw_trim_caches()


def shard_lpt(circuit_types, world):
    """zkw_shard_lpt: owner rank of every instance of an ordered instance list (no GPU needed)"""
    t = np.ascontiguousarray(circuit_types, dtype=np.uint8)
    owner = np.zeros(t.size, np.uint32)
    _check(load().zkw_shard_lpt(_np_ptr(t), t.size, world, _np_ptr(owner)))
    return [int(x) for x in owner]


class Comm:
    """zkw_comm: the RCCL communicator behind zkw_gather_closed_form_inputs. `unique_id`: bytes from Comm.unique_id() on
    rank 0, handed to the other ranks by the host (here: torch.distributed broadcast); world == 1 needs none."""

    @staticmethod
    def unique_id() -> bytes:
        buf = np.zeros(128, np.uint8)
        _check(load().zkw_comm_unique_id(_np_ptr(buf)))
        return buf.tobytes()

    def __init__(self, ctx, rank=0, world=1, unique_id=None):
        self.handle = C.c_void_p(None)
        idbuf = np.frombuffer(unique_id, np.uint8).copy() if unique_id is not None else None
        _check(load().zkw_comm_init(ctx.handle, _np_ptr(idbuf) if idbuf is not None else None, rank, world, C.byref(self.handle)))
        self.ctx, self.rank, self.world = ctx, rank, world

    @classmethod
    def tcp(cls, ctx, address, port, rank, world, timeout_ms=30000):
        """zkw_comm_init_tcp: the same collective over sockets; ctx=None -> host-memory communicator (needs no GPU)"""
        self = cls.__new__(cls)
        self.handle = C.c_void_p(None)
        _check(load().zkw_comm_init_tcp(ctx.handle if ctx is not None else None, address.encode(), port, rank, world, timeout_ms, C.byref(self.handle)))
        self.ctx, self.rank, self.world = ctx, rank, world
        return self

    @classmethod
    def rccl(cls, ctx, rank=0, world=1, unique_id=None):
        """zkw_comm_init_rccl: the RCCL transport whatever the world size (world == 1: ncclCommInitRank over one rank)"""
        self = cls.__new__(cls)
        self.handle = C.c_void_p(None)
        idbuf = np.frombuffer(unique_id if unique_id is not None else cls.unique_id(), np.uint8).copy()
        _check(load().zkw_comm_init_rccl(ctx.handle, _np_ptr(idbuf), rank, world, C.byref(self.handle)))
        self.ctx, self.rank, self.world = ctx, rank, world
        return self

    def exchange(self, src_ptr, dst_ptr, nbytes, peer):
        """zkw_comm_exchange: grouped send to + receive from `peer` (device pointers; enqueued on the stream)"""
        _check(load().zkw_comm_exchange(self.handle, C.c_void_p(src_ptr), C.c_void_p(dst_ptr), nbytes, peer))

    def synchronize(self):
        _check(load().zkw_comm_synchronize(self.handle))

    def gather_records(self, owner, mine, root=0):
        """zkw_gather_records: owner[n] (zkw_shard_lpt), mine = this rank's records [k][w] u64 in list order (host);
        returns the n records in list order on the root, None elsewhere"""
        owner = np.ascontiguousarray(owner, dtype=np.uint32)
        mine = np.ascontiguousarray(mine, dtype=np.uint64)
        assert mine.ndim == 2 and mine.shape[0] == int((owner == self.rank).sum())
        out = np.zeros((owner.size, mine.shape[1]), np.uint64) if self.rank == root else None
        _check(load().zkw_gather_records(self.handle, _np_ptr(owner) if owner.size else None, owner.size, _np_ptr(mine) if mine.size else None,
                                         mine.shape[1] * 8, root, _np_ptr(out) if out is not None and out.size else None))
        return out

    def gather(self, records_dev_ptr, counts, record_bytes, root, recv_dev_ptr):
        cnt = np.ascontiguousarray(counts, dtype=np.uint64)
        _check(load().zkw_gather_closed_form_inputs(self.handle, C.c_void_p(records_dev_ptr), _np_ptr(cnt), record_bytes, root,
                                                    C.c_void_p(recv_dev_ptr) if recv_dev_ptr else None))

    def destroy(self):
        if self.handle:
            load().zkw_comm_destroy(self.handle)
            self.handle = C.c_void_p(None)

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass
