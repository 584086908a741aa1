"""CPU-side checks of the drop-in boundary: libzkw.so loads and exports every symbol include/zkw.h
declares; record layouts agree between the header, numpy and the oracle; no compute without a GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built_lib():
    from era_zkevm_test_harness_amd import build

    return build.build()


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "zkw.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(zkw_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(built_lib):
    from era_zkevm_test_harness_amd import native

    lib = ctypes.CDLL(built_lib)
    declared = _declared_symbols()
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/zkw.h but not exported"
    assert sorted(n for n, _, _ in native.SYMBOLS) == declared
    native.load()


def test_no_cpu_fallback_without_gpu(built_lib):
    import torch

    from era_zkevm_test_harness_amd import native

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(native.ZkwError) as ei:
        native.Context(0)
    assert ei.value.code == native.ERR_NO_DEVICE
    assert "no CPU fallback" in str(ei.value)


def test_record_layouts_match_header():
    from era_zkevm_test_harness_amd import native
    from oracle import pyoracle

    src = r"""
    #include <stdio.h>
    #include <stddef.h>
    #include "zkw_types.h"
    int main(void){
      printf("%zu %zu %zu %zu\n", sizeof(zkw_mem_query), sizeof(zkw_queue_state12), sizeof(zkw_ram_fsm), sizeof(zkw_ram_instance));
      printf("%zu %zu %zu %zu\n", sizeof(zkw_callstack_entry), offsetof(zkw_callstack_entry, pc), sizeof(zkw_log_query), sizeof(zkw_decommit_query));
      printf("%zu %zu %zu %zu\n", sizeof(zkw_precompile_fsm), sizeof(zkw_precompile_instance), offsetof(zkw_precompile_fsm, buffer_bytes), offsetof(zkw_precompile_instance, first_request));
      printf("%zu %zu %zu %zu\n", sizeof(zkw_storage_application_fsm), sizeof(zkw_storage_application_instance), offsetof(zkw_storage_application_instance, hidden_fsm_input), offsetof(zkw_storage_application_instance, first_item));
      printf("%zu %zu %zu %zu\n", offsetof(zkw_mem_query, value), offsetof(zkw_ram_fsm, previous_sorting_key),
             offsetof(zkw_ram_instance, hidden_fsm_input), offsetof(zkw_ram_instance, first_item));
      printf("%zu %zu %zu %zu\n", sizeof(zkw_vm_instance), sizeof(zkw_vm_aux_parameters), sizeof(zkw_storage_log_detailed_state), sizeof(zkw_vm_tracer_streams));
      printf("%zu %zu %zu %zu\n", offsetof(zkw_vm_instance, auxilary_final_parameters), offsetof(zkw_vm_instance, memory_queue_final_state),
             offsetof(zkw_vm_tracer_streams, vm_memory_queries), offsetof(zkw_vm_tracer_streams, global_end_of_storage_log));
      return 0; }
    """
    import subprocess
    import tempfile

    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", os.path.join(d, "t"), os.path.join(d, "t.c")])
        out = subprocess.check_output([os.path.join(d, "t")]).decode().split()
    sizes = [int(x) for x in out]
    for mod in (native, pyoracle):
        assert sizes[:4] == [mod.MEM_QUERY.itemsize, mod.QUEUE_STATE12.itemsize, mod.RAM_FSM.itemsize, mod.RAM_INSTANCE.itemsize]
        assert sizes[4:8] == [mod.CALLSTACK_ENTRY.itemsize, mod.CALLSTACK_ENTRY.fields["pc"][1], mod.LOG_QUERY.itemsize,
                              mod.DECOMMIT_QUERY.itemsize]
        assert sizes[8:12] == [mod.PRECOMPILE_FSM.itemsize, mod.PRECOMPILE_INSTANCE.itemsize,
                               mod.PRECOMPILE_FSM.fields["buffer_bytes"][1], mod.PRECOMPILE_INSTANCE.fields["first_request"][1]]
        assert sizes[12:16] == [mod.STORAGE_APPLICATION_FSM.itemsize, mod.STORAGE_APPLICATION_INSTANCE.itemsize,
                                mod.STORAGE_APPLICATION_INSTANCE.fields["hidden_fsm_input"][1],
                                mod.STORAGE_APPLICATION_INSTANCE.fields["first_item"][1]]
        assert sizes[16] == mod.MEM_QUERY.fields["value"][1]
        assert sizes[17] == mod.RAM_FSM.fields["previous_sorting_key"][1]
        assert sizes[18] == mod.RAM_INSTANCE.fields["hidden_fsm_input"][1]
        assert sizes[19] == mod.RAM_INSTANCE.fields["first_item"][1]
        assert sizes[20:24] == [mod.VM_INSTANCE.itemsize, mod.VM_AUX_PARAMETERS.itemsize, mod.STORAGE_LOG_DETAILED_STATE.itemsize,
                                ctypes.sizeof(mod.VmTracerStreams)]
        assert sizes[24:28] == [mod.VM_INSTANCE.fields["auxilary_final_parameters"][1], mod.VM_INSTANCE.fields["memory_queue_final_state"][1],
                                mod.VmTracerStreams.vm_memory_queries.offset, mod.VmTracerStreams.global_end_of_storage_log.offset]


def test_synthetic_trace_is_valid_memory():
    from era_zkevm_test_harness_amd import synthetic

    q = synthetic.ram_trace(5000, seed=9, pages=2, indices=64)
    assert np.all(np.diff(q["timestamp"].astype(np.int64)) > 0)
    mem = {}
    for rec in q:
        key = (int(rec["page"]), int(rec["index"]))
        val = (tuple(int(x) for x in rec["value"]), int(rec["value_is_pointer"]))
        if rec["rw_flag"]:
            mem[key] = val
        else:
            assert key in mem and mem[key] == val
    assert 0.5 < 1 - q["rw_flag"].mean() < 0.75


def test_circuit_geometry_table():
    """No GPU needed: the per-type geometry is a table. Witness bytes = (columns + 1 multiplicity) * 2^20 * 8."""
    from era_zkevm_test_harness_amd import native

    ram = native.circuit_geometry(8)
    assert (int(ram["num_columns_under_copy_permutation"]), int(ram["lookup_width"]), int(ram["lookup_repetitions"])) == (133, 1, 15)
    assert int(ram["capacity"]) == 136714 and int(ram["size_hint_variables"]) == (1 << 26) + (1 << 25)
    cols = lambda g: int(g["num_columns_under_copy_permutation"]) + int(g["lookup_width"]) * int(g["lookup_repetitions"])
    assert [cols(native.circuit_geometry(t)) for t in range(1, 14)] == [154, 148, 152, 150, 128, 152, 128, 148, 148, 138, 138, 138, 144]
    assert int(native.circuit_geometry(11)["max_allowed_constraint_degree"]) == 18
    with pytest.raises(native.ZkwError):
        native.circuit_geometry(14)
