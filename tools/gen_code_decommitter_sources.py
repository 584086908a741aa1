#!/usr/bin/env python3
"""The CodeDecommitter circuit (type 3) is the Sha256RoundFunction netlist at 18 lookups per row (include/
zkw_code_decommitter_circuit_spec.h, tools/gen_sha256_circuit.py): its oracle and its kernels are the SHA-256 ones with the
DC_ layout constants instead of SC_. This script derives
    oracle/code_decommitter_circuit.c                              from oracle/sha256_circuit.c
    era_zkevm_test_harness_amd/csrc/code_decommitter_circuit_kernels.cuh   from .../csrc/sha256_circuit_kernels.cuh
by renaming; tests/test_spec_generators.py checks that the committed copies are current."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RENAMES = [(r"\bSC_", "DC_"), (r"\bsc_op\b", "dc_op"), (r"\bsc_gate\b", "dc_gate"), (r"\bg_sc_", "g_dc_"), (r"\bk_sc_", "k_dc_"),
           (r"\bScSynthJob\b", "DcSynthJob"), (r"\bScHistPlan\b", "DcHistPlan"), (r"\bsc_hist_plan\b", "dc_hist_plan"), (r"\bc_sc_", "c_dc_"), (r"\bsc_table\b", "dc_table"), (r"\bsc_prev_byte\b", "dc_prev_byte"),
           (r"\bsc_pads_per_cycle\b", "dc_pads_per_cycle"), (r"orc_sha256_round_", "orc_code_decommitter_round_"),
           (r"zkw_sha256_circuit_spec\.h", "zkw_code_decommitter_circuit_spec.h")]


def derive(src, dst, banner):
    text = open(os.path.join(ROOT, src)).read()
    for pat, rep in RENAMES:
        text = re.sub(pat, rep, text)
    open(os.path.join(ROOT, dst), "w").write(banner + text)
    print(dst)


def main():
    derive("oracle/sha256_circuit.c", "oracle/code_decommitter_circuit.c",
           "/* GENERATED from oracle/sha256_circuit.c by tools/gen_code_decommitter_sources.py (SC_ -> DC_: the same netlist at 18 lookups\n"
           " * per row, CodeDecommitter circuit, type 3) — do not edit. */\n")
    derive("era_zkevm_test_harness_amd/csrc/sha256_circuit_kernels.cuh", "era_zkevm_test_harness_amd/csrc/code_decommitter_circuit_kernels.cuh",
           "// GENERATED from sha256_circuit_kernels.cuh by tools/gen_code_decommitter_sources.py (SC_ -> DC_: the same netlist at 18 lookups\n"
           "// per row, CodeDecommitter circuit, type 3) — do not edit.\n")


if __name__ == "__main__":
    main()
