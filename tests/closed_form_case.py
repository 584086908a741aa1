"""Cells of the closed-form sections (tools/gen_ram_circuit.py, class ClosedForm) whose tampering a checker must catch — shared by the
oracle tests and the GPU parity tests: what the reference derives inside the circuit (FS challenges, start-flag selection, commitments,
public input) and these traces therefore constrain."""
from era_zkevm_test_harness_amd.ram_circuit import boundary_row, spec_macros


def ram_closed_form_tampers(capacity):
    """(name, col, row)"""
    M = spec_macros()
    b = boundary_row(capacity)
    r = lambda name: b + M["ROWOFF_" + name]  # noqa: E731
    return [("challenge in BND_IN", M["BND_IN_G_c0_1"], r("BND_IN")), ("challenge in BND_IN (repetition 1)", M["BND_IN_G_c0_1"] + 8 + 3, r("BND_IN")),
            ("public input", M["PI_pi0"] + 2, r("PI")), ("start flag", M["SEL0_start"], r("SEL0")), ("start flag copy", M["SEL1_start1"], r("SEL1")),
            ("initial head / FSM head", M["BND_IN_uh0"], r("BND_IN")), ("selected value", M["SEL0_s0_t"], r("SEL0")),
            ("observable-input word", M["OI0_OI0_i0"] + 3, r("OI0")), ("FSM-input word", M["FI0_FI0_i0"] + 5, r("FI1")),
            ("FSM-output word", M["FO0_FO0_i0"] + 1, r("FO2")), ("commitment of the observable input", M["CP0_CP0_i0"] + 2, r("CP0")),
            ("completion flag", M["BND_OUT_completion"], r("BND_OUT")), ("value byte", M["VIN_VIN_v5_b0"], r("VIN")),
            ("challenge sponge state", 60, r("CH2")), ("compact-form sponge output", M["CP2_CP2_o0"], r("CP2"))]


def section_tampers(header, prefix, rows_per_cycle, capacity, extra=()):
    """the same for a circuit whose section is built with dsl.Selections (flag cells named `flag`, an observable output gated by
    completion): (name, col, row). extra: (name, cell macro, row name) of circuit-specific cells."""
    M = spec_macros(header, prefix)
    b = rows_per_cycle * ((capacity + 63) // 64 * 64)
    r = lambda name: b + M["ROWOFF_" + name]  # noqa: E731
    last_ch = max(int(k[9:]) for k in M if k.startswith("ROWOFF_CH"))
    last_cp = max(int(k[9:]) for k in M if k.startswith("ROWOFF_CP"))
    out = [("challenge in BND_IN", M["BND_IN_G_c0_1"], r("BND_IN")), ("challenge in BND_IN (repetition 1)", M["BND_IN_G_c1_1"] + 2, r("BND_IN")),
           ("public input", M["PI_pi0"] + 2, r("PI")), ("start flag", M["SEL0_flag"], r("SEL0")),
           ("selected value", M["SEL0_s0_t"], r("SEL0")), ("selection operand", M["SEL0_s0_b"], r("SEL0")),
           ("observable-input word", M["OI0_OI0_i0"] + 3, r("OI0")), ("FSM-input word", M["FI0_FI0_i0"] + 5, r("FI1")),
           ("FSM-output word", M["FO0_FO0_i0"] + 1, r("FO1")), ("commitment of the observable input", M["CP0_CP0_i0"] + 2, r("CP0")),
           ("completion flag", M["BND_OUT_completion"], r("BND_OUT")), ("completion flag copy", M["OSEL0_flag"], r("OSEL0")),
           ("observable-output word", M["OO0_OO0_i0"] + 1, r("OO0")),
           ("challenge sponge state", 60, r(f"CH{last_ch - 1}")), ("compact-form sponge output", M[f"CP{last_cp}_CP{last_cp}_o0"], r(f"CP{last_cp}"))]
    if "ROWOFF_SEL1" in M:
        out.append(("start flag copy", M["SEL1_flag"], r("SEL1")))
    return out + [(name, M[cell], r(row)) for name, cell, row in extra]


def decommit_sorter_tampers(capacity):
    return section_tampers("zkw_decommit_sorter_circuit_spec.h", "DS", 7, capacity,
                           extra=(("page byte of the open group", "GIN_gpage_b0", "GIN"), ("open group's encoding", "GIN_gge2", "GIN"),
                                  ("open-group flag", "BND_IN_gvalid", "BND_IN")))
