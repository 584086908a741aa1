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


def section_tampers(header, prefix, rows_per_cycle, capacity, extra=(), challenges=True):
    """the same for a circuit whose section is built with dsl.Selections (flag cells named `flag`, an observable output gated by
    completion): (name, col, row). extra: (name, cell macro, row name) of circuit-specific cells."""
    M = spec_macros(header, prefix)
    b = rows_per_cycle * ((capacity + 63) // 64 * 64)
    r = lambda name: b + M["ROWOFF_" + name]  # noqa: E731
    last_cp = max(int(k[9:]) for k in M if k.startswith("ROWOFF_CP"))
    out = []
    if challenges:
        last_ch = max(int(k[9:]) for k in M if k.startswith("ROWOFF_CH"))
        out = [("challenge in BND_IN", M["BND_IN_G_c0_1"], r("BND_IN")), ("challenge in BND_IN (repetition 1)", M["BND_IN_G_c1_1"] + 2, r("BND_IN")),
               ("challenge sponge state", 60, r(f"CH{last_ch - 1}"))]
    out += [("public input", M["PI_pi0"] + 2, r("PI")), ("start flag", M["SEL0_flag"], r("SEL0")),
           ("selected value", M["SEL0_s0_t"], r("SEL0")), ("selection operand", M["SEL0_s0_b"], r("SEL0")),
           ("observable-input word", M["OI0_OI0_i0"] + 3, r("OI0")), ("FSM-input word", M["FI0_FI0_i0"] + 5, r("FI1")),
           ("FSM-output word", M["FO0_FO0_i0"] + 1, r("FO1")), ("commitment of the observable input", M["CP0_CP0_i0"] + 2, r("CP0")),
           ("completion flag", M["BND_OUT_completion"], r("BND_OUT")), ("completion flag copy", M["OSEL0_flag"], r("OSEL0")),
           ("observable-output word", M["OO0_OO0_i0"] + 1, r("OO0")),
           ("compact-form sponge output", M[f"CP{last_cp}_CP{last_cp}_o0"], r(f"CP{last_cp}"))]
    if "ROWOFF_SEL1" in M:
        out.append(("start flag copy", M["SEL1_flag"], r("SEL1")))
    return out + [(name, M[cell], r(row)) for name, cell, row in extra]


def decommit_sorter_tampers(capacity):
    return section_tampers("zkw_decommit_sorter_circuit_spec.h", "DS", 7, capacity,
                           extra=(("page byte of the open group", "GIN_gpage_b0", "GIN"), ("open group's encoding", "GIN_gge2", "GIN"),
                                  ("open-group flag", "BND_IN_gvalid", "BND_IN"), ("handed-over first-encountered timestamp", "GOUT_gof", "GOUT"),
                                  ("handed-over group's encoding", "GOUT_goge1", "GOUT")))


def events_sorter_tampers(capacity):
    return section_tampers("zkw_events_sorter_circuit_spec.h", "ES", 13, capacity,
                           extra=(("previous-record flag", "BND_IN_valid", "BND_IN"), ("handed-over previous key", "BND_OUT_kts", "BND_OUT"),
                                  ("previous record's normalised encoding (register)", "BND_IN_ne3", "BND_IN"), ("its re-derivation from the FSM input", "NIE_NI_ne9", "NIE"),
                                  ("a key byte of the FSM input's previous_item", "NIB1_NI_w7_b2", "NIB1"), ("a limb of the handed-over previous_item", "NOB0_NO_w5", "NOB0"),
                                  ("the handed-over record's encoding", "NOE_NO_ne17", "NOE")))


def storage_sorter_tampers(capacity):
    return section_tampers("zkw_storage_sorter_circuit_spec.h", "SS", 22, capacity,
                           extra=(("open-cell flag", "BND_IN_valid", "BND_IN"), ("handed-over depth", "BND_OUT_depth", "BND_OUT"),
                                  ("cycle index", "BND_IN_cidx", "BND_IN"), ("open cell's key chunk (register)", "BND_IN_kc4", "BND_IN"),
                                  ("its re-derivation from the FSM input", "KIE_KI_kc9", "KIE"), ("a byte of the FSM input's packed key", "KIB1_KI_p5_b1", "KIB1"),
                                  ("a limb of the handed-over packed key", "KOB0_KO_p2", "KOB0"), ("the handed-over key's chunk", "KOE_KO_kc17", "KOE"),
                                  ("previous_key word of the FSM input", "FI5_FI5_i6", "FI5")))


def log_demux_tampers(capacity):
    t = section_tampers("zkw_log_demux_circuit_spec.h", "LD", 12, capacity, challenges=False)
    return t


def n_boundary_rows(header, prefix):
    M = spec_macros(header, prefix)
    return M["NUM_ROW_TYPES"] - M["ROWS_PER_CYCLE"]


def check_section(synth, check, public_inputs, header, prefix, rows_per_cycle, capacity, n_instances, tampers, challenges=None):
    """shared body of the oracle-level section tests: every instance's trace is satisfied, its PI row (derived in-trace) is the
    builder's public input, the start flag is the instance's, the challenges of BND_IN are the challenge sponge's outputs
    (challenges: [2][1 + per_rep] of the builder), and each listed tampering is caught"""
    import numpy as np

    P = 0xFFFFFFFF00000001
    M = spec_macros(header, prefix)
    b = rows_per_cycle * ((capacity + 63) // 64 * 64)
    for idx in range(n_instances):
        t = synth(idx)
        assert check(t)[0] == 0
        assert np.array_equal(t[M["PI_pi0"]:M["PI_pi0"] + 4, b + M["ROWOFF_PI"]], public_inputs[idx]), idx
        assert int(t[M["SEL0_flag"], b + M["ROWOFF_SEL0"]]) == (1 if idx == 0 else 0)
        if challenges is not None:
            per_rep = challenges.shape[1] - 1
            n_absorb = 2 if per_rep == 20 else 4
            for rep in range(2):
                for k in range(1, per_rep + 1):
                    j = per_rep * rep + k - 1
                    row = n_absorb - 1 + j // 8
                    assert int(t[118 + j % 8, b + M[f"ROWOFF_CH{row}"]]) == int(challenges[rep, k])
                    assert int(t[M[f"BND_IN_G_c{rep}_{k}"], b + M["ROWOFF_BND_IN"]]) == int(challenges[rep, k])
        for name, c, r in tampers:
            t2 = t.copy()
            t2[c, r] = (int(t2[c, r]) + 1) % P
            assert check(t2)[0] > 0, (idx, name)


def gpu_tamper_parity(base, n_rows, host, gpu_check, oracle_check, tampers):
    """GPU parity body: write each tampered cell into the device trace (base = device address of the slot), compare the GPU checker's
    verdict with the oracle checker's on the same tampered trace, restore"""
    import ctypes as C

    import numpy as np

    P = 0xFFFFFFFF00000001
    hip = C.CDLL("libamdhip64.so")
    for name, c, r in tampers:
        addr = base + (int(c) * n_rows + int(r)) * 8
        new = np.array([(int(host[c, r]) + 1) % P], np.uint64)
        assert hip.hipMemcpy(C.c_void_p(addr), new.ctypes.data_as(C.c_void_p), C.c_size_t(8), 1) == 0
        bad, first = gpu_check()
        h2 = host.copy()
        h2[c, r] = new[0]
        obad, ofirst = oracle_check(h2)
        assert bad > 0 and bad == obad and first[0] == ofirst[0], (name, bad, first, obad, ofirst)
        assert hip.hipMemcpy(C.c_void_p(addr), np.array([host[c, r]], np.uint64).ctypes.data_as(C.c_void_p), C.c_size_t(8), 1) == 0
    assert gpu_check()[0] == 0
