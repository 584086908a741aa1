"""CPU: the disk formats around the hot path (wire.py) reproduce the reference's committed artifacts byte for byte, and the
layout descriptions of this library (zkw_circuit_layout_of) are consistent with the geometry and comparable with the
reference's finalization hints."""
import json
import os

import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("fname,ctype", [("finalization_hint_8.json", 8), ("finalization_hint_11.json", 11), ("vk_8.json", 8)])
def test_reference_artifacts_round_trip_byte_exact(fname, ctype):
    from era_zkevm_test_harness_amd import wire

    raw = open(os.path.join(GOLD, "reference_setup", fname)).read()
    t, payload = wire.loads(raw)
    assert t == ctype
    assert wire.dumps(t, payload) == raw
    if fname.startswith("vk"):
        fp = payload["fixed_parameters"]
        assert fp["domain_size"] == 1 << 20 and fp["parameters"]["num_columns_under_copy_permutation"] == 133
        assert len(payload["setup_merkle_tree_cap"]) == 16 and all(len(d) == 4 for d in payload["setup_merkle_tree_cap"])


def test_layouts_against_reference_hints(tmp_path):
    from era_zkevm_test_harness_amd import native, wire

    ref = {int(k): v for k, v in json.load(open(os.path.join(GOLD, "reference_finalization_hints.json"))).items()}
    assert {t: ref[t]["name"] for t in ref} == wire.CIRCUIT_NAMES
    table = wire.rows_used_table(ref)
    synth = {t for t, (_, _, ours) in table.items() if ours is not None}
    assert synth == {2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13}
    for t in (3, 5, 6, 7, 10, 13):  # the netlist circuits: cycle-major, their own column counts
        lay = native.circuit_layout(t)
        assert lay["fits"] and int(lay["region_stride"]) == 0 and int(lay["rows_used"]) + int(lay["nop_rows"]) == 1 << 20
        assert int(lay["rows_used"]) == table[t][2] <= (1 << 20)
        # the PI row closes the netlist part; types 6, 3, 5 and 13 carry their queue section (Poseidon2 rows of the pops / pushes) below it
        # (type 7 also its EC section below that, and its lookup tables are longer than its gates: rows_used = 197 632 table rows)
        assert (int(lay["queue_rows_per_cycle"]) > 0) == (t in (6, 3, 5, 13, 7))
        # the closed-form section (flags, words, ties, commitment sponges: docs/KERNELS.md 3.22) closes every one of them
        assert int(lay["closed_form_rows"]) > int(lay["closed_form_header_rows"]) > 0
        assert int(lay["closed_form_first_row"]) + int(lay["closed_form_rows"]) == int(lay["rows_used"]) or t == 7
        pi_row = int(lay["queue_first_row"]) - 1 if int(lay["queue_rows_per_cycle"]) else int(lay["closed_form_first_row"]) - 1
        assert wire.finalization_hint_of_layout(t)["public_inputs"][0][1] == pi_row
        if t == 7:
            assert int(lay["ec_first_row"]) + int(lay["capacity"]) * int(lay["ec_rows_per_cycle"]) <= int(lay["rows_used"]) == int(lay["total_table_rows"])
    for t in sorted(synth - {3, 5, 6, 7, 10, 13}):
        name, ref_rows, ours = table[t]
        lay = native.circuit_layout(t)
        geo = native.circuit_geometry(t)
        assert lay["fits"] and int(lay["capacity"]) == int(geo["capacity"]) and int(lay["trace_len"]) == 1 << 20
        assert int(lay["rows_used"]) + int(lay["nop_rows"]) == 1 << 20 == ref[t]["final_trace_len"]
        assert int(lay["rows_used"]) == int(lay["rows_per_cycle"]) * int(lay["region_stride"]) + (ours - int(lay["rows_per_cycle"]) * int(lay["region_stride"]))
        assert int(lay["num_columns"]) == int(geo["num_columns_under_copy_permutation"]) + int(geo["lookup_width"]) * int(geo["lookup_repetitions"]) + 1
        # the reference's own rows: PI row + 1 + nop == 2^20
        assert ref_rows + ref[t]["nop_gates_to_add"] == 1 << 20
        # same file shape as the reference's hint, for this library's layout; and it reads back
        hint = wire.finalization_hint_of_layout(t)
        assert set(hint) == {"row_finalization_hints", "column_finalization_hints", "nop_gates_to_add", "final_trace_len", "public_inputs"}
        assert [c for c, _ in hint["public_inputs"]] == [0, 1, 2, 3] and len({r for _, r in hint["public_inputs"]}) == 1
        path = os.path.join(tmp_path, os.path.basename(wire.base_layer_paths(str(tmp_path), t)[1]))
        wire.dump(path, t, hint)
        assert wire.load(path) == (t, hint)
    assert not native.circuit_layout(1)["synthesizable"]
    with pytest.raises(native.ZkwError):
        native.circuit_layout(14)


def test_verification_key_of_a_layout_has_the_reference_shape():
    """wire.verification_key_payload: the key of this library's RAMPermutation layout next to the reference's vk_8.json (a fixture copy of
    setup/base_layer/vk_8.json): same fields, same geometry / lookup parameters / domain / table length / cap size; the PI cells are the
    layout's, the selector description is this library's single row-type column"""
    import numpy as np

    from era_zkevm_test_harness_amd import native, wire

    ref_type, ref = wire.load(os.path.join(GOLD, "reference_setup", "vk_8.json"))
    assert ref_type == 8
    cap = np.arange(64, dtype=np.uint64).reshape(16, 4)
    vk = wire.verification_key_payload(8, cap)
    assert list(vk) == list(ref) and list(vk["fixed_parameters"]) == list(ref["fixed_parameters"])
    fp, rp = vk["fixed_parameters"], ref["fixed_parameters"]
    for k in ("lookup_parameters", "domain_size", "total_tables_len", "fri_lde_factor", "cap_size", "quotient_degree"):
        assert fp[k] == rp[k], k
    for k in ("num_columns_under_copy_permutation", "num_witness_columns", "max_allowed_constraint_degree"):
        assert fp["parameters"][k] == rp["parameters"][k], k
    lay = native.circuit_layout(8)
    assert fp["public_inputs_locations"] == [[int(c), int(r)] for c, r in zip(lay["public_input_column"], lay["public_input_row"])]
    assert fp["parameters"]["num_constant_columns"] == 1 and list(fp["selectors_placement"]) == ["RowTypeColumn"]
    assert vk["setup_merkle_tree_cap"] == cap.tolist()
    assert wire.loads(wire.dumps(8, vk)) == (8, vk)
    for t in (2, 3, 4, 5, 6, 7, 9, 10, 11, 12, 13):
        assert wire.verification_key_payload(t, cap)["fixed_parameters"]["parameters"]["num_columns_under_copy_permutation"] == native.circuit_geometry(t)["num_columns_under_copy_permutation"]
    with pytest.raises(ValueError):
        wire.verification_key_payload(1, cap)  # MainVM has no layout here
    with pytest.raises(ValueError):
        wire.verification_key_payload(8, cap[:3])
