"""Disk formats of the artifacts around the hot path: the reference stores verification keys, finalization
hints and proofs as `serde_json::to_string_pretty` of enums keyed by the circuit's name
(src/data_source/local_file_data_source.rs:51-56, 62-160; files setup/base_layer/{vk,finalization_hint}_N.json,
test_proofs/base_layer/basic_circuit_proof_T_I.json). This module reads and writes that format byte-for-byte (field
elements are plain u64, digests [u64; 4]) and emits, in the same format, the finalization hint of THIS library's layouts
(`zkw_circuit_layout_of`), so that a host written against `LocalFileDataSource` finds the files it expects. Host-side
code: no GPU, no oracle.
"""
import json
import os

from . import native

# BaseLayerCircuitType (circuit_definitions/src/circuit_definitions/base_layer/mod.rs:55-71) -> the variant names of
# ZkSyncBaseLayerStorage, which are the top-level keys of every stored artifact
CIRCUIT_NAMES = {1: "MainVM", 2: "CodeDecommittmentsSorter", 3: "CodeDecommitter", 4: "LogDemuxer", 5: "KeccakRoundFunction",
                 6: "Sha256RoundFunction", 7: "ECRecover", 8: "RAMPermutation", 9: "StorageSorter", 10: "StorageApplication",
                 11: "EventsSorter", 12: "L1MessagesSorter", 13: "L1MessagesHasher"}
CIRCUIT_TYPES = {v: k for k, v in CIRCUIT_NAMES.items()}


def loads(text):
    """(circuit type, payload) of one stored artifact (vk, finalization hint, proof)"""
    obj = json.loads(text)
    if len(obj) != 1:
        raise ValueError("a stored artifact is an enum: exactly one top-level key")
    name, payload = next(iter(obj.items()))
    if name not in CIRCUIT_TYPES:
        raise ValueError(f"unknown circuit name {name!r}")
    return CIRCUIT_TYPES[name], payload


def dumps(circuit_type, payload):
    """serde_json::to_string_pretty of the enum variant: two-space indent, one element per line, key order kept"""
    return json.dumps({CIRCUIT_NAMES[circuit_type]: payload}, indent=2)


def load(path):
    with open(path) as f:
        return loads(f.read())


def dump(path, circuit_type, payload):
    with open(path, "w") as f:
        f.write(dumps(circuit_type, payload))


def base_layer_paths(root, circuit_type, instance=None):
    """file names of LocalFileDataSource (local_file_data_source.rs:62-160)"""
    if instance is None:
        return (os.path.join(root, "setup", "base_layer", f"vk_{circuit_type}.json"),
                os.path.join(root, "setup", "base_layer", f"finalization_hint_{circuit_type}.json"))
    return os.path.join(root, "test_proofs", "base_layer", f"basic_circuit_proof_{circuit_type}_{instance}.json")


def finalization_hint_of_layout(circuit_type, capacity=0):
    """FinalizationHintsForProver of this library's layout of `circuit_type` in the reference's JSON shape:
    public_inputs = [(column, row)] x 4, nop_gates_to_add = zero-padding rows, final_trace_len = 2^20. The row / column
    finalization hints of boojum's resolver (variable-placement bookkeeping of ITS layout) have no counterpart: the
    layouts here are fixed tables, so those lists are empty."""
    lay = native.circuit_layout(circuit_type, capacity)
    if not lay["synthesizable"]:
        raise ValueError(f"circuit type {circuit_type} has no layout in this library yet")
    return {"row_finalization_hints": [[]], "column_finalization_hints": [[]], "nop_gates_to_add": int(lay["nop_rows"]),
            "final_trace_len": int(lay["trace_len"]),
            "public_inputs": [[int(c), int(r)] for c, r in zip(lay["public_input_column"], lay["public_input_row"])]}


def verification_key_payload(circuit_type, setup_merkle_tree_cap, capacity=0, lde_factor=2):
    """VerificationKey of THIS library's layout of `circuit_type` in the reference's JSON shape (vk_N.json: fixed_parameters +
    setup_merkle_tree_cap; written by generate_base_layer_vks_and_proofs, src/tests/complex_tests/mod.rs:560-640, read back by
    LocalFileDataSource). `setup_merkle_tree_cap`: [cap_size][4], what zkw_setup_commit returns for the layout's setup columns (sigma
    columns, the selector column, the lookup-table columns). Host-side, no GPU.
    Field by field against the reference's key: geometry, lookup parameters, domain size, total_tables_len and cap size are the wrapper's
    (zkw_circuit_geometry_of: the layouts keep the reference's column counts); public_inputs_locations are the layout's PI cells. The
    layouts select gates by ONE constant column holding the row's type (zkw_setup_row_selectors) instead of boojum's tree of selector
    polynomials, so num_constant_columns = 1, extra_constant_polys_for_selectors = 0, the table id rides in the lookup-table columns
    (table_ids_column_idxes = []), and selectors_placement is a variant of its own ({"RowTypeColumn": ...}) that names the column and
    how many distinct values it takes. A key of this library's layout does not pair with the reference's proofs."""
    geo = native.circuit_geometry(circuit_type)
    lay = native.circuit_layout(circuit_type, capacity)
    if not lay["synthesizable"]:
        raise ValueError(f"circuit type {circuit_type} has no layout in this library yet")
    cap = [[int(x) for x in digest] for digest in setup_merkle_tree_cap]
    if not cap or any(len(d) != 4 for d in cap) or len(cap) & (len(cap) - 1):
        raise ValueError("setup_merkle_tree_cap: a power-of-two number of 4-element digests")
    n_rows = int(lay["trace_len"])
    sel = native.setup_row_selectors(circuit_type, int(lay["capacity"]), n_rows)
    degree = int(geo["max_allowed_constraint_degree"])
    return {
        "fixed_parameters": {
            "parameters": {"num_columns_under_copy_permutation": int(geo["num_columns_under_copy_permutation"]),
                           "num_witness_columns": int(geo["num_witness_columns"]), "num_constant_columns": 1,
                           "max_allowed_constraint_degree": degree},
            "lookup_parameters": {"UseSpecializedColumnsWithTableIdAsConstant": {
                "width": int(geo["lookup_width"]), "num_repetitions": int(geo["lookup_repetitions"]), "share_table_id": True}},
            "domain_size": n_rows,
            "total_tables_len": int(lay["total_table_rows"]),
            "public_inputs_locations": [[int(c), int(r)] for c, r in zip(lay["public_input_column"], lay["public_input_row"])],
            "extra_constant_polys_for_selectors": 0,
            "table_ids_column_idxes": [],
            "quotient_degree": degree,
            "selectors_placement": {"RowTypeColumn": {"column": 0, "num_values": int(len(set(sel.tolist())))}},
            "fri_lde_factor": int(lde_factor),
            "cap_size": len(cap),
        },
        "setup_merkle_tree_cap": cap,
    }


def verification_key_of_layout(ctx, circuit_type, capacity=0, lde_factor=2, cap_size=16):
    """the key of this library's layout with the cap computed on the device (zkw_setup_commit: setup columns -> monomial form -> LDE ->
    Poseidon2 Merkle tree -> cap; prover_utils.rs:48-197 is the reference's route to the same field of its key)"""
    lay = native.circuit_layout(circuit_type, capacity)
    log_n = int(lay["trace_len"]).bit_length() - 1
    cap = ctx.setup_commit(circuit_type, int(lay["capacity"]), log_n, lde_factor, cap_size)
    return verification_key_payload(circuit_type, cap, capacity, lde_factor)


def rows_used_table(reference_hints):
    """rows used per circuit type: the reference's layout (from its committed hints: PI row + 1) next to this library's
    (None where there is no layout yet); `reference_hints`: {type: hint payload}"""
    out = {}
    for t in range(1, 14):
        ref = reference_hints[t]["public_inputs"][0][1] + 1
        lay = native.circuit_layout(t)
        out[t] = (CIRCUIT_NAMES[t], ref, int(lay["rows_used"]) if lay["synthesizable"] else None)
    return out
