"""GPU: zkw_closed_form_public_inputs (ClosedFormInputCompactForm + public input, postprocessing/mod.rs:353-369) for the
circuits whose builders keep no compact forms (3, 5, 6, 7, 10, 13) against the oracle on the same instance records —
records produced by the ORACLE's builders, two blocks back to back so that the 'first instance of the block' rule is used."""
import numpy as np
import pytest

from era_zkevm_test_harness_amd import synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from era_zkevm_test_harness_amd import native

    c = native.Context(0)
    yield c
    c.close()


def _records(oracle, seed):
    from oracle import block as ob

    b = synthetic.block_after_vm(seed=seed)
    caps = {ob.DECOMMITS_SORTER: 5, ob.CODE_DECOMMITTER: 7, ob.LOG_DEMUXER: 64, ob.KECCAK256: 3, ob.SHA256: 4, ob.ECRECOVER: 2,
            ob.RAM_PERMUTATION: 1000, ob.STORAGE_SORTER: 40, ob.STORAGE_APPLICATION: 5, ob.EVENTS_SORTER: 16, ob.L1_MESSAGES_SORTER: 9}
    tree = oracle.Tree()
    rng = np.random.default_rng(seed)
    for _ in range(10):
        tree.insert_leaf(rng.bytes(32), rng.bytes(32))
    a = ob.create_artifacts_after_vm(b, caps)
    sto = a["witnesses"]["storage_sorter"]
    for q in sto["result_q"]:
        if q["read_value"].any():
            tree.insert_leaf(oracle.derive_final_address(q), b"".join(int(x).to_bytes(4, "big") for x in q["read_value"][::-1]))
    sap = oracle.storage_application_build(tree, sto["result_q"], sto["result_new_tails"], 5)
    w = a["witnesses"]
    return {3: w["code_decommitter"]["instances"], 5: w["keccak256"]["instances"], 6: w["sha256"]["instances"],
            7: w["ecrecover"]["instances"], 10: sap["instances"], 13: w["l1_messages_hasher"]["instances"]}


def test_closed_forms_match_the_oracle(ctx, oracle):
    r1, r2 = _records(oracle, 1), _records(oracle, 5)
    for ctype in (3, 5, 6, 7, 10, 13):
        inst = np.concatenate([r1[ctype], r2[ctype]])
        assert inst["start_flag"].sum() == 2 and inst.size >= 2
        exp_cf, exp_pi = oracle.closed_form_public_inputs(ctype, inst)
        cf, pi = ctx.closed_form_public_inputs(ctype, inst)
        assert np.array_equal(cf, exp_cf), ctype
        assert np.array_equal(pi, exp_pi), ctype
        # the second block's observable input is its own first instance's, not the first block's
        k = r1[ctype].size
        assert not np.array_equal(cf[0, 2:6], cf[k, 2:6]) and (cf[k:, 2:6] == cf[k, 2:6]).all()


def test_closed_forms_edge_cases(ctx):
    from era_zkevm_test_harness_amd import native as nv

    cf, pi = ctx.closed_form_public_inputs(7, np.zeros(0, nv.PRECOMPILE_INSTANCE))
    assert cf.shape == (0, 18) and pi.shape == (0, 4)
    lib = nv.load()
    rec = np.zeros(1, nv.PRECOMPILE_INSTANCE)
    out = np.zeros(22, np.uint64)
    for bad in (0, 1, 2, 4, 8, 9, 11, 12, 14):  # MainVM and the types that carry their compact forms in the witness
        rc = lib.zkw_closed_form_public_inputs(ctx.handle, bad, rec.ctypes.data, 1, out.ctypes.data, out[18:].ctypes.data)
        assert rc == nv.ERR_INVALID, bad
