"""The DEVICE Poseidon2 against the reference's own committed artifacts — no oracle in the loop.

tests/golden/{merkle_pair_kat_ram,fri_leaf_kat_ram,reference_merkle_paths_kat}.json are Merkle paths of the proofs the
reference commits under test_proofs/ (hasher: `GoldilocksPoseidon2Sponge<AbsorptionModeOverwrite>`, src/prover_utils.rs:43;
harvested by tests/golden/make_reference_kats.py). A Merkle node is a one-item push into a zero full-width queue state, a
leaf hash is a chain of ceil(n/8) pushes with a zero-padded last chunk (the overwrite sponge) — i.e. exactly what the
queue-chain kernel (`zkw_queue_push_chain_full_batch`, FullWidthQueueSimulator::push, circuit_encodings/src/lib.rs:391-429)
computes, in each of its three layouts: one state per 16-lane DPP row (`p2::Coop`), per quad (`p2::Coop4`), per lane
(`p2::permute`)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
FORMS = [16, 4, 2, 1]


@pytest.fixture(scope="module")
def ctx():
    from era_zkevm_test_harness_amd import native

    c = native.Context(0)
    yield c
    c.close()


def _chains(ctx, form, items):
    """items: list of u64 sequences; returns the 4-word digest of the overwrite sponge over each (one launch)."""
    lens, rows = [], []
    for it in items:
        it = list(it)
        it += [0] * (-len(it) % 8)
        lens.append(len(it) // 8)
        rows.append(np.array(it, np.uint64).reshape(-1, 8))
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    ctx.set_chain_form(form)
    try:
        tails = ctx.queue_push_chain_full_batch(np.concatenate(rows), offsets)
    finally:
        ctx.set_chain_form(0)
    return [tuple(int(x) for x in tails[int(offsets[k + 1]) - 1][:4]) for k in range(len(items))]


@pytest.mark.parametrize("form", FORMS)
def test_device_fri_leaf_kat(ctx, form):
    kat = json.load(open(os.path.join(GOLD, "fri_leaf_kat_ram.json")))
    assert set(_chains(ctx, form, kat["leaves"])) == {tuple(c) for c in kat["cap"]}


@pytest.mark.parametrize("form", FORMS)
def test_device_exact_sibling_pair_kat(ctx, form):
    kat = json.load(open(os.path.join(GOLD, "merkle_pair_kat_ram.json")))
    cap = {tuple(c) for c in kat["cap"]}
    parents = {tuple(p) for p in kat["parents"]}
    ab = _chains(ctx, form, [e["pair"][0] + e["pair"][1] for e in kat["pairs"]])
    ba = _chains(ctx, form, [e["pair"][1] + e["pair"][0] for e in kat["pairs"]])
    seen = 0
    for e, y0, y1 in zip(kat["pairs"], ab, ba):
        u = e["uncle"]
        tops = _chains(ctx, form, [list(y0) + u, u + list(y0), list(y1) + u, u + list(y1)])
        good = [y for y, t in ((y0, tops[0]), (y0, tops[1]), (y1, tops[2]), (y1, tops[3])) if t in cap]
        assert len(good) == 1
        seen += good[0] in parents
    assert seen == 20


@pytest.mark.parametrize("form", FORMS)
def test_device_whole_query_paths(ctx, form):
    """leaf sponges of up to 160 elements (20 pushes, ragged last chunk) and 2..17 node levels up to the committed caps,
    incl. `setup_merkle_tree_cap` of setup/*/vk_N.json; all paths advance level by level in one launch per level"""
    kats = json.load(open(os.path.join(GOLD, "reference_merkle_paths_kat.json")))
    paths = []  # [leaf_elements, proof, idx, cap]
    for k in kats:
        for q in k["queries"]:
            for name, o in q["oracles"].items():
                paths.append([o["leaf_elements"], o["proof"], q["index"], k["caps"][name]])
            idx = q["index"]
            for lvl, o in enumerate(q["fri"]):
                idx >>= 3 if len(o["leaf_elements"]) == 16 else 2
                paths.append([o["leaf_elements"], o["proof"], idx, k["fri_caps"][lvl]])
    cur = _chains(ctx, form, [p[0] for p in paths])
    for lvl in range(max(len(p[1]) for p in paths)):
        live = [i for i, p in enumerate(paths) if lvl < len(p[1])]
        items = []
        for i in live:
            sib, c = paths[i][1][lvl], list(cur[i])
            items.append(sib + c if (paths[i][2] >> lvl) & 1 else c + sib)
        for i, d in zip(live, _chains(ctx, form, items)):
            cur[i] = d
    for p, c in zip(paths, cur):
        assert list(c) == p[3][p[2] >> len(p[1])]
    assert len(paths) == 68


def test_device_commit_kernel_on_reference_leaves(ctx):
    """the per-lane form as the commitment kernels use it (`zkw_commit_encodings`): for an 8-element item the
    length-specialised sponge is one permutation of (item || 000 || 8); check it against the chain form on the fixture
    leaves with the same capacity word — ties the fourth device user of p2::permute to the pinned forms"""
    kat = json.load(open(os.path.join(GOLD, "fri_leaf_kat_ram.json")))
    leaves = np.array(kat["leaves"], np.uint64)
    got = ctx.commit_variable_length_encodable_items(leaves)
    tin = np.zeros((len(leaves), 12), np.uint64)
    tin[:, 11] = 8
    tails = ctx.queue_push_chain_full_batch(leaves, np.arange(len(leaves) + 1, dtype=np.uint64), tin)
    assert np.array_equal(got, tails[:, :4])
