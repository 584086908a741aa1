"""GPU parity: the setup side as field elements (csrc/ntt_kernels.cuh, csrc/zkw_commit.hip) against the oracle (oracle/commit.c) —
NTT at every size the kernels split differently, LDE against the oracle and against Horner evaluation at the full size, the Merkle tree
cell for cell and through the path rule the reference's proofs pin, and the setup commitment of a real layout end to end."""
import numpy as np
import pytest

from era_zkevm_test_harness_amd import native

pytestmark = pytest.mark.gpu
P = 0xFFFFFFFF00000001


@pytest.fixture(scope="module")
def ctx():
    c = native.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("log_n", [0, 1, 2, 5, 10, 13, 14, 15, 17, 20])
def test_ntt_matches_oracle(ctx, oracle, log_n):
    """one-pass kernel up to 2^13, the four-step split above (14 = 7 + 7, 15 = 7 + 8, 17 = 8 + 9, 20 = 10 + 10); forward and inverse"""
    rng = np.random.default_rng(log_n)
    n_cols = 3 if log_n < 20 else 2
    x = rng.integers(0, P, (n_cols, 1 << log_n), dtype=np.uint64)
    x[0, : min(4, 1 << log_n)] = [P - 1, 0, 1, P - 2][: min(4, 1 << log_n)]
    f = ctx.ntt(x)
    assert int(f.max()) < P
    assert np.array_equal(f, oracle.ntt(x))
    assert np.array_equal(ctx.ntt(f, inverse=True), x)
    assert np.array_equal(ctx.ntt(x, inverse=True), oracle.ntt(x, inverse=True))


def test_ntt_accepts_non_canonical_input(ctx, oracle):
    x = np.array([[P, P + 5, 2**64 - 1, 7] * 4], np.uint64)
    assert np.array_equal(ctx.ntt(x), oracle.ntt(x % np.uint64(P)))


@pytest.mark.parametrize("log_n,lde", [(6, 2), (12, 4), (14, 2), (16, 8)])
def test_lde_matches_oracle(ctx, oracle, log_n, lde):
    rng = np.random.default_rng(log_n)
    vals = rng.integers(0, P, (5, 1 << log_n), dtype=np.uint64)
    assert np.array_equal(ctx.lde(vals, lde), oracle.lde(vals, lde))


def test_lde_full_size_is_evaluation_on_the_cosets(ctx, oracle):
    """2^20 points, factor 2 (the reference's fri_lde_factor): spot checks by Horner evaluation of the interpolating polynomial"""
    rng = np.random.default_rng(20)
    log_n = 20
    vals = rng.integers(0, P, (2, 1 << log_n), dtype=np.uint64)
    ext = ctx.lde(vals, 2)
    coeffs = ctx.ntt(vals, inverse=True)
    w, gamma = oracle.root_of_unity(log_n), oracle.root_of_unity(log_n + 1)
    for col in range(2):
        assert oracle.poly_eval(coeffs[col], pow(w, 123457, P)) == int(vals[col, 123457])
        for c in range(2):
            for i in (0, 1, 1 << 19, (1 << 20) - 1, 777777):
                assert oracle.poly_eval(coeffs[col], 7 * pow(gamma, c, P) * pow(w, i, P) % P) == int(ext[c, col, i])


@pytest.mark.parametrize("n_sets,n_cols,n,cap", [(1, 1, 16, 16), (2, 11, 32, 4), (2, 8, 64, 1), (4, 150, 256, 16)])
def test_merkle_tree_matches_oracle(ctx, oracle, n_sets, n_cols, n, cap):
    rng = np.random.default_rng(n_cols)
    cols = rng.integers(0, P, (n_sets, n_cols, n), dtype=np.uint64)
    got_cap, tree = ctx.merkle_tree_with_cap(cols, cap, want_tree=True)
    want = oracle.merkle_tree_with_cap(cols, cap)
    assert np.array_equal(tree, want) and np.array_equal(got_cap, want[-cap:])
    assert np.array_equal(ctx.merkle_tree_with_cap(cols, cap), got_cap)


def test_setup_commit_of_a_layout(ctx, oracle):
    """RAMPermutation at capacity 300 in 2^12 rows: the columns are sigma as field elements, the selector column, the lookup-table columns; the cap of the device
    pipeline == the oracle's pipeline on the same columns; a leaf of the tree opens to that cap; a different capacity commits differently"""
    ctype, cap, log_n = 8, 300, 12
    n = 1 << log_n
    cols = ctx.setup_columns(ctype, cap, log_n)
    sigma = native.setup_copy_permutation(ctype, cap, n)
    sel = native.setup_row_selectors(ctype, cap, n)
    tab = native.setup_lookup_tables(ctype, n)
    G = sigma.shape[0]
    assert cols.shape == (G + 1 + tab.shape[0], n) and np.array_equal(cols[G], sel.astype(np.uint64)) and np.array_equal(cols[G + 1:], tab)
    w = oracle.root_of_unity(log_n)
    om = oracle.gl_powers(w, n)
    for c, r in ((0, 0), (5, 17), (130, 2000), (132, n - 1)):
        t = int(sigma[c, r])
        assert int(cols[c, r]) == pow(7, t >> log_n, P) * int(om[t & (n - 1)]) % P
    ident = sigma == np.arange(sigma.size, dtype=np.uint64).reshape(sigma.shape)
    assert 0 < ident.sum() < sigma.size  # some cells are under a copy constraint, some are not
    commit = ctx.setup_commit(ctype, cap, log_n)
    want = oracle.merkle_tree_with_cap(oracle.lde(cols, 2), 16)
    assert np.array_equal(commit, want[-16:])
    assert not np.array_equal(ctx.setup_commit(ctype, cap - 1, log_n), commit)


def test_setup_commit_at_production_size(ctx, oracle):
    """EventsSorter at its production capacity, 2^20 rows, LDE x 2, cap 16 (133 columns): the device cap equals a tree built from the device's
    LDE columns, and sampled leaves open to it by the reference's path rule with the oracle's hashes"""
    ctype, log_n = 11, 20
    n = 1 << log_n
    commit = ctx.setup_commit(ctype, 0, log_n)
    assert commit.shape == (16, 4) and int(commit.max()) < P and len({tuple(x) for x in commit}) == 16
    cols = ctx.setup_columns(ctype, 0, log_n)
    ext = ctx.lde(cols, 2)
    cap, tree = ctx.merkle_tree_with_cap(ext, 16, want_tree=True)
    assert np.array_equal(cap, commit)
    for leaf in (0, 12345, n + 7, 2 * n - 1):
        s, i = divmod(leaf, n)
        cur, idx, off, width = oracle.hash_leaf(ext[s, :, i]), leaf, 0, 2 * n
        assert np.array_equal(cur, tree[leaf])
        while width > 16:
            sib = tree[off + (idx ^ 1)]
            cur = oracle.hash_node(sib, cur) if idx & 1 else oracle.hash_node(cur, sib)
            off += width
            width //= 2
            idx >>= 1
        assert np.array_equal(cur, commit[idx])
    # spot-check the extension of the last column (the table ids) against Horner evaluation
    coeffs = ctx.ntt(cols[-1:], inverse=True)[0]
    gamma, w = oracle.root_of_unity(log_n + 1), oracle.root_of_unity(log_n)
    for c, i in ((0, 3), (1, 99999)):
        assert oracle.poly_eval(coeffs, 7 * pow(gamma, c, P) * pow(w, i, P) % P) == int(ext[c, -1, i])


def test_setup_commit_of_every_synthesized_type(ctx, oracle):
    """all twelve layouts commit (2^18 rows, reduced capacities): twelve different caps; ECRecover's (the newest copy classes: its EC section)
    equals the oracle's pipeline over the device's columns, and its columns are sigma as field elements + the selector column"""
    log_n = 18
    caps = {2: 1000, 3: 300, 4: 1000, 5: 50, 6: 300, 7: 2, 8: 1000, 9: 1000, 10: 4, 11: 1000, 12: 1000, 13: 100}
    commits = {t: ctx.setup_commit(t, c, log_n) for t, c in caps.items()}
    assert all(v.shape == (16, 4) and int(v.max()) < P for v in commits.values())
    distinct = {tuple(v.reshape(-1)) for t, v in commits.items() if t != 12}   # (11 and 12 share one layout)
    assert len(distinct) == 11 and np.array_equal(commits[11], commits[12])
    cols = ctx.setup_columns(7, 2, log_n)
    sigma = native.setup_copy_permutation(7, 2, 1 << log_n)
    assert cols.shape[0] == sigma.shape[0] + 1 + 4 and np.array_equal(cols[sigma.shape[0] + 1:], native.setup_lookup_tables(7, 1 << log_n))
    om = oracle.gl_powers(oracle.root_of_unity(log_n), 1 << log_n)
    rng = np.random.default_rng(7)
    for cell in rng.integers(0, sigma.size, 200):
        c, r = divmod(int(cell), 1 << log_n)
        t = int(sigma[c, r])
        assert int(cols[c, r]) == pow(7, t >> log_n, P) * int(om[t & ((1 << log_n) - 1)]) % P
    want = oracle.merkle_tree_with_cap(oracle.lde(cols, 2), 16)
    assert np.array_equal(commits[7], want[-16:])


def test_verification_key_of_a_layout(ctx, tmp_path):
    """the key of the EventsSorter layout in the reference's vk_N.json shape: the cap is zkw_setup_commit's at 2^20 rows / LDE x 2 / cap 16, the
    file round-trips through the reference's enum-keyed JSON, and the geometry fields repeat the wrapper's"""
    from era_zkevm_test_harness_amd import wire

    vk = wire.verification_key_of_layout(ctx, 11)
    assert np.array_equal(np.array(vk["setup_merkle_tree_cap"], np.uint64), ctx.setup_commit(11, 0, 20))
    fp = vk["fixed_parameters"]
    assert fp["domain_size"] == 1 << 20 and fp["fri_lde_factor"] == 2 and fp["cap_size"] == 16 and fp["total_tables_len"] == 256
    path = tmp_path / "vk_11.json"
    wire.dump(path, 11, vk)
    assert wire.load(path) == (11, vk)
