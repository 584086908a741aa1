"""Oracle level: the setup side as field elements (oracle/commit.c) — the transform against the definition, the extension against
Horner evaluation, the tree against the pinned leaf / node hashes. Textbook properties, so that the GPU tests (tests/test_gpu_commit.py)
compare against something that is itself checked."""
import numpy as np

P = 0xFFFFFFFF00000001


def test_root_of_unity(oracle):
    assert oracle.root_of_unity(32) == 0x185629DCDA58878C  # boojum's / plonky2's 2^32-th root: 7^((p - 1) / 2^32)
    for k in (1, 5, 20):
        w = oracle.root_of_unity(k)
        assert pow(w, 1 << k, P) == 1 and pow(w, 1 << (k - 1), P) == P - 1


def test_ntt_is_the_dft(oracle):
    rng = np.random.default_rng(1)
    for log_n in (0, 1, 3, 5):
        n = 1 << log_n
        x = rng.integers(0, P, n, dtype=np.uint64)
        w = oracle.root_of_unity(log_n) if log_n else 1
        want = [sum(int(x[j]) * pow(w, j * k, P) for j in range(n)) % P for k in range(n)]
        assert [int(v) for v in oracle.ntt(x)] == want
        assert np.array_equal(oracle.ntt(oracle.ntt(x), inverse=True), x)


def test_ntt_round_trip_and_linearity(oracle):
    rng = np.random.default_rng(2)
    a, b = rng.integers(0, P, (2, 1 << 12), dtype=np.uint64)
    assert np.array_equal(oracle.ntt(oracle.ntt(a), inverse=True), a)
    s = np.array([(int(x) + int(y)) % P for x, y in zip(a, b)], np.uint64)
    fa, fb, fs = oracle.ntt(a), oracle.ntt(b), oracle.ntt(s)
    assert all((int(x) + int(y)) % P == int(z) for x, y, z in zip(fa, fb, fs))


def test_lde_is_evaluation_on_the_cosets(oracle):
    rng = np.random.default_rng(3)
    log_n, n = 8, 256
    vals = rng.integers(0, P, (3, n), dtype=np.uint64)
    ext = oracle.lde(vals, 4)
    coeffs = oracle.ntt(vals, inverse=True)
    w, gamma = oracle.root_of_unity(log_n), oracle.root_of_unity(log_n + 2)
    for col in range(3):
        assert all(oracle.poly_eval(coeffs[col], pow(w, i, P)) == int(vals[col, i]) for i in (0, 1, 77, 255))  # the polynomial interpolates
        for c in range(4):
            for i in (0, 5, 200):
                x = 7 * pow(gamma, c, P) * pow(w, i, P) % P
                assert oracle.poly_eval(coeffs[col], x) == int(ext[c, col, i])


def test_merkle_tree_with_cap(oracle):
    rng = np.random.default_rng(4)
    cols = rng.integers(0, P, (2, 11, 32), dtype=np.uint64)  # 2 cosets x 11 columns x 32 positions: 64 leaves of 11 elements (ragged last chunk)
    tree = oracle.merkle_tree_with_cap(cols, 4)
    assert tree.shape == (2 * 64 - 4, 4)
    for leaf in (0, 31, 32, 63):
        s, i = divmod(leaf, 32)
        assert np.array_equal(tree[leaf], oracle.hash_leaf(cols[s, :, i]))
    # a path from a leaf to the cap, by the rule the reference's proofs pin (tests/test_gpu_reference_kats.py): bit l of the index picks the side
    for leaf in (3, 40, 63):
        cur, idx, off, width = tree[leaf], leaf, 0, 64
        while width > 4:
            sib = tree[off + (idx ^ 1)]
            cur = oracle.hash_node(sib, cur) if idx & 1 else oracle.hash_node(cur, sib)
            off += width
            width //= 2
            idx >>= 1
            assert np.array_equal(cur, tree[off + idx])
        assert np.array_equal(cur, tree[-4 + idx])
