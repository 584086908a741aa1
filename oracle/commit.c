/* TEST INFRASTRUCTURE — CPU restatement, never linked into the product (see oracle.h).
 *
 * The setup side as field elements: NTT over Goldilocks, low-degree extension onto cosets, Poseidon2 Merkle tree with a cap — the shape of
 * create_base_layer_setup_data (src/prover_utils.rs:48-197: SetupStorage in monomial form and as an LDE, MerkleTreeWithCap over it; the bodies
 * are boojum's, absent). The hashing conventions are the ones every Merkle path of the reference's committed proofs pins
 * (orc_poseidon2_hash_leaf / _hash_node, tests/golden/reference_merkle_paths_kat.json); the enumeration of the LDE points is this
 * library's (leaf c * n + i = 7 * w_(lde n)^c * w_n^i): PARITY UNPINNED at that level. Textbook forms, written for obviousness: an
 * iterative radix-2 transform with an explicit bit reversal, Horner evaluation for the spot checks. */
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

#define P 0xFFFFFFFF00000001ULL

uint64_t orc_root_of_unity(uint32_t log_n) { return orc_gl_pow(orc_gl_pow(7, (P - 1) >> 32), 1ULL << (32 - log_n)); }

void orc_gl_powers(uint64_t base, size_t n, uint64_t *out) {
    uint64_t x = 1;
    for (size_t i = 0; i < n; i++) { out[i] = x; x = orc_gl_mul(x, base % P); }
}

/* in place, natural order in and out: X[k] = sum_j x[j] w^(jk), w = w_n (inverse: w^-1, then 1/n) */
void orc_ntt(uint64_t *x, uint32_t log_n, int inverse) {
    const size_t n = (size_t)1 << log_n;
    for (size_t i = 0; i < n; i++) { /* bit reversal */
        size_t r = 0;
        for (uint32_t b = 0; b < log_n; b++) r |= ((i >> b) & 1) << (log_n - 1 - b);
        if (r > i) { uint64_t t = x[i]; x[i] = x[r]; x[r] = t; }
    }
    uint64_t w = orc_root_of_unity(log_n);
    if (inverse) w = orc_gl_pow(w, P - 2);
    for (uint32_t s = 1; s <= log_n; s++) { /* decimation in time */
        const size_t m = (size_t)1 << s, half = m / 2;
        const uint64_t wm = orc_gl_pow(w, n / m);
        for (size_t k = 0; k < n; k += m) {
            uint64_t t = 1;
            for (size_t j = 0; j < half; j++) {
                const uint64_t u = x[k + j] % P, v = orc_gl_mul(x[k + j + half] % P, t);
                x[k + j] = orc_gl_add(u, v);
                x[k + j + half] = orc_gl_sub(u, v);
                t = orc_gl_mul(t, wm);
            }
        }
    }
    if (inverse) {
        const uint64_t ninv = orc_gl_pow((uint64_t)n % P, P - 2);
        for (size_t i = 0; i < n; i++) x[i] = orc_gl_mul(x[i], ninv);
    }
}

/* values [n_cols][n] on the domain -> out [lde][n_cols][n]: coset c = the points 7 * w_(lde n)^c * w_n^i */
void orc_lde(const uint64_t *values, uint32_t log_n, size_t n_cols, uint32_t lde_factor, uint64_t *out) {
    const size_t n = (size_t)1 << log_n;
    uint32_t log_lde = 0;
    while ((1u << log_lde) < lde_factor) log_lde++;
    const uint64_t gamma = orc_root_of_unity(log_n + log_lde);
    uint64_t *c = (uint64_t *)malloc(n * 8);
    for (size_t col = 0; col < n_cols; col++) {
        memcpy(c, values + col * n, n * 8);
        orc_ntt(c, log_n, 1);
        for (uint32_t k = 0; k < lde_factor; k++) {
            uint64_t *o = out + ((size_t)k * n_cols + col) * n;
            const uint64_t shift = orc_gl_mul(7, orc_gl_pow(gamma, k));
            uint64_t s = 1;
            for (size_t j = 0; j < n; j++) { o[j] = orc_gl_mul(c[j], s); s = orc_gl_mul(s, shift); }
            orc_ntt(o, log_n, 0);
        }
    }
    free(c);
}

/* the polynomial through `values` on the domain, evaluated at x (O(n) inverse transform once per call is the caller's: coeffs in) */
uint64_t orc_poly_eval(const uint64_t *coeffs, size_t n, uint64_t x) {
    uint64_t acc = 0;
    for (size_t i = n; i-- > 0;) acc = orc_gl_add(orc_gl_mul(acc, x % P), coeffs[i] % P);
    return acc;
}

/* leaf_cols [n_sets][n_cols][n] -> tree: every level, leaves first (4 * (2 * n_sets * n - cap_size) words); the cap is the last level */
void orc_merkle_tree_with_cap(const uint64_t *leaf_cols, size_t n_sets, size_t n_cols, size_t n, uint32_t cap_size, uint64_t *tree) {
    const size_t n_leaves = n_sets * n;
    uint64_t *row = (uint64_t *)malloc(n_cols * 8);
    for (size_t s = 0; s < n_sets; s++)
        for (size_t i = 0; i < n; i++) {
            for (size_t c = 0; c < n_cols; c++) row[c] = leaf_cols[(s * n_cols + c) * n + i];
            orc_poseidon2_hash_leaf(row, n_cols, tree + 4 * (s * n + i));
        }
    free(row);
    uint64_t *below = tree;
    for (size_t m = n_leaves / 2; m >= cap_size && m >= 1; m /= 2) {
        uint64_t *level = below + 8 * m;
        for (size_t i = 0; i < m; i++) orc_poseidon2_hash_node(below + 8 * i, below + 8 * i + 4, level + 4 * i);
        below = level;
        if (m == 1) break;
    }
}
