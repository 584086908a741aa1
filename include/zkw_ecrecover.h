/* zkw_ecrecover.h — the EC SECTION of the ECRecover circuit (type 7): record types of the generated spec
 * (include/zkw_ecrecover_ec_spec.h, format and statement: tools/gen_ecrecover_circuit.py), secp256k1 / Goldilocks helpers, the
 * evaluation of a cycle's value tape, and the relation every item states over its cells. Plain C that also compiles as HIP device
 * code; shared by the kernels (csrc/ecrecover_kernels.cuh) and the host side. NOT by the test oracle: oracle/ecrecover_eval.c evaluates
 * and checks the items with code of its own and includes only the format header include/zkw_ecrecover_layout.h. A third restatement is
 * the Python evaluator of the generator (tests compare tapes); the reference's circuit body (ecrecover_function_entry_point, era-zkevm_circuits) is absent
 * from /root/reference, so the placement is this library's own — geometry and tables are base_layer/ecrecover.rs:30-41,138-176.
 *
 * Rows of cycle c: EC_FIRST_ROW + c * EC_ROWS_PER_CYCLE + run.row0 + instance * type.n_rows + row; a row = 80 general-purpose
 * cells + 16 lookup slots of width 3 of ONE table. Items (32-bit words, w0 = kind | row << 4 | col << 16 | aux << 24):
 *   LIN    aux = known cells; n_new; const lo, hi; known x {ref, coef (i32)}; new x {tape index, shift | width << 8}
 *   SEL    refs b, x, y; out tape index                      cells [b, x, y, o]
 *   FMA    aux = 1: d is NEW; refs a, b, c; d (tape index / ref)  cells [a, b, c, d]
 *   MUL    aux = modulus (0 P, 1 N); refs a0, b0, r0 (limb i = ref + i); q tape0, carry tape0   cells a 0.., b 16.., q 32.., r 48.., c 64..78
 *   HINT   aux = kind; arguments (no cells)
 *   LOOKUP row, col = slot; aux = inputs; table (| EC_ROWTAB_PER_INSTANCE); in0, in1; out tape0   cells [in.., out..] at 80 + 3 slot
 * References: kind << 28 | payload — 0 TAPE t, 1 PREV k (state element k of the previous segment), 2 GLOB k, 3 GLOBJ (base | stride << 16:
 * global base + stride * instance), 4 CONST v, 5 BIG (idx << 4 | limb), 6 IN k (input byte), 0xFFFFFFFF none. */
#ifndef ZKW_ECRECOVER_H
#define ZKW_ECRECOVER_H
#include "zkw_ecrecover_layout.h"
/* loops over the words of a 256-bit value are unrolled, so that such a value lives in registers on the GPU (a loop the compiler keeps
   rolled indexes the value at run time, which puts it in scratch memory: tests/test_kernel_resources.py) */
#if defined(__HIPCC__) || defined(__clang__)
#define EC_UNROLL _Pragma("unroll")
#else
#define EC_UNROLL _Pragma("GCC unroll 16")
#endif

/* ---- Goldilocks (canonical inputs and outputs) ------------------------------------------------------------------------ */
EC_HD uint64_t ec_gl_add(uint64_t a, uint64_t b) {
    uint64_t s = a + b;
    if (s < a || s >= EC_GL_P) s -= EC_GL_P;
    return s;
}
EC_HD uint64_t ec_gl_sub(uint64_t a, uint64_t b) { return a >= b ? a - b : a + (EC_GL_P - b); }
EC_HD uint64_t ec_gl_mul(uint64_t a, uint64_t b) {
    const uint64_t a0 = (uint32_t)a, a1 = a >> 32, b0 = (uint32_t)b, b1 = b >> 32;
    const uint64_t p00 = a0 * b0, p01 = a0 * b1, p10 = a1 * b0, p11 = a1 * b1;
    const uint64_t mid = (p00 >> 32) + (uint32_t)p01 + (uint32_t)p10;
    const uint64_t lo = (uint32_t)p00 | (mid << 32);
    const uint64_t hi = p11 + (p01 >> 32) + (p10 >> 32) + (mid >> 32);
    /* x = hi_hi 2^96 + hi_lo 2^64 + lo, 2^64 = 2^32 - 1, 2^96 = -1 */
    const uint64_t hi_hi = hi >> 32, hi_lo = (uint32_t)hi;
    uint64_t t = lo - hi_hi;
    if (lo < hi_hi) t -= 0xFFFFFFFFull; /* + p */
    const uint64_t u = hi_lo * 0xFFFFFFFFull;
    uint64_t r = t + u;
    if (r < t) r += 0xFFFFFFFFull;
    if (r >= EC_GL_P) r -= EC_GL_P;
    return r;
}
EC_HD uint64_t ec_gl_inv(uint64_t a) { /* a^(p - 2); 0 -> 0 */
    uint64_t r = 1, b = a;
    uint64_t e = EC_GL_P - 2;
    while (e) {
        if (e & 1) r = ec_gl_mul(r, b);
        b = ec_gl_mul(b, b);
        e >>= 1;
    }
    return r;
}
EC_HD uint64_t ec_gl_from_i64(int64_t v) { return v >= 0 ? (uint64_t)v % EC_GL_P : EC_GL_P - ((uint64_t)(-v) % EC_GL_P); }

/* ---- 256-bit arithmetic modulo m = 2^256 - c (the secp256k1 base and scalar fields), 32-bit words little end first ------
   Every array with run-time indices lives in a caller-provided WORKSPACE (`ec_ws`: the stack on a CPU, a slice of LDS in the
   kernels — per-lane arrays with run-time indices would otherwise become scratch memory, which every HSA queue that ran the kernel
   keeps for every wave slot of the chip: tests/test_kernel_resources.py); 256-bit values (`ec_u256`) are indexed by constants only. */
typedef struct ec_u256 { uint32_t w[8]; } ec_u256;
typedef struct ec_mod { const uint32_t *m, *c, *inv_e; uint32_t nc; } ec_mod;
typedef struct ec_ws {
    uint32_t va[16], vb[16], vc[16]; /* the limb vectors of a MUL row / of a hint's operand: lazy limbs, < 2^32 (ec_get_vec) */
    uint32_t A[9], B[9];             /* two of them as integers */
    uint32_t big[52];                /* ec_mul_witness: T[20] | Q[12] | Qc[20]; ec_reduce: cur[24] | nxt[24] */
    uint32_t pad;                    /* 119 words = 476 bytes: an odd word stride between the lanes' slices of LDS, and five 64-lane
                                        workgroups of k_ec_segments on a CU (30.5 KB each) instead of two (the struct was 1 044 bytes) */
} ec_ws;

static const uint32_t EC_P_M[8] = {0xFFFFFC2Fu, 0xFFFFFFFEu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
static const uint32_t EC_P_C[5] = {977u, 1u, 0u, 0u, 0u};
static const uint32_t EC_P_INV_E[8] = {0xFFFFFC2Du, 0xFFFFFFFEu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}; /* p - 2 */
static const uint32_t EC_P_SQRT_E[8] = {0xBFFFFF0Cu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0x3FFFFFFFu}; /* (p + 1) / 4 */
static const uint32_t EC_N_M[8] = {0xD0364141u, 0xBFD25E8Cu, 0xAF48A03Bu, 0xBAAEDCE6u, 0xFFFFFFFEu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
static const uint32_t EC_N_C[5] = {0x2FC9BEBFu, 0x402DA173u, 0x50B75FC4u, 0x45512319u, 0x1u};
static const uint32_t EC_N_INV_E[8] = {0xD036413Fu, 0xBFD25E8Cu, 0xAF48A03Bu, 0xBAAEDCE6u, 0xFFFFFFFEu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}; /* n - 2 */

EC_HD ec_mod ec_modulus(uint32_t which) { /* 0: P = 2^256 - 2^32 - 977, 1: N */
    ec_mod M;
    if (which == 0) { M.m = EC_P_M; M.c = EC_P_C; M.inv_e = EC_P_INV_E; M.nc = 2; }
    else { M.m = EC_N_M; M.c = EC_N_C; M.inv_e = EC_N_INV_E; M.nc = 5; }
    return M;
}
EC_HD int ec_cmp8(const uint32_t *a, const uint32_t *b) { /* branch-free: the borrow of a - b and whether any word differs */
    uint64_t br = 0;
    uint32_t diff = 0;
    EC_UNROLL for (int i = 0; i < 8; i++) {
        const uint64_t d = (uint64_t)a[i] - b[i] - br;
        br = (d >> 32) & 1;
        diff |= a[i] ^ b[i];
    }
    return br ? -1 : diff ? 1 : 0;
}
EC_HD void ec_sub8(uint32_t *a, const uint32_t *b) { /* a -= b */
    uint64_t br = 0;
    EC_UNROLL for (int i = 0; i < 8; i++) {
        const uint64_t d = (uint64_t)a[i] - b[i] - br;
        a[i] = (uint32_t)d;
        br = (d >> 32) & 1;
    }
}
/* out[na + nb] = a * b */
EC_HD void ec_mul_words(const uint32_t *a, int na, const uint32_t *b, int nb, uint32_t *out) {
    for (int i = 0; i < na + nb; i++) out[i] = 0;
    for (int i = 0; i < na; i++) {
        uint64_t carry = 0;
        const uint64_t ai = a[i];
        for (int j = 0; j < nb; j++) {
            const uint64_t t = ai * b[j] + out[i + j] + carry;
            out[i + j] = (uint32_t)t;
            carry = t >> 32;
        }
        out[i + nb] = (uint32_t)carry;
    }
}
/* x (n <= 20 words, not inside W->big) mod m */
EC_HD ec_u256 ec_reduce(const uint32_t *x, int n, const ec_mod *M, ec_ws *W) {
    uint32_t *cur = W->big, *nxt = W->big + 24;
    int len = n;
    for (int i = 0; i < 24; i++) cur[i] = i < n ? x[i] : 0;
    while (len > 8) { /* x = hi 2^256 + lo = hi c + lo */
        int top = len;
        while (top > 8 && cur[top - 1] == 0) top--;
        if (top <= 8) break;
        const int nh = top - 8;
        ec_mul_words(cur + 8, nh, M->c, (int)M->nc, nxt);
        int nl = nh + (int)M->nc;
        if (nl < 9) { for (int i = nl; i < 9; i++) nxt[i] = 0; nl = 9; }
        else nxt[nl++] = 0;
        uint64_t carry = 0;
        for (int i = 0; i < nl; i++) {
            const uint64_t t = (uint64_t)nxt[i] + (i < 8 ? cur[i] : 0) + carry;
            nxt[i] = (uint32_t)t;
            carry = t >> 32;
        }
        for (int i = 0; i < 24; i++) cur[i] = i < nl ? nxt[i] : 0;
        len = nl;
    }
    ec_u256 r;
    r.w[0] = cur[0]; r.w[1] = cur[1]; r.w[2] = cur[2]; r.w[3] = cur[3]; r.w[4] = cur[4]; r.w[5] = cur[5]; r.w[6] = cur[6]; r.w[7] = cur[7];
    while (ec_cmp8(r.w, M->m) >= 0) ec_sub8(r.w, M->m);
    return r;
}
/* ---- the multiplication the inversions and the curve arithmetic spend their time in: constant shapes only, so that on the GPU every
   word lives in a register (the generic ec_reduce above walks arrays of run-time length in the workspace).
   x = hi 2^256 + lo = hi c + lo, folded a fixed number of times:  P (c = 2^32 + 977): 16 -> 11 -> 9 -> 9 words;
   N (c of 129 bits): 16 -> 14 -> 10 -> 9 -> 9 words; then at most two subtractions of m. */
/* acc[0 .. NA) += h[0 .. NH) * c[0 .. NC) (NA >= NH + NC; no carry out of acc by the callers' bounds) */
#define EC_FOLD_MAC(acc, NA, h, NH, c, NC)                                        \
    EC_UNROLL for (int i_ = 0; i_ < (NH); i_++) {                                 \
        uint64_t cy_ = 0;                                                         \
        const uint64_t hi_ = (h)[i_];                                             \
        EC_UNROLL for (int j_ = 0; j_ < (NC); j_++) {                             \
            const uint64_t v_ = hi_ * (c)[j_] + (acc)[i_ + j_] + cy_;             \
            (acc)[i_ + j_] = (uint32_t)v_;                                        \
            cy_ = v_ >> 32;                                                       \
        }                                                                         \
        EC_UNROLL for (int k_ = i_ + (NC); k_ < (NA); k_++) {                     \
            const uint64_t v_ = (uint64_t)(acc)[k_] + cy_;                        \
            (acc)[k_] = (uint32_t)v_;                                             \
            cy_ = v_ >> 32;                                                       \
        }                                                                         \
    }
EC_HD ec_u256 ec_finish8(const uint32_t *x9, const uint32_t *m) { /* x9 < 2^256 + small: the last fold's 9 words, word 8 is 0 or 1 */
    ec_u256 r;
    EC_UNROLL for (int i = 0; i < 8; i++) r.w[i] = x9[i];
    uint32_t top = x9[8];
    for (int rep = 0; rep < 3; rep++) {
        if (!top && ec_cmp8(r.w, m) < 0) break;
        uint64_t br = 0;
        EC_UNROLL for (int i = 0; i < 8; i++) {
            const uint64_t d = (uint64_t)r.w[i] - m[i] - br;
            r.w[i] = (uint32_t)d;
            br = (d >> 32) & 1;
        }
        top -= (uint32_t)br;
    }
    return r;
}
EC_HD ec_u256 ec_reduce16_p(const uint32_t *t) {
    const uint32_t c[2] = {977u, 1u};
    uint32_t a1[11], a2[9], a3[9];
    EC_UNROLL for (int i = 0; i < 11; i++) a1[i] = i < 8 ? t[i] : 0;
    EC_FOLD_MAC(a1, 11, t + 8, 8, c, 2)
    EC_UNROLL for (int i = 0; i < 9; i++) a2[i] = i < 8 ? a1[i] : 0;
    EC_FOLD_MAC(a2, 9, a1 + 8, 3, c, 2)
    EC_UNROLL for (int i = 0; i < 9; i++) a3[i] = i < 8 ? a2[i] : 0;
    EC_FOLD_MAC(a3, 9, a2 + 8, 1, c, 2)
    return ec_finish8(a3, EC_P_M);
}
EC_HD ec_u256 ec_reduce16_n(const uint32_t *t) {
    const uint32_t c[5] = {0x2FC9BEBFu, 0x402DA173u, 0x50B75FC4u, 0x45512319u, 0x1u};
    uint32_t a1[14], a2[11], a3[9], a4[9];
    EC_UNROLL for (int i = 0; i < 14; i++) a1[i] = i < 8 ? t[i] : 0;
    EC_FOLD_MAC(a1, 14, t + 8, 8, c, 5)   /* < 2^385: 13 words, word 13 stays 0 */
    EC_UNROLL for (int i = 0; i < 11; i++) a2[i] = i < 8 ? a1[i] : 0;
    EC_FOLD_MAC(a2, 11, a1 + 8, 5, c, 5)  /* hi < 2^129 (5 words; word 13 of a1 is 0): < 2^259 */
    EC_UNROLL for (int i = 0; i < 9; i++) a3[i] = i < 8 ? a2[i] : 0;
    EC_FOLD_MAC(a3, 9, a2 + 8, 1, c, 5)   /* hi < 8 */
    EC_UNROLL for (int i = 0; i < 9; i++) a4[i] = i < 8 ? a3[i] : 0;
    EC_FOLD_MAC(a4, 9, a3 + 8, 1, c, 5)   /* hi <= 1 */
    return ec_finish8(a4, EC_N_M);
}
EC_HD ec_u256 ec_mulmod(const ec_u256 *a, const ec_u256 *b, const ec_mod *M, ec_ws *W) {
    (void)W;
    uint32_t t[16];
    EC_UNROLL for (int i = 0; i < 16; i++) t[i] = 0;
    EC_UNROLL for (int i = 0; i < 8; i++) {
        uint64_t carry = 0;
        const uint64_t ai = a->w[i];
        EC_UNROLL for (int j = 0; j < 8; j++) {
            const uint64_t v = ai * b->w[j] + t[i + j] + carry;
            t[i + j] = (uint32_t)v;
            carry = v >> 32;
        }
        t[i + 8] = (uint32_t)carry;
    }
    return M->nc == 2 ? ec_reduce16_p(t) : ec_reduce16_n(t);
}
EC_HD ec_u256 ec_submod(const ec_u256 *a, const ec_u256 *b, const ec_mod *M) { /* a, b < m */
    ec_u256 r = *a;
    if (ec_cmp8(a->w, b->w) >= 0) { ec_sub8(r.w, b->w); return r; }
    uint64_t carry = 0; /* a + m - b */
    EC_UNROLL for (int i = 0; i < 8; i++) {
        const uint64_t t = (uint64_t)a->w[i] + M->m[i] + carry;
        r.w[i] = (uint32_t)t;
        carry = t >> 32;
    }
    ec_sub8(r.w, b->w);
    return r;
}
EC_HD ec_u256 ec_addmod(const ec_u256 *a, const ec_u256 *b, const ec_mod *M, ec_ws *W) { /* a, b < m */
    (void)W;
    ec_u256 r;
    uint64_t carry = 0;
    EC_UNROLL for (int i = 0; i < 8; i++) {
        const uint64_t s = (uint64_t)a->w[i] + b->w[i] + carry;
        r.w[i] = (uint32_t)s;
        carry = s >> 32;
    }
    if (carry || ec_cmp8(r.w, M->m) >= 0) ec_sub8(r.w, M->m); /* (a borrow out of the top cancels the carry) */
    return r;
}
EC_HD int ec_is_zero8(const ec_u256 *a) {
    uint32_t o = 0;
    EC_UNROLL for (int i = 0; i < 8; i++) o |= a->w[i];
    return o == 0;
}
EC_HD ec_u256 ec_zero256(void) {
    ec_u256 z;
    EC_UNROLL for (int i = 0; i < 8; i++) z.w[i] = 0;
    return z;
}
/* a^e mod m, e = 8 words in constant memory (m - 2: the inverse; (p + 1) / 4: the square root) */
EC_HD ec_u256 ec_powmod(const ec_u256 *a, const uint32_t *e, const ec_mod *M, ec_ws *W) {
    ec_u256 r = *a;
    int started = 0;
    for (int wi = 7; wi >= 0; wi--) {
        const uint32_t word = e[wi];
        for (int bit = 31; bit >= 0; bit--) {
            if (started) r = ec_mulmod(&r, &r, M, W);
            if ((word >> bit) & 1) {
                if (started) r = ec_mulmod(&r, a, M, W);
                started = 1; /* (the first set bit: r = a already) */
            }
        }
    }
    return r;
}
/* a^-1 mod m (m odd, 0 < a < m, gcd(a, m) = 1) by the binary extended Euclidean algorithm on two pairs (u, x), (v, y) with u = x a,
   v = y a (mod m): the pair whose turn it is — the even one, else the larger — comes first; an even u is halved (and x with it, after
   adding m when x is odd), an odd u >= v loses v (and x loses y). About 1.4 x 512 steps of ~60 instructions against the 506 modular
   multiplications (~700 instructions each) of a^(m - 2): the 544 quotients lambda = dy / dx of a cycle were 85 % of the EC section's
   fill. Every 256-bit value stays in registers (constant indices only). */
EC_HD ec_u256 ec_invmod(const ec_u256 *a, const ec_mod *M, ec_ws *W) {
    (void)W;
    uint32_t u[8], v[8], x[8], y[8], m[8];
    EC_UNROLL for (int i = 0; i < 8; i++) { u[i] = a->w[i]; v[i] = m[i] = M->m[i]; x[i] = i == 0; y[i] = 0; }
    for (int step = 0; step < 1100; step++) {
        uint32_t ru = 0, rv = 0;
        EC_UNROLL for (int i = 1; i < 8; i++) { ru |= u[i]; rv |= v[i]; }
        if ((ru == 0 && u[0] == 1) || (rv == 0 && v[0] == 1)) break;
        /* whose turn: u if it is even; else v if it is even; else the larger of the two */
        int ge = 1; /* u >= v */
        {
            int lt = 0, gt = 0; /* from the top word down: the first difference decides */
            EC_UNROLL for (int i = 7; i >= 0; i--) {
                const int l = !lt && !gt && u[i] < v[i], g = !lt && !gt && u[i] > v[i];
                lt |= l; gt |= g;
            }
            ge = !lt;
        }
        const int u_even = !(u[0] & 1), v_even = !(v[0] & 1);
        const int second = !u_even && (v_even || !ge); /* the v pair's turn: swap the roles */
        if (second) {
            EC_UNROLL for (int i = 0; i < 8; i++) {
                const uint32_t tu = u[i], tx = x[i];
                u[i] = v[i]; v[i] = tu; x[i] = y[i]; y[i] = tx;
            }
        }
        if (!(u[0] & 1)) { /* halve u and x */
            EC_UNROLL for (int i = 0; i < 8; i++) u[i] = (u[i] >> 1) | (i < 7 ? u[i + 1] << 31 : 0);
            const uint32_t odd = x[0] & 1;
            uint64_t c = 0;
            EC_UNROLL for (int i = 0; i < 8; i++) { c += (uint64_t)x[i] + (odd ? m[i] : 0); x[i] = (uint32_t)c; c >>= 32; }
            EC_UNROLL for (int i = 0; i < 8; i++) x[i] = (x[i] >> 1) | (i < 7 ? x[i + 1] << 31 : (uint32_t)c << 31);
        } else { /* u, v odd and u >= v: u -= v, x -= y (mod m) */
            uint64_t br = 0;
            EC_UNROLL for (int i = 0; i < 8; i++) { const uint64_t d = (uint64_t)u[i] - v[i] - br; u[i] = (uint32_t)d; br = (d >> 32) & 1; }
            br = 0;
            EC_UNROLL for (int i = 0; i < 8; i++) { const uint64_t d = (uint64_t)x[i] - y[i] - br; x[i] = (uint32_t)d; br = (d >> 32) & 1; }
            if (br) {
                uint64_t c = 0;
                EC_UNROLL for (int i = 0; i < 8; i++) { c += (uint64_t)x[i] + m[i]; x[i] = (uint32_t)c; c >>= 32; }
            }
        }
    }
    uint32_t ru = 0;
    EC_UNROLL for (int i = 1; i < 8; i++) ru |= u[i];
    const int first = ru == 0 && u[0] == 1;
    ec_u256 r;
    EC_UNROLL for (int i = 0; i < 8; i++) r.w[i] = first ? x[i] : y[i];
    return r;
}
/* a vector of 16 (possibly lazy: up to 2^24 each) limbs as an integer of 9 words */
EC_HD void ec_from_limbs16(const uint32_t *l, uint32_t *out) {
    uint64_t acc = 0;
    EC_UNROLL for (int k = 0; k < 16; k += 2) {
        acc += (uint64_t)l[k] + ((uint64_t)l[k + 1] << 16);
        out[k / 2] = (uint32_t)acc;
        acc >>= 32;
    }
    out[8] = (uint32_t)acc;
}
EC_HD void ec_to_limbs16(const ec_u256 *a, ec_tp l, size_t ts) { /* limb k at l[k * ts] */
    EC_UNROLL for (int k = 0; k < 8; k++) {
        l[(size_t)(2 * k) * ts] = a->w[k] & 0xFFFFu;
        l[(size_t)(2 * k + 1) * ts] = a->w[k] >> 16;
    }
}

/* ---- reading references ------------------------------------------------------------------------------------------------- */
typedef struct ec_eval_ctx {
    const ec_spec *S;
    uint64_t *tape;          /* the cycle's tape: value t at tape[t * ts] */
    uint32_t ts;             /* 1: a tape of its own; the kernels interleave the cycles of an instance (ts = the capacity rounded up to 8:
                                the lanes of a wave are cycles, and value t of neighbouring cycles shares a cache line) */
    const uint8_t *in;       /* 128 input bytes */
    ec_ws *W;
    uint32_t base, prev_base, prev_type, inst;
} ec_eval_ctx;

EC_HD uint64_t ec_get(const ec_eval_ctx *E, uint32_t ref) {
    const uint32_t t = ec_ref_tape(E->S, ref, E->base, E->prev_base, E->prev_type, E->inst);
    return t != EC_NONE ? ((ec_ctp)E->tape)[(size_t)t * E->ts] : ec_ref_const(E->S, ref, E->in);
}
/* the 16 limbs a reference names; returns nonzero when one of them does not fit 32 bits (no limb vector of a satisfiable cycle does:
   canonical limbs are 16 bits, lazy sums of a few of them with the limbs of 4 m stay below 2^24) */
EC_HD uint32_t ec_get_vec(const ec_eval_ctx *E, uint32_t ref0, uint32_t *out) {
    /* where the sixteen values live first, then the sixteen loads side by side (one after the other they were sixteen round trips) */
    uint32_t t[16];
    uint64_t v[16], hi = 0;
    EC_UNROLL for (int i = 0; i < 16; i++) t[i] = ec_ref_tape(E->S, ref0 + (uint32_t)i, E->base, E->prev_base, E->prev_type, E->inst);
    EC_UNROLL for (int i = 0; i < 16; i++) v[i] = t[i] != EC_NONE ? ((ec_ctp)E->tape)[(size_t)t[i] * E->ts] : ec_ref_const(E->S, ref0 + (uint32_t)i, E->in);
    EC_UNROLL for (int i = 0; i < 16; i++) {
        out[i] = (uint32_t)v[i];
        hi |= v[i] >> 32;
    }
    return hi != 0;
}
/* q (16 limbs, the top one up to 24 bits) and the 15 carries (+ 2^31) of a MUL row (element k at [k * ts]); a, b, r: limb vectors (W->va /
   vb / vc or any memory); returns 0 when a * b + 8 m - r is not a non-negative multiple of m (no witness) */
EC_HD int ec_mul_witness(const uint32_t *a, const uint32_t *b, const uint32_t *r, uint32_t which, ec_tp q, ec_tp c, size_t ts, ec_ws *W) {
    const ec_mod M = ec_modulus(which);
    uint32_t *A = W->A, *B = W->B, *T = W->big, *Q = W->big + 20, *Qc = W->big + 32;
    ec_from_limbs16(a, A);
    ec_from_limbs16(b, B);
    ec_mul_words(A, 9, B, 9, T); /* 18 words */
    T[18] = T[19] = 0;
    uint64_t carry = 0; /* + 8 m */
    for (int i = 0; i < 20; i++) {
        uint64_t add = 0;
        if (i < 9) add = i < 8 ? (((uint64_t)M.m[i] << 3) & 0xFFFFFFFFull) | (i ? M.m[i - 1] >> 29 : 0) : (uint64_t)(M.m[7] >> 29);
        const uint64_t t = (uint64_t)T[i] + add + carry;
        T[i] = (uint32_t)t;
        carry = t >> 32;
    }
    uint64_t br = 0, racc = 0; /* - r: its nine words come off the limbs as they are needed */
    for (int i = 0; i < 20; i++) {
        uint64_t rw = 0;
        if (i < 8) {
            racc += (uint64_t)r[2 * i] + ((uint64_t)r[2 * i + 1] << 16);
            rw = (uint32_t)racc;
            racc >>= 32;
        } else if (i == 8) rw = (uint32_t)racc;
        const uint64_t d = (uint64_t)T[i] - rw - br;
        T[i] = (uint32_t)d;
        br = (d >> 32) & 1;
    }
    if (br) return 0;
    /* Q = T / m exactly, m = 2^256 - c: Q <- ceil((T + Q c) / 2^256) from Q = T >> 256 (the sum's low eight words only carry) */
    for (int i = 0; i < 12; i++) Q[i] = T[8 + i];
    for (int it = 0; it < 4; it++) {
        ec_mul_words(Q, 12, M.c, (int)M.nc, Qc); /* 12 + nc words */
        for (int i = 12 + (int)M.nc; i < 20; i++) Qc[i] = 0;
        uint64_t cy = 0;
        for (int i = 0; i < 20; i++) {
            const uint64_t t = (uint64_t)T[i] + Qc[i] + (i < 8 ? 0xFFFFFFFFull : 0) + cy;
            if (i >= 8) Q[i - 8] = (uint32_t)t;
            cy = t >> 32;
        }
    }
    if (Q[8] >> 8 || Q[9] || Q[10] || Q[11]) return 0; /* q < 2^264 */
    ec_mul_words(Q, 12, M.m, 8, Qc);
    for (int i = 0; i < 20; i++)
        if (Qc[i] != T[i]) return 0;
    for (int k = 0; k < 15; k++) q[(size_t)k * ts] = (Q[k / 2] >> (16 * (k & 1))) & 0xFFFFu;
    q[(size_t)15 * ts] = (Q[7] >> 16) | ((uint64_t)Q[8] << 16);
    /* carries of the 32-bit positions. The 512 products below take the limbs of q off the quotient's words and the limbs of m off the
       workspace (Qc is free by now) — read back from the tape / through M.m they were 1 024 loads a MUL row, one after the other */
    for (int j = 0; j < 16; j++) Qc[j] = (M.m[j / 2] >> (16 * (j & 1))) & 0xFFFFu;
    int64_t cin = 0;
    for (int k = 0; k < 16; k++) {
        int64_t d = 0;
        for (int half = 0; half < 2; half++) {
            const int t = 2 * k + half;
            int64_t s = 0;
            for (int i = 0; i < 16; i++) {
                const int j = t - i;
                if (j < 0 || j > 15) continue;
                const int64_t mj = (int64_t)Qc[j];
                const int64_t qi = i < 15 ? (int64_t)((Q[i / 2] >> (16 * (i & 1))) & 0xFFFFu) : (int64_t)((Q[7] >> 16) | ((uint64_t)Q[8] << 16));
                s += (int64_t)a[i] * (int64_t)b[j] - (qi - (i == 0 ? EC_KMUL : 0)) * mj;
            }
            if (t < 16) s -= (int64_t)r[t];
            d += half ? s * 65536 : s;
        }
        const int64_t tot = d + cin;
        if (tot & 0xFFFFFFFFll) return 0;
        cin = tot >> 32;
        if (k < 15) {
            if (cin <= -(1ll << 31) || cin >= (1ll << 31)) return 0;
            c[(size_t)k * ts] = (uint64_t)(cin + (1ll << 31));
        } else if (cin != 0) return 0;
    }
    return 1;
}

/* the limb vector a reference names, reduced mod m */
EC_HD ec_u256 ec_get_reduced(const ec_eval_ctx *E, uint32_t ref0, const ec_mod *M, uint32_t *wide) {
    *wide |= ec_get_vec(E, ref0, E->W->va);
    ec_from_limbs16(E->W->va, E->W->A);
    return ec_reduce(E->W->A, 9, M, E->W);
}

/* evaluates the items of one segment instance onto the tape; returns 0, or 1 + the item's index when the inputs have no witness
   (a division by zero in the incomplete addition, a broken assertion) */
EC_HD int ec_eval_items_of(ec_eval_ctx *E, uint32_t type, uint32_t first, uint32_t count, const int small_only);
EC_HD int ec_eval_items(ec_eval_ctx *E, uint32_t type, uint32_t first, uint32_t count) { return ec_eval_items_of(E, type, first, count, 0); }
EC_HD int ec_eval_segment(ec_eval_ctx *E, uint32_t type) { return ec_eval_items(E, type, 0, E->S->types[type].n_items); }
/* items [first, first + count) of the segment type, in order (the whole segment, or one of its parts: EC_PART_ITEMS_INIT). small_only (a
   constant at every call site: the branch folds): the list holds no MUL row and no hint — a segment's LEAVES — so E->W is not used (nor
   the registers of the 256-bit arithmetic); such an item in the list is reported as one without a witness */
EC_HD int ec_eval_items_of(ec_eval_ctx *E_in, uint32_t type, uint32_t first, uint32_t count, const int small_only) {
    /* every lane of a wave walks the SAME list (its lanes are cycles): the position in the list and what a reference decodes through are
       uniform, and saying so (EC_UNIFORM) lets the item words and the spec's tables come by scalar loads; only tape values and input bytes
       are per lane */
    ec_eval_ctx U = *E_in, *E = &U;
    U.base = EC_UNIFORM(U.base); U.prev_base = EC_UNIFORM(U.prev_base); U.prev_type = EC_UNIFORM(U.prev_type); U.inst = EC_UNIFORM(U.inst); U.ts = EC_UNIFORM(U.ts);
    type = EC_UNIFORM(type); first = EC_UNIFORM(first); count = EC_UNIFORM(count);
    const ec_spec *S = E->S;
    const ec_ctype T = (ec_ctype)S->types + type;
    ec_cw w = (ec_cw)S->items + EC_UNIFORM(T->item0 + ((ec_cw)S->item_index)[T->index0 + first]);
    const size_t ts = E->ts;
    ec_tp tape = (ec_tp)E->tape + (size_t)E->base * ts; /* value k of this segment at tape[k * ts] */
    ec_ws *W = E->W;
    for (uint32_t n = first; n < first + count; n++, w += ec_item_words_of(w[0], w[1])) {
        const uint32_t kind = w[0] & 15, aux = w[0] >> 24;
        if (kind == EC_I_LIN) {
            const uint32_t nk = aux, nn = w[1];
            ec_cw kn = w + 4, nw = w + 4 + 2 * nk;
            if (nn == 1 && nw[1] == 0) { /* one NEW cell = the whole sum, in the field (it may be "negative") */
                uint64_t acc = ec_gl_from_i64((int64_t)((uint64_t)w[2] | ((uint64_t)w[3] << 32)));
                for (uint32_t i = 0; i < nk; i++) {
                    const uint64_t v = ec_get(E, kn[2 * i]);
                    acc = ec_gl_add(acc, ec_gl_mul(v, ec_gl_from_i64((int32_t)kn[2 * i + 1])));
                }
                tape[(size_t)(nw[0]) * ts] = acc;
                continue;
            }
            int64_t s = (int64_t)((uint64_t)w[2] | ((uint64_t)w[3] << 32));
            for (uint32_t i = 0; i < nk; i++) s += (int64_t)ec_get(E, kn[2 * i]) * (int64_t)(int32_t)kn[2 * i + 1];
            if (nn == 0) { if (s != 0) return 1 + (int)n; continue; }
            if (s < 0) return 1 + (int)n;
            for (uint32_t i = 0; i < nn; i++) {
                const uint32_t sh = nw[2 * i + 1] & 0xFF, wd = nw[2 * i + 1] >> 8;
                uint64_t x = (uint64_t)s >> sh;
                if (wd) x &= (1ull << wd) - 1;
                tape[(size_t)(nw[2 * i]) * ts] = x;
            }
        } else if (kind == EC_I_SEL) {
            tape[(size_t)(w[4]) * ts] = ec_get(E, w[1]) ? ec_get(E, w[2]) : ec_get(E, w[3]);
        } else if (kind == EC_I_FMA) {
            const uint64_t v = ec_gl_add(ec_gl_mul(ec_get(E, w[1]) % EC_GL_P, ec_get(E, w[2]) % EC_GL_P), ec_get(E, w[3]) % EC_GL_P);
            if (aux) tape[(size_t)(w[4]) * ts] = v;
            else if (v != ec_get(E, w[4]) % EC_GL_P) return 1 + (int)n;
        } else if (small_only && (kind == EC_I_MUL || kind == EC_I_HINT)) {
            return 1 + (int)n;
        } else if (kind == EC_I_MUL) {
            if (ec_get_vec(E, w[1], W->va) | ec_get_vec(E, w[2], W->vb) | ec_get_vec(E, w[3], W->vc)) return 1 + (int)n;
            if (!ec_mul_witness(W->va, W->vb, W->vc, aux, tape + (size_t)w[4] * ts, tape + (size_t)w[5] * ts, ts, W)) return 1 + (int)n;
        } else if (kind == EC_I_LOOKUP) {
            const uint64_t a = ec_get(E, w[2]);
            if ((w[1] & 0xFF) == EC_T_XOR8) {
                const uint64_t b = ec_get(E, w[3]);
                if (a > 255 || b > 255) return 1 + (int)n;
                tape[(size_t)(w[4]) * ts] = a ^ b;
            } else {
                if (a > 255) return 1 + (int)n;
                const uint32_t tb = (w[1] & 0xFF) - EC_T_FIXED0 + 8 * E->inst;
                tape[(size_t)(w[4]) * ts] = ((ec_cw)S->fixed)[((size_t)tb * 256 + a) * 2];
                tape[(size_t)(w[4] + 1) * ts] = ((ec_cw)S->fixed)[((size_t)tb * 256 + a) * 2 + 1];
            }
        } else if (small_only) {
            return 1 + (int)n;
        } else if (aux == EC_H_MULSUB || aux == EC_H_DIV) {
            const ec_mod M = ec_modulus(w[1]);
            uint32_t wide = 0;
            const ec_u256 a = ec_get_reduced(E, w[2], &M, &wide), b = ec_get_reduced(E, w[3], &M, &wide);
            if (wide) return 1 + (int)n;
            ec_u256 res;
            if (aux == EC_H_DIV) {
                if (ec_is_zero8(&b)) return 1 + (int)n;
                const ec_u256 bi = ec_invmod(&b, &M, W);
                res = ec_mulmod(&a, &bi, &M, W);
                ec_to_limbs16(&res, tape + (size_t)w[4] * ts, ts);
            } else {
                res = ec_mulmod(&a, &b, &M, W);
                if (w[4] != EC_NONE) { const ec_u256 cc = ec_get_reduced(E, w[4], &M, &wide); res = ec_submod(&res, &cc, &M); }
                if (w[5] != EC_NONE) { const ec_u256 dd = ec_get_reduced(E, w[5], &M, &wide); res = ec_submod(&res, &dd, &M); }
                if (wide) return 1 + (int)n;
                ec_to_limbs16(&res, tape + (size_t)w[6] * ts, ts);
            }
        } else if (aux == EC_H_SQRT) {
            const ec_mod M = ec_modulus(0);
            uint32_t wide = 0;
            const ec_u256 t = ec_get_reduced(E, w[1], &M, &wide);
            if (wide) return 1 + (int)n;
            ec_u256 y = ec_powmod(&t, EC_P_SQRT_E, &M, W);
            const ec_u256 y2 = ec_mulmod(&y, &y, &M, W), zero = ec_zero256();
            uint64_t e_nr = 0;
            if (ec_cmp8(y2.w, t.w) != 0) { /* no root: a root of -t proves it (p = 3 mod 4) */
                e_nr = 1;
                const ec_u256 nt = ec_submod(&zero, &t, &M);
                y = ec_powmod(&nt, EC_P_SQRT_E, &M, W);
            } else if ((y.w[0] & 1) != (ec_get(E, w[2]) & 1)) {
                y = ec_submod(&zero, &y, &M);
            }
            ec_to_limbs16(&y, tape + (size_t)w[3] * ts, ts);
            tape[(size_t)(w[3] + 16) * ts] = e_nr;
        } else if (aux == EC_H_ISZERO) {
            const uint64_t x = ec_get(E, w[1]) % EC_GL_P;
            tape[(size_t)(w[2]) * ts] = x ? ec_gl_inv(x) : 0;
            tape[(size_t)(w[2] + 1) * ts] = x ? 0 : 1;
        } else { /* EC_H_GE: a >= the constant */
            int ge = 1;
            for (int i = 15; i >= 0; i--) {
                const uint64_t av = ec_get(E, w[1] + (uint32_t)i), cst = ((ec_cw)S->bigs)[w[2] * 16 + (uint32_t)i];
                if (av != cst) { ge = av > cst; break; }
            }
            tape[(size_t)(w[3]) * ts] = (uint64_t)ge;
        }
    }
    return 0;
}

/* the whole cycle: tape[EC_TAPE_PER_CYCLE] from the 128 input bytes. Returns 0, or (run << 24 | instance << 12 | 1 + item) of the
   first item without a witness */
EC_HD uint32_t ec_eval_cycle_strided(const ec_spec *S, const uint8_t *in, uint64_t *tape, uint32_t ts, ec_ws *W) {
    ec_eval_ctx E;
    E.S = S; E.tape = tape; E.ts = ts; E.in = in; E.W = W;
    E.prev_base = 0; E.prev_type = 0;
    for (uint32_t r = 0; r < EC_NUM_RUNS; r++) {
        const ec_run *R = &S->runs[r];
        const ec_seg_type *T = &S->types[R->type];
        for (uint32_t j = 0; j < R->count; j++) {
            E.base = R->tape0 + j * T->n_tape;
            E.inst = j;
            const int bad = ec_eval_segment(&E, R->type);
            if (bad) return (r << 24) | (j << 12) | (uint32_t)bad;
            E.prev_base = E.base;
            E.prev_type = R->type;
        }
    }
    return 0;
}
EC_HD uint32_t ec_eval_cycle(const ec_spec *S, const uint8_t *in, uint64_t *tape, ec_ws *W) { return ec_eval_cycle_strided(S, in, tape, 1, W); }

/* ---- the relations, from cells alone. `cell(col)` reads the item's row; returns 0 when the relation holds ------------------ */
typedef struct ec_row_view { const uint64_t *trace; size_t n_rows, row; } ec_row_view;
EC_HD uint64_t ec_rv(const ec_row_view *v, uint32_t col) { return v->trace[(size_t)col * v->n_rows + v->row]; }
EC_HD int ec_canon(uint64_t x) { return x < EC_GL_P; }

EC_HD int ec_check_item(const ec_spec *S, const uint32_t *w, const ec_row_view *v, uint32_t inst) {
    const uint32_t kind = w[0] & 15, aux = w[0] >> 24, col = (w[0] >> 16) & 0xFF;
    if (kind == EC_I_LIN) {
        const uint32_t nk = aux, nn = w[1];
        uint64_t acc = ec_gl_from_i64((int64_t)((uint64_t)w[2] | ((uint64_t)w[3] << 32)));
        for (uint32_t i = 0; i < nk; i++) {
            const uint64_t x = ec_rv(v, col + i);
            if (!ec_canon(x)) return 1;
            acc = ec_gl_add(acc, ec_gl_mul(x, ec_gl_from_i64((int32_t)w[4 + 2 * i + 1])));
        }
        for (uint32_t i = 0; i < nn; i++) {
            const uint64_t x = ec_rv(v, col + nk + i);
            if (!ec_canon(x)) return 1;
            acc = ec_gl_sub(acc, ec_gl_mul(x, 1ull << (w[4 + 2 * nk + 2 * i + 1] & 0xFF)));
        }
        return acc != 0;
    }
    if (kind == EC_I_SEL) {
        const uint64_t b = ec_rv(v, col), x = ec_rv(v, col + 1), y = ec_rv(v, col + 2), o = ec_rv(v, col + 3);
        if (!ec_canon(b) || !ec_canon(x) || !ec_canon(y) || !ec_canon(o)) return 1;
        return ec_gl_sub(ec_gl_add(ec_gl_mul(b, ec_gl_sub(x, y)), y), o) != 0;
    }
    if (kind == EC_I_FMA) {
        const uint64_t a = ec_rv(v, col), b = ec_rv(v, col + 1), c = ec_rv(v, col + 2), d = ec_rv(v, col + 3);
        if (!ec_canon(a) || !ec_canon(b) || !ec_canon(c) || !ec_canon(d)) return 1;
        return ec_gl_add(ec_gl_mul(a, b), c) != d;
    }
    if (kind == EC_I_MUL) {
        const ec_mod M = ec_modulus(aux);
        uint64_t cin = 0; /* carry c_{k-1} - 2^31, in the field */
        for (int c = 0; c < 80; c++)
            if (!ec_canon(ec_rv(v, (uint32_t)c))) return 1;
        for (int k = 0; k < 16; k++) {
            uint64_t d = 0;
            for (int half = 0; half < 2; half++) {
                const int t = 2 * k + half;
                uint64_t s = 0;
                for (int i = 0; i < 16; i++) {
                    const int j = t - i;
                    if (j < 0 || j > 15) continue;
                    const uint64_t mj = (M.m[j / 2] >> (16 * (j & 1))) & 0xFFFFu;
                    s = ec_gl_add(s, ec_gl_mul(ec_rv(v, (uint32_t)i), ec_rv(v, 16u + (uint32_t)j)));
                    const uint64_t qi = i == 0 ? ec_gl_sub(ec_rv(v, 32), EC_KMUL) : ec_rv(v, 32u + (uint32_t)i);
                    s = ec_gl_sub(s, ec_gl_mul(qi, mj));
                }
                if (t < 16) s = ec_gl_sub(s, ec_rv(v, 48u + (uint32_t)t));
                d = ec_gl_add(d, half ? ec_gl_mul(s, 65536) : s);
            }
            const uint64_t cout = k < 15 ? ec_gl_sub(ec_rv(v, 64u + (uint32_t)k), 1ull << 31) : 0;
            if (ec_gl_add(d, cin) != ec_gl_mul(cout, 1ull << 32)) return 1;
            cin = cout;
        }
        return ec_rv(v, 79) != 0;
    }
    if (kind == EC_I_LOOKUP) {
        const uint32_t c0 = EC_G + EC_W * col;
        const uint64_t a = ec_rv(v, c0), b = ec_rv(v, c0 + 1), c = ec_rv(v, c0 + 2);
        if ((w[1] & 0xFF) == EC_T_XOR8) return a > 255 || b > 255 || c != (a ^ b);
        const uint32_t tb = (w[1] & 0xFF) - EC_T_FIXED0 + 8 * inst;
        return a > 255 || b != S->fixed[((size_t)tb * 256 + a) * 2] || c != S->fixed[((size_t)tb * 256 + a) * 2 + 1];
    }
    return 0; /* hints state nothing */
}

/* ---- Jacobian arithmetic without case distinctions (the kernels' accumulator chain): in place on three 256-bit values; a degenerate
   input (the point at infinity, equal x in the addition) leaves z == 0, which the caller checks once at the end -------------------- */
EC_HD void ec_jdbl(ec_u256 *X, ec_u256 *Y, ec_u256 *Z, const ec_mod *M) { /* a = 0: 7 multiplications */
    const ec_u256 a = ec_mulmod(X, X, M, 0), b = ec_mulmod(Y, Y, M, 0), c = ec_mulmod(&b, &b, M, 0);
    ec_u256 t = ec_addmod(X, &b, M, 0);
    t = ec_mulmod(&t, &t, M, 0);
    t = ec_submod(&t, &a, M);
    t = ec_submod(&t, &c, M);
    const ec_u256 d = ec_addmod(&t, &t, M, 0);
    ec_u256 e = ec_addmod(&a, &a, M, 0);
    e = ec_addmod(&e, &a, M, 0);
    const ec_u256 f = ec_mulmod(&e, &e, M, 0), d2 = ec_addmod(&d, &d, M, 0);
    const ec_u256 yz = ec_mulmod(Y, Z, M, 0);
    *X = ec_submod(&f, &d2, M);
    ec_u256 c8 = ec_addmod(&c, &c, M, 0);
    c8 = ec_addmod(&c8, &c8, M, 0);
    c8 = ec_addmod(&c8, &c8, M, 0);
    ec_u256 dx = ec_submod(&d, X, M);
    dx = ec_mulmod(&e, &dx, M, 0);
    *Y = ec_submod(&dx, &c8, M);
    *Z = ec_addmod(&yz, &yz, M, 0);
}
EC_HD void ec_jmadd(ec_u256 *X, ec_u256 *Y, ec_u256 *Z, const ec_u256 *x2, const ec_u256 *y2, const ec_mod *M) { /* + the affine (x2, y2): 11 */
    const ec_u256 zz = ec_mulmod(Z, Z, M, 0), zzz = ec_mulmod(&zz, Z, M, 0);
    const ec_u256 u2 = ec_mulmod(x2, &zz, M, 0), s2 = ec_mulmod(y2, &zzz, M, 0);
    const ec_u256 h = ec_submod(&u2, X, M), r = ec_submod(&s2, Y, M);
    const ec_u256 h2 = ec_mulmod(&h, &h, M, 0), h3 = ec_mulmod(&h2, &h, M, 0), xh2 = ec_mulmod(X, &h2, M, 0);
    ec_u256 t = ec_mulmod(&r, &r, M, 0);
    t = ec_submod(&t, &h3, M);
    t = ec_submod(&t, &xh2, M);
    const ec_u256 x3 = ec_submod(&t, &xh2, M);
    ec_u256 v = ec_submod(&xh2, &x3, M);
    v = ec_mulmod(&r, &v, M, 0);
    const ec_u256 yh3 = ec_mulmod(Y, &h3, M, 0);
    *Z = ec_mulmod(Z, &h, M, 0);
    *X = x3;
    *Y = ec_submod(&v, &yh3, M);
}

/* ---- the 256 FixedBaseMul tables (host): word i of x and of y of byte * 2^(8 C) * G, (0, 0) for byte 0 ---------------------- */
typedef struct ec_jac { ec_u256 x, y, z; } ec_jac; /* z == 0: infinity */
EC_HD ec_jac ec_jac_double(const ec_jac *p, const ec_mod *M, ec_ws *W) {
    if (ec_is_zero8(&p->z)) return *p;
    ec_jac r;
    const ec_u256 a = ec_mulmod(&p->x, &p->x, M, W), b = ec_mulmod(&p->y, &p->y, M, W), c = ec_mulmod(&b, &b, M, W);
    ec_u256 t = ec_addmod(&p->x, &b, M, W);
    t = ec_mulmod(&t, &t, M, W);
    t = ec_submod(&t, &a, M);
    t = ec_submod(&t, &c, M);
    const ec_u256 d = ec_addmod(&t, &t, M, W);
    ec_u256 e = ec_addmod(&a, &a, M, W);
    e = ec_addmod(&e, &a, M, W);
    const ec_u256 f = ec_mulmod(&e, &e, M, W);
    ec_u256 d2 = ec_addmod(&d, &d, M, W);
    r.x = ec_submod(&f, &d2, M);
    ec_u256 c8 = ec_addmod(&c, &c, M, W);
    c8 = ec_addmod(&c8, &c8, M, W);
    c8 = ec_addmod(&c8, &c8, M, W);
    ec_u256 dx = ec_submod(&d, &r.x, M);
    dx = ec_mulmod(&e, &dx, M, W);
    r.y = ec_submod(&dx, &c8, M);
    const ec_u256 yz = ec_mulmod(&p->y, &p->z, M, W);
    r.z = ec_addmod(&yz, &yz, M, W);
    return r;
}
EC_HD ec_jac ec_jac_add(const ec_jac *p, const ec_jac *q, const ec_mod *M, ec_ws *W) {
    if (ec_is_zero8(&p->z)) return *q;
    if (ec_is_zero8(&q->z)) return *p;
    const ec_u256 z1z1 = ec_mulmod(&p->z, &p->z, M, W), z2z2 = ec_mulmod(&q->z, &q->z, M, W);
    const ec_u256 u1 = ec_mulmod(&p->x, &z2z2, M, W), u2 = ec_mulmod(&q->x, &z1z1, M, W);
    ec_u256 s1 = ec_mulmod(&p->y, &q->z, M, W);
    s1 = ec_mulmod(&s1, &z2z2, M, W);
    ec_u256 s2 = ec_mulmod(&q->y, &p->z, M, W);
    s2 = ec_mulmod(&s2, &z1z1, M, W);
    const ec_u256 h = ec_submod(&u2, &u1, M), rr = ec_submod(&s2, &s1, M);
    if (ec_is_zero8(&h)) {
        if (ec_is_zero8(&rr)) return ec_jac_double(p, M, W);
        ec_jac inf = *p;
        for (int i = 0; i < 8; i++) inf.z.w[i] = 0;
        return inf;
    }
    const ec_u256 h2 = ec_mulmod(&h, &h, M, W), h3 = ec_mulmod(&h2, &h, M, W), u1h2 = ec_mulmod(&u1, &h2, M, W);
    ec_jac r;
    ec_u256 t = ec_mulmod(&rr, &rr, M, W);
    t = ec_submod(&t, &h3, M);
    t = ec_submod(&t, &u1h2, M);
    r.x = ec_submod(&t, &u1h2, M);
    ec_u256 v = ec_submod(&u1h2, &r.x, M);
    v = ec_mulmod(&rr, &v, M, W);
    const ec_u256 s1h3 = ec_mulmod(&s1, &h3, M, W);
    r.y = ec_submod(&v, &s1h3, M);
    const ec_u256 zz = ec_mulmod(&p->z, &q->z, M, W);
    r.z = ec_mulmod(&zz, &h, M, W);
    return r;
}
#include <stdlib.h>
/* out[EC_FIXED_WORDS]; boojum's create_fixed_base_mul_table<i, C> (gadgets/tables/fixed_base_mul_table, absent crate; the contents
   are the public curve: row `byte` of table (i, C) = 32-bit word i of the affine x and y of byte * 2^(8 C) * G) */
static inline void ec_build_fixed_tables(uint32_t *out) {
    const ec_mod M = ec_modulus(0);
    ec_ws ws;
    ec_jac *pts = (ec_jac *)malloc(sizeof(ec_jac) * 32 * 256);
    ec_u256 *pre = (ec_u256 *)malloc(sizeof(ec_u256) * 32 * 256);
    ec_jac base;
    const uint32_t gx[8] = {0x16F81798u, 0x59F2815Bu, 0x2DCE28D9u, 0x029BFCDBu, 0xCE870B07u, 0x55A06295u, 0xF9DCBBACu, 0x79BE667Eu};
    const uint32_t gy[8] = {0xFB10D4B8u, 0x9C47D08Fu, 0xA6855419u, 0xFD17B448u, 0x0E1108A8u, 0x5DA4FBFCu, 0x26A3C465u, 0x483ADA77u};
    for (int i = 0; i < 8; i++) { base.x.w[i] = gx[i]; base.y.w[i] = gy[i]; base.z.w[i] = i == 0; }
    for (int C = 0; C < 32; C++) {
        ec_jac cur = base;
        for (int i = 0; i < 8; i++) cur.z.w[i] = 0; /* infinity */
        for (int b = 0; b < 256; b++) {
            pts[C * 256 + b] = cur;
            cur = ec_jac_add(&cur, &base, &M, &ws);
        }
        for (int d = 0; d < 8; d++) base = ec_jac_double(&base, &M, &ws);
    }
    /* one inversion for all z (Montgomery's trick), infinity skipped */
    ec_u256 acc;
    for (int i = 0; i < 8; i++) acc.w[i] = i == 0;
    for (int k = 0; k < 32 * 256; k++) {
        pre[k] = acc;
        if (!ec_is_zero8(&pts[k].z)) acc = ec_mulmod(&acc, &pts[k].z, &M, &ws);
    }
    ec_u256 inv = ec_invmod(&acc, &M, &ws);
    for (int k = 32 * 256 - 1; k >= 0; k--) {
        const int C = k / 256, b = k % 256;
        ec_u256 x, y;
        for (int i = 0; i < 8; i++) x.w[i] = y.w[i] = 0;
        if (!ec_is_zero8(&pts[k].z)) {
            const ec_u256 zi = ec_mulmod(&inv, &pre[k], &M, &ws);
            inv = ec_mulmod(&inv, &pts[k].z, &M, &ws);
            const ec_u256 zi2 = ec_mulmod(&zi, &zi, &M, &ws), zi3 = ec_mulmod(&zi2, &zi, &M, &ws);
            x = ec_mulmod(&pts[k].x, &zi2, &M, &ws);
            y = ec_mulmod(&pts[k].y, &zi3, &M, &ws);
        }
        for (int i = 0; i < 8; i++) {
            out[((size_t)(8 * C + i) * 256 + (size_t)b) * 2] = x.w[i];
            out[((size_t)(8 * C + i) * 256 + (size_t)b) * 2 + 1] = y.w[i];
        }
    }
    free(pts);
    free(pre);
}
#endif /* ZKW_ECRECOVER_H */
