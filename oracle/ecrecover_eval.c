/* TEST INFRASTRUCTURE — ecrecover_eval.c: the oracle's OWN evaluator and relation checker of the ECRecover circuit's EC section.
 *
 * The library evaluates a cycle's value tape and checks an item's relation with include/zkw_ecrecover.h (constant-shape 256-bit
 * arithmetic with special-form folds, Fermat inversions, Jacobian chains: written for the GPU). Until round 5 the oracle compiled that
 * same header, so "GPU == oracle" on these cells proved the schedule and not the semantics. This file shares NO code with
 * it: it reads the generated spec through the format header include/zkw_ecrecover_layout.h only and restates the item semantics of
 * tools/gen_ecrecover_circuit.py (Cycle.eval_segment, the statement in that file's docstring) the plain way —
 *   a little-endian big-number type with schoolbook multiplication and bit-serial long division (quotients and remainders by ANY
 *   modulus: no use of the moduli's special form), inverses by the binary extended Euclidean algorithm (not by exponentiation),
 *   the FixedBaseMul tables by repeated affine addition of 2^(8C) G.
 * Semantics restated (the circuit body `ecrecover_function_entry_point` is in the absent era-zkevm_circuits; geometry and tables are
 * circuit_definitions/src/circuit_definitions/base_layer/ecrecover.rs:30-41,138-176; a request's reads and writes
 * src/witness/individual_circuits/ecrecover.rs:143-178): LIN / SEL / FMA / MUL / LOOKUP / HINT items as listed in the layout header.
 * Results are pinned by public secp256k1 vectors (tests/test_oracle_ecrecover_circuit.py) and by the generator's Python evaluator. */
#include <stdlib.h>
#include <string.h>
#include "oracle.h"
#include "../include/zkw_ecrecover_layout.h"

/* ---- big numbers: 20 x 32-bit words (640 bits), unsigned ---------------------------------------------------------------- */
#define BW 20
typedef struct big { uint32_t w[BW]; } big;

static big big_zero(void) { big r; memset(&r, 0, sizeof r); return r; }
static big big_small(uint64_t v) { big r = big_zero(); r.w[0] = (uint32_t)v; r.w[1] = (uint32_t)(v >> 32); return r; }
static int big_cmp(const big *a, const big *b) {
    for (int i = BW - 1; i >= 0; i--)
        if (a->w[i] != b->w[i]) return a->w[i] < b->w[i] ? -1 : 1;
    return 0;
}
static int big_is_zero(const big *a) { for (int i = 0; i < BW; i++) if (a->w[i]) return 0; return 1; }
static big big_add(const big *a, const big *b) {
    big r; uint64_t c = 0;
    for (int i = 0; i < BW; i++) { c += (uint64_t)a->w[i] + b->w[i]; r.w[i] = (uint32_t)c; c >>= 32; }
    return r;
}
/* a - b; *borrow = 1 when a < b (the result is then a - b + 2^640) */
static big big_sub(const big *a, const big *b, int *borrow) {
    big r; int64_t c = 0;
    for (int i = 0; i < BW; i++) { c += (int64_t)a->w[i] - (int64_t)b->w[i]; r.w[i] = (uint32_t)c; c >>= 32; }
    if (borrow) *borrow = c != 0;
    return r;
}
static big big_mul(const big *a, const big *b) { /* truncated to BW words: callers keep the product below 2^640 */
    big r = big_zero();
    for (int i = 0; i < BW; i++) {
        if (!a->w[i]) continue;
        uint64_t c = 0;
        for (int j = 0; i + j < BW; j++) { c += (uint64_t)a->w[i] * b->w[j] + r.w[i + j]; r.w[i + j] = (uint32_t)c; c >>= 32; }
    }
    return r;
}
static int big_bit(const big *a, int i) { return (a->w[i >> 5] >> (i & 31)) & 1; }
static void big_shl1(big *a) { uint32_t c = 0; for (int i = 0; i < BW; i++) { const uint32_t n = a->w[i] >> 31; a->w[i] = (a->w[i] << 1) | c; c = n; } }
static void big_shr1(big *a) { uint32_t c = 0; for (int i = BW - 1; i >= 0; i--) { const uint32_t n = a->w[i] & 1; a->w[i] = (a->w[i] >> 1) | (c << 31); c = n; } }
/* bit-serial long division: q = floor(n / m), r = n mod m */
static void big_divmod(const big *n, const big *m, big *q, big *r) {
    big qq = big_zero(), rr = big_zero();
    int top = BW * 32 - 1;
    while (top >= 0 && !big_bit(n, top)) top--;
    for (int i = top; i >= 0; i--) {
        big_shl1(&rr);
        rr.w[0] |= (uint32_t)big_bit(n, i);
        if (big_cmp(&rr, m) >= 0) { rr = big_sub(&rr, m, NULL); qq.w[i >> 5] |= 1u << (i & 31); }
    }
    if (q) *q = qq;
    if (r) *r = rr;
}
static big big_mod(const big *n, const big *m) { big r; big_divmod(n, m, NULL, &r); return r; }
static big big_mulmod(const big *a, const big *b, const big *m) { const big p = big_mul(a, b); return big_mod(&p, m); }
static big big_submod(const big *a, const big *b, const big *m) { /* a, b < m */
    int br; big r = big_sub(a, b, &br);
    if (br) r = big_add(&r, m);
    return r;
}
static big big_powmod(const big *a, const big *e, const big *m) {
    big r = big_small(1), base = big_mod(a, m);
    int top = BW * 32 - 1;
    while (top >= 0 && !big_bit(e, top)) top--;
    for (int i = top; i >= 0; i--) { r = big_mulmod(&r, &r, m); if (big_bit(e, i)) r = big_mulmod(&r, &base, m); }
    return r;
}
/* a^-1 mod m for an odd m and 0 < a < m with gcd(a, m) = 1: the binary extended Euclidean algorithm */
static big big_invmod(const big *a, const big *m) {
    big u = *a, v = *m, x1 = big_small(1), x2 = big_zero();
    const big one = big_small(1);
    while (big_cmp(&u, &one) != 0 && big_cmp(&v, &one) != 0) {
        while (!(u.w[0] & 1)) { big_shr1(&u); if (x1.w[0] & 1) x1 = big_add(&x1, m); big_shr1(&x1); }
        while (!(v.w[0] & 1)) { big_shr1(&v); if (x2.w[0] & 1) x2 = big_add(&x2, m); big_shr1(&x2); }
        if (big_cmp(&u, &v) >= 0) { u = big_sub(&u, &v, NULL); x1 = big_submod(&x1, &x2, m); }
        else { v = big_sub(&v, &u, NULL); x2 = big_submod(&x2, &x1, m); }
    }
    return big_cmp(&u, &one) == 0 ? x1 : x2;
}
/* sum of limb_i * 2^(16 i): limbs may be "lazy" (wider than 16 bits) */
static big big_from_limbs16(const uint64_t *l, int n) {
    big r = big_zero();
    for (int i = 0; i < n; i++) {
        big t = big_zero();
        const int word = i / 2, sh = 16 * (i & 1);
        const unsigned __int128 v = (unsigned __int128)l[i] << sh;
        t.w[word] = (uint32_t)v; t.w[word + 1] = (uint32_t)(v >> 32); t.w[word + 2] = (uint32_t)(v >> 64);
        r = big_add(&r, &t);
    }
    return r;
}
static void big_to_limbs16(const big *a, uint64_t *l, int n) { for (int i = 0; i < n; i++) l[i] = (a->w[i / 2] >> (16 * (i & 1))) & 0xFFFFu; }

/* ---- secp256k1: the two moduli as numbers (SEC 2, section 2.4.1) ------------------------------------------------------------- */
static big modulus(uint32_t which) {
    big m = big_zero();
    static const uint32_t P[8] = {0xFFFFFC2Fu, 0xFFFFFFFEu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    static const uint32_t N[8] = {0xD0364141u, 0xBFD25E8Cu, 0xAF48A03Bu, 0xBAAEDCE6u, 0xFFFFFFFEu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    memcpy(m.w, which ? N : P, 32);
    return m;
}
typedef struct point { big x, y; int inf; } point;
static point pt_add(const point *p, const point *q, const big *m) { /* affine, complete by case distinction */
    if (p->inf) return *q;
    if (q->inf) return *p;
    big lam;
    if (big_cmp(&p->x, &q->x) == 0) {
        const big ysum = big_add(&p->y, &q->y), ys = big_mod(&ysum, m);
        if (big_is_zero(&ys)) { point o; memset(&o, 0, sizeof o); o.inf = 1; return o; }
        const big three = big_small(3), xx = big_mulmod(&p->x, &p->x, m), num = big_mulmod(&three, &xx, m);
        const big two_y = big_mod(&ysum, m), den = big_invmod(&two_y, m);
        lam = big_mulmod(&num, &den, m);
    } else {
        const big dy = big_submod(&q->y, &p->y, m), dx = big_submod(&q->x, &p->x, m), den = big_invmod(&dx, m);
        lam = big_mulmod(&dy, &den, m);
    }
    point r; r.inf = 0;
    const big l2 = big_mulmod(&lam, &lam, m), t = big_submod(&l2, &p->x, m);
    r.x = big_submod(&t, &q->x, m);
    const big d = big_submod(&p->x, &r.x, m), ld = big_mulmod(&lam, &d, m);
    r.y = big_submod(&ld, &p->y, m);
    return r;
}
/* FixedBaseMulTable<i, C>[byte] = (word i of x, word i of y) of byte * 2^(8 C) * G, (0, 0) for byte 0; out[((8 C + i) * 256 + byte) * 2 + {0, 1}] */
void orc_ec_build_fixed_own(uint32_t *out) {
    const big m = modulus(0);
    static const uint32_t GX[8] = {0x16F81798u, 0x59F2815Bu, 0x2DCE28D9u, 0x029BFCDBu, 0xCE870B07u, 0x55A06295u, 0xF9DCBBACu, 0x79BE667Eu};
    static const uint32_t GY[8] = {0xFB10D4B8u, 0x9C47D08Fu, 0xA6855419u, 0xFD17B448u, 0x0E1108A8u, 0x5DA4FBFCu, 0x26A3C465u, 0x483ADA77u};
    point base; memset(&base, 0, sizeof base);
    memcpy(base.x.w, GX, 32); memcpy(base.y.w, GY, 32);
    for (int C = 0; C < 32; C++) {
        point cur; memset(&cur, 0, sizeof cur); cur.inf = 1;
        for (int byte = 0; byte < 256; byte++) {
            for (int i = 0; i < 8; i++) {
                out[((size_t)(8 * C + i) * 256 + byte) * 2] = cur.inf ? 0 : cur.x.w[i];
                out[((size_t)(8 * C + i) * 256 + byte) * 2 + 1] = cur.inf ? 0 : cur.y.w[i];
            }
            cur = pt_add(&cur, &base, &m);
        }
        for (int k = 0; k < 8; k++) base = pt_add(&base, &base, &m); /* 2^8 base */
    }
}

/* ---- the value tape of a cycle ---------------------------------------------------------------------------------------------- */
typedef struct ectx { const ec_spec *S; uint64_t *tape; const uint8_t *in; uint32_t base, pbase, ptype, inst; } ectx;
static uint64_t getv(const ectx *E, uint32_t ref) {
    const uint32_t t = ec_ref_tape(E->S, ref, E->base, E->pbase, E->ptype, E->inst);
    return t != EC_NONE ? E->tape[t] : ec_ref_const(E->S, ref, E->in);
}
static void getvec(const ectx *E, uint32_t ref0, uint64_t *out) { for (uint32_t i = 0; i < 16; i++) out[i] = getv(E, ref0 + i); }
static uint64_t fe_from_i64(int64_t v) { return v >= 0 ? (uint64_t)v % EC_GL_P : (EC_GL_P - ((uint64_t)(-v) % EC_GL_P)) % EC_GL_P; }
static uint64_t limb_of_m(const big *m, int j) { return (m->w[j / 2] >> (16 * (j & 1))) & 0xFFFFu; }

/* q (16 limbs) and carries (15, stored + 2^31) of a MUL row: a * b + 8 m = q m + r; 0 when there is no witness */
static int mul_row(const uint64_t *a, const uint64_t *b, const uint64_t *r, uint32_t which, uint64_t *q, uint64_t *c) {
    const big m = modulus(which), A = big_from_limbs16(a, 16), B = big_from_limbs16(b, 16), R = big_from_limbs16(r, 16);
    big num = big_mul(&A, &B);
    const big k = big_small(EC_KMUL), km = big_mul(&k, &m);
    num = big_add(&num, &km);
    int br;
    num = big_sub(&num, &R, &br);
    if (br) return 0;
    big Q, rem;
    big_divmod(&num, &m, &Q, &rem);
    if (!big_is_zero(&rem)) return 0;
    for (int i = 9; i < BW; i++) if (Q.w[i]) return 0;
    if (Q.w[8] >> 8) return 0; /* q < 2^264 */
    for (int i = 0; i < 15; i++) q[i] = (Q.w[i / 2] >> (16 * (i & 1))) & 0xFFFFu;
    q[15] = (Q.w[7] >> 16) | ((uint64_t)Q.w[8] << 16);
    __int128 carry = 0;
    for (int kk = 0; kk < 16; kk++) {
        __int128 d = 0;
        for (int half = 0; half < 2; half++) {
            const int t = 2 * kk + half;
            __int128 s = 0;
            for (int i = 0; i < 16; i++) {
                const int j = t - i;
                if (j < 0 || j > 15) continue;
                s += (__int128)a[i] * b[j] - ((__int128)q[i] - (i == 0 ? EC_KMUL : 0)) * (__int128)limb_of_m(&m, j);
            }
            if (t < 16) s -= (__int128)r[t];
            d += half ? s * 65536 : s;
        }
        const __int128 tot = d + carry;
        if (tot & 0xFFFFFFFF) return 0;
        carry = tot >> 32; /* arithmetic */
        if (kk < 15) {
            if (carry <= -((__int128)1 << 31) || carry >= ((__int128)1 << 31)) return 0;
            c[kk] = (uint64_t)(carry + ((__int128)1 << 31));
        } else if (carry != 0) return 0;
    }
    return 1;
}

static big vec_mod(const ectx *E, uint32_t ref0, const big *m) { uint64_t l[16]; getvec(E, ref0, l); const big v = big_from_limbs16(l, 16); return big_mod(&v, m); }

static int eval_segment(ectx *E, uint32_t type) {
    const ec_spec *S = E->S;
    const ec_seg_type *T = &S->types[type];
    const uint32_t *w = S->items + T->item0;
    uint64_t *tape = E->tape + E->base;
    for (uint32_t n = 0; n < T->n_items; n++, w += ec_item_words(w)) {
        const uint32_t kind = w[0] & 15, aux = w[0] >> 24;
        if (kind == EC_I_LIN) {
            const uint32_t nk = aux, nn = w[1];
            const uint32_t *kn = w + 4, *nw = w + 4 + 2 * nk;
            const int64_t cst = (int64_t)((uint64_t)w[2] | ((uint64_t)w[3] << 32));
            if (nn == 1 && nw[1] == 0) { /* the NEW cell is the sum itself, as a field element */
                uint64_t acc = fe_from_i64(cst);
                for (uint32_t i = 0; i < nk; i++) acc = orc_gl_add(acc, orc_gl_mul(getv(E, kn[2 * i]) % EC_GL_P, fe_from_i64((int32_t)kn[2 * i + 1])));
                tape[nw[0]] = acc;
                continue;
            }
            __int128 s = cst;
            for (uint32_t i = 0; i < nk; i++) s += (__int128)getv(E, kn[2 * i]) * (int32_t)kn[2 * i + 1];
            if (nn == 0) { if (s != 0) return 1 + (int)n; continue; } /* an assertion */
            if (s < 0) return 1 + (int)n;                              /* digits of a negative number */
            for (uint32_t i = 0; i < nn; i++) {
                const uint32_t sh = nw[2 * i + 1] & 0xFF, wd = nw[2 * i + 1] >> 8;
                unsigned __int128 x = (unsigned __int128)s >> sh;
                if (wd) x &= (((unsigned __int128)1) << wd) - 1;
                tape[nw[2 * i]] = (uint64_t)x;
            }
        } else if (kind == EC_I_SEL) {
            tape[w[4]] = getv(E, w[1]) ? getv(E, w[2]) : getv(E, w[3]);
        } else if (kind == EC_I_FMA) {
            const uint64_t v = orc_gl_add(orc_gl_mul(getv(E, w[1]) % EC_GL_P, getv(E, w[2]) % EC_GL_P), getv(E, w[3]) % EC_GL_P);
            if (aux) tape[w[4]] = v;
            else if (v != getv(E, w[4]) % EC_GL_P) return 1 + (int)n;
        } else if (kind == EC_I_MUL) {
            uint64_t a[16], b[16], r[16];
            getvec(E, w[1], a); getvec(E, w[2], b); getvec(E, w[3], r);
            if (!mul_row(a, b, r, aux, tape + w[4], tape + w[5])) return 1 + (int)n;
        } else if (kind == EC_I_LOOKUP) {
            const uint64_t a = getv(E, w[2]);
            if ((w[1] & 0xFF) == EC_T_XOR8) {
                const uint64_t b = getv(E, w[3]);
                if (a > 255 || b > 255) return 1 + (int)n;
                uint64_t x = 0;
                for (int bit = 0; bit < 8; bit++) x |= (uint64_t)((((a >> bit) & 1) + ((b >> bit) & 1)) & 1) << bit;
                tape[w[4]] = x;
            } else {
                if (a > 255) return 1 + (int)n;
                const uint32_t tb = (w[1] & 0xFF) - EC_T_FIXED0 + 8 * E->inst;
                tape[w[4]] = S->fixed[((size_t)tb * 256 + a) * 2];
                tape[w[4] + 1] = S->fixed[((size_t)tb * 256 + a) * 2 + 1];
            }
        } else if (aux == EC_H_MULSUB || aux == EC_H_DIV) {
            const big m = modulus(w[1]);
            const big a = vec_mod(E, w[2], &m), b = vec_mod(E, w[3], &m);
            big res;
            if (aux == EC_H_DIV) {
                if (big_is_zero(&b)) return 1 + (int)n;
                const big bi = big_invmod(&b, &m);
                res = big_mulmod(&a, &bi, &m);
                big_to_limbs16(&res, tape + w[4], 16);
            } else {
                res = big_mulmod(&a, &b, &m);
                if (w[4] != EC_NONE) { const big cc = vec_mod(E, w[4], &m); res = big_submod(&res, &cc, &m); }
                if (w[5] != EC_NONE) { const big dd = vec_mod(E, w[5], &m); res = big_submod(&res, &dd, &m); }
                big_to_limbs16(&res, tape + w[6], 16);
            }
        } else if (aux == EC_H_SQRT) { /* p = 3 mod 4: y = t^((p + 1) / 4); no root of t -> a root of -t proves it */
            const big m = modulus(0), t = vec_mod(E, w[1], &m), one = big_small(1), zero = big_zero();
            big e = big_add(&m, &one);
            big_shr1(&e); big_shr1(&e);
            big y = big_powmod(&t, &e, &m);
            const big y2 = big_mulmod(&y, &y, &m);
            uint64_t e_nr = 0;
            if (big_cmp(&y2, &t) != 0) {
                e_nr = 1;
                const big nt = big_submod(&zero, &t, &m);
                y = big_powmod(&nt, &e, &m);
            } else if ((y.w[0] & 1) != (getv(E, w[2]) & 1)) {
                y = big_submod(&zero, &y, &m);
            }
            big_to_limbs16(&y, tape + w[3], 16);
            tape[w[3] + 16] = e_nr;
        } else if (aux == EC_H_ISZERO) {
            const uint64_t x = getv(E, w[1]) % EC_GL_P;
            tape[w[2]] = x ? orc_gl_inv(x) : 0;
            tape[w[2] + 1] = x ? 0 : 1;
        } else { /* EC_H_GE: the limb vector >= the constant */
            uint64_t l[16], cst[16];
            getvec(E, w[1], l);
            for (int i = 0; i < 16; i++) cst[i] = S->bigs[w[2] * 16 + (uint32_t)i];
            const big a = big_from_limbs16(l, 16), cbig = big_from_limbs16(cst, 16);
            tape[w[3]] = big_cmp(&a, &cbig) >= 0;
        }
    }
    return 0;
}

/* the whole cycle: 0, or (run << 24 | instance << 12 | 1 + item) of the first item without a witness */
uint32_t orc_ec_eval_cycle_own(const ec_spec *S, const uint8_t *in, uint64_t *tape) {
    ectx E;
    E.S = S; E.tape = tape; E.in = in; E.pbase = 0; E.ptype = 0;
    for (uint32_t r = 0; r < EC_NUM_RUNS; r++) {
        const ec_run *R = &S->runs[r];
        const ec_seg_type *T = &S->types[R->type];
        for (uint32_t j = 0; j < R->count; j++) {
            E.base = R->tape0 + j * T->n_tape;
            E.inst = j;
            const int bad = eval_segment(&E, R->type);
            if (bad) return (r << 24) | (j << 12) | (uint32_t)bad;
            E.pbase = E.base;
            E.ptype = R->type;
        }
    }
    return 0;
}

/* ---- the relation an item states over the cells of its row; 0 when it holds ---------------------------------------------------- */
#define CELL(col) trace[(size_t)(col) * n_rows + row]
int orc_ec_check_item_own(const ec_spec *S, const uint32_t *w, const uint64_t *trace, size_t n_rows, size_t row, uint32_t inst) {
    const uint32_t kind = w[0] & 15, aux = w[0] >> 24, col = (w[0] >> 16) & 0xFF;
    if (kind == EC_I_LIN) { /* sum coef_i cell_i + const == sum 2^shift_j new_j */
        const uint32_t nk = aux, nn = w[1];
        uint64_t lhs = fe_from_i64((int64_t)((uint64_t)w[2] | ((uint64_t)w[3] << 32))), rhs = 0;
        for (uint32_t i = 0; i < nk; i++) {
            const uint64_t x = CELL(col + i);
            if (x >= EC_GL_P) return 1;
            lhs = orc_gl_add(lhs, orc_gl_mul(x, fe_from_i64((int32_t)w[4 + 2 * i + 1])));
        }
        for (uint32_t i = 0; i < nn; i++) {
            const uint64_t x = CELL(col + nk + i);
            if (x >= EC_GL_P) return 1;
            rhs = orc_gl_add(rhs, orc_gl_mul(x, orc_gl_pow(2, w[4 + 2 * nk + 2 * i + 1] & 0xFF)));
        }
        return lhs != rhs;
    }
    if (kind == EC_I_SEL || kind == EC_I_FMA) {
        uint64_t c[4];
        for (int i = 0; i < 4; i++) { c[i] = CELL(col + (uint32_t)i); if (c[i] >= EC_GL_P) return 1; }
        if (kind == EC_I_SEL) return orc_gl_add(orc_gl_mul(c[0], orc_gl_sub(c[1], c[2])), c[2]) != c[3]; /* b (x - y) + y == o */
        return orc_gl_add(orc_gl_mul(c[0], c[1]), c[2]) != c[3];                                          /* a b + c == d */
    }
    if (kind == EC_I_MUL) { /* position k: D_k + c_(k-1) == 2^32 c_k, carries stored + 2^31, c_15 = 0; cell 79 empty */
        const big m = modulus(aux);
        for (uint32_t c = 0; c < 80; c++) if (CELL(c) >= EC_GL_P) return 1;
        const uint64_t two31 = 1ull << 31, two32 = 1ull << 32;
        uint64_t cin = 0;
        for (int k = 0; k < 16; k++) {
            uint64_t d = 0;
            for (int half = 0; half < 2; half++) {
                const int t = 2 * k + half;
                uint64_t s = 0;
                for (int i = 0; i < 16; i++) {
                    const int j = t - i;
                    if (j < 0 || j > 15) continue;
                    s = orc_gl_add(s, orc_gl_mul(CELL((uint32_t)i), CELL(16u + (uint32_t)j)));
                    const uint64_t qi = i == 0 ? orc_gl_sub(CELL(32), EC_KMUL) : CELL(32u + (uint32_t)i);
                    s = orc_gl_sub(s, orc_gl_mul(qi, limb_of_m(&m, j)));
                }
                if (t < 16) s = orc_gl_sub(s, CELL(48u + (uint32_t)t));
                d = orc_gl_add(d, half ? orc_gl_mul(s, 65536) : s);
            }
            const uint64_t cout = k < 15 ? orc_gl_sub(CELL(64u + (uint32_t)k), two31) : 0;
            if (orc_gl_add(d, cin) != orc_gl_mul(cout, two32)) return 1;
            cin = cout;
        }
        return CELL(79) != 0;
    }
    if (kind == EC_I_LOOKUP) { /* membership: (a, b, a xor b) / (byte, word i of x, word i of y) */
        const uint32_t c0 = EC_G + EC_W * col;
        const uint64_t a = CELL(c0), b = CELL(c0 + 1), c = CELL(c0 + 2);
        if ((w[1] & 0xFF) == EC_T_XOR8) {
            if (a > 255 || b > 255) return 1;
            uint64_t x = 0;
            for (int bit = 0; bit < 8; bit++) x |= (uint64_t)((((a >> bit) & 1) + ((b >> bit) & 1)) & 1) << bit;
            return c != x;
        }
        const uint32_t tb = (w[1] & 0xFF) - EC_T_FIXED0 + 8 * inst;
        return a > 255 || b != S->fixed[((size_t)tb * 256 + a) * 2] || c != S->fixed[((size_t)tb * 256 + a) * 2 + 1];
    }
    return 0; /* hints state nothing */
}
