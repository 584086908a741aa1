/* oracle.c — TEST INFRASTRUCTURE: CPU restatement of the reference hot path (see oracle.h).
 * Written for obviousness; the field ops use the 2^64 = 2^32 - 1 reduction (checked against one-`%` reference forms
 * in the tests) so that the cpu_baseline leg of bench.py times a competent scalar CPU path.
 */
#include "oracle.h"
#include "../include/zkw_poseidon2_params.h"
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
#define P ZKW_GOLDILOCKS_P

/* ------------------------------------------------------------------ field */
/* Reference forms (one `%` on unsigned __int128 each): what the fast forms below are checked against in
 * tests/test_oracle_field_hash.py. */
uint64_t orc_gl_add_ref(uint64_t a, uint64_t b) { return (uint64_t)(((u128)a + b) % P); }
uint64_t orc_gl_sub_ref(uint64_t a, uint64_t b) { return (uint64_t)(((u128)(a % P) + P - (b % P)) % P); }
uint64_t orc_gl_mul_ref(uint64_t a, uint64_t b) { return (uint64_t)(((u128)a * b) % P); }

/* Fast forms (boojum's GoldilocksField does the same: 2^64 = 2^32 - 1 and 2^96 = -1 mod p). Any u64 in, canonical out.
 * Defined inline in oracle.h; these are the exported symbols (ctypes, tests). */
#undef orc_gl_reduce128
#undef orc_gl_add
#undef orc_gl_sub
#undef orc_gl_mul
uint64_t orc_gl_reduce128(u128 w) { return orc_gl_reduce128_inl(w); }
uint64_t orc_gl_add(uint64_t a, uint64_t b) { return orc_gl_add_inl(a, b); }
uint64_t orc_gl_sub(uint64_t a, uint64_t b) { return orc_gl_sub_inl(a, b); }
uint64_t orc_gl_mul(uint64_t a, uint64_t b) { return orc_gl_mul_inl(a, b); }
#define orc_gl_reduce128 orc_gl_reduce128_inl
#define orc_gl_add orc_gl_add_inl
#define orc_gl_sub orc_gl_sub_inl
#define orc_gl_mul orc_gl_mul_inl
uint64_t orc_gl_pow(uint64_t a, uint64_t e) {
    uint64_t r = 1, b = a % P;
    while (e) {
        if (e & 1) r = orc_gl_mul(r, b);
        b = orc_gl_mul(b, b);
        e >>= 1;
    }
    return r;
}
uint64_t orc_gl_inv(uint64_t a) { return orc_gl_pow(a, P - 2); }

/* ------------------------------------------------------------------ Poseidon2 */
static uint64_t sbox7(uint64_t x) {
    uint64_t x2 = orc_gl_mul(x, x), x3 = orc_gl_mul(x2, x), x4 = orc_gl_mul(x2, x2);
    return orc_gl_mul(x3, x4);
}

/* external linear layer: circ(2*M4, M4, M4), M4 = [[5,7,1,3],[4,6,1,1],[1,3,5,7],[1,1,4,6]] */
static void p2_external(uint64_t s[12]) {
    static const uint64_t M4[4][4] = {{5, 7, 1, 3}, {4, 6, 1, 1}, {1, 3, 5, 7}, {1, 1, 4, 6}};
    uint64_t t[12];
    for (int c = 0; c < 3; c++)
        for (int i = 0; i < 4; i++) {
            u128 acc = 0;
            for (int j = 0; j < 4; j++) acc += (u128)M4[i][j] * s[4 * c + j];
            t[4 * c + i] = orc_gl_reduce128(acc);
        }
    for (int i = 0; i < 4; i++) {
        uint64_t col = orc_gl_add(orc_gl_add(t[i], t[4 + i]), t[8 + i]);
        for (int c = 0; c < 3; c++) s[4 * c + i] = orc_gl_add(t[4 * c + i], col);
    }
}

/* internal linear layer: y_i = x_i * 2^shift_i + sum_j x_j */
static void p2_internal(uint64_t s[12]) {
    uint64_t sum = 0;
    for (int i = 0; i < 12; i++) sum = orc_gl_add(sum, s[i]);
    for (int i = 0; i < 12; i++)
        s[i] = orc_gl_add(orc_gl_mul(s[i], 1ULL << P2_INTERNAL_DIAG_SHIFTS[i]), sum);
}

/* the obvious form: what the fast form below is tested against (tests/test_oracle_field_hash.py) */
void orc_poseidon2_permutation_ref(uint64_t s[12]) {
    int r = 0;
    p2_external(s);
    for (int k = 0; k < P2_HALF_FULL_ROUNDS; k++, r++) {
        for (int i = 0; i < 12; i++) s[i] = sbox7(orc_gl_add(s[i], P2_ROUND_CONSTANTS[12 * r + i]));
        p2_external(s);
    }
    for (int k = 0; k < P2_PARTIAL_ROUNDS; k++, r++) {
        s[0] = sbox7(orc_gl_add(s[0], P2_ROUND_CONSTANTS[12 * r]));
        p2_internal(s);
    }
    for (int k = 0; k < P2_HALF_FULL_ROUNDS; k++, r++) {
        for (int i = 0; i < 12; i++) s[i] = sbox7(orc_gl_add(s[i], P2_ROUND_CONSTANTS[12 * r + i]));
        p2_external(s);
    }
}

/* The same permutation written the way a competent scalar CPU implementation does it (boojum's generic
 * `State`: unreduced u128 sums in the linear layers, one reduction per output, branch-free reductions, values kept
 * in "weak" form = any u64 congruent to the element, canonicalised once at the end). This is what every oracle
 * routine calls and what bench.py's cpu_baseline leg times. */
static inline uint64_t w_red(u128 w) { /* weak result */
    const uint64_t lo = (uint64_t)w, hi = (uint64_t)(w >> 64), hh = hi >> 32, hl = hi & ORC_GL_EPS;
    uint64_t t0, r;
    const uint64_t bo = __builtin_sub_overflow(lo, hh, &t0);
    t0 -= (0 - bo) & ORC_GL_EPS;
    const uint64_t t1 = (hl << 32) - hl;
    const uint64_t ca = __builtin_add_overflow(t0, t1, &r);
    return r + ((0 - ca) & ORC_GL_EPS);
}
static inline uint64_t w_mul(uint64_t a, uint64_t b) { return w_red((u128)a * b); }
static inline uint64_t w_sbox(uint64_t x) {
    const uint64_t x2 = w_mul(x, x), x3 = w_mul(x2, x), x4 = w_mul(x2, x2);
    return w_mul(x3, x4);
}
static inline void w_external(uint64_t s[12]) {
    u128 t[12];
    for (int c = 0; c < 3; c++) { /* the Poseidon2 addition chain for M4, on unreduced sums (< 2^68) */
        const u128 x0 = s[4 * c], x1 = s[4 * c + 1], x2 = s[4 * c + 2], x3 = s[4 * c + 3];
        const u128 t0 = x0 + x1, t1 = x2 + x3, t2 = 2 * x1 + t1, t3 = 2 * x3 + t0, t4 = 4 * t1 + t3, t5 = 4 * t0 + t2;
        t[4 * c] = t3 + t5; t[4 * c + 1] = t5; t[4 * c + 2] = t2 + t4; t[4 * c + 3] = t4;
    }
    for (int i = 0; i < 4; i++) {
        const u128 col = t[i] + t[4 + i] + t[8 + i];
        for (int c = 0; c < 3; c++) s[4 * c + i] = w_red(t[4 * c + i] + col);
    }
}
static inline void w_internal(uint64_t s[12]) {
    u128 sum = 0;
    for (int i = 0; i < 12; i++) sum += s[i];
    for (int i = 0; i < 12; i++) s[i] = w_red(((u128)s[i] << P2_INTERNAL_DIAG_SHIFTS[i]) + sum);
}
static inline uint64_t w_add(uint64_t a, uint64_t b) { /* weak + canonical constant */
    uint64_t r;
    const uint64_t ca = __builtin_add_overflow(a, b, &r);
    uint64_t r2;
    const uint64_t cb = __builtin_add_overflow(r, (0 - ca) & ORC_GL_EPS, &r2);
    return r2 + ((0 - cb) & ORC_GL_EPS);
}
void orc_poseidon2_permutation(uint64_t s[12]) {
    int r = 0;
    w_external(s);
    for (int k = 0; k < P2_HALF_FULL_ROUNDS; k++, r++) {
        for (int i = 0; i < 12; i++) s[i] = w_sbox(w_add(s[i], P2_ROUND_CONSTANTS[12 * r + i]));
        w_external(s);
    }
    for (int k = 0; k < P2_PARTIAL_ROUNDS; k++, r++) {
        s[0] = w_sbox(w_add(s[0], P2_ROUND_CONSTANTS[12 * r]));
        w_internal(s);
    }
    for (int k = 0; k < P2_HALF_FULL_ROUNDS; k++, r++) {
        for (int i = 0; i < 12; i++) s[i] = w_sbox(w_add(s[i], P2_ROUND_CONSTANTS[12 * r + i]));
        w_external(s);
    }
    for (int i = 0; i < 12; i++) s[i] = s[i] >= P ? s[i] - P : s[i];
}

void orc_absorb_multiple_rounds(uint64_t state[12], const uint64_t *to_absorb, size_t n_rounds,
                                uint64_t *states_out) {
    for (size_t r = 0; r < n_rounds; r++) {
        for (int i = 0; i < 8; i++) state[i] = to_absorb[8 * r + i] % P; /* AbsorptionModeOverwrite */
        orc_poseidon2_permutation(state);
        if (states_out) memcpy(states_out + 12 * r, state, 12 * sizeof(uint64_t));
    }
}

void orc_poseidon2_hash_node(const uint64_t left[4], const uint64_t right[4], uint64_t out[4]) {
    uint64_t s[12] = {0};
    memcpy(s, left, 32);
    memcpy(s + 4, right, 32);
    orc_poseidon2_permutation(s);
    memcpy(out, s, 32);
}

void orc_poseidon2_hash_leaf(const uint64_t *elems, size_t n, uint64_t out[4]) {
    uint64_t s[12] = {0};
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        memcpy(s, elems + i, 64);
        orc_poseidon2_permutation(s);
    }
    if (i < n) {
        uint64_t buf[8] = {0};
        memcpy(buf, elems + i, (n - i) * 8);
        memcpy(s, buf, 64);
        orc_poseidon2_permutation(s);
    }
    memcpy(out, s, 32);
}

/* ------------------------------------------------------------------ encodings */
/* memory_query.rs:24-118. linear_combination(&[(x, c)...]) = sum x*c in the field; every term here
   is < 2^56 so no reduction is ever triggered, but we go through the field ops anyway. */
void orc_encode_memory_query(const zkw_mem_query *q, uint64_t out[8]) {
    const uint32_t *v = q->value;
    uint8_t d5[4], d6[4], d7[4];
    for (int i = 0; i < 4; i++) {
        d5[i] = (uint8_t)(v[5] >> (8 * i)); /* to_le_bytes */
        d6[i] = (uint8_t)(v[6] >> (8 * i));
        d7[i] = (uint8_t)(v[7] >> (8 * i));
    }
    const uint64_t S32 = 1ULL << 32, S33 = 1ULL << 33, S40 = 1ULL << 40, S48 = 1ULL << 48;
#define LC2(a, b, cb, c, cc, d, cd)                                                              \
    orc_gl_add(orc_gl_add(orc_gl_add((a), orc_gl_mul((b), (cb))), orc_gl_mul((c), (cc))),         \
               orc_gl_mul((d), (cd)))
    out[0] = q->timestamp;
    out[1] = q->page;
    out[2] = orc_gl_add(orc_gl_add(q->index, orc_gl_mul(q->rw_flag ? 1 : 0, S32)),
                        orc_gl_mul(q->value_is_pointer ? 1 : 0, S33));
    out[3] = LC2(v[0], d5[0], S32, d5[1], S40, d5[2], S48);
    out[4] = LC2(v[1], d5[3], S32, d6[0], S40, d6[1], S48);
    out[5] = LC2(v[2], d6[2], S32, d6[3], S40, d7[0], S48);
    out[6] = LC2(v[3], d7[1], S32, d7[2], S40, d7[3], S48);
    out[7] = v[4];
#undef LC2
}

void orc_encode_memory_queries(const zkw_mem_query *q, size_t n, uint64_t *out) {
    for (size_t i = 0; i < n; i++) orc_encode_memory_query(q + i, out + 8 * i);
}

/* log_query.rs:102-396 */
static void encode_log_query(const zkw_log_query *q, uint32_t ext_ts, int has_ext, uint64_t out[20]) {
    uint8_t key_bytes[32], address_bytes[20];
    for (int i = 0; i < 32; i++) key_bytes[i] = (uint8_t)(q->key[i / 4] >> (8 * (i % 4)));       /* to_little_endian */
    for (int i = 0; i < 20; i++) address_bytes[i] = (uint8_t)(q->address[i / 4] >> (8 * (i % 4))); /* H160 bytes reversed */
    const uint64_t S32 = 1ULL << 32, S40 = 1ULL << 40, S48 = 1ULL << 48;
    uint8_t tail_bytes[60]; /* the 3-byte riders of v0..v16 in order: key[0..32] then address[0..19] */
    memcpy(tail_bytes, key_bytes, 32);
    memcpy(tail_bytes + 32, address_bytes, 20);
    for (int k = 0; k < 17; k++) {
        uint64_t base = k < 8 ? q->read_value[k] : (k < 16 ? q->written_value[k - 8] : q->timestamp);
        uint64_t v = base;
        v = orc_gl_add(v, orc_gl_mul(tail_bytes[3 * k], S32));
        v = orc_gl_add(v, orc_gl_mul(tail_bytes[3 * k + 1], S40));
        v = orc_gl_add(v, orc_gl_mul(tail_bytes[3 * k + 2], S48));
        out[k] = v;
    }
    /* v17 = tx_number + address_bytes[19]<<32 + aux_byte<<40 + shard_id<<48 */
    uint64_t v17 = q->tx_number_in_block;
    v17 = orc_gl_add(v17, orc_gl_mul(address_bytes[19], S32));
    v17 = orc_gl_add(v17, orc_gl_mul(q->aux_byte, S40));
    v17 = orc_gl_add(v17, orc_gl_mul(q->shard_id, S48));
    out[17] = v17;
    out[18] = (uint64_t)(q->rw_flag ? 1 : 0) + 2 * (uint64_t)(q->is_service ? 1 : 0);
    out[19] = q->rollback ? 1 : 0;
    if (has_ext) /* scale_and_accumulate, log_query.rs:414-421 */
        out[ZKW_EXTENDED_TIMESTAMP_ENCODING_ELEMENT] =
            orc_gl_add(out[ZKW_EXTENDED_TIMESTAMP_ENCODING_ELEMENT],
                       orc_gl_mul(ext_ts, 1ULL << ZKW_EXTENDED_TIMESTAMP_ENCODING_OFFSET));
}

void orc_encode_log_queries(const zkw_log_query *q, size_t n, const uint32_t *ext_ts, uint64_t *out) {
    for (size_t i = 0; i < n; i++) encode_log_query(q + i, ext_ts ? ext_ts[i] : 0, ext_ts != NULL, out + 20 * i);
}

/* decommittment_request.rs:9-74 */
void orc_encode_decommit_queries(const zkw_decommit_query *q, size_t n, uint64_t *out) {
    const uint64_t S32 = 1ULL << 32, S40 = 1ULL << 40, S48 = 1ULL << 48;
    for (size_t i = 0; i < n; i++) {
        const zkw_decommit_query *d = q + i;
        uint8_t pb[4], tb[4];
        for (int k = 0; k < 4; k++) { pb[k] = (uint8_t)(d->memory_page >> (8 * k)); tb[k] = (uint8_t)(d->timestamp >> (8 * k)); }
        uint64_t *o = out + 8 * i;
        o[0] = orc_gl_add(orc_gl_add(orc_gl_add(d->hash[0], orc_gl_mul(pb[0], S32)), orc_gl_mul(pb[1], S40)), orc_gl_mul(pb[2], S48));
        o[1] = orc_gl_add(orc_gl_add(orc_gl_add(d->hash[1], orc_gl_mul(pb[3], S32)), orc_gl_mul(tb[0], S40)), orc_gl_mul(tb[1], S48));
        o[2] = orc_gl_add(orc_gl_add(orc_gl_add(d->hash[2], orc_gl_mul(tb[2], S32)), orc_gl_mul(tb[3], S40)), orc_gl_mul(d->is_fresh ? 1 : 0, S48));
        for (int k = 3; k < 8; k++) o[k] = d->hash[k];
    }
}

void orc_encode_recursion_request(uint64_t circuit_type, const uint64_t pi[4], uint64_t out[8]) {
    out[0] = circuit_type % P;
    for (int k = 0; k < 4; k++) out[1 + k] = pi[k] % P;
    out[5] = out[6] = out[7] = 0;
}

/* ------------------------------------------------------------------ queues */
void orc_queue_push_chain_full(const uint64_t *enc, size_t n, const uint64_t tail_in[12], uint64_t *tails) {
    uint64_t state[12];
    memcpy(state, tail_in, sizeof state);
    for (size_t i = 0; i < n; i++) {
        orc_absorb_multiple_rounds(state, enc + 8 * i, 1, NULL); /* lib.rs:404-409, ROUNDS = 1 */
        memcpy(tails + 12 * i, state, sizeof state);             /* lib.rs:415 stores the NEW tail */
    }
}

void orc_queue_push_chain_log(const uint64_t *enc, size_t n, const uint64_t tail_in[4],
                              uint64_t *old_tails, uint64_t *new_tails) {
    uint64_t tail[4];
    memcpy(tail, tail_in, sizeof tail);
    for (size_t i = 0; i < n; i++) {
        uint64_t to_hash[24], state[12] = {0}; /* R::initial_state() = zeros, lib.rs:196 */
        memcpy(to_hash, enc + 20 * i, 20 * 8);
        memcpy(to_hash + 20, tail, 32); /* lib.rs:192-194 */
        if (old_tails) memcpy(old_tails + 4 * i, tail, 32);
        orc_absorb_multiple_rounds(state, to_hash, 3, NULL);
        memcpy(tail, state, 32); /* state_into_commitment::<4>, lib.rs:200-201 */
        memcpy(new_tails + 4 * i, tail, 32);
    }
}

/* ------------------------------------------------------------------ FS challenges (utils.rs:498-550) */
void orc_fs_challenges(const uint64_t *tail_u, uint32_t len_u, const uint64_t *tail_s, uint32_t len_s,
                       int state_w, int n_chal, uint64_t *out) {
    uint64_t fs_input[2 * 12 + 2];
    int m = 0;
    for (int i = 0; i < state_w; i++) fs_input[m++] = tail_u[i];
    fs_input[m++] = (uint64_t)len_u % P; /* from_u64_with_reduction */
    for (int i = 0; i < state_w; i++) fs_input[m++] = tail_s[i];
    fs_input[m++] = (uint64_t)len_s % P;

    uint64_t state[12] = {0};
    state[11] = (uint64_t)m; /* Poseidon2Goldilocks::specialize_for_len: length in the last element */
    int i = 0;
    for (; i + 8 <= m; i += 8) orc_absorb_multiple_rounds(state, fs_input + i, 1, NULL);
    if (i < m) {
        uint64_t padded[8] = {0};
        memcpy(padded, fs_input + i, (size_t)(m - i) * 8);
        orc_absorb_multiple_rounds(state, padded, 1, NULL);
    }
    int can_take = 8;
    for (int rep = 0; rep < 2; rep++) {
        out[rep * n_chal] = 1; /* F::ONE, utils.rs:533 */
        for (int k = 1; k < n_chal; k++) {
            if (can_take == 0) {
                orc_poseidon2_permutation(state);
                can_take = 8;
            }
            out[rep * n_chal + k] = state[8 - can_take];
            can_take--;
        }
    }
}

/* ------------------------------------------------------------------ grand products (utils.rs:554-697) */
#define GP_CHUNK ((size_t)1 << 16) /* PARALLELIZATION_CHUNK_SIZE, utils.rs:552 */

static void gp_chunk_local(const uint64_t *src, size_t n, int width, const uint64_t *ch, uint64_t *dst) {
    uint64_t gp = 1;
    for (size_t i = 0; i < n; i++) {
        uint64_t acc = ch[width] % P;
        for (int j = 0; j < width; j++) acc = orc_gl_add(acc, orc_gl_mul(src[i * width + j], ch[j]));
        gp = orc_gl_mul(gp, acc);
        dst[i] = gp;
    }
}

static void gp_fold(uint64_t *z, size_t n) {
    size_t n_chunks = (n + GP_CHUNK - 1) / GP_CHUNK;
    uint64_t acc = 1;
    for (size_t c = 0; c < n_chunks; c++) {
        size_t lo = c * GP_CHUNK, hi = lo + GP_CHUNK < n ? lo + GP_CHUNK : n;
        uint64_t last = z[hi - 1];
        if (c > 0)
            for (size_t i = lo; i < hi; i++) z[i] = orc_gl_mul(z[i], acc);
        acc = orc_gl_mul(acc, last);
    }
}

int orc_grand_product_chains(const uint64_t *lhs, const uint64_t *rhs, size_t n, int width,
                             const uint64_t *ch, uint64_t *lhs_z, uint64_t *rhs_z) {
    for (size_t lo = 0; lo < n; lo += GP_CHUNK) {
        size_t len = lo + GP_CHUNK < n ? GP_CHUNK : n - lo;
        gp_chunk_local(lhs + lo * width, len, width, ch, lhs_z + lo);
        gp_chunk_local(rhs + lo * width, len, width, ch, rhs_z + lo);
    }
    if (n == 0) return 0;
    gp_fold(lhs_z, n);
    gp_fold(rhs_z, n);
    return lhs_z[n - 1] == rhs_z[n - 1] ? 0 : -1;
}

typedef struct {
    const uint64_t *lhs, *rhs, *ch;
    uint64_t *lz, *rz;
    size_t n;
    int width, tid, nthreads;
} gp_job;

static void *gp_worker(void *p) {
    gp_job *j = (gp_job *)p;
    size_t n_chunks = (j->n + GP_CHUNK - 1) / GP_CHUNK;
    for (size_t c = (size_t)j->tid; c < 2 * n_chunks; c += (size_t)j->nthreads) {
        size_t cc = c % n_chunks, lo = cc * GP_CHUNK, len = lo + GP_CHUNK < j->n ? GP_CHUNK : j->n - lo;
        if (c < n_chunks)
            gp_chunk_local(j->lhs + lo * j->width, len, j->width, j->ch, j->lz + lo);
        else
            gp_chunk_local(j->rhs + lo * j->width, len, j->width, j->ch, j->rz + lo);
    }
    return NULL;
}

int orc_grand_product_chains_mt(const uint64_t *lhs, const uint64_t *rhs, size_t n, int width,
                                const uint64_t *ch, uint64_t *lhs_z, uint64_t *rhs_z, int threads) {
    if (threads < 1) threads = 1;
    if (threads > 64) threads = 64;
    pthread_t th[64];
    gp_job jobs[64];
    for (int t = 0; t < threads; t++) {
        jobs[t] = (gp_job){lhs, rhs, ch, lhs_z, rhs_z, n, width, t, threads};
        pthread_create(&th[t], NULL, gp_worker, &jobs[t]);
    }
    for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    if (n == 0) return 0;
    gp_fold(lhs_z, n);
    gp_fold(rhs_z, n);
    return lhs_z[n - 1] == rhs_z[n - 1] ? 0 : -1;
}

/* ------------------------------------------------------------------ RAM permutation builder */
typedef struct {
    uint32_t page, index, ts;
    size_t orig;
} ram_sort_key;

/* ram_permutation.rs:50-53: location (page, then index — zk_evm's MemoryLocation::cmp), then
   timestamp; par_sort_by is stable, so the original position breaks ties. */
static int ram_key_cmp(const void *a, const void *b) {
    const ram_sort_key *x = (const ram_sort_key *)a, *y = (const ram_sort_key *)b;
    if (x->page != y->page) return x->page < y->page ? -1 : 1;
    if (x->index != y->index) return x->index < y->index ? -1 : 1;
    if (x->ts != y->ts) return x->ts < y->ts ? -1 : 1;
    return x->orig < y->orig ? -1 : (x->orig > y->orig ? 1 : 0);
}

static void qs_placeholder(zkw_queue_state12 *s) { memset(s, 0, sizeof *s); }

int64_t orc_ram_build_instances(const zkw_mem_query *q, size_t n, uint32_t capacity,
                                uint32_t num_non_deterministic_heap_queries, zkw_mem_query *sorted_q,
                                uint64_t *unsorted_enc, uint64_t *sorted_enc, uint64_t *unsorted_tails,
                                uint64_t *sorted_tails, uint64_t *challenges, uint64_t *lhs_z,
                                uint64_t *rhs_z, zkw_ram_instance *instances) {
    if (n == 0 || capacity == 0) return -2; /* ram_permutation.rs:43-46 asserts non-empty */
    /* sort, ram_permutation.rs:48-53 */
    ram_sort_key *keys = (ram_sort_key *)malloc(n * sizeof *keys);
    for (size_t i = 0; i < n; i++) keys[i] = (ram_sort_key){q[i].page, q[i].index, q[i].timestamp, i};
    qsort(keys, n, sizeof *keys, ram_key_cmp);
    for (size_t i = 0; i < n; i++) sorted_q[i] = q[keys[i].orig];
    free(keys);

    /* unsorted chain (src/witness/oracle.rs:894-903) and sorted chain (ram_permutation.rs:59-71) */
    const uint64_t zero12[12] = {0};
    orc_encode_memory_queries(q, n, unsorted_enc);
    orc_encode_memory_queries(sorted_q, n, sorted_enc);
    orc_queue_push_chain_full(unsorted_enc, n, zero12, unsorted_tails);
    orc_queue_push_chain_full(sorted_enc, n, zero12, sorted_tails);

    /* challenges, ram_permutation.rs:80-90: N = 12, MEMORY_QUERY_PACKED_WIDTH + 1 = 9, 2 repetitions */
    const uint64_t *u_final = unsorted_tails + 12 * (n - 1), *s_final = sorted_tails + 12 * (n - 1);
    orc_fs_challenges(u_final, (uint32_t)n, s_final, (uint32_t)n, 12, 9, challenges);

    /* grand products, ram_permutation.rs:115-138 */
    for (int rep = 0; rep < 2; rep++)
        if (orc_grand_product_chains(unsorted_enc, sorted_enc, n, 8, challenges + 9 * rep, lhs_z + rep * n,
                                     rhs_z + rep * n) != 0)
            return -3;

    /* instances, ram_permutation.rs:239-453 */
    size_t num_circuits = (n + capacity - 1) / capacity;
    uint64_t cur_lhs[2] = {1, 1}, cur_rhs[2] = {1, 1};
    uint32_t prev_sorting_key[3] = {0}, prev_full_key[2] = {0}, prev_value[8] = {0}, prev_is_ptr = 0;
    uint32_t cur_nondet = 0;
    zkw_queue_state12 last_u, last_s;
    qs_placeholder(&last_u);
    qs_placeholder(&last_s);

    for (size_t idx = 0; idx < num_circuits; idx++) {
        size_t lo = idx * capacity, hi = lo + capacity < n ? lo + capacity : n, last = hi - 1;
        zkw_ram_instance *w = instances + idx;
        memset(w, 0, sizeof *w);
        w->start_flag = idx == 0;
        w->completion_flag = idx == num_circuits - 1;
        w->first_item = lo;
        w->num_items = hi - lo;

        uint32_t nondet_in_chunk = 0; /* ram_permutation.rs:309-317 (no is_ptr test out of circuit) */
        for (size_t i = lo; i < hi; i++)
            if (sorted_q[i].rw_flag && sorted_q[i].timestamp == 0 && sorted_q[i].page == ZKW_BOOTLOADER_HEAP_PAGE)
                nondet_in_chunk++;
        uint32_t new_nondet = cur_nondet + nondet_in_chunk;

        /* observable input: global final states (head = 0: the block-wide simulators never pop) */
        memset(w->unsorted_queue_initial_state.head, 0, 96);
        memcpy(w->unsorted_queue_initial_state.tail, u_final, 96);
        w->unsorted_queue_initial_state.length = (uint32_t)n;
        memset(w->sorted_queue_initial_state.head, 0, 96);
        memcpy(w->sorted_queue_initial_state.tail, s_final, 96);
        w->sorted_queue_initial_state.length = (uint32_t)n;
        w->non_deterministic_bootloader_memory_snapshot_length = num_non_deterministic_heap_queries;

        zkw_ram_fsm *fi = &w->hidden_fsm_input, *fo = &w->hidden_fsm_output;
        memcpy(fi->lhs_accumulator, cur_lhs, 16);
        memcpy(fi->rhs_accumulator, cur_rhs, 16);
        fi->current_unsorted_queue_state = last_u;
        fi->current_sorted_queue_state = last_s;
        memcpy(fi->previous_sorting_key, prev_sorting_key, 12);
        memcpy(fi->previous_full_key, prev_full_key, 8);
        memcpy(fi->previous_value, prev_value, 32);
        fi->previous_is_ptr = prev_is_ptr;
        fi->num_nondeterministic_writes = cur_nondet;

        for (int rep = 0; rep < 2; rep++) {
            fo->lhs_accumulator[rep] = lhs_z[rep * n + last];
            fo->rhs_accumulator[rep] = rhs_z[rep * n + last];
        }
        /* ram_permutation.rs:355-365: head := tail after this chunk, tail := global final tail,
           length := remaining items */
        memcpy(fo->current_unsorted_queue_state.head, unsorted_tails + 12 * last, 96);
        memcpy(fo->current_unsorted_queue_state.tail, u_final, 96);
        fo->current_unsorted_queue_state.length = (uint32_t)(n - hi);
        memcpy(fo->current_sorted_queue_state.head, sorted_tails + 12 * last, 96);
        memcpy(fo->current_sorted_queue_state.tail, s_final, 96);
        fo->current_sorted_queue_state.length = (uint32_t)(n - hi);

        const zkw_mem_query *lq = sorted_q + last;
        uint32_t sk[3] = {lq->timestamp, lq->index, lq->page}, fk[2] = {lq->index, lq->page};
        memcpy(fo->previous_sorting_key, sk, 12);
        memcpy(fo->previous_full_key, fk, 8);
        memcpy(fo->previous_value, lq->value, 32);
        fo->previous_is_ptr = lq->value_is_pointer ? 1 : 0;
        fo->num_nondeterministic_writes = new_nondet;

        if ((hi - lo) % capacity != 0) { /* padding reset, ram_permutation.rs:414-432 */
            memset(fo->previous_sorting_key, 0, 12);
            memset(fo->previous_full_key, 0, 8);
            memset(fo->previous_value, 0, 32);
            fo->previous_is_ptr = 0;
        }

        memcpy(cur_lhs, fo->lhs_accumulator, 16);
        memcpy(cur_rhs, fo->rhs_accumulator, 16);
        memcpy(prev_sorting_key, sk, 12); /* ram_permutation.rs:437-440 keep the un-reset values */
        memcpy(prev_full_key, fk, 8);
        memcpy(prev_value, lq->value, 32);
        prev_is_ptr = lq->value_is_pointer ? 1 : 0;
        cur_nondet = new_nondet;
        last_u = fo->current_unsorted_queue_state;
        last_s = fo->current_sorted_queue_state;
    }
    return (int64_t)num_circuits;
}
