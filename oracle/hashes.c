/* hashes.c — TEST INFRASTRUCTURE: SHA-256 compression, Keccak-f[1600]/Keccak-256 and Blake2s-256,
 * the standard algorithms the reference's round-function builders replay out of circuit
 * (src/witness/individual_circuits/sha256_round_function.rs:231, keccak256_round_function.rs:312-327,
 * decommit_code.rs:320, src/witness/tree/mod.rs:401-424). Pinned by public known-answer vectors in
 * tests/test_oracle_field_hash.py (FIPS 180-4 "abc", Keccak-256(""), RFC 7693 "abc").
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>

/* ---------------- SHA-256 */
static const uint32_t K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
    0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
static inline uint32_t ror32(uint32_t x, int r) { return (x >> r) | (x << (32 - r)); }

const uint32_t ORC_SHA256_IV[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a,
                                   0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};

/* one compression: state (8 x u32) updated with a 64-byte block */
void orc_sha256_compress(uint32_t state[8], const uint8_t block[64]) {
    uint32_t w[64];
    for (int i = 0; i < 16; i++)
        w[i] = ((uint32_t)block[4 * i] << 24) | ((uint32_t)block[4 * i + 1] << 16) |
               ((uint32_t)block[4 * i + 2] << 8) | block[4 * i + 3];
    for (int i = 16; i < 64; i++) {
        uint32_t s0 = ror32(w[i - 15], 7) ^ ror32(w[i - 15], 18) ^ (w[i - 15] >> 3);
        uint32_t s1 = ror32(w[i - 2], 17) ^ ror32(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = state[0], b = state[1], c = state[2], d = state[3], e = state[4], f = state[5],
             g = state[6], h = state[7];
    for (int i = 0; i < 64; i++) {
        uint32_t S1 = ror32(e, 6) ^ ror32(e, 11) ^ ror32(e, 25), ch = (e & f) ^ (~e & g);
        uint32_t t1 = h + S1 + ch + K256[i] + w[i];
        uint32_t S0 = ror32(a, 2) ^ ror32(a, 13) ^ ror32(a, 22), mj = (a & b) ^ (a & c) ^ (b & c);
        uint32_t t2 = S0 + mj;
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    state[0] += a; state[1] += b; state[2] += c; state[3] += d;
    state[4] += e; state[5] += f; state[6] += g; state[7] += h;
}

void orc_sha256(const uint8_t *msg, size_t len, uint8_t out[32]) {
    uint32_t st[8];
    memcpy(st, ORC_SHA256_IV, sizeof st);
    size_t i = 0;
    for (; i + 64 <= len; i += 64) orc_sha256_compress(st, msg + i);
    uint8_t buf[128] = {0};
    size_t rem = len - i;
    memcpy(buf, msg + i, rem);
    buf[rem] = 0x80;
    size_t total = rem + 9 <= 64 ? 64 : 128;
    uint64_t bits = (uint64_t)len * 8;
    for (int k = 0; k < 8; k++) buf[total - 1 - k] = (uint8_t)(bits >> (8 * k));
    orc_sha256_compress(st, buf);
    if (total == 128) orc_sha256_compress(st, buf + 64);
    for (int k = 0; k < 8; k++) {
        out[4 * k] = (uint8_t)(st[k] >> 24); out[4 * k + 1] = (uint8_t)(st[k] >> 16);
        out[4 * k + 2] = (uint8_t)(st[k] >> 8); out[4 * k + 3] = (uint8_t)st[k];
    }
}

/* ---------------- Keccak-f[1600], Keccak-256 (pad 0x01 ... 0x80, rate 136) */
static const uint64_t KRC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
    0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
    0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
    0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
    0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
static const int KROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
static inline uint64_t rol64(uint64_t x, int r) { return r ? (x << r) | (x >> (64 - r)) : x; }

/* state index = x + 5*y */
void orc_keccak_f1600(uint64_t a[25]) {
    for (int round = 0; round < 24; round++) {
        uint64_t c[5], d[5], b[25];
        for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
        for (int x = 0; x < 5; x++) d[x] = c[(x + 4) % 5] ^ rol64(c[(x + 1) % 5], 1);
        for (int i = 0; i < 25; i++) a[i] ^= d[i % 5];
        for (int x = 0; x < 5; x++)
            for (int y = 0; y < 5; y++) b[y + 5 * ((2 * x + 3 * y) % 5)] = rol64(a[x + 5 * y], KROT[x + 5 * y]);
        for (int y = 0; y < 5; y++)
            for (int x = 0; x < 5; x++) a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
        a[0] ^= KRC[round];
    }
}

void orc_keccak256(const uint8_t *msg, size_t len, uint8_t out[32]) {
    uint64_t st[25] = {0};
    uint8_t block[136];
    size_t i = 0;
    for (;;) {
        size_t take = len - i < 136 ? len - i : 136;
        memset(block, 0, 136);
        memcpy(block, msg + i, take);
        int last = take < 136;
        if (last) { block[take] ^= 0x01; block[135] ^= 0x80; }
        for (int k = 0; k < 17; k++) {
            uint64_t lane = 0;
            for (int b = 0; b < 8; b++) lane |= (uint64_t)block[8 * k + b] << (8 * b);
            st[k] ^= lane;
        }
        orc_keccak_f1600(st);
        i += take;
        if (last) break;
    }
    for (int k = 0; k < 4; k++)
        for (int b = 0; b < 8; b++) out[8 * k + b] = (uint8_t)(st[k] >> (8 * b));
}

/* ---------------- Blake2s-256 (unkeyed) */
static const uint32_t B2S_IV[8] = {0x6A09E667, 0xBB67AE85, 0x3C6EF372, 0xA54FF53A,
                                   0x510E527F, 0x9B05688C, 0x1F83D9AB, 0x5BE0CD19};
static const uint8_t B2S_SIGMA[10][16] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
    {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
    {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
    {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
    {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};

static void b2s_compress(uint32_t h[8], const uint8_t block[64], uint64_t t, int last) {
    uint32_t m[16], v[16];
    for (int i = 0; i < 16; i++)
        m[i] = (uint32_t)block[4 * i] | ((uint32_t)block[4 * i + 1] << 8) | ((uint32_t)block[4 * i + 2] << 16) |
               ((uint32_t)block[4 * i + 3] << 24);
    for (int i = 0; i < 8; i++) { v[i] = h[i]; v[i + 8] = B2S_IV[i]; }
    v[12] ^= (uint32_t)t; v[13] ^= (uint32_t)(t >> 32);
    if (last) v[14] = ~v[14];
#define G(a, b, c, d, x, y)                                                                      \
    v[a] = v[a] + v[b] + (x); v[d] = ror32(v[d] ^ v[a], 16); v[c] = v[c] + v[d]; v[b] = ror32(v[b] ^ v[c], 12); \
    v[a] = v[a] + v[b] + (y); v[d] = ror32(v[d] ^ v[a], 8);  v[c] = v[c] + v[d]; v[b] = ror32(v[b] ^ v[c], 7);
    for (int r = 0; r < 10; r++) {
        const uint8_t *s = B2S_SIGMA[r];
        G(0, 4, 8, 12, m[s[0]], m[s[1]]) G(1, 5, 9, 13, m[s[2]], m[s[3]])
        G(2, 6, 10, 14, m[s[4]], m[s[5]]) G(3, 7, 11, 15, m[s[6]], m[s[7]])
        G(0, 5, 10, 15, m[s[8]], m[s[9]]) G(1, 6, 11, 12, m[s[10]], m[s[11]])
        G(2, 7, 8, 13, m[s[12]], m[s[13]]) G(3, 4, 9, 14, m[s[14]], m[s[15]])
    }
#undef G
    for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[i + 8];
}

void orc_blake2s256(const uint8_t *msg, size_t len, uint8_t out[32]) {
    uint32_t h[8];
    memcpy(h, B2S_IV, sizeof h);
    h[0] ^= 0x01010000u ^ 32u;
    uint8_t block[64];
    size_t i = 0;
    while (len - i > 64) { b2s_compress(h, msg + i, (uint64_t)i + 64, 0); i += 64; }
    memset(block, 0, 64);
    memcpy(block, msg + i, len - i);
    b2s_compress(h, block, (uint64_t)len, 1);
    for (int k = 0; k < 8; k++) {
        out[4 * k] = (uint8_t)h[k]; out[4 * k + 1] = (uint8_t)(h[k] >> 8);
        out[4 * k + 2] = (uint8_t)(h[k] >> 16); out[4 * k + 3] = (uint8_t)(h[k] >> 24);
    }
}
