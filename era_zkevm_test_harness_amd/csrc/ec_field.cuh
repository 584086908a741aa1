// ec_field.cuh — the base field of secp256k1 (p = 2^256 - 2^32 - 977) for the accumulator chain of the ECRecover circuit's EC section
// (ecrecover_kernels.cuh k_ec_chain / k_ec_affine), two forms of the same arithmetic, canonical in and canonical out:
//   ecf   a value is eight 32-bit words in eight registers of ONE lane (k_ec_affine: a lane per point);
//   ecl   a value is ONE register, limb i in lane i of a 16-lane row (k_ec_chain: a wave per request). A wave instruction costs its cycles
//         whatever the number of active lanes, so the 64 partial products of a multiplication are 8 v_mad_u64_u32 on 16 lanes instead of 64
//         on one, the words move between lanes by DPP row shifts, and carries are settled by a carry-lookahead over the lanes' ballot.
// Depends only on include/zkw_ecrecover.h (ec_u256) so that tests/csrc_gpu/ec_field_test.hip can run both forms against the host arithmetic.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/zkw_ecrecover.h"

namespace zkw {
typedef uint32_t u32;
typedef uint64_t u64;

// ---- the base field of secp256k1 for the accumulator chain: OUTLINED multiplication ------------------------------------------------
// The chain below is ~6 500 multiplications mod p = 2^256 - 2^32 - 977 on ONE lane per request. With ec_mulmod inlined everywhere
// (include/zkw_ecrecover.h: forced, because its operands travel as pointers) k_ec_chain was 62 000 instructions of straight-line code,
// 22 000 of them register moves, far beyond the instruction cache: 17 ms per call whatever the batch (ECRecover at 0.003 of
// HBM). Here the multiplication is ONE function of ~260 instructions that every call site CALLS: operands and result by value (eight
// VGPRs each, no stack, no scratch), product scanning (a column at a time: one v_mad_u64_u32 + one add-with-carry per partial product
// into a 96-bit accumulator), the fold hi * (2^32 + 977) + lo word by word through the same multiplier, one conditional subtraction.
// Canonical in, canonical out, like ec_mulmod.
namespace ecf {
__device__ __forceinline__ void mac(u64& acc, u32& ext, u32 a, u32 b) {  // (ext : acc) += a * b
    u64 c, dead;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(c) : "v"(a), "v"(b));
    asm("v_addc_co_u32_e64 %0, %1, %0, 0, %2" : "+v"(ext), "=s"(dead) : "s"(c));
}
__device__ __forceinline__ u64 mad32(u32 a, u32 b, u64 acc) {  // a * b + acc (no overflow by construction)
    u64 r, dead;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(r), "=s"(dead) : "v"(a), "v"(b), "v"(acc));
    return r;
}
// x (8 words) + carry-in chain helpers are written with 64-bit sums: the compiler keeps the carries in the adds
__device__ __forceinline__ ec_u256 cond_sub_p(const u32* r, u32 top) {  // r + top * 2^256 < 2 p  ->  mod p
    // r >= p  <=>  r + (2^32 + 977) carries out of 256 bits
    u32 u[8];
    u64 c = (u64)r[0] + 977u;
    u[0] = (u32)c; c >>= 32;
    c += (u64)r[1] + 1u;
    u[1] = (u32)c; c >>= 32;
#pragma unroll
    for (int i = 2; i < 8; i++) { c += r[i]; u[i] = (u32)c; c >>= 32; }
    const bool ge = top != 0 || c != 0;
    ec_u256 o;
#pragma unroll
    for (int i = 0; i < 8; i++) o.w[i] = ge ? u[i] : r[i];
    return o;
}
__device__ __attribute__((noinline)) ec_u256 mul(ec_u256 a, ec_u256 b) {
    // product scanning: one 96-bit accumulator walks the columns. (A form with one accumulator per column — eight independent chains in
    // flight — was measured too: 15.0 instead of 13.0 ms per call; a lone wave issues a v_mad_u64_u32 every ~8 cycles whether it depends
    // on the previous one or not (profiles/r05/valu_ceiling.json, one wave per SIMD), so the extra moves of that form only cost.)
    u32 t[16];
    u64 acc = 0;
    u32 ext = 0;
#pragma unroll
    for (int k = 0; k < 15; k++) {
#pragma unroll
        for (int i = 0; i < 8; i++)
            if (k - i >= 0 && k - i < 8) mac(acc, ext, a.w[i], b.w[k - i]);
        t[k] = (u32)acc;
        acc = (acc >> 32) | ((u64)ext << 32);
        ext = 0;
    }
    t[15] = (u32)acc;
    // fold: lo + hi * 977 + (hi << 32); word j: lo_j + 977 hi_j + hi_(j-1) + carry  (< 2^42 + 2^33: one 64-bit accumulator)
    u32 r[8];
    u64 cy = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        cy = mad32(t[8 + j], 977u, cy);
        cy = mad32(t[j], 1u, cy);
        if (j) cy = mad32(t[8 + j - 1], 1u, cy);
        r[j] = (u32)cy;
        cy >>= 32;
    }
    const u64 R = cy + t[15];  // what is left above 2^256: < 2^33 + 2^11
    // second fold: R * (2^32 + 977) onto the low words (R * 977 < 2^45; R << 32 spans words 1..2)
    u64 c2 = (u64)r[0] + (R & 0xFFFFFFFFull) * 977u;
    r[0] = (u32)c2; c2 >>= 32;
    c2 += (u64)r[1] + (R >> 32) * 977u + (R & 0xFFFFFFFFull);
    r[1] = (u32)c2; c2 >>= 32;
    c2 += (u64)r[2] + (R >> 32);
    r[2] = (u32)c2; c2 >>= 32;
#pragma unroll
    for (int i = 3; i < 8; i++) { c2 += r[i]; r[i] = (u32)c2; c2 >>= 32; }
    return cond_sub_p(r, (u32)c2);  // (a wrap leaves a tiny low part: one subtraction of p settles either case)
}
__device__ __attribute__((noinline)) ec_u256 add(ec_u256 a, ec_u256 b) {  // a, b < p
    u32 r[8];
    u64 c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { c += (u64)a.w[i] + b.w[i]; r[i] = (u32)c; c >>= 32; }
    return cond_sub_p(r, (u32)c);
}
__device__ __attribute__((noinline)) ec_u256 sub(ec_u256 a, ec_u256 b) {  // a, b < p
    u32 r[8];
    long long c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) { c += (long long)a.w[i] - (long long)b.w[i]; r[i] = (u32)c; c >>= 32; }
    if (c == 0) { ec_u256 o;
#pragma unroll
        for (int i = 0; i < 8; i++) o.w[i] = r[i];
        return o; }
    // borrowed: + p = - (2^32 + 977) mod 2^256
    ec_u256 o;
    long long d = (long long)r[0] - 977;
    o.w[0] = (u32)d; d >>= 32;
    d += (long long)r[1] - 1;
    o.w[1] = (u32)d; d >>= 32;
#pragma unroll
    for (int i = 2; i < 8; i++) { d += r[i]; o.w[i] = (u32)d; d >>= 32; }
    return o;
}
// Jacobian doubling (a = 0, 7 multiplications) and mixed addition (11), the formulas of ec_jdbl / ec_jmadd (include/zkw_ecrecover.h)
__device__ __forceinline__ void jdbl(ec_u256& X, ec_u256& Y, ec_u256& Z) {
    const ec_u256 a = mul(X, X), b = mul(Y, Y), c = mul(b, b);
    ec_u256 t = add(X, b);
    t = mul(t, t);
    t = sub(t, a);
    t = sub(t, c);
    const ec_u256 d = add(t, t);
    ec_u256 e = add(a, a);
    e = add(e, a);
    const ec_u256 f = mul(e, e), d2 = add(d, d), yz = mul(Y, Z);
    X = sub(f, d2);
    ec_u256 c8 = add(c, c);
    c8 = add(c8, c8);
    c8 = add(c8, c8);
    ec_u256 dx = sub(d, X);
    dx = mul(e, dx);
    Y = sub(dx, c8);
    Z = add(yz, yz);
}
__device__ __forceinline__ void jmadd(ec_u256& X, ec_u256& Y, ec_u256& Z, const ec_u256& x2, const ec_u256& y2) {
    const ec_u256 zz = mul(Z, Z), zzz = mul(zz, Z), u2 = mul(x2, zz), s2 = mul(y2, zzz);
    const ec_u256 h = sub(u2, X), r = sub(s2, Y);
    const ec_u256 h2 = mul(h, h), h3 = mul(h2, h), xh2 = mul(X, h2);
    ec_u256 t = mul(r, r);
    t = sub(t, h3);
    t = sub(t, xh2);
    const ec_u256 x3 = sub(t, xh2);
    ec_u256 v = sub(xh2, x3);
    v = mul(r, v);
    const ec_u256 yh3 = mul(Y, h3);
    Z = mul(Z, h);
    X = x3;
    Y = sub(v, yh3);
}
}  // namespace ecf

// ---- limb per lane -----------------------------------------------------------------------------------------------------------------
// Callers run with lanes 0..15 of the wave active; a value's lanes 8..15 hold zero. Every function returns canonical limbs (< p).
namespace ecl {
template <int N> __device__ __forceinline__ u32 shr(u32 v) {  // lane k <- lane k - N of the row (zero below it)
    if constexpr (N == 0) return v;
    else return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x110 + N, 0xF, 0xF, false);
}
template <int N> __device__ __forceinline__ u32 shl(u32 v) {  // lane k <- lane k + N of the row (zero above it)
    return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x100 + N, 0xF, 0xF, false);
}
__device__ __forceinline__ u32 lane16() { return threadIdx.x & 15u; }
__device__ __forceinline__ void mac_s(u64& acc, u32& ext, u32 a_uniform, u32 b) {  // (ext : acc) += a * b, a in a scalar register
    u64 c, dead;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc), "=s"(c) : "s"(a_uniform), "v"(b));
    asm("v_addc_co_u32_e64 %0, %1, %0, 0, %2" : "+v"(ext), "=s"(dead) : "s"(c));
}
// limbs X_k < 2^33 (k < 8; zero above) with every carry settled: lane k gets +1 when a carry reaches it — from lane k - 1's bit 32, or
// rippling through lanes that hold 2^32 - 1 (a lane whose bit 32 is set holds a small low word: it never also propagates). The lanes a
// carry reaches are the bits the addition P + (G << 1) changes in P. carry: the carry out of lane 7 (uniform)
__device__ __forceinline__ u32 settle(u64 X, u32& carry) {
    const u32 l = (u32)X;
    const u64 G = __builtin_amdgcn_ballot_w64((u32)(X >> 32) != 0) & 0xFFull, P = __builtin_amdgcn_ballot_w64(l == 0xFFFFFFFFu) & 0xFFull;
    const u64 I = (P + (G << 1)) ^ P;
    carry = (u32)(I >> 8) & 1u;
    return l + ((u32)((I & 0xFFull) >> lane16()) & 1u);
}
// the limbs of c = 2^32 + 977 = 2^256 - p
__device__ __forceinline__ u32 c_limb() { const u32 k = lane16(); return k == 0 ? 977u : k == 1 ? 1u : 0u; }
// r (settled limbs, value < 2^256) + carry * 2^256 < 2 p  ->  mod p
__device__ __forceinline__ u32 canon(u32 r, u32 carry) {
    u32 c2;
    const u32 y = settle((u64)r + c_limb(), c2);  // r + c: carries out of 2^256 exactly when r >= p
    return (carry | c2) ? y : r;
}
__device__ __forceinline__ u32 add(u32 a, u32 b) {  // a, b < p
    u32 co;
    const u32 s = settle((u64)a + b, co);
    return canon(s, co);
}
__device__ __forceinline__ u32 sub(u32 a, u32 b) {  // a, b < p
    // borrows by the same lookahead: a lane generates one when a_k < b_k and passes one on when a_k == b_k
    const u64 B = __builtin_amdgcn_ballot_w64(a < b) & 0xFFull, P = __builtin_amdgcn_ballot_w64(a == b) & 0xFFull;
    const u64 I = (P + (B << 1)) ^ P;
    const u32 d = a - b - ((u32)((I & 0xFFull) >> lane16()) & 1u);
    if (!((I >> 8) & 1)) return d;  // (uniform)
    // borrowed: + p = - c mod 2^256
    const u32 c = c_limb();
    const u64 B2 = __builtin_amdgcn_ballot_w64(d < c) & 0xFFull, P2 = __builtin_amdgcn_ballot_w64(d == c) & 0xFFull;
    const u64 I2 = (P2 + (B2 << 1)) ^ P2;
    return d - c - ((u32)((I2 & 0xFFull) >> lane16()) & 1u);
}
__device__ __forceinline__ u32 mul(u32 a, u32 b) {
    // column k = sum over i of a_i * b_(k - i) in lane k (k < 15): a_i from a scalar register, b moved up the row by i lanes
    u64 acc = 0;
    u32 ext = 0;
    mac_s(acc, ext, __builtin_amdgcn_readlane(a, 0), shr<0>(b));
    mac_s(acc, ext, __builtin_amdgcn_readlane(a, 1), shr<1>(b));
    mac_s(acc, ext, __builtin_amdgcn_readlane(a, 2), shr<2>(b));
    mac_s(acc, ext, __builtin_amdgcn_readlane(a, 3), shr<3>(b));
    mac_s(acc, ext, __builtin_amdgcn_readlane(a, 4), shr<4>(b));
    mac_s(acc, ext, __builtin_amdgcn_readlane(a, 5), shr<5>(b));
    mac_s(acc, ext, __builtin_amdgcn_readlane(a, 6), shr<6>(b));
    mac_s(acc, ext, __builtin_amdgcn_readlane(a, 7), shr<7>(b));
    // word k of the product (lazy: < 2^34) = low word of column k + middle word of column k - 1 + top bits of column k - 2
    const u64 W = (u64)(u32)acc + shr<1>((u32)(acc >> 32)) + shr<2>(ext);
    const u32 k = lane16();
    // fold the words above 2^256: hi * c = 977 hi + (hi << 32): lane k < 8 takes 977 W_(k + 8) and W_(k + 7) (k >= 1); lane 8 takes W_15
    const u64 H = (u64)shl<8>((u32)W) | (u64)shl<8>((u32)(W >> 32)) << 32;
    u64 H1 = (u64)shl<7>((u32)W) | (u64)shl<7>((u32)(W >> 32)) << 32;
    if (k == 0 || k > 8) H1 = 0;
    const u64 R = (k < 8 ? W : 0ull) + 977ull * H + H1;  // < 2^45, lanes 0..8
    // a carry round: T_k = low word + the upper bits of lane k - 1 (< 2^32 + 2^13; lane 9: the upper bits of lane 8)
    u64 T = (u64)(u32)R + shr<1>((u32)(R >> 32));
    // what stands above 2^256 (lanes 8, 9) times c onto lanes 0, 1
    const u64 v = (u64)(u32)__builtin_amdgcn_readlane((u32)T, 8) + ((u64)((u32)__builtin_amdgcn_readlane((u32)(T >> 32), 8) + (u32)__builtin_amdgcn_readlane((u32)T, 9)) << 32);  // < 2^35 (the builtin returns an int: no sign extension)
    if (k >= 8) T = 0;
    T += k == 0 ? 977ull * v : k == 1 ? v : 0ull;  // lane 0 < 2^46, lane 1 < 2^36
    u64 U = (u64)(u32)T + shr<1>((u32)(T >> 32));  // < 2^32 + 2^14; lane 8: the carry out of lane 7 (0 / 1)
    if (__builtin_amdgcn_readlane((u32)U, 8)) U += c_limb();  // (uniform, rare) above 2^256 once more: the low part is tiny, + c settles it
    if (k >= 8) U = 0;
    u32 co;
    u32 r = settle(U, co);
    if (co) { u32 c2; r = settle((u64)r + c_limb(), c2); }  // (uniform, rare) the ripple itself left 2^256: the rest is tiny
    return canon(r, 0);
}
// Jacobian doubling (a = 0) and mixed addition: the formulas of ecf::jdbl / ecf::jmadd
__device__ __forceinline__ void jdbl(u32& X, u32& Y, u32& Z) {
    const u32 a = mul(X, X), b = mul(Y, Y), c = mul(b, b);
    u32 t = add(X, b);
    t = mul(t, t);
    t = sub(t, a);
    t = sub(t, c);
    const u32 d = add(t, t);
    u32 e = add(a, a);
    e = add(e, a);
    const u32 f = mul(e, e), d2 = add(d, d), yz = mul(Y, Z);
    X = sub(f, d2);
    u32 c8 = add(c, c);
    c8 = add(c8, c8);
    c8 = add(c8, c8);
    u32 dx = sub(d, X);
    dx = mul(e, dx);
    Y = sub(dx, c8);
    Z = add(yz, yz);
}
__device__ __forceinline__ void jmadd(u32& X, u32& Y, u32& Z, u32 x2, u32 y2) {
    const u32 zz = mul(Z, Z), zzz = mul(zz, Z), u2 = mul(x2, zz), s2 = mul(y2, zzz);
    const u32 h = sub(u2, X), r = sub(s2, Y);
    const u32 h2 = mul(h, h), h3 = mul(h2, h), xh2 = mul(X, h2);
    u32 t = mul(r, r);
    t = sub(t, h3);
    t = sub(t, xh2);
    const u32 x3 = sub(t, xh2);
    u32 v = sub(xh2, x3);
    v = mul(r, v);
    const u32 yh3 = mul(Y, h3);
    Z = mul(Z, h);
    X = x3;
    Y = sub(v, yh3);
}
// a^((p + 1) / 4): the square root of a quadratic residue (p = 3 mod 4). Square-and-multiply over the constant exponent: uniform control flow
__device__ __forceinline__ u32 pow_sqrt(u32 a) {
    u32 r = a;
    bool started = false;
#pragma unroll 1
    for (int wi = 7; wi >= 0; wi--) {
        const u32 word = EC_P_SQRT_E[wi];
#pragma unroll 1
        for (int bit = 31; bit >= 0; bit--) {
            if (started) r = mul(r, r);
            if ((word >> bit) & 1) {
                if (started) r = mul(r, a);
                started = true;  // (the first set bit: r = a already)
            }
        }
    }
    return r;
}
}  // namespace ecl
}  // namespace zkw
