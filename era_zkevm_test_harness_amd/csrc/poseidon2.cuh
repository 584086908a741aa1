// poseidon2.cuh — Poseidon2 over Goldilocks, width 12 / rate 8 / capacity 4, for gfx950.
//
// Replaces boojum's `Poseidon2Goldilocks` round function that the reference reaches through
// `ZkSyncDefaultRoundFunction` (circuit_encodings/src/lib.rs:12-15) in every queue simulator
// (lib.rs:198-203, 405-409), in produce_fs_challenges (src/witness/utils.rs:515-528) and in the
// Poseidon2 flattened gate of every base-layer circuit (vm_main.rs:55-118, ram_permutation.rs:52-96).
//
// Two device formulations:
//   * p2::permute(s[12])          one state per lane, everything in registers (row-parallel passes:
//                                 trace materialisation, gate checks; 64 independent states per wave).
//   * p2::Coop::permute(x)        one state per 16-lane DPP row, element g of the state in lane g
//                                 (12 active lanes, 4 idle), 4 states per wave. Cross-lane traffic is
//                                 DPP only (quad_perm inside the 4x4 blocks, row_ror across blocks), no
//                                 LDS. This is the latency-oriented form for the serial queue chains.
#pragma once
#include "gl64.cuh"
#include "../../include/zkw_poseidon2_params.h"

namespace p2 {
using gl::u32;
using gl::u64;

static __constant__ u64 c_rc[P2_TOTAL_ROUNDS * P2_WIDTH] = P2_ROUND_CONSTANTS_INIT;
static __constant__ u32 c_shift[P2_WIDTH] = P2_INTERNAL_DIAG_SHIFTS_INIT;

__host__ __device__ __forceinline__ u64 rc_at(int i) {
#if defined(__HIP_DEVICE_COMPILE__)
    return c_rc[i];
#else
    return P2_ROUND_CONSTANTS[i];
#endif
}
__host__ __device__ __forceinline__ u32 shift_at(int i) {
#if defined(__HIP_DEVICE_COMPILE__)
    return c_shift[i];
#else
    return P2_INTERNAL_DIAG_SHIFTS[i];
#endif
}

// ---------------------------------------------------------------- lazy-reduction helpers
// value = lo + hi * 2^64 with a small hi: sums of a few dozen field elements are accumulated without
// reducing and brought back with one `lo + hi * (2^32 - 1)` at the end (2^64 = 2^32 - 1 mod p).
struct Wide {
    u64 lo;
    u32 hi;
};
GL_HD Wide wide(u64 x) { Wide r; r.lo = x; r.hi = 0; return r; }
GL_HD Wide wadd(Wide a, Wide b) {
    Wide r;
    r.lo = a.lo + b.lo;
    r.hi = a.hi + b.hi + (r.lo < a.lo ? 1u : 0u);
    return r;
}
// a * 2^s for 1 <= s <= 31
GL_HD Wide wshl(Wide a, u32 s) {
    Wide r;
    r.lo = a.lo << s;
    r.hi = (a.hi << s) | (u32)(a.lo >> (64 - s));
    return r;
}
// hi < 2^31
GL_HD u64 wreduce(Wide a) {
    u64 t = ((u64)a.hi << 32) - a.hi;  // hi * EPS
    u64 r = a.lo + t;
    u64 c = r < t ? gl::EPS : 0;
    return r + c;  // cannot wrap again: after a wrap r < t < 2^63
}

// ---------------------------------------------------------------- one state per lane
// M4 = [[5,7,1,3],[4,6,1,1],[1,3,5,7],[1,1,4,6]] by the Poseidon2 addition chain, on unreduced sums
GL_HD void m4w(const u64 x[4], Wide t[4]) {
    Wide t0 = wadd(wide(x[0]), wide(x[1])), t1 = wadd(wide(x[2]), wide(x[3]));
    Wide t2 = wadd(wshl(wide(x[1]), 1), t1), t3 = wadd(wshl(wide(x[3]), 1), t0);
    Wide t4 = wadd(wshl(t1, 2), t3), t5 = wadd(wshl(t0, 2), t2);
    t[0] = wadd(t3, t5); t[1] = t5; t[2] = wadd(t2, t4); t[3] = t4;
}

#if defined(__HIP_DEVICE_COMPILE__)
// ---- the linear layers of the lane form on the device, written against the measured issue cost of gfx950 (profiles/r05/valu_ceiling.json:
// nearly every integer instruction that is not a plain 32-bit add / logic op costs the same ~4 cycles, so the count decides). Both layers are
// sums of a few field elements with small coefficients, so they are done on the two 32-bit WORD PLANES of the state separately — a plane's
// sums stay below 2^40 in a 64-bit accumulator, with no carry between the planes to track — and every sum is one instruction:
// v_mad_u64_u32 (32-bit word x small coefficient + 64-bit accumulator) or v_lshl_add_u64 ((acc << k) + acc). An output L + 2^32 H is then
// brought back to 64 bits with 2^64 = EPS in five instructions (wp_reduce). Against the Wide (lo64 + hi32) form above — three instructions
// per addition, a 64-bit compare for every carry: external layer ~250 -> ~160 instructions, internal layer ~190 -> ~110.
__device__ __forceinline__ u64 mad_acc(u32 a, u32 k, u64 acc) {  // a * k + acc, no overflow by construction
    u64 r, dead;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(r), "=s"(dead) : "v"(a), "s"(k), "v"(acc));
    return r;
}
__device__ __forceinline__ u64 mad_acc1(u32 a, u64 acc) {  // a + acc
    u64 r, dead;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %3" : "=v"(r), "=s"(dead) : "v"(a), "v"(acc));
    return r;
}
__device__ __forceinline__ u64 mad_acc2(u32 a, u64 acc) {  // 2 a + acc
    u64 r, dead;
    asm("v_mad_u64_u32 %0, %1, %2, 2, %3" : "=v"(r), "=s"(dead) : "v"(a), "v"(acc));
    return r;
}
// L + 2^32 H mod p for L, H < 2^63 (weak out): the high word of L takes the low word of H (carry c, worth 2^64 = EPS), the high word of H and c
// go in as (Hhi + c) * EPS through the multiplier (carry c2; after that wrap the value is below 2^63, so adding EPS once more cannot wrap)
__device__ __forceinline__ u64 wp_reduce(u64 L, u64 H) {
    const u32 l0 = (u32)L, l1 = (u32)(L >> 32), h0 = (u32)H, h1 = (u32)(H >> 32);
    u32 n1, k, f;
    u64 c, c2, dead, r;
    asm("v_add_co_u32_e64 %0, %1, %2, %3" : "=v"(n1), "=s"(c) : "v"(l1), "v"(h0));
    asm("v_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(k), "=s"(dead) : "v"(h1), "s"(c));
    const u64 x = ((u64)n1 << 32) | l0;
    asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(r), "=s"(c2) : "v"(k), "v"(x));
    asm("v_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(f) : "s"(c2));
    asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(r), "=s"(dead) : "v"(f), "v"(r));  // + EPS after a wrap
    return r;
}
// M4 = [[5,7,1,3],[4,6,1,1],[1,3,5,7],[1,1,4,6]] of one word plane of a block by the Poseidon2 addition chain: 8 instructions (+ one zero extension each for t0, t1)
__device__ __forceinline__ void m4_plane(u32 a0, u32 a1, u32 a2, u32 a3, u64 y[4]) {
    const u64 t0 = mad_acc1(a0, (u64)a1), t1 = mad_acc1(a2, (u64)a3);
    const u64 t2 = mad_acc2(a1, t1), t3 = mad_acc2(a3, t0);
    const u64 t4 = (t1 << 2) + t3, t5 = (t0 << 2) + t2;  // v_lshl_add_u64
    y[0] = t3 + t5; y[1] = t5; y[2] = t2 + t4; y[3] = t4;
}
#endif

// external layer circ(2*M4, M4, M4): every output is < 64 * 2^64 before its single reduction
GL_HD void external(u64 s[12]) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(P2_LINEAR_WIDE)
    u64 L[12], H[12];
#pragma unroll
    for (int b = 0; b < 3; b++) {
        m4_plane((u32)s[4 * b], (u32)s[4 * b + 1], (u32)s[4 * b + 2], (u32)s[4 * b + 3], L + 4 * b);
        m4_plane((u32)(s[4 * b] >> 32), (u32)(s[4 * b + 1] >> 32), (u32)(s[4 * b + 2] >> 32), (u32)(s[4 * b + 3] >> 32), H + 4 * b);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {  // a plane's entries are < 16 * 2^32, the sums below < 64 * 2^32
        const u64 cl = L[i] + L[4 + i] + L[8 + i], ch = H[i] + H[4 + i] + H[8 + i];
#pragma unroll
        for (int b = 0; b < 3; b++) s[4 * b + i] = wp_reduce(L[4 * b + i] + cl, H[4 * b + i] + ch);
    }
#else
    Wide t[12];
    m4w(s, t); m4w(s + 4, t + 4); m4w(s + 8, t + 8);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        Wide col = wadd(wadd(t[i], t[4 + i]), t[8 + i]);
        s[i] = wreduce(wadd(t[i], col)); s[4 + i] = wreduce(wadd(t[4 + i], col)); s[8 + i] = wreduce(wadd(t[8 + i], col));
    }
#endif
}

// internal layer: y_i = x_i * 2^shift_i + sum_j x_j
GL_HD void internal(u64 s[12]) {
    constexpr u32 SH[12] = P2_INTERNAL_DIAG_SHIFTS_INIT;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(P2_LINEAR_WIDE)
    u64 sl = (u64)(u32)s[0], sh = s[0] >> 32;  // the planes' sums: < 12 * 2^32
#pragma unroll
    for (int i = 1; i < 12; i++) { sl = mad_acc1((u32)s[i], sl); sh = mad_acc1((u32)(s[i] >> 32), sh); }
#pragma unroll
    for (int i = 0; i < 12; i++) {  // word * 2^shift + the plane's sum: < 2^47
        const u64 l = SH[i] ? mad_acc((u32)s[i], 1u << SH[i], sl) : mad_acc1((u32)s[i], sl);
        const u64 h = SH[i] ? mad_acc((u32)(s[i] >> 32), 1u << SH[i], sh) : mad_acc1((u32)(s[i] >> 32), sh);
        s[i] = wp_reduce(l, h);
    }
#else
    Wide sum = wide(s[0]);
#pragma unroll
    for (int i = 1; i < 12; i++) sum = wadd(sum, wide(s[i]));
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = wreduce(wadd(SH[i] ? wshl(wide(s[i]), SH[i]) : wide(s[i]), sum));
#endif
}

// LAT: the S-boxes through gl::mul_lat (the compiler-scheduled multiplication) instead of the 14-instruction form. For the few one-lane
// sponges whose loops stay rolled (commitments, challenges): there the carry flags of the short form's SGPR pairs spill (48 B of scratch
// per lane, which tests/test_kernel_resources.py forbids), and their time does not matter.
#ifndef P2_SBOX_GROUP
#define P2_SBOX_GROUP 3
#endif
template <bool LAT = false>
GL_HD void full_round(u64 s[12], int r) {
#pragma unroll
    for (int i = 0; i < 12; i++) {
        s[i] = LAT ? gl::pow7_lat(gl::add_canon(s[i], rc_at(12 * r + i))) : gl::pow7(gl::add_canon(s[i], rc_at(12 * r + i)));
#if defined(__HIP_DEVICE_COMPILE__)
        // the twelve S-boxes of a round are independent and the scheduler would interleave them all: every multiplication in flight
        // holds three carry flags in SGPR pairs, twelve of them exhaust the 102 SGPRs, the spills go to VGPRs and the fills drop to 3
        // waves per SIMD. A scheduling barrier after every P2_SBOX_GROUP S-boxes keeps that many in flight (the waves of a full SIMD
        // supply the rest of the parallelism).
        if (!LAT && (i + 1) % P2_SBOX_GROUP == 0) __builtin_amdgcn_sched_barrier(0);
#endif
    }
    external(s);
}

template <bool LAT = false>
GL_HD void partial_round(u64 s[12], int r) {
    s[0] = LAT ? gl::pow7_lat(gl::add_canon(s[0], rc_at(12 * r))) : gl::pow7(gl::add_canon(s[0], rc_at(12 * r)));
    internal(s);
}

// weak in, weak out
template <bool LAT = false>
GL_HD void permute(u64 s[12]) {
    external(s);
    int r = 0;
    for (int k = 0; k < P2_HALF_FULL_ROUNDS; k++, r++) full_round<LAT>(s, r);
    for (int k = 0; k < P2_PARTIAL_ROUNDS; k++, r++) partial_round<LAT>(s, r);
    for (int k = 0; k < P2_HALF_FULL_ROUNDS; k++, r++) full_round<LAT>(s, r);
}
GL_HD void permute_lat(u64 s[12]) { permute<true>(s); }

#if defined(__HIPCC__)
// ---------------------------------------------------------------- one state per 16-lane row
// DPP controls (GFX9 encoding): quad_perm = sel0 | sel1<<2 | sel2<<4 | sel3<<6; row_ror:n = 0x120 + n.
template <int CTRL>
__device__ __forceinline__ u32 dpp32(u32 v) {
    return (u32)__builtin_amdgcn_update_dpp((int)0, (int)v, CTRL, 0xF, 0xF, false);
}
template <int CTRL>
__device__ __forceinline__ u64 dpp64(u64 v) {
    u32 lo = dpp32<CTRL>((u32)v), hi = dpp32<CTRL>((u32)(v >> 32));
    return ((u64)hi << 32) | lo;
}
constexpr int QP_ROT1 = 1 | (2 << 2) | (3 << 4) | (0 << 6);  // lane j reads lane j+1 (mod 4)
constexpr int QP_ROT2 = 2 | (3 << 2) | (0 << 4) | (1 << 6);
constexpr int QP_ROT3 = 3 | (0 << 2) | (1 << 4) | (2 << 6);
constexpr int QP_SWAP1 = 1 | (0 << 2) | (3 << 4) | (2 << 6);
constexpr int QP_BCAST0 = 0;                                   // every lane of a quad reads its lane 0
constexpr int ROW_ROR4 = 0x124, ROW_ROR8 = 0x128, ROW_ROR12 = 0x12C;

template <int CTRL>
__device__ __forceinline__ Wide wdpp(Wide a) {
    Wide r;
    r.lo = dpp64<CTRL>(a.lo);
    r.hi = dpp32<CTRL>(a.hi);
    return r;
}
// x + c for a CANONICAL c (round constant): one wrap at most, so one conditional "+ EPS" (gl::add handles weak + weak)
__device__ __forceinline__ u64 add_rc_sched(u64 x, u64 c) {
    u32 r0, r1, m;
    asm("v_add_co_u32 %0, vcc, %3, %5\n\t"
        "v_addc_co_u32 %1, vcc, %4, %6, vcc\n\t"
        "v_cndmask_b32 %2, 0, -1, vcc\n\t"
        "v_add_co_u32 %0, vcc, %0, %2\n\t"
        "v_addc_co_u32 %1, vcc, 0, %1, vcc"
        : "=&v"(r0), "=&v"(r1), "=&v"(m)
        : "v"((u32)x), "v"((u32)(x >> 32)), "v"((u32)c), "v"((u32)(c >> 32))
        : "vcc");
    return ((u64)r1 << 32) | r0;
}
__device__ __forceinline__ u64 pow7_sched(u64 x) {  // one dependent chain per wave: the hand-scheduled multiplication (gl64.cuh)
    const u64 x2 = gl::mul_sched(x, x), x3 = gl::mul_sched(x2, x), x4 = gl::mul_sched(x2, x2);
    return gl::mul_sched(x3, x4);
}

struct Coop {
    // per-lane constants, loaded once per kernel
    u64 rc_full[2 * P2_HALF_FULL_ROUNDS];  // c_rc[12*round + g] for the 8 full rounds (0 for idle lanes)
    u32 ka, kb, kd;                        // row of M4 seen from this lane: ka*a + kb*b + c + kd*d
    u32 shift;                             // internal diag shift of element g
    u32 rsh, hi_mask;                      // 32 - shift (mod 32) and all-ones unless shift == 0: the word x * 2^shift spills
    bool active;                           // g < 12
    bool first;                            // g == 0
    bool second;                           // g == 1: lane 0's helper in the partial rounds' S-box (pow7_pair)

    __device__ __forceinline__ void init(int g) {
        active = g < 12;
        first = g == 0;
        second = g == 1;
#pragma unroll
        for (int k = 0; k < 2 * P2_HALF_FULL_ROUNDS; k++) {
            int round = k < P2_HALF_FULL_ROUNDS ? k : k + P2_PARTIAL_ROUNDS;
            rc_full[k] = active ? c_rc[12 * round + g] : 0;
        }
        bool even = (g & 1) == 0;
        ka = even ? 5u : 6u;  // even lane j: 5 x_j + 7 x_{j+1} + x_{j+2} + 3 x_{j+3}
        kb = even ? 7u : 1u;  // odd lane j:  6 x_j +   x_{j+1} + x_{j+2} + 4 x_{j+3}  (indices mod 4)
        kd = even ? 3u : 4u;
        shift = active ? c_shift[g] : 0;
        rsh = (32u - shift) & 31u;
        hi_mask = shift ? 0xffffffffu : 0u;
    }

    // external layer on the distributed state; x weak, idle lanes must hold 0 and get 0 back
    __device__ __forceinline__ u64 external(u64 x) const {
        u64 a = x, b = dpp64<QP_ROT1>(x), c = dpp64<QP_ROT2>(x), d = dpp64<QP_ROT3>(x);
        u64 L = (a & gl::EPS) * ka + (b & gl::EPS) * kb + (c & gl::EPS) + (d & gl::EPS) * kd;  // < 18 * 2^32
        u64 H = (a >> 32) * ka + (b >> 32) * kb + (c >> 32) + (d >> 32) * kd;
        // t = L + H * 2^32 (the M4 row of this lane, 96 bits); col = t summed over the row's four quads (the idle quad holds
        // 0); y = t + col reduced. Hand-scheduled like `internal`: DPP operands taken by the adds themselves, carries in VCC.
        u32 r0, r1, t0, t1, t2, u0, u1, u2, e0, e1, e2;
        asm volatile(
            "v_add_co_u32 %3, vcc, %11, %13\n\t"          // t1 = L.hi + H.lo
            "v_addc_co_u32 %4, vcc, 0, %14, vcc\n\t"      // t2 = H.hi + carry        (t0 = L.lo = %12)
            "s_nop 0\n\t"
            "v_add_co_u32_dpp %5, vcc, %12, %12 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"   // u = t + ror8(t)
            "v_addc_co_u32_dpp %6, vcc, %3, %3, vcc row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_addc_co_u32_dpp %7, vcc, %4, %4, vcc row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_co_u32_dpp %5, vcc, %5, %5 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"      // col = u + ror4(u)
            "v_addc_co_u32_dpp %6, vcc, %6, %6, vcc row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
            "v_addc_co_u32_dpp %7, vcc, %7, %7, vcc row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_co_u32 %0, vcc, %12, %5\n\t"           // y = t + col
            "v_addc_co_u32 %1, vcc, %3, %6, vcc\n\t"
            "v_addc_co_u32 %2, vcc, %4, %7, vcc\n\t"      // y2 < 2^32, worth y2 * (2^32 - 1)
            "v_sub_co_u32 %8, vcc, 0, %2\n\t"
            "v_subbrev_co_u32 %9, vcc, 0, %2, vcc\n\t"
            "v_add_co_u32 %0, vcc, %0, %8\n\t"
            "v_addc_co_u32 %1, vcc, %1, %9, vcc\n\t"
            "v_cndmask_b32 %10, 0, -1, vcc\n\t"
            "v_add_co_u32 %0, vcc, %0, %10\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc"
            : "=&v"(r0), "=&v"(r1), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(u0), "=&v"(u1), "=&v"(u2), "=&v"(e0), "=&v"(e1), "=&v"(e2)
            : "v"((u32)(L >> 32)), "v"((u32)L), "v"((u32)H), "v"((u32)(H >> 32))
            : "vcc");
        const u64 y = ((u64)r1 << 32) | r0;
        return active ? y : 0;
    }

    // Internal layer y_g = x_g * 2^shift_g + sum over the row, hand-scheduled: the compiler's form of the same arithmetic is
    // ~75 instructions per call (a v_mov_b32_dpp per 32-bit word and butterfly step, 64-bit compares to recover carries,
    // zero-extension moves); here every butterfly step is three adds-with-carry that take their DPP operand directly
    // (12 instructions for the 96-bit row sum), the reduction keeps its carries in VCC (7), ~30 in all. 22 calls per
    // permutation, and a lone chain wave pays ~4 cycles per instruction whatever it is.
    __device__ __forceinline__ u64 internal(u64 x) const {
        const u32 x0 = (u32)x, x1 = (u32)(x >> 32);
        u32 s0, s1, s2;
        const u32 zero = 0;
        // s = sum of x over the 16 lanes of the row as a 96-bit number (idle lanes hold 0). A DPP read needs two wait
        // states after the VALU write of its source: s_nop before the first step; inside, three instructions separate a
        // register's write from its next DPP read.
        asm volatile(
            "s_nop 1\n\t"
            "v_add_co_u32_dpp %0, vcc, %3, %3 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_addc_co_u32_dpp %1, vcc, %4, %4, vcc row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
            "v_addc_co_u32 %2, vcc, 0, %5, vcc\n\t"
            "v_add_co_u32_dpp %0, vcc, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
            "v_addc_co_u32_dpp %1, vcc, %1, %1, vcc row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
            "v_addc_co_u32_dpp %2, vcc, %2, %2, vcc row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
            "v_add_co_u32_dpp %0, vcc, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
            "v_addc_co_u32_dpp %1, vcc, %1, %1, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
            "v_addc_co_u32_dpp %2, vcc, %2, %2, vcc quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
            "v_add_co_u32_dpp %0, vcc, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_addc_co_u32_dpp %1, vcc, %1, %1, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
            "v_addc_co_u32_dpp %2, vcc, %2, %2, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
            : "=&v"(s0), "=&v"(s1), "=&v"(s2)
            : "v"(x0), "v"(x1), "v"(zero)
            : "vcc");
        // m = x * 2^shift as 96 bits; rsh / hi_mask are per-lane constants (shift 0: no high word)
        const u64 mlo = x << shift;
        const u32 m2 = (x1 >> rsh) & hi_mask;
        u32 r0, r1;
        u32 t0, t1, t2, t3;
        asm volatile(
            "v_add_co_u32 %0, vcc, %6, %9\n\t"            // y = m + s (96 bits): y0
            "v_addc_co_u32 %1, vcc, %7, %10, vcc\n\t"     // y1
            "v_addc_co_u32 %2, vcc, %8, %11, vcc\n\t"     // y2 < 2^32: the part above 2^64, worth y2 * (2^32 - 1)
            "v_sub_co_u32 %3, vcc, 0, %2\n\t"             // y2 * EPS = (y2 << 32) - y2: low word
            "v_subbrev_co_u32 %4, vcc, 0, %2, vcc\n\t"    //                              high word
            "v_add_co_u32 %0, vcc, %0, %3\n\t"
            "v_addc_co_u32 %1, vcc, %1, %4, vcc\n\t"
            "v_cndmask_b32 %5, 0, -1, vcc\n\t"            // wrapped past 2^64 (= EPS): add it back; cannot wrap twice
            "v_add_co_u32 %0, vcc, %0, %5\n\t"
            "v_addc_co_u32 %1, vcc, 0, %1, vcc"
            : "=&v"(r0), "=&v"(r1), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
            : "v"((u32)mlo), "v"((u32)(mlo >> 32)), "v"(m2), "v"(s0), "v"(s1), "v"(s2)
            : "vcc");
        const u64 y = ((u64)r1 << 32) | r0;
        return active ? y : 0;
    }

    // weak in / weak out; x = element g of the state (0 in idle lanes)
    __device__ __forceinline__ u64 permute(u64 x) const {
        x = external(x);
#pragma unroll
        for (int k = 0; k < P2_HALF_FULL_ROUNDS; k++) x = external(pow7_sched(add_rc_sched(x, rc_full[k])));
        for (int k = 0; k < P2_PARTIAL_ROUNDS; k++) {
            u64 rc = c_rc[12 * (P2_HALF_FULL_ROUNDS + k)];  // wave-uniform -> scalar load
            u64 sx = pow7_pair(add_rc_sched(x, rc));
            x = internal(first ? sx : x);
        }
#pragma unroll
        for (int k = 0; k < P2_HALF_FULL_ROUNDS; k++)
            x = external(pow7_sched(add_rc_sched(x, rc_full[P2_HALF_FULL_ROUNDS + k])));
        return x;
    }

    // The ONE S-box of a partial round (element 0) on two lanes of the row's first quad: after the square, lane 0 multiplies on to the cube
    // while lane 1 squares again, and lane 0 takes the fourth power across the quad — three multiplications deep instead of four (a lone
    // wave's time is its instruction count). Only lane 0's result means anything; the other lanes' is discarded by the caller.
    __device__ __forceinline__ u64 pow7_pair(u64 t) const {
        const u64 t0 = dpp64<QP_BCAST0>(t);            // lanes 0..3 of a quad: lane 0's element
        const u64 x2 = gl::mul_sched(t0, t0);
        const u64 b = second ? x2 : t0;
        const u64 y = gl::mul_sched(x2, b);            // lane 0: x^3, lane 1: x^4
        return gl::mul_sched(y, dpp64<QP_SWAP1>(y));   // lane 0: x^3 * x^4
    }
};

// One flattened Poseidon2 gate by the 16 lanes of a row (Coop, the form of the queue-chain kernels: lane g holds element g, the
// linear layers cross lanes by DPP): every lane stores its element after each full round, lane 0 the S-box output of each partial
// round — the 130 variables in the order of orc_poseidon2_flattened. x: this lane's input (0 in lanes 12..15); returns its output.
template <class Put>
__device__ __forceinline__ u64 coop_flattened(const Coop& co, u64 x, u32 g, Put&& put) {
    if (co.active) put(g, gl::canon(x));
    x = co.external(x);
#pragma unroll
    for (int k = 0; k < P2_HALF_FULL_ROUNDS; k++) {
        x = co.external(pow7_sched(add_rc_sched(x, co.rc_full[k])));
        if (co.active) put(12 * (k + 1) + g, gl::canon(x));
    }
    for (int k = 0; k < P2_PARTIAL_ROUNDS; k++) {
        const u64 rc = c_rc[12 * (P2_HALF_FULL_ROUNDS + k)];
        const u64 sx = co.pow7_pair(add_rc_sched(x, rc));  // (lane 0's value is the S-box output; the other lanes' is not used)
        if (co.first) put(12 * (P2_HALF_FULL_ROUNDS + 1) + k, gl::canon(sx));
        x = co.internal(co.first ? sx : x);
    }
#pragma unroll
    for (int k = 0; k < P2_HALF_FULL_ROUNDS; k++) {
        x = co.external(pow7_sched(add_rc_sched(x, co.rc_full[P2_HALF_FULL_ROUNDS + k])));
        if (co.active) put(12 * (P2_HALF_FULL_ROUNDS + 1) + P2_PARTIAL_ROUNDS + 12 * k + g, gl::canon(x));
    }
    return gl::canon(x);
}


// ---------------------------------------------------------------- one state per QUAD (4 lanes)
// Lane j of the quad holds elements {j, 4+j, 8+j} (column j of the three 4-element blocks), 16 states per
// wave. The 4x4 block products cross lanes with quad_perm DPP only; the sum over blocks and most of the
// internal layer stay inside the lane, and the three elements of a lane give the in-order wave three
// independent S-box chains to interleave. ~8.1k instructions per permutation step for 16 states, against
// ~5.9k for 4 states in the row-of-16 form above.
#if defined(P2_Q4_PARTIAL_SCHED)
#define Q4_PARTIAL_MUL gl::mul_sched
#else
#define Q4_PARTIAL_MUL gl::mul_lat
#endif
#if defined(P2_Q4_FULL_CYC)
#define Q4_FULL_POW7 gl::pow7
#else
#define Q4_FULL_POW7 gl::pow7_lat
#endif
struct Coop4 {
    u64 rc_full[2 * P2_HALF_FULL_ROUNDS][3];
    u32 ka, kb, kd;
    u32 shift[3];
    bool first;  // lane 0 of the quad: owns element 0
    bool second; // lane 1: lane 0's helper in the partial rounds' S-box

    __device__ __forceinline__ void init(int j) {
        first = j == 0;
        second = j == 1;
#pragma unroll
        for (int k = 0; k < 2 * P2_HALF_FULL_ROUNDS; k++) {
            int round = k < P2_HALF_FULL_ROUNDS ? k : k + P2_PARTIAL_ROUNDS;
#pragma unroll
            for (int c = 0; c < 3; c++) rc_full[k][c] = c_rc[12 * round + 4 * c + j];
        }
        bool even = (j & 1) == 0;
        ka = even ? 5u : 6u;
        kb = even ? 7u : 1u;
        kd = even ? 3u : 4u;
#pragma unroll
        for (int c = 0; c < 3; c++) shift[c] = c_shift[4 * c + j];
    }

    __device__ __forceinline__ Wide m4_row(u64 x) const {
        u64 a = x, b = dpp64<QP_ROT1>(x), c = dpp64<QP_ROT2>(x), d = dpp64<QP_ROT3>(x);
        u64 L = (a & gl::EPS) * ka + (b & gl::EPS) * kb + (c & gl::EPS) + (d & gl::EPS) * kd;
        u64 H = (a >> 32) * ka + (b >> 32) * kb + (c >> 32) + (d >> 32) * kd;
        Wide t;
        t.lo = L + (H << 32);
        t.hi = (u32)(H >> 32) + (t.lo < L ? 1u : 0u);
        return t;
    }

    __device__ __forceinline__ void external(u64 x[3]) const {
        Wide t0 = m4_row(x[0]), t1 = m4_row(x[1]), t2 = m4_row(x[2]);
        Wide col = wadd(wadd(t0, t1), t2);
        x[0] = wreduce(wadd(t0, col));
        x[1] = wreduce(wadd(t1, col));
        x[2] = wreduce(wadd(t2, col));
    }

    __device__ __forceinline__ void internal(u64 x[3]) const {
        Wide s;
        s.lo = x[0]; s.hi = 0;
        Wide s1; s1.lo = x[1]; s1.hi = 0;
        Wide s2; s2.lo = x[2]; s2.hi = 0;
        s = wadd(wadd(s, s1), s2);
        s = wadd(s, wdpp<QP_ROT2>(s));
        s = wadd(s, wdpp<QP_SWAP1>(s));
#pragma unroll
        for (int c = 0; c < 3; c++) {
            Wide m;
            m.lo = x[c] << shift[c];
            m.hi = (u32)((x[c] >> 1) >> (63 - shift[c]));
            x[c] = wreduce(wadd(m, s));
        }
    }

    // weak in / weak out
    __device__ __forceinline__ void permute(u64 x[3]) const {
        external(x);
#pragma unroll
        for (int k = 0; k < P2_HALF_FULL_ROUNDS; k++) {
#pragma unroll
            for (int c = 0; c < 3; c++) x[c] = Q4_FULL_POW7(gl::add_canon(x[c], rc_full[k][c]));
            external(x);
        }
        for (int k = 0; k < P2_PARTIAL_ROUNDS; k++) {
            u64 rc = c_rc[12 * (P2_HALF_FULL_ROUNDS + k)];
            // (the hand-scheduled multiplication of the row form was tried here, where a lane has ONE dependent S-box: the
            // chain kernel got 4 % slower, 891 against 932 M permutations/s — with ~2 waves per SIMD the other wave fills
            // the gaps the compiler's schedule leaves, and the opaque asm blocks only cost)
            // the round's one S-box on two lanes of the quad, as in the row form (Coop::pow7_pair): lane 0 goes on to the cube while
            // lane 1 squares again — three multiplications per partial round instead of four for the whole wave
            const u64 t0 = dpp64<QP_BCAST0>(gl::add_canon(x[0], rc));
            const u64 x2 = Q4_PARTIAL_MUL(t0, t0);
            const u64 y = Q4_PARTIAL_MUL(x2, second ? x2 : t0);
            const u64 sx = Q4_PARTIAL_MUL(y, dpp64<QP_SWAP1>(y));
            x[0] = first ? sx : x[0];
            internal(x);
        }
#pragma unroll
        for (int k = 0; k < P2_HALF_FULL_ROUNDS; k++) {
#pragma unroll
            for (int c = 0; c < 3; c++) x[c] = Q4_FULL_POW7(gl::add_canon(x[c], rc_full[P2_HALF_FULL_ROUNDS + k][c]));
            external(x);
        }
    }
};

// ---------------------------------------------------------------- one state per PAIR of lanes: 32 states per wave
// Lane j (0 / 1) of a pair holds elements 4c + 2j, 4c + 2j + 1 of the three 4-blocks (x[2c], x[2c + 1]). M4 is symmetric under
// exchanging the halves of a block — with own = (o0, o1), partner = (p0, p1):  B = 4 (o0 + o1) + 2 o1 + (p0 + p1),
// A = B + 2 p1 + (o0 + o1) are rows (0, 1) of the block for lane 0 and rows (2, 3) for lane 1 — so both lanes run the same code
// on one exchanged sum and one exchanged element per block. Against the quad form: a partial round's single S-box idles one
// lane of two instead of three of four and a lane's six S-boxes per full round interleave: ~270 wave-instructions per
// permutation against ~506 (the lane form: ~190), at about the latency the quad form has with two waves per SIMD.
struct Coop2 {
    u64 rc_full[2 * P2_HALF_FULL_ROUNDS][6];
    u32 shift[6];
    bool first;  // lane 0 of the pair: owns element 0

    __device__ __forceinline__ void init(int j) {
        first = j == 0;
#pragma unroll
        for (int k = 0; k < 2 * P2_HALF_FULL_ROUNDS; k++) {
            const int round = k < P2_HALF_FULL_ROUNDS ? k : k + P2_PARTIAL_ROUNDS;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                rc_full[k][2 * c] = c_rc[12 * round + 4 * c + 2 * j];
                rc_full[k][2 * c + 1] = c_rc[12 * round + 4 * c + 2 * j + 1];
            }
        }
#pragma unroll
        for (int c = 0; c < 3; c++) {
            shift[2 * c] = c_shift[4 * c + 2 * j];
            shift[2 * c + 1] = c_shift[4 * c + 2 * j + 1];
        }
    }

    __device__ __forceinline__ void external(u64 x[6]) const {
        Wide A[3], B[3];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const Wide own = wadd(wide(x[2 * c]), wide(x[2 * c + 1]));
            const Wide oth = wdpp<QP_SWAP1>(own);
            const u64 p1 = dpp64<QP_SWAP1>(x[2 * c + 1]);
            B[c] = wadd(wadd(wshl(own, 2), wshl(wide(x[2 * c + 1]), 1)), oth);
            A[c] = wadd(wadd(B[c], wshl(wide(p1), 1)), own);
        }
        const Wide colA = wadd(wadd(A[0], A[1]), A[2]), colB = wadd(wadd(B[0], B[1]), B[2]);
#pragma unroll
        for (int c = 0; c < 3; c++) {
            x[2 * c] = wreduce(wadd(A[c], colA));
            x[2 * c + 1] = wreduce(wadd(B[c], colB));
        }
    }

    __device__ __forceinline__ void internal(u64 x[6]) const {
        Wide s = wide(x[0]);
#pragma unroll
        for (int i = 1; i < 6; i++) s = wadd(s, wide(x[i]));
        s = wadd(s, wdpp<QP_SWAP1>(s));
#pragma unroll
        for (int i = 0; i < 6; i++) {
            Wide m;
            m.lo = x[i] << shift[i];
            m.hi = (u32)((x[i] >> 1) >> (63 - shift[i]));
            x[i] = wreduce(wadd(m, s));
        }
    }

    // weak in / weak out
    __device__ __forceinline__ void permute(u64 x[6]) const {
        external(x);
#pragma unroll
        for (int k = 0; k < P2_HALF_FULL_ROUNDS; k++) {
#pragma unroll
            for (int i = 0; i < 6; i++) x[i] = gl::pow7_lat(gl::add(x[i], rc_full[k][i]));
            external(x);
        }
        for (int k = 0; k < P2_PARTIAL_ROUNDS; k++) {
            const u64 rc = c_rc[12 * (P2_HALF_FULL_ROUNDS + k)];
            const u64 sx = gl::pow7_lat(gl::add(x[0], rc));
            x[0] = first ? sx : x[0];
            internal(x);
        }
#pragma unroll
        for (int k = 0; k < P2_HALF_FULL_ROUNDS; k++) {
#pragma unroll
            for (int i = 0; i < 6; i++) x[i] = gl::pow7_lat(gl::add(x[i], rc_full[P2_HALF_FULL_ROUNDS + k][i]));
            external(x);
        }
    }
};
#endif  // __HIPCC__

}  // namespace p2
