// gl64.cuh — Goldilocks field p = 2^64 - 2^32 + 1 for gfx950 (and the host side of libzkw).
//
// Replaces boojum's GoldilocksField as used by the reference at circuit_encodings/src/lib.rs:664-713,
// src/witness/utils.rs:511,578-590 (add_assign / mul_assign / from_u64_with_reduction).
//
// Representation: a u64 that is congruent to the element but NOT necessarily canonical ("weak" form,
// any value in [0, 2^64)). Every operation accepts weak inputs and returns a weak output; `canon()`
// maps to the unique representative < p and is applied once, at the point where a value is stored to
// a buffer that crosses the C ABI (the reference compares canonical values).
//
// Reduction identity: 2^64 = 2^32 - 1 (=: EPS) and 2^96 = -1 (mod p).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gl {

typedef uint64_t u64;
typedef uint32_t u32;

constexpr u64 P = 0xFFFFFFFF00000001ULL;
constexpr u64 EPS = 0xFFFFFFFFULL;

#define GL_HD __host__ __device__ __forceinline__

GL_HD u64 canon(u64 x) { return x >= P ? x - P : x; }

// a + b (mod p), weak in / weak out. The second fix-up only triggers for non-canonical operands.
GL_HD u64 add(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__)
    // the carry-out of the 64-bit add (v_add_co / v_addc_co) instead of a second 64-bit compare
    unsigned long long s, t;
    const bool c = __builtin_uaddll_overflow(a, b, &s);
    const bool c2 = __builtin_uaddll_overflow(s, c ? EPS : 0, &t);
    return t + (c2 ? EPS : 0);
#else
    u64 s = a + b;
    u64 c = s < a ? EPS : 0;
    u64 t = s + c;
    u64 c2 = t < s ? EPS : 0;
    return t + c2;
#endif
}

// a + c (mod p) for a CANONICAL c (a round constant): the sum wraps at most once, and after a wrap it is below c < p, so adding EPS cannot
// wrap again — four instructions instead of the nine of the general add. Weak a in, weak out.
GL_HD u64 add_canon(u64 a, u64 c) {
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned long long s;
    const bool w = __builtin_uaddll_overflow(a, c, &s);
    u32 f = w ? 1u : 0u;
    u64 r, dead;
    asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(r), "=s"(dead) : "v"(f), "v"((u64)s));  // + EPS after a wrap
    return r;
#else
    u64 s = a + c;
    return s < a ? s + EPS : s;
#endif
}

// a - b (mod p)
GL_HD u64 sub(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned long long d, t;
    const bool br = __builtin_usubll_overflow(a, b, &d);
    const bool br2 = __builtin_usubll_overflow(d, br ? EPS : 0, &t);
    return t - (br2 ? EPS : 0);
#else
    u64 d = a - b;
    u64 br = a < b ? EPS : 0;
    u64 t = d - br;
    u64 br2 = t > d ? EPS : 0;
    return t - br2;
#endif
}

GL_HD u64 neg(u64 a) { return sub(0, a); }

// (hi * 2^64 + lo) mod p
GL_HD u64 reduce128(u64 lo, u64 hi) {
    u64 hh = hi >> 32, hl = hi & EPS;
#if defined(__HIP_DEVICE_COMPILE__)
    // the overflow builtins keep the borrow / carry of the 64-bit subtract / add instead of re-deriving them with a
    // second 64-bit compare: 86 instead of 94 VALU instructions per S-box (tools note in DESIGN.md 3.2)
    unsigned long long t0, r;
    const bool borrow = __builtin_usubll_overflow(lo, hh, &t0);
    t0 -= borrow ? EPS : 0;  // wrapped by 2^64 = EPS: take it back out
    const u64 t1 = (hl << 32) - hl;  // hl * EPS < 2^64
    const bool carry = __builtin_uaddll_overflow(t0, t1, &r);
    r += carry ? EPS : 0;    // cannot overflow a second time
    return r;
#else
    u64 t0 = lo - hh;
    if (lo < hh) t0 -= EPS;
    u64 t1 = hl * EPS;
    u64 r = t0 + t1;
    if (r < t1) r += EPS;
    return r;
#endif
}

#if defined(__HIPCC__)
// a * b mod p, hand-scheduled for gfx950: four independent 32 x 32 products (the compiler's chain feeds each product's
// high half into the next one's addend through a v_mov and multiplies by 2^32 - 1 with a fifth v_mad_u64_u32), three
// adds-with-carry to line the partial products up and a 12-instruction reduction whose borrow and carries stay in VCC:
// 20 instructions against ~23 + s_nops. Weak in, weak out.
__device__ __forceinline__ u64 mul_sched(u64 a, u64 b) {
#if !defined(__HIP_DEVICE_COMPILE__)
    return 0;  // device-only (the host pass of hipcc only needs the declaration)
#else
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    const u32 zero = 0;
    u64 p, q, h;
    u32 cm;
    asm("v_mad_u64_u32 %0, s[100:101], %4, %6, 0\n\t"      // p = a0 * b0
        "v_mad_u64_u32 %1, s[100:101], %4, %7, 0\n\t"      // q = a0 * b1
        "v_mad_u64_u32 %2, s[100:101], %5, %7, 0\n\t"      // h = a1 * b1
        "v_mad_u64_u32 %1, vcc, %5, %6, %1\n\t"            // q += a1 * b0, carry out of 64 bits
        "v_addc_co_u32 %3, vcc, 0, %8, vcc"                   // cm = that carry (worth 2^96)
        : "=&v"(p), "=&v"(q), "=&v"(h), "=&v"(cm)
        : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(zero)
        : "vcc", "s100", "s101");
    const u32 p0 = (u32)p, p1 = (u32)(p >> 32), q0 = (u32)q, q1 = (u32)(q >> 32), h0 = (u32)h, h1 = (u32)(h >> 32);
    u32 r0, r1, w1, g0, g1, m, t0, t1;
    asm("v_add_co_u32 %2, vcc, %8, %10\n\t"                // product = p + (q << 32) + (h << 64) + (cm << 96): word 1
        "v_addc_co_u32 %3, vcc, %12, %11, vcc\n\t"         // word 2 = hl
        "v_addc_co_u32 %4, vcc, %13, %14, vcc\n\t"         // word 3 = hh
        "v_sub_co_u32 %0, vcc, %9, %4\n\t"                 // lo - hh            (2^96 = -1)
        "v_subbrev_co_u32 %1, vcc, 0, %2, vcc\n\t"
        "v_cndmask_b32 %5, 0, -1, vcc\n\t"                 // borrowed: the wrap added 2^64 = EPS, take it out again
        "v_sub_co_u32 %0, vcc, %0, %5\n\t"
        "v_subbrev_co_u32 %1, vcc, 0, %1, vcc\n\t"
        "v_sub_co_u32 %6, vcc, 0, %3\n\t"                  // hl * EPS = (hl << 32) - hl
        "v_subbrev_co_u32 %7, vcc, 0, %3, vcc\n\t"
        "v_add_co_u32 %0, vcc, %0, %6\n\t"
        "v_addc_co_u32 %1, vcc, %1, %7, vcc\n\t"
        "v_cndmask_b32 %5, 0, -1, vcc\n\t"                 // carried: add EPS; cannot carry twice
        "v_add_co_u32 %0, vcc, %0, %5\n\t"
        "v_addc_co_u32 %1, vcc, 0, %1, vcc"
        : "=&v"(r0), "=&v"(r1), "=&v"(w1), "=&v"(g0), "=&v"(g1), "=&v"(m), "=&v"(t0), "=&v"(t1)
        : "v"(p1), "v"(p0), "v"(q0), "v"(q1), "v"(h0), "v"(h1), "v"(cm)
        : "vcc");
    return ((u64)r1 << 32) | r0;
#endif
}
#endif

#if defined(__HIPCC__)
// a * b mod p priced by the MEASURED issue cost of each instruction class on gfx950 (tools/ubench_valu_ceiling.hip, profiles/r05:
// per wave-instruction and SIMD a 32-bit VALU op costs 2.2 cycles, v_mad_u64_u32 4.1, an add / subtract that writes or reads a carry 4.2,
// a 64-bit add or compare ~4) rather than by instruction count. For kernels that keep a SIMD full (the trace fills: 4-8 waves per SIMD)
// cycles are what a multiplication costs; the compiler's form is 21 instructions = 68.7 cycles (six v_mov to pair a half with a zero
// register, four 64-bit adds, two 64-bit compares), this one 14 instructions = ~51:
//   four independent 32 x 32 products, the cross terms added inside the multiplier (carry cm, worth 2^96 = -1), three adds-with-carry
//   for the exact words w1..w3, w2 * (2^32 - 1) + (w1:w0) in one more v_mad_u64_u32 (carry c1, worth 2^64 = EPS), w3 and cm taken off
//   by one borrow chain (borrow b1, worth -EPS), and ONE 64-bit add of (c1 - b1) * EPS built by three v_cndmask from the two flags
//   (both set: the wrap and the borrow cancel). Carries live in SGPR pairs, the flag algebra runs on the scalar unit.
// One asm statement per instruction: the compiler still schedules and interleaves independent multiplications. Weak in, weak out.
__device__ __forceinline__ u64 mul_cyc(u64 a, u64 b) {
#if !defined(__HIP_DEVICE_COMPILE__)
    return 0;  // device-only
#else
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    u64 p, q, h, x, cm, c1, cw, cw2, bw, b1f, dead;
    asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(p), "=s"(dead) : "v"(a0), "v"(b0));
    asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(q), "=s"(dead) : "v"(a0), "v"(b1));
    asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(h), "=s"(dead) : "v"(a1), "v"(b1));
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(q), "=s"(cm) : "v"(a1), "v"(b0), "v"(q));
    const u32 p0 = (u32)p, p1 = (u32)(p >> 32), q0 = (u32)q, q1 = (u32)(q >> 32), h0 = (u32)h, h1 = (u32)(h >> 32);
    u32 w1, w2, w3, y0, y1, dlo, dhi;
    asm("v_add_co_u32_e64 %0, %1, %2, %3" : "=v"(w1), "=s"(cw) : "v"(p1), "v"(q0));
    asm("v_addc_co_u32_e64 %0, %1, %2, %3, %4" : "=v"(w2), "=s"(cw2) : "v"(h0), "v"(q1), "s"(cw));
    asm("v_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(w3), "=s"(dead) : "v"(h1), "s"(cw2));  // h1 <= 2^32 - 2: no carry out
    const u64 lo = ((u64)w1 << 32) | p0;
    asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(x), "=s"(c1) : "v"(w2), "v"(lo));
    const u32 x0 = (u32)x, x1 = (u32)(x >> 32);
    asm("v_subb_co_u32_e64 %0, %1, %2, %3, %4" : "=v"(y0), "=s"(bw) : "v"(x0), "v"(w3), "s"(cm));
    asm("v_subbrev_co_u32_e64 %0, %1, 0, %2, %3" : "=v"(y1), "=s"(b1f) : "v"(x1), "s"(bw));
    const u64 mp = c1 & ~b1f, mn = b1f & ~c1;  // +EPS / -EPS (scalar unit)
    asm("v_cndmask_b32_e64 %0, 0, -1, %1" : "=v"(dhi) : "s"(mn));
    asm("v_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(dlo) : "s"(mn));
    asm("v_cndmask_b32_e64 %0, %1, -1, %2" : "=v"(dlo) : "v"(dlo), "s"(mp));
    u64 d = ((u64)dhi << 32) | dlo;  // -EPS = (1, 0xFFFFFFFF) mod 2^64
    asm("" : "+v"(d));               // one register pair, one v_lshl_add_u64 (the optimizer otherwise adds the halves one by one)
    return (((u64)y1 << 32) | y0) + d;
#endif
}
#endif

// a * b mod p, the compiler-scheduled form: four chained v_mad_u64_u32 + reduce128, 21 instructions that the compiler interleaves with
// whatever else the lane has to do. For LATENCY-bound code — the serial queue chains, one wave per SIMD (p2::Coop4 / Coop2): with the
// 14-instruction form below k_chain_full_q4 was 11 % slower (4.91 -> 5.45 s per step, profiles/r05/README.md), its carry chains through
// SGPR pairs stall a lone wave. Weak in, weak out.
GL_HD u64 mul_lat(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(GL_MUL_SCHEDULED_EVERYWHERE)
    // measured (bench.py): the hand-scheduled form only pays where a wave runs ONE dependent S-box chain (the row-form
    // queue chains, p2::Coop: 9.99 -> 7.44 us per permutation with the other hand-scheduled pieces). Kernels with several
    // independent multiplications per lane (quad-form chains, the per-lane permutation of the trace fills) lose 3-4 %:
    // the compiler interleaves independent multiplications instruction by instruction, opaque asm blocks it cannot.
    return mul_sched(a, b);
#else
#if defined(__HIP_DEVICE_COMPILE__)
    // four chained v_mad_u64_u32 give both halves of the 128-bit product; computing `a * b` and
    // `__umul64hi(a, b)` separately costs 5 mads + 2 v_mul_lo_u32 (+16 % instructions per S-box, measured)
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    const u64 t = (u64)a0 * b0;
    const u64 t1 = (u64)a0 * b1 + (t >> 32);
    const u64 t2 = (u64)a1 * b0 + (u32)t1;
    u64 lo = (t2 << 32) | (u32)t;
    u64 hi = (u64)a1 * b1 + (t1 >> 32) + (t2 >> 32);
#else
    unsigned __int128 w = (unsigned __int128)a * b;
    u64 lo = (u64)w, hi = (u64)(w >> 64);
#endif
    return reduce128(lo, hi);
#endif
}

// a * b mod p. On the device: the 14-instruction form (mul_cyc) — every kernel that keeps its SIMDs full is bound by VALU issue cycles,
// and nearly every instruction of a multiplication is a half-rate one (profiles/r05/valu_ceiling.json), so fewer instructions is the
// lever: k_ram_fill_poseidon 2.75 -> 2.31 s per step, row C 0.49 -> 0.31 s. GL_MUL_COMPILER_FORM restores the old default (A/B builds).
GL_HD u64 mul(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(GL_MUL_COMPILER_FORM)
    return mul_cyc(a, b);
#else
    return mul_lat(a, b);
#endif
}

GL_HD u64 sqr(u64 a) { return mul(a, a); }

// a*b + c
GL_HD u64 mul_add(u64 a, u64 b, u64 c) { return add(mul(a, b), c); }

// x^7 (the Poseidon2 S-box over Goldilocks)
GL_HD u64 pow7(u64 x) {
    u64 x2 = sqr(x);
    u64 x3 = mul(x2, x);
    u64 x4 = sqr(x2);
    return mul(x3, x4);
}
GL_HD u64 pow7_lat(u64 x) {  // for the latency-bound chain forms (see mul_lat)
    u64 x2 = mul_lat(x, x);
    u64 x3 = mul_lat(x2, x);
    u64 x4 = mul_lat(x2, x2);
    return mul_lat(x3, x4);
}

// x * 2^s for 0 <= s < 32
GL_HD u64 mul_pow2(u64 x, u32 s) {
    u64 lo = x << s;
    u64 hi = s ? (x >> (64 - s)) : 0;  // < 2^32
    u64 t1 = (hi << 32) - hi;          // hi * EPS
    u64 r = lo + t1;
    if (r < t1) r += EPS;
    return r;
}

// small-constant multiple, k < 2^32
GL_HD u64 mul_small(u64 x, u32 k) {
    u64 l = (x & EPS) * k;   // < 2^64
    u64 h = (x >> 32) * k;   // < 2^64 ; value = l + h * 2^32
    u64 hh = h >> 32, hl = h & EPS;  // h*2^32 = hl*2^32 + hh*2^64
    u64 r = l + (hl << 32);
    u64 c = r < l ? EPS : 0;
    u64 t = hh * EPS;  // hh < 2^32
    u64 r2 = r + c;    // r < 2^64 - ... after wrap, cannot overflow
    u64 r3 = r2 + t;
    if (r3 < t) r3 += EPS;
    return r3;
}

GL_HD u64 pow(u64 a, u64 e) {
    u64 r = 1;
    while (e) {
        if (e & 1) r = mul(r, a);
        a = sqr(a);
        e >>= 1;
    }
    return r;
}

// a^(p - 2) by an addition chain: p - 2 = (2^31 - 1) * 2^33 + (2^32 - 1), and with e_k = a^(2^k - 1): e_2k = e_k^(2^k) * e_k,
// e_(k+1) = e_k^2 * a — 64 squarings + 9 multiplications (square-and-multiply over the 63 one-bits of p - 2: 64 + 63)
GL_HD u64 sqr_n(u64 a, int n) {
    for (int i = 0; i < n; i++) a = sqr(a);
    return a;
}
GL_HD u64 inv(u64 a) {
    const u64 e2 = mul(sqr(a), a);
    const u64 e3 = mul(sqr(e2), a);
    const u64 e6 = mul(sqr_n(e3, 3), e3);
    const u64 e12 = mul(sqr_n(e6, 6), e6);
    const u64 e24 = mul(sqr_n(e12, 12), e12);
    const u64 e30 = mul(sqr_n(e24, 6), e6);
    const u64 e31 = mul(sqr(e30), a);
    const u64 e32 = mul(sqr(e31), a);
    return mul(sqr_n(e31, 33), e32);
}

}  // namespace gl
