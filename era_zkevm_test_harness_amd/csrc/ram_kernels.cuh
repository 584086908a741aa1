// ram_kernels.cuh — device kernels of the RAM-permutation witness path (gfx950).
//
// Reference functions replaced (paths relative to the reference root):
//   k_encode_mem        MemoryQuery::encoding_witness           circuit_encodings/src/memory_query.rs:24-118
//   k_chain_full        FullWidthQueueSimulator::push chain     circuit_encodings/src/lib.rs:391-429
//                       (unsorted queue: src/witness/oracle.rs:894-903; sorted: W/ram_permutation.rs:61-71)
//   k_fs_challenges     produce_fs_challenges                   src/witness/utils.rs:498-550
//   k_gp_local/_tiles/_apply  compute_grand_product_chains      src/witness/utils.rs:554-697
//   k_ram_sort_keys / k_gather_queries   par_sort_by            W/ram_permutation.rs:48-53
//   k_ram_instances     per-instance FSM snapshots              W/ram_permutation.rs:239-453
#pragma once
#include "../../include/zkw_types.h"
#include "poseidon2.cuh"

namespace zkw {
using gl::u32;
using gl::u64;

// ------------------------------------------------------------------------------------------------
// K1: encode memory queries. One query per lane: 48 B in (3 x 16 B), 64 B out (4 x 16 B), both fully
// used cache lines. AoS output [n][8] is what the queue chain and the grand product consume.
__device__ __forceinline__ void encode_mem_query(const zkw_mem_query& q, u64 out[8]) {
    const u32* v = q.value;
    out[0] = q.timestamp;
    out[1] = q.page;
    out[2] = (u64)q.index | ((u64)(q.rw_flag ? 1 : 0) << 32) | ((u64)(q.value_is_pointer ? 1 : 0) << 33);
    // bytes of limbs 5,6,7 ride in bits 32..55 of limbs 0..3 (memory_query.rs:53-113)
    u64 b5 = v[5], b6 = v[6], b7 = v[7];
    out[3] = (u64)v[0] | ((b5 & 0xFFFFFF) << 32);
    out[4] = (u64)v[1] | ((b5 >> 24) << 32) | ((b6 & 0xFFFF) << 40);
    out[5] = (u64)v[2] | ((b6 >> 16) << 32) | ((b7 & 0xFF) << 48);
    out[6] = (u64)v[3] | ((b7 >> 8) << 32);
    out[7] = v[4];
}

static __device__ __forceinline__ void k_encode_mem(const VB& vb, const zkw_mem_query* __restrict__ q, size_t n,
                                                    u64* __restrict__ enc) {
    size_t i = (size_t)vb.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)vb.nx * blockDim.x;
    for (; i < n; i += stride) {
        const uint4* src = reinterpret_cast<const uint4*>(q + i);
        uint4 w0 = src[0], w1 = src[1], w2 = src[2];
        zkw_mem_query m;
        uint4* dst = reinterpret_cast<uint4*>(&m);
        dst[0] = w0; dst[1] = w1; dst[2] = w2;
        u64 e[8];
        encode_mem_query(m, e);
        ulonglong2* o = reinterpret_cast<ulonglong2*>(enc + 8 * i);
        o[0] = make_ulonglong2(e[0], e[1]); o[1] = make_ulonglong2(e[2], e[3]);
        o[2] = make_ulonglong2(e[4], e[5]); o[3] = make_ulonglong2(e[6], e[7]);
    }
}

// ------------------------------------------------------------------------------------------------
// K2: full-width queue chains. One chain per 16-lane DPP row (4 chains per wave, 1 wave per block so
// that every chain gets its own SIMD issue slot when chains are few). tails[i] = permute(enc[i] ||
// capacity(tails[i-1])). The absorb loads are prefetched one item ahead; stores are fire-and-forget.
struct ChainJob {
    const u64* enc;      // [n][8], or nullptr: the items are memory queries, encoded on the fly from `q`
    u64* tails;          // [n][12], or nullptr when only the compact outputs below are wanted
    const u64* tail_in;  // [12] or nullptr (= zeros)
    u64 n;
    // compact outputs (RAM builder): the next absorption only needs the capacity words of a tail, and any full tail
    // is one permutation away from (enc[i], caps[i-1]); full tails are kept at the instance boundaries only
    u64* caps;           // [n][4] elements 8..11 of every tail, or nullptr
    u64* marks;          // [ceil(n / period)][12] full tails at items period-1, 2*period-1, ... and n-1, or nullptr
    u64 period;
    const zkw_mem_query* q;  // [n] when enc == nullptr (the RAM builder keeps no encodings: 48 B per item instead of 64)
    const u32* perm;         // [n] or nullptr: item i is q[perm[i]] (the sorted queue as a permutation of the batch: q is
                             // then the batch's base pointer; no sorted copy of the queries is kept)
};

// a memory query as three 16-byte words, and word g (0..7) of its encoding (memory_query.rs:24-118)
struct RawQuery { uint4 a, b, c; };
__device__ __forceinline__ RawQuery load_raw_query(const zkw_mem_query* q) {
    const uint4* src = reinterpret_cast<const uint4*>(q);
    RawQuery r;
    r.a = src[0]; r.b = src[1]; r.c = src[2];
    return r;
}
__device__ __forceinline__ void encode_raw_query(const RawQuery& r, u64 e[8]) {
    zkw_mem_query m;
    uint4* dm = reinterpret_cast<uint4*>(&m);
    dm[0] = r.a; dm[1] = r.b; dm[2] = r.c;
    encode_mem_query(m, e);
}
__device__ __forceinline__ u64 pick8(const u64 e[8], int g) {
    u64 v = e[0];
#pragma unroll
    for (int k = 1; k < 8; k++) v = g == k ? e[k] : v;
    return v;
}

// The stores of item i are issued at the top of iteration i + 1, before the prefetch of item i + 2: gfx9 counts
// loads and stores in one in-order vmcnt, so consuming the prefetched encoding waits for every earlier store too.
// Stored late, those stores are a whole permutation (>= 10 us) old by then; stored right after the permutation their
// write latency was exposed once per step (15 -> 22.5 us per step beyond ~8.5 k concurrent chains).
struct ChainOut {
    u64* tails;
    u64* caps;
    u64* marks;
};

// WAVES = 1: one wave per workgroup. WAVES = 4 (`k_chain_full_x4`, the chain service's launches): the four waves of a workgroup go to
// the four SIMDs of ONE CU and the launch asks for more than half of a CU's LDS (unused), so that no two chain waves — of this launch or
// of another chain launch that runs next to it — share a SIMD (as `k_chain_full_q4x4` below; measured with 96 blocks in flight: three
// concurrent one-wave launches ran 1.4 - 1.5 x as long as alone).
template <int WAVES>
static __device__ __forceinline__ void chain_full_body(const ChainJob* __restrict__ jobs, int n_jobs) {
    __builtin_amdgcn_s_setprio(3);  // a serial chain is latency-bound: its wave issues before the fill waves sharing the SIMD
    const int lane = threadIdx.x & 63, g = lane & 15;
    const int chain = (blockIdx.x * WAVES + (int)(threadIdx.x >> 6)) * 4 + (lane >> 4);
    p2::Coop co;
    co.init(g);
    ChainJob job;
    memset(&job, 0, sizeof job);
    if (chain < n_jobs) job = jobs[chain];
    u64 x = (co.active && job.tail_in) ? job.tail_in[g] : 0;
    const bool absorbs = g < 8;
    const bool from_q = job.enc == nullptr;
    RawQuery rq_next;
    rq_next.a = rq_next.b = rq_next.c = make_uint4(0, 0, 0, 0);
    u64 e_next = 0;
    u64 at_next = 0;  // index (into q) of item i + 2, fetched one iteration before its query
    if (absorbs && job.n > 0) {
        if (from_q) rq_next = load_raw_query(job.q + (job.perm ? job.perm[0] : 0)); else e_next = job.enc[g];
    }
    if (absorbs && from_q && job.n > 1) at_next = job.perm ? job.perm[1] : 1;
    u64 next_mark = job.marks ? job.period : ~0ull, mark_idx = 0;  // item count at which the next full tail is kept
    u64 pend = 0, pend_i = 0;  // canonical tail of the previous item, not stored yet
    bool have_pend = false, pend_mark = false;
    auto flush = [&]() {
        if (!have_pend) return;
        if (co.active && job.tails) job.tails[12 * pend_i + g] = pend;
        if (g >= 8 && g < 12 && job.caps) job.caps[4 * pend_i + (g - 8)] = pend;
        if (pend_mark) {
            if (co.active) job.marks[12 * mark_idx + g] = pend;
            mark_idx++;
        }
        have_pend = false;
    };
    for (u64 i = 0; __any(i < job.n); i++) {
        const bool live = i < job.n;
        u64 e = e_next;
        if (from_q) {
            u64 ew[8];
            encode_raw_query(rq_next, ew);
            e = pick8(ew, g);
        }
        flush();
        if (absorbs && i + 1 < job.n) {
            if (from_q) {
                rq_next = load_raw_query(job.q + at_next);
                if (i + 2 < job.n) at_next = job.perm ? job.perm[i + 2] : i + 2;
            } else {
                e_next = job.enc[8 * (i + 1) + g];
            }
        }
        u64 y = co.permute(absorbs ? e : x);  // AbsorptionModeOverwrite
        if (live) {
            x = y;
            pend = gl::canon(y);
            pend_i = i;
            have_pend = true;
            pend_mark = job.marks && (i + 1 == next_mark || i + 1 == job.n);
            if (pend_mark) next_mark += job.period;
        }
    }
    flush();
}
static __global__ __launch_bounds__(64) void k_chain_full(const ChainJob* __restrict__ jobs, int n_jobs) { chain_full_body<1>(jobs, n_jobs); }
static __global__ __launch_bounds__(256) void k_chain_full_x4(const ChainJob* __restrict__ jobs, int n_jobs) { chain_full_body<4>(jobs, n_jobs); }

// Quad form: 16 chains per wave (p2::Coop4). Lane j of a quad loads enc[j], enc[4+j] and stores tails[j],
// tails[4+j], tails[8+j]: 32 contiguous bytes per quad per access.
// WAVES = 1: one wave per workgroup (placed wherever a wave slot is free). WAVES = 4 (`k_chain_full_q4x4`): the four waves of a
// workgroup go to the four SIMDs of ONE CU, and the launch asks for more than half of a CU's LDS (unused) so that a CU takes one such
// workgroup: every chain wave of a launch of <= 16 384 chains has a SIMD without another chain wave on it, whatever else (the
// other pipeline's fills) occupies the chip when the launch arrives — see dev_chains.
template <int WAVES>
static __device__ __forceinline__ void chain_full_q4_body(const ChainJob* __restrict__ jobs, int n_jobs) {
    __builtin_amdgcn_s_setprio(3);  // a serial chain is latency-bound: its wave issues before the fill waves sharing the SIMD
    const int lane = threadIdx.x & 63, j = lane & 3;
    const int chain = blockIdx.x * (16 * WAVES) + (threadIdx.x >> 2);
    p2::Coop4 co;
    co.init(j);
    ChainJob job;
    memset(&job, 0, sizeof job);
    if (chain < n_jobs) job = jobs[chain];
    u64 x[3];
#pragma unroll
    for (int c = 0; c < 3; c++) x[c] = job.tail_in ? job.tail_in[4 * c + j] : 0;
    u64 e0 = 0, e1 = 0;
    const bool from_q = job.enc == nullptr;
    RawQuery rq_next;
    rq_next.a = rq_next.b = rq_next.c = make_uint4(0, 0, 0, 0);
    u64 at_next = 0;  // index (into q) of item i + 2, fetched one iteration before its query
    if (job.n > 0) {
        if (from_q) rq_next = load_raw_query(job.q + (job.perm ? job.perm[0] : 0)); else { e0 = job.enc[j]; e1 = job.enc[4 + j]; }
    }
    if (from_q && job.n > 1) at_next = job.perm ? job.perm[1] : 1;
    u64 next_mark = job.marks ? job.period : ~0ull, mark_idx = 0;
    u64 pend[3] = {0, 0, 0}, pend_i = 0;
    bool have_pend = false, pend_mark = false;
    auto flush = [&]() {
        if (!have_pend) return;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            if (job.tails) job.tails[12 * pend_i + 4 * c + j] = pend[c];
            if (c == 2 && job.caps) job.caps[4 * pend_i + j] = pend[c];
            if (pend_mark) job.marks[12 * mark_idx + 4 * c + j] = pend[c];
        }
        if (pend_mark) mark_idx++;
        have_pend = false;
    };
    for (u64 i = 0; __any(i < job.n); i++) {
        const bool live = i < job.n;
        if (from_q) {
            u64 ew[8];
            encode_raw_query(rq_next, ew);
            e0 = pick8(ew, j);
            e1 = pick8(ew, 4 + j);
        }
        u64 y[3] = {e0, e1, x[2]};  // AbsorptionModeOverwrite: rate part replaced, capacity kept
        flush();
        if (i + 1 < job.n) {
            if (from_q) {
                rq_next = load_raw_query(job.q + at_next);
                if (i + 2 < job.n) at_next = job.perm ? job.perm[i + 2] : i + 2;
            } else {
                e0 = job.enc[8 * (i + 1) + j]; e1 = job.enc[8 * (i + 1) + 4 + j];
            }
        }
        co.permute(y);
        if (live) {
#pragma unroll
            for (int c = 0; c < 3; c++) {
                x[c] = y[c];
                pend[c] = gl::canon(y[c]);
            }
            pend_i = i;
            have_pend = true;
            pend_mark = job.marks && (i + 1 == next_mark || i + 1 == job.n);
            if (pend_mark) next_mark += job.period;
        }
    }
    flush();
}
static __global__ __launch_bounds__(64) void k_chain_full_q4(const ChainJob* __restrict__ jobs, int n_jobs) { chain_full_q4_body<1>(jobs, n_jobs); }
static __global__ __launch_bounds__(256) void k_chain_full_q4x4(const ChainJob* __restrict__ jobs, int n_jobs) { chain_full_q4_body<4>(jobs, n_jobs); }

// Pair form: 32 chains per wave (p2::Coop2). Lane j of a pair holds elements 4c + 2j, 4c + 2j + 1: it loads / stores 16
// contiguous bytes per block of four. Half the waves of the quad form for the same queues and ~half its wave-instructions per
// permutation: what a launch of tens of thousands of queues takes away from the trace fills it overlaps.
static __global__ __launch_bounds__(64) void k_chain_full_p2(const ChainJob* __restrict__ jobs, int n_jobs) {
    __builtin_amdgcn_s_setprio(3);  // a serial chain is latency-bound: its wave issues before the fill waves sharing the SIMD
    const int lane = threadIdx.x & 63, j = lane & 1;
    const int chain = blockIdx.x * 32 + (lane >> 1);
    p2::Coop2 co;
    co.init(j);
    ChainJob job;
    memset(&job, 0, sizeof job);
    if (chain < n_jobs) job = jobs[chain];
    u64 x[6];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        x[2 * c] = job.tail_in ? job.tail_in[4 * c + 2 * j] : 0;
        x[2 * c + 1] = job.tail_in ? job.tail_in[4 * c + 2 * j + 1] : 0;
    }
    u64 e[4] = {0, 0, 0, 0};  // rate elements 2j, 2j + 1, 4 + 2j, 5 + 2j of the next item
    const bool from_q = job.enc == nullptr;
    RawQuery rq_next;
    rq_next.a = rq_next.b = rq_next.c = make_uint4(0, 0, 0, 0);
    u64 at_next = 0;  // index (into q) of item i + 2, fetched one iteration before its query
    auto load_enc = [&](u64 i) {
        const ulonglong2 lo = *reinterpret_cast<const ulonglong2*>(job.enc + 8 * i + 2 * j), hi = *reinterpret_cast<const ulonglong2*>(job.enc + 8 * i + 4 + 2 * j);
        e[0] = lo.x; e[1] = lo.y; e[2] = hi.x; e[3] = hi.y;
    };
    if (job.n > 0) {
        if (from_q) rq_next = load_raw_query(job.q + (job.perm ? job.perm[0] : 0)); else load_enc(0);
    }
    if (from_q && job.n > 1) at_next = job.perm ? job.perm[1] : 1;
    u64 next_mark = job.marks ? job.period : ~0ull, mark_idx = 0;
    u64 pend[6] = {0, 0, 0, 0, 0, 0}, pend_i = 0;
    bool have_pend = false, pend_mark = false;
    auto flush = [&]() {  // the stores of item i go out at the top of iteration i + 1 (see k_chain_full)
        if (!have_pend) return;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const ulonglong2 v = make_ulonglong2(pend[2 * c], pend[2 * c + 1]);
            if (job.tails) *reinterpret_cast<ulonglong2*>(job.tails + 12 * pend_i + 4 * c + 2 * j) = v;
            if (c == 2 && job.caps) *reinterpret_cast<ulonglong2*>(job.caps + 4 * pend_i + 2 * j) = v;
            if (pend_mark) *reinterpret_cast<ulonglong2*>(job.marks + 12 * mark_idx + 4 * c + 2 * j) = v;
        }
        if (pend_mark) mark_idx++;
        have_pend = false;
    };
    for (u64 i = 0; __any(i < job.n); i++) {
        const bool live = i < job.n;
        if (from_q) {
            u64 ew[8];
            encode_raw_query(rq_next, ew);
            e[0] = j ? ew[2] : ew[0];
            e[1] = j ? ew[3] : ew[1];
            e[2] = j ? ew[6] : ew[4];
            e[3] = j ? ew[7] : ew[5];
        }
        u64 y[6] = {e[0], e[1], e[2], e[3], x[4], x[5]};  // AbsorptionModeOverwrite: rate part replaced, capacity kept
        flush();
        if (i + 1 < job.n) {
            if (from_q) {
                rq_next = load_raw_query(job.q + at_next);
                if (i + 2 < job.n) at_next = job.perm ? job.perm[i + 2] : i + 2;
            } else {
                load_enc(i + 1);
            }
        }
        co.permute(y);
        if (live) {
#pragma unroll
            for (int c = 0; c < 6; c++) {
                x[c] = y[c];
                pend[c] = gl::canon(y[c]);
            }
            pend_i = i;
            have_pend = true;
            pend_mark = job.marks && (i + 1 == next_mark || i + 1 == job.n);
            if (pend_mark) next_mark += job.period;
        }
    }
    flush();
}

// Lane form: ONE CHAIN PER LANE, 64 chains per wave (p2::permute, the whole state in the lane's registers). The
// cooperative forms above buy latency with idle lanes (during the 22 partial rounds only one S-box per state is live:
// 506 wave-instructions per permutation in the quad form); per lane a permutation costs ~210. With tens of thousands
// of queues in one launch (bench.py: 2 x 14 440) latency per step is irrelevant and VALU issue slots are what the
// chains take away from the trace fills they overlap: 29 k chains are 451 waves, fewer than half the SIMDs, instead of
// 1 805 waves on every SIMD twice. Accesses are per-lane (48 B in, 32 B out per step, each lane on its own stream): a few
// hundred bytes per wave every ~25 us.
static __global__ __launch_bounds__(64) void k_chain_full_lane(const ChainJob* __restrict__ jobs, int n_jobs) {
    __builtin_amdgcn_s_setprio(3);  // a serial chain is latency-bound: its wave issues before the fill waves sharing the SIMD
    const int chain = blockIdx.x * 64 + (threadIdx.x & 63);
    ChainJob job;
    memset(&job, 0, sizeof job);
    if (chain < n_jobs) job = jobs[chain];
    u64 s[12];
#pragma unroll
    for (int k = 0; k < 12; k++) s[k] = job.tail_in ? job.tail_in[k] : 0;
    const bool from_q = job.enc == nullptr;
    RawQuery rq_next;
    rq_next.a = rq_next.b = rq_next.c = make_uint4(0, 0, 0, 0);
    u64 e_next[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    u64 at_next = 0;
    if (job.n > 0) {
        if (from_q) rq_next = load_raw_query(job.q + (job.perm ? job.perm[0] : 0));
        else {
            const ulonglong2* src = reinterpret_cast<const ulonglong2*>(job.enc);
#pragma unroll
            for (int k = 0; k < 4; k++) { const ulonglong2 v = src[k]; e_next[2 * k] = v.x; e_next[2 * k + 1] = v.y; }
        }
    }
    if (from_q && job.n > 1) at_next = job.perm ? job.perm[1] : 1;
    u64 next_mark = job.marks ? job.period : ~0ull, mark_idx = 0;
    u64 pend[12], pend_i = 0;
#pragma unroll
    for (int k = 0; k < 12; k++) pend[k] = 0;
    bool have_pend = false, pend_mark = false;
    auto flush = [&]() {  // the stores of item i go out at the top of iteration i + 1 (see k_chain_full)
        if (!have_pend) return;
        if (job.tails) {
            ulonglong2* d = reinterpret_cast<ulonglong2*>(job.tails + 12 * pend_i);
#pragma unroll
            for (int k = 0; k < 6; k++) d[k] = make_ulonglong2(pend[2 * k], pend[2 * k + 1]);
        }
        if (job.caps) {
            ulonglong2* d = reinterpret_cast<ulonglong2*>(job.caps + 4 * pend_i);
            d[0] = make_ulonglong2(pend[8], pend[9]);
            d[1] = make_ulonglong2(pend[10], pend[11]);
        }
        if (pend_mark) {
            ulonglong2* d = reinterpret_cast<ulonglong2*>(job.marks + 12 * mark_idx);
#pragma unroll
            for (int k = 0; k < 6; k++) d[k] = make_ulonglong2(pend[2 * k], pend[2 * k + 1]);
            mark_idx++;
        }
        have_pend = false;
    };
    for (u64 i = 0; __any(i < job.n); i++) {
        const bool live = i < job.n;
        u64 e[8];
        if (from_q) encode_raw_query(rq_next, e);
        else {
#pragma unroll
            for (int k = 0; k < 8; k++) e[k] = e_next[k];
        }
        flush();
        if (i + 1 < job.n) {
            if (from_q) {
                rq_next = load_raw_query(job.q + at_next);
                if (i + 2 < job.n) at_next = job.perm ? job.perm[i + 2] : i + 2;
            } else {
                const ulonglong2* src = reinterpret_cast<const ulonglong2*>(job.enc + 8 * (i + 1));
#pragma unroll
                for (int k = 0; k < 4; k++) { const ulonglong2 v = src[k]; e_next[2 * k] = v.x; e_next[2 * k + 1] = v.y; }
            }
        }
        if (live) {
#pragma unroll
            for (int k = 0; k < 8; k++) s[k] = e[k];  // AbsorptionModeOverwrite
            p2::permute(s);
#pragma unroll
            for (int k = 0; k < 12; k++) pend[k] = gl::canon(s[k]);
            pend_i = i;
            have_pend = true;
            pend_mark = job.marks && (i + 1 == next_mark || i + 1 == job.n);
            if (pend_mark) next_mark += job.period;
        }
    }
    flush();
}

// ------------------------------------------------------------------------------------------------
// K5: Fiat-Shamir challenges, one job per lane (a handful of permutations; latency-irrelevant).
struct FsJob {
    const u64* tail_u;  // [state_w]
    const u64* tail_s;  // [state_w]
    u32 len_u, len_s;
    u64* out;           // [2][n_chal]
};

// No per-lane arrays with run-time indices: they would live in scratch memory, and every HSA queue that ever ran the
// kernel keeps scratch-per-lane x every wave slot of the chip (117 MB for 224 B per lane) out of the runtime's 4 GB
// scratch aperture; with 32 hardware queues in use that exhausted it (HSA_STATUS_ERROR_OUT_OF_RESOURCES).
static __device__ __forceinline__ void k_fs_challenges(const VB& vb, const FsJob* __restrict__ jobs, int n_jobs, int state_w, int n_chal) {
    int j = vb.x * blockDim.x + threadIdx.x;
    if (j >= n_jobs) return;
    const FsJob job = jobs[j];
    const int m = 2 * state_w + 2;
    // element e of the transcript: tail_u, len_u, tail_s, len_s
    auto elem = [&](int e) -> u64 {
        if (e >= m) return 0;
        if (e < state_w) return job.tail_u[e];
        if (e == state_w) return job.len_u;  // < p
        if (e < 2 * state_w + 1) return job.tail_s[e - state_w - 1];
        return job.len_s;
    };
    u64 s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = 0;
    s[11] = (u64)m;  // specialize_for_len
    for (int i = 0; i < m; i += 8) {
#pragma unroll
        for (int k = 0; k < 8; k++) s[k] = elem(i + k);
        p2::permute(s);
    }
    int can_take = 8;
    for (int rep = 0; rep < 2; rep++) {
        job.out[rep * n_chal] = 1;
        for (int k = 1; k < n_chal; k++) {
            if (can_take == 0) { p2::permute(s); can_take = 8; }
            u64 v = 0;
#pragma unroll
            for (int q = 0; q < 8; q++) v = (q == 8 - can_take) ? s[q] : v;
            job.out[rep * n_chal + k] = gl::canon(v);
            can_take--;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K6: grand-product chains. A "segment" is one side (lhs or rhs) of one sorter: rows[n][W] -> z[rep][n]
// for both repetitions in ONE pass over the rows (the reference reads them once per repetition).
// Three launches: tile-local inclusive scan (+ tile aggregates), scan of the aggregates per segment,
// tile-prefix application. Tiles are GP_TILE rows; a block is 256 lanes = 4 waves.
constexpr int GP_BLOCK = 256;
constexpr int GP_SLABS = 4;
constexpr int GP_TILE = GP_BLOCK * GP_SLABS;

struct GpSeg {
    const u64* rows;   // [n][W], or nullptr: W = 8 and the rows are the encodings of the memory queries `mem_q`
    u64* z;            // [n_reps][n]
    const u64* chal;   // [n_reps][W+1]
    u64 n;
    u32 first_tile;    // index of this segment's first tile in the launch
    u32 n_tiles;
    const zkw_mem_query* mem_q;  // [n] when rows == nullptr
    const u32* perm;             // [n] or nullptr: row i is the encoding of mem_q[perm[i]]
};
struct GpTile {
    u32 seg;
    u32 tile;  // tile index inside the segment
};

// inclusive product scan across the 64 lanes of a wave
__device__ __forceinline__ u64 wave_scan_mul(u64 v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        u64 o = __shfl_up(v, d, 64);
        if (lane >= d) v = gl::mul(v, o);
    }
    return v;
}

template <int W, int REPS>
static __device__ __forceinline__ void k_gp_local(const VB& vb, const GpSeg* __restrict__ segs,
                                                       const GpTile* __restrict__ tiles,
                                                       u64* __restrict__ tile_aggr /* [n_tiles_total][REPS] */) {
    __shared__ u64 sh_ch[REPS][W + 1];
    __shared__ u64 sh_wave[REPS][GP_BLOCK / 64];
    const GpTile t = tiles[vb.x];
    const GpSeg seg = segs[t.seg];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int k = tid; k < REPS * (W + 1); k += GP_BLOCK) sh_ch[k / (W + 1)][k % (W + 1)] = seg.chal[k];
    __syncthreads();
    u64 carry[REPS];
#pragma unroll
    for (int r = 0; r < REPS; r++) carry[r] = 1;
    const u64 base = (u64)t.tile * GP_TILE;
    for (int slab = 0; slab < GP_SLABS; slab++) {
        const u64 row = base + (u64)slab * GP_BLOCK + tid;
        const bool live = row < seg.n;
        u64 term[REPS];
#pragma unroll
        for (int r = 0; r < REPS; r++) term[r] = 1;  // neutral for rows past the end
        if (live) {
            u64 e[W];
            if (W == 8 && seg.rows == nullptr) {
                u64 e8[8];
                encode_raw_query(load_raw_query(seg.mem_q + (seg.perm ? seg.perm[row] : row)), e8);
#pragma unroll
                for (int k = 0; k < (W < 8 ? W : 8); k++) e[k] = e8[k];
            } else {
                const ulonglong2* src = reinterpret_cast<const ulonglong2*>(seg.rows + row * W);
#pragma unroll
                for (int k = 0; k < W / 2; k++) { ulonglong2 w = src[k]; e[2 * k] = w.x; e[2 * k + 1] = w.y; }
            }
#pragma unroll
            for (int r = 0; r < REPS; r++) {
                u64 acc = sh_ch[r][W];
#pragma unroll
                for (int k = 0; k < W; k++) acc = gl::add(acc, gl::mul(e[k], sh_ch[r][k]));
                term[r] = acc;
            }
        }
#pragma unroll
        for (int r = 0; r < REPS; r++) {
            u64 v = wave_scan_mul(term[r], lane);
            if (lane == 63) sh_wave[r][wave] = v;
            __syncthreads();
            u64 pre = carry[r];
            for (int w = 0; w < wave; w++) pre = gl::mul(pre, sh_wave[r][w]);
            v = gl::mul(v, pre);
            if (live) seg.z[(u64)r * seg.n + row] = v;  // weak; canonicalised by k_gp_apply
            u64 tot = carry[r];
            for (int w = 0; w < GP_BLOCK / 64; w++) tot = gl::mul(tot, sh_wave[r][w]);
            carry[r] = tot;
            __syncthreads();
        }
    }
    if (tid == 0) {
#pragma unroll
        for (int r = 0; r < REPS; r++) tile_aggr[(u64)vb.x * REPS + r] = carry[r];
    }
}

// exclusive scan of the tile aggregates inside each segment: one lane per (segment, repetition)
template <int REPS>
static __device__ __forceinline__ void k_gp_tiles(const VB& vb, const GpSeg* __restrict__ segs, int n_segs, u64* __restrict__ tile_aggr) {
    int j = vb.x * blockDim.x + threadIdx.x;
    if (j >= n_segs * REPS) return;
    const GpSeg seg = segs[j / REPS];
    const int r = j % REPS;
    u64 acc = 1;
    for (u32 t = 0; t < seg.n_tiles; t++) {
        u64* p = tile_aggr + (u64)(seg.first_tile + t) * REPS + r;
        u64 v = *p;
        *p = acc;
        acc = gl::mul(acc, v);
    }
}

template <int REPS>
static __device__ __forceinline__ void k_gp_apply(const VB& vb, const GpSeg* __restrict__ segs,
                                                       const GpTile* __restrict__ tiles,
                                                       const u64* __restrict__ tile_prefix) {
    const GpTile t = tiles[vb.x];
    const GpSeg seg = segs[t.seg];
    const u64 base = (u64)t.tile * GP_TILE;
#pragma unroll
    for (int r = 0; r < REPS; r++) {
        const u64 pre = tile_prefix[(u64)vb.x * REPS + r];
        u64* z = seg.z + (u64)r * seg.n;
        for (int slab = 0; slab < GP_SLABS; slab++) {
            const u64 row = base + (u64)slab * GP_BLOCK + threadIdx.x;
            if (row < seg.n) z[row] = gl::canon(gl::mul(z[row], pre));
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K7 support: sort keys and the gather that applies the sorting permutation.
// Sorting order (W/ram_permutation.rs:50-53): (page, index) then timestamp, stable.
static __device__ __forceinline__ void k_ram_sort_keys(const VB& vb, const zkw_mem_query* __restrict__ q, size_t n, u32* __restrict__ ts,
                                u64* __restrict__ cell, u32* __restrict__ iota, const u64* __restrict__ seg_off,
                                int n_segs) {
    size_t i = (size_t)vb.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    ts[i] = q[i].timestamp;
    cell[i] = ((u64)q[i].page << 32) | q[i].index;
    iota[i] = (u32)i;  // global position; segments never mix, so this is also the stable tiebreak
    (void)seg_off; (void)n_segs;
}

static __device__ __forceinline__ void k_gather_u32_by_u32(const VB& vb, const u32* __restrict__ src, const u32* __restrict__ idx, size_t n,
                                    u32* __restrict__ dst) {
    size_t i = (size_t)vb.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}

static __device__ __forceinline__ void k_gather_u64_by_u32(const VB& vb, const u64* __restrict__ src, const u32* __restrict__ idx, size_t n,
                                    u64* __restrict__ dst) {
    size_t i = (size_t)vb.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}

// sorted_q[i] = q[perm[i]] and its encoding in the same pass (the sorted side never needs the
// un-encoded query again except for the FSM snapshots, which read sorted_q).
static __device__ __forceinline__ void k_gather_encode(const VB& vb, const zkw_mem_query* __restrict__ q,
                                                       const u32* __restrict__ perm, size_t n,
                                                       zkw_mem_query* __restrict__ sorted_q,
                                                       u64* __restrict__ sorted_enc) {
    size_t i = (size_t)vb.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint4* src = reinterpret_cast<const uint4*>(q + perm[i]);
    uint4 w0 = src[0], w1 = src[1], w2 = src[2];
    uint4* dq = reinterpret_cast<uint4*>(sorted_q + i);
    dq[0] = w0; dq[1] = w1; dq[2] = w2;
    zkw_mem_query m;
    uint4* dm = reinterpret_cast<uint4*>(&m);
    dm[0] = w0; dm[1] = w1; dm[2] = w2;
    if (!sorted_enc) return;
    u64 e[8];
    encode_mem_query(m, e);
    ulonglong2* o = reinterpret_cast<ulonglong2*>(sorted_enc + 8 * i);
    o[0] = make_ulonglong2(e[0], e[1]); o[1] = make_ulonglong2(e[2], e[3]);
    o[2] = make_ulonglong2(e[4], e[5]); o[3] = make_ulonglong2(e[6], e[7]);
}

// ------------------------------------------------------------------------------------------------
// a10: per-instance records (W/ram_permutation.rs:239-453). One block per memory block (queue);
// the instance loop is sequential in the reference because of the running FSM values, but every
// field is a pure function of the block-wide arrays, so each instance is filled independently.
struct RamBlock {
    const zkw_mem_query* sorted_q;  // the BATCH's queries; sorted item i of this block is sorted_q[sorted_perm[i]]
    const u32* sorted_perm;         // [n]
    const u64* u_marks;             // [n_instances][12] unsorted queue tail after the last item of each instance
    const u64* s_marks;             // [n_instances][12] the same for the sorted queue
    const u64* lhs_z;               // [2][n]
    const u64* rhs_z;               // [2][n]
    zkw_ram_instance* instances;    // [ceil(n/capacity)]
    u32* nondet_prefix;             // [n_instances] scratch: nondeterministic writes per chunk
    u64 n;
    u32 capacity;
    u32 num_nondet_heap_queries;
};

__device__ __forceinline__ void copy12(u64* dst, const u64* src) {
    for (int k = 0; k < 12; k++) dst[k] = src[k];
}

// pass 1: count nondeterministic writes per chunk (rw && ts == 0 && page == BOOTLOADER_HEAP_PAGE)
static __device__ __forceinline__ void k_ram_count_nondet(const VB& vb, const RamBlock* __restrict__ blocks) {
    const RamBlock b = blocks[vb.y];
    const u64 n_inst = (b.n + b.capacity - 1) / b.capacity;
    __shared__ u32 sh[4];
    for (u64 inst = vb.x; inst < n_inst; inst += vb.nx) {
        const u64 lo = inst * b.capacity, hi = lo + b.capacity < b.n ? lo + b.capacity : b.n;
        u32 cnt = 0;
        for (u64 i = lo + threadIdx.x; i < hi; i += blockDim.x) {
            const zkw_mem_query* q = b.sorted_q + b.sorted_perm[i];
            cnt += (q->rw_flag && q->timestamp == 0 && q->page == ZKW_BOOTLOADER_HEAP_PAGE) ? 1u : 0u;
        }
        for (int d = 32; d > 0; d >>= 1) cnt += __shfl_down(cnt, d, 64);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = cnt;
        __syncthreads();
        if (threadIdx.x == 0) b.nondet_prefix[inst] = sh[0] + sh[1] + sh[2] + sh[3];
        __syncthreads();
    }
}

// pass 2: one lane per instance
static __device__ __forceinline__ void k_ram_instances(const VB& vb, const RamBlock* __restrict__ blocks) {
    const RamBlock b = blocks[vb.y];
    const u64 n_inst = (b.n + b.capacity - 1) / b.capacity;
    const u64 idx = (u64)vb.x * blockDim.x + threadIdx.x;
    if (idx >= n_inst) return;
    const u64 n = b.n, lo = idx * b.capacity, hi = lo + b.capacity < n ? lo + b.capacity : n;
    zkw_ram_instance& w = b.instances[idx];  // filled in place: a local copy would live in scratch memory
    memset(&w, 0, sizeof w);
    w.start_flag = idx == 0;
    w.completion_flag = idx == n_inst - 1;
    w.first_item = lo;
    w.num_items = hi - lo;
    const u64* u_final = b.u_marks + 12 * (n_inst - 1);
    const u64* s_final = b.s_marks + 12 * (n_inst - 1);
    copy12(w.unsorted_queue_initial_state.tail, u_final);
    w.unsorted_queue_initial_state.length = (u32)n;
    copy12(w.sorted_queue_initial_state.tail, s_final);
    w.sorted_queue_initial_state.length = (u32)n;
    w.non_deterministic_bootloader_memory_snapshot_length = b.num_nondet_heap_queries;

    u32 nondet_before = 0;
    for (u64 k = 0; k < idx; k++) nondet_before += b.nondet_prefix[k];

    // FSM output of chunk j (without the padding reset) is a function of item `end_j - 1`
    auto fill = [&](zkw_ram_fsm& f, u64 end /* items consumed so far, > 0 */, u32 nondet) {
        const u64 l = end - 1;
        for (int r = 0; r < 2; r++) { f.lhs_accumulator[r] = b.lhs_z[r * n + l]; f.rhs_accumulator[r] = b.rhs_z[r * n + l]; }
        copy12(f.current_unsorted_queue_state.head, b.u_marks + 12 * (l / b.capacity));
        copy12(f.current_unsorted_queue_state.tail, u_final);
        f.current_unsorted_queue_state.length = (u32)(n - end);
        copy12(f.current_sorted_queue_state.head, b.s_marks + 12 * (l / b.capacity));
        copy12(f.current_sorted_queue_state.tail, s_final);
        f.current_sorted_queue_state.length = (u32)(n - end);
        const zkw_mem_query* q = b.sorted_q + b.sorted_perm[l];
        f.previous_sorting_key[0] = q->timestamp; f.previous_sorting_key[1] = q->index; f.previous_sorting_key[2] = q->page;
        f.previous_full_key[0] = q->index; f.previous_full_key[1] = q->page;
        for (int k = 0; k < 8; k++) f.previous_value[k] = q->value[k];
        f.previous_is_ptr = q->value_is_pointer ? 1 : 0;
        f.num_nondeterministic_writes = nondet;
    };
    if (idx == 0) {
        for (int r = 0; r < 2; r++) { w.hidden_fsm_input.lhs_accumulator[r] = 1; w.hidden_fsm_input.rhs_accumulator[r] = 1; }
    } else {
        fill(w.hidden_fsm_input, lo, nondet_before);
    }
    fill(w.hidden_fsm_output, hi, nondet_before + b.nondet_prefix[idx]);
    if ((hi - lo) % b.capacity != 0) {  // padding reset, W/ram_permutation.rs:414-432
        zkw_ram_fsm& f = w.hidden_fsm_output;
        for (int k = 0; k < 3; k++) f.previous_sorting_key[k] = 0;
        for (int k = 0; k < 2; k++) f.previous_full_key[k] = 0;
        for (int k = 0; k < 8; k++) f.previous_value[k] = 0;
        f.previous_is_ptr = 0;
    }
}


// ------------------------------------------------------------------------------------------------
// Full tails on demand: tails[i] = permute(enc[i] || caps[i-1]) (zero capacity at the first item of a queue).
// One item per lane; offsets[] are the queue boundaries inside the batch.
static __device__ __forceinline__ void k_tails_expand(const VB& vb, const u64* __restrict__ enc, const u64* __restrict__ caps,
                                                     const u64* __restrict__ offsets, int n_queues, size_t n,
                                                     u64* __restrict__ tails) {
    const size_t i = (size_t)vb.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int lo = 0, hi = n_queues;  // largest b with offsets[b] <= i
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (offsets[mid] <= i) lo = mid; else hi = mid;
    }
    u64 s[12];
#pragma unroll
    for (int k = 0; k < 8; k++) s[k] = enc[8 * i + k];
    const bool first = offsets[lo] == i;
#pragma unroll
    for (int k = 0; k < 4; k++) s[8 + k] = first ? 0 : caps[4 * (i - 1) + k];
    p2::permute(s);
#pragma unroll
    for (int k = 0; k < 12; k++) tails[12 * i + k] = gl::canon(s[k]);
}

}  // namespace zkw
