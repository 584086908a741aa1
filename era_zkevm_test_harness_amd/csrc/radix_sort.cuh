// radix_sort.cuh — the ordering step of the sorter circuits as this library's own kernels (round 6; rounds 1-5 called rocPRIM):
// a stable least-significant-digit radix sort of (key, u32 value) pairs, 8 bits per pass. Replaces rayon `par_sort_by` at
// src/witness/individual_circuits/ram_permutation.rs:48-53, sort_decommit_requests.rs, events_sort_dedup.rs, storage sorting
// (circuit_sequencer_api/src/sort_storage_access.rs:31-42). Stable passes => the result equals the reference's stable comparison sort.
//
// Written as kernel BODIES (zkw_launch.h): inside zkw_blocks_run the K blocks' sorts of a stage travel as one launch per kernel and pass.
//
// A pass over n pairs, tile = 4 waves x 8 rounds x 64 keys = 2 048 keys (30 KB of LDS: five workgroups per CU; 16 rounds — two workgroups per
// CU — ran the scatter of 1.9 G pairs at 1.8 TB/s), wave w of a tile owns the keys [512 w, 512 (w + 1)) of it:
//   k_rs_hist     per tile: 256 digit counts (LDS atomics) -> hist[digit][tile]
//   k_rs_scan_a   exclusive scan of hist (digit-major) in chunks of 16 384 entries, chunk totals aside
//   k_rs_scan_b   exclusive scan of the chunk totals (one workgroup)
//   k_rs_scatter  per tile: every key's rank among the tile's keys of its digit — ballot match inside a wave (8 ballots), wave counts
//                 through LDS, waves in order — + hist + chunk offset = its place; key and value written there.
// HBM traffic per pass and pair: keys read twice, key + value read and written once: 2 x sizeof(K) + 2 x (sizeof(K) + 4) bytes.
#pragma once
#include "zkw_ctx.h"

namespace zkw {

constexpr int RS_ROUNDS = 8, RS_WAVES = 4, RS_TILE = RS_WAVES * RS_ROUNDS * 64, RS_CHUNK = 16384;

template <class K>
static __device__ __forceinline__ unsigned rs_digit(K key, unsigned shift, unsigned mask) { return (unsigned)(key >> shift) & mask; }

template <class K>
static __device__ __forceinline__ void k_rs_hist(const VB& vb, const K* __restrict__ keys, size_t n, unsigned shift, unsigned mask, u32* __restrict__ hist, u32 n_tiles) {
    __shared__ u32 cnt[256];
    const int t = threadIdx.x;
    cnt[t] = 0;
    __syncthreads();
    const size_t base = (size_t)vb.x * RS_TILE;
#pragma unroll 4
    for (int r = 0; r < RS_TILE / 256; r++) {
        const size_t i = base + (size_t)r * 256 + t;
        if (i < n) atomicAdd(&cnt[rs_digit(keys[i], shift, mask)], 1u);
    }
    __syncthreads();
    hist[(size_t)t * n_tiles + vb.x] = cnt[t];
}

// in place: hist[i] <- exclusive prefix inside its chunk; chunk_tot[c] <- the chunk's total. 1024 threads x 16 entries.
static __device__ __forceinline__ void k_rs_scan_a(const VB& vb, u32* __restrict__ hist, size_t n_entries, u32* __restrict__ chunk_tot) {
    __shared__ u32 s_wave[16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const size_t base = (size_t)vb.x * RS_CHUNK + (size_t)t * 16;
    u32 v[16], sum = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        v[k] = base + k < n_entries ? hist[base + k] : 0;
        sum += v[k];
    }
    u32 incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const u32 y = __shfl_up(incl, off, 64);
        if (lane >= off) incl += y;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    u32 before = 0;
    for (int w = 0; w < wave; w++) before += s_wave[w];
    u32 run = before + incl - sum;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        if (base + k < n_entries) hist[base + k] = run;
        run += v[k];
    }
    if (t == 1023) chunk_tot[vb.x] = before + incl;
}

// exclusive scan of the chunk totals in place (one workgroup of 1024, any count)
static __device__ __forceinline__ void k_rs_scan_b(const VB& vb, u32* __restrict__ chunk_tot, u32 n_chunks) {
    __shared__ u32 s_wave[16];
    __shared__ u32 carry;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) carry = 0;
    __syncthreads();
    for (u32 base = 0; base < n_chunks; base += 1024) {
        const u32 i = base + t;
        const u32 v = i < n_chunks ? chunk_tot[i] : 0;
        u32 incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const u32 y = __shfl_up(incl, off, 64);
            if (lane >= off) incl += y;
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        u32 before = carry;
        for (int w = 0; w < wave; w++) before += s_wave[w];
        if (i < n_chunks) chunk_tot[i] = before + incl - v;
        __syncthreads();
        if (t == 1023) carry = before + incl;
        __syncthreads();
    }
}

template <class K>
static __device__ __forceinline__ void k_rs_scatter(const VB& vb, const K* __restrict__ kin, const u32* __restrict__ vin, size_t n, unsigned shift, unsigned mask,
                                    const u32* __restrict__ hist, const u32* __restrict__ chunk_off, u32 n_tiles, K* __restrict__ kout, u32* __restrict__ vout) {
    // The tile is sorted by digit in LDS first and written out in that order: a digit's keys of the tile are then consecutive lanes'
    // consecutive addresses. Writing every pair straight to its place (the first form of this kernel) ran a pass over random digits at
    // 0.8 TB/s — 8- and 4-byte stores scattered over 256 runs per tile — against 3.9 TB/s when all digits were equal.
    __shared__ K s_key[RS_TILE];
    __shared__ u32 s_val[RS_TILE];
    __shared__ u32 wcnt[RS_WAVES][256];  // phase A: a wave's digit counts; phase B: the next free place (in the tile) of (wave, digit)
    __shared__ u32 s_first[256];         // where digit d's run starts in the sorted tile
    __shared__ u32 s_gbase[256];         // global place of the run's first pair minus s_first[d]
    __shared__ u32 s_wsum[RS_WAVES];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
#pragma unroll
    for (int w = 0; w < RS_WAVES; w++) wcnt[w][t] = 0;
    __syncthreads();
    const size_t tile0 = (size_t)vb.x * RS_TILE, base = tile0 + (size_t)wave * (RS_ROUNDS * 64) + lane;
    K key[RS_ROUNDS];
    u32 val[RS_ROUNDS];
    unsigned long long same[RS_ROUNDS];  // the lanes of this wave whose key of the round has the same digit
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; r++) {  // every load of the tile in flight before the first ballot
        const size_t i = base + (size_t)r * 64;
        key[r] = i < n ? kin[i] : (K)0;
        val[r] = i < n ? vin[i] : 0u;
    }
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; r++) {
        const size_t i = base + (size_t)r * 64;
        const bool ok = i < n;
        const unsigned d = rs_digit(key[r], shift, mask);
        unsigned long long m = __ballot(ok);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const unsigned long long bal = __ballot((d >> b) & 1);
            m &= ((d >> b) & 1) ? bal : ~bal;
        }
        same[r] = ok ? m : 0ull;
        // the lowest lane of a digit's group adds the group to the wave's count (one writer per digit and round; rounds in order)
        if (ok && (m & ((1ull << lane) - 1)) == 0) wcnt[wave][d] += (u32)__popcll(m);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    {  // digit t: its run in the sorted tile (exclusive scan of the tile's digit counts), where each wave's keys of it go, where the run goes globally
        u32 c[RS_WAVES], total = 0;
#pragma unroll
        for (int w = 0; w < RS_WAVES; w++) { c[w] = wcnt[w][t]; total += c[w]; }
        u32 incl = total;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const u32 y = __shfl_up(incl, off, 64);
            if (lane >= off) incl += y;
        }
        if (lane == 63) s_wsum[wave] = incl;
        __syncthreads();
        u32 first = incl - total;
        for (int w = 0; w < wave; w++) first += s_wsum[w];
        s_first[t] = first;
        const size_t e = (size_t)t * n_tiles + vb.x;
        s_gbase[t] = hist[e] + chunk_off[e / RS_CHUNK] - first;
        u32 at = first;
#pragma unroll
        for (int w = 0; w < RS_WAVES; w++) { wcnt[w][t] = at; at += c[w]; }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; r++) {
        const unsigned long long m = same[r];
        if (m) {
            const unsigned d = rs_digit(key[r], shift, mask);
            const u32 at = wcnt[wave][d] + (u32)__popcll(m & ((1ull << lane) - 1));
            s_key[at] = key[r];
            s_val[at] = val[r];
        }
        __builtin_amdgcn_wave_barrier();
        if (m && (m & ((1ull << lane) - 1)) == 0) wcnt[wave][rs_digit(key[r], shift, mask)] += (u32)__popcll(m);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    const u32 in_tile = (u32)(n - tile0 < (size_t)RS_TILE ? n - tile0 : (size_t)RS_TILE);
#pragma unroll 4
    for (u32 p = t; p < in_tile; p += 256) {
        const K k = s_key[p];
        const u32 at = s_gbase[rs_digit(k, shift, mask)] + p;
        kout[at] = k;
        vout[at] = s_val[p];
    }
}

static inline size_t rs_tiles(size_t n) { return (n + RS_TILE - 1) / RS_TILE; }
static inline size_t radix_temp_bytes(size_t n) {
    const size_t entries = 256 * rs_tiles(n), chunks = (entries + RS_CHUNK - 1) / RS_CHUNK;
    return ((n * 8 + 255) & ~(size_t)255) + ((n * 4 + 255) & ~(size_t)255) + entries * 4 + chunks * 4 + 1024;
}

// kin / vin -> kout / vout sorted by the low `end_bit` bits of the key, stable. tmp: radix_temp_bytes(n) bytes. kin == kout is NOT allowed
// (vin == vout neither); kin and vin are left as they were.
template <class K>
static int radix_sort_pairs(zkw_ctx* ctx, void* tmp, size_t tmp_bytes, const K* kin, K* kout, const u32* vin, u32* vout, size_t n, unsigned end_bit) {
    if (n == 0) return ZKW_OK;
    if (n >= (1ull << 32)) return fail(ZKW_ERR_INVALID, "radix_sort_pairs: more than 2^32 - 1 pairs");
    if (tmp_bytes < radix_temp_bytes(n)) return fail(ZKW_ERR_INVALID, "radix_sort_pairs: temporary storage too small");
    if (end_bit == 0) end_bit = 1;
    const unsigned passes = (end_bit + 7) / 8;
    const u32 n_tiles = (u32)rs_tiles(n);
    const size_t entries = (size_t)256 * n_tiles;
    const u32 n_chunks = (u32)((entries + RS_CHUNK - 1) / RS_CHUNK);
    char* p = static_cast<char*>(tmp);
    K* tk = reinterpret_cast<K*>(p);
    u32* tv = reinterpret_cast<u32*>(p + ((n * 8 + 255) & ~(size_t)255));
    u32* hist = reinterpret_cast<u32*>(p + ((n * 8 + 255) & ~(size_t)255) + ((n * 4 + 255) & ~(size_t)255));
    u32* chunk = hist + entries;
    const K* src_k = kin;
    const u32* src_v = vin;
    for (unsigned pass = 0; pass < passes; pass++) {
        // the last pass must land in (kout, vout): passes alternate out, tmp, out, ... when their number is odd, tmp, out, ... when even
        const bool to_out = ((passes - pass) & 1) == 1;
        K* dst_k = to_out ? kout : tk;
        u32* dst_v = to_out ? vout : tv;
        const unsigned shift = 8 * pass, bits = std::min(8u, end_bit - shift), mask = (1u << bits) - 1;
        ZKW_LAUNCH_T(ctx, (k_rs_hist<K>), "k_rs_hist", n_tiles, 256, src_k, n, shift, mask, hist, n_tiles);
        ZKW_LAUNCH(ctx, k_rs_scan_a, n_chunks, 1024, hist, entries, chunk);
        ZKW_LAUNCH(ctx, k_rs_scan_b, 1, 1024, chunk, n_chunks);
        ZKW_LAUNCH_T(ctx, (k_rs_scatter<K>), "k_rs_scatter", n_tiles, 256, src_k, src_v, n, shift, mask, (const u32*)hist, (const u32*)chunk, n_tiles, dst_k, dst_v);
        src_k = dst_k;
        src_v = dst_v;
    }
    return ZKW_OK;
}

}  // namespace zkw
