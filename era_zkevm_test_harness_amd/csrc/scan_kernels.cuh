// scan_kernels.cuh — tiled prefix counts / sums. flag_prefix: count of a per-item flag: prefix[k] = #{ i < k : flag(i) }, k = 0..n. Three launches (per-tile
// inclusive counts with wave ballots, an exclusive scan of the tile totals, the tile offsets added), any n; replaces the
// single-workgroup loops the synthesis of the sorter circuits used for their compaction indices.
#pragma once
#include "zkw_ctx.h"

namespace zkw {

constexpr int FLAG_PREFIX_TILE = 1024;

template <class Flag>
static __device__ __forceinline__ void k_flag_prefix_tiles(const VB& vb, Flag flag, size_t n, u32* __restrict__ prefix, u32* __restrict__ tile_sums) {
    __shared__ u32 s_wave[FLAG_PREFIX_TILE / 64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const size_t i = (size_t)vb.x * FLAG_PREFIX_TILE + t;
    const bool f = i < n && flag(i) != 0;
    const unsigned long long bal = __ballot(f);
    const u32 incl = __popcll(bal & ((2ull << lane) - 1));
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    u32 before = 0;
    for (int w = 0; w < wave; w++) before += s_wave[w];
    if (i < n) prefix[i + 1] = before + incl;
    if (t == FLAG_PREFIX_TILE - 1) tile_sums[vb.x] = before + incl;
    if (i == 0) prefix[0] = 0;
}

// exclusive scan of the tile totals in place: one workgroup, n_tiles = n / 1024 elements
static __device__ __forceinline__ void k_flag_prefix_offsets(const VB& vb, u32* __restrict__ tile_sums, u32 n_tiles) {
    __shared__ u32 s[1024];
    __shared__ u32 carry;
    const int t = threadIdx.x;
    if (t == 0) carry = 0;
    __syncthreads();
    for (u32 base = 0; base < n_tiles; base += 1024) {
        const u32 i = base + t;
        const u32 v = i < n_tiles ? tile_sums[i] : 0;
        s[t] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const u32 x = t >= off ? s[t - off] : 0;
            __syncthreads();
            s[t] += x;
            __syncthreads();
        }
        if (i < n_tiles) tile_sums[i] = carry + s[t] - v;
        __syncthreads();
        if (t == 0) carry += s[1023];
        __syncthreads();
    }
}

static __device__ __forceinline__ void k_flag_prefix_apply(const VB& vb, u32* __restrict__ prefix, const u32* __restrict__ tile_offsets, size_t n) {
    const size_t i = (size_t)vb.x * blockDim.x + threadIdx.x;
    if (i < n) prefix[i + 1] += tile_offsets[i / FLAG_PREFIX_TILE];
}

// d_prefix: [n + 1] on the device; stream-ordered on the context's stream
template <class Flag>
static int flag_prefix(zkw_ctx* ctx, const char* name, Flag flag, size_t n, u32* d_prefix) {
    if (n == 0) return ctx->memset_async(d_prefix, 0, sizeof(u32)) == hipSuccess ? ZKW_OK : fail(ZKW_ERR_HIP, "memset failed");
    const unsigned n_tiles = (unsigned)((n + FLAG_PREFIX_TILE - 1) / FLAG_PREFIX_TILE);
    u32* d_tiles = nullptr;
    ZKW_TRY(ctx->scratch_t<u32>("flag_prefix_tiles", n_tiles, &d_tiles));
    { Prof _p(ctx, name); ZKW_LAUNCH_T(ctx, (k_flag_prefix_tiles<Flag>), "k_flag_prefix_tiles", n_tiles, FLAG_PREFIX_TILE, flag, n, d_prefix, d_tiles); }
    ZKW_TRY(launch_check(name));
    if (n_tiles > 1) {
        ZKW_LAUNCH(ctx, k_flag_prefix_offsets, 1, 1024, d_tiles, n_tiles);
        ZKW_TRY(launch_check("k_flag_prefix_offsets"));
        ZKW_LAUNCH(ctx, k_flag_prefix_apply, (unsigned)((n + 255) / 256), 256, d_prefix, d_tiles, n);
        ZKW_TRY(launch_check("k_flag_prefix_apply"));
    }
    return ZKW_OK;
}

// ---- K routes at once: route(i) in [-1, K); count[c][i] = #{ j <= i : route(j) == c } (inclusive), totals[c] = count[c][n - 1].
// The same three launches with K counters side by side (the log demuxer's six stable compactions).
template <int K, class Route>
static __device__ __forceinline__ void k_route_prefix_tiles(const VB& vb, Route route, size_t n, u32* __restrict__ count /* [K][n] */, u32* __restrict__ tile_sums /* [K][n_tiles] */) {
    __shared__ u32 s_wave[K][FLAG_PREFIX_TILE / 64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const size_t i = (size_t)vb.x * FLAG_PREFIX_TILE + t;
    const int r = i < n ? route(i) : -1;
    u32 incl[K];
#pragma unroll
    for (int c = 0; c < K; c++) {
        const unsigned long long bal = __ballot(r == c);
        incl[c] = __popcll(bal & ((2ull << lane) - 1));
        if (lane == 63) s_wave[c][wave] = incl[c];
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < K; c++) {
        u32 before = 0;
        for (int w = 0; w < wave; w++) before += s_wave[c][w];
        if (i < n) count[(size_t)c * n + i] = before + incl[c];
        if (t == FLAG_PREFIX_TILE - 1) tile_sums[(size_t)c * vb.nx + vb.x] = before + incl[c];
    }
}
template <int K>
static __device__ __forceinline__ void k_route_prefix_apply(const VB& vb, u32* __restrict__ count, const u32* __restrict__ tile_offsets, size_t n, u32 n_tiles) {
    const size_t i = (size_t)vb.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
#pragma unroll
    for (int c = 0; c < K; c++) count[(size_t)c * n + i] += tile_offsets[(size_t)c * n_tiles + i / FLAG_PREFIX_TILE];
}
template <int K, class Route>
static int route_prefix(zkw_ctx* ctx, const char* name, Route route, size_t n, u32* d_count /* [K][n] */) {
    if (n == 0) return ZKW_OK;
    const unsigned n_tiles = (unsigned)((n + FLAG_PREFIX_TILE - 1) / FLAG_PREFIX_TILE);
    u32* d_tiles = nullptr;
    ZKW_TRY(ctx->scratch_t<u32>("route_prefix_tiles", (size_t)K * n_tiles, &d_tiles));
    { Prof _p(ctx, name); ZKW_LAUNCH_T(ctx, (k_route_prefix_tiles<K, Route>), "k_route_prefix_tiles", n_tiles, FLAG_PREFIX_TILE, route, n, d_count, d_tiles); }
    ZKW_TRY(launch_check(name));
    if (n_tiles > 1) {
        for (int c = 0; c < K; c++) ZKW_LAUNCH(ctx, k_flag_prefix_offsets, 1, 1024, d_tiles + (size_t)c * n_tiles, n_tiles);
        ZKW_TRY(launch_check("k_flag_prefix_offsets"));
        ZKW_LAUNCH_T(ctx, (k_route_prefix_apply<K>), "k_route_prefix_apply", (unsigned)((n + 255) / 256), 256, d_count, d_tiles, n, n_tiles);
        ZKW_TRY(launch_check("k_route_prefix_apply"));
    }
    return ZKW_OK;
}

// ---- K SUMS side by side (64-bit, wrap-around: signed deltas work): out[c][i] = sum over j < i of val(j)[c] (exclusive), out[c][n] =
// totals[c] = the sum over all items. val(i, v) fills v[0..K). The same three launches; replaces the single-workgroup sweeps of
// k_precompile_counts (rounds / queries / reads per request) and k_stack_depth (depth, push rank).
template <int K, class Val>
static __device__ __forceinline__ void k_sum_prefix_tiles(const VB& vb, Val val, size_t n, u64* __restrict__ out /* [K][n + 1] */, u64* __restrict__ tile_sums /* [K][n_tiles] */) {
    __shared__ u64 s_wave[K][FLAG_PREFIX_TILE / 64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const size_t i = (size_t)vb.x * FLAG_PREFIX_TILE + t;
    u64 v[K], incl[K];
#pragma unroll
    for (int c = 0; c < K; c++) v[c] = 0;
    if (i < n) val(i, v);
#pragma unroll
    for (int c = 0; c < K; c++) {
        u64 x = v[c];
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const u64 y = (u64)__shfl_up((unsigned long long)x, off, 64);
            if (lane >= off) x += y;
        }
        incl[c] = x;
        if (lane == 63) s_wave[c][wave] = x;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < K; c++) {
        u64 before = 0;
        for (int w = 0; w < wave; w++) before += s_wave[c][w];
        if (i < n) out[(size_t)c * (n + 1) + i] = before + incl[c] - v[c];
        if (t == FLAG_PREFIX_TILE - 1) tile_sums[(size_t)c * vb.nx + vb.x] = before + incl[c];
    }
}
// exclusive scan of one sum's tile totals in place (one workgroup), its grand total to *total and to out_last (= out[c][n])
static __device__ __forceinline__ void k_sum_prefix_offsets(const VB& vb, u64* __restrict__ tile_sums, u32 n_tiles, u64* __restrict__ total, u64* __restrict__ out_last) {
    __shared__ u64 s[1024];
    __shared__ u64 carry;
    const int t = threadIdx.x;
    if (t == 0) carry = 0;
    __syncthreads();
    for (u32 base = 0; base < n_tiles; base += 1024) {
        const u32 i = base + t;
        const u64 v = i < n_tiles ? tile_sums[i] : 0;
        s[t] = v;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const u64 x = t >= off ? s[t - off] : 0;
            __syncthreads();
            s[t] += x;
            __syncthreads();
        }
        if (i < n_tiles) tile_sums[i] = carry + s[t] - v;
        __syncthreads();
        if (t == 0) carry += s[1023];
        __syncthreads();
    }
    if (t == 0) { *total = carry; *out_last = carry; }
}
template <int K>
static __device__ __forceinline__ void k_sum_prefix_apply(const VB& vb, u64* __restrict__ out, const u64* __restrict__ tile_offsets, size_t n, u32 n_tiles) {
    const size_t i = (size_t)vb.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
#pragma unroll
    for (int c = 0; c < K; c++) out[(size_t)c * (n + 1) + i] += tile_offsets[(size_t)c * n_tiles + i / FLAG_PREFIX_TILE];
}
// d_out: [K][n + 1], d_totals: [K] on the device; stream-ordered on the context's stream
template <int K, class Val>
static int sum_prefix(zkw_ctx* ctx, const char* name, Val val, size_t n, u64* d_out, u64* d_totals) {
    const unsigned n_tiles = (unsigned)((n + FLAG_PREFIX_TILE - 1) / FLAG_PREFIX_TILE);
    u64* d_tiles = nullptr;
    ZKW_TRY(ctx->scratch_t<u64>("sum_prefix_tiles", (size_t)K * (n_tiles ? n_tiles : 1), &d_tiles));
    if (n) {
        { Prof _p(ctx, name); ZKW_LAUNCH_T(ctx, (k_sum_prefix_tiles<K, Val>), "k_sum_prefix_tiles", n_tiles, FLAG_PREFIX_TILE, val, n, d_out, d_tiles); }
        ZKW_TRY(launch_check(name));
    }
    for (int c = 0; c < K; c++) ZKW_LAUNCH(ctx, k_sum_prefix_offsets, 1, 1024, d_tiles + (size_t)c * n_tiles, n_tiles, d_totals + c, d_out + (size_t)c * (n + 1) + n);
    ZKW_TRY(launch_check("k_sum_prefix_offsets"));
    if (n_tiles > 1) {
        ZKW_LAUNCH_T(ctx, (k_sum_prefix_apply<K>), "k_sum_prefix_apply", (unsigned)((n + 255) / 256), 256, d_out, d_tiles, n, n_tiles);
        ZKW_TRY(launch_check("k_sum_prefix_apply"));
    }
    return ZKW_OK;
}

}  // namespace zkw
