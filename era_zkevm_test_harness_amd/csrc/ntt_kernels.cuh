// ntt_kernels.cuh — the setup side as field elements on gfx950: number-theoretic transform over Goldilocks, low-degree extension onto
// cosets, Poseidon2 Merkle tree with a cap. What the reference does between a synthesized circuit and its verification key
// (src/prover_utils.rs:48-197 create_base_layer_setup_data: SetupBaseStorage -> SetupStorage (monomial form, LDE x fri_lde_factor) ->
// MerkleTreeWithCap over the LDE, cap 16 -> VerificationKey.setup_merkle_tree_cap); the bodies are boojum's (absent crate). Pinned by
// the reference's data: leaf hash = overwrite-mode sponge over a row's values, node hash = permutation of (left || right || 0000), one
// binary tree over all LDE positions cut at the cap (every Merkle path of the committed proofs, tests/golden/reference_merkle_paths_kat.json).
// NOT pinned: which evaluation point a leaf index stands for (boojum's coset / bit-reversal enumeration) — DESIGN.md 3.21.
//
// NTT. n = 2^log_n points per column, columns back to back (column-major, like the traces). One pass in LDS for log_n <= 13; else the
// four-step split n = n1 * n2 (n1 = 2^a, n2 = 2^b, a, b <= 10): pass 1 transforms over j1 (stride n2: a workgroup takes a tile of
// 8192 / n1 neighbouring j2, i.e. 64-byte segments) and multiplies by w^(j2 k1); pass 2 transforms the contiguous rows over j2 and stores
// X[k1 + n1 k2] (64-byte segments again). Both passes: 8192 points = 64 KB of LDS per workgroup of 1024 lanes (two workgroups per CU = 8 waves
// per SIMD; 256 lanes measured 1.24 / 1.44 TB/s, 512 1.60 / 1.85, 1024 1.62 / 2.07), decimation in frequency two stages per barrier
// (natural order in, bit-reversed positions out: the store un-reverses), twiddles of the sub-transform in LDS.
// What bounds it: VALU issue, not HBM. A pass moves 16 B per point and spends 5 multiplications + 10 additions of a 64-bit prime field on
// it with 32-bit ALUs (~20 and ~6 instructions each, + addressing, + the inter-pass twiddle in pass 1): ~230 - 280 wave-instructions per
// 64 points, a floor of ~0.8 - 1.0 ms per pass of 131 x 2^20 points at 6.1e11 wave-instructions/s against 0.28 ms at 8 TB/s
// (DESIGN.md 3.21 has the measured rates).
#pragma once
#include "poseidon2.cuh"

namespace zkw {

constexpr int NTT_THREADS = 1024;
constexpr int NTT_TILE = 8192;      // points per workgroup (2^13)
constexpr int NTT_MAX_SINGLE = 13;  // log_n of the one-pass kernel
constexpr int NTT_TW = 1024;        // w^e = tw_hi[e >> 10] * tw_lo[e & 1023] for e < 2^20

struct NttArgs {
    const u64* in;
    u64* out;
    u32 log_n, a, b;        // n = 2^log_n; two-pass split a + b = log_n (one pass: a = log_n, b = 0)
    const u64 *tw_a, *tw_b;  // w_(2^a)^i for i < 2^(a-1); w_(2^b)^i
    const u64 *tw_lo, *tw_hi;    // w_n^i (i < 1024), w_n^(1024 i) (i < 1024): the inter-pass twiddles
    const u64 *pre_lo, *pre_hi;  // s^i, s^(1024 i): input point j is multiplied by s^j (coset shift); null = no scaling
    u64 post;                    // every output is multiplied by this (1 / n of the inverse transform; 1 = nothing)
};

__device__ __forceinline__ u32 bitrev(u32 x, u32 bits) { return bits ? __brev(x) >> (32 - bits) : 0; }

// `cnt` sequences of length L = 2^m at lds[seq * (L + 1) + i] (the odd stride keeps neighbouring sequences off one bank), tw = w_L^i in LDS.
// Natural order in, position p holds X[bitrev(p)] out. Decimation in frequency, TWO stages per barrier: a lane takes the four points
// {pos, pos + h, pos + 2h, pos + 3h} of a group of 4h through the stages of half-size 2h and h in registers (the second twiddle of the first
// stage is tw[i + L / 4] = w^i * w_4) — half the LDS traffic and half the barriers of the radix-2 walk; a last radix-2 stage when m is odd.
__device__ __forceinline__ void ntt_dif_lds(u64* lds, const u64* tw, u32 m, u32 cnt) {
    const u32 L = 1u << m;
    int s = (int)m - 1;
    for (; s >= 1; s -= 2) {
        const u32 h = 1u << (s - 1), total = cnt << (m - 2);
        for (u32 q = threadIdx.x; q < total; q += NTT_THREADS) {
            const u32 seq = q >> (m - 2), r = q & (L / 4 - 1), pos = r & (h - 1);
            const u32 i0 = seq * (L + 1) + ((r >> (s - 1)) << (s + 1)) + pos;
            const u64 a = lds[i0], b = lds[i0 + h], c = lds[i0 + 2 * h], d = lds[i0 + 3 * h];
            const u32 i1 = pos << (m - 1 - s), i2 = pos << (m - s);
            const u64 a1 = gl::add(a, c), c1 = gl::mul(gl::sub(a, c), tw[i1]);
            const u64 b1 = gl::add(b, d), d1 = gl::mul(gl::sub(b, d), tw[i1 + L / 4]);
            const u64 w2 = tw[i2];
            lds[i0] = gl::add(a1, b1);
            lds[i0 + h] = gl::mul(gl::sub(a1, b1), w2);
            lds[i0 + 2 * h] = gl::add(c1, d1);
            lds[i0 + 3 * h] = gl::mul(gl::sub(c1, d1), w2);
        }
        __syncthreads();
    }
    if (s == 0) {  // half-size 1: no twiddle
        const u32 total = cnt << (m - 1);
        for (u32 bf = threadIdx.x; bf < total; bf += NTT_THREADS) {
            const u32 seq = bf >> (m - 1), r = bf & (L / 2 - 1), i0 = seq * (L + 1) + 2 * r;
            const u64 u = lds[i0], v = lds[i0 + 1];
            lds[i0] = gl::add(u, v);
            lds[i0 + 1] = gl::sub(u, v);
        }
        __syncthreads();
    }
}

__device__ __forceinline__ u64 tw_pow(const u64* lo, const u64* hi, u32 e) { return gl::mul(hi[e >> 10], lo[e & (NTT_TW - 1)]); }

// one pass: a whole sequence per workgroup. grid (1, n_cols)
static __global__ __launch_bounds__(NTT_THREADS) void k_ntt_single(NttArgs A) {
    extern __shared__ __attribute__((aligned(16))) u64 ntt_lds[];
    const u32 m = A.log_n, n = 1u << m;
    u64* tw = ntt_lds + n + 1;
    const u64* in = A.in + (size_t)blockIdx.y * n;
    u64* out = A.out + (size_t)blockIdx.y * n;
    for (u32 i = threadIdx.x; i < n / 2; i += NTT_THREADS) tw[i] = A.tw_a[i];
    for (u32 i = threadIdx.x; i < n; i += NTT_THREADS) {
        u64 x = in[i];
        if (A.pre_lo) x = gl::mul(x, tw_pow(A.pre_lo, A.pre_hi, i));
        ntt_lds[i] = x;
    }
    __syncthreads();
    if (m) ntt_dif_lds(ntt_lds, tw, m, 1);
    for (u32 p = threadIdx.x; p < n; p += NTT_THREADS) {
        u64 x = ntt_lds[p];
        if (A.post != 1) x = gl::mul(x, A.post);
        out[bitrev(p, m)] = gl::canon(x);
    }
}

// pass 1 of the split: grid (n2 / T, n_cols), T = NTT_TILE / n1 neighbouring j2 per workgroup
static __global__ __launch_bounds__(NTT_THREADS) void k_ntt_pass1(NttArgs A) {
    extern __shared__ __attribute__((aligned(16))) u64 ntt_lds[];
    const u32 n1 = 1u << A.a, n2 = 1u << A.b, lt = 13 - A.a, T = 1u << lt, j2_0 = blockIdx.x * T;  // NTT_TILE = 2^13
    u64* tw = ntt_lds + T * (n1 + 1);
    const u64* in = A.in + ((size_t)blockIdx.y << A.log_n);
    u64* out = A.out + ((size_t)blockIdx.y << A.log_n);
    for (u32 i = threadIdx.x; i < n1 / 2; i += NTT_THREADS) tw[i] = A.tw_a[i];
    for (u32 idx = threadIdx.x; idx < NTT_TILE; idx += NTT_THREADS) {
        const u32 t = idx & (T - 1), j1 = idx >> lt, j = j1 * n2 + j2_0 + t;
        u64 x = in[j];
        if (A.pre_lo) x = gl::mul(x, tw_pow(A.pre_lo, A.pre_hi, j));
        ntt_lds[t * (n1 + 1) + j1] = x;
    }
    __syncthreads();
    ntt_dif_lds(ntt_lds, tw, A.a, T);
    for (u32 idx = threadIdx.x; idx < NTT_TILE; idx += NTT_THREADS) {
        const u32 t = idx & (T - 1), p = idx >> lt, k1 = bitrev(p, A.a), j2 = j2_0 + t;
        out[(size_t)k1 * n2 + j2] = gl::mul(ntt_lds[t * (n1 + 1) + p], tw_pow(A.tw_lo, A.tw_hi, j2 * k1));  // j2 k1 < n <= 2^20
    }
}

// pass 2: grid (n1 / T, n_cols), T = NTT_TILE / n2 rows k1 per workgroup
static __global__ __launch_bounds__(NTT_THREADS) void k_ntt_pass2(NttArgs A) {
    extern __shared__ __attribute__((aligned(16))) u64 ntt_lds[];
    const u32 n1 = 1u << A.a, n2 = 1u << A.b, lt = 13 - A.b, T = 1u << lt, k1_0 = blockIdx.x * T;
    u64* tw = ntt_lds + T * (n2 + 1);
    const u64* in = A.in + ((size_t)blockIdx.y << A.log_n);
    u64* out = A.out + ((size_t)blockIdx.y << A.log_n);
    for (u32 i = threadIdx.x; i < n2 / 2; i += NTT_THREADS) tw[i] = A.tw_b[i];
    for (u32 idx = threadIdx.x; idx < NTT_TILE; idx += NTT_THREADS) {
        const u32 r = idx >> A.b, j2 = idx & (n2 - 1);
        ntt_lds[r * (n2 + 1) + j2] = in[(size_t)(k1_0 + r) * n2 + j2];
    }
    __syncthreads();
    ntt_dif_lds(ntt_lds, tw, A.b, T);
    for (u32 idx = threadIdx.x; idx < NTT_TILE; idx += NTT_THREADS) {
        const u32 r = idx & (T - 1), p = idx >> lt, k2 = bitrev(p, A.b);
        u64 x = ntt_lds[r * (n2 + 1) + p];
        if (A.post != 1) x = gl::mul(x, A.post);
        out[(size_t)k2 * n1 + k1_0 + r] = gl::canon(x);
    }
}

// ---------------------------------------------------------------- Merkle tree with a cap
// leaf i = sponge over (col 0 .. n_cols-1 at position i), eight at a time, overwrite mode, zero-padded last chunk: one lane per leaf,
// reads coalesced across the lanes of a wave (column-major input, `stride` elements between columns)
static __global__ __launch_bounds__(256) void k_merkle_leaves(const u64* __restrict__ cols, size_t n_cols, size_t stride, size_t n, u64* __restrict__ digests) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 s[12], nx[8];
#pragma unroll
    for (int k = 0; k < 12; k++) s[k] = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) nx[k] = (size_t)k < n_cols ? cols[(size_t)k * stride + i] : 0;
    for (size_t c0 = 0; c0 < n_cols; c0 += 8) {
#pragma unroll
        for (int k = 0; k < 8; k++) s[k] = nx[k];
#pragma unroll
        for (int k = 0; k < 8; k++) nx[k] = c0 + 8 + k < n_cols ? cols[(c0 + 8 + k) * stride + i] : 0;  // the next chunk's loads fly during the permutation
        p2::permute(s);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) digests[4 * i + k] = gl::canon(s[k]);
}

static __global__ __launch_bounds__(256) void k_merkle_nodes(const u64* __restrict__ below, size_t n_nodes, u64* __restrict__ level) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    u64 s[12];
#pragma unroll
    for (int k = 0; k < 8; k++) s[k] = below[8 * i + k];
#pragma unroll
    for (int k = 8; k < 12; k++) s[k] = 0;
    p2::permute(s);
#pragma unroll
    for (int k = 0; k < 4; k++) level[4 * i + k] = gl::canon(s[k]);
}

// sigma as field elements: cell index (col' * n + row') -> k_col' * w^row' with the coset representatives k_j = g^j (g = 7) and w^r from a table
static __global__ __launch_bounds__(256) void k_sigma_to_field(const u64* __restrict__ sigma_idx, size_t n_cells, u32 log_n, const u64* __restrict__ omega_pow,
                                                               const u64* __restrict__ coset_rep, u64* __restrict__ out) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_cells; i += stride) {
        const u64 t = sigma_idx[i];
        out[i] = gl::canon(gl::mul(coset_rep[t >> log_n], omega_pow[t & ((1ull << log_n) - 1)]));
    }
}
// w^i for i < n from the two power tables
static __global__ __launch_bounds__(256) void k_powers(const u64* __restrict__ lo, const u64* __restrict__ hi, size_t n, u64* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = gl::canon(tw_pow(lo, hi, (u32)i));
}
static __global__ __launch_bounds__(256) void k_bytes_to_field(const uint8_t* __restrict__ in, size_t n, u64* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i];
}

}  // namespace zkw
