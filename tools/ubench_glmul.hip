// ubench_glmul.hip — candidates for the Goldilocks multiplication priced by the MEASURED issue cost of each instruction class
// (tools/ubench_valu_ceiling.hip, profiles/r05/valu_ceiling.json: 32-bit VALU 2.2 cycles per wave-instruction per SIMD, v_mad_u64_u32 4.1,
// carry-writing adds 4.2, 64-bit add / compare ~4) instead of by instruction count. Every variant is checked against gl::mul on edge
// values and 2^20 random pairs (canonical results must be equal), then timed at 1 / 2 / 4 / 8 waves per SIMD on the whole chip.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I era_zkevm_test_harness_amd/csrc -I include -o tools/ubench_glmul tools/ubench_glmul.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "poseidon2.cuh"
using gl::u32;
using gl::u64;

template <int V> __device__ __forceinline__ u64 mulv(u64 a, u64 b);
template <> __device__ __forceinline__ u64 mulv<0>(u64 a, u64 b) { return gl::mul_lat(a, b); }
template <> __device__ __forceinline__ u64 mulv<1>(u64 a, u64 b) { return gl::mul_sched(a, b); }
template <> __device__ __forceinline__ u64 mulv<2>(u64 a, u64 b) { return gl::mul_cyc(a, b); }

template <int V>
__global__ __launch_bounds__(256) void k_mul(u64* out, u64 seed, int iters) {
    extern __shared__ unsigned char lds[];
    u64 m[4];
#pragma unroll
    for (int i = 0; i < 4; i++) m[i] = seed + i * 0x9E3779B97F4A7C15ULL + threadIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
#pragma unroll
            for (int i = 0; i < 4; i++) m[i] = mulv<V>(m[i], m[(i + 1) & 3]);
        }
    }
    u64 s = m[0] ^ m[1] ^ m[2] ^ m[3];
    if (s == 0x1234567ULL) out[blockIdx.x * 256 + threadIdx.x] = s + lds[threadIdx.x];
}

template <int V>
__global__ void k_check(const u64* a, const u64* b, size_t n, unsigned long long* n_bad, u64* first) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 r = gl::canon(mulv<V>(a[i], b[i])), e = gl::canon(gl::mul_lat(a[i], b[i]));
    if (r != e && atomicAdd(n_bad, 1ull) == 0) { first[0] = a[i]; first[1] = b[i]; first[2] = r; first[3] = e; }
}

__global__ void k_perm(u64* st, size_t n) {  // the lane form as the trace fills run it
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 s[12];
    for (int k = 0; k < 12; k++) s[k] = st[12 * i + k];
    p2::permute(s);
    for (int k = 0; k < 12; k++) st[12 * i + k] = gl::canon(s[k]);
}
__global__ void k_perm_q4(u64* st, size_t n) {  // the quad form (the chain kernels): lane j of a quad holds elements j, 4 + j, 8 + j
    size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    size_t i = t / 4;
    int j = (int)(t & 3);
    p2::Coop4 co;
    co.init(j);
    u64 x[3];
    const bool live = i < n;
    for (int c = 0; c < 3; c++) x[c] = live ? st[12 * i + 4 * c + j] : 0;
    co.permute(x);
    if (live) for (int c = 0; c < 3; c++) st[12 * i + 4 * c + j] = gl::canon(x[c]);
}

static u64 sm(u64& s) { u64 z = (s += 0x9E3779B97F4A7C15ULL); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL; z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL; return z ^ (z >> 31); }

template <int V> int check(const char* name) {
    std::vector<u64> edge = {0, 1, 2, 0xFFFFFFFFull, 0x100000000ull, 0x100000001ull, gl::P - 1, gl::P, gl::P + 1, 0xFFFFFFFFFFFFFFFFull, 0xFFFFFFFF00000000ull,
                             0xFFFFFFFEFFFFFFFFull, 0x8000000000000000ull, 0x7FFFFFFFFFFFFFFFull, 0xFFFFFFFF, 0xFFFFFFFFFFFFFFFEull, 0x00000001FFFFFFFFull};
    std::vector<u64> a, b;
    for (u64 x : edge) for (u64 y : edge) { a.push_back(x); b.push_back(y); }
    u64 s = 7;
    for (int i = 0; i < (1 << 20); i++) {
        u64 x = sm(s), y = sm(s);
        if (i % 7 == 0) x |= 0xFFFFFFFF00000000ull;  // high words of all ones: the carries the reduction has to survive
        if (i % 11 == 0) y |= 0xFFFFFFFF00000000ull;
        if (i % 13 == 0) x &= 0xFFFFFFFFull;
        a.push_back(x); b.push_back(y);
    }
    size_t n = a.size();
    u64 *da, *db, *df; unsigned long long* dn;
    hipMalloc(&da, n * 8); hipMalloc(&db, n * 8); hipMalloc(&df, 32); hipMalloc(&dn, 8);
    hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice); hipMemset(dn, 0, 8);
    hipLaunchKernelGGL((k_check<V>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, da, db, n, dn, df);
    unsigned long long bad = 0; u64 f[4] = {0, 0, 0, 0};
    hipMemcpy(&bad, dn, 8, hipMemcpyDeviceToHost); hipMemcpy(f, df, 32, hipMemcpyDeviceToHost);
    fprintf(stderr, "check %-10s: %llu of %zu differ", name, bad, n);
    if (bad) fprintf(stderr, " (first: %llx * %llx = %llx, expected %llx)", (unsigned long long)f[0], (unsigned long long)f[1], (unsigned long long)f[2], (unsigned long long)f[3]);
    fprintf(stderr, "\n");
    hipFree(da); hipFree(db); hipFree(df); hipFree(dn);
    return bad != 0;
}

static int check_perm() {
    // device forms against the HOST form of the same header (the Wide lo64 + hi32 linear layers, __int128 products)
    const size_t n = 1 << 14;
    std::vector<u64> st(12 * n), ref;
    u64 s = 99;
    const u64 edge[6] = {0, 1, gl::P - 1, gl::P, 0xFFFFFFFFFFFFFFFFull, 0xFFFFFFFF00000000ull};
    for (size_t i = 0; i < n; i++)
        for (int k = 0; k < 12; k++) {
            u64 v = sm(s);
            if (i < 64) v = edge[(i + k * (i / 6 + 1)) % 6];           // states of edge values only
            else if (i % 5 == 0 && k % 3 == 0) v |= 0xFFFFFFFF00000000ull;  // (weak inputs: the layers must take any u64)
            st[12 * i + k] = v;
        }
    ref = st;
    for (size_t i = 0; i < n; i++) { p2::permute(&ref[12 * i]); for (int k = 0; k < 12; k++) ref[12 * i + k] = gl::canon(ref[12 * i + k]); }
    int bad_total = 0;
    for (int form = 0; form < 2; form++) {
        u64* d; hipMalloc(&d, st.size() * 8); hipMemcpy(d, st.data(), st.size() * 8, hipMemcpyHostToDevice);
        if (form == 0) hipLaunchKernelGGL(k_perm, dim3((unsigned)(n / 64)), dim3(64), 0, 0, d, n);
        else hipLaunchKernelGGL(k_perm_q4, dim3((unsigned)(4 * n / 64)), dim3(64), 0, 0, d, n);
        std::vector<u64> got(st.size()); hipMemcpy(got.data(), d, st.size() * 8, hipMemcpyDeviceToHost); hipFree(d);
        size_t bad = 0;
        for (size_t i = 0; i < st.size(); i++) bad += got[i] != ref[i];
        fprintf(stderr, "check permutation (%s form, %zu states): %zu words differ from the host form\n", form ? "quad" : "lane", n, bad);
        bad_total |= bad != 0;
    }
    return bad_total;
}

template <int V> void timeit(const char* name, u64* out, int n_cu, bool last) {
    printf(" {\"variant\": \"%s\", \"mul_per_s_by_waves_per_simd\": {", name);
    bool first = true;
    for (int w : {1, 2, 4, 8}) {
        const size_t lds = (160 * 1024) / w / 256 * 256;
        hipFuncSetAttribute((const void*)k_mul<V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        int occ = 0;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)k_mul<V>, 256, lds);
        if (occ < w) continue;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float best = 1e30f;
        const int iters = 2000;
        for (int rep = 0; rep < 4; rep++) {
            hipEventRecord(e0);
            hipLaunchKernelGGL((k_mul<V>), dim3(n_cu * w), dim3(256), lds, 0, out, 12345ull, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep && ms < best) best = ms;
        }
        const double muls = (double)n_cu * w * 256 * 64.0 * iters;  // lane multiplications
        const double per_simd_wave_muls = muls / 64 / (n_cu * 4.0) / (best * 1e-3);
        printf("%s\"%d\": {\"ms\": %.3f, \"lane_mul_per_s\": %.4e, \"cycles_per_wave_mul_per_simd_at_2.4GHz\": %.2f}", first ? "" : ", ", w, best, muls / (best * 1e-3), 2.4e9 / per_simd_wave_muls);
        first = false;
    }
    printf("}}%s\n", last ? "" : ",");
}

int main() {
    int n_cu = 0;
    hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, 0);
    int bad = check<1>("mul_sched") | check<2>("mul_cyc") | check_perm();
    u64* out; hipMalloc(&out, (size_t)n_cu * 8 * 256 * 8);
    printf("{\"checks_failed\": %d, \"variants\": [\n", bad);
    timeit<0>("gl::mul (compiler form, 21 wave-instructions)", out, n_cu, false);
    timeit<1>("gl::mul_sched (hand-scheduled for one wave, 20 instructions)", out, n_cu, false);
    timeit<2>("gl::mul_cyc (priced by class cycles)", out, n_cu, true);
    printf("]}\n");
    return bad;
}
