// p2_search2.hip — research tool (not part of the product or the oracle): widens tools/p2_search.hip from "all orderings
// of one set of twelve internal-diagonal shifts" to "all orderings of EVERY 12-subset of {0..MAXS-1}", against an exact
// known answer from the reference's committed proof (tools/p2_pair_kat.txt, harvested from
// test_proofs/base_layer/basic_circuit_proof_8_0.json: two sibling Merkle nodes of the quotient oracle whose parent must be
// one of the 31 listed level-16 nodes). Usage: p2_search2 tools/p2_pair_kat.txt [MAXS=16] [form=0]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../era_zkevm_test_harness_amd/csrc/gl64.cuh"
#include "../era_zkevm_test_harness_amd/csrc/poseidon2_constants.h"
using gl::u64; using gl::u32;
__constant__ u64 c_rc[360];
__constant__ u64 c_pre[2 * 12];
__constant__ u64 c_top0[32];
__constant__ int c_ntop;
struct Vals { u32 v[12]; };
__host__ __device__ inline void m4(u64& x0, u64& x1, u64& x2, u64& x3) {
    u64 t0 = gl::add(x0, x1), t1 = gl::add(x2, x3);
    u64 t2 = gl::add(gl::add(x1, x1), t1), t3 = gl::add(gl::add(x3, x3), t0);
    u64 t14 = gl::add(t1, t1); t14 = gl::add(t14, t14);
    u64 t04 = gl::add(t0, t0); t04 = gl::add(t04, t04);
    u64 t4 = gl::add(t14, t3), t5 = gl::add(t04, t2);
    x0 = gl::add(t3, t5); x1 = t5; x2 = gl::add(t2, t4); x3 = t4;
}
__host__ __device__ inline void ext(u64* s) {
    m4(s[0], s[1], s[2], s[3]); m4(s[4], s[5], s[6], s[7]); m4(s[8], s[9], s[10], s[11]);
    for (int i = 0; i < 4; i++) {
        u64 col = gl::add(gl::add(s[i], s[4 + i]), s[8 + i]);
        s[i] = gl::add(s[i], col); s[4 + i] = gl::add(s[4 + i], col); s[8 + i] = gl::add(s[8 + i], col);
    }
}
__global__ void k_search(Vals vals, u64 first, u64 count, int form, u64* hits, unsigned* n_hits, u64 tag) {
    u64 idx = first + (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= first + count) return;
    u32 sh[12];
    {
        u32 avail[12];
        for (int i = 0; i < 12; i++) avail[i] = vals.v[i];
        u64 r = idx, fact = 39916800ULL;
        for (int i = 0; i < 12; i++) {
            u32 d = (u32)(r / fact); r %= fact; sh[i] = avail[d];
            for (int k = d; k < 11 - i; k++) avail[k] = avail[k + 1];
            if (i < 11) fact /= (11 - i);
        }
    }
    for (int in = 0; in < 2; in++) {
        u64 s[12];
        for (int i = 0; i < 12; i++) s[i] = c_pre[in * 12 + i];
        int r = 4;
        for (int k = 0; k < 22; k++, r++) {
            s[0] = gl::pow7(gl::add(s[0], c_rc[12 * r]));
            u64 sum = s[0];
            for (int i = 1; i < 12; i++) sum = gl::add(sum, s[i]);
            for (int i = 0; i < 12; i++) {
                u64 y = gl::add(gl::mul_pow2(s[i], sh[i]), sum);
                if (form == 1) y = gl::sub(y, s[i]);
                if (form == 2) y = gl::add(y, s[i]);
                s[i] = y;
            }
        }
        for (int k = 0; k < 4; k++, r++) {
            for (int i = 0; i < 12; i++) s[i] = gl::pow7(gl::add(s[i], c_rc[12 * r + i]));
            ext(s);
        }
        const u64 o = gl::canon(s[0]);
        bool hit = false;
        for (int c = 0; c < c_ntop; c++) hit |= (o == c_top0[c]);
        if (hit) { unsigned k = atomicAdd(n_hits, 1u); if (k < 32) { hits[3 * k] = tag; hits[3 * k + 1] = idx; hits[3 * k + 2] = in; } }
    }
}
int main(int argc, char** argv) {
    if (argc < 2) return 2;
    const int MAXS = argc > 2 ? atoi(argv[2]) : 16, form = argc > 3 ? atoi(argv[3]) : 0;
    FILE* f = fopen(argv[1], "r"); if (!f) return 2;
    u64 a[4], b[4]; for (int i = 0; i < 4; i++) if (fscanf(f, "%lu", &a[i]) != 1) return 2;
    for (int i = 0; i < 4; i++) if (fscanf(f, "%lu", &b[i]) != 1) return 2;
    int nt; if (fscanf(f, "%d", &nt) != 1) return 2;
    u64 top0[32]; for (int t = 0; t < nt; t++) { u64 w[4]; for (int i = 0; i < 4; i++) if (fscanf(f, "%lu", &w[i]) != 1) return 2; top0[t] = w[0] % gl::P; }
    fclose(f);
    hipMemcpyToSymbol(HIP_SYMBOL(c_rc), P2_ROUND_CONSTANTS, sizeof(u64) * 360);
    hipMemcpyToSymbol(HIP_SYMBOL(c_top0), top0, sizeof(u64) * nt); hipMemcpyToSymbol(HIP_SYMBOL(c_ntop), &nt, sizeof nt);
    u64 pre[24];
    for (int order = 0; order < 2; order++) {
        u64 s[12] = {0}; memcpy(s, order ? b : a, 32); memcpy(s + 4, order ? a : b, 32);
        ext(s);
        for (int r = 0; r < 4; r++) { for (int i = 0; i < 12; i++) s[i] = gl::pow7(gl::add(s[i], P2_ROUND_CONSTANTS[12 * r + i])); ext(s); }
        memcpy(pre + 12 * order, s, 96);
    }
    hipMemcpyToSymbol(HIP_SYMBOL(c_pre), pre, sizeof pre);
    u64* d_hits; unsigned* d_nh; hipMalloc(&d_hits, 96 * 8); hipMalloc(&d_nh, 4); hipMemset(d_nh, 0, 4);
    const u64 TOTAL = 479001600ULL;
    unsigned long long subsets = 0;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0);
    for (u32 mask = 0; mask < (1u << MAXS); mask++) {
        if (__builtin_popcount(mask) != 12) continue;
        Vals v; int k = 0; for (int s = 0; s < MAXS; s++) if (mask >> s & 1) v.v[k++] = s;
        const u64 CH = 1ULL << 27;
        for (u64 first = 0; first < TOTAL; first += CH) {
            u64 cnt = TOTAL - first < CH ? TOTAL - first : CH;
            hipLaunchKernelGGL(k_search, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, 0, v, first, cnt, form, d_hits, d_nh, (u64)mask);
        }
        subsets++;
        if (subsets % 50 == 0) {
            hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned nh; hipMemcpy(&nh, d_nh, 4, hipMemcpyDeviceToHost);
            printf("%llu subsets, %.0f s, %u first-word hits so far\n", subsets, ms / 1e3, nh); fflush(stdout);
        }
    }
    hipDeviceSynchronize();
    unsigned nh; u64 hits[96]; hipMemcpy(&nh, d_nh, 4, hipMemcpyDeviceToHost); hipMemcpy(hits, d_hits, sizeof hits, hipMemcpyDeviceToHost);
    printf("MAXS=%d form=%d: %llu subsets x 12! orderings, %u first-word hits (expected by chance %.3f)\n", MAXS, form, subsets, nh,
           (double)subsets * TOTAL * 2 * nt / 1.8446744e19);
    for (unsigned k = 0; k < nh && k < 32; k++) printf("  HIT mask=0x%lx perm_index=%lu order=%lu\n", hits[3 * k], hits[3 * k + 1], hits[3 * k + 2]);
    return 0;
}
