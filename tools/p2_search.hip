// p2_search.hip — research tool (not part of the product or the oracle): searches the ordering of the
// internal-diagonal shifts of boojum's Poseidon2 against the Merkle-node known-answer pairs harvested
// from the reference's committed proofs (tests/golden/merkle_kat_ram.txt: 31 level-16 sibling digests +
// the 16-entry cap of test_proofs/base_layer/basic_circuit_proof_8_0.json `quotient_query`).
// Usage: p2_search <kat file>   (runs on one gfx950 GPU, ~1 minute)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../era_zkevm_test_harness_amd/csrc/gl64.cuh"
#include "../era_zkevm_test_harness_amd/csrc/poseidon2_constants.h"

using gl::u64;
using gl::u32;

__constant__ u64 c_rc[360];
__constant__ u64 c_pre[64 * 12];  // states after the first 4 full rounds, per input
__constant__ u64 c_cap0[16];      // first word of each cap digest (canonical)
__constant__ int c_nin;

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

__global__ void k_search(u64 first, u64 count, int form, u64* hits, unsigned* n_hits) {
    const u32 VALS[12] = {0, 2, 3, 4, 5, 6, 8, 9, 11, 12, 13, 14};
    u64 idx = first + (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= first + count) return;
    // factoradic decode -> permutation of VALS
    u32 sh[12];
    {
        u32 avail[12];
        for (int i = 0; i < 12; i++) avail[i] = VALS[i];
        u64 r = idx;
        u64 fact = 39916800ULL;  // 11!
        for (int i = 0; i < 12; i++) {
            u32 d = (u32)(r / fact);
            r %= fact;
            sh[i] = avail[d];
            for (int k = d; k < 11 - i; k++) avail[k] = avail[k + 1];
            if (i < 11) fact /= (11 - i);
        }
    }
    for (int in = 0; in < c_nin; in++) {
        u64 s[12];
        for (int i = 0; i < 12; i++) s[i] = c_pre[in * 12 + i];
        int r = 4;
        for (int k = 0; k < 22; k++, r++) {
            s[0] = gl::pow7(gl::add(s[0], c_rc[12 * r]));
            u64 sum = s[0];
            for (int i = 1; i < 12; i++) sum = gl::add(sum, s[i]);
            for (int i = 0; i < 12; i++) {
                u64 m = gl::mul_pow2(s[i], sh[i]);
                u64 y = gl::add(m, sum);
                if (form == 1) y = gl::sub(y, s[i]);
                if (form == 2) y = gl::add(y, s[i]);
                s[i] = y;
            }
        }
        for (int k = 0; k < 4; k++, r++) {
            for (int i = 0; i < 12; i++) s[i] = gl::pow7(gl::add(s[i], c_rc[12 * r + i]));
            ext(s);
        }
        u64 o = gl::canon(s[0]);
        bool hit = false;
        for (int c = 0; c < 16; c++) hit |= (o == c_cap0[c]);
        if (hit) {
            unsigned k = atomicAdd(n_hits, 1u);
            if (k < 64) { hits[2 * k] = idx; hits[2 * k + 1] = (u64)in | ((u64)form << 32); }
        }
    }
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s kat.txt\n", argv[0]); return 2; }
    FILE* f = fopen(argv[1], "r");
    if (!f) { perror("kat"); return 2; }
    int n = 0;
    if (fscanf(f, "%d", &n) != 1) return 2;
    std::vector<u64> L(n * 4);
    for (int i = 0; i < n * 4; i++) if (fscanf(f, "%lu", &L[i]) != 1) return 2;
    u64 cap[64];
    for (int i = 0; i < 64; i++) if (fscanf(f, "%lu", &cap[i]) != 1) return 2;
    fclose(f);
    u64 cap0[16];
    for (int c = 0; c < 16; c++) cap0[c] = cap[4 * c] % gl::P;
    hipMemcpyToSymbol(HIP_SYMBOL(c_rc), P2_ROUND_CONSTANTS, sizeof(u64) * 360);
    hipMemcpyToSymbol(HIP_SYMBOL(c_cap0), cap0, sizeof cap0);
    u64* d_hits; unsigned* d_nh;
    hipMalloc(&d_hits, 128 * 8); hipMalloc(&d_nh, 4);
    const u64 TOTAL = 479001600ULL;
    for (int premds = 1; premds >= 0; premds--) {
        // inputs: X = L[0] against every other digest, both orders
        std::vector<u64> pre;
        int nin = 0;
        for (int y = 1; y < n && nin < 64; y++)
            for (int order = 0; order < 2 && nin < 64; order++) {
                u64 s[12] = {0};
                const u64* a = &L[0]; const u64* b = &L[4 * y];
                memcpy(s, order ? b : a, 32); memcpy(s + 4, order ? a : b, 32);
                if (premds) ext(s);
                for (int r = 0; r < 4; r++) { for (int i = 0; i < 12; i++) s[i] = gl::pow7(gl::add(s[i], P2_ROUND_CONSTANTS[12 * r + i])); ext(s); }
                pre.insert(pre.end(), s, s + 12);
                nin++;
            }
        hipMemcpyToSymbol(HIP_SYMBOL(c_pre), pre.data(), pre.size() * 8);
        hipMemcpyToSymbol(HIP_SYMBOL(c_nin), &nin, sizeof nin);
        for (int form = 0; form < 3; form++) {
            hipMemset(d_nh, 0, 4);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0);
            const u64 CH = 1ULL << 26;
            for (u64 first = 0; first < TOTAL; first += CH) {
                u64 cnt = TOTAL - first < CH ? TOTAL - first : CH;
                hipLaunchKernelGGL(k_search, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, 0, first, cnt, form, d_hits, d_nh);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            unsigned nh = 0; u64 hits[128];
            hipMemcpy(&nh, d_nh, 4, hipMemcpyDeviceToHost); hipMemcpy(hits, d_hits, sizeof hits, hipMemcpyDeviceToHost);
            printf("premds=%d form=%d inputs=%d: %u first-word hits in %.1f s (expected by chance: %.3f)\n", premds, form, nin, nh, ms / 1e3,
                   (double)TOTAL * nin * 16 / 1.8446744e19);
            for (unsigned k = 0; k < nh && k < 64; k++) printf("  HIT perm_index=%lu input=%lu\n", hits[2 * k], hits[2 * k + 1] & 0xffffffff);
            fflush(stdout);
        }
    }
    return 0;
}
