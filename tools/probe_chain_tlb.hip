// probe_chain_tlb.hip — does the slow regime of the queue-chain kernel come from address translation? 16 chains per wave
// (quad form), each chain streams 48 B in and 32 B out per permutation, prefetched one step ahead like k_chain_full_q4.
// layout 0: chain-major (chain c owns [c * n_items ...]: 16 K..29 K concurrent streams, each on its own pages);
// layout 1: item-major (item i of all chains contiguous: the 16 chains of a wave read 768 contiguous bytes).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../era_zkevm_test_harness_amd/csrc/poseidon2.cuh"
using namespace p2;
__global__ __launch_bounds__(64) void k_chain(const uint4* in, uint4* out, size_t n_chains, size_t n_items, int layout, int iters) {
    const int lane = threadIdx.x & 63, j = lane & 3;
    const size_t chain = (size_t)blockIdx.x * 16 + (lane >> 2);
    Coop4 co; co.init(j);
    u64 x[3] = {1, 2, 3};
    auto at = [&](size_t i) -> size_t { return layout ? (i * n_chains + chain) * 3 : (chain * n_items + i) * 3; };
    uint4 nxt = j < 3 ? in[at(0) + j] : make_uint4(0, 0, 0, 0);
    for (int i = 0; i < iters; i++) {
        const uint4 cur = nxt;
        if (j < 3 && (size_t)(i + 1) < n_items) nxt = in[at(i + 1) + j];
        x[0] ^= ((u64)cur.y << 32) | cur.x; x[1] ^= ((u64)cur.w << 32) | cur.z;
        co.permute(x);
        if (j < 2) { const size_t o = layout ? ((size_t)i * n_chains + chain) * 2 + j : (chain * n_items + i) * 2 + j; out[o] = make_uint4((unsigned)x[0], (unsigned)(x[0] >> 32), (unsigned)x[1], (unsigned)(x[1] >> 32)); }
    }
}
int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    const size_t n_items = 136714;
    for (int layout = 0; layout < 2; layout++)
    for (unsigned waves : {256u, 512u, 600u, 1024u, 1805u}) {
        const size_t chains = (size_t)waves * 16;
        uint4 *in, *out;
        if (hipMalloc(&in, chains * n_items * 48) != hipSuccess || hipMalloc(&out, chains * n_items * 32) != hipSuccess) { printf("alloc failed at %u waves\n", waves); return 1; }
        hipMemset(in, 3, chains * n_items * 48);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a); hipLaunchKernelGGL(k_chain, dim3(waves), dim3(64), 0, 0, in, out, chains, n_items, layout, iters); hipEventRecord(b);
        hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b);
        printf("layout %s waves %4u (%6zu chains): %.2f us per step\n", layout ? "item-major " : "chain-major", waves, chains, ms * 1e3 / iters);
        hipFree(in); hipFree(out);
    }
    return 0;
}
