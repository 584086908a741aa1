// ubench_perm.hip — developer microbenchmark: latency of ONE Poseidon2 permutation in the cooperative row form
// (p2::Coop, what k_chain_full runs per queue item) on a single wave: ns per permutation over a dependent loop.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I<dir with gl64.cuh/poseidon2.cuh> -o ubench_perm ubench_perm.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include "poseidon2.cuh"
using namespace p2;
__global__ __launch_bounds__(64) void k_perm_loop(u64* io, int iters) {
    const int g = threadIdx.x & 15;
    Coop co; co.init(g);
    u64 x = io[threadIdx.x];
    for (int i = 0; i < iters; i++) x = co.permute(x);
    io[threadIdx.x] = x;
}
int main() {
    u64* io; hipMalloc(&io, 64 * 8);
    u64 h[64]; for (int i = 0; i < 64; i++) h[i] = (i & 15) < 12 ? 1000 + i : 0;
    for (int rep = 0; rep < 3; rep++) {
        hipMemcpy(io, h, sizeof h, hipMemcpyHostToDevice);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        const int iters = 20000;
        hipEventRecord(a); hipLaunchKernelGGL(k_perm_loop, dim3(1), dim3(64), 0, 0, io, iters); hipEventRecord(b);
        hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b);
        u64 o[64]; hipMemcpy(o, io, sizeof o, hipMemcpyDeviceToHost);
        printf("%.3f us per permutation (checksum %llx)\n", ms * 1e3 / iters, (unsigned long long)(o[0] ^ o[17] ^ o[35]));
    }
    return 0;
}
