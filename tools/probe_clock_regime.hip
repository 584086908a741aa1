// probe_clock_regime.hip — is the "slow regime" of the queue-chain kernels (14.7 -> 22.3 us per step beyond
// ~8 400 concurrent chains, after a few seconds of load) a clock state? Every wave runs the chain kernel's arithmetic
// (the quad-form Poseidon2 permutation, p2::Coop4) for `iters` dependent permutations and reads both counters around it:
// clock64() = s_memtime (shader cycles) and wall_clock64() = s_memrealtime (constant 100 MHz). Their ratio is the
// effective shader clock the wave saw; wall time per permutation and cycles per permutation separate "slower clock" from
// "more cycles" (contention).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "../era_zkevm_test_harness_amd/csrc/poseidon2.cuh"
using namespace p2;
__global__ __launch_bounds__(64) void k_load(u64* io, long long* stamps, int iters) {
    const int j = threadIdx.x & 3;
    Coop4 co; co.init(j);
    u64 x[3];
    for (int c = 0; c < 3; c++) x[c] = io[(blockIdx.x * 64 + threadIdx.x) * 3 + c];
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; i++) co.permute(x);
    const long long c1 = clock64(), w1 = wall_clock64();
    for (int c = 0; c < 3; c++) io[(blockIdx.x * 64 + threadIdx.x) * 3 + c] = x[c];
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = w1 - w0; }
}
int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 100000;
    for (int rep = 0; rep < 2; rep++)
    for (unsigned waves : {256u, 512u, 600u, 768u, 1024u, 2048u}) {
        u64* io; long long* st;
        hipMalloc(&io, (size_t)waves * 64 * 3 * 8); hipMemset(io, 1, (size_t)waves * 64 * 3 * 8); hipMalloc(&st, waves * 16);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a); hipLaunchKernelGGL(k_load, dim3(waves), dim3(64), 0, 0, io, st, iters); hipEventRecord(b);
        hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b);
        std::vector<long long> h(waves * 2); hipMemcpy(h.data(), st, waves * 16, hipMemcpyDeviceToHost);
        std::vector<double> mhz, cyc, us;
        for (unsigned w = 0; w < waves; w++) { mhz.push_back(100.0 * h[2 * w] / h[2 * w + 1]); cyc.push_back((double)h[2 * w] / iters); us.push_back(h[2 * w + 1] / 100.0 / iters); }
        std::sort(mhz.begin(), mhz.end()); std::sort(cyc.begin(), cyc.end()); std::sort(us.begin(), us.end());
        printf("waves %4u: kernel %.0f ms | effective clock MHz min/med/max %.0f %.0f %.0f | cycles per permutation med %.0f max %.0f | us per permutation med %.2f max %.2f\n",
               waves, ms, mhz.front(), mhz[waves / 2], mhz.back(), cyc[waves / 2], cyc.back(), us[waves / 2], us.back());
        hipFree(io); hipFree(st);
    }
    return 0;
}
