#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
#include <vector>
__global__ void k_long(long long* out, long long ticks) { long long t0 = wall_clock64(); while (wall_clock64() - t0 < ticks) {} if (out) *out = t0; }
__global__ void k_short(int* p) { if (p) *p = 1; }
int main(int argc, char** argv) {
    int prio_long = argc > 1 ? atoi(argv[1]) : 0;  // 0: normal, 1: high priority for the long stream
    int lo, hi; hipDeviceGetStreamPriorityRange(&lo, &hi); printf("priority range lo=%d hi=%d\n", lo, hi);
    hipStream_t L; if (prio_long) hipStreamCreateWithPriority(&L, hipStreamNonBlocking, hi); else hipStreamCreateWithFlags(&L, hipStreamNonBlocking);
    const int NS = 40; std::vector<hipStream_t> S(NS); for (auto& s : S) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipLaunchKernelGGL(k_long, dim3(1), dim3(64), 0, L, nullptr, 100000000LL);  // 1 s at 100 MHz
    double worst = 0; int blocked = 0;
    for (int i = 0; i < NS; i++) {
        auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(k_short, dim3(1), dim3(64), 0, S[i], nullptr); hipStreamSynchronize(S[i]);
        double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        if (ms > worst) worst = ms; if (ms > 50) blocked++;
    }
    hipStreamSynchronize(L);
    printf("long stream %s priority: %d of %d short streams waited > 50 ms behind it (worst %.1f ms)\n", prio_long ? "HIGH" : "normal", blocked, NS, worst);
    return 0;
}
