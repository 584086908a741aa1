// How many long kernels on different streams run concurrently? N streams (normal or high priority), one 0.5 s kernel each.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <vector>
__global__ void k_long(long long ticks) { long long t0 = wall_clock64(); while (wall_clock64() - t0 < ticks) {} }
int main(int argc, char** argv) {
    int lo, hi; hipDeviceGetStreamPriorityRange(&lo, &hi);
    for (int prio = 0; prio < 2; prio++)
        for (int n : {1, 2, 4, 8, 16, 32}) {
            std::vector<hipStream_t> S(n);
            for (auto& s : S) { if (prio) hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi); else hipStreamCreateWithFlags(&s, hipStreamNonBlocking); }
            auto t0 = std::chrono::steady_clock::now();
            for (auto& s : S) hipLaunchKernelGGL(k_long, dim3(1), dim3(64), 0, s, 50000000LL);
            for (auto& s : S) hipStreamSynchronize(s);
            double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            printf("%s priority, %2d streams x 0.5 s kernel: %.0f ms\n", prio ? "HIGH  " : "normal", n, ms);
            for (auto& s : S) hipStreamDestroy(s);
        }
    return 0;
}
