// probe_wave_placement.hip — where does the dispatcher put the waves of a grid of one-wave (or four-wave) workgroups?
// Every wave records HW_ID / XCC_ID while all waves of the launch are resident (they spin until the last one has
// arrived), then the host histograms waves per SIMD. Motivation: the queue-chain kernels want one wave per SIMD
// (the "cliff" beyond ~8 400 chains).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
__global__ void k_where(unsigned* out, unsigned* arrived, unsigned total_waves, int spin) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) / 64;
    if ((threadIdx.x & 63) == 0) {
        out[2 * wave] = hw;
        out[2 * wave + 1] = xcc;
        atomicAdd(arrived, 1u);
    }
    // keep the wave resident until everybody is (bounded, in case the grid does not fit)
    for (int i = 0; i < spin; i++) {
        if (__hip_atomic_load(arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= total_waves) break;
        __builtin_amdgcn_s_sleep(32);
    }
}
int main(int argc, char** argv) {
    const int wg_threads = argc > 1 ? atoi(argv[1]) : 64;
    for (unsigned waves : {256u, 512u, 525u, 600u, 768u, 1024u, 1200u, 1805u, 2048u}) {
        const unsigned wpw = wg_threads / 64, grid = (waves + wpw - 1) / wpw, total = grid * wpw;
        unsigned *out, *arr;
        hipMalloc(&out, total * 8); hipMalloc(&arr, 4); hipMemset(arr, 0, 4);
        hipLaunchKernelGGL(k_where, dim3(grid), dim3(wg_threads), 0, 0, out, arr, total, 200000);
        hipDeviceSynchronize();
        std::vector<unsigned> h(total * 2); hipMemcpy(h.data(), out, total * 8, hipMemcpyDeviceToHost);
        std::map<unsigned long long, int> per_simd, per_cu;
        for (unsigned w = 0; w < total; w++) {
            const unsigned hw = h[2 * w], xcc = h[2 * w + 1] & 0xf;
            const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            const unsigned long long cukey = ((unsigned long long)xcc << 16) | (se << 8) | (sh << 4) | cu;
            per_cu[cukey]++; per_simd[(cukey << 2) | simd]++;
        }
        int hist[9] = {0}, mx = 0;
        for (auto& kv : per_simd) { hist[kv.second < 8 ? kv.second : 8]++; if (kv.second > mx) mx = kv.second; }
        int cuh[17] = {0};
        for (auto& kv : per_cu) cuh[kv.second < 16 ? kv.second : 16]++;
        printf("wg=%3d waves=%5u: CUs used %3zu, SIMDs used %4zu, max waves/SIMD %d | SIMDs with 1/2/3/4 waves: %d %d %d %d | CUs with 1..8 waves:",
               wg_threads, total, per_cu.size(), per_simd.size(), mx, hist[1], hist[2], hist[3], hist[4]);
        for (int i = 1; i <= 8; i++) printf(" %d", cuh[i]);
        printf("\n");
        hipFree(out); hipFree(arr);
    }
    return 0;
}
