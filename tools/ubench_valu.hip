// ubench_valu.hip — developer microbenchmark: issue cost (cycles per wave64 instruction, one wave on a SIMD)
// of the integer-multiply flavours the Goldilocks kernels can be built from. Prints cycles/instruction
// for a dependent chain and for 4 independent chains.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint64_t u64; typedef uint32_t u32;
#define REP 256
template <int MODE, int ILP>
__global__ void k(u64* out, u64 seed, long long* cyc) {
    u64 a[ILP]; u32 b = (u32)seed | 1;
    for (int i = 0; i < ILP; i++) a[i] = seed + i * 0x9E3779B97F4A7C15ULL + threadIdx.x;
    long long t0 = clock64();
    for (int it = 0; it < 64; it++) {
#pragma unroll
        for (int r = 0; r < REP / ILP; r++) {
#pragma unroll
            for (int i = 0; i < ILP; i++) {
                if (MODE == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"((u32)a[i]), "v"(b) : "vcc");
                if (MODE == 1) { u32 x = (u32)a[i]; asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(b)); a[i] = x; }
                if (MODE == 2) { u32 x = (u32)a[i]; asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x) : "v"(b)); a[i] = x; }
                if (MODE == 3) { u32 x = (u32)a[i]; asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x) : "v"(b)); a[i] = x; }
                if (MODE == 4) { u32 x = (u32)a[i]; asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(x) : "v"(b)); a[i] = x; }
                if (MODE == 5) { asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(a[i]) : "v"(seed)); }
                if (MODE == 6) { double d = __longlong_as_double(a[i]); asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d) : "v"(1.0000001)); a[i] = __double_as_longlong(d); }
                if (MODE == 7) { u32 x = (u32)a[i]; asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(b)); a[i] = x; }
                if (MODE == 8) { u32 x = (u32)a[i]; asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf" : "+v"(x)); a[i] = x; }
                if (MODE == 9) { u32 x = (u32)a[i]; asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(x) : "v"(b)); a[i] = x; }
            }
        }
    }
    long long t1 = clock64();
    u64 s = 0; for (int i = 0; i < ILP; i++) s += a[i];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) *cyc = t1 - t0;
}
template <int MODE, int ILP> void run(const char* name) {
    u64* out; long long* cyc; hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 8);
    hipLaunchKernelGGL((k<MODE, ILP>), dim3(1), dim3(64), 0, 0, out, 12345ULL, cyc); hipDeviceSynchronize();
    hipLaunchKernelGGL((k<MODE, ILP>), dim3(1), dim3(64), 0, 0, out, 12345ULL, cyc); hipDeviceSynchronize();
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-22s ILP=%d: %6.2f clock64 ticks / instr\n", name, ILP, (double)c / (64.0 * REP));
    hipFree(out); hipFree(cyc);
}
int main() {
#define R(M, N) run<M, 1>(N); run<M, 4>(N);
    R(7, "v_add_u32") R(5, "v_lshl_add_u64") R(0, "v_mad_u64_u32") R(1, "v_mul_lo_u32") R(2, "v_mul_hi_u32")
    R(3, "v_mul_u32_u24") R(4, "v_mul_hi_u32_u24") R(9, "v_mad_u32_u24") R(6, "v_fma_f64") R(8, "v_mov_b32_dpp quad")
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeWallClockRate, 0); printf("wall clock rate kHz: %d\n", clk);
    hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0); printf("sclk kHz: %d\n", clk);
    return 0;
}
