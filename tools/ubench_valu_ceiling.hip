// ubench_valu_ceiling.hip — the VALU issue ceiling of an MI355X for the instruction classes the Goldilocks kernels are made of,
// MEASURED at 1 / 2 / 4 / 8 waves per SIMD on the whole chip (every CU busy: the clocks are the ones a real kernel sees).
//
// Why: bench.py's roofline_valu priced kernels against 256 CUs x 4 SIMDs x 2.4 GHz / 4 cycles. One kernel of the library ran at
// 1.07 of that "peak" (VERDICT r4, weak 2), so the peak was wrong. This program measures what one SIMD issues per second for
//   (0) v_add_u32                 plain 32-bit VALU
//   (1) v_mad_u64_u32             the 32 x 32 -> 64 multiply-add every field multiplication is built from
//   (2) v_add_co / v_addc_co      the carry chains around it
//   (3) gl::mul                   one Goldilocks multiplication (compiler-scheduled form of gl64.cuh), 4 independent chains per lane
//   (4) p2::permute               Poseidon2, one state per lane           (the mix of k_ram_fill_poseidon)
//   (5) p2::Coop4::permute        Poseidon2, one state per quad of lanes  (the mix of k_chain_full_q4)
// with no memory traffic inside the timed loop. Modes 0-2 have a known instruction count (asm volatile, REP per iteration);
// the last three classes (indices 32-34) are counted by `rocprofv3 --pmc SQ_INSTS_VALU` on this same binary (tools/run_round_profiles.sh step 5c,
// tools/make_valu_ceiling.py).
//
// Occupancy is pinned: a launch is 256 CUs x W workgroups of 256 threads (one wave per SIMD each) and every workgroup asks for
// 160 KiB / W of LDS, so at most W fit on a CU and all 256 x W are resident at once when the dispatcher spreads them evenly
// (a second round would show as a 2x step in the times: checked by the program, which prints "uneven" then).
//
// Output: one JSON object on stdout (profiles/r05/valu_ceiling.json).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I era_zkevm_test_harness_amd/csrc -I include -o tools/ubench_valu_ceiling tools/ubench_valu_ceiling.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "poseidon2.cuh"
using namespace p2;

#define REP 512
template <int MODE>
__global__ __launch_bounds__(256) void k_cls(u64* out, u64 seed, int iters) {
    extern __shared__ unsigned char lds[];
    u32 a[8];
    u32 b = (u32)seed | 1;
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = (u32)seed + i * 0x9E3779B9u + threadIdx.x;
    u64 m[4];
#pragma unroll
    for (int i = 0; i < 4; i++) m[i] = seed + i * 0x9E3779B97F4A7C15ULL + threadIdx.x;
    const u64 sm = seed * 0x5555555555555555ULL;  // a lane mask in an SGPR pair
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < REP / 8; r++) {
            if (MODE == 0) {
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            }
            if (MODE == 1) {
#pragma unroll
                for (int j = 0; j < 2; j++)
#pragma unroll
                    for (int i = 0; i < 4; i++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(m[i]) : "v"((u32)a[i]), "v"(b) : "vcc");
            }
            if (MODE == 2) {
#pragma unroll
                for (int i = 0; i < 8; i += 2)
                    asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %2, vcc" : "+v"(a[i]), "+v"(a[i + 1]) : "v"(b) : "vcc");
            }
#define CLS8(M, TXT) if (MODE == M) { _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile(TXT : "+v"(a[i]) : "v"(b), "s"(sm)); }
#define CLS8V(M, TXT) if (MODE == M) { _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile(TXT : "+v"(a[i]) : "v"(b), "s"(sm) : "vcc"); }
#define CLS4(M, TXT) if (MODE == M) { _Pragma("unroll") for (int j = 0; j < 2; j++) _Pragma("unroll") for (int i = 0; i < 4; i++) asm volatile(TXT : "+v"(m[i]) : "v"(b), "s"(sm), "v"(m[(i + 1) & 3])); }
#define CLS4V(M, TXT) if (MODE == M) { _Pragma("unroll") for (int j = 0; j < 2; j++) _Pragma("unroll") for (int i = 0; i < 4; i++) asm volatile(TXT : "+v"(m[i]) : "v"(b), "s"(sm), "v"(m[(i + 1) & 3]) : "vcc"); }
            CLS4(10, "v_lshl_add_u64 %0, %0, 1, %3")
            CLS8(11, "v_cndmask_b32_e64 %0, %0, %1, %2")
            CLS4V(12, "v_cmp_gt_u64_e32 vcc, %0, %3")
            CLS8(13, "v_mov_b32_dpp %0, %0 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf")
            CLS8(14, "v_mul_lo_u32 %0, %0, %1")
            CLS8(15, "v_mul_hi_u32 %0, %0, %1")
            CLS8(16, "v_add3_u32 %0, %0, %1, %1")
            CLS4(17, "v_lshlrev_b64 %0, 1, %0")
            CLS8(18, "v_add_co_u32_e64 %0, s[20:21], %0, %1")
            CLS8(19, "v_alignbit_b32 %0, %0, %1, 7")
            CLS8(20, "v_mad_u32_u24 %0, %0, %1, %0")
            CLS8(21, "v_addc_co_u32_e64 %0, s[20:21], %0, %1, %2")
            CLS8(22, "v_xor_b32 %0, %0, %1")
            CLS8(23, "v_mov_b32 %0, %1")
            CLS8(24, "v_add_u32_e64 %0, %0, %1")
            CLS8(25, "v_fma_f32 %0, %0, %1, %0")
            CLS8(26, "v_fmac_f32_e32 %0, %0, %1")
            CLS8V(27, "v_cndmask_b32_e32 %0, %0, %1, vcc")
            CLS8V(28, "v_add_co_u32_e32 %0, vcc, %0, %1")
            CLS8(29, "v_and_or_b32 %0, %0, %1, %1")
            CLS8(30, "v_lshlrev_b32_e32 %0, 3, %0")
            CLS8(31, "v_mul_u32_u24_e32 %0, %0, %1")
            CLS8V(32, "v_addc_co_u32_e32 %0, vcc, %0, %1, vcc")
            CLS8(33, "v_sub_u32_e32 %0, %0, %1")
            CLS8(34, "v_add_u32_e32 %0, 0x12345, %0")
            CLS8(35, "v_add_u32_e32 %0, s20, %0")
            CLS8(36, "v_mul_hi_u32_u24_e32 %0, %0, %1")
            CLS8(37, "v_pk_add_u16 %0, %0, %1")
            CLS8(38, "v_mov_b32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1")
        }
    }
    u64 s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += a[i];
#pragma unroll
    for (int i = 0; i < 4; i++) s += m[i];
    if (s == 0x1234567ULL) out[blockIdx.x * 256 + threadIdx.x] = s + lds[threadIdx.x];  // (never: keeps the loop alive)
}

__global__ __launch_bounds__(256) void k_glmul(u64* out, u64 seed, int iters) {
    extern __shared__ unsigned char lds[];
    u64 m[4];
#pragma unroll
    for (int i = 0; i < 4; i++) m[i] = seed + i * 0x9E3779B97F4A7C15ULL + threadIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
#pragma unroll
            for (int i = 0; i < 4; i++) m[i] = gl::mul_lat(m[i], m[(i + 1) & 3]);
        }
    }
    u64 s = m[0] ^ m[1] ^ m[2] ^ m[3];
    if (s == 0x1234567ULL) out[blockIdx.x * 256 + threadIdx.x] = s + lds[threadIdx.x];
}

__global__ __launch_bounds__(256) void k_p2_lane(u64* out, u64 seed, int iters) {
    extern __shared__ unsigned char lds[];
    u64 s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = seed + i * 0x9E3779B97F4A7C15ULL + threadIdx.x;
    for (int it = 0; it < iters; it++) permute(s);
    u64 x = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) x ^= s[i];
    if (x == 0x1234567ULL) out[blockIdx.x * 256 + threadIdx.x] = x + lds[threadIdx.x];
}

__global__ __launch_bounds__(256) void k_p2_quad(u64* out, u64 seed, int iters) {
    extern __shared__ unsigned char lds[];
    Coop4 co;
    co.init(threadIdx.x & 3);
    u64 x[3];
#pragma unroll
    for (int i = 0; i < 3; i++) x[i] = seed + i * 0x9E3779B97F4A7C15ULL + threadIdx.x;
    for (int it = 0; it < iters; it++) co.permute(x);
    u64 s = x[0] ^ x[1] ^ x[2];
    if (s == 0x1234567ULL) out[blockIdx.x * 256 + threadIdx.x] = s + lds[threadIdx.x];
}

struct Mode {
    const char* name;
    const void* fn;
    double wave_insts_per_iter;  // known for the asm classes, 0 = counted by SQ_INSTS_VALU
    double units_per_wave_iter;  // what one wave completes per iteration (multiplications, permutations)
    const char* unit;
    int iters;
};

int main(int argc, char** argv) {
    int only = argc > 1 ? atoi(argv[1]) : -1;  // one mode (for the PMC pass: one kernel name per dispatch anyway)
    int n_cu = 0;
    hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, 0);
    int sclk_khz = 0;
    hipDeviceGetAttribute(&sclk_khz, hipDeviceAttributeClockRate, 0);
    Mode modes[] = {
        {"v_add_u32", (const void*)k_cls<0>, REP, REP, "instructions", 4000},
        {"v_mad_u64_u32", (const void*)k_cls<1>, REP, REP, "instructions", 1000},
        {"v_add_co_u32+v_addc_co_u32", (const void*)k_cls<2>, REP, REP, "instructions", 4000},
        {"v_lshl_add_u64", (const void*)k_cls<10>, REP, REP, "instructions", 1000},
        {"v_cndmask_b32 (SGPR mask)", (const void*)k_cls<11>, REP, REP, "instructions", 2000},
        {"v_cmp_gt_u64", (const void*)k_cls<12>, REP, REP, "instructions", 1000},
        {"v_mov_b32_dpp quad_perm", (const void*)k_cls<13>, REP, REP, "instructions", 2000},
        {"v_mul_lo_u32", (const void*)k_cls<14>, REP, REP, "instructions", 1000},
        {"v_mul_hi_u32", (const void*)k_cls<15>, REP, REP, "instructions", 1000},
        {"v_add3_u32", (const void*)k_cls<16>, REP, REP, "instructions", 2000},
        {"v_lshlrev_b64", (const void*)k_cls<17>, REP, REP, "instructions", 1000},
        {"v_add_co_u32 (SGPR-pair carry out, independent)", (const void*)k_cls<18>, REP, REP, "instructions", 2000},
        {"v_alignbit_b32", (const void*)k_cls<19>, REP, REP, "instructions", 2000},
        {"v_mad_u32_u24", (const void*)k_cls<20>, REP, REP, "instructions", 2000},
        {"v_addc_co_u32 (SGPR-pair carry in and out, independent)", (const void*)k_cls<21>, REP, REP, "instructions", 2000},
        {"v_xor_b32", (const void*)k_cls<22>, REP, REP, "instructions", 2000},
        {"v_mov_b32", (const void*)k_cls<23>, REP, REP, "instructions", 2000},
        {"v_add_u32_e64 (VOP3 encoding of a VOP2 op)", (const void*)k_cls<24>, REP, REP, "instructions", 2000},
        {"v_fma_f32 (VOP3)", (const void*)k_cls<25>, REP, REP, "instructions", 2000},
        {"v_fmac_f32_e32 (VOP2)", (const void*)k_cls<26>, REP, REP, "instructions", 2000},
        {"v_cndmask_b32_e32 (vcc)", (const void*)k_cls<27>, REP, REP, "instructions", 2000},
        {"v_add_co_u32_e32 (vcc out, independent)", (const void*)k_cls<28>, REP, REP, "instructions", 2000},
        {"v_and_or_b32", (const void*)k_cls<29>, REP, REP, "instructions", 2000},
        {"v_lshlrev_b32_e32", (const void*)k_cls<30>, REP, REP, "instructions", 2000},
        {"v_mul_u32_u24_e32", (const void*)k_cls<31>, REP, REP, "instructions", 2000},
        {"v_addc_co_u32_e32 (vcc in and out, chained through vcc)", (const void*)k_cls<32>, REP, REP, "instructions", 2000},
        {"v_sub_u32_e32", (const void*)k_cls<33>, REP, REP, "instructions", 2000},
        {"v_add_u32_e32 with a 32-bit literal", (const void*)k_cls<34>, REP, REP, "instructions", 2000},
        {"v_add_u32_e32 with an SGPR operand", (const void*)k_cls<35>, REP, REP, "instructions", 2000},
        {"v_mul_hi_u32_u24_e32", (const void*)k_cls<36>, REP, REP, "instructions", 2000},
        {"v_pk_add_u16 (VOP3P)", (const void*)k_cls<37>, REP, REP, "instructions", 2000},
        {"v_mov_b32_sdwa", (const void*)k_cls<38>, REP, REP, "instructions", 2000},
        {"gl::mul x4 chains", (const void*)k_glmul, 0, 64 * 64, "multiplications", 2000},
        {"p2::permute (lane form)", (const void*)k_p2_lane, 0, 64, "permutations", 200},
        {"p2::Coop4::permute (quad form)", (const void*)k_p2_quad, 0, 16, "permutations", 400},
    };
    u64* out;
    hipMalloc(&out, (size_t)n_cu * 8 * 256 * 8);
    printf("{\"device_cus\": %d, \"sclk_khz\": %d, \"simds\": %d, \"classes\": [\n", n_cu, sclk_khz, n_cu * 4);
    bool first_mode = true;
    for (int mi = 0; mi < (int)(sizeof(modes) / sizeof(modes[0])); mi++) {
        if (only >= 0 && only != mi) continue;
        const Mode& md = modes[mi];
        printf("%s {\"class\": \"%s\", \"unit\": \"%s\", \"by_waves_per_simd\": {", first_mode ? "" : ",", md.name, md.unit);
        first_mode = false;
        bool first_w = true;
        for (int w : {1, 2, 4, 8}) {
            const size_t lds = (160 * 1024) / w / 256 * 256;
            hipFuncSetAttribute(md.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            int occ = 0;
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, md.fn, 256, lds);
            if (occ < w) continue;  // registers do not allow w waves per SIMD of this kernel
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            u64 seed = 12345;
            int iters = md.iters;
            void* args[] = {&out, &seed, &iters};
            float best = 1e30f, worst = 0;
            for (int rep = 0; rep < 4; rep++) {
                hipEventRecord(e0);
                if (hipLaunchKernel(md.fn, dim3(n_cu * w), dim3(256), args, lds, 0) != hipSuccess) { fprintf(stderr, "launch failed\n"); return 1; }
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (rep) { best = ms < best ? ms : best; worst = ms > worst ? ms : worst; }
            }
            const double waves = (double)n_cu * 4 * w;
            const double units_per_s = waves * md.units_per_wave_iter * md.iters / (best * 1e-3);
            printf("%s\"%d\": {\"ms\": %.4f, \"ms_worst\": %.4f, \"units_per_s\": %.6e, \"units_per_s_per_simd\": %.6e", first_w ? "" : ", ", w, best, worst,
                   units_per_s, units_per_s / (n_cu * 4.0));
            if (md.wave_insts_per_iter > 0) {
                const double ips = waves * md.wave_insts_per_iter * md.iters / (best * 1e-3);
                printf(", \"wave_insts_per_s\": %.6e, \"cycles_per_wave_inst_per_simd_at_2.4GHz\": %.3f", ips, 2.4e9 / (ips / (n_cu * 4.0)));
            }
            printf("}");
            first_w = false;
        }
        printf("}}\n");
    }
    printf("]}\n");
    hipFree(out);
    return 0;
}
