// ubench_fill.hip — store-only ceilings for the launch geometries of the trace fill kernels (zkw trace v1:
// 16 slots x 149 columns x 2^20 rows of u64, column-major). Build: hipcc --offload-arch=gfx950 -O3 -o ubench_fill ubench_fill.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned long long u64;
constexpr size_t N_ROWS = 1 << 20, COLS = 149, CAP = 136714, SLOT = N_ROWS * COLS;

// row-type fill: lane = cycle, writes `ncols` columns at row region*CAP + cycle (8-byte stores)
template <int NT>
__global__ __launch_bounds__(256) void k_rows8(u64* __restrict__ base, int region, int ncols, u64 v, size_t rstride = CAP) {
    const size_t cyc = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (cyc >= CAP) return;
    u64* q = base + (size_t)blockIdx.y * SLOT + (size_t)region * rstride + cyc;
    for (int c = 0; c < ncols; c++) {
        if (NT) __builtin_nontemporal_store(v + c, q + (size_t)c * N_ROWS); else q[(size_t)c * N_ROWS] = v + c;
    }
}
// two cycles per lane, 16-byte stores (region*CAP even assumed by choosing region 0/2/4)
__global__ __launch_bounds__(256) void k_rows16(u64* __restrict__ base, int region, int ncols, u64 v, size_t rstride = CAP) {
    const size_t pair = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (2 * pair >= CAP) return;
    ulonglong2* q = reinterpret_cast<ulonglong2*>(base + (size_t)blockIdx.y * SLOT + (size_t)region * rstride) + pair;
    for (int c = 0; c < ncols; c++) q[(size_t)c * (N_ROWS / 2)] = make_ulonglong2(v + c, v);
}
// the tail: rows [6*CAP, N_ROWS) of 148 columns, as k_ram_fill_tail does it (grid.x blocks stride over one column at a time)
__global__ __launch_bounds__(256) void k_tail_strided(u64* __restrict__ base) {
    u64* trace = base + (size_t)blockIdx.y * SLOT;
    const size_t first = 6 * CAP, n_pairs = (N_ROWS - first) / 2;
    const size_t stride = (size_t)gridDim.x * blockDim.x, tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int col = 0; col < 148; col++) {
        ulonglong2* c2 = reinterpret_cast<ulonglong2*>(trace + (size_t)col * N_ROWS + first);
        for (size_t k = tid; k < n_pairs; k += stride) c2[k] = make_ulonglong2(0, 0);
    }
}
// the tail with one block per (column, chunk): grid.x = 148 * chunks
__global__ __launch_bounds__(256) void k_tail_flat(u64* __restrict__ base, int chunks) {
    u64* trace = base + (size_t)blockIdx.y * SLOT;
    const int col = blockIdx.x / chunks, ch = blockIdx.x % chunks;
    const size_t first = 6 * CAP, n_pairs = (N_ROWS - first) / 2;
    const size_t per = (n_pairs + chunks - 1) / chunks, lo = ch * per, hi = lo + per < n_pairs ? lo + per : n_pairs;
    ulonglong2* c2 = reinterpret_cast<ulonglong2*>(trace + (size_t)col * N_ROWS + first);
    for (size_t k = lo + threadIdx.x; k < hi; k += 256) c2[k] = make_ulonglong2(0, 0);
}

int main() {
    const int NJ = 16;
    u64* d;
    if (hipMalloc(&d, NJ * SLOT * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    auto run = [&](const char* name, double bytes, auto launch) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(a);
        for (int r = 0; r < 5; r++) launch();
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%-34s %.3f ms/launch  %.1f GB/s\n", name, ms / 5, 5.0 * bytes / ms / 1e6);
    };
    const double row_bytes = (double)NJ * 148 * CAP * 8;
    const unsigned gx = (CAP + 255) / 256;
    run("rows 8B plain 148 cols", row_bytes, [&] { hipLaunchKernelGGL(k_rows8<0>, dim3(gx, NJ), dim3(256), 0, 0, d, 2, 148, 1ull); });
    run("rows 8B nt 148 cols", row_bytes, [&] { hipLaunchKernelGGL(k_rows8<1>, dim3(gx, NJ), dim3(256), 0, 0, d, 2, 148, 1ull); });
    run("rows 16B 148 cols", row_bytes, [&] { hipLaunchKernelGGL(k_rows16, dim3((gx + 1) / 2, NJ), dim3(256), 0, 0, d, 2, 148, 1ull); });
    run("rows 8B plain 148 cols, 64 jobs-ish", row_bytes, [&] { hipLaunchKernelGGL(k_rows8<0>, dim3(gx, NJ), dim3(256), 0, 0, d, 4, 148, 1ull); });
    for (int region : {1, 2, 3, 5}) {
        char nm[64];
        snprintf(nm, sizeof nm, "rows 8B region %d stride CAP", region);
        run(nm, row_bytes, [&] { hipLaunchKernelGGL(k_rows8<0>, dim3(gx, NJ), dim3(256), 0, 0, d, region, 148, 1ull, (size_t)CAP); });
        snprintf(nm, sizeof nm, "rows 8B region %d stride 136768", region);
        run(nm, row_bytes, [&] { hipLaunchKernelGGL(k_rows8<0>, dim3(gx, NJ), dim3(256), 0, 0, d, region, 148, 1ull, (size_t)136768); });
        snprintf(nm, sizeof nm, "rows 16B region %d stride 136768", region);
        run(nm, row_bytes, [&] { hipLaunchKernelGGL(k_rows16, dim3((gx + 1) / 2, NJ), dim3(256), 0, 0, d, region, 148, 1ull, (size_t)136768); });
    }
    const double tail_bytes = (double)NJ * 148 * (N_ROWS - 6 * CAP) * 8;
    for (int g : {128, 512, 2048})
        { char nm[64]; snprintf(nm, sizeof nm, "tail strided grid.x %d", g);
          run(nm, tail_bytes, [&] { hipLaunchKernelGGL(k_tail_strided, dim3(g, NJ), dim3(256), 0, 0, d); }); }
    for (int ch : {1, 4, 16, 64})
        { char nm[64]; snprintf(nm, sizeof nm, "tail flat chunks %d", ch);
          run(nm, tail_bytes, [&] { hipLaunchKernelGGL(k_tail_flat, dim3(148 * ch, NJ), dim3(256), 0, 0, d, ch); }); }
    run("hipMemsetAsync 16 slots", (double)NJ * SLOT * 8, [&] { hipMemsetAsync(d, 0, NJ * SLOT * 8, 0); });
    return 0;
}
