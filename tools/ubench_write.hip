// ubench_write.hip — write-only HBM bandwidth on gfx950 by store flavour (plain / nontemporal, 8 / 16 bytes per lane,
// column-strided like the trace fills). Build: hipcc --offload-arch=gfx950 -O3 -o ubench_write ubench_write.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned long long u64;

template <int MODE>
__global__ __launch_bounds__(256) void k_write(u64* __restrict__ p, size_t n_u64, u64 v) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (MODE == 0) {  // 8 B plain
        for (; i < n_u64; i += stride) p[i] = v + i;
    } else if (MODE == 1) {  // 8 B nontemporal
        for (; i < n_u64; i += stride) __builtin_nontemporal_store(v + i, p + i);
    } else if (MODE == 2) {  // 16 B plain
        ulonglong2* q = reinterpret_cast<ulonglong2*>(p);
        for (; i < n_u64 / 2; i += stride) q[i] = make_ulonglong2(v + i, v);
    } else if (MODE == 3) {  // 16 B nontemporal
        typedef u64 v2 __attribute__((ext_vector_type(2)));
        v2* q = reinterpret_cast<v2*>(p);
        for (; i < n_u64 / 2; i += stride) { v2 x = {v + i, v}; __builtin_nontemporal_store(x, q + i); }
    }
}

// trace-like: block (x = row tile of 256 rows, y = column group); each lane writes `cols` columns, 8 B each, column stride n_rows
template <int NT>
__global__ __launch_bounds__(256) void k_cols(u64* __restrict__ p, size_t n_rows, int cols, u64 v) {
    const size_t row = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= n_rows) return;
    u64* q = p + (size_t)blockIdx.y * cols * n_rows + row;
    for (int c = 0; c < cols; c++) {
        if (NT) __builtin_nontemporal_store(v + c, q + (size_t)c * n_rows); else q[(size_t)c * n_rows] = v + c;
    }
}

int main() {
    const size_t bytes = 4ull << 30, n = bytes / 8;
    u64* d;
    hipMalloc(&d, bytes);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    auto run = [&](const char* name, auto launch) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(a);
        for (int r = 0; r < 5; r++) launch();
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%-28s %.1f GB/s\n", name, 5.0 * bytes / ms / 1e6);
    };
    for (int grid : {2048, 8192, 32768}) {
        printf("grid %d\n", grid);
        run("8B plain", [&] { hipLaunchKernelGGL(k_write<0>, dim3(grid), dim3(256), 0, 0, d, n, 1ull); });
        run("8B nt", [&] { hipLaunchKernelGGL(k_write<1>, dim3(grid), dim3(256), 0, 0, d, n, 1ull); });
        run("16B plain", [&] { hipLaunchKernelGGL(k_write<2>, dim3(grid), dim3(256), 0, 0, d, n, 1ull); });
        run("16B nt", [&] { hipLaunchKernelGGL(k_write<3>, dim3(grid), dim3(256), 0, 0, d, n, 1ull); });
    }
    run("hipMemsetAsync", [&] { hipMemsetAsync(d, 0, bytes, 0); });
    const size_t n_rows = 1 << 20;
    const int total_cols = (int)(n / n_rows);  // 512 columns of 8 MiB
    for (int cols : {8, 32, 128}) {
        char nm[64];
        snprintf(nm, sizeof nm, "cols %d plain", cols);
        run(nm, [&] { hipLaunchKernelGGL(k_cols<0>, dim3(n_rows / 256, total_cols / cols), dim3(256), 0, 0, d, n_rows, cols, 1ull); });
        snprintf(nm, sizeof nm, "cols %d nt", cols);
        run(nm, [&] { hipLaunchKernelGGL(k_cols<1>, dim3(n_rows / 256, total_cols / cols), dim3(256), 0, 0, d, n_rows, cols, 1ull); });
    }
    return 0;
}
