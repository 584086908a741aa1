// zkw_launch.h — one kernel body, two launch forms.
//
// A builder kernel is written as a __device__ BODY whose first parameter is its virtual block position (VB: index and count of the
// workgroups of ITS job) and launched through zkw_launch<Body, BS>(ctx, name, grid, args...):
//   * a context on its own (zkw_block_run, every direct call of a builder): k_single<Body> on the context's stream — the launch the
//     __global__ kernel of rounds 1-5 was;
//   * a context that belongs to a batch (zkw_blocks_run: K blocks in flight as fibers of one host thread, zkw_batch.h): nothing is launched.
//     The call leaves (kernel, grid, packed arguments) with the batch, and when every fiber of the batch has run into a point where it needs
//     results, the launches that the K blocks made of the same kernel leave as ONE launch of k_multi<Body>: a job table (the packed
//     argument tuples back to back) and the prefix sums of the jobs' grids; a workgroup finds its job by binary search over the prefix and
//     runs the body with its position inside that job. One launch per (stage, all blocks) instead of one per (stage, block).
// The body is the same code in both forms, so the results are the same bits.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <cstring>
#include <type_traits>
#include <utility>

namespace zkw {

struct VB {
    unsigned x, y, z, nx, ny, nz;  // this workgroup's position in its job's grid, and that grid
};

// a plain aggregate of the body's arguments (host and device lay it out alike: one compiler)
template <class... A> struct Tup;
template <> struct Tup<> {};
template <class H, class... R> struct Tup<H, R...> {
    H h;
    Tup<R...> r;
};
inline void tup_pack(Tup<>&) {}
template <class H, class... R> inline void tup_pack(Tup<H, R...>& t, const H& h, const R&... r) {
    t.h = h;
    tup_pack(t.r, r...);
}
template <auto Body, class... U> __device__ __forceinline__ void tup_apply(const VB& vb, const Tup<>&, const U&... u) { Body(vb, u...); }
template <auto Body, class H, class... R, class... U> __device__ __forceinline__ void tup_apply(const VB& vb, const Tup<H, R...>& t, const U&... u) {
    tup_apply<Body>(vb, t.r, u..., t.h);
}

template <auto Body, int BS, class... A> __global__ __launch_bounds__(BS) void k_single(A... a) {
    Body(VB{blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x, gridDim.y, gridDim.z}, a...);
}
// prefix[j] <= blockIdx.x < prefix[j + 1]: job j; gxy[2 j], gxy[2 j + 1] = the job's grid in x and y (its z extent follows from the prefix)
template <auto Body, int BS, class... A> __global__ __launch_bounds__(BS) void k_multi(const Tup<A...>* __restrict__ jobs, const unsigned* __restrict__ prefix,
                                                                                      const unsigned* __restrict__ gxy, int n_jobs) {
    int lo = 0, hi = n_jobs;
    while (hi - lo > 1) {
        const int m = (lo + hi) >> 1;
        if (prefix[m] <= blockIdx.x) lo = m; else hi = m;
    }
    const unsigned local = blockIdx.x - prefix[lo], total = prefix[lo + 1] - prefix[lo], nx = gxy[2 * lo], ny = gxy[2 * lo + 1];
    // (the divisions run on the vector unit; the results are wave-uniform and belong in scalar registers — a 1024-thread kernel has 64
    // vector registers per lane and none to spare for six copies of a constant)
    auto uni = [](unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); };
    tup_apply<Body>(VB{uni(local % nx), uni((local / nx) % ny), uni(local / (nx * ny)), nx, ny, uni(total / (nx * ny))}, jobs[uni((unsigned)lo)]);
}

// what the batch needs to know about a kernel to merge its launches
struct BatchKernel {
    const void* multi_fn;  // k_multi<Body, BS, A...>
    unsigned bs, tup_bytes, tup_align;
    const char* name;
};

// A body may take a large descriptor as `const D&`: as a launch of its own the reference binds to the kernel argument, as a job of a merged
// launch to the job table in global memory — by value it would be copied into scratch there (it is indexed dynamically). The argument tuple
// holds the decayed types.
template <class F> struct LaunchSig;
template <class... P> struct LaunchSig<void (*)(const VB&, P...)> {
    template <class X> using Dec = typename std::remove_cv<typename std::remove_reference<X>::type>::type;
    using T = Tup<Dec<P>...>;
    template <auto Body, int BS> static void single(hipStream_t st, dim3 grid, size_t lds_bytes, const Dec<P>&... a) {
        hipLaunchKernelGGL((k_single<Body, BS, Dec<P>...>), grid, dim3(BS), lds_bytes, st, a...);
    }
    template <auto Body, int BS> static const void* single_fn() { return reinterpret_cast<const void*>(&k_single<Body, BS, Dec<P>...>); }
    template <auto Body, int BS> static const BatchKernel* desc(const char* name) {
        static BatchKernel k{reinterpret_cast<const void*>(&k_multi<Body, BS, Dec<P>...>), (unsigned)BS, (unsigned)sizeof(T), (unsigned)alignof(T), name};
        if (name && name[0] && !k.name[0]) k.name = name;
        return &k;
    }
    static void pack(T& t, const Dec<P>&... a) { tup_pack(t, a...); }
};

}  // namespace zkw
