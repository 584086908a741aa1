// circuit_check_host.h — the spec-generic satisfiability check's host side (kernels: ram_circuit_kernels.cuh)
#pragma once
#include "zkw_ctx.h"
#include "ram_circuit_kernels.cuh"

template <class S>
static int check_satisfied(zkw_ctx* ctx, const zkw_trace* t, size_t slot, uint32_t capacity, uint64_t* n_violations, uint64_t* first_bad) {
    HIP_TRY(hipSetDevice(ctx->device));
    if (t->n_cols < (size_t)(S::G + S::L + 1)) return fail(ZKW_ERR_INVALID, "trace has %zu columns, the circuit needs %d", t->n_cols, S::G + S::L + 1);
    const u64* trace = t->data + slot * t->slot_elems();
    const size_t n_rows = t->n_rows;
    CheckResult* d_res = nullptr;
    u32* d_hist = nullptr;
    ZKW_TRY(ctx->scratch_t<CheckResult>("check_res", 1, &d_res));
    ZKW_TRY(ctx->scratch_t<u32>("check_hist", 256, &d_hist));
    CheckResult init{0ull, ~0ull};
    HIP_TRY(hipMemcpyAsync(d_res, &init, sizeof init, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemsetAsync(d_hist, 0, 256 * sizeof(u32), ctx->stream));
    const size_t lds = (size_t)(S::G + S::L) * CHK_ROWS * sizeof(u64);
    { Prof _p(ctx, "k_check_rows"); hipLaunchKernelGGL((k_check_rows<S>), dim3((capacity + CHK_ROWS - 1) / CHK_ROWS, S::NUM_ROW_TYPES), dim3(CHK_ROWS), lds, ctx->stream, trace, capacity, n_rows, d_res); }
    ZKW_TRY(launch_check("k_check_rows"));
    { Prof _p(ctx, "k_check_links"); hipLaunchKernelGGL((k_check_links<S>), dim3((capacity + 255) / 256), dim3(256), 0, ctx->stream, trace, capacity, n_rows, d_res); }
    ZKW_TRY(launch_check("k_check_links"));
    { Prof _p(ctx, "k_check_lookups"); hipLaunchKernelGGL((k_check_lookups<S>), dim3(1024), dim3(256), 0, ctx->stream, trace, capacity, n_rows, d_hist, d_res); }
    ZKW_TRY(launch_check("k_check_lookups"));
    { Prof _p(ctx, "k_check_mult"); hipLaunchKernelGGL(k_check_mult, dim3(256), dim3(256), 0, ctx->stream, trace, n_rows, S::G + S::L, d_hist, d_res); }
    ZKW_TRY(launch_check("k_check_mult"));
    CheckResult res;
    ZKW_TRY(ctx->read_small(&res, d_res, sizeof res));
    *n_violations = res.violations;
    if (first_bad) *first_bad = res.violations ? res.first_bad : 0;
    return ZKW_OK;
}
