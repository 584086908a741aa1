// closed_forms_host.h — host side of the closed-form commitments (kernels: public_input_kernels.cuh)
#pragma once
#include "zkw_ctx.h"
#include "public_input_kernels.cuh"

// a20 for the 4-wide log-queue circuits: compact closed-form inputs [ni][18] followed by the public inputs [ni][4]
// in one allocation (ClosedFormInputCompactForm::from_full_form + commit, postprocessing/mod.rs:353-369)
template <class T>
static int closed_form_public_inputs(zkw_ctx* ctx, const typename T::Inst* d_inst, size_t ni, u64** cf_pi) {
    if (!*cf_pi) HIP_TRY(dev_malloc((void**)cf_pi, ni * (COMPACT_FORM_LEN + 4) * sizeof(u64)));
    u64 *compact = *cf_pi, *pis = *cf_pi + COMPACT_FORM_LEN * ni;
    { Prof _p(ctx, "k_closed_form_commitments"); ZKW_LAUNCH_T(ctx, (k_closed_form_commitments<T>), "k_closed_form_commitments", (unsigned)ni, 64, d_inst, ni, compact); }
    ZKW_TRY(launch_check("k_closed_form_commitments"));
    { Prof _p(ctx, "k_commit_encodings"); ZKW_LAUNCH(ctx, k_commit_encodings, blocks_for(ni, 64), 64, compact, ni, (u32)COMPACT_FORM_LEN, pis); }
    return launch_check("k_commit_encodings");
}

// a20 for the circuits whose builders keep no compact forms themselves (3, 5, 6, 7, 10, 13): commitments straight from
// the instance records
template <class T>
static int closed_form_from_records(zkw_ctx* ctx, const void* instances, size_t n, uint64_t* compact, uint64_t* public_inputs) {
    const typename T::Inst* d_inst = nullptr;
    ZKW_TRY(ctx->in("cf_records", static_cast<const typename T::Inst*>(instances), n, &d_inst));
    u64 *d_cf = nullptr, *d_pi = nullptr;
    ZKW_TRY(ctx->out("cf_compact", reinterpret_cast<u64*>(compact), n * COMPACT_FORM_LEN, &d_cf));
    ZKW_TRY(ctx->out("cf_pi", reinterpret_cast<u64*>(public_inputs), n * 4, &d_pi));
    { Prof _p(ctx, "k_closed_form_commitments"); ZKW_LAUNCH_T(ctx, (k_closed_form_commitments<T>), "k_closed_form_commitments", (unsigned)n, 64, d_inst, n, d_cf); }
    ZKW_TRY(launch_check("k_closed_form_commitments"));
    { Prof _p(ctx, "k_commit_encodings"); ZKW_LAUNCH(ctx, k_commit_encodings, blocks_for(n, 64), 64, d_cf, n, (u32)COMPACT_FORM_LEN, d_pi); }
    ZKW_TRY(launch_check("k_commit_encodings"));
    ZKW_TRY(ctx->finish_out(reinterpret_cast<u64*>(compact), d_cf, n * COMPACT_FORM_LEN));
    ZKW_TRY(ctx->finish_out(reinterpret_cast<u64*>(public_inputs), d_pi, n * 4));
    return ctx->sync_if_host();
}

