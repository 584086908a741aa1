// zkw_sorters.hip — the four sorter / demuxer circuit types behind include/zkw.h: CodeDecommittmentsSorter (2), Events /
// L1Messages sorter (11, 12), LogDemuxer (4), StorageSorter (9): witness builders, synthesis, satisfiability checks.
#include "zkw_ctx.h"
#include "circuit_check_host.h"
#include "closed_forms_host.h"
#include "decommit_kernels.cuh"
#include "events_kernels.cuh"
#include "demux_kernels.cuh"
#include "storage_kernels.cuh"
#include "decommit_sorter_circuit_kernels.cuh"
#include "events_sorter_circuit_kernels.cuh"
#include "log_demux_circuit_kernels.cuh"
#include "storage_sorter_circuit_kernels.cuh"
#include "radix_sort.cuh"

// ------------------------------------------------------------------------------------------------ decommit sorter
struct zkw_decommit_witness {
    zkw_ctx* ctx = nullptr;
    size_t n = 0, n_instances = 0, n_dedup = 0;
    uint32_t capacity = 0;
    zkw_decommit_query *sorted_q = nullptr, *dedup_q = nullptr;
    u64 *unsorted_enc = nullptr, *sorted_enc = nullptr, *unsorted_tails = nullptr, *sorted_tails = nullptr;
    u64 *dedup_enc = nullptr, *dedup_tails = nullptr, *challenges = nullptr, *lhs_z = nullptr, *rhs_z = nullptr;
    zkw_decommit_sorter_instance* instances = nullptr;
    zkw_queue_state12 dedup_in;   // state of the deduplicated queue before the block (host copy)
    u32* fresh_prefix = nullptr;  // [n + 1], computed by the first synthesis call
    u64 *compact_forms = nullptr, *public_inputs = nullptr;  // [n_instances][18], [n_instances][4]
    u32 *fresh_count = nullptr, *last_fresh = nullptr;  // context scratch shared by the two phases of the builder
    bool finished = false;  // zkw_decommit_sorter_finish has run: tails, challenges, chains, instances are valid
    void release() {
        void* ptrs[] = {sorted_q, dedup_q, unsorted_enc, sorted_enc, unsorted_tails, sorted_tails, dedup_enc,
                        dedup_tails, challenges, lhs_z, rhs_z, instances, fresh_prefix, compact_forms, public_inputs};
        for (void* p : ptrs)
            if (p) dev_free(p);
    }
};

// phase 1 (contents): encodings, the stable (hash, timestamp) sort, the deduplicated queue. No hashing.
static int decommit_prepare(zkw_ctx* ctx, zkw_decommit_witness* w, const zkw_decommit_query* d_q) {
    const size_t n = w->n;
    const unsigned grid = blocks_for(n, 256);
    // unsorted side
    { Prof _p(ctx, "k_encode_decommit"); ZKW_LAUNCH(ctx, k_encode_decommit, grid, 256, d_q, n, w->unsorted_enc); }
    ZKW_TRY(launch_check("k_encode_decommit"));
    // sort: timestamp, then the hash from its least to its most significant 64 bits (stable LSD)
    u32 *ts = nullptr, *k32 = nullptr, *v0 = nullptr, *v1 = nullptr;
    u64 *hk[4] = {nullptr, nullptr, nullptr, nullptr}, *k64a = nullptr, *k64b = nullptr;
    void* tmp = nullptr;
    size_t tmp_bytes = radix_temp_bytes(n);
    ZKW_TRY(ctx->scratch_t<u32>("sort_ts", n, &ts));
    ZKW_TRY(ctx->scratch_t<u32>("sort_k32", n, &k32));
    ZKW_TRY(ctx->scratch_t<u32>("sort_v0", n, &v0));
    ZKW_TRY(ctx->scratch_t<u32>("sort_v1", n, &v1));
    const char* hn[4] = {"dsort_h0", "dsort_h1", "dsort_h2", "dsort_h3"};
    for (int k = 0; k < 4; k++) ZKW_TRY(ctx->scratch_t<u64>(hn[k], n, &hk[k]));
    ZKW_TRY(ctx->scratch_t<u64>("sort_k64a", n, &k64a));
    ZKW_TRY(ctx->scratch_t<u64>("sort_k64b", n, &k64b));
    ZKW_TRY(ctx->scratch("sort_tmp", tmp_bytes + 256, &tmp));
    { Prof _p(ctx, "k_decommit_sort_keys"); ZKW_LAUNCH(ctx, k_decommit_sort_keys, grid, 256, d_q, n, ts, hk[0], hk[1], hk[2], hk[3], v0); }
    ZKW_TRY(launch_check("k_decommit_sort_keys"));
    { Prof _p(ctx, "radix_sort"); ZKW_TRY(radix_sort_pairs<u32>(ctx, tmp, tmp_bytes, ts, k32, v0, v1, n, 32)); }
    u32 *cur = v1, *nxt = v0;
    for (int k = 0; k < 4; k++) {
        { Prof _p(ctx, "k_gather_u64_by_u32"); ZKW_LAUNCH(ctx, k_gather_u64_by_u32, grid, 256, hk[k], cur, n, k64a); }
        ZKW_TRY(launch_check("k_gather_u64_by_u32"));
        { Prof _p(ctx, "radix_sort"); ZKW_TRY(radix_sort_pairs<u64>(ctx, tmp, tmp_bytes, k64a, k64b, cur, nxt, n, 64)); }
        u32* t = cur; cur = nxt; nxt = t;
    }
    { Prof _p(ctx, "k_decommit_gather_encode"); ZKW_LAUNCH(ctx, k_decommit_gather_encode, grid, 256, d_q, cur, n, w->sorted_q, w->sorted_enc); }
    ZKW_TRY(launch_check("k_decommit_gather_encode"));
    // deduplicated queue = the fresh requests in sorted order
    u32 *fresh_count = nullptr, *last_fresh = nullptr, *totals = nullptr;
    ZKW_TRY(ctx->scratch_t<u32>("dec_fresh", n, &fresh_count));
    ZKW_TRY(ctx->scratch_t<u32>("dec_lastf", n, &last_fresh));
    ZKW_TRY(ctx->scratch_t<u32>("dec_totals", 2, &totals));
    u32 *fresh_prefix = nullptr, *fresh_pos = nullptr;
    ZKW_TRY(ctx->scratch_t<u32>("dec_fresh_prefix", n + 1, &fresh_prefix));
    ZKW_TRY(ctx->scratch_t<u32>("dec_fresh_pos", n, &fresh_pos));
    HIP_TRY(ctx->memset_async(totals, 0, 2 * sizeof(u32)));
    ZKW_TRY(flag_prefix(ctx, "k_decommit_fresh_prefix", DecommitFreshFlag{w->sorted_q}, n, fresh_prefix));
    { Prof _p(ctx, "k_decommit_dedup"); ZKW_LAUNCH(ctx, k_decommit_dedup, grid, 256, w->sorted_q, w->sorted_enc, n, fresh_prefix, fresh_count, fresh_pos, w->dedup_q, w->dedup_enc, totals); }
    ZKW_TRY(launch_check("k_decommit_dedup"));
    { Prof _p(ctx, "k_decommit_last_fresh"); ZKW_LAUNCH(ctx, k_decommit_last_fresh, grid, 256, fresh_count, fresh_pos, n, last_fresh); }
    ZKW_TRY(launch_check("k_decommit_last_fresh"));
    u32 h_totals[2] = {0, 0};
    ZKW_TRY(ctx->read_small(h_totals, totals, sizeof h_totals));
    if (h_totals[1]) return fail(ZKW_ERR_CHECK_FAILED, "decommit requests with the same hash disagree on page or are not "
                                                       "timestamp-ordered (sort_decommit_requests.rs:99-114)");
    w->n_dedup = h_totals[0];
    w->fresh_count = fresh_count;
    w->last_fresh = last_fresh;
    return ZKW_OK;
}

// phase 2 (hashes): the three queue chains in one launch, challenges, grand products, instance records
static int decommit_finish(zkw_ctx* ctx, zkw_decommit_witness* w) {
    const size_t n = w->n;
    const zkw_queue_state12& dedup_in = w->dedup_in;
    u32 *fresh_count = w->fresh_count, *last_fresh = w->last_fresh;
    zkw_queue_state12* d_dedup_in = nullptr;
    std::vector<zkw_queue_state12> din(1, dedup_in);
    ZKW_TRY(ctx->upload("dec_dedup_in", din, &d_dedup_in));
    std::vector<ChainJob> chains;
    chains.push_back(ChainJob{w->unsorted_enc, w->unsorted_tails, nullptr, n});
    chains.push_back(ChainJob{w->sorted_enc, w->sorted_tails, nullptr, n});
    chains.push_back(ChainJob{w->dedup_enc, w->dedup_tails, d_dedup_in->tail, w->n_dedup});
    ZKW_TRY(dev_chains(ctx, chains));
    std::vector<FsJob> fs(1);
    fs[0] = FsJob{w->unsorted_tails + 12 * (n - 1), w->sorted_tails + 12 * (n - 1), (u32)n, (u32)n, w->challenges};
    ZKW_TRY(dev_fs(ctx, fs, 12, 9));
    std::vector<GpSeg> segs;
    segs.push_back(GpSeg{w->unsorted_enc, w->lhs_z, w->challenges, n, 0, 0});
    segs.push_back(GpSeg{w->sorted_enc, w->rhs_z, w->challenges, n, 0, 0});
    ZKW_TRY(dev_grand_products(ctx, segs, 8, 2));
    std::vector<DecommitBlock> blk(1);
    blk[0] = DecommitBlock{w->sorted_q, w->unsorted_tails, w->sorted_tails, w->dedup_tails, w->lhs_z, w->rhs_z,
                           fresh_count, last_fresh, w->instances, dedup_in, n, w->capacity};
    DecommitBlock* d_blk = nullptr;
    ZKW_TRY(ctx->upload("dec_block", blk, &d_blk));
    { Prof _p(ctx, "k_decommit_instances"); ZKW_LAUNCH(ctx, k_decommit_instances, blocks_for(w->n_instances, 64), 64, d_blk); }
    return launch_check("k_decommit_instances");
}

extern "C" void zkw_decommit_witness_free(zkw_decommit_witness* w);

extern "C" int zkw_decommit_sorter_prepare(zkw_ctx* ctx, const zkw_decommit_query* q, size_t n, uint32_t capacity,
                                           const zkw_queue_state12* dedup_in, zkw_decommit_witness** out) {
    if (!ctx || !q || !out || capacity == 0) return fail(ZKW_ERR_INVALID, "zkw_decommit_sorter_prepare: bad argument");
    if (n == 0) return fail(ZKW_ERR_INVALID, "VM should have made some code decommits (sort_decommit_requests.rs:38-41)");
    if (n >= (1ull << 32)) return fail(ZKW_ERR_INVALID, "too many requests");
    HIP_TRY(hipSetDevice(ctx->device));
    zkw_decommit_witness* w = new zkw_decommit_witness();
    w->ctx = ctx;
    w->n = n;
    w->capacity = capacity;
    w->n_instances = (n + capacity - 1) / capacity;
    hipError_t e = hipSuccess;
    auto alloc = [&](void** p, size_t bytes) { if (e == hipSuccess) e = dev_malloc(p, bytes + 64); };
    alloc((void**)&w->sorted_q, n * sizeof(zkw_decommit_query));
    alloc((void**)&w->dedup_q, n * sizeof(zkw_decommit_query));
    alloc((void**)&w->unsorted_enc, n * 64); alloc((void**)&w->sorted_enc, n * 64); alloc((void**)&w->dedup_enc, n * 64);
    alloc((void**)&w->unsorted_tails, n * 96); alloc((void**)&w->sorted_tails, n * 96); alloc((void**)&w->dedup_tails, n * 96);
    alloc((void**)&w->challenges, 18 * 8); alloc((void**)&w->lhs_z, n * 16); alloc((void**)&w->rhs_z, n * 16);
    alloc((void**)&w->instances, w->n_instances * sizeof(zkw_decommit_sorter_instance));
    alloc((void**)&w->compact_forms, w->n_instances * COMPACT_FORM_LEN * 8);
    alloc((void**)&w->public_inputs, w->n_instances * 32);
    if (e != hipSuccess) {
        w->release();
        delete w;
        return fail(ZKW_ERR_OOM, "zkw_decommit_sorter_prepare: hipMalloc failed: %s", hipGetErrorString(e));
    }
    memset(&w->dedup_in, 0, sizeof w->dedup_in);
    if (dedup_in) w->dedup_in = *dedup_in;
    const zkw_decommit_query* d_q = nullptr;
    int rc = ctx->in("dec_q", q, n, &d_q);
    if (rc == ZKW_OK) rc = decommit_prepare(ctx, w, d_q);
    if (rc != ZKW_OK) {
        w->release();
        delete w;
        return rc;
    }
    ctx_retain(ctx);
    *out = w;
    return ZKW_OK;
}

extern "C" int zkw_decommit_sorter_finish(zkw_ctx* ctx, zkw_decommit_witness* w) {
    if (!ctx || !w || w->ctx != ctx) return fail(ZKW_ERR_INVALID, "zkw_decommit_sorter_finish: bad argument");
    if (w->finished) return ZKW_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t n = w->n;
    int rc = decommit_finish(ctx, w);
    if (rc == ZKW_OK) {  // a20: compact forms and public inputs (postprocessing/mod.rs:353-369)
        const size_t ni = w->n_instances;
        { Prof _p(ctx, "k_ds_commitments"); ZKW_LAUNCH(ctx, k_ds_commitments, blocks_for(4 * ni, 64), 64, w->instances, ni, w->compact_forms); }
        rc = launch_check("k_ds_commitments");
        if (rc == ZKW_OK) {
            { Prof _p(ctx, "k_commit_encodings"); ZKW_LAUNCH(ctx, k_commit_encodings, blocks_for(ni, 64), 64, w->compact_forms, ni, (u32)COMPACT_FORM_LEN, w->public_inputs); }
            rc = launch_check("k_commit_encodings");
        }
    }
    if (rc == ZKW_OK) rc = ctx->sync_if_host();
    if (rc == ZKW_OK && ctx->ptr_mode == ZKW_PTR_HOST) {  // lhs == rhs at the end (utils.rs:685-696)
        u64 ends[4];
        for (int r = 0; r < 2 && rc == ZKW_OK; r++) {
            if (hipMemcpy(&ends[2 * r], w->lhs_z + (size_t)r * n + n - 1, 8, hipMemcpyDeviceToHost) != hipSuccess ||
                hipMemcpy(&ends[2 * r + 1], w->rhs_z + (size_t)r * n + n - 1, 8, hipMemcpyDeviceToHost) != hipSuccess)
                rc = fail(ZKW_ERR_HIP, "readback failed");
            else if (ends[2 * r] != ends[2 * r + 1])
                rc = fail(ZKW_ERR_CHECK_FAILED, "grand products differ in repetition %d", r);
        }
    }
    if (rc == ZKW_OK) w->finished = true;
    return rc;
}

extern "C" int zkw_decommit_sorter_build(zkw_ctx* ctx, const zkw_decommit_query* q, size_t n, uint32_t capacity,
                                         const zkw_queue_state12* dedup_in, zkw_decommit_witness** out) {
    if (!out) return fail(ZKW_ERR_INVALID, "zkw_decommit_sorter_build: bad argument");
    zkw_decommit_witness* w = nullptr;
    ZKW_TRY(zkw_decommit_sorter_prepare(ctx, q, n, capacity, dedup_in, &w));
    const int rc = zkw_decommit_sorter_finish(ctx, w);
    if (rc != ZKW_OK) {
        zkw_decommit_witness_free(w);
        return rc;
    }
    *out = w;
    return ZKW_OK;
}

extern "C" size_t zkw_decommit_witness_num_instances(const zkw_decommit_witness* w) { return w ? w->n_instances : 0; }
extern "C" size_t zkw_decommit_witness_num_dedup(const zkw_decommit_witness* w) { return w ? w->n_dedup : 0; }

static const void* dec_array(const zkw_decommit_witness* w, int what, size_t* bytes) {
    const size_t n = w->n, nd = w->n_dedup;
    switch (what) {
        case ZKW_DEC_SORTED_QUERIES: *bytes = n * sizeof(zkw_decommit_query); return w->sorted_q;
        case ZKW_DEC_UNSORTED_ENC: *bytes = n * 64; return w->unsorted_enc;
        case ZKW_DEC_SORTED_ENC: *bytes = n * 64; return w->sorted_enc;
        case ZKW_DEC_UNSORTED_TAILS: *bytes = n * 96; return w->unsorted_tails;
        case ZKW_DEC_SORTED_TAILS: *bytes = n * 96; return w->sorted_tails;
        case ZKW_DEC_DEDUP_QUERIES: *bytes = nd * sizeof(zkw_decommit_query); return w->dedup_q;
        case ZKW_DEC_DEDUP_TAILS: *bytes = nd * 96; return w->dedup_tails;
        case ZKW_DEC_CHALLENGES: *bytes = 18 * 8; return w->challenges;
        case ZKW_DEC_LHS_Z: *bytes = n * 16; return w->lhs_z;
        case ZKW_DEC_RHS_Z: *bytes = n * 16; return w->rhs_z;
        case ZKW_DEC_INSTANCES: *bytes = w->n_instances * sizeof(zkw_decommit_sorter_instance); return w->instances;
        case ZKW_DEC_COMPACT_FORMS: *bytes = w->n_instances * COMPACT_FORM_LEN * 8; return w->compact_forms;
        case ZKW_DEC_PUBLIC_INPUTS: *bytes = w->n_instances * 32; return w->public_inputs;
        default: *bytes = 0; return nullptr;
    }
}
extern "C" size_t zkw_decommit_witness_bytes(const zkw_decommit_witness* w, int what) {
    size_t b = 0;
    if (w) (void)dec_array(w, what, &b);
    return b;
}
extern "C" const void* zkw_decommit_witness_device_ptr(const zkw_decommit_witness* w, int what) {
    size_t b = 0;
    return w ? dec_array(w, what, &b) : nullptr;
}
extern "C" int zkw_decommit_witness_get(const zkw_decommit_witness* w, int what, void* dst, size_t dst_bytes) {
    if (!w || !dst) return fail(ZKW_ERR_INVALID, "zkw_decommit_witness_get: null argument");
    size_t bytes = 0;
    const void* src = dec_array(w, what, &bytes);
    if (!src && bytes == 0 && what > ZKW_DEC_PUBLIC_INPUTS) return fail(ZKW_ERR_INVALID, "unknown array %d", what);
    if (dst_bytes < bytes) return fail(ZKW_ERR_INVALID, "need %zu bytes, got %zu", bytes, dst_bytes);
    if (bytes == 0) return ZKW_OK;
    zkw_ctx* ctx = w->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(ctx->copy_async(dst, src, bytes, ctx->ptr_mode == ZKW_PTR_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost));
    return ctx->sync_if_host();
}
extern "C" void zkw_decommit_witness_free(zkw_decommit_witness* w) {
    if (!w) return;
    (void)hipSetDevice(w->ctx->device);
    (void)w->ctx->sync_stream();
    w->release();
    zkw_ctx* owner = w->ctx;
    delete w;
    ctx_release(owner);
}

// ------------------------------------------------------------------------------------------------ events sorter
struct zkw_events_witness {
    zkw_ctx* ctx = nullptr;
    size_t n = 0, n_instances = 0, n_result = 0;
    uint32_t capacity = 0;
    zkw_log_query *sorted_q = nullptr, *result_q = nullptr;
    u64* enc_all = nullptr;    // [3n][20]: unsorted | sorted | result (one array so that one prehash covers all)
    u64* tails_all = nullptr;  // [5n][4]: unsorted old | unsorted new | sorted old | sorted new | result new
    u64 *challenges = nullptr, *lhs_z = nullptr, *rhs_z = nullptr;
    zkw_events_sorter_instance* instances = nullptr;
    zkw_queue_state4 result_in;  // state of the result queue before the block (host copy)
    u32* kept_prefix = nullptr;  // [n + 1], computed by the first synthesis call
    u64* cf_pi = nullptr;        // compact forms [ni][18] | public inputs [ni][4]
    void release() {
        void* ptrs[] = {sorted_q, result_q, enc_all, tails_all, challenges, lhs_z, rhs_z, instances, kept_prefix, cf_pi};
        for (void* p : ptrs)
            if (p) dev_free(p);
    }
};

static int events_run(zkw_ctx* ctx, zkw_events_witness* w, const zkw_log_query* d_q, const zkw_queue_state4& result_in) {
    const size_t n = w->n;
    const unsigned grid = blocks_for(n, 256);
    u64 *u_enc = w->enc_all, *s_enc = w->enc_all + 20 * n, *r_enc = w->enc_all + 40 * n;
    u64 *u_old = w->tails_all, *u_new = u_old + 4 * n, *s_old = u_new + 4 * n, *s_new = s_old + 4 * n, *r_new = s_new + 4 * n;
    { Prof _p(ctx, "k_encode_log"); ZKW_LAUNCH(ctx, k_encode_log, grid, 256, d_q, n, (const u32*)nullptr, u_enc); }
    ZKW_TRY(launch_check("k_encode_log"));
    // stable sort by (timestamp, rollback): 33-bit key
    u64 *key = nullptr, *key_out = nullptr;
    u32 *v0 = nullptr, *v1 = nullptr, *kept = nullptr, *totals = nullptr;
    void* tmp = nullptr;
    size_t tmp_bytes = radix_temp_bytes(n);
    ZKW_TRY(ctx->scratch_t<u64>("sort_k64a", n, &key));
    ZKW_TRY(ctx->scratch_t<u64>("sort_k64b", n, &key_out));
    ZKW_TRY(ctx->scratch_t<u32>("sort_v0", n, &v0));
    ZKW_TRY(ctx->scratch_t<u32>("sort_v1", n, &v1));
    ZKW_TRY(ctx->scratch("sort_tmp", tmp_bytes + 256, &tmp));
    { Prof _p(ctx, "k_events_sort_keys"); ZKW_LAUNCH(ctx, k_events_sort_keys, grid, 256, d_q, n, key, v0); }
    ZKW_TRY(launch_check("k_events_sort_keys"));
    { Prof _p(ctx, "radix_sort"); ZKW_TRY(radix_sort_pairs<u64>(ctx, tmp, tmp_bytes, key, key_out, v0, v1, n, 33)); }
    { Prof _p(ctx, "k_log_gather_encode"); ZKW_LAUNCH(ctx, k_log_gather_encode, grid, 256, d_q, v1, n, w->sorted_q, s_enc); }
    ZKW_TRY(launch_check("k_log_gather_encode"));
    ZKW_TRY(ctx->scratch_t<u32>("evt_kept", n, &kept));
    ZKW_TRY(ctx->scratch_t<u32>("evt_totals", 2, &totals));
    u32* kept_prefix = nullptr;
    ZKW_TRY(ctx->scratch_t<u32>("evt_kept_prefix", n + 1, &kept_prefix));
    HIP_TRY(ctx->memset_async(totals, 0, 2 * sizeof(u32)));
    ZKW_TRY(flag_prefix(ctx, "k_events_kept_prefix", EventsKeptFlag{w->sorted_q, n}, n, kept_prefix));
    { Prof _p(ctx, "k_events_dedup"); ZKW_LAUNCH(ctx, k_events_dedup, grid, 256, w->sorted_q, n, kept_prefix, kept, w->result_q, r_enc, totals); }
    ZKW_TRY(launch_check("k_events_dedup"));
    u32 h_totals[2] = {0, 0};
    ZKW_TRY(ctx->read_small(h_totals, totals, sizeof h_totals));
    if (h_totals[1]) return fail(ZKW_ERR_CHECK_FAILED, "event queue is not a sequence of forward events each optionally followed by "
                                                       "its own rollback (events_sort_dedup.rs:344-356, 512-533): %u violations", h_totals[1]);
    w->n_result = h_totals[0];
    zkw_queue_state4* d_rin = nullptr;
    std::vector<zkw_queue_state4> rin(1, result_in);
    ZKW_TRY(ctx->upload("evt_result_in", rin, &d_rin));
    std::vector<LogChainJob> chains;
    chains.push_back(LogChainJob{u_enc, nullptr, u_old, u_new, nullptr, n});
    chains.push_back(LogChainJob{s_enc, nullptr, s_old, s_new, nullptr, n});
    chains.push_back(LogChainJob{r_enc, nullptr, nullptr, r_new, d_rin->tail, w->n_result});
    ZKW_TRY(dev_log_chains(ctx, w->enc_all, 3 * n, chains));
    std::vector<FsJob> fs(1);
    fs[0] = FsJob{u_new + 4 * (n - 1), s_new + 4 * (n - 1), (u32)n, (u32)n, w->challenges};
    ZKW_TRY(dev_fs(ctx, fs, 4, 21));
    std::vector<GpSeg> segs;
    segs.push_back(GpSeg{u_enc, w->lhs_z, w->challenges, n, 0, 0});
    segs.push_back(GpSeg{s_enc, w->rhs_z, w->challenges, n, 0, 0});
    ZKW_TRY(dev_grand_products(ctx, segs, 20, 2));
    std::vector<EventsBlock> blk(1);
    blk[0] = EventsBlock{w->sorted_q, u_new, s_new, r_new, w->lhs_z, w->rhs_z, kept, w->instances, result_in, n, w->capacity};
    EventsBlock* d_blk = nullptr;
    ZKW_TRY(ctx->upload("evt_block", blk, &d_blk));
    { Prof _p(ctx, "k_events_instances"); ZKW_LAUNCH(ctx, k_events_instances, blocks_for(w->n_instances, 64), 64, d_blk); }
    return launch_check("k_events_instances");
}

extern "C" int zkw_events_sorter_build(zkw_ctx* ctx, const zkw_log_query* q, size_t n, uint32_t capacity,
                                       const zkw_queue_state4* result_in, zkw_events_witness** out) {
    if (!ctx || !out || capacity == 0 || (n && !q)) return fail(ZKW_ERR_INVALID, "zkw_events_sorter_build: bad argument");
    if (n >= (1ull << 31)) return fail(ZKW_ERR_INVALID, "too many log queries");
    HIP_TRY(hipSetDevice(ctx->device));
    zkw_events_witness* w = new zkw_events_witness();
    w->ctx = ctx;
    w->n = n;
    w->capacity = capacity;
    w->n_instances = n ? (n + capacity - 1) / capacity : 1;
    const size_t m = n ? n : 1;
    hipError_t e = hipSuccess;
    auto alloc = [&](void** p, size_t bytes) { if (e == hipSuccess) e = dev_malloc(p, bytes + 64); };
    alloc((void**)&w->sorted_q, m * sizeof(zkw_log_query));
    alloc((void**)&w->result_q, m * sizeof(zkw_log_query));
    alloc((void**)&w->enc_all, 3 * m * 160);
    alloc((void**)&w->tails_all, 5 * m * 32);
    alloc((void**)&w->challenges, 42 * 8);
    alloc((void**)&w->lhs_z, m * 16);
    alloc((void**)&w->rhs_z, m * 16);
    alloc((void**)&w->instances, w->n_instances * sizeof(zkw_events_sorter_instance));
    if (e != hipSuccess) {
        w->release();
        delete w;
        return fail(ZKW_ERR_OOM, "zkw_events_sorter_build: hipMalloc failed: %s", hipGetErrorString(e));
    }
    zkw_queue_state4 rin;
    memset(&rin, 0, sizeof rin);
    if (result_in) rin = *result_in;
    w->result_in = rin;
    int rc = ZKW_OK;
    if (n == 0) {  // events_sort_dedup.rs:27-76: one dummy instance, accumulators forced to ONE
        zkw_events_sorter_instance inst;
        memset(&inst, 0, sizeof inst);
        inst.start_flag = inst.completion_flag = 1;
        for (int r = 0; r < 2; r++) {
            inst.hidden_fsm_input.lhs_accumulator[r] = inst.hidden_fsm_input.rhs_accumulator[r] = 1;
            inst.hidden_fsm_output.lhs_accumulator[r] = inst.hidden_fsm_output.rhs_accumulator[r] = 1;
        }
        // the circuit derives its challenges whatever the queue holds (the trace's closed-form section does): those of two empty queues
        u64* d_zero = nullptr;
        rc = ctx->scratch_t<u64>("empty_queue_tail", 4, &d_zero);
        if (rc == ZKW_OK && (hipMemcpy(w->instances, &inst, sizeof inst, hipMemcpyHostToDevice) != hipSuccess ||
                             ctx->memset_async(d_zero, 0, 4 * sizeof(u64)) != hipSuccess))
            rc = fail(ZKW_ERR_HIP, "copy failed");
        if (rc == ZKW_OK) {
            std::vector<FsJob> fs(1, FsJob{d_zero, d_zero, 0u, 0u, w->challenges});
            rc = dev_fs(ctx, fs, 4, 21);
        }
    } else {
        const zkw_log_query* d_q = nullptr;
        rc = ctx->in("evt_q", q, n, &d_q);
        if (rc == ZKW_OK) rc = events_run(ctx, w, d_q, rin);
    }
    if (rc == ZKW_OK) rc = closed_form_public_inputs<CfEventsSorter>(ctx, w->instances, w->n_instances, &w->cf_pi);
    if (rc == ZKW_OK) rc = ctx->sync_if_host();
    if (rc != ZKW_OK) {
        w->release();
        delete w;
        return rc;
    }
    ctx_retain(ctx);
    *out = w;
    return ZKW_OK;
}

extern "C" size_t zkw_events_witness_num_instances(const zkw_events_witness* w) { return w ? w->n_instances : 0; }
extern "C" size_t zkw_events_witness_num_results(const zkw_events_witness* w) { return w ? w->n_result : 0; }

static const void* evt_array(const zkw_events_witness* w, int what, size_t* bytes) {
    const size_t n = w->n, nr = w->n_result;
    switch (what) {
        case ZKW_EVT_SORTED_QUERIES: *bytes = n * sizeof(zkw_log_query); return w->sorted_q;
        case ZKW_EVT_UNSORTED_ENC: *bytes = n * 160; return w->enc_all;
        case ZKW_EVT_SORTED_ENC: *bytes = n * 160; return w->enc_all + 20 * n;
        case ZKW_EVT_UNSORTED_OLD_TAILS: *bytes = n * 32; return w->tails_all;
        case ZKW_EVT_UNSORTED_NEW_TAILS: *bytes = n * 32; return w->tails_all + 4 * n;
        case ZKW_EVT_SORTED_OLD_TAILS: *bytes = n * 32; return w->tails_all + 8 * n;
        case ZKW_EVT_SORTED_NEW_TAILS: *bytes = n * 32; return w->tails_all + 12 * n;
        case ZKW_EVT_RESULT_QUERIES: *bytes = nr * sizeof(zkw_log_query); return w->result_q;
        case ZKW_EVT_RESULT_NEW_TAILS: *bytes = nr * 32; return w->tails_all + 16 * n;
        case ZKW_EVT_CHALLENGES: *bytes = 42 * 8; return w->challenges;
        case ZKW_EVT_LHS_Z: *bytes = n * 16; return w->lhs_z;
        case ZKW_EVT_RHS_Z: *bytes = n * 16; return w->rhs_z;
        case ZKW_EVT_INSTANCES: *bytes = w->n_instances * sizeof(zkw_events_sorter_instance); return w->instances;
        case ZKW_EVT_COMPACT_FORMS: *bytes = w->n_instances * COMPACT_FORM_LEN * 8; return w->cf_pi;
        case ZKW_EVT_PUBLIC_INPUTS: *bytes = w->n_instances * 32; return w->cf_pi + COMPACT_FORM_LEN * w->n_instances;
        default: *bytes = 0; return nullptr;
    }
}
extern "C" size_t zkw_events_witness_bytes(const zkw_events_witness* w, int what) {
    size_t b = 0;
    if (w) (void)evt_array(w, what, &b);
    return b;
}
extern "C" const void* zkw_events_witness_device_ptr(const zkw_events_witness* w, int what) {
    size_t b = 0;
    return w ? evt_array(w, what, &b) : nullptr;
}
extern "C" int zkw_events_witness_get(const zkw_events_witness* w, int what, void* dst, size_t dst_bytes) {
    if (!w || !dst) return fail(ZKW_ERR_INVALID, "zkw_events_witness_get: null argument");
    if (what < 0 || what > ZKW_EVT_PUBLIC_INPUTS) return fail(ZKW_ERR_INVALID, "unknown array %d", what);
    size_t bytes = 0;
    const void* src = evt_array(w, what, &bytes);
    if (dst_bytes < bytes) return fail(ZKW_ERR_INVALID, "need %zu bytes, got %zu", bytes, dst_bytes);
    if (bytes == 0) return ZKW_OK;
    zkw_ctx* ctx = w->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(ctx->copy_async(dst, src, bytes, ctx->ptr_mode == ZKW_PTR_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost));
    return ctx->sync_if_host();
}
extern "C" void zkw_events_witness_free(zkw_events_witness* w) {
    if (!w) return;
    (void)hipSetDevice(w->ctx->device);
    (void)w->ctx->sync_stream();
    w->release();
    zkw_ctx* owner = w->ctx;
    delete w;
    ctx_release(owner);
}

// ------------------------------------------------------------------------------------------------ log demuxer
struct zkw_demux_witness {
    zkw_ctx* ctx = nullptr;
    size_t n = 0, n_instances = 0, routed = 0;
    uint32_t capacity = 0;
    uint64_t offsets[7] = {0, 0, 0, 0, 0, 0, 0};
    zkw_log_query* out_q = nullptr;
    u64* enc_all = nullptr;    // [2n][20]: input | routed
    u64* tails_all = nullptr;  // [4n][4]: in old | in new | out old | out new
    u64* d_offsets = nullptr;  // [8]
    u32* route_count = nullptr;  // [6][n] inclusive prefix counts per route (kept for synthesis)
    bool default_params = true;
    zkw_log_demux_instance* instances = nullptr;
    u64* cf_pi = nullptr;  // compact forms [ni][18] | public inputs [ni][4]
    void release() {
        void* ptrs[] = {out_q, enc_all, tails_all, d_offsets, route_count, instances, cf_pi};
        for (void* p : ptrs)
            if (p) dev_free(p);
    }
};

static int demux_run(zkw_ctx* ctx, zkw_demux_witness* w, const zkw_log_query* d_q, const zkw_demux_params& params) {
    const size_t n = w->n;
    u64 *in_enc = w->enc_all, *out_enc = w->enc_all + 20 * n;
    u64 *in_old = w->tails_all, *in_new = in_old + 4 * n, *out_old = in_new + 4 * n, *out_new = out_old + 4 * n;
    { Prof _p(ctx, "k_encode_log"); ZKW_LAUNCH(ctx, k_encode_log, blocks_for(n, 256), 256, d_q, n, (const u32*)nullptr, in_enc); }
    ZKW_TRY(launch_check("k_encode_log"));
    u32* route_count = w->route_count;
    HIP_TRY(ctx->memset_async(w->d_offsets, 0, 8 * sizeof(u64)));
    ZKW_TRY((route_prefix<6>(ctx, "k_demux_route_prefix", DemuxRoute{d_q, params}, n, route_count)));
    ZKW_LAUNCH(ctx, k_demux_offsets, 1, 64, route_count, n, w->d_offsets);
    ZKW_TRY(launch_check("k_demux_offsets"));
    { Prof _p(ctx, "k_demux_route"); ZKW_LAUNCH(ctx, k_demux_route, blocks_for(n, 256), 256, d_q, in_enc, n, params, route_count, w->out_q, out_enc, w->d_offsets); }
    ZKW_TRY(launch_check("k_demux_route"));
    u64 h_tot[8];
    ZKW_TRY(ctx->read_small(h_tot, w->d_offsets, sizeof h_tot));
    if (h_tot[7]) return fail(ZKW_ERR_CHECK_FAILED, "%llu log queries have an aux byte / shard / rollback combination the "
                                                    "reference treats as unreachable (log_demux.rs:174-249)", (unsigned long long)h_tot[7]);
    for (int k = 0; k < 7; k++) w->offsets[k] = h_tot[k];
    w->routed = h_tot[6];
    std::vector<LogChainJob> chains;
    chains.push_back(LogChainJob{in_enc, nullptr, in_old, in_new, nullptr, n});
    for (int k = 0; k < 6; k++) {
        const size_t lo = w->offsets[k], cnt = w->offsets[k + 1] - lo;
        chains.push_back(LogChainJob{out_enc + 20 * lo, nullptr, out_old + 4 * lo, out_new + 4 * lo, nullptr, cnt});
    }
    ZKW_TRY(dev_log_chains(ctx, w->enc_all, n + w->routed, chains));
    std::vector<DemuxBlock> blk(1);
    blk[0].in_new_tails = in_new;
    blk[0].out_new_tails = out_new;
    blk[0].route_count = route_count;
    blk[0].instances = w->instances;
    for (int k = 0; k < 7; k++) blk[0].offsets[k] = w->offsets[k];
    blk[0].n = n;
    blk[0].capacity = w->capacity;
    DemuxBlock* d_blk = nullptr;
    ZKW_TRY(ctx->upload("dmx_block", blk, &d_blk));
    { Prof _p(ctx, "k_demux_instances"); ZKW_LAUNCH(ctx, k_demux_instances, blocks_for(w->n_instances, 64), 64, d_blk); }
    return launch_check("k_demux_instances");
}

extern "C" int zkw_log_demux_build(zkw_ctx* ctx, const zkw_log_query* q, size_t n, uint32_t capacity,
                                   const zkw_demux_params* params, zkw_demux_witness** out) {
    if (!ctx || !out || capacity == 0 || (n && !q)) return fail(ZKW_ERR_INVALID, "zkw_log_demux_build: bad argument");
    if (n >= (1ull << 31)) return fail(ZKW_ERR_INVALID, "too many log queries");
    HIP_TRY(hipSetDevice(ctx->device));
    zkw_demux_witness* w = new zkw_demux_witness();
    w->ctx = ctx;
    w->n = n;
    w->capacity = capacity;
    w->n_instances = n ? (n + capacity - 1) / capacity : 1;
    const size_t m = n ? n : 1;
    hipError_t e = hipSuccess;
    auto alloc = [&](void** p, size_t bytes) { if (e == hipSuccess) e = dev_malloc(p, bytes + 64); };
    alloc((void**)&w->out_q, m * sizeof(zkw_log_query));
    alloc((void**)&w->enc_all, 2 * m * 160);
    alloc((void**)&w->tails_all, 4 * m * 32);
    alloc((void**)&w->d_offsets, 8 * 8);
    alloc((void**)&w->route_count, 6 * m * sizeof(u32));
    alloc((void**)&w->instances, w->n_instances * sizeof(zkw_log_demux_instance));
    if (e != hipSuccess) {
        w->release();
        delete w;
        return fail(ZKW_ERR_OOM, "zkw_log_demux_build: hipMalloc failed: %s", hipGetErrorString(e));
    }
    zkw_demux_params p = ZKW_DEMUX_PARAMS_DEFAULT;
    if (params) p = *params;
    {
        const zkw_demux_params d = ZKW_DEMUX_PARAMS_DEFAULT;
        w->default_params = memcmp(&p, &d, sizeof d) == 0;
    }
    int rc = ZKW_OK;
    if (n == 0) {  // log_demux.rs:51-107
        zkw_log_demux_instance inst;
        memset(&inst, 0, sizeof inst);
        inst.start_flag = inst.completion_flag = 1;
        if (hipMemcpy(w->instances, &inst, sizeof inst, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemset(w->d_offsets, 0, 64) != hipSuccess)
            rc = fail(ZKW_ERR_HIP, "copy failed");
    } else {
        const zkw_log_query* d_q = nullptr;
        rc = ctx->in("dmx_q", q, n, &d_q);
        if (rc == ZKW_OK) rc = demux_run(ctx, w, d_q, p);
    }
    if (rc == ZKW_OK) rc = closed_form_public_inputs<CfLogDemux>(ctx, w->instances, w->n_instances, &w->cf_pi);
    if (rc == ZKW_OK) rc = ctx->sync_if_host();
    if (rc != ZKW_OK) {
        w->release();
        delete w;
        return rc;
    }
    ctx_retain(ctx);
    *out = w;
    return ZKW_OK;
}

extern "C" size_t zkw_demux_witness_num_instances(const zkw_demux_witness* w) { return w ? w->n_instances : 0; }
static const void* dmx_array(const zkw_demux_witness* w, int what, size_t* bytes) {
    const size_t n = w->n, r = w->routed;
    switch (what) {
        case ZKW_DMX_IN_ENC: *bytes = n * 160; return w->enc_all;
        case ZKW_DMX_IN_OLD_TAILS: *bytes = n * 32; return w->tails_all;
        case ZKW_DMX_IN_NEW_TAILS: *bytes = n * 32; return w->tails_all + 4 * n;
        case ZKW_DMX_OUT_QUERIES: *bytes = r * sizeof(zkw_log_query); return w->out_q;
        case ZKW_DMX_OUT_ENC: *bytes = r * 160; return w->enc_all + 20 * n;
        case ZKW_DMX_OUT_OLD_TAILS: *bytes = r * 32; return w->tails_all + 8 * n;
        case ZKW_DMX_OUT_NEW_TAILS: *bytes = r * 32; return w->tails_all + 12 * n;
        case ZKW_DMX_OUT_OFFSETS: *bytes = 7 * 8; return w->d_offsets;
        case ZKW_DMX_INSTANCES: *bytes = w->n_instances * sizeof(zkw_log_demux_instance); return w->instances;
        case ZKW_DMX_COMPACT_FORMS: *bytes = w->n_instances * COMPACT_FORM_LEN * 8; return w->cf_pi;
        case ZKW_DMX_PUBLIC_INPUTS: *bytes = w->n_instances * 32; return w->cf_pi + COMPACT_FORM_LEN * w->n_instances;
        default: *bytes = 0; return nullptr;
    }
}
extern "C" size_t zkw_demux_witness_bytes(const zkw_demux_witness* w, int what) {
    size_t b = 0;
    if (w) (void)dmx_array(w, what, &b);
    return b;
}
extern "C" const void* zkw_demux_witness_device_ptr(const zkw_demux_witness* w, int what) {
    size_t b = 0;
    return w ? dmx_array(w, what, &b) : nullptr;
}
extern "C" int zkw_demux_witness_get(const zkw_demux_witness* w, int what, void* dst, size_t dst_bytes) {
    if (!w || !dst) return fail(ZKW_ERR_INVALID, "zkw_demux_witness_get: null argument");
    if (what < 0 || what > ZKW_DMX_PUBLIC_INPUTS) return fail(ZKW_ERR_INVALID, "unknown array %d", what);
    size_t bytes = 0;
    const void* src = dmx_array(w, what, &bytes);
    if (dst_bytes < bytes) return fail(ZKW_ERR_INVALID, "need %zu bytes, got %zu", bytes, dst_bytes);
    if (bytes == 0) return ZKW_OK;
    zkw_ctx* ctx = w->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(ctx->copy_async(dst, src, bytes, ctx->ptr_mode == ZKW_PTR_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost));
    return ctx->sync_if_host();
}
extern "C" void zkw_demux_witness_free(zkw_demux_witness* w) {
    if (!w) return;
    (void)hipSetDevice(w->ctx->device);
    (void)w->ctx->sync_stream();
    w->release();
    zkw_ctx* owner = w->ctx;
    delete w;
    ctx_release(owner);
}

// ------------------------------------------------------------------------------------------------ storage sorter
struct zkw_storage_witness {
    zkw_ctx* ctx = nullptr;
    size_t n = 0, n_instances = 0, n_result = 0;
    uint32_t capacity = 0;
    zkw_log_query *sorted_q = nullptr, *result_q = nullptr;
    u32* sorted_ext = nullptr;
    u64* enc_all = nullptr;    // [3n][20]: unsorted plain | sorted (ext) | result : the three hashed queues
    u64* lhs_enc = nullptr;    // [n][20]: unsorted with extended timestamp (permutation argument only)
    u64* tails_all = nullptr;  // [5n][4]
    u64 *challenges = nullptr, *lhs_z = nullptr, *rhs_z = nullptr;
    u32* scans = nullptr;  // [4][n]: D, S, R, E of the per-cell prefix passes (k_storage_ds / k_storage_emit; kept for synthesis)
    zkw_storage_sorter_instance* instances = nullptr;
    u64* cf_pi = nullptr;  // compact forms [ni][18] | public inputs [ni][4]
    void release() {
        void* ptrs[] = {sorted_q, result_q, sorted_ext, enc_all, lhs_enc, tails_all, challenges, lhs_z, rhs_z, scans, instances, cf_pi};
        for (void* p : ptrs)
            if (p) dev_free(p);
    }
};

static int storage_run(zkw_ctx* ctx, zkw_storage_witness* w, const zkw_log_query* d_q) {
    const size_t n = w->n;
    const unsigned grid = blocks_for(n, 256);
    u64 *u_enc = w->enc_all, *s_enc = w->enc_all + 20 * n, *r_enc = w->enc_all + 40 * n;
    u64 *u_old = w->tails_all, *u_new = u_old + 4 * n, *s_old = u_new + 4 * n, *s_new = s_old + 4 * n, *r_new = s_new + 4 * n;
    // sort keys; the initial order IS the extended timestamp, so 7 stable passes (key low..high, address low..high)
    u64 *kk[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}, *k64a = nullptr, *k64b = nullptr;
    u32 *a2 = nullptr, *k32a = nullptr, *k32b = nullptr, *v0 = nullptr, *v1 = nullptr;
    void* tmp = nullptr;
    size_t tmp_bytes = radix_temp_bytes(n);
    const char* kn[6] = {"ssort_k0", "ssort_k1", "ssort_k2", "ssort_k3", "ssort_a0", "ssort_a1"};
    for (int k = 0; k < 6; k++) ZKW_TRY(ctx->scratch_t<u64>(kn[k], n, &kk[k]));
    ZKW_TRY(ctx->scratch_t<u32>("ssort_a2", n, &a2));
    ZKW_TRY(ctx->scratch_t<u64>("sort_k64a", n, &k64a));
    ZKW_TRY(ctx->scratch_t<u64>("sort_k64b", n, &k64b));
    ZKW_TRY(ctx->scratch_t<u32>("sort_ts", n, &k32a));
    ZKW_TRY(ctx->scratch_t<u32>("sort_k32", n, &k32b));
    ZKW_TRY(ctx->scratch_t<u32>("sort_v0", n, &v0));
    ZKW_TRY(ctx->scratch_t<u32>("sort_v1", n, &v1));
    ZKW_TRY(ctx->scratch("sort_tmp", tmp_bytes + 256, &tmp));
    u32* iota = nullptr;
    ZKW_TRY(ctx->scratch_t<u32>("ssort_iota", n, &iota));
    { Prof _p(ctx, "k_storage_sort_keys"); ZKW_LAUNCH(ctx, k_storage_sort_keys, grid, 256, d_q, n, kk[0], kk[1], kk[2], kk[3], kk[4], kk[5], a2, iota); }
    ZKW_TRY(launch_check("k_storage_sort_keys"));
    // plain and extended encodings of the unsorted side
    { Prof _p(ctx, "k_encode_log"); ZKW_LAUNCH(ctx, k_encode_log, grid, 256, d_q, n, (const u32*)nullptr, u_enc); }
    ZKW_TRY(launch_check("k_encode_log"));
    { Prof _p(ctx, "k_encode_log"); ZKW_LAUNCH(ctx, k_encode_log, grid, 256, d_q, n, (const u32*)iota, w->lhs_enc); }
    ZKW_TRY(launch_check("k_encode_log"));
    HIP_TRY(ctx->copy_async(v0, iota, n * sizeof(u32), hipMemcpyDeviceToDevice));
    u32 *cur = v0, *nxt = v1;
    for (int k = 0; k < 6; k++) {
        { Prof _p(ctx, "k_gather_u64_by_u32"); ZKW_LAUNCH(ctx, k_gather_u64_by_u32, grid, 256, kk[k], cur, n, k64a); }
        ZKW_TRY(launch_check("k_gather_u64_by_u32"));
        { Prof _p(ctx, "radix_sort"); ZKW_TRY(radix_sort_pairs<u64>(ctx, tmp, tmp_bytes, k64a, k64b, cur, nxt, n, 64)); }
        u32* t = cur; cur = nxt; nxt = t;
    }
    { Prof _p(ctx, "k_gather_u32_by_u32"); ZKW_LAUNCH(ctx, k_gather_u32_by_u32, grid, 256, a2, cur, n, k32a); }
    ZKW_TRY(launch_check("k_gather_u32_by_u32"));
    { Prof _p(ctx, "radix_sort"); ZKW_TRY(radix_sort_pairs<u32>(ctx, tmp, tmp_bytes, k32a, k32b, cur, nxt, n, 32)); }
    { u32* t = cur; cur = nxt; nxt = t; }
    { Prof _p(ctx, "k_storage_gather_encode"); ZKW_LAUNCH(ctx, k_storage_gather_encode, grid, 256, d_q, cur, n, w->sorted_q, w->sorted_ext, s_enc); }
    ZKW_TRY(launch_check("k_storage_gather_encode"));
    // per-cell registers and the deduplicated queue
    StorageScan sc;
    u32* totals = nullptr;
    sc.D = reinterpret_cast<int*>(w->scans);
    sc.S = w->scans + n;
    sc.R = w->scans + 2 * n;
    sc.E = w->scans + 3 * n;
    ZKW_TRY(ctx->scratch_t<u32>("sto_totals", 2, &totals));
    {   // tiled prefix passes (storage_kernels.cuh): depth deltas, cell starts, reads at depth zero, emitting cells
        u64 *d_dpfx = nullptr, *d_dtot = nullptr;
        u32 *d_spfx = nullptr, *d_rpfx = nullptr, *d_epfx = nullptr, *d_first = nullptr;
        ZKW_TRY(ctx->scratch_t<u64>("sto_dpfx", n + 1, &d_dpfx));
        ZKW_TRY(ctx->scratch_t<u64>("sto_dtot", 1, &d_dtot));
        ZKW_TRY(ctx->scratch_t<u32>("sto_spfx", n + 1, &d_spfx));
        ZKW_TRY(ctx->scratch_t<u32>("sto_rpfx", n + 1, &d_rpfx));
        ZKW_TRY(ctx->scratch_t<u32>("sto_epfx", n + 1, &d_epfx));
        ZKW_TRY(ctx->scratch_t<u32>("sto_first", n, &d_first));
        HIP_TRY(ctx->memset_async(totals, 0, 2 * sizeof(u32)));
        ZKW_TRY((sum_prefix<1>(ctx, "k_storage_depth_prefix", StoDelta{w->sorted_q}, n, d_dpfx, d_dtot)));
        ZKW_TRY(flag_prefix(ctx, "k_storage_start_prefix", StoIsStart{w->sorted_q}, n, d_spfx));
        { Prof _p(ctx, "k_storage_first"); ZKW_LAUNCH(ctx, k_storage_first, grid, 256, d_spfx, n, d_first); }
        ZKW_TRY(launch_check("k_storage_first"));
        { Prof _p(ctx, "k_storage_ds"); ZKW_LAUNCH(ctx, k_storage_ds, grid, 256, w->sorted_q, n, d_dpfx, d_spfx, d_first, sc, totals + 1); }
        ZKW_TRY(launch_check("k_storage_ds"));
        ZKW_TRY(flag_prefix(ctx, "k_storage_read_prefix", StoReadAtZero{w->sorted_q, sc, totals + 1}, n, d_rpfx));
        ZKW_TRY(flag_prefix(ctx, "k_storage_emit_prefix", StoEmits{sc, d_rpfx, n}, n, d_epfx));
        { Prof _p(ctx, "k_storage_emit"); ZKW_LAUNCH(ctx, k_storage_emit, grid, 256, w->sorted_q, n, sc, d_rpfx, d_epfx, w->result_q, r_enc, totals); }
        ZKW_TRY(launch_check("k_storage_emit"));
    }
    u32 h_totals[2] = {0, 0};
    ZKW_TRY(ctx->read_small(h_totals, totals, sizeof h_totals));
    if (h_totals[1]) return fail(ZKW_ERR_CHECK_FAILED, "storage log is not a consistent history (%u violations of the asserts at "
                                                       "sort_storage_access.rs:64-203)", h_totals[1]);
    w->n_result = h_totals[0];
    std::vector<LogChainJob> chains;
    chains.push_back(LogChainJob{u_enc, nullptr, u_old, u_new, nullptr, n});
    chains.push_back(LogChainJob{s_enc, nullptr, s_old, s_new, nullptr, n});
    chains.push_back(LogChainJob{r_enc, nullptr, nullptr, r_new, nullptr, w->n_result});
    ZKW_TRY(dev_log_chains(ctx, w->enc_all, 3 * n, chains));
    std::vector<FsJob> fs(1);
    fs[0] = FsJob{u_new + 4 * (n - 1), s_new + 4 * (n - 1), (u32)n, (u32)n, w->challenges};
    ZKW_TRY(dev_fs(ctx, fs, 4, 21));
    std::vector<GpSeg> segs;
    segs.push_back(GpSeg{w->lhs_enc, w->lhs_z, w->challenges, n, 0, 0});
    segs.push_back(GpSeg{s_enc, w->rhs_z, w->challenges, n, 0, 0});
    ZKW_TRY(dev_grand_products(ctx, segs, 20, 2));
    std::vector<StorageBlock> blk(1);
    blk[0] = StorageBlock{w->sorted_q, w->sorted_ext, u_new, s_new, r_new, w->lhs_z, w->rhs_z, sc, w->instances, n, w->capacity};
    StorageBlock* d_blk = nullptr;
    ZKW_TRY(ctx->upload("sto_block", blk, &d_blk));
    { Prof _p(ctx, "k_storage_instances"); ZKW_LAUNCH(ctx, k_storage_instances, blocks_for(w->n_instances, 64), 64, d_blk); }
    return launch_check("k_storage_instances");
}

extern "C" int zkw_storage_sorter_build(zkw_ctx* ctx, const zkw_log_query* q, size_t n, uint32_t capacity,
                                        zkw_storage_witness** out) {
    if (!ctx || !out || capacity == 0 || (n && !q)) return fail(ZKW_ERR_INVALID, "zkw_storage_sorter_build: bad argument");
    if (n >= (1ull << 31)) return fail(ZKW_ERR_INVALID, "too many log queries");
    HIP_TRY(hipSetDevice(ctx->device));
    zkw_storage_witness* w = new zkw_storage_witness();
    w->ctx = ctx;
    w->n = n;
    w->capacity = capacity;
    w->n_instances = n ? (n + capacity - 1) / capacity : 1;
    const size_t m = n ? n : 1;
    hipError_t e = hipSuccess;
    auto alloc = [&](void** p, size_t bytes) { if (e == hipSuccess) e = dev_malloc(p, bytes + 64); };
    alloc((void**)&w->sorted_q, m * sizeof(zkw_log_query));
    alloc((void**)&w->result_q, m * sizeof(zkw_log_query));
    alloc((void**)&w->sorted_ext, m * 4);
    alloc((void**)&w->enc_all, 3 * m * 160);
    alloc((void**)&w->lhs_enc, m * 160);
    alloc((void**)&w->tails_all, 5 * m * 32);
    alloc((void**)&w->challenges, 42 * 8);
    alloc((void**)&w->lhs_z, m * 16);
    alloc((void**)&w->rhs_z, m * 16);
    alloc((void**)&w->scans, 4 * m * sizeof(u32));
    alloc((void**)&w->instances, w->n_instances * sizeof(zkw_storage_sorter_instance));
    if (e != hipSuccess) {
        w->release();
        delete w;
        return fail(ZKW_ERR_OOM, "zkw_storage_sorter_build: hipMalloc failed: %s", hipGetErrorString(e));
    }
    int rc = ZKW_OK;
    if (n == 0) {  // storage_sort_dedup.rs:23-70
        zkw_storage_sorter_instance inst;
        memset(&inst, 0, sizeof inst);
        inst.start_flag = inst.completion_flag = 1;
        for (int r = 0; r < 2; r++) inst.hidden_fsm_output.lhs_accumulator[r] = inst.hidden_fsm_output.rhs_accumulator[r] = 1;
        inst.hidden_fsm_output.cycle_idx = 4;
        // the circuit derives its challenges whatever the queue holds (the trace's closed-form section does): those of two empty queues
        u64* d_zero = nullptr;
        rc = ctx->scratch_t<u64>("empty_queue_tail", 4, &d_zero);
        if (rc == ZKW_OK && (hipMemcpy(w->instances, &inst, sizeof inst, hipMemcpyHostToDevice) != hipSuccess ||
                             ctx->memset_async(d_zero, 0, 4 * sizeof(u64)) != hipSuccess))
            rc = fail(ZKW_ERR_HIP, "copy failed");
        if (rc == ZKW_OK) {
            std::vector<FsJob> fs(1, FsJob{d_zero, d_zero, 0u, 0u, w->challenges});
            rc = dev_fs(ctx, fs, 4, 21);
        }
    } else {
        const zkw_log_query* d_q = nullptr;
        rc = ctx->in("sto_q", q, n, &d_q);
        if (rc == ZKW_OK) rc = storage_run(ctx, w, d_q);
    }
    if (rc == ZKW_OK) rc = closed_form_public_inputs<CfStorageSorter>(ctx, w->instances, w->n_instances, &w->cf_pi);
    if (rc == ZKW_OK) rc = ctx->sync_if_host();
    if (rc != ZKW_OK) {
        w->release();
        delete w;
        return rc;
    }
    ctx_retain(ctx);
    *out = w;
    return ZKW_OK;
}

extern "C" size_t zkw_storage_witness_num_instances(const zkw_storage_witness* w) { return w ? w->n_instances : 0; }
extern "C" size_t zkw_storage_witness_num_results(const zkw_storage_witness* w) { return w ? w->n_result : 0; }
static const void* sto_array(const zkw_storage_witness* w, int what, size_t* bytes) {
    const size_t n = w->n, nr = w->n_result;
    switch (what) {
        case ZKW_STO_SORTED_QUERIES: *bytes = n * sizeof(zkw_log_query); return w->sorted_q;
        case ZKW_STO_SORTED_EXT_TS: *bytes = n * 4; return w->sorted_ext;
        case ZKW_STO_UNSORTED_ENC: *bytes = n * 160; return w->enc_all;
        case ZKW_STO_LHS_ENC: *bytes = n * 160; return w->lhs_enc;
        case ZKW_STO_SORTED_ENC: *bytes = n * 160; return w->enc_all + 20 * n;
        case ZKW_STO_UNSORTED_OLD_TAILS: *bytes = n * 32; return w->tails_all;
        case ZKW_STO_UNSORTED_NEW_TAILS: *bytes = n * 32; return w->tails_all + 4 * n;
        case ZKW_STO_SORTED_OLD_TAILS: *bytes = n * 32; return w->tails_all + 8 * n;
        case ZKW_STO_SORTED_NEW_TAILS: *bytes = n * 32; return w->tails_all + 12 * n;
        case ZKW_STO_RESULT_QUERIES: *bytes = nr * sizeof(zkw_log_query); return w->result_q;
        case ZKW_STO_RESULT_NEW_TAILS: *bytes = nr * 32; return w->tails_all + 16 * n;
        case ZKW_STO_CHALLENGES: *bytes = 42 * 8; return w->challenges;
        case ZKW_STO_LHS_Z: *bytes = n * 16; return w->lhs_z;
        case ZKW_STO_RHS_Z: *bytes = n * 16; return w->rhs_z;
        case ZKW_STO_INSTANCES: *bytes = w->n_instances * sizeof(zkw_storage_sorter_instance); return w->instances;
        case ZKW_STO_COMPACT_FORMS: *bytes = w->n_instances * COMPACT_FORM_LEN * 8; return w->cf_pi;
        case ZKW_STO_PUBLIC_INPUTS: *bytes = w->n_instances * 32; return w->cf_pi + COMPACT_FORM_LEN * w->n_instances;
        default: *bytes = 0; return nullptr;
    }
}
extern "C" size_t zkw_storage_witness_bytes(const zkw_storage_witness* w, int what) {
    size_t b = 0;
    if (w) (void)sto_array(w, what, &b);
    return b;
}
extern "C" const void* zkw_storage_witness_device_ptr(const zkw_storage_witness* w, int what) {
    size_t b = 0;
    return w ? sto_array(w, what, &b) : nullptr;
}
extern "C" int zkw_storage_witness_get(const zkw_storage_witness* w, int what, void* dst, size_t dst_bytes) {
    if (!w || !dst) return fail(ZKW_ERR_INVALID, "zkw_storage_witness_get: null argument");
    if (what < 0 || what > ZKW_STO_PUBLIC_INPUTS) return fail(ZKW_ERR_INVALID, "unknown array %d", what);
    size_t bytes = 0;
    const void* src = sto_array(w, what, &bytes);
    if (dst_bytes < bytes) return fail(ZKW_ERR_INVALID, "need %zu bytes, got %zu", bytes, dst_bytes);
    if (bytes == 0) return ZKW_OK;
    zkw_ctx* ctx = w->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(ctx->copy_async(dst, src, bytes, ctx->ptr_mode == ZKW_PTR_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost));
    return ctx->sync_if_host();
}
extern "C" void zkw_storage_witness_free(zkw_storage_witness* w) {
    if (!w) return;
    (void)hipSetDevice(w->ctx->device);
    (void)w->ctx->sync_stream();
    w->release();
    zkw_ctx* owner = w->ctx;
    delete w;
    ctx_release(owner);
}

// ------------------------------------------------------------------------------------------------ decommit sorter synthesis (a21, type 2)
extern "C" int zkw_decommit_sorter_synthesize(zkw_ctx* ctx, const zkw_decommit_witness* cw, size_t first_instance, size_t n_instances,
                                              zkw_trace* t, size_t first_slot) {
    zkw_decommit_witness* w = const_cast<zkw_decommit_witness*>(cw);
    if (!ctx || !w || !t || w->ctx != ctx || t->ctx->device != ctx->device) return fail(ZKW_ERR_INVALID, "zkw_decommit_sorter_synthesize: bad argument");
    if (first_instance + n_instances > w->n_instances) return fail(ZKW_ERR_INVALID, "instance range out of bounds");
    if (n_instances > t->n_slots) return fail(ZKW_ERR_INVALID, "more instances (%zu) than trace slots (%zu)", n_instances, t->n_slots);
    const u32 capacity = w->capacity;
    const size_t n_rows = t->n_rows;
    if (DS_MIN_ROWS(capacity) > n_rows)
        return fail(ZKW_ERR_INVALID, "capacity %u needs %llu rows, trace has %zu", capacity, (unsigned long long)DS_MIN_ROWS(capacity), n_rows);
    if (n_rows & 1) return fail(ZKW_ERR_INVALID, "trace length must be even (it is a power of two in every circuit)");
    if (n_instances == 0) return ZKW_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    if (!w->fresh_prefix) {
        HIP_TRY(dev_malloc((void**)&w->fresh_prefix, (w->n + 2) * sizeof(u32)));
        ZKW_TRY(flag_prefix(ctx, "k_ds_fresh_prefix", DsFreshFlag{w->sorted_q}, w->n, w->fresh_prefix));
    }
    u32* d_hist = nullptr;
    ZKW_TRY(ctx->scratch_t<u32>("ds_hist", n_instances * 256, &d_hist));
    HIP_TRY(ctx->memset_async(d_hist, 0, n_instances * 256 * sizeof(u32)));
    SlotClaims claims(t);
    std::vector<DsSynthJob> jobs(n_instances);
    for (size_t k = 0; k < n_instances; k++) {
        DsSynthJob& j = jobs[k];
        j.inst = w->instances + first_instance + k;
        j.sorted_q = w->sorted_q;
        j.unsorted_enc = w->unsorted_enc; j.sorted_enc = w->sorted_enc;
        j.unsorted_tails = w->unsorted_tails; j.sorted_tails = w->sorted_tails;
        j.dedup_enc = w->dedup_enc; j.dedup_tails = w->dedup_tails;
        j.fresh_prefix = w->fresh_prefix;
        j.challenges = w->challenges;
        j.lhs_z = w->lhs_z; j.rhs_z = w->rhs_z;
        j.n_block = w->n;
        memcpy(j.rq_tail_in, w->dedup_in.tail, 96);
        j.rq_len_in = w->dedup_in.length;
        {   // a slot whose previous tenant was this layout keeps its zero padding rows (zkw_ctx.h slot_tag; every other writer resets the tag)
            const uint64_t tag = ((uint64_t)2 << 56) ^ ((uint64_t)capacity << 24) ^ (uint64_t)n_rows ^ 0x5A00000000000000ull;
            const size_t slot = (first_slot + k) % t->n_slots;
            bool clean = false;
            j.trace = claims.claim(slot, tag, &clean);
            j.tail_clean = clean;
        }
        j.hist = d_hist + 256 * k;
        j.public_input = w->public_inputs + 4 * (first_instance + k);
        j.first_inst = w->instances;  // one block per witness
    }
    DsSynthJob* d_jobs = nullptr;
    ZKW_TRY(ctx->upload("ds_jobs", jobs, &d_jobs));
    const unsigned nj = (unsigned)n_instances;
    const u32 rstride = (u32)DS_REGION_STRIDE(capacity);
    const dim3 g64((rstride + 63) / 64, nj), g256((rstride + 255) / 256, nj);
    { Prof _p(ctx, "k_ds_fill_poseidon"); ZKW_LAUNCH_D(ctx, (k_ds_fill_poseidon<0>), "k_ds_fill_poseidon", g64, 64, 0, d_jobs, capacity, n_rows); }
    ZKW_TRY(launch_check("k_ds_fill_poseidon<0>"));
    { Prof _p(ctx, "k_ds_fill_poseidon"); ZKW_LAUNCH_D(ctx, (k_ds_fill_poseidon<1>), "k_ds_fill_poseidon", g64, 64, 0, d_jobs, capacity, n_rows); }
    ZKW_TRY(launch_check("k_ds_fill_poseidon<1>"));
    { Prof _p(ctx, "k_ds_fill_poseidon"); ZKW_LAUNCH_D(ctx, (k_ds_fill_poseidon<2>), "k_ds_fill_poseidon", g64, 64, 0, d_jobs, capacity, n_rows); }
    ZKW_TRY(launch_check("k_ds_fill_poseidon<2>"));
    { Prof _p(ctx, "k_ds_fill_row_A"); ZKW_LAUNCH_D(ctx, (k_ds_fill_row<DS_ROW_A>), "k_ds_fill_row", g256, 256, 0, d_jobs, capacity, n_rows); }
    ZKW_TRY(launch_check("k_ds_fill_row<A>"));
    { Prof _p(ctx, "k_ds_fill_row_B"); ZKW_LAUNCH_D(ctx, (k_ds_fill_row<DS_ROW_B>), "k_ds_fill_row", g256, 256, 0, d_jobs, capacity, n_rows); }
    ZKW_TRY(launch_check("k_ds_fill_row<B>"));
    { Prof _p(ctx, "k_ds_fill_row_C"); ZKW_LAUNCH_D(ctx, (k_ds_fill_row<DS_ROW_C>), "k_ds_fill_row", g256, 256, 0, d_jobs, capacity, n_rows); }
    ZKW_TRY(launch_check("k_ds_fill_row<C>"));
    { Prof _p(ctx, "k_ds_fill_row_D"); ZKW_LAUNCH_D(ctx, (k_ds_fill_row<DS_ROW_D>), "k_ds_fill_row", g256, 256, 0, d_jobs, capacity, n_rows); }
    ZKW_TRY(launch_check("k_ds_fill_row<D>"));
    { Prof _p(ctx, "k_ds_fill_tail"); ZKW_LAUNCH(ctx, k_ds_fill_tail, nj * ((DS_G + DS_L + 1) * TAIL_CHUNKS + 1), 256, d_jobs, nj, capacity, n_rows); }
    ZKW_TRY(launch_check("k_ds_fill_tail"));
    return claims.commit_if(ctx->sync_if_host());
}

// ------------------------------------------------------------------------------------------------ events / L1 messages sorter synthesis (a21, types 11 / 12)
extern "C" int zkw_events_sorter_synthesize(zkw_ctx* ctx, const zkw_events_witness* cw, size_t first_instance, size_t n_instances,
                                            zkw_trace* t, size_t first_slot) {
    zkw_events_witness* w = const_cast<zkw_events_witness*>(cw);
    if (!ctx || !w || !t || w->ctx != ctx || t->ctx->device != ctx->device) return fail(ZKW_ERR_INVALID, "zkw_events_sorter_synthesize: bad argument");
    if (first_instance + n_instances > w->n_instances) return fail(ZKW_ERR_INVALID, "instance range out of bounds");
    if (n_instances > t->n_slots) return fail(ZKW_ERR_INVALID, "more instances (%zu) than trace slots (%zu)", n_instances, t->n_slots);
    const u32 capacity = w->capacity;
    const size_t n_rows = t->n_rows, n = w->n;
    if (ES_MIN_ROWS(capacity) > n_rows)
        return fail(ZKW_ERR_INVALID, "capacity %u needs %llu rows, trace has %zu", capacity, (unsigned long long)ES_MIN_ROWS(capacity), n_rows);
    if (n_rows & 1) return fail(ZKW_ERR_INVALID, "trace length must be even (it is a power of two in every circuit)");
    if (n_instances == 0) return ZKW_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    if (!w->kept_prefix) {
        HIP_TRY(dev_malloc((void**)&w->kept_prefix, (n + 2) * sizeof(u32)));
        ZKW_TRY(flag_prefix(ctx, "k_es_kept_prefix", EsKeptFlag{w->sorted_q, n}, n, w->kept_prefix));
    }
    u32* d_hist = nullptr;
    ZKW_TRY(ctx->scratch_t<u32>("es_hist", n_instances * 256, &d_hist));
    HIP_TRY(ctx->memset_async(d_hist, 0, n_instances * 256 * sizeof(u32)));
    const size_t m = n ? n : 1;
    u64 *u_enc = w->enc_all, *s_enc = w->enc_all + 20 * m;
    u64 *u_new = w->tails_all + 4 * m, *s_new = w->tails_all + 12 * m, *r_new = w->tails_all + 16 * m;
    SlotClaims claims(t);
    std::vector<EsSynthJob> jobs(n_instances);
    for (size_t k = 0; k < n_instances; k++) {
        EsSynthJob& j = jobs[k];
        j.inst = w->instances + first_instance + k;
        j.sorted_q = w->sorted_q;
        j.unsorted_enc = u_enc; j.sorted_enc = s_enc;
        j.unsorted_new_tails = u_new; j.sorted_new_tails = s_new; j.result_new_tails = r_new;
        j.kept_prefix = w->kept_prefix;
        j.challenges = w->challenges;
        j.lhs_z = w->lhs_z; j.rhs_z = w->rhs_z;
        j.n_block = n;
        memcpy(j.rq_tail_in, w->result_in.tail, 32);
        j.rq_len_in = w->result_in.length;
        j.public_input = w->cf_pi + COMPACT_FORM_LEN * w->n_instances + 4 * (first_instance + k);
        j.first_inst = w->instances;  // one block per witness
        {   // a slot whose previous tenant was this layout keeps its zero padding rows (zkw_ctx.h slot_tag; every other writer resets the tag)
            const uint64_t tag = ((uint64_t)11 << 56) ^ ((uint64_t)capacity << 24) ^ (uint64_t)n_rows ^ 0x5A00000000000000ull;
            const size_t slot = (first_slot + k) % t->n_slots;
            bool clean = false;
            j.trace = claims.claim(slot, tag, &clean);
            j.tail_clean = clean;
        }
        j.hist = d_hist + 256 * k;
    }
    EsSynthJob* d_jobs = nullptr;
    ZKW_TRY(ctx->upload("es_jobs", jobs, &d_jobs));
    const unsigned nj = (unsigned)n_instances;
    const u32 rstride = (u32)ES_REGION_STRIDE(capacity);
    const dim3 g64((rstride + 63) / 64, nj), g256((rstride + 255) / 256, nj);
    { Prof _p(ctx, "k_es_fill_queue"); ZKW_LAUNCH_D(ctx, (k_es_fill_queue<0>), "k_es_fill_queue", g64, 64, 0, d_jobs, capacity, n_rows); }
    ZKW_TRY(launch_check("k_es_fill_queue<0>"));
    { Prof _p(ctx, "k_es_fill_queue"); ZKW_LAUNCH_D(ctx, (k_es_fill_queue<1>), "k_es_fill_queue", g64, 64, 0, d_jobs, capacity, n_rows); }
    ZKW_TRY(launch_check("k_es_fill_queue<1>"));
    { Prof _p(ctx, "k_es_fill_queue"); ZKW_LAUNCH_D(ctx, (k_es_fill_queue<2>), "k_es_fill_queue", g64, 64, 0, d_jobs, capacity, n_rows); }
    ZKW_TRY(launch_check("k_es_fill_queue<2>"));
#define ES_LAUNCH_ROW(R) { Prof _p(ctx, "k_es_fill_row"); ZKW_LAUNCH_D(ctx, (k_es_fill_row<ES_ROW_##R>), "k_es_fill_row", g256, 256, 0, d_jobs, capacity, n_rows); } \
    ZKW_TRY(launch_check("k_es_fill_row<" #R ">"));
    ES_LAUNCH_ROW(A) ES_LAUNCH_ROW(NTV) ES_LAUNCH_ROW(W) ES_LAUNCH_ROW(Q)  // NTV after the queue kernels: it writes the range checks into their rows' lookup columns
#undef ES_LAUNCH_ROW
    { Prof _p(ctx, "k_es_fill_tail"); ZKW_LAUNCH(ctx, k_es_fill_tail, nj * ((ES_G + ES_L + 1) * TAIL_CHUNKS + 1), 256, d_jobs, nj, capacity, n_rows); }
    ZKW_TRY(launch_check("k_es_fill_tail"));
    return claims.commit_if(ctx->sync_if_host());
}

extern "C" int zkw_events_sorter_check_satisfied(zkw_ctx* ctx, const zkw_trace* t, size_t slot, uint32_t capacity,
                                                 uint64_t* n_violations, uint64_t* first_bad) {
    if (!ctx || !t || t->ctx->device != ctx->device || slot >= t->n_slots || !n_violations)
        return fail(ZKW_ERR_INVALID, "zkw_events_sorter_check_satisfied: bad argument");
    if (ES_MIN_ROWS(capacity) > t->n_rows) return fail(ZKW_ERR_INVALID, "capacity does not fit the trace");
    return check_satisfied<SpecEventsSorter>(ctx, t, slot, capacity, n_violations, first_bad);
}

// ------------------------------------------------------------------------------------------------ LogDemuxer synthesis
extern "C" int zkw_log_demux_synthesize(zkw_ctx* ctx, const zkw_demux_witness* w, size_t first_instance, size_t n_instances,
                                        zkw_trace* t, size_t first_slot) {
    if (!ctx || !w || !t || w->ctx != ctx || t->ctx->device != ctx->device) return fail(ZKW_ERR_INVALID, "zkw_log_demux_synthesize: bad argument");
    if (first_instance + n_instances > w->n_instances) return fail(ZKW_ERR_INVALID, "instance range out of bounds");
    if (n_instances > t->n_slots) return fail(ZKW_ERR_INVALID, "more instances (%zu) than trace slots (%zu)", n_instances, t->n_slots);
    if (!w->default_params)
        return fail(ZKW_ERR_INVALID, "the LogDemuxer circuit hard-wires ZKW_DEMUX_PARAMS_DEFAULT; this witness was built with other routing constants");
    if (t->n_cols < LD_COLS) return fail(ZKW_ERR_INVALID, "trace has %zu columns, the LogDemuxer needs %d (zkw_trace_create_with_columns)", t->n_cols, LD_COLS);
    const u32 capacity = w->capacity;
    const size_t n_rows = t->n_rows, n = w->n;
    if (LD_MIN_ROWS(capacity) > n_rows)
        return fail(ZKW_ERR_INVALID, "capacity %u needs %llu rows, trace has %zu", capacity, (unsigned long long)LD_MIN_ROWS(capacity), n_rows);
    if (n_rows & 1) return fail(ZKW_ERR_INVALID, "trace length must be even (it is a power of two in every circuit)");
    if (n_instances == 0) return ZKW_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    u32* d_hist = nullptr;
    ZKW_TRY(ctx->scratch_t<u32>("ld_hist", n_instances * 256, &d_hist));
    HIP_TRY(ctx->memset_async(d_hist, 0, n_instances * 256 * sizeof(u32)));
    SlotClaims claims(t);
    std::vector<LdSynthJob> jobs(n_instances);
    for (size_t k = 0; k < n_instances; k++) {
        LdSynthJob& j = jobs[k];
        j.inst = w->instances + first_instance + k;
        j.in_enc = w->enc_all;
        j.in_new_tails = w->tails_all + 4 * n;
        j.out_new_tails = w->tails_all + 12 * n;
        j.route_count = w->route_count;
        for (int c = 0; c < 7; c++) j.offsets[c] = w->offsets[c];
        j.n_block = n;
        j.public_input = w->cf_pi + COMPACT_FORM_LEN * w->n_instances + 4 * (first_instance + k);
        j.first_inst = w->instances;  // one block per witness
        {   // a slot whose previous tenant was this layout keeps its zero padding rows (zkw_ctx.h slot_tag; every other writer resets the tag)
            const uint64_t tag = ((uint64_t)4 << 56) ^ ((uint64_t)capacity << 24) ^ (uint64_t)n_rows ^ 0x5A00000000000000ull;
            const size_t slot = (first_slot + k) % t->n_slots;
            bool clean = false;
            j.trace = claims.claim(slot, tag, &clean);
            j.tail_clean = clean;
        }
        j.hist = d_hist + 256 * k;
    }
    LdSynthJob* d_jobs = nullptr;
    ZKW_TRY(ctx->upload("ld_jobs", jobs, &d_jobs));
    const unsigned nj = (unsigned)n_instances;
    const u32 rstride = (u32)LD_REGION_STRIDE(capacity);
    const dim3 g64((rstride + 63) / 64, nj), g256((rstride + 255) / 256, nj);
    { Prof _p(ctx, "k_ld_fill_queue"); ZKW_LAUNCH_D(ctx, (k_ld_fill_queue<0>), "k_ld_fill_queue", g64, 64, 0, d_jobs, capacity, n_rows); }
    ZKW_TRY(launch_check("k_ld_fill_queue<0>"));
    { Prof _p(ctx, "k_ld_fill_queue"); ZKW_LAUNCH_D(ctx, (k_ld_fill_queue<1>), "k_ld_fill_queue", g64, 64, 0, d_jobs, capacity, n_rows); }
    ZKW_TRY(launch_check("k_ld_fill_queue<1>"));
#define LD_LAUNCH_ROW(R) { Prof _p(ctx, "k_ld_fill_row"); ZKW_LAUNCH_D(ctx, (k_ld_fill_row<LD_ROW_##R>), "k_ld_fill_row", g256, 256, 0, d_jobs, capacity, n_rows); } \
    ZKW_TRY(launch_check("k_ld_fill_row<" #R ">"));
    LD_LAUNCH_ROW(X0) LD_LAUNCH_ROW(X1) LD_LAUNCH_ROW(X2) LD_LAUNCH_ROW(X3) LD_LAUNCH_ROW(R) LD_LAUNCH_ROW(Q)
#undef LD_LAUNCH_ROW
    { Prof _p(ctx, "k_ld_fill_tail"); ZKW_LAUNCH(ctx, k_ld_fill_tail, nj * ((LD_G + LD_L + 1) * TAIL_CHUNKS + 1), 256, d_jobs, nj, capacity, n_rows); }
    ZKW_TRY(launch_check("k_ld_fill_tail"));
    return claims.commit_if(ctx->sync_if_host());
}

extern "C" int zkw_log_demux_check_satisfied(zkw_ctx* ctx, const zkw_trace* t, size_t slot, uint32_t capacity,
                                             uint64_t* n_violations, uint64_t* first_bad) {
    if (!ctx || !t || t->ctx->device != ctx->device || slot >= t->n_slots || !n_violations)
        return fail(ZKW_ERR_INVALID, "zkw_log_demux_check_satisfied: bad argument");
    if (LD_MIN_ROWS(capacity) > t->n_rows) return fail(ZKW_ERR_INVALID, "capacity does not fit the trace");
    return check_satisfied<SpecLogDemux>(ctx, t, slot, capacity, n_violations, first_bad);
}

// compact closed-form inputs and public inputs of a precompile witness's instances, computed once and kept with the witness
// ------------------------------------------------------------------------------------------------ StorageSorter synthesis
extern "C" int zkw_storage_sorter_synthesize(zkw_ctx* ctx, const zkw_storage_witness* w, size_t first_instance, size_t n_instances,
                                             zkw_trace* t, size_t first_slot) {
    if (!ctx || !w || !t || w->ctx != ctx || t->ctx->device != ctx->device) return fail(ZKW_ERR_INVALID, "zkw_storage_sorter_synthesize: bad argument");
    if (first_instance + n_instances > w->n_instances) return fail(ZKW_ERR_INVALID, "instance range out of bounds");
    if (n_instances > t->n_slots) return fail(ZKW_ERR_INVALID, "more instances (%zu) than trace slots (%zu)", n_instances, t->n_slots);
    if (t->n_cols < SS_COLS) return fail(ZKW_ERR_INVALID, "trace has %zu columns, the StorageSorter needs %d", t->n_cols, SS_COLS);
    const u32 capacity = w->capacity;
    const size_t n_rows = t->n_rows, n = w->n;
    if (SS_MIN_ROWS(capacity) > n_rows)
        return fail(ZKW_ERR_INVALID, "capacity %u needs %llu rows, trace has %zu", capacity, (unsigned long long)SS_MIN_ROWS(capacity), n_rows);
    if (n_rows & 1) return fail(ZKW_ERR_INVALID, "trace length must be even (it is a power of two in every circuit)");
    if (n_instances == 0) return ZKW_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    u32* d_hist = nullptr;
    ZKW_TRY(ctx->scratch_t<u32>("ss_hist", n_instances * 256, &d_hist));
    HIP_TRY(ctx->memset_async(d_hist, 0, n_instances * 256 * sizeof(u32)));
    SlotClaims claims(t);
    std::vector<SsSynthJob> jobs(n_instances);
    for (size_t k = 0; k < n_instances; k++) {
        SsSynthJob& j = jobs[k];
        j.inst = w->instances + first_instance + k;
        j.unsorted_enc = w->enc_all; j.sorted_enc = w->enc_all + 20 * n;
        j.unsorted_new_tails = w->tails_all + 4 * n; j.sorted_new_tails = w->tails_all + 12 * n; j.result_new_tails = w->tails_all + 16 * n;
        j.challenges = w->challenges;
        j.lhs_z = w->lhs_z; j.rhs_z = w->rhs_z;
        j.sc.D = reinterpret_cast<int*>(w->scans); j.sc.S = w->scans + n; j.sc.R = w->scans + 2 * n; j.sc.E = w->scans + 3 * n;
        j.n_block = n;
        j.public_input = w->cf_pi + COMPACT_FORM_LEN * w->n_instances + 4 * (first_instance + k);
        j.first_inst = w->instances;  // one block per witness
        {   // a slot whose previous tenant was this layout keeps its zero padding rows (zkw_ctx.h slot_tag; every other writer resets the tag)
            const uint64_t tag = ((uint64_t)9 << 56) ^ ((uint64_t)capacity << 24) ^ (uint64_t)n_rows ^ 0x5A00000000000000ull;
            const size_t slot = (first_slot + k) % t->n_slots;
            bool clean = false;
            j.trace = claims.claim(slot, tag, &clean);
            j.tail_clean = clean;
        }
        j.hist = d_hist + 256 * k;
    }
    SsSynthJob* d_jobs = nullptr;
    ZKW_TRY(ctx->upload("ss_jobs", jobs, &d_jobs));
    const unsigned nj = (unsigned)n_instances;
    const u32 rstride = (u32)SS_REGION_STRIDE(capacity);
    const dim3 g64((rstride + 63) / 64, nj), g256((rstride + 255) / 256, nj);
    { Prof _p(ctx, "k_ss_fill_queue"); ZKW_LAUNCH_D(ctx, (k_ss_fill_queue<0>), "k_ss_fill_queue", g64, 64, 0, d_jobs, capacity, n_rows); }
    ZKW_TRY(launch_check("k_ss_fill_queue<0>"));
    { Prof _p(ctx, "k_ss_fill_queue"); ZKW_LAUNCH_D(ctx, (k_ss_fill_queue<1>), "k_ss_fill_queue", g64, 64, 0, d_jobs, capacity, n_rows); }
    ZKW_TRY(launch_check("k_ss_fill_queue<1>"));
    { Prof _p(ctx, "k_ss_fill_queue"); ZKW_LAUNCH_D(ctx, (k_ss_fill_queue<2>), "k_ss_fill_queue", g64, 64, 0, d_jobs, capacity, n_rows); }
    ZKW_TRY(launch_check("k_ss_fill_queue<2>"));
#define SS_LAUNCH_ROW(R) { Prof _p(ctx, "k_ss_fill_row"); ZKW_LAUNCH_D(ctx, (k_ss_fill_row<SS_ROW_##R>), "k_ss_fill_row", g256, 256, 0, d_jobs, capacity, n_rows); } \
    ZKW_TRY(launch_check("k_ss_fill_row<" #R ">"));
    SS_LAUNCH_ROW(A) SS_LAUNCH_ROW(X0) SS_LAUNCH_ROW(X1) SS_LAUNCH_ROW(X2) SS_LAUNCH_ROW(X3) SS_LAUNCH_ROW(X4) SS_LAUNCH_ROW(X5)
    SS_LAUNCH_ROW(X6) SS_LAUNCH_ROW(X7) SS_LAUNCH_ROW(K) SS_LAUNCH_ROW(C1) SS_LAUNCH_ROW(C2) SS_LAUNCH_ROW(Q)
#undef SS_LAUNCH_ROW
    { Prof _p(ctx, "k_ss_fill_tail"); ZKW_LAUNCH(ctx, k_ss_fill_tail, nj * ((SS_G + SS_L + 1) * TAIL_CHUNKS + 1), 256, d_jobs, nj, capacity, n_rows); }
    ZKW_TRY(launch_check("k_ss_fill_tail"));
    return claims.commit_if(ctx->sync_if_host());
}

extern "C" int zkw_storage_sorter_check_satisfied(zkw_ctx* ctx, const zkw_trace* t, size_t slot, uint32_t capacity,
                                                  uint64_t* n_violations, uint64_t* first_bad) {
    if (!ctx || !t || t->ctx->device != ctx->device || slot >= t->n_slots || !n_violations)
        return fail(ZKW_ERR_INVALID, "zkw_storage_sorter_check_satisfied: bad argument");
    if (SS_MIN_ROWS(capacity) > t->n_rows) return fail(ZKW_ERR_INVALID, "capacity does not fit the trace");
    return check_satisfied<SpecStorageSorter>(ctx, t, slot, capacity, n_violations, first_bad);
}

