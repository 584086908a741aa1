// zkw_precompiles.hip — the hash-circuit side of include/zkw.h: CodeDecommitter (3), the keccak256 / sha256 / ecrecover
// round functions (5, 6, 7), StorageApplication (10), L1MessagesHasher (13): witness builders, and the netlist engine that
// synthesizes and checks types 3, 5, 6, 13 ("zkw trace v4").
#include "zkw_ctx.h"
#include "closed_forms_host.h"
#include "decommitter_kernels.cuh"
#include <array>
#include <functional>
#include "precompile_kernels.cuh"
#include "storage_application_kernels.cuh"
#include "netlist_kernels.cuh"
#include "netlist_queue_kernels.cuh"
#include "netlist_closed_form_kernels.cuh"
#include "ecrecover_kernels.cuh"
#include "radix_sort.cuh"

// ------------------------------------------------------------------------------------------------ code decommitter
struct zkw_decommitter_witness {
    zkw_ctx* ctx = nullptr;
    size_t n_requests = 0, total_words = 0, total_rounds = 0, n_instances = 0;
    zkw_mem_query* mem_q = nullptr;
    u64 *mem_enc = nullptr, *mem_tails = nullptr;
    u32* round_states = nullptr;
    zkw_decommitter_instance* instances = nullptr;
    zkw_sha256_round_record* sha256_rounds = nullptr;  // [total_rounds]: the cycles of the circuit
    u32 capacity = 0;
    u64* cf_pi = nullptr;  // compact forms [ni][18] | public inputs [ni][4], made by the first synthesis call
    // the queues of the circuit's queue section (netlist_queue_kernels.cuh): the popped requests and the states of their queue, what
    // every round does to the queues, the memory queue's state before the first write
    zkw_decommit_query* requests = nullptr;
    u64* dedup_tails = nullptr;
    RoundOps* round_ops = nullptr;
    zkw_queue_state12 mem_in{};
    void release() {
        void* ptrs[] = {mem_q, mem_enc, mem_tails, round_states, instances, sha256_rounds, cf_pi, requests, dedup_tails, round_ops};
        for (void* p : ptrs)
            if (p) dev_free(p);
    }
};

extern "C" int zkw_decommitter_memory_queries(zkw_ctx* ctx, const zkw_decommit_query* requests, size_t n_requests,
                                              const uint32_t* words, const uint64_t* word_offsets, zkw_mem_query* out) {
    if (!ctx || !requests || !words || !word_offsets || !out || n_requests == 0)
        return fail(ZKW_ERR_INVALID, "zkw_decommitter_memory_queries: bad argument");
    HIP_TRY(hipSetDevice(ctx->device));
    std::vector<uint64_t> woff(n_requests + 1);
    for (size_t k = 0; k <= n_requests; k++) woff[k] = word_offsets[k] - word_offsets[0];
    const size_t total = woff[n_requests];
    const zkw_decommit_query* d_req = nullptr;
    const u32* d_words = nullptr;
    u64* d_woff = nullptr;
    zkw_mem_query* d_out = nullptr;
    ZKW_TRY(ctx->in("dcm_req", requests, n_requests, &d_req));
    ZKW_TRY(ctx->in("dcm_words", words + 8 * word_offsets[0], total * 8, &d_words));
    ZKW_TRY(ctx->upload("dcm_woff", woff, &d_woff));
    ZKW_TRY(ctx->out("dcm_mq_out", out, total, &d_out));
    DecommitterJob job{d_req, d_words, d_woff, nullptr, nullptr, d_out, nullptr, nullptr, n_requests, nullptr};
    if (total) {
        { Prof _p(ctx, "k_decommitter_mem_queries"); ZKW_LAUNCH(ctx, k_decommitter_mem_queries, blocks_for(total, 256), 256, job, (u64)total); }
        ZKW_TRY(launch_check("k_decommitter_mem_queries"));
    }
    ZKW_TRY(ctx->finish_out(out, d_out, total));
    return ctx->sync_if_host();
}

extern "C" int zkw_decommitter_build_with_tails(zkw_ctx* ctx, const zkw_decommit_query* requests, const uint64_t* dedup_tails,
                                     size_t n_requests, const uint32_t* words, const uint64_t* word_offsets,
                                     uint32_t capacity, const zkw_queue_state12* mem_in, const uint64_t* given_mem_tails,
                                     zkw_decommitter_witness** out) {
    if (!ctx || !requests || !dedup_tails || !words || !word_offsets || !mem_in || !out || capacity == 0 || n_requests == 0)
        return fail(ZKW_ERR_INVALID, "zkw_decommitter_build: bad argument");
    HIP_TRY(hipSetDevice(ctx->device));
    std::vector<uint64_t> woff(n_requests + 1), roff(n_requests + 1, 0);
    for (size_t k = 0; k <= n_requests; k++) woff[k] = word_offsets[k] - word_offsets[0];
    for (size_t k = 0; k < n_requests; k++) {
        if (word_offsets[k + 1] <= word_offsets[k]) return fail(ZKW_ERR_INVALID, "request %zu has no bytecode (decommit_code.rs:236)", k);
        roff[k + 1] = roff[k] + (woff[k + 1] - woff[k] + 1) / 2;
    }
    zkw_decommitter_witness* w = new zkw_decommitter_witness();
    w->ctx = ctx;
    w->n_requests = n_requests;
    w->total_words = woff[n_requests];
    w->total_rounds = roff[n_requests];
    w->n_instances = (w->total_rounds + capacity - 1) / capacity;
    hipError_t e = hipSuccess;
    auto alloc = [&](void** p, size_t bytes) { if (e == hipSuccess) e = dev_malloc(p, bytes + 64); };
    alloc((void**)&w->mem_q, w->total_words * sizeof(zkw_mem_query));
    alloc((void**)&w->mem_enc, w->total_words * 64);
    alloc((void**)&w->mem_tails, w->total_words * 96);
    alloc((void**)&w->round_states, w->total_rounds * 32);
    alloc((void**)&w->sha256_rounds, w->total_rounds * sizeof(zkw_sha256_round_record));
    alloc((void**)&w->requests, n_requests * sizeof(zkw_decommit_query));
    alloc((void**)&w->dedup_tails, n_requests * 96);
    alloc((void**)&w->round_ops, w->total_rounds * sizeof(RoundOps));
    w->mem_in = *mem_in;
    w->capacity = capacity;
    alloc((void**)&w->instances, w->n_instances * sizeof(zkw_decommitter_instance));
    auto bail = [&](int rc) { w->release(); delete w; return rc; };
    if (e != hipSuccess) return bail(fail(ZKW_ERR_OOM, "zkw_decommitter_build: hipMalloc failed: %s", hipGetErrorString(e)));
    const zkw_decommit_query* d_req = nullptr;
    const u64* d_dt = nullptr;
    const u32* d_words = nullptr;
    u64 *d_woff = nullptr, *d_roff = nullptr;
    u32* d_viol = nullptr;
    int rc = ctx->in("dcm_req", requests, n_requests, &d_req);
    if (rc == ZKW_OK) rc = ctx->in("dcm_dt", dedup_tails, n_requests * 12, &d_dt);
    if (rc == ZKW_OK) rc = ctx->in("dcm_words", words + 8 * word_offsets[0], w->total_words * 8, &d_words);
    if (rc == ZKW_OK) rc = ctx->upload("dcm_woff", woff, &d_woff);
    if (rc == ZKW_OK) rc = ctx->upload("dcm_roff", roff, &d_roff);
    if (rc == ZKW_OK) rc = ctx->scratch_t<u32>("dcm_viol", 1, &d_viol);
    if (rc != ZKW_OK) return bail(rc);
    if (ctx->memset_async(d_viol, 0, 4) != hipSuccess) return bail(fail(ZKW_ERR_HIP, "memset failed"));
    if (ctx->copy_async(w->requests, d_req, n_requests * sizeof(zkw_decommit_query), hipMemcpyDeviceToDevice) != hipSuccess ||
        ctx->copy_async(w->dedup_tails, d_dt, n_requests * 96, hipMemcpyDeviceToDevice) != hipSuccess)
        return bail(fail(ZKW_ERR_HIP, "copy of the decommit requests failed"));
    DecommitterJob job{d_req, d_words, d_woff, d_roff, w->round_states, w->mem_q, w->mem_enc, d_viol, n_requests, w->sha256_rounds, w->round_ops};
    { Prof _p(ctx, "k_decommitter_sha"); ZKW_LAUNCH(ctx, k_decommitter_sha, blocks_for(n_requests, 64), 64, job); }
    if ((rc = launch_check("k_decommitter_sha")) != ZKW_OK) return bail(rc);
    { Prof _p(ctx, "k_decommitter_mem_queries"); ZKW_LAUNCH(ctx, k_decommitter_mem_queries, blocks_for(w->total_words, 256), 256, job, (u64)w->total_words); }
    if ((rc = launch_check("k_decommitter_mem_queries")) != ZKW_OK) return bail(rc);
    zkw_queue_state12* d_min = nullptr;
    std::vector<zkw_queue_state12> minv(1, *mem_in);
    if ((rc = ctx->upload("dcm_mem_in", minv, &d_min)) != ZKW_OK) return bail(rc);
    if (given_mem_tails) {  // the caller has already hashed the memory queue this slice belongs to (zkw_block_run)
        if (ctx->copy_async(w->mem_tails, given_mem_tails, w->total_words * 96, ctx->ptr_mode == ZKW_PTR_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice) != hipSuccess)
            return bail(fail(ZKW_ERR_HIP, "copy of the given memory-queue states failed"));
    } else {
        std::vector<ChainJob> chains(1, ChainJob{w->mem_enc, w->mem_tails, d_min->tail, w->total_words});
        if ((rc = dev_chains(ctx, chains)) != ZKW_OK) return bail(rc);
    }
    std::vector<DecommitterBlock> blk(1);
    blk[0].job = job;
    blk[0].dedup_tails = d_dt;
    blk[0].mem_tails = w->mem_tails;
    blk[0].instances = w->instances;
    blk[0].mem_in = *mem_in;
    blk[0].total_rounds = w->total_rounds;
    blk[0].total_words = w->total_words;
    blk[0].capacity = capacity;
    DecommitterBlock* d_blk = nullptr;
    if ((rc = ctx->upload("dcm_block", blk, &d_blk)) != ZKW_OK) return bail(rc);
    { Prof _p(ctx, "k_decommitter_instances"); ZKW_LAUNCH(ctx, k_decommitter_instances, blocks_for(w->n_instances, 64), 64, d_blk); }
    if ((rc = launch_check("k_decommitter_instances")) != ZKW_OK) return bail(rc);
    u32 viol = 0;
    if (ctx->read_small(&viol, d_viol, 4) != ZKW_OK)
        return bail(fail(ZKW_ERR_HIP, "readback failed"));
    if (viol) return bail(fail(ZKW_ERR_CHECK_FAILED, "%u bytecodes do not match their decommit request (length parity, word count or "
                                                     "SHA-256 digest, decommit_code.rs:241-244, 323-337)", viol));
    ctx_retain(ctx);
    *out = w;
    return ZKW_OK;
}

extern "C" int zkw_decommitter_build(zkw_ctx* ctx, const zkw_decommit_query* requests, const uint64_t* dedup_tails,
                                     size_t n_requests, const uint32_t* words, const uint64_t* word_offsets,
                                     uint32_t capacity, const zkw_queue_state12* mem_in, zkw_decommitter_witness** out) {
    return zkw_decommitter_build_with_tails(ctx, requests, dedup_tails, n_requests, words, word_offsets, capacity, mem_in, nullptr, out);
}

extern "C" size_t zkw_decommitter_witness_num_instances(const zkw_decommitter_witness* w) { return w ? w->n_instances : 0; }
static const void* dcm_array(const zkw_decommitter_witness* w, int what, size_t* bytes) {
    switch (what) {
        case ZKW_DCM_MEM_QUERIES: *bytes = w->total_words * sizeof(zkw_mem_query); return w->mem_q;
        case ZKW_DCM_MEM_ENC: *bytes = w->total_words * 64; return w->mem_enc;
        case ZKW_DCM_MEM_TAILS: *bytes = w->total_words * 96; return w->mem_tails;
        case ZKW_DCM_ROUND_STATES: *bytes = w->total_rounds * 32; return w->round_states;
        case ZKW_DCM_INSTANCES: *bytes = w->n_instances * sizeof(zkw_decommitter_instance); return w->instances;
        case ZKW_DCM_SHA256_ROUNDS: *bytes = w->total_rounds * sizeof(zkw_sha256_round_record); return w->sha256_rounds;
        default: *bytes = 0; return nullptr;
    }
}
extern "C" size_t zkw_decommitter_witness_bytes(const zkw_decommitter_witness* w, int what) {
    size_t b = 0;
    if (w) (void)dcm_array(w, what, &b);
    return b;
}
extern "C" const void* zkw_decommitter_witness_device_ptr(const zkw_decommitter_witness* w, int what) {
    size_t b = 0;
    return w ? dcm_array(w, what, &b) : nullptr;
}
extern "C" int zkw_decommitter_witness_get(const zkw_decommitter_witness* w, int what, void* dst, size_t dst_bytes) {
    if (!w || !dst) return fail(ZKW_ERR_INVALID, "zkw_decommitter_witness_get: null argument");
    if (what < 0 || what > ZKW_DCM_SHA256_ROUNDS) return fail(ZKW_ERR_INVALID, "unknown array %d", what);
    size_t bytes = 0;
    const void* src = dcm_array(w, what, &bytes);
    if (dst_bytes < bytes) return fail(ZKW_ERR_INVALID, "need %zu bytes, got %zu", bytes, dst_bytes);
    if (bytes == 0) return ZKW_OK;
    zkw_ctx* ctx = w->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(ctx->copy_async(dst, src, bytes, ctx->ptr_mode == ZKW_PTR_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost));
    return ctx->sync_if_host();
}
extern "C" void zkw_decommitter_witness_free(zkw_decommitter_witness* w) {
    if (!w) return;
    (void)hipSetDevice(w->ctx->device);
    (void)w->ctx->sync_stream();
    w->release();
    zkw_ctx* owner = w->ctx;
    delete w;
    ctx_release(owner);
}

// ------------------------------------------------------------------------------------------------ L1 messages hasher
extern "C" int zkw_linear_keccak256(zkw_ctx* ctx, const zkw_log_query* messages, size_t n, uint8_t* hash_out) {
    if (!ctx || !hash_out || (n && !messages)) return fail(ZKW_ERR_INVALID, "zkw_linear_keccak256: null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    const zkw_log_query* d_q = nullptr;
    uint8_t* d_out = nullptr;
    ZKW_TRY(ctx->in("lk_q", messages, n, &d_q));
    ZKW_TRY(ctx->out("lk_out", hash_out, 32, &d_out));
    { Prof _p(ctx, "k_linear_keccak256"); ZKW_LAUNCH(ctx, k_linear_keccak256, 1, 64, d_q, n, d_out, (zkw_keccak_round_record*)nullptr, (const u64*)nullptr, (const u64*)nullptr, false); }
    ZKW_TRY(launch_check("k_linear_keccak256"));
    ZKW_TRY(ctx->finish_out(hash_out, d_out, 32));
    return ctx->sync_if_host();
}

// RecursionQueueSimulator::split_by(RECURSION_ARITY) as create_leaf_witnesses uses it (src/witness/recursive_aggregation.rs:
// 98-117, circuit_encodings/src/lib.rs:472-506): leaf k covers the requests [k * arity, min((k + 1) * arity, n)); its queue
// starts at the state the previous leaf ended with (head = tail before its first request), ends at the state after its last
// request. Pure host arithmetic over the states zkw_queue_push_chain_full returned: no device work.
extern "C" int zkw_recursion_queue_split(const uint64_t* states, size_t n, uint32_t arity, zkw_queue_state12* leaf_states,
                                         size_t max_leaves, size_t* n_leaves) {
    if (!n_leaves || arity == 0 || (n && !states)) return fail(ZKW_ERR_INVALID, "zkw_recursion_queue_split: bad argument");
    const size_t leaves = (n + arity - 1) / arity;  // an empty queue has no leaves (split_by returns an empty vector)
    *n_leaves = leaves;
    if (leaves > max_leaves || (leaves && !leaf_states)) return fail(ZKW_ERR_INVALID, "zkw_recursion_queue_split: %zu leaves, room for %zu", leaves, max_leaves);
    for (size_t k = 0; k < leaves; k++) {
        const size_t first = k * arity, end = std::min(n, first + arity);
        zkw_queue_state12& q = leaf_states[k];
        memset(&q, 0, sizeof q);
        if (first) memcpy(q.head, states + 12 * (first - 1), 96);
        memcpy(q.tail, states + 12 * (end - 1), 96);
        q.length = (uint32_t)(end - first);
    }
    return ZKW_OK;
}

// ------------------------------------------------------------------------------------------------ precompile round functions (a16)
struct zkw_precompile_witness {
    zkw_ctx* ctx = nullptr;
    size_t n_requests = 0, n_queries = 0, total_rounds = 0, total_reads = 0, n_instances = 0;
    u64 *mem_enc = nullptr, *mem_tails = nullptr;
    zkw_precompile_instance* instances = nullptr;
    zkw_keccak_round_record* keccak_rounds = nullptr;  // keccak256 only: [total_rounds], the cycles of the circuit
    zkw_sha256_round_record* sha256_rounds = nullptr;  // sha256 only
    int kind = 0;
    u32 capacity = 0;
    u64* cf_pi = nullptr;  // compact forms [ni][18] | public inputs [ni][4], made by the first synthesis call
    // keccak256 / sha256 — the queues of the circuit's queue section (netlist_queue_kernels.cuh): the precompile calls and the states of their
    // queue, the memory queries, what every round does to the queues, the memory queue's state before the first query
    zkw_log_query* requests = nullptr;
    u64* req_tails = nullptr;
    zkw_mem_query* mem_q = nullptr;
    RoundOps* round_ops = nullptr;
    zkw_queue_state12 mem_in{};
    void release() {
        void* ptrs[] = {mem_enc, mem_tails, instances, keccak_rounds, sha256_rounds, cf_pi, requests, req_tails, mem_q, round_ops};
        for (void* p : ptrs)
            if (p) dev_free(p);
    }
};

extern "C" int zkw_precompile_build_with_tails(zkw_ctx* ctx, int kind, const zkw_log_query* requests, const uint64_t* request_tails,
                                    size_t n_requests, const zkw_mem_query* mem_queries, size_t n_queries, uint32_t capacity,
                                    const zkw_queue_state12* mem_in, const uint64_t* given_mem_tails,
                                    zkw_precompile_witness** out) {
    if (!ctx || !mem_in || !out || capacity == 0 || kind < ZKW_PRECOMPILE_KECCAK256 || kind > ZKW_PRECOMPILE_ECRECOVER ||
        (n_requests && (!requests || !request_tails)) || (n_queries && !mem_queries))
        return fail(ZKW_ERR_INVALID, "zkw_precompile_build: bad argument");
    if (n_requests == 0 && n_queries) return fail(ZKW_ERR_INVALID, "memory queries without a precompile request");
    HIP_TRY(hipSetDevice(ctx->device));
    const zkw_log_query* d_req = nullptr;
    const u64* d_rt = nullptr;
    const zkw_mem_query* d_mq = nullptr;
    u64 *d_roff = nullptr, *d_qoff = nullptr, *d_rdoff = nullptr, *d_meta = nullptr;
    u64 meta[4] = {0, 0, 0, 0};
    if (n_requests) {
        ZKW_TRY(ctx->in("pc_req", requests, n_requests, &d_req));
        ZKW_TRY(ctx->in("pc_rt", request_tails, n_requests * 4, &d_rt));
        if (n_queries) ZKW_TRY(ctx->in("pc_mq", mem_queries, n_queries, &d_mq));
        u64* d_off = nullptr;  // [3][n_requests + 1]: exclusive prefix sums of rounds, queries, reads (totals at [n_requests])
        ZKW_TRY(ctx->scratch_t<u64>("pc_off", 3 * (n_requests + 1), &d_off));
        d_roff = d_off; d_qoff = d_off + (n_requests + 1); d_rdoff = d_off + 2 * (n_requests + 1);
        ZKW_TRY(ctx->scratch_t<u64>("pc_meta", 4, &d_meta));
        HIP_TRY(ctx->memset_async(d_meta, 0, 4 * sizeof(u64)));
        ZKW_TRY((sum_prefix<3>(ctx, "k_precompile_counts", PrecompileShape{kind, d_req, reinterpret_cast<u32*>(d_meta + 3)}, n_requests, d_off, d_meta)));
        ZKW_TRY(ctx->read_small(meta, d_meta, sizeof meta));
        if (meta[3]) return fail(ZKW_ERR_INVALID, "a precompile request without rounds (the first round carries `new_request`)");
        if (meta[1] != n_queries)
            return fail(ZKW_ERR_INVALID, "the requests need %llu memory queries, %zu given", (unsigned long long)meta[1], n_queries);
    }
    zkw_precompile_witness* w = new zkw_precompile_witness();
    w->ctx = ctx;
    w->n_requests = n_requests;
    w->n_queries = n_queries;
    w->total_rounds = meta[0];
    w->total_reads = meta[2];
    w->n_instances = n_requests ? (w->total_rounds + capacity - 1) / capacity : 1;
    w->kind = kind;
    w->capacity = capacity;
    hipError_t e = hipSuccess;
    auto alloc = [&](void** p, size_t bytes) { if (e == hipSuccess) e = dev_malloc(p, bytes + 64); };
    alloc((void**)&w->mem_enc, n_queries * 64);
    alloc((void**)&w->mem_tails, n_queries * 96);
    alloc((void**)&w->instances, w->n_instances * sizeof(zkw_precompile_instance));
    if (kind == ZKW_PRECOMPILE_KECCAK256) alloc((void**)&w->keccak_rounds, w->total_rounds * sizeof(zkw_keccak_round_record));
    if (kind == ZKW_PRECOMPILE_SHA256) alloc((void**)&w->sha256_rounds, w->total_rounds * sizeof(zkw_sha256_round_record));
    alloc((void**)&w->requests, n_requests * sizeof(zkw_log_query));  // (the queue sections of the three circuits)
    alloc((void**)&w->req_tails, n_requests * 32);
    alloc((void**)&w->mem_q, n_queries * sizeof(zkw_mem_query));
    alloc((void**)&w->round_ops, w->total_rounds * sizeof(RoundOps));
    w->mem_in = *mem_in;
    auto bail = [&](int rc) { w->release(); delete w; return rc; };
    if (e != hipSuccess) return bail(fail(ZKW_ERR_OOM, "zkw_precompile_build: hipMalloc failed: %s", hipGetErrorString(e)));
    int rc = ZKW_OK;
    PrecompileSnap* d_snaps = nullptr;
    u32* d_viol = nullptr;
    if ((rc = ctx->scratch_t<u32>("pc_viol", 1, &d_viol)) != ZKW_OK) return bail(rc);
    if (ctx->memset_async(d_viol, 0, 4) != hipSuccess) return bail(fail(ZKW_ERR_HIP, "memset failed"));
    if (n_requests) {
        if ((rc = ctx->scratch_t<PrecompileSnap>("pc_snaps", w->n_instances, &d_snaps)) != ZKW_OK) return bail(rc);
        if (n_queries) {
            if ((rc = dev_encode(ctx, d_mq, n_queries, w->mem_enc)) != ZKW_OK) return bail(rc);
            zkw_queue_state12* d_min = nullptr;
            std::vector<zkw_queue_state12> minv(1, *mem_in);
            if ((rc = ctx->upload("pc_mem_in", minv, &d_min)) != ZKW_OK) return bail(rc);
            if (given_mem_tails) {  // already hashed by the caller as part of the whole memory queue (zkw_block_run)
                if (ctx->copy_async(w->mem_tails, given_mem_tails, n_queries * 96, ctx->ptr_mode == ZKW_PTR_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice) != hipSuccess)
                    return bail(fail(ZKW_ERR_HIP, "copy of the given memory-queue states failed"));
            } else {
                std::vector<ChainJob> chains(1, ChainJob{w->mem_enc, w->mem_tails, d_min->tail, n_queries});
                if ((rc = dev_chains(ctx, chains)) != ZKW_OK) return bail(rc);
            }
        }
        if ((ctx->copy_async(w->requests, d_req, n_requests * sizeof(zkw_log_query), hipMemcpyDeviceToDevice) != hipSuccess ||
             ctx->copy_async(w->req_tails, d_rt, n_requests * 32, hipMemcpyDeviceToDevice) != hipSuccess ||
             (n_queries && ctx->copy_async(w->mem_q, d_mq, n_queries * sizeof(zkw_mem_query), hipMemcpyDeviceToDevice) != hipSuccess)))
            return bail(fail(ZKW_ERR_HIP, "copy of the precompile calls failed"));
        PrecompileJob job{kind, d_req, d_mq, d_roff, d_qoff, d_rdoff, d_snaps, d_viol, n_requests, w->total_rounds, capacity, w->keccak_rounds, w->sha256_rounds, w->round_ops};
        { Prof _p(ctx, "k_precompile_walk"); ZKW_LAUNCH(ctx, k_precompile_walk, blocks_for(n_requests, 64), 64, job); }
        if ((rc = launch_check("k_precompile_walk")) != ZKW_OK) return bail(rc);
    }
    std::vector<PrecompileBlock> blk(1);
    blk[0].kind = kind;
    blk[0].snaps = d_snaps;
    blk[0].req_tails = d_rt;
    blk[0].mem_tails = w->mem_tails;
    blk[0].instances = w->instances;
    blk[0].mem_in = *mem_in;
    blk[0].n_requests = n_requests;
    blk[0].total_rounds = w->total_rounds;
    blk[0].n_instances = w->n_instances;
    blk[0].capacity = capacity;
    PrecompileBlock* d_blk = nullptr;
    if ((rc = ctx->upload("pc_block", blk, &d_blk)) != ZKW_OK) return bail(rc);
    { Prof _p(ctx, "k_precompile_instances"); ZKW_LAUNCH(ctx, k_precompile_instances, blocks_for(w->n_instances, 64), 64, d_blk); }
    if ((rc = launch_check("k_precompile_instances")) != ZKW_OK) return bail(rc);
    u32 viol = 0;
    if (ctx->read_small(&viol, d_viol, 4) != ZKW_OK)
        return bail(fail(ZKW_ERR_HIP, "readback failed"));
    if (viol) return bail(fail(ZKW_ERR_CHECK_FAILED, "%u requests whose memory queries do not fit their ABI (read/write flags, word "
                                                     "index or count: the asserts of the round walks)", viol));
    ctx_retain(ctx);
    *out = w;
    return ZKW_OK;
}

extern "C" int zkw_precompile_build(zkw_ctx* ctx, int kind, const zkw_log_query* requests, const uint64_t* request_tails,
                                    size_t n_requests, const zkw_mem_query* mem_queries, size_t n_queries, uint32_t capacity,
                                    const zkw_queue_state12* mem_in, zkw_precompile_witness** out) {
    return zkw_precompile_build_with_tails(ctx, kind, requests, request_tails, n_requests, mem_queries, n_queries, capacity, mem_in, nullptr, out);
}

extern "C" size_t zkw_precompile_witness_num_instances(const zkw_precompile_witness* w) { return w ? w->n_instances : 0; }
extern "C" size_t zkw_precompile_witness_num_rounds(const zkw_precompile_witness* w) { return w ? w->total_rounds : 0; }
static const void* pc_array(const zkw_precompile_witness* w, int what, size_t* bytes) {
    switch (what) {
        case ZKW_PRC_MEM_ENC: *bytes = w->n_queries * 64; return w->mem_enc;
        case ZKW_PRC_MEM_TAILS: *bytes = w->n_queries * 96; return w->mem_tails;
        case ZKW_PRC_INSTANCES: *bytes = w->n_instances * sizeof(zkw_precompile_instance); return w->instances;
        case ZKW_PRC_KECCAK_ROUNDS: *bytes = w->keccak_rounds ? w->total_rounds * sizeof(zkw_keccak_round_record) : 0; return w->keccak_rounds;
        case ZKW_PRC_SHA256_ROUNDS: *bytes = w->sha256_rounds ? w->total_rounds * sizeof(zkw_sha256_round_record) : 0; return w->sha256_rounds;
        default: *bytes = 0; return nullptr;
    }
}
extern "C" size_t zkw_precompile_witness_bytes(const zkw_precompile_witness* w, int what) {
    size_t b = 0;
    if (w) (void)pc_array(w, what, &b);
    return b;
}
extern "C" const void* zkw_precompile_witness_device_ptr(const zkw_precompile_witness* w, int what) {
    size_t b = 0;
    return w ? pc_array(w, what, &b) : nullptr;
}
extern "C" int zkw_precompile_witness_get(const zkw_precompile_witness* w, int what, void* dst, size_t dst_bytes) {
    if (!w || !dst) return fail(ZKW_ERR_INVALID, "zkw_precompile_witness_get: null argument");
    if (what < 0 || what > ZKW_PRC_SHA256_ROUNDS) return fail(ZKW_ERR_INVALID, "unknown array %d", what);
    size_t bytes = 0;
    const void* src = pc_array(w, what, &bytes);
    if (dst_bytes < bytes) return fail(ZKW_ERR_INVALID, "need %zu bytes, got %zu", bytes, dst_bytes);
    if (bytes == 0) return ZKW_OK;
    zkw_ctx* ctx = w->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(ctx->copy_async(dst, src, bytes, ctx->ptr_mode == ZKW_PTR_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost));
    return ctx->sync_if_host();
}
extern "C" void zkw_precompile_witness_free(zkw_precompile_witness* w) {
    if (!w) return;
    (void)hipSetDevice(w->ctx->device);
    (void)w->ctx->sync_stream();
    w->release();
    zkw_ctx* owner = w->ctx;
    delete w;
    ctx_release(owner);
}

// ------------------------------------------------------------------------------------------------ storage application (a17)
struct zkw_storage_application_witness {
    zkw_ctx* ctx = nullptr;
    size_t n = 0, n_instances = 0;
    u32 *keys = nullptr, *paths = nullptr, *roots = nullptr;
    u64* leaf_indexes = nullptr;
    zkw_storage_application_instance* instances = nullptr;
    SapItem* items = nullptr;  // the leaf before / after every query: what synthesis needs once the builder's scratch is gone
    u32* walk_hashes = nullptr;  // [n][2][257][8]: the running hashes of every query's walks (level 0 = the leaf hash), for synthesis
    u64* cf_pi = nullptr;      // compact forms [n_instances][18] then public inputs [n_instances][4] (a20), on first synthesis
    u32 capacity = 0;
    void release() {
        void* ptrs[] = {keys, paths, roots, leaf_indexes, instances, items, walk_hashes, cf_pi};
        for (void* p : ptrs)
            if (p) dev_free(p);
    }
};

extern "C" int zkw_storage_application_build(zkw_ctx* ctx, const zkw_log_query* queries, const uint64_t* query_tails, size_t n,
                                             const uint64_t* init_leaf_indexes, const uint8_t* init_merkle_paths,
                                             const uint8_t initial_root[32], uint64_t initial_next_enumeration_index,
                                             uint32_t capacity, zkw_storage_application_witness** out) {
    if (!ctx || !out || !initial_root || capacity < 2 || (n && (!queries || !query_tails || !init_leaf_indexes || !init_merkle_paths)))
        return fail(ZKW_ERR_INVALID, "zkw_storage_application_build: bad argument");
    if (n >= (1ull << 31)) return fail(ZKW_ERR_INVALID, "too many storage queries");
    HIP_TRY(hipSetDevice(ctx->device));
    zkw_storage_application_witness* w = new zkw_storage_application_witness();
    w->ctx = ctx;
    w->n = n;
    hipError_t e = hipSuccess;
    auto alloc = [&](void** p, size_t bytes) { if (e == hipSuccess) e = dev_malloc(p, bytes + 64); };
    alloc((void**)&w->keys, n * 32);
    alloc((void**)&w->paths, n * 256 * 32);
    alloc((void**)&w->roots, n * 32);
    alloc((void**)&w->leaf_indexes, n * 8);
    alloc((void**)&w->items, n * sizeof(SapItem));
    alloc((void**)&w->walk_hashes, n * 2 * 257 * 32);
    w->capacity = capacity;
    auto bail = [&](int rc) { w->release(); delete w; return rc; };
    if (e != hipSuccess) return bail(fail(ZKW_ERR_OOM, "zkw_storage_application_build: hipMalloc failed: %s", hipGetErrorString(e)));
    int rc = ZKW_OK;
    SapJob job;
    memset(&job, 0, sizeof job);
    const u64* d_qt = nullptr;
    const uint8_t* d_ip = nullptr;
    u64 *d_snap = nullptr, *d_meta = nullptr;
    uint8_t* d_hash = nullptr;
    u32* d_viol = nullptr;
    auto TRY = [&](int r) { if (rc == ZKW_OK) rc = r; };
    if (n) {
        TRY(ctx->in("sap_q", queries, n, &job.queries));
        TRY(ctx->in("sap_qt", query_tails, n * 4, &d_qt));
        TRY(ctx->in("sap_ii", init_leaf_indexes, n, &job.init_index));
        TRY(ctx->in("sap_ip", init_merkle_paths, n * 256 * 32, &d_ip));
    }
    job.init_paths = reinterpret_cast<const u32*>(d_ip);
    job.keys = w->keys; job.paths = w->paths; job.roots = w->roots; job.walk_hashes = w->walk_hashes;
    TRY(ctx->scratch_t<u64>("sap_newidx", n + 1, &job.new_index));
    TRY(ctx->scratch_t<u32>("sap_prevw", n + 1, &job.prev_write));
    TRY(ctx->scratch_t<u32>("sap_chunk", n + 1, &job.chunk_of));
    TRY(ctx->scratch_t<u32>("sap_fwu", n + 1, &job.first_writes_upto));
    TRY(ctx->scratch_t<u64>("sap_cend", n + 2, &job.chunk_end));
    TRY(ctx->scratch_t<u32>("sap_jstar", (n + 1) * 256, &job.jstar));
    TRY(ctx->scratch_t<u32>("sap_A0", (n + 1) * 8, &job.A0));
    TRY(ctx->scratch_t<u32>("sap_A1", (n + 1) * 8, &job.A1));
    TRY(ctx->scratch_t<u32>("sap_C0", (n + 1) * 8, &job.C0));
    TRY(ctx->scratch_t<u32>("sap_C1", (n + 1) * 8, &job.C1));
    TRY(ctx->scratch_t<u32>("sap_R0", (n + 1) * 8, &job.R0));
    TRY(ctx->scratch_t<u32>("sap_R1", (n + 1) * 8, &job.R1));
    TRY(ctx->scratch_t<u32>("sap_viol", 1, &d_viol));
    TRY(ctx->scratch_t<u64>("sap_meta", 2, &d_meta));
    TRY(ctx->scratch_t<u64>("sap_snap", (n + 1) * 25, &d_snap));
    TRY(ctx->scratch_t<uint8_t>("sap_hash", 32, &d_hash));
    if (rc != ZKW_OK) return bail(rc);
    job.violations = d_viol;
    job.meta = d_meta;
    job.n = n;
    job.next_enumeration_index = initial_next_enumeration_index;
    memcpy(job.initial_root, initial_root, 32);
    job.capacity = capacity;
    if (ctx->memset_async(d_viol, 0, 4) != hipSuccess) return bail(fail(ZKW_ERR_HIP, "memset failed"));
    u64 meta[2] = {1, initial_next_enumeration_index};
    if (n) {
        const unsigned g64 = blocks_for(n, 64);
        { Prof _p(ctx, "k_sap_keys"); ZKW_LAUNCH(ctx, k_sap_keys, g64, 64, job); }
        TRY(launch_check("k_sap_keys"));
        { Prof _p(ctx, "k_sap_scan"); ZKW_LAUNCH(ctx, k_sap_scan, 1, 1024, job); }
        TRY(launch_check("k_sap_scan"));
        { Prof _p(ctx, "k_sap_items"); ZKW_LAUNCH(ctx, k_sap_items, g64, 64, job, w->items); }
        TRY(launch_check("k_sap_items"));
        { Prof _p(ctx, "k_sap_pairs"); ZKW_LAUNCH(ctx, k_sap_pairs, g64, 64, job); }
        TRY(launch_check("k_sap_pairs"));
        { Prof _p(ctx, "k_sap_leaves"); ZKW_LAUNCH(ctx, k_sap_leaves, g64, 64, job); }
        TRY(launch_check("k_sap_leaves"));
        static_assert(ZKW_STORAGE_TREE_DEPTH == 256, "k_sap_levels walks 256 levels");
        if (n <= SAP_PERSISTENT_MAX) {
            Prof _p(ctx, "k_sap_levels");
            ZKW_LAUNCH(ctx, k_sap_levels, 1, SAP_PERSISTENT_THREADS, job);
        } else {
            for (int L = 0; L < ZKW_STORAGE_TREE_DEPTH && rc == ZKW_OK; L++) {
                Prof _p(ctx, "k_sap_level");
                ZKW_LAUNCH(ctx, k_sap_level, g64, 64, job, L);
            }
        }
        TRY(launch_check("k_sap_level"));
        { Prof _p(ctx, "k_sap_roots"); ZKW_LAUNCH(ctx, k_sap_roots, g64, 64, job); }
        TRY(launch_check("k_sap_roots"));
        if (rc != ZKW_OK) return bail(rc);
        if (ctx->copy_async(w->leaf_indexes, job.init_index, n * 8, hipMemcpyDeviceToDevice) != hipSuccess ||
            ctx->read_small(meta, d_meta, sizeof meta) != ZKW_OK)
            return bail(fail(ZKW_ERR_HIP, "readback failed"));
    }
    w->n_instances = n ? (size_t)meta[0] : 1;
    if (dev_malloc((void**)&w->instances, w->n_instances * sizeof(zkw_storage_application_instance) + 64) != hipSuccess)
        return bail(fail(ZKW_ERR_OOM, "zkw_storage_application_build: hipMalloc failed"));
    SapKeccakOut ko{d_snap, d_hash};
    { Prof _p(ctx, "k_sap_keccak"); ZKW_LAUNCH(ctx, k_sap_keccak, 1, 64, job, ko); }
    TRY(launch_check("k_sap_keccak"));
    std::vector<SapBlock> blk(1);
    blk[0].job = job;
    blk[0].query_tails = d_qt;
    blk[0].snapshots = d_snap;
    blk[0].final_hash = d_hash;
    blk[0].instances = w->instances;
    blk[0].n_instances = w->n_instances;
    SapBlock* d_blk = nullptr;
    TRY(ctx->upload("sap_block", blk, &d_blk));
    if (rc != ZKW_OK) return bail(rc);
    { Prof _p(ctx, "k_sap_instances"); ZKW_LAUNCH(ctx, k_sap_instances, blocks_for(w->n_instances, 64), 64, d_blk); }
    TRY(launch_check("k_sap_instances"));
    if (rc != ZKW_OK) return bail(rc);
    u32 viol = 0;
    if (ctx->read_small(&viol, d_viol, 4) != ZKW_OK)
        return bail(fail(ZKW_ERR_HIP, "readback failed"));
    if (viol) return bail(fail(ZKW_ERR_CHECK_FAILED, "%u storage queries contradict the tree: the pre-state proof does not lead to the "
                                                     "initial root, the read value is not the leaf's (storage_application.rs:221,276), "
                                                     "or a slot occurs twice", viol));
    ctx_retain(ctx);
    *out = w;
    return ZKW_OK;
}

extern "C" size_t zkw_storage_application_witness_num_instances(const zkw_storage_application_witness* w) { return w ? w->n_instances : 0; }
static const void* sap_array(const zkw_storage_application_witness* w, int what, size_t* bytes) {
    switch (what) {
        case ZKW_SAP_DERIVED_KEYS: *bytes = w->n * 32; return w->keys;
        case ZKW_SAP_MERKLE_PATHS: *bytes = w->n * 256 * 32; return w->paths;
        case ZKW_SAP_LEAF_INDEXES: *bytes = w->n * 8; return w->leaf_indexes;
        case ZKW_SAP_ROOTS: *bytes = w->n * 32; return w->roots;
        case ZKW_SAP_INSTANCES: *bytes = w->n_instances * sizeof(zkw_storage_application_instance); return w->instances;
        default: *bytes = 0; return nullptr;
    }
}
extern "C" size_t zkw_storage_application_witness_bytes(const zkw_storage_application_witness* w, int what) {
    size_t b = 0;
    if (w) (void)sap_array(w, what, &b);
    return b;
}
extern "C" const void* zkw_storage_application_witness_device_ptr(const zkw_storage_application_witness* w, int what) {
    size_t b = 0;
    return w ? sap_array(w, what, &b) : nullptr;
}
extern "C" int zkw_storage_application_witness_get(const zkw_storage_application_witness* w, int what, void* dst, size_t dst_bytes) {
    if (!w || !dst) return fail(ZKW_ERR_INVALID, "zkw_storage_application_witness_get: null argument");
    if (what < 0 || what > ZKW_SAP_INSTANCES) return fail(ZKW_ERR_INVALID, "unknown array %d", what);
    size_t bytes = 0;
    const void* src = sap_array(w, what, &bytes);
    if (dst_bytes < bytes) return fail(ZKW_ERR_INVALID, "need %zu bytes, got %zu", bytes, dst_bytes);
    if (bytes == 0) return ZKW_OK;
    zkw_ctx* ctx = w->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(ctx->copy_async(dst, src, bytes, ctx->ptr_mode == ZKW_PTR_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost));
    return ctx->sync_if_host();
}
extern "C" void zkw_storage_application_witness_free(zkw_storage_application_witness* w) {
    if (!w) return;
    (void)hipSetDevice(w->ctx->device);
    (void)w->ctx->sync_stream();
    w->release();
    zkw_ctx* owner = w->ctx;
    delete w;
    ctx_release(owner);
}

// (work_ctx: whose stream and scratch do the work — the witness's own context, or the private one of a joint call over many witnesses)
static int precompile_closed_forms_with(zkw_ctx* ctx, zkw_precompile_witness* w, const uint64_t** compact, const uint64_t** public_inputs);
extern "C" int zkw_precompile_closed_forms(zkw_ctx* ctx, zkw_precompile_witness* w, const uint64_t** compact, const uint64_t** public_inputs) {
    if (!ctx || !w || w->ctx != ctx) return fail(ZKW_ERR_INVALID, "zkw_precompile_closed_forms: bad argument");
    return precompile_closed_forms_with(ctx, w, compact, public_inputs);
}
static int precompile_closed_forms_with(zkw_ctx* ctx, zkw_precompile_witness* w, const uint64_t** compact, const uint64_t** public_inputs) {
    HIP_TRY(hipSetDevice(ctx->device));
    if (!w->cf_pi) {
        if (w->kind == ZKW_PRECOMPILE_KECCAK256) ZKW_TRY(closed_form_public_inputs<CfPrecompile<ZKW_PRECOMPILE_KECCAK256>>(ctx, w->instances, w->n_instances, &w->cf_pi));
        else if (w->kind == ZKW_PRECOMPILE_SHA256) ZKW_TRY(closed_form_public_inputs<CfPrecompile<ZKW_PRECOMPILE_SHA256>>(ctx, w->instances, w->n_instances, &w->cf_pi));
        else ZKW_TRY(closed_form_public_inputs<CfPrecompile<ZKW_PRECOMPILE_ECRECOVER>>(ctx, w->instances, w->n_instances, &w->cf_pi));
    }
    if (compact) *compact = w->cf_pi;
    if (public_inputs) *public_inputs = w->cf_pi + COMPACT_FORM_LEN * w->n_instances;
    return ZKW_OK;
}

// ------------------------------------------------------------------------------------------------ netlist circuits ("zkw trace v4")
// Sha256RoundFunction (6), CodeDecommitter (3), Keccak256RoundFunction (5), L1MessagesHasher (13): one engine (netlist_kernels.cuh),
// four generated specs on the reference's geometry and table sets. The device copy of a spec (its arrays, the general-purpose cell
// map, the key layout and the histogram plan) is built once per device and circuit and never freed.
namespace {
struct NlCached { NlDev host; NlDev* dev = nullptr; const NlqFreeHome* free_home = nullptr; /* circuits with a queue section: the cell of every FREE element */
                  const NlqFreeHome* link_home = nullptr; /* [operation][64]: the cell (within the cycle) a linked value cell copies, for the fill */ };
std::mutex g_nl_mu;
std::map<std::pair<int, int>, NlCached>& nl_cache() { static auto* m = new std::map<std::pair<int, int>, NlCached>(); return *m; }

template <class T>
int nl_to_device(const T* src, size_t n, const T** out) {
    void* p = nullptr;
    if (hipMalloc(&p, (n ? n : 1) * sizeof(T)) != hipSuccess) return fail(ZKW_ERR_OOM, "netlist spec: hipMalloc failed");
    if (n && hipMemcpy(p, src, n * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return fail(ZKW_ERR_HIP, "netlist spec: upload failed");
    *out = static_cast<const T*>(p);
    return ZKW_OK;
}

int nl_get(zkw_ctx* ctx, int circuit_type, const NlCached** out) {
    const nl_spec* hs = nl_host_spec(circuit_type);
    if (!hs) return fail(ZKW_ERR_INVALID, "circuit type %d is not a netlist circuit", circuit_type);
    std::lock_guard<std::mutex> g(g_nl_mu);
    NlCached& c = nl_cache()[{ctx->device, circuit_type}];
    if (c.dev) { *out = &c; return ZKW_OK; }
    HIP_TRY(hipSetDevice(ctx->device));
    NlDev d;
    memset(&d, 0, sizeof d);
    d.s = *hs;
    ZKW_TRY(nl_to_device(hs->tables, hs->n_tables, &d.s.tables));
    ZKW_TRY(nl_to_device(hs->step_types, hs->n_step_types, &d.s.step_types));
    ZKW_TRY(nl_to_device(hs->ops, hs->n_ops, &d.s.ops));
    ZKW_TRY(nl_to_device(hs->gates, hs->n_gates, &d.s.gates));
    ZKW_TRY(nl_to_device(hs->terms, hs->n_terms, &d.s.terms));
    ZKW_TRY(nl_to_device(hs->hints, hs->n_hints ? hs->n_hints : 1, &d.s.hints));
    ZKW_TRY(nl_to_device(hs->out, (size_t)hs->n_step_types * hs->state, &d.s.out));
    ZKW_TRY(nl_to_device(hs->order, hs->n_order, &d.s.order));
    ZKW_TRY(nl_to_device(hs->level_start, hs->n_level_starts, &d.s.level_start));
    ZKW_TRY(nl_to_device(hs->homes, hs->n_values, &d.s.homes));
    ZKW_TRY(nl_to_device(hs->cycle, hs->steps_per_cycle, &d.s.cycle));
    size_t n_rowend = 0;
    for (u32 k = 0; k < hs->n_step_types; k++) n_rowend += hs->step_types[k].rows;
    ZKW_TRY(nl_to_device(hs->gate_row_end, n_rowend, &d.s.gate_row_end));
    const NlV V(*hs);
    // general-purpose cell map: [type: cell0 + col * rows + row] = dense reference of the cell
    std::vector<u32> cell0(hs->n_step_types);
    std::vector<uint16_t> cmap;
    u32 max_items = 0;
    for (u32 k = 0; k < hs->n_step_types; k++) {
        const nl_step_type& T = hs->step_types[k];
        cell0[k] = (u32)cmap.size();
        cmap.resize(cmap.size() + (size_t)hs->g * T.rows, 0xFFFF);
        uint16_t* m = cmap.data() + cell0[k];
        for (int f = 0; f < NL_HDR_FIELDS; f++) m[(size_t)f * T.rows] = (uint16_t)(V.hdr + f);
        for (u32 gi = 0; gi < T.n_gates; gi++) {
            const nl_gate& gt = hs->gates[T.gate0 + gi];
            for (u32 i = 0; i < (u32)gt.n_known + gt.n_new; i++)
                m[(size_t)(gt.col + i) * T.rows + gt.row] = V.dense(hs->terms[T.term0 + gt.first_term + i].ref);
        }
        max_items = std::max(max_items, T.n_ops + T.n_gates + T.rows);
    }
    ZKW_TRY(nl_to_device(cmap.data(), cmap.size(), &d.cellmap));
    ZKW_TRY(nl_to_device(cell0.data(), cell0.size(), &d.cell0));
    // keys of a cycle: step after step, [slot][lookup row] inside a step
    std::vector<u32> key0(hs->steps_per_cycle);
    u32 keys = 0;
    for (u32 s = 0; s < hs->steps_per_cycle; s++) {
        key0[s] = keys;
        keys += hs->r * hs->step_types[hs->cycle[s].type].lookup_rows;
    }
    d.keys_per_cycle = keys;
    ZKW_TRY(nl_to_device(key0.data(), key0.size(), &d.step_key0));
    // histogram plan: the row runs of every table in every step of a cycle; slices in proportion to the lookups
    std::vector<NlHistEntry> entries;
    std::vector<u32> first(hs->n_tables + 1, 0), slice0(hs->n_tables + 1, 0);
    std::vector<unsigned long long> weight(hs->n_tables, 0);
    for (u32 tb = 0; tb < hs->n_tables; tb++) {
        first[tb] = (u32)entries.size();
        for (u32 s = 0; s < hs->steps_per_cycle; s++) {
            const nl_step_type& T = hs->step_types[hs->cycle[s].type];
            u32 r0 = ~0u, r1 = 0;
            for (u32 r = 0; r < T.lookup_rows; r++)
                if (hs->ops[T.op0 + r * hs->r].table == tb + 1) { r0 = std::min(r0, r); r1 = r + 1; }
            if (r1) { entries.push_back(NlHistEntry{s, r0, r1, key0[s], T.lookup_rows}); weight[tb] += (r1 - r0) * hs->r; }
        }
    }
    first[hs->n_tables] = (u32)entries.size();
    std::vector<u32> slices(hs->n_tables, 1);
    int in_use = 0;  // a table no lookup of the netlist uses (ECRecover's FixedBaseMul tables: its EC section counts those) gets no slice
    for (u32 tb = 0; tb < hs->n_tables; tb++) { slices[tb] = weight[tb] ? 1 : 0; in_use += weight[tb] ? 1 : 0; }
    for (int left = 64 - in_use; left > 0; left--) {  // the next slice goes to the table with the most lookups per slice
        u32 best = 0;
        for (u32 tb = 1; tb < hs->n_tables; tb++)
            if (weight[tb] * slices[best] > weight[best] * slices[tb]) best = tb;
        slices[best]++;
    }
    for (u32 tb = 0; tb < hs->n_tables; tb++) slice0[tb + 1] = slice0[tb] + slices[tb];
    d.n_hist_slices = slice0[hs->n_tables];
    ZKW_TRY(nl_to_device(entries.data(), entries.size(), &d.hist_entries));
    ZKW_TRY(nl_to_device(first.data(), first.size(), &d.hist_first));
    ZKW_TRY(nl_to_device(slice0.data(), slice0.size(), &d.hist_slice0));
    // the gates' known cells, run-length packed (netlist_kernels.cuh NlDev): consecutive cells whose dense references step by 1 and whose
    // shifts step by `step` with one sign fold into one entry, provided every cell of the run is < 2^step (nibbles at step 4, bytes at
    // step 8: true for this format's values, which are nibbles or bytes by construction of the generators)
    std::vector<uint32_t> pk;
    std::vector<uint16_t> pk_first;
    std::vector<u32> pk0(hs->n_step_types);
    for (u32 k = 0; k < hs->n_step_types; k++) {
        const nl_step_type& T = hs->step_types[k];
        pk0[k] = (u32)pk.size();
        for (u32 gi = 0; gi < T.n_gates; gi++) {
            const nl_gate& gt = hs->gates[T.gate0 + gi];
            const nl_term* tm = hs->terms + T.term0 + gt.first_term;
            pk_first.push_back((uint16_t)(pk.size() - pk0[k]));
            for (u32 i = 1; i < gt.n_new; i++) {  // the fill describes a gate's NEW cells as (first value, count, first shift, step)
                const nl_term *a = tm + gt.n_known + i - 1, *b = a + 1;
                if (b->ref != a->ref + 1 || (i > 1 && (b->code & 0x7F) - (a->code & 0x7F) != (a->code & 0x7F) - (a[-1].code & 0x7F)))
                    return fail(ZKW_ERR_INVALID, "netlist circuit %d: the NEW cells of gate %u are not consecutive values at evenly spaced shifts", circuit_type, gi);
            }
            for (u32 i = 0; i < gt.n_known;) {
                if (tm[i].code & NL_TERM_LATE) { i++; continue; }  // in the constraint, not in the fill's evaluation
                const u32 ref = V.dense(tm[i].ref), code = tm[i].code & 0xFF;
                u32 cnt = 1, step = 0;
                if (i + 1 < gt.n_known && !(tm[i + 1].code & NL_TERM_LATE) && V.dense(tm[i + 1].ref) == ref + 1 && (tm[i + 1].code & 0x80) == (code & 0x80) && (tm[i + 1].code & 0x7F) > (code & 0x7F)) {
                    step = (tm[i + 1].code & 0x7F) - (code & 0x7F);
                    const bool nibble_run = hs->w == 4 && step == 4, byte_run = hs->w == 3 && step == 8;  // values < 2^step
                    if (nibble_run || byte_run)
                        while (cnt < 8 && i + cnt < gt.n_known && V.dense(tm[i + cnt].ref) == ref + cnt && tm[i + cnt].code == code + cnt * step) cnt++;
                    else step = 0;
                }
                if (cnt == 1) step = 0;
                pk.push_back(ref | (cnt - 1) << 16 | code << 20 | step << 28);
                i += cnt;
            }
        }
        pk_first.push_back((uint16_t)(pk.size() - pk0[k]));  // closes the step type's last gate
    }
    if (pk.empty()) pk.push_back(0);
    d.n_pk_terms = (u32)pk.size();
    ZKW_TRY(nl_to_device(pk.data(), pk.size(), &d.pk_terms));
    ZKW_TRY(nl_to_device(pk_first.data(), pk_first.size(), &d.pk_first));
    ZKW_TRY(nl_to_device(pk0.data(), pk0.size(), &d.pk0));
    d.max_items = max_items;
    d.vsize = V.size;
    // 16 waves per workgroup (= cycles in flight) wherever the LDS holds them: the Keccak family fits two such workgroups per CU,
    // the SHA-256 family (a compression in three steps, so that a cycle's values are a third) one
    d.lds_bytes = NlLds(*hs, V.size, d.n_pk_terms, 8).total;
    d.lds_bytes16 = NlLds(*hs, V.size, d.n_pk_terms, 16).total;
    d.fill_waves = d.lds_bytes16 <= 160 * 1024 ? 16 : 8;
    if (getenv("ZKW_NL_VERBOSE")) fprintf(stderr, "[zkw] netlist circuit %d: %u terms packed into %u, LDS %u bytes for %u waves\n", circuit_type, hs->n_terms, d.n_pk_terms, d.fill_waves == 16 ? d.lds_bytes16 : d.lds_bytes, d.fill_waves);
    // ---- the lane-per-cycle path: instruction stream with LDS slots (linear scan over the evaluation order), state sources,
    // per-row metadata of a cycle
    {
        auto enc_src = [&](u32 ref, const std::vector<uint16_t>& slot_of) -> uint16_t {
            if (ref < NL_REF_HDR) return (uint16_t)(NL_SRC_VAL << 13 | slot_of[ref]);
            if (ref < NL_REF_PREV) return (uint16_t)(NL_SRC_HDR << 13 | (ref - NL_REF_HDR));
            if (ref < NL_REF_CYC) return (uint16_t)(NL_SRC_PREV << 13 | (ref - NL_REF_PREV));
            if (ref < NL_REF_FREE) return (uint16_t)(NL_SRC_CYC << 13 | (ref - NL_REF_CYC));
            if (ref < NL_REF_RC) return (uint16_t)(NL_SRC_FREE << 13 | (ref - NL_REF_FREE));
            if (ref < NL_REF_CONST) return (uint16_t)(NL_SRC_RC << 13 | (ref - NL_REF_RC));
            return (uint16_t)(NL_SRC_IMM << 13 | (ref - NL_REF_CONST));
        };
        std::vector<u32> prog, prog0(hs->n_step_types);
        std::vector<uint16_t> out_src((size_t)hs->n_step_types * hs->state);
        u32 max_slots = 1;
        for (u32 k = 0; k < hs->n_step_types; k++) {
            const nl_step_type& T = hs->step_types[k];
            const u32 n_items = hs->level_start[T.level0 + T.n_levels];
            // the items in evaluation order as (reads, writes); a fused hint is a hint followed by its lookup
            struct Item { int kind; u32 idx; };  // 0 lookup slot, 1 gate, 2 hint
            std::vector<Item> items;
            for (u32 e = 0; e < n_items; e++) {
                const u32 it = hs->order[T.order0 + e];
                if (it >= NL_ORDER_HINT) items.push_back({2, it - NL_ORDER_HINT});
                else if (it >= NL_ORDER_GATE) items.push_back({1, it - NL_ORDER_GATE});
                else if (it >= NL_ORDER_FUSED) { items.push_back({2, it - NL_ORDER_FUSED}); items.push_back({0, hs->hints[T.hint0 + (it - NL_ORDER_FUSED)].fused_slot}); }
                else items.push_back({0, it});
            }
            const long END = (long)items.size();
            std::vector<long> last_use(T.n_values, -1);
            auto use = [&](u32 ref, long at) { if (ref < NL_REF_HDR && at > last_use[ref]) last_use[ref] = at; };
            for (long i = 0; i < END; i++) {
                const Item& it = items[i];
                if (it.kind == 0) {
                    const nl_op& op = hs->ops[T.op0 + it.idx];
                    const nl_table& tb = hs->tables[op.table - 1];
                    for (u32 a = 0; a < tb.n_in; a++) use(op.in[a], i);
                } else if (it.kind == 1) {
                    const nl_gate& g = hs->gates[T.gate0 + it.idx];
                    const nl_term* tm = hs->terms + T.term0 + g.first_term;
                    for (u32 a = 0; a < g.n_known; a++) use(tm[a].ref, (tm[a].code & NL_TERM_LATE) ? END : i);
                } else {
                    const nl_hint& h = hs->hints[T.hint0 + it.idx];
                    use(h.ref_a, i); use(h.ref_b, i);
                }
            }
            for (u32 e = 0; e < hs->state; e++) use(hs->out[(size_t)k * hs->state + e], END);
            // linear scan: a value's slot is free again after the item that reads it last
            std::vector<uint16_t> slot_of(T.n_values, 0xFFFF);
            std::vector<uint16_t> free_slots;
            std::vector<std::vector<u32>> dies_at(END + 1);
            u32 next_slot = 0;
            auto define = [&](u32 v, long at) -> uint16_t {
                if (last_use[v] < 0) return 0xFFFF;  // never read: not kept
                uint16_t sl;
                if (!free_slots.empty()) { sl = free_slots.back(); free_slots.pop_back(); } else sl = (uint16_t)next_slot++;
                slot_of[v] = sl;
                if (last_use[v] < END) dies_at[last_use[v]].push_back(v);
                (void)at;
                return sl;
            };
            prog0[k] = (u32)prog.size();
            std::vector<std::array<u32, 3>> late;  // (src ref, row, col) of the gates' late cells
            for (long i = 0; i < END; i++) {
                const Item& it = items[i];
                if (it.kind == 0) {
                    const nl_op& op = hs->ops[T.op0 + it.idx];
                    const nl_table& tb = hs->tables[op.table - 1];
                    uint16_t src[3] = {(uint16_t)(NL_SRC_IMM << 13), (uint16_t)(NL_SRC_IMM << 13), (uint16_t)(NL_SRC_IMM << 13)};
                    for (u32 a = 0; a < tb.n_in; a++) src[a] = enc_src(op.in[a], slot_of);
                    for (u32 v : dies_at[i]) free_slots.push_back(slot_of[v]);  // read first, then written: an output may take an input's slot
                    uint16_t dst[3] = {0xFFFF, 0xFFFF, 0xFFFF};
                    if (op.out != 0xFFFF)
                        for (u32 o = 0; o < tb.n_out; o++) dst[o] = define(op.out + o, i);
                    prog.push_back(NL_I_LOOKUP | tb.fn << 4 | tb.param << 8 | tb.n_in << 12 | tb.n_out << 14);
                    prog.push_back(src[0] | (u32)src[1] << 16);
                    prog.push_back(src[2] | (u32)dst[0] << 16);
                    prog.push_back(dst[1] | (u32)dst[2] << 16);
                    prog.push_back((1 + it.idx / hs->r) << 16 | (hs->g + hs->w * (it.idx % hs->r)));
                } else if (it.kind == 1) {
                    const nl_gate& g = hs->gates[T.gate0 + it.idx];
                    const nl_term* tm = hs->terms + T.term0 + g.first_term;
                    bool has_late = false;
                    std::vector<u32> words;
                    for (u32 a = 0; a < g.n_known; a++) {
                        words.push_back(enc_src(tm[a].ref, slot_of) | (u32)(tm[a].code & 0x1FF) << 16);
                        if (tm[a].code & NL_TERM_LATE) { has_late = true; late.push_back({tm[a].ref, g.row, (u32)g.col + a}); }
                    }
                    for (u32 v : dies_at[i]) free_slots.push_back(slot_of[v]);
                    const u32 sh0 = g.n_new ? (tm[g.n_known].code & 0x7F) : 0, step = g.n_new > 1 ? (tm[g.n_known + 1].code & 0x7F) - sh0 : 8;
                    prog.push_back(NL_I_GATE | (u32)g.n_known << 4 | (u32)g.n_new << 12 | (has_late ? 1u : 0u) << 20 | step << 24);
                    prog.push_back(g.constant);
                    prog.push_back((u32)g.row << 16 | g.col);
                    prog.push_back(sh0);
                    prog.insert(prog.end(), words.begin(), words.end());
                    for (u32 a = 0; a < g.n_new; a++) prog.push_back(define(tm[g.n_known + a].ref, i));
                } else {
                    const nl_hint& h = hs->hints[T.hint0 + it.idx];
                    const uint16_t sa = enc_src(h.ref_a, slot_of), sb = enc_src(h.ref_b, slot_of);
                    for (u32 v : dies_at[i]) free_slots.push_back(slot_of[v]);
                    prog.push_back(NL_I_HINT | (u32)h.lo_a << 4 | (u32)h.n_a << 8 | (u32)h.lo_b << 12 | (u32)h.n_b << 16);
                    prog.push_back(sa | (u32)sb << 16);
                    prog.push_back(define(h.value, i));
                }
            }
            for (auto& lc : late) { prog.push_back(NL_I_LATE); prog.push_back(enc_src(lc[0], slot_of)); prog.push_back(lc[1] << 16 | lc[2]); }
            prog.push_back(NL_I_END);
            for (u32 e = 0; e < hs->state; e++) out_src[(size_t)k * hs->state + e] = enc_src(hs->out[(size_t)k * hs->state + e], slot_of);
            max_slots = std::max(max_slots, next_slot);
            if (next_slot >= (1u << 13)) return fail(ZKW_ERR_INVALID, "netlist circuit %d: %u live values", circuit_type, next_slot);
        }
        std::vector<NlRowMeta> rowmeta(hs->rows_per_cycle);
        for (u32 st = 0; st < hs->steps_per_cycle; st++) {
            const nl_step_type& T = hs->step_types[hs->cycle[st].type];
            for (u32 r = 0; r < T.rows; r++) {
                NlRowMeta& m = rowmeta[hs->cycle[st].row0 + r];
                memset(&m, 0, sizeof m);
                m.rowend = (uint8_t)hs->gate_row_end[T.rowend0 + r];
                m.flags = (uint8_t)((r <= T.gate_rows ? 1 : 0) | (r >= 1 && r <= T.lookup_rows ? 2 : 0));
                m.lookup_rows = (uint16_t)T.lookup_rows;
                if (m.flags & 2) {
                    const nl_table& tb = hs->tables[hs->ops[T.op0 + (r - 1) * hs->r].table - 1];
                    m.keyfmt = (uint8_t)(tb.n_in | tb.in_bits << 4);
                    m.key_base = key0[st] + (r - 1);
                }
            }
        }
        prog.resize(prog.size() + 256, NL_I_END);  // (the fetch window reads up to three chunks ahead)
        ZKW_TRY(nl_to_device(prog.data(), prog.size(), &d.prog));
        ZKW_TRY(nl_to_device(prog0.data(), prog0.size(), &d.prog0));
        ZKW_TRY(nl_to_device(out_src.data(), out_src.size(), &d.out_src));
        ZKW_TRY(nl_to_device(rowmeta.data(), rowmeta.size(), &d.rowmeta));
        d.max_slots = max_slots;
        d.walk_lds = (max_slots + 3 * hs->state + hs->max_free) * 64;
        if (getenv("ZKW_NL_VERBOSE")) fprintf(stderr, "[zkw] netlist circuit %d, lane-per-cycle path: %zu program words, %u LDS slots, %u bytes of LDS per wave\n", circuit_type, prog.size(), max_slots, d.walk_lds);
    }
    if (nlq_desc_of(circuit_type)) {  // the one cell of a cycle that uses FREE element i (tools/netlist.py asserts there is exactly one)
        std::vector<NlqFreeHome> fh(hs->free_per_cycle, NlqFreeHome{0xFFFF, 0xFFFF});
        u32 off = 0;
        for (u32 st = 0; st < hs->steps_per_cycle; st++) {
            const nl_cycle_step& cs = hs->cycle[st];
            const nl_step_type& T = hs->step_types[cs.type];
            for (u32 j = 0; j < T.n_ops; j++) {
                const nl_op& op = hs->ops[T.op0 + j];
                if (op.out == 0xFFFF) continue;
                for (u32 i = 0; i < hs->tables[op.table - 1].n_in; i++)
                    if (op.in[i] >= NL_REF_FREE && op.in[i] < NL_REF_RC) fh[off + op.in[i] - NL_REF_FREE] = NlqFreeHome{(uint16_t)(cs.row0 + 1 + j / hs->r), (uint16_t)(hs->g + hs->w * (j % hs->r) + i)};
            }
            for (u32 gi = 0; gi < T.n_gates; gi++) {
                const nl_gate& gt = hs->gates[T.gate0 + gi];
                for (u32 i = 0; i < gt.n_known; i++) {
                    const u32 ref = hs->terms[T.term0 + gt.first_term + i].ref;
                    if (ref >= NL_REF_FREE && ref < NL_REF_RC) fh[off + ref - NL_REF_FREE] = NlqFreeHome{(uint16_t)(cs.row0 + gt.row), (uint16_t)(gt.col + i)};
                }
            }
            off += T.n_free;
        }
        ZKW_TRY(nl_to_device(fh.data(), fh.size(), &c.free_home));
        // the fill's shortcut: where a linked value cell's source sits inside ITS OWN cycle (a FREE element's cell; a state element
        // after the cycle = the value the last step leaves); 0xFFFF = resolve it the long way (nl_home_cell), as the checker always does
        const nlq_desc* qd = nlq_desc_of(circuit_type);
        std::vector<NlqFreeHome> lh((size_t)NLQ_MAX_OPS * 64, NlqFreeHome{0xFFFF, 0xFFFF});
        const nl_cycle_step& ls = hs->cycle[hs->steps_per_cycle - 1];
        const nl_step_type& LT = hs->step_types[ls.type];
        for (u32 j = 0; j < qd->n_ops; j++)
            for (u32 cell = NLQ_MEM_NIBBLE0; cell < NLQ_MEM_NIBBLE0 + 64; cell++) {
                if (!nlq_comp_linked(&qd->ops[j], cell) || qd->ops[j].link == NLQ_LINK_LH_MESSAGE) continue;  // (links that depend on the cycle: the long way)
                uint32_t next = 0;
                const u32 ref = nlq_link_ref(&qd->ops[j], cell, &next);
                NlqFreeHome& o = lh[(size_t)j * 64 + (cell - NLQ_MEM_NIBBLE0)];
                if (!next) { o = fh[ref - NL_REF_FREE]; continue; }
                const u32 r2 = hs->out[(size_t)ls.type * hs->state + (ref - NL_REF_CYC)];
                if (r2 >= NL_REF_HDR) continue;
                const nl_home h = hs->homes[LT.home0 + r2];
                if (h.kind == 1) { const nl_gate& gt = hs->gates[LT.gate0 + h.item]; o = NlqFreeHome{(uint16_t)(ls.row0 + gt.row), (uint16_t)(gt.col + h.cell)}; }
                else o = NlqFreeHome{(uint16_t)(ls.row0 + 1 + h.item / hs->r), (uint16_t)(hs->g + hs->w * (h.item % hs->r) + h.cell)};
            }
        ZKW_TRY(nl_to_device(lh.data(), lh.size(), &c.link_home));
    }
    const NlDev* dd = nullptr;
    ZKW_TRY(nl_to_device(&d, 1, &dd));
    c.host = d;
    c.dev = const_cast<NlDev*>(dd);
    *out = &c;
    return ZKW_OK;
}

struct NlInstance { u64 first_round; u32 n_active; const u64* public_input; const zkw_trace* t; size_t slot; bool fresh = false; /* the hash state before the instance is zero, not what round first_round - 1 left (independent queues in one call) */ };

template <int W, int R, int WAVES>
int nl_launch_fill_w(zkw_ctx* ctx, const NlCached* nc, const NlJob* d_jobs, unsigned nj, u32 capacity, size_t n_rows) {
    using L = Launcher<&k_nl_fill<W, R, WAVES>, 64 * WAVES>;
    static bool attr_set[16] = {};
    if (!attr_set[ctx->device & 15]) {  // more than the default 64 KB of dynamic LDS
        // (the 16-wave form is a launch of its own only: as a job of a merged launch its 64 registers per lane do not hold the job lookup too)
        if constexpr (WAVES == 16) HIP_TRY(hipFuncSetAttribute(L::S::template single_fn<&k_nl_fill<W, R, WAVES>, 64 * WAVES>(), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        else ZKW_TRY(L::allow_dynamic_lds(160 * 1024));
        attr_set[ctx->device & 15] = true;
    }
#ifdef ZKW_PROBE_BUILD  // measurement builds only (ZKW_PROBE_BUILD=1 python -m era_zkevm_test_harness_amd.build --force): 1 = no level walk, 2 = no
                        // streaming — the traces are then INVALID, so the knob does not exist in the library that ships
    static const u32 probe = [] { const char* e = getenv("ZKW_NL_PROBE"); return e ? (u32)atoi(e) : 0u; }();
    if (probe) fprintf(stderr, "libzkw: ZKW_NL_PROBE=%u — k_nl_fill skips work, the traces are invalid\n", probe);
#else
    const u32 probe = 0;
#endif
    // as many workgroups as the LDS lets a CU hold: the write phase is a stream of stores and wants waves in flight
    const unsigned lds = WAVES == 16 ? nc->host.lds_bytes16 : nc->host.lds_bytes;
    const unsigned per_cu = std::max<unsigned>(1, std::min<unsigned>(4, (160u * 1024u) / std::max<unsigned>(1, lds)));
    const unsigned blocks = std::min<unsigned>((capacity + WAVES - 1) / WAVES, std::max<unsigned>(1, 256 * per_cu / nj));
    if constexpr (WAVES == 16) {
        Prof _p(ctx, "k_nl_fill");
        L::S::template single<&k_nl_fill<W, R, WAVES>, 64 * WAVES>(ctx->stream, dim3(blocks, nj), lds, nc->dev, d_jobs, capacity, n_rows, probe);
    } else {
        Prof _p(ctx, "k_nl_fill");
        ZKW_LAUNCH_D(ctx, (k_nl_fill<W, R, WAVES>), "k_nl_fill", dim3(blocks, nj), 64 * WAVES, lds, nc->dev, d_jobs, capacity, n_rows, probe);
    }
    return launch_check("k_nl_fill");
}
// the lane-per-cycle path (k_nl_walk + k_nl_expand), for netlists whose live values fit the LDS
template <int W, int R>
int nl_launch_fill_lanes(zkw_ctx* ctx, const NlCached* nc, const NlJob* d_jobs, unsigned nj, u32 capacity, size_t n_rows) {
    static bool attr_set[16] = {};
    if (!attr_set[ctx->device & 15]) {
        ZKW_TRY((Launcher<&k_nl_walk<W, R>, 64>::allow_dynamic_lds(160 * 1024)));
        attr_set[ctx->device & 15] = true;
    }
    const nl_spec& S = nc->host.s;
    const unsigned tiles = (capacity + 63) / 64, row_blocks = (S.rows_per_cycle + 63) / 64;
    const size_t per_job = ((size_t)tiles * S.mult_col * S.rows_per_cycle * 64 + 255) & ~(size_t)255;
    uint8_t* d_bytes = nullptr;
    ZKW_TRY(ctx->scratch_t<uint8_t>("nl_bytes", per_job * nj, &d_bytes));
    { Prof _p(ctx, "k_nl_walk"); ZKW_LAUNCH_D(ctx, (k_nl_walk<W, R>), "k_nl_walk", dim3(tiles, nj), 64, nc->host.walk_lds, nc->dev, d_jobs, capacity, d_bytes, per_job, nc->host.prog, nc->host.prog0, nc->host.out_src); }
    ZKW_TRY(launch_check("k_nl_walk"));
    { Prof _p(ctx, "k_nl_expand"); ZKW_LAUNCH_D(ctx, (k_nl_expand<W, R>), "k_nl_expand", dim3((S.g + R) * row_blocks, tiles, nj), 64, 0, nc->dev, d_jobs, capacity, n_rows, d_bytes, per_job); }
    return launch_check("k_nl_expand");
}

using NlAfterFill = std::function<int()>;  // runs once the fill kernel is queued, before the histogram / finish / queue-section kernels (the closed-form sponges fork here)
template <int W, int R>
int nl_launch_fill(zkw_ctx* ctx, const NlCached* nc, const NlJob* d_jobs, unsigned nj, u32 capacity, size_t n_rows, const NlAfterFill& after_fill) {
    // the lane-per-cycle form is opt-in (zkw_set_netlist_fill_form, or ZKW_NL_LANES=1 for a whole process): bit-identical, measured
    // SLOWER than the wave-per-cycle form on every circuit
    static const int lanes_env = [] { const char* e = getenv("ZKW_NL_LANES"); return e ? atoi(e) : 0; }();
    if ((lanes_env || ctx->netlist_fill_form == 1) && nc->host.walk_lds <= 160 * 1024) {
        ZKW_TRY((nl_launch_fill_lanes<W, R>(ctx, nc, d_jobs, nj, capacity, n_rows)));
        if (after_fill) ZKW_TRY(after_fill());
        { Prof _p(ctx, "k_nl_hist"); ZKW_LAUNCH_D(ctx, (k_nl_hist<R>), "k_nl_hist", dim3(nc->host.n_hist_slices, nc->host.s.total_table_rows > NL_HIST_HALF ? 2 : 1, nj), NL_HIST_THREADS, 0, nc->dev, d_jobs, capacity, n_rows); }
        return launch_check("k_nl_hist");
    }
    // (a call with few cycles keeps 8 waves per workgroup: twice the workgroups, so that every CU has one)
    if (nc->host.fill_waves == 16 && (size_t)((capacity + 15) / 16) * nj >= 128 && !ctx->batched()) ZKW_TRY((nl_launch_fill_w<W, R, 16>(ctx, nc, d_jobs, nj, capacity, n_rows)));
    else ZKW_TRY((nl_launch_fill_w<W, R, 8>(ctx, nc, d_jobs, nj, capacity, n_rows)));
    if (after_fill) ZKW_TRY(after_fill());
    { Prof _p(ctx, "k_nl_hist"); ZKW_LAUNCH_D(ctx, (k_nl_hist<R>), "k_nl_hist", dim3(nc->host.n_hist_slices, nc->host.s.total_table_rows > NL_HIST_HALF ? 2 : 1, nj), NL_HIST_THREADS, 0, nc->dev, d_jobs, capacity, n_rows); }
    return launch_check("k_nl_hist");
}

// synthesis of instances of one netlist circuit: `prepare` turns the builder's records into the engine's inputs (header bits, free
// elements, the state before every cycle) at the pointers of the jobs it is handed; `capacity` is in cycles
using NlPrepare = std::function<int(std::vector<NlPrepJob>&)>;
// `hold`: the caller has more launches to make into the same slots (the queue, EC and closed-form sections): the slots' layout tags are handed to
// it uncommitted and it commits them after ITS last launch (committed here, a failure in those later launches left a slot tagged with
// a layout whose lower sections were never written)
int nl_synthesize_with(zkw_ctx* ctx, int circuit_type, const NlPrepare& prepare, const std::vector<NlInstance>& inst, u32 capacity, size_t n_rows,
                       const NlAfterFill& after_fill = {}, SlotClaims* hold = nullptr) {
    const NlCached* nc = nullptr;
    ZKW_TRY(nl_get(ctx, circuit_type, &nc));
    const nl_spec& S = nc->host.s;
    if ((nc->host.fill_waves == 16 ? nc->host.lds_bytes16 : nc->host.lds_bytes) > 160 * 1024) return fail(ZKW_ERR_INVALID, "netlist of circuit %d needs %u bytes of LDS", circuit_type, nc->host.lds_bytes);
    const size_t used = nlcf_used_rows(circuit_type, &S, capacity);  // netlist rows + the queue, EC and closed-form sections
    if (used > n_rows || S.total_table_rows > n_rows)
        return fail(ZKW_ERR_INVALID, "capacity %u needs %zu rows (tables: %u), trace has %zu", capacity, used, S.total_table_rows, n_rows);
    const size_t ni = inst.size();
    if (ni == 0) return ZKW_OK;
    uint8_t *d_hdr = nullptr, *d_free = nullptr, *d_state = nullptr;
    uint16_t* d_keys = nullptr;
    const size_t hdr_n = capacity, free_n = (size_t)capacity * S.free_per_cycle, state_n = (size_t)(capacity + 1) * S.state, keys_n = (size_t)capacity * nc->host.keys_per_cycle;
    ZKW_TRY(ctx->scratch_t<uint8_t>("nl_hdr", ni * hdr_n, &d_hdr));
    ZKW_TRY(ctx->scratch_t<uint8_t>("nl_free", ni * free_n + 1, &d_free));
    ZKW_TRY(ctx->scratch_t<uint8_t>("nl_state", ni * state_n, &d_state));
    ZKW_TRY(ctx->scratch_t<uint16_t>("nl_keys", ni * keys_n, &d_keys));
    u32* d_hist = nullptr;
    const size_t hist_n = (size_t)nc->host.n_hist_slices * 2 * NL_HIST_HALF;
    ZKW_TRY(ctx->scratch_t<u32>("nl_hist", ni * hist_n, &d_hist));
    std::vector<NlPrepJob> prep(ni);
    std::vector<NlJob> jobs(ni);
    SlotClaims claims;  // the slots' layout tags are written after the last launch: a failed call leaves them "unknown" (zkw_ctx.h)
    const size_t bnd = NL_BOUNDARY_ROW(&S, capacity);
    for (size_t k = 0; k < ni; k++) {
        prep[k] = NlPrepJob{nullptr, inst[k].first_round, inst[k].n_active, d_hdr + k * hdr_n, d_free + k * free_n, d_state + k * state_n, S.state};
        // The fill writes the lookup cells of every row above the boundary and the general-purpose cells of the header / gate rows;
        // everything else is zero. A slot whose previous tenant was the same layout (circuit, capacity, rows) already has those
        // zeros: nothing to clear (the multiplicity column is rewritten over the tables' rows). Otherwise: clear it.
        const uint64_t tag = ((uint64_t)circuit_type << 56) ^ ((uint64_t)capacity << 24) ^ (uint64_t)n_rows ^ 0x5A00000000000000ull;
        bool clean = false;
        u64* tr = claims.claim(inst[k].t, inst[k].slot, tag, &clean);
        jobs[k] = NlJob{prep[k].hdr_bits, prep[k].free_elems, prep[k].state_before, inst[k].public_input, tr, d_keys + k * keys_n, d_hist + k * hist_n};
        if (!clean) {
            HIP_TRY(ctx->memset_async(tr, 0, (size_t)S.g * n_rows * sizeof(u64)));  // general-purpose columns
            ZKW_LAUNCH_2D(ctx, k_zero_strip, (unsigned)((n_rows - bnd + 255) / 256), S.mult_col - S.g, 256, tr + (size_t)S.g * n_rows + bnd, n_rows, n_rows - bnd);
            ZKW_TRY(launch_check("k_zero_strip"));
            HIP_TRY(ctx->memset_async(tr + (size_t)S.mult_col * n_rows, 0, n_rows * sizeof(u64)));
        }
    }
    NlJob* d_jobs = nullptr;
    ZKW_TRY(ctx->upload("nl_jobs", jobs, &d_jobs));
    const unsigned nj = (unsigned)ni;
    ZKW_TRY(prepare(prep));
    static_assert(SA_W == LH_W && SA_R == LH_R, "StorageApplication shares the fill instantiation of the LinearHasher geometry");
    switch (circuit_type) {
        case 6: ZKW_TRY((nl_launch_fill<SC_W, SC_R>(ctx, nc, d_jobs, nj, capacity, n_rows, after_fill))); break;
        case 3: ZKW_TRY((nl_launch_fill<DC_W, DC_R>(ctx, nc, d_jobs, nj, capacity, n_rows, after_fill))); break;
        case 5: ZKW_TRY((nl_launch_fill<KC_W, KC_R>(ctx, nc, d_jobs, nj, capacity, n_rows, after_fill))); break;
        case 7: ZKW_TRY((nl_launch_fill<EK_W, EK_R>(ctx, nc, d_jobs, nj, capacity, n_rows, after_fill))); break;
        default: ZKW_TRY((nl_launch_fill<LH_W, LH_R>(ctx, nc, d_jobs, nj, capacity, n_rows, after_fill))); break;  // 13 and 10: 3 x 26
    }
    { Prof _p(ctx, "k_nl_finish"); ZKW_LAUNCH_2D(ctx, k_nl_finish, (std::max(S.state, S.total_table_rows) + 255) / 256, nj, 256, nc->dev, d_jobs, capacity, n_rows); }
    const int rc = launch_check("k_nl_finish");
    if (rc == ZKW_OK && hold) {
        hold->pending.insert(hold->pending.end(), claims.pending.begin(), claims.pending.end());
        return ZKW_OK;
    }
    return claims.commit_if(rc);
}

// ... from the block's round records (`sha_like`: zkw_sha256_round_record, else zkw_keccak_round_record)
int nl_synthesize(zkw_ctx* ctx, int circuit_type, bool sha_like, const void* d_rounds, const std::vector<NlInstance>& inst, u32 capacity, size_t n_rows,
                  const NlAfterFill& after_fill = {}, SlotClaims* hold = nullptr) {
    return nl_synthesize_with(ctx, circuit_type, [&](std::vector<NlPrepJob>& prep) {
        const size_t rec_bytes = sha_like ? sizeof(zkw_sha256_round_record) : sizeof(zkw_keccak_round_record);
        for (size_t k = 0; k < prep.size(); k++) {
            if (inst[k].fresh) { prep[k].rounds = static_cast<const char*>(d_rounds) + inst[k].first_round * rec_bytes; prep[k].first_round = 0; }
            else prep[k].rounds = d_rounds;
        }
        NlPrepJob* d_prep = nullptr;
        ZKW_TRY(ctx->upload("nl_prep", prep, &d_prep));
        const unsigned nj = (unsigned)prep.size();
        if (sha_like) { Prof _p(ctx, "k_nl_prepare"); ZKW_LAUNCH_2D(ctx, k_nl_prepare_sha, capacity + 1, nj, 128, d_prep, capacity); }
        else { Prof _p(ctx, "k_nl_prepare"); ZKW_LAUNCH_2D(ctx, k_nl_prepare_keccak, capacity + 1, nj, 256, d_prep, capacity); }
        return launch_check("k_nl_prepare");
    }, inst, capacity, n_rows, after_fill, hold);
}

// the queue section of the instances nl_synthesize has just filled (include/zkw_netlist_queue.h): request-queue pops and memory-queue
// pushes as Poseidon2 rows below the netlist, on the same stream (the fill reads the linked netlist cells back)
struct NlqQueues { NlqQueueIn q[NLQ_MAX_QUEUES]; const RoundOps* round_ops; };
int nlq_synthesize(zkw_ctx* ctx, int circuit_type, const NlqQueues& Q, const std::vector<NlInstance>& inst, u32 capacity, size_t n_rows,
                   const std::vector<NlqQueues>* per_instance = nullptr /* independent queues per instance (L1MessagesHasher) */) {
    const nlq_desc* d = nlq_desc_of(circuit_type);
    if (!d || inst.empty()) return ZKW_OK;
    const NlCached* nc = nullptr;
    ZKW_TRY(nl_get(ctx, circuit_type, &nc));
    const size_t ni = inst.size(), feed_n = (size_t)capacity * d->n_ops;
    nlq_feed* d_feed = nullptr;
    ZKW_TRY(ctx->scratch_t<nlq_feed>("nlq_feed", ni * feed_n, &d_feed));
    std::vector<NlqFeedJob> fj(ni);
    std::vector<NlqJob> jobs(ni);
    for (size_t k = 0; k < ni; k++) {
        const NlqQueues& QK = per_instance ? (*per_instance)[k] : Q;
        fj[k] = NlqFeedJob{QK.round_ops, inst[k].first_round, inst[k].n_active, d_feed + k * feed_n, QK.q[0].n_items};
        jobs[k].feed = fj[k].feed;
        jobs[k].trace = inst[k].t->data + inst[k].slot * inst[k].t->slot_elems();  // (the slot keeps the tag nl_synthesize_with gave it)
        for (u32 q = 0; q < NLQ_MAX_QUEUES; q++) jobs[k].queues[q] = QK.q[q];
    }
    NlqFeedJob* d_fj = nullptr;
    NlqJob* d_jobs = nullptr;
    ZKW_TRY(ctx->upload("nlq_feed_jobs", fj, &d_fj));
    ZKW_TRY(ctx->upload("nlq_jobs", jobs, &d_jobs));
    const unsigned cb = (capacity + 63) / 64;
    { Prof _p(ctx, "k_nlq_feed"); ZKW_LAUNCH_2D(ctx, k_nlq_feed, cb, (unsigned)ni, 64, circuit_type, d_fj, capacity, d->n_ops); }
    ZKW_TRY(launch_check("k_nlq_feed"));
    { Prof _p(ctx, "k_nlq_fill"); ZKW_LAUNCH_D(ctx, (k_nlq_fill), "k_nlq_fill", dim3((capacity + 3) / 4, d->n_ops, (unsigned)ni), 64, 0, nc->dev, nc->free_home, nc->link_home, *d, d_jobs, capacity, n_rows); }
    return launch_check("k_nlq_fill");
}

// ---- the closed-form section (include/zkw_netlist_closed_form.h; netlist_closed_form_kernels.cuh) of the instances first_index + k of a
// witness's records (d_inst) in the slots of inst[k]. nlcf_begin: flags, words, the four sponges and the compact form, on the context's
// side stream — a chain of up to 58 dependent permutations in ONE wave per instance (0.1 - 0.6 ms of latency, no work). It reads only the
// records; WHEN it runs was measured (8 instances per call, tools/probe_netlist_perf.py): on the main stream Keccak 2.08 -> 2.67 ms, SHA-256
// 3.01 -> 3.13; on the side stream from the start of the call, NEXT TO k_nl_fill, Keccak 2.26 but SHA-256 3.35 — the fill is a persistent grid
// that fills every CU's registers, a CU that holds a sponge wave first cannot take its fill workgroup until the sponge is done, and that
// workgroup's whole share ends late; so the fork is AFTER the fill kernel (NlAfterFill): the sponges run next to the histogram, finish and
// queue-section kernels, which leave room. nlcf_end: joins, then the tie cells (copies of the registers the fills have written) on the main stream.
struct NlcfCall { NlcfJob* d_jobs = nullptr; size_t n = 0; bool forked = false; };
template <class T>
int nlcf_begin(zkw_ctx* ctx, int circuit_type, const typename T::Inst* d_inst, size_t first_index, const std::vector<NlInstance>& inst, u32 cycles, size_t n_rows,
               NlcfCall* call, const char* jobs_name = "nlcf_jobs", bool on_main_stream = false) {
    const nlcf_desc* d = nlcf_desc_of(circuit_type);
    const nl_spec* S = nl_host_spec(circuit_type);
    if (!d || !S) return fail(ZKW_ERR_INVALID, "circuit type %d has no closed-form section", circuit_type);
    if (nlcf_used_rows(circuit_type, S, cycles) > n_rows) return fail(ZKW_ERR_INVALID, "the closed-form section does not fit the trace");
    static_assert(sizeof(u64) * 4 * T::MAXLEN <= 60 * 1024, "the encodings are staged in LDS");
    std::vector<NlcfJob> jobs(inst.size());
    for (size_t k = 0; k < inst.size(); k++) jobs[k] = NlcfJob{inst[k].t->data + inst[k].slot * inst[k].t->slot_elems(), (u64)(first_index + k)};
    ZKW_TRY(ctx->upload(jobs_name, jobs, &call->d_jobs));
    call->n = jobs.size();
    if (jobs.empty()) return ZKW_OK;
    if (ctx->batched() || on_main_stream) {  // a context of a batch has one stream: the sponges of all the group's blocks travel as one launch, behind the fills
                                             // (on_main_stream: the caller's own fork is open — ECRecover's EC section)
        ZKW_LAUNCH_D(ctx, (k_nlcf_sponges<T>), "k_nlcf_sponges", dim3((unsigned)jobs.size()), 256, 0, *d, d_inst, (const NlcfJob*)call->d_jobs, S->g, n_rows, (u64)nlcf_first_row(circuit_type, S, cycles));
        return ZKW_OK;
    }
    hipStream_t side = nullptr;
    ZKW_TRY(ctx->side_fork(&side));
    call->forked = true;
    Launcher<&k_nlcf_sponges<T>, 256>::S::template single<&k_nlcf_sponges<T>, 256>(side, dim3((unsigned)jobs.size()), 0, *d, d_inst, call->d_jobs, S->g, n_rows, (u64)nlcf_first_row(circuit_type, S, cycles));
    return launch_check("k_nlcf_sponges");
}
int nlcf_end(zkw_ctx* ctx, int circuit_type, const NlcfCall& call, u32 cycles, size_t n_rows) {
    if (call.n == 0) return ZKW_OK;
    const nlcf_desc* d = nlcf_desc_of(circuit_type);
    const NlCached* nc = nullptr;
    ZKW_TRY(nl_get(ctx, circuit_type, &nc));
    if (call.forked) ZKW_TRY(ctx->side_join());
    const u32 cells = nlcf_header_cells(d) - nlcf_group_cell0(d, 0);
    if (cells == 0) return ZKW_OK;
    const nlq_desc* qd = nlq_desc_of(circuit_type);
    const nlq_desc none{};
    { Prof _p(ctx, "k_nlcf_ties"); ZKW_LAUNCH_2D(ctx, k_nlcf_ties, (cells + 255) / 256, (unsigned)call.n, 256, *d, qd ? *qd : none, nc->dev, call.d_jobs, cycles, n_rows, (u64)nlcf_first_row(circuit_type, &nc->host.s, cycles)); }
    return launch_check("k_nlcf_ties");
}

// ---- the EC section of the ECRecover circuit (ecrecover_kernels.cuh): the spec on the device, once per device
struct EcCached { ec_spec host; const ec_spec* dev = nullptr; u32 segments_per_cycle = 0; EcStreamDev stream{}; const EcTask* tasks_main = nullptr; const EcTask* tasks_muls = nullptr; const EcTask* tasks_leaves = nullptr; u32 n_main = 0, n_muls = 0, n_leaves = 0; };
std::map<int, EcCached>& ec_cache() { static auto* m = new std::map<int, EcCached>(); return *m; }
size_t ec_first_row(u32 capacity) { return nlq_used_rows(nl_host_spec(7), nlq_desc_of(7), capacity); }
size_t ec_used_rows(u32 capacity) { return ec_first_row(capacity) + (size_t)capacity * EC_ROWS_PER_CYCLE; }
int ec_get(zkw_ctx* ctx, const EcCached** out) {
    std::lock_guard<std::mutex> g(g_nl_mu);
    EcCached& c = ec_cache()[ctx->device];
    if (c.dev) { *out = &c; return ZKW_OK; }
    HIP_TRY(hipSetDevice(ctx->device));
    static const std::vector<uint32_t>* fixed = [] {  // the 256 FixedBaseMul tables (8 192 curve points), built once per process
        auto* v = new std::vector<uint32_t>(EC_FIXED_WORDS);
        ec_build_fixed_tables(v->data());
        return v;
    }();
    ec_spec d;
    ZKW_TRY(nl_to_device(h_ecs_types, EC_NUM_TYPES, &d.types));
    ZKW_TRY(nl_to_device(h_ecs_runs, EC_NUM_RUNS, &d.runs));
    ZKW_TRY(nl_to_device(h_ecs_items, EC_NUM_ITEM_WORDS, &d.items));
    ZKW_TRY(nl_to_device(h_ecs_item_index, EC_NUM_ITEMS, &d.item_index));
    ZKW_TRY(nl_to_device(h_ecs_cells, EC_NUM_CELLS, &d.cells));
    ZKW_TRY(nl_to_device(h_ecs_homes, EC_NUM_HOMES, &d.homes));
    ZKW_TRY(nl_to_device(h_ecs_outs, (size_t)EC_NUM_TYPES * EC_STATE, &d.outs));
    ZKW_TRY(nl_to_device(h_ecs_rowtab, EC_NUM_ROWTAB, &d.rowtab));
    ZKW_TRY(nl_to_device(h_ecs_globs, EC_GL_COUNT, &d.globs));
    ZKW_TRY(nl_to_device(h_ecs_bigs, (size_t)EC_NUM_BIGS * 16, &d.bigs));
    ZKW_TRY(nl_to_device(h_ecs_in_home, 128, &d.in_home));
    ZKW_TRY(nl_to_device(h_ecs_key_byte, 64, &d.key_byte));
    ZKW_TRY(nl_to_device(fixed->data(), fixed->size(), &d.fixed));
    const ec_spec* dd = nullptr;
    ZKW_TRY(nl_to_device(&d, 1, &dd));
    c.host = ec_spec{h_ecs_types, h_ecs_runs, h_ecs_items, h_ecs_item_index, h_ecs_cells, h_ecs_homes, h_ecs_outs, h_ecs_rowtab, h_ecs_globs, h_ecs_bigs, h_ecs_in_home, h_ecs_key_byte, fixed->data()};
    for (u32 r = 0; r < EC_NUM_RUNS; r++) c.segments_per_cycle += h_ecs_runs[r].count;
    {   // the rows of a cycle with every reference resolved (k_ec_stream): kind << 30 | payload, column-major
        std::vector<u32> refs((size_t)EC_ROW_CELLS * EC_ROWS_PER_CYCLE);
        std::vector<uint16_t> row_table(EC_ROWS_PER_CYCLE), xor_index(EC_ROWS_PER_CYCLE, 0xFFFF);
        std::vector<u32> fix_rows;
        u32 n_xor = 0;
        for (u32 r = 0; r < EC_ROWS_PER_CYCLE; r++) {
            u32 run, inst, row, prun, pinst;
            ec_locate_row(&c.host, r, &run, &inst, &row);
            ec_prev_segment(&c.host, run, inst, &prun, &pinst);
            const ec_seg_type& T = h_ecs_types[h_ecs_runs[run].type];
            const u32 base = h_ecs_runs[run].tape0 + inst * T.n_tape, pbase = h_ecs_runs[prun].tape0 + pinst * h_ecs_types[h_ecs_runs[prun].type].n_tape;
            for (u32 col = 0; col < EC_ROW_CELLS; col++) {
                const u32 ref = h_ecs_cells[T.cell0 + (size_t)row * EC_ROW_CELLS + col];
                u32 e = 0;  // an empty cell: the constant 0
                if (ref != EC_NONE) {
                    const u32 t = ec_ref_tape(&c.host, ref, base, pbase, h_ecs_runs[prun].type, inst);
                    if (t != EC_NONE) e = 1u << 30 | t;
                    else if ((ref >> 28) == EC_K_IN) e = 2u << 30 | (ref & 0x0FFFFFFFu);
                    else {
                        const uint64_t v = ec_ref_const(&c.host, ref, nullptr);
                        if (v >> 30) return fail(ZKW_ERR_INVALID, "ECRecover spec: a constant cell of %llu does not fit the resolved row table", (unsigned long long)v);
                        e = (u32)v;
                    }
                }
                refs[(size_t)col * EC_ROWS_PER_CYCLE + r] = e;
            }
            row_table[r] = (uint16_t)ec_row_table(&c.host, run, inst, row);
            if (row_table[r] == EC_T_XOR8) xor_index[r] = (uint16_t)n_xor++;
            else if (row_table[r]) fix_rows.push_back(r | (u32)row_table[r] << 16);
        }
        ZKW_TRY(nl_to_device(refs.data(), refs.size(), &c.stream.refs));
        ZKW_TRY(nl_to_device(row_table.data(), row_table.size(), &c.stream.row_table));
        ZKW_TRY(nl_to_device(xor_index.data(), xor_index.size(), &c.stream.xor_index));
        c.stream.n_xor_rows = n_xor;
        ZKW_TRY(nl_to_device(fix_rows.data(), fix_rows.size(), &c.stream.fix_rows));
        c.stream.n_fix_rows = (u32)fix_rows.size();
    }
    {   // the item lists of k_ec_segments / k_ec_leaves: MAIN of the segments after PRE (PRE's is the chain kernel's), MULS, LEAVES
        std::vector<EcTask> mains, muls, leaves;
        for (u32 r = 0; r < EC_NUM_RUNS; r++)
            for (u32 inst = 0; inst < h_ecs_runs[r].count; inst++) {
                u32 first = 0;
                for (u32 p = 0; p < EC_MAX_PARTS; p++) {
                    const u32 n = EC_PART_ITEMS[h_ecs_runs[r].type][p];
                    if (n && !(r == 0 && p == 0)) (p == 0 ? mains : p == 1 ? muls : leaves).push_back(EcTask{r, inst, first, n});
                    first += n;
                }
                if (first != h_ecs_types[h_ecs_runs[r].type].n_items) return fail(ZKW_ERR_INVALID, "ECRecover spec: the parts of segment type %u do not add up", h_ecs_runs[r].type);
            }
        ZKW_TRY(nl_to_device(mains.data(), mains.size(), &c.tasks_main));
        ZKW_TRY(nl_to_device(muls.data(), muls.size(), &c.tasks_muls));
        ZKW_TRY(nl_to_device(leaves.data(), leaves.size(), &c.tasks_leaves));
        c.n_main = (u32)mains.size();
        c.n_muls = (u32)muls.size();
        c.n_leaves = (u32)leaves.size();
    }
    c.dev = dd;
    *out = &c;
    return ZKW_OK;
}

int nl_check(zkw_ctx* ctx, int circuit_type, const zkw_trace* t, size_t slot, u32 capacity, uint64_t* n_violations, uint64_t* first_bad) {
    const NlCached* nc = nullptr;
    ZKW_TRY(nl_get(ctx, circuit_type, &nc));
    const nl_spec& S = nc->host.s;
    if (t->n_cols < S.cols) return fail(ZKW_ERR_INVALID, "trace has %zu columns, the circuit needs %u", t->n_cols, S.cols);
    const nlq_desc* qd = nlq_desc_of(circuit_type);
    if (nlcf_used_rows(circuit_type, &S, capacity) > t->n_rows) return fail(ZKW_ERR_INVALID, "capacity does not fit the trace");
    HIP_TRY(hipSetDevice(ctx->device));
    const u64* trace = t->data + slot * t->slot_elems();
    const size_t n_rows = t->n_rows;
    CheckResult* d_res = nullptr;
    u32* d_hist = nullptr;
    ZKW_TRY(ctx->scratch_t<CheckResult>("check_res", 1, &d_res));
    ZKW_TRY(ctx->scratch_t<u32>("nl_check_hist", S.total_table_rows, &d_hist));
    CheckResult init{0ull, ~0ull};
    HIP_TRY(ctx->copy_async(d_res, &init, sizeof init, hipMemcpyHostToDevice));
    HIP_TRY(ctx->memset_async(d_hist, 0, S.total_table_rows * sizeof(u32)));
    { Prof _p(ctx, "k_nl_check_steps"); hipLaunchKernelGGL(k_nl_check_steps, dim3((nc->host.max_items + 255) / 256, capacity * S.steps_per_cycle), dim3(256), 0, ctx->stream, nc->dev, trace, capacity, n_rows, d_hist, d_res); }
    ZKW_TRY(launch_check("k_nl_check_steps"));
    u64 e_begin = 0, e_end = 0;
    if (circuit_type == 7) {  // the EC section below the queue section: its own checker, which adds its lookups to the histogram first
        const EcCached* ec = nullptr;
        ZKW_TRY(ec_get(ctx, &ec));
        e_begin = ec_first_row(capacity); e_end = ec_used_rows(capacity);
        if (e_end > n_rows) return fail(ZKW_ERR_INVALID, "capacity does not fit the trace");
        { Prof _p(ctx, "k_ec_check_items"); hipLaunchKernelGGL(k_ec_check_items, dim3((EC_MAX_ITEMS + 255) / 256, ec->segments_per_cycle, capacity), dim3(256), 0, ctx->stream, ec->dev, trace, capacity, n_rows, (size_t)e_begin, d_res); }
        ZKW_TRY(launch_check("k_ec_check_items"));
        { Prof _p(ctx, "k_ec_check_rows"); hipLaunchKernelGGL(k_ec_check_rows, dim3((EC_ROWS_PER_CYCLE + 63) / 64, capacity), dim3(64), 0, ctx->stream, ec->dev, nc->dev, *qd, trace, capacity, n_rows, (size_t)e_begin, d_hist, d_res); }
        ZKW_TRY(launch_check("k_ec_check_rows"));
        { Prof _p(ctx, "k_ec_check_links"); hipLaunchKernelGGL(k_ec_check_links, dim3(capacity), dim3(128), 0, ctx->stream, ec->dev, nc->dev, nc->free_home, trace, n_rows, (size_t)e_begin, d_res); }
        ZKW_TRY(launch_check("k_ec_check_links"));
    }
    { Prof _p(ctx, "k_nl_check_tail"); hipLaunchKernelGGL(k_nl_check_tail, dim3(1024), dim3(256), 0, ctx->stream, nc->dev, trace, capacity, n_rows, d_hist, d_res, (u64)NL_USED_ROWS(&S, capacity), (u64)nlq_used_rows(&S, qd, capacity), e_begin, e_end,
                                                        (u64)nlcf_first_row(circuit_type, &S, capacity), (u64)nlcf_used_rows(circuit_type, &S, capacity)); }
    ZKW_TRY(launch_check("k_nl_check_tail"));
    if (const nlcf_desc* cd = nlcf_desc_of(circuit_type)) {  // the closed-form section: its own checker (its lookup cells: the tail kernel)
        u32 ties = 0;
        for (u32 gi = 0; gi < cd->n_groups; gi++) ties += cd->g[gi].count;
        const u32 tie_blocks = (ties + 255) / 256, perm_blocks = (nlcf_n_perms(cd) + 15) / 16;
        const nlq_desc none{};
        { Prof _p(ctx, "k_nlcf_check"); hipLaunchKernelGGL(k_nlcf_check, dim3(tie_blocks + perm_blocks + 1), dim3(256), 0, ctx->stream, *cd, qd ? *qd : none, nc->dev, trace, capacity, n_rows,
                                                         (u64)nlcf_first_row(circuit_type, &S, capacity), tie_blocks, perm_blocks, d_res); }
        ZKW_TRY(launch_check("k_nlcf_check"));
    }
    if (qd) {
        { Prof _p(ctx, "k_nlq_check"); hipLaunchKernelGGL(k_nlq_check, dim3((capacity + 63) / 64, qd->n_ops), dim3(64), 0, ctx->stream, nc->dev, nc->free_home, *qd, *nlq_rels_of(circuit_type), trace, capacity, n_rows, d_res); }
        ZKW_TRY(launch_check("k_nlq_check"));
    }
    CheckResult res;
    ZKW_TRY(ctx->read_small(&res, d_res, sizeof res));
    *n_violations = res.violations;
    if (first_bad) *first_bad = res.violations ? res.first_bad : 0;
    return ZKW_OK;
}

// instance i of a precompile-style witness covers the rounds [i * capacity, min((i + 1) * capacity, total)) (none for the dummy instance)
std::vector<NlInstance> nl_instances(size_t first_instance, size_t n_instances, u32 capacity, bool any, u64 total_rounds, const u64* cf_pi, size_t n_all,
                                     zkw_trace* t, size_t first_slot) {
    std::vector<NlInstance> v(n_instances);
    for (size_t k = 0; k < n_instances; k++) {
        const size_t i = first_instance + k;
        v[k].first_round = (u64)i * capacity;
        v[k].n_active = any ? (u32)std::min<u64>(capacity, total_rounds - v[k].first_round) : 0;
        v[k].public_input = cf_pi + COMPACT_FORM_LEN * n_all + 4 * i;
        v[k].t = t;
        v[k].slot = (first_slot + k) % t->n_slots;
    }
    return v;
}
}  // namespace

// ZkSyncBaseLayerCircuit::synthesis for Keccak256RoundFunction (type 5): 86 + 3 x 14 columns, Xor8 / And8 / ByteSplit<1..4>
extern "C" int zkw_keccak_round_synthesize(zkw_ctx* ctx, zkw_precompile_witness* w, size_t first_instance, size_t n_instances,
                                           zkw_trace* t, size_t first_slot) {
    if (!ctx || !w || !t || w->ctx != ctx || t->ctx->device != ctx->device) return fail(ZKW_ERR_INVALID, "zkw_keccak_round_synthesize: bad argument");
    if (w->kind != ZKW_PRECOMPILE_KECCAK256) return fail(ZKW_ERR_INVALID, "zkw_keccak_round_synthesize: not a keccak256 witness");
    if (first_instance + n_instances > w->n_instances) return fail(ZKW_ERR_INVALID, "instance range out of bounds");
    if (n_instances > t->n_slots) return fail(ZKW_ERR_INVALID, "more instances (%zu) than trace slots (%zu)", n_instances, t->n_slots);
    if (t->n_cols < KC_COLS) return fail(ZKW_ERR_INVALID, "trace has %zu columns, the Keccak256RoundFunction circuit needs %d (zkw_trace_create_with_columns)", t->n_cols, KC_COLS);
    if (n_instances == 0) return ZKW_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    ZKW_TRY(zkw_precompile_closed_forms(ctx, w, nullptr, nullptr));  // public inputs of the block's instances (a20), once
    const std::vector<NlInstance> inst = nl_instances(first_instance, n_instances, w->capacity, w->n_requests != 0, w->total_rounds, w->cf_pi, w->n_instances, t, first_slot);
    NlcfCall cf;  // the closed-form section: sponges next to the fills, the ties after them
    SlotClaims claims;  // the slots' tags: committed after the closed-form section's last launch
    ZKW_TRY(nl_synthesize(ctx, 5, false, w->keccak_rounds, inst, w->capacity, t->n_rows, [&] {
        return nlcf_begin<CfPrecompile<ZKW_PRECOMPILE_KECCAK256>>(ctx, 5, w->instances, first_instance, inst, w->capacity, t->n_rows, &cf);
    }, &claims));
    NlqQueues Q{};  // the precompile calls are popped, the memory queries (unaligned reads, the digest write) pushed
    Q.q[0] = NlqQueueIn{w->requests, w->req_tails, {0}, w->n_requests};
    Q.q[1] = NlqQueueIn{w->mem_q, w->mem_tails, {0}, w->n_queries};
    memcpy(Q.q[1].init, w->mem_in.tail, sizeof w->mem_in.tail);
    Q.round_ops = w->round_ops;
    ZKW_TRY(nlq_synthesize(ctx, 5, Q, inst, w->capacity, t->n_rows));
    return claims.commit_if(nlcf_end(ctx, 5, cf, w->capacity, t->n_rows));
}
extern "C" int zkw_keccak_round_check_satisfied(zkw_ctx* ctx, const zkw_trace* t, size_t slot, uint32_t capacity, uint64_t* n_violations, uint64_t* first_bad) {
    if (!ctx || !t || t->ctx->device != ctx->device || slot >= t->n_slots || !n_violations || capacity == 0)
        return fail(ZKW_ERR_INVALID, "zkw_keccak_round_check_satisfied: bad argument");
    return nl_check(ctx, 5, t, slot, capacity, n_violations, first_bad);
}

// ZkSyncBaseLayerCircuit::synthesis for ECRecover (type 7): 80 + 3 x 16 columns, Xor8 / And8 / 8 x 32 FixedBaseMul / ByteSplit<1..4> =
// 197 632 table rows (base_layer/ecrecover.rs:30-41,138-176). One cycle per request (ecrecover.rs:143-178: four reads, two writes):
// the EC section recovers the key from the read values, the netlist hashes it, the queue section pops the call and pushes the queries.
// The instances [first[k], first[k] + count[k]) of witness ws[k], k = 0 .. n_ws - 1, into consecutive slots from first_slot — ONE launch of
// every EC kernel over all of them. The accumulator chain of a request is one wave and ~2.5 ms whatever the batch (k_ec_chain; round 5: one lane, 13 ms), so the
// instances of MANY blocks in one call cost what one block's cost (zkw_blocks_synthesize); the queue sections are per witness (their queues are).
// The witnesses may belong to other contexts of the same device: their builders must have finished (zkw_block_run returns after them).
static int ecrecover_synthesize_many(zkw_ctx* ctx, zkw_precompile_witness* const* ws, const size_t* first, const size_t* count, size_t n_ws, zkw_trace* t, size_t first_slot) {
    if (n_ws == 0) return ZKW_OK;
    const u32 capacity = ws[0]->capacity;
    const size_t n_rows = t->n_rows;
    size_t total = 0;
    for (size_t k = 0; k < n_ws; k++) {
        zkw_precompile_witness* w = ws[k];
        if (!w || w->kind != ZKW_PRECOMPILE_ECRECOVER) return fail(ZKW_ERR_INVALID, "zkw_ecrecover_synthesize: not an ecrecover witness");
        if (w->ctx->device != ctx->device || w->capacity != capacity) return fail(ZKW_ERR_INVALID, "zkw_ecrecover_synthesize: witnesses of another device or capacity");
        if (first[k] + count[k] > w->n_instances) return fail(ZKW_ERR_INVALID, "instance range out of bounds");
        total += count[k];
    }
    if (total > t->n_slots) return fail(ZKW_ERR_INVALID, "more instances (%zu) than trace slots (%zu)", total, t->n_slots);
    if (t->n_cols < EK_COLS) return fail(ZKW_ERR_INVALID, "trace has %zu columns, the ECRecover circuit needs %d (zkw_trace_create_with_columns)", t->n_cols, EK_COLS);
    if (total == 0) return ZKW_OK;
    if (ec_used_rows(capacity) > n_rows) return fail(ZKW_ERR_INVALID, "capacity %u needs %zu rows, trace has %zu", capacity, ec_used_rows(capacity), n_rows);
    HIP_TRY(hipSetDevice(ctx->device));
    const EcCached* ec = nullptr;
    ZKW_TRY(ec_get(ctx, &ec));
    std::vector<NlInstance> inst;
    std::vector<size_t> w_of, start_of(n_ws + 1, 0);  // instance -> witness; witness -> its first instance of the call
    for (size_t k = 0; k < n_ws; k++) {
        zkw_precompile_witness* w = ws[k];
        ZKW_TRY(precompile_closed_forms_with(ctx, w, nullptr, nullptr));  // on THIS call's context: the witness's own may be busy on another thread
        const std::vector<NlInstance> one = nl_instances(first[k], count[k], capacity, w->n_requests != 0, w->total_rounds, w->cf_pi, w->n_instances, t, first_slot + inst.size());
        inst.insert(inst.end(), one.begin(), one.end());
        w_of.insert(w_of.end(), one.size(), k);
        start_of[k + 1] = inst.size();
    }
    const size_t ni = inst.size();
    u64* d_tape = nullptr;
    u32* d_status = nullptr;
    uint8_t* d_inputs = nullptr;
    const size_t tape_per_instance = (size_t)EC_TAPE_PER_CYCLE * ec_tape_stride(capacity);  // the cycles of an instance interleaved (ecrecover_kernels.cuh)
    ZKW_TRY(ctx->scratch_t<u64>("ec_tape", ni * tape_per_instance, &d_tape));
    ZKW_TRY(ctx->scratch_t<uint8_t>("ec_inputs", ni * capacity * (size_t)128, &d_inputs));
    ZKW_TRY(ctx->scratch_t<u32>("ec_status", 1, &d_status));
    HIP_TRY(ctx->memset_async(d_status, 0, 4));
    std::vector<EcJob> jobs(ni);
    EcJob* d_jobs = nullptr;
    const unsigned cb = (capacity + 1 + 63) / 64, nj = (unsigned)ni;
    // the serial form (one lane walks a whole cycle, an inversion per quotient) is kept for cross-checks: ZKW_EC_SERIAL=1, same tape
    static const bool serial = [] { const char* e = getenv("ZKW_EC_SERIAL"); return e && atoi(e) != 0; }();
    const u32 n_cycles = (u32)(ni * capacity);
    const unsigned chunks = (n_cycles + EC_TAPE_LANES - 1) / EC_TAPE_LANES;
    const dim3 stream_grid((EC_ROWS_PER_CYCLE + EC_STREAM_ROWS - 1) / EC_STREAM_ROWS, (capacity + 7) / 8, nj);
    uint16_t* d_keys = nullptr;  // the Xor8 lookups' keys: [instance][cycle][Xor8 row of the cycle][16]
    ZKW_TRY(ctx->scratch_t<uint16_t>("ec_xor_keys", ni * capacity * (size_t)ec->stream.n_xor_rows * EC_R, &d_keys));
    bool forked = false;
    // the netlist's inputs come from the EC tapes: evaluate them inside the engine's `prepare` step
    SlotClaims claims;  // the slots' tags: committed after the call's last launch
    ZKW_TRY(nl_synthesize_with(ctx, 7, [&](std::vector<NlPrepJob>& prep) -> int {
        for (size_t k = 0; k < ni; k++)
            jobs[k] = EcJob{ws[w_of[k]]->mem_q, inst[k].first_round, inst[k].n_active, d_inputs + k * capacity * (size_t)128, d_tape + k * tape_per_instance,
                            inst[k].t->data + inst[k].slot * inst[k].t->slot_elems(), prep[k].hdr_bits, prep[k].free_elems, prep[k].state_before};
        ZKW_TRY(ctx->upload("ec_jobs", jobs, &d_jobs));
        { Prof _p(ctx, "k_ec_inputs"); ZKW_LAUNCH_2D(ctx, k_ec_inputs, capacity, nj, 128, d_jobs); }
        ZKW_TRY(launch_check("k_ec_inputs"));
        if (serial) {
            { Prof _p(ctx, "k_ec_tape"); ZKW_LAUNCH_2D(ctx, k_ec_tape, (capacity + EC_TAPE_LANES - 1) / EC_TAPE_LANES, nj, EC_TAPE_LANES, ec->dev, d_jobs, capacity, d_status); }
            ZKW_TRY(launch_check("k_ec_tape"));
        } else {
            EcChainScratch sc{};
            ZKW_TRY(ctx->scratch_t<ec_jac>("ec_chain_pts", ni * capacity * (size_t)EC_CHAIN_POINTS, &sc.pts));
            { Prof _p(ctx, "k_ec_chain"); ZKW_LAUNCH_2D(ctx, k_ec_chain, capacity, nj, 64, ec->dev, d_jobs, capacity, d_status, sc); }
            ZKW_TRY(launch_check("k_ec_chain"));
            { Prof _p(ctx, "k_ec_affine"); ZKW_LAUNCH(ctx, k_ec_affine, ((size_t)n_cycles * EC_CHAIN_POINTS + 63) / 64, 64, ec->dev, d_jobs, capacity, n_cycles, d_status, sc); }
            ZKW_TRY(launch_check("k_ec_affine"));
            { Prof _p(ctx, "k_ec_segments_main"); ZKW_LAUNCH_2D(ctx, k_ec_segments, ec->n_main, chunks, EC_TAPE_LANES, ec->dev, d_jobs, capacity, n_cycles, ec->tasks_main, d_status); }
            ZKW_TRY(launch_check("k_ec_segments"));
        }
        // the netlist's inputs need the globals (PRE's MAIN: the chain kernel) and the state POST leaves (its MAIN) only
        { Prof _p(ctx, "k_ec_prepare"); ZKW_LAUNCH_2D(ctx, k_ec_prepare, cb, nj, 64, ec->dev, d_jobs, capacity); }
        ZKW_TRY(launch_check("k_ec_prepare"));
        if (!serial) {
            // the rest of the EC section — MUL rows, leaves, the rows themselves — beside the netlist's fill, histogram and queue section:
            // on the side stream (a context of a batch has one stream: behind the fills there)
            hipStream_t st = ctx->stream;
            if (!ctx->batched()) { ZKW_TRY(ctx->side_fork(&st)); forked = true; }
            if (forked) {
                Launcher<&k_ec_segments, EC_TAPE_LANES>::S::template single<&k_ec_segments, EC_TAPE_LANES>(st, dim3(ec->n_muls, chunks), 0, ec->dev, (const EcJob*)d_jobs, capacity, n_cycles, ec->tasks_muls, d_status);
                ZKW_TRY(launch_check("k_ec_segments"));
                Launcher<&k_ec_leaves, EC_TAPE_LANES>::S::template single<&k_ec_leaves, EC_TAPE_LANES>(st, dim3(ec->n_leaves, chunks), 0, ec->dev, (const EcJob*)d_jobs, capacity, n_cycles, ec->tasks_leaves, d_status);
                ZKW_TRY(launch_check("k_ec_leaves"));
                Launcher<&k_ec_stream, EC_STREAM_THREADS>::S::template single<&k_ec_stream, EC_STREAM_THREADS>(st, stream_grid, 0, ec->stream, (const EcJob*)d_jobs, capacity, n_rows, ec_first_row(capacity), d_keys);
                ZKW_TRY(launch_check("k_ec_stream"));
            } else {
                { Prof _p(ctx, "k_ec_segments_muls"); ZKW_LAUNCH_2D(ctx, k_ec_segments, ec->n_muls, chunks, EC_TAPE_LANES, ec->dev, d_jobs, capacity, n_cycles, ec->tasks_muls, d_status); }
                ZKW_TRY(launch_check("k_ec_segments"));
                { Prof _p(ctx, "k_ec_leaves"); ZKW_LAUNCH_2D(ctx, k_ec_leaves, ec->n_leaves, chunks, EC_TAPE_LANES, ec->dev, d_jobs, capacity, n_cycles, ec->tasks_leaves, d_status); }
                ZKW_TRY(launch_check("k_ec_leaves"));
            }
        }
        return ZKW_OK;
    }, inst, capacity, n_rows, {}, &claims));
    for (size_t k = 0; k < n_ws; k++) {  // the queue sections: per witness (its request and memory queues)
        if (start_of[k + 1] == start_of[k]) continue;
        zkw_precompile_witness* w = ws[k];
        NlqQueues Q{};
        Q.q[0] = NlqQueueIn{w->requests, w->req_tails, {0}, w->n_requests};
        Q.q[1] = NlqQueueIn{w->mem_q, w->mem_tails, {0}, w->n_queries};
        memcpy(Q.q[1].init, w->mem_in.tail, sizeof w->mem_in.tail);
        Q.round_ops = w->round_ops;
        const std::vector<NlInstance> sub(inst.begin() + start_of[k], inst.begin() + start_of[k + 1]);
        ZKW_TRY(nlq_synthesize(ctx, 7, Q, sub, capacity, n_rows));
        NlcfCall cf;  // the closed-form section (a 34-word FSM: eight dependent permutations) on the main stream: the side stream is the EC section's
        char name[32];
        snprintf(name, sizeof name, "nlcf_jobs_%zu", k % 64);
        ZKW_TRY((nlcf_begin<CfPrecompile<ZKW_PRECOMPILE_ECRECOVER>>(ctx, 7, w->instances, first[k], sub, capacity, n_rows, &cf, name, true)));
        ZKW_TRY(nlcf_end(ctx, 7, cf, capacity, n_rows));
    }
    if (forked) ZKW_TRY(ctx->side_join());
    else { Prof _p(ctx, "k_ec_stream"); ZKW_LAUNCH_D(ctx, (k_ec_stream), "k_ec_stream", stream_grid, EC_STREAM_THREADS, 0, ec->stream, d_jobs, capacity, n_rows, ec_first_row(capacity), d_keys); }
    // the Xor8 multiplicities onto the column the netlist's finish kernel has written
    { Prof _p(ctx, "k_ec_hist"); ZKW_LAUNCH_2D(ctx, k_ec_hist, 2, nj, EC_HIST_THREADS, ec->stream, d_jobs, capacity, n_rows, ec_first_row(capacity), (u32)EK_MULT_COL, d_keys); }
    ZKW_TRY(launch_check("k_ec_hist"));
    u32 status = 0;
    ZKW_TRY(ctx->read_small(&status, d_status, 4));
    if (status) return fail(ZKW_ERR_CHECK_FAILED, "zkw_ecrecover_synthesize: request %llu (instance %u of the call) has no witness (the accumulator of the incomplete addition met x1 == x2)",
                            (unsigned long long)(inst[(status - 1) >> 16].first_round + ((status - 1) & 0xFFFF)), (unsigned)((status - 1) >> 16));
    return claims.commit_if(ZKW_OK);
}

extern "C" int zkw_ecrecover_synthesize(zkw_ctx* ctx, zkw_precompile_witness* w, size_t first_instance, size_t n_instances, zkw_trace* t, size_t first_slot) {
    if (!ctx || !w || !t || w->ctx != ctx || t->ctx->device != ctx->device) return fail(ZKW_ERR_INVALID, "zkw_ecrecover_synthesize: bad argument");
    return ecrecover_synthesize_many(ctx, &w, &first_instance, &n_instances, 1, t, first_slot);
}
// every instance of every witness (one per block), in order, into slots first_slot ..: see ecrecover_synthesize_many
extern "C" int zkw_ecrecover_synthesize_multi(zkw_ctx* ctx, zkw_precompile_witness* const* witnesses, size_t n_witnesses, zkw_trace* t, size_t first_slot) {
    if (!ctx || !t || (n_witnesses && !witnesses) || t->ctx->device != ctx->device) return fail(ZKW_ERR_INVALID, "zkw_ecrecover_synthesize_multi: bad argument");
    std::vector<size_t> first(n_witnesses, 0), count(n_witnesses, 0);
    for (size_t k = 0; k < n_witnesses; k++) {
        if (!witnesses[k]) return fail(ZKW_ERR_INVALID, "zkw_ecrecover_synthesize_multi: null witness");
        count[k] = witnesses[k]->n_instances;
    }
    return ecrecover_synthesize_many(ctx, witnesses, first.data(), count.data(), n_witnesses, t, first_slot);
}
extern "C" int zkw_ecrecover_check_satisfied(zkw_ctx* ctx, const zkw_trace* t, size_t slot, uint32_t capacity, uint64_t* n_violations, uint64_t* first_bad) {
    if (!ctx || !t || t->ctx->device != ctx->device || slot >= t->n_slots || !n_violations || capacity == 0)
        return fail(ZKW_ERR_INVALID, "zkw_ecrecover_check_satisfied: bad argument");
    return nl_check(ctx, 7, t, slot, capacity, n_violations, first_bad);
}

// ZkSyncBaseLayerCircuit::synthesis for Sha256RoundFunction (type 6): 116 + 4 x 9 columns, TriXor4 / Ch4 / Maj4 / Split4BitChunk<1, 2>
extern "C" int zkw_sha256_round_synthesize(zkw_ctx* ctx, zkw_precompile_witness* w, size_t first_instance, size_t n_instances,
                                           zkw_trace* t, size_t first_slot) {
    if (!ctx || !w || !t || w->ctx != ctx || t->ctx->device != ctx->device) return fail(ZKW_ERR_INVALID, "zkw_sha256_round_synthesize: bad argument");
    if (w->kind != ZKW_PRECOMPILE_SHA256) return fail(ZKW_ERR_INVALID, "zkw_sha256_round_synthesize: not a sha256 witness");
    if (first_instance + n_instances > w->n_instances) return fail(ZKW_ERR_INVALID, "instance range out of bounds");
    if (n_instances > t->n_slots) return fail(ZKW_ERR_INVALID, "more instances (%zu) than trace slots (%zu)", n_instances, t->n_slots);
    if (t->n_cols < SC_COLS) return fail(ZKW_ERR_INVALID, "trace has %zu columns, the Sha256RoundFunction circuit needs %d (zkw_trace_create_with_columns)", t->n_cols, SC_COLS);
    if (n_instances == 0) return ZKW_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    ZKW_TRY(zkw_precompile_closed_forms(ctx, w, nullptr, nullptr));
    const std::vector<NlInstance> inst = nl_instances(first_instance, n_instances, w->capacity, w->n_requests != 0, w->total_rounds, w->cf_pi, w->n_instances, t, first_slot);
    NlcfCall cf;
    SlotClaims claims;  // the slots' tags: committed after the closed-form section's last launch
    ZKW_TRY(nl_synthesize(ctx, 6, true, w->sha256_rounds, inst, w->capacity, t->n_rows, [&] {
        return nlcf_begin<CfPrecompile<ZKW_PRECOMPILE_SHA256>>(ctx, 6, w->instances, first_instance, inst, w->capacity, t->n_rows, &cf);
    }, &claims));
    NlqQueues Q{};  // the precompile calls are popped (the head runs through the states their pushes left), the memory queries pushed
    Q.q[0] = NlqQueueIn{w->requests, w->req_tails, {0}, w->n_requests};
    Q.q[1] = NlqQueueIn{w->mem_q, w->mem_tails, {0}, w->n_queries};
    memcpy(Q.q[1].init, w->mem_in.tail, sizeof w->mem_in.tail);
    Q.round_ops = w->round_ops;
    ZKW_TRY(nlq_synthesize(ctx, 6, Q, inst, w->capacity, t->n_rows));
    return claims.commit_if(nlcf_end(ctx, 6, cf, w->capacity, t->n_rows));
}
extern "C" int zkw_sha256_round_check_satisfied(zkw_ctx* ctx, const zkw_trace* t, size_t slot, uint32_t capacity, uint64_t* n_violations, uint64_t* first_bad) {
    if (!ctx || !t || t->ctx->device != ctx->device || slot >= t->n_slots || !n_violations || capacity == 0)
        return fail(ZKW_ERR_INVALID, "zkw_sha256_round_check_satisfied: bad argument");
    return nl_check(ctx, 6, t, slot, capacity, n_violations, first_bad);
}

// ------------------------------------------------------------------------------------------------ CodeDecommitter synthesis
// ZkSyncBaseLayerCircuit::synthesis for CodeDecommitter (type 3): the SHA-256 netlist on 108 + 4 x 11 columns, one cycle per round
// of the unpacked bytecodes (a cycle is one round: BeginNew shares the cycle of a bytecode's first round)
extern "C" int zkw_code_decommitter_synthesize(zkw_ctx* ctx, zkw_decommitter_witness* w, size_t first_instance, size_t n_instances,
                                               zkw_trace* t, size_t first_slot) {
    if (!ctx || !w || !t || w->ctx != ctx || t->ctx->device != ctx->device) return fail(ZKW_ERR_INVALID, "zkw_code_decommitter_synthesize: bad argument");
    if (first_instance + n_instances > w->n_instances) return fail(ZKW_ERR_INVALID, "instance range out of bounds");
    if (n_instances > t->n_slots) return fail(ZKW_ERR_INVALID, "more instances (%zu) than trace slots (%zu)", n_instances, t->n_slots);
    if (t->n_cols < DC_COLS) return fail(ZKW_ERR_INVALID, "trace has %zu columns, the CodeDecommitter circuit needs %d (zkw_trace_create_with_columns)", t->n_cols, DC_COLS);
    if (n_instances == 0) return ZKW_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    if (!w->cf_pi) ZKW_TRY(closed_form_public_inputs<CfDecommitter>(ctx, w->instances, w->n_instances, &w->cf_pi));
    const std::vector<NlInstance> inst = nl_instances(first_instance, n_instances, w->capacity, true, w->total_rounds, w->cf_pi, w->n_instances, t, first_slot);
    NlcfCall cf;
    SlotClaims claims;  // the slots' tags: committed after the closed-form section's last launch
    ZKW_TRY(nl_synthesize(ctx, 3, true, w->sha256_rounds, inst, w->capacity, t->n_rows, [&] {
        return nlcf_begin<CfDecommitter>(ctx, 3, w->instances, first_instance, inst, w->capacity, t->n_rows, &cf);
    }, &claims));
    NlqQueues Q{};  // the decommit requests are popped, the code words written to memory
    Q.q[0] = NlqQueueIn{w->requests, w->dedup_tails, {0}, w->n_requests};
    Q.q[1] = NlqQueueIn{w->mem_q, w->mem_tails, {0}, w->total_words};
    memcpy(Q.q[1].init, w->mem_in.tail, sizeof w->mem_in.tail);
    Q.round_ops = w->round_ops;
    ZKW_TRY(nlq_synthesize(ctx, 3, Q, inst, w->capacity, t->n_rows));
    return claims.commit_if(nlcf_end(ctx, 3, cf, w->capacity, t->n_rows));
}
extern "C" int zkw_code_decommitter_check_satisfied(zkw_ctx* ctx, const zkw_trace* t, size_t slot, uint32_t capacity, uint64_t* n_violations, uint64_t* first_bad) {
    if (!ctx || !t || t->ctx->device != ctx->device || slot >= t->n_slots || !n_violations || capacity == 0)
        return fail(ZKW_ERR_INVALID, "zkw_code_decommitter_check_satisfied: bad argument");
    return nl_check(ctx, 3, t, slot, capacity, n_violations, first_bad);
}

// ------------------------------------------------------------------------------------------------ StorageApplication synthesis
// ZkSyncBaseLayerCircuit::synthesis for StorageApplication (type 10): the Merkle walks of the instance's tree queries as Blake2s
// compressions on 60 + 3 x 26 columns, Xor8 / And8 / ByteSplit<1..4, 7> (base_layer/storage_apply.rs:28-39,124-140). A walk is 257
// cycles (leaf hash + 256 levels); capacity = cycles_per_storage_application walks (a read is one walk, a write two).
extern "C" int zkw_storage_application_synthesize(zkw_ctx* ctx, zkw_storage_application_witness* w, size_t first_instance, size_t n_instances,
                                                  zkw_trace* t, size_t first_slot) {
    if (!ctx || !w || !t || w->ctx != ctx || t->ctx->device != ctx->device) return fail(ZKW_ERR_INVALID, "zkw_storage_application_synthesize: bad argument");
    if (first_instance + n_instances > w->n_instances) return fail(ZKW_ERR_INVALID, "instance range out of bounds");
    if (n_instances > t->n_slots) return fail(ZKW_ERR_INVALID, "more instances (%zu) than trace slots (%zu)", n_instances, t->n_slots);
    if (t->n_cols < SA_COLS) return fail(ZKW_ERR_INVALID, "trace has %zu columns, the StorageApplication circuit needs %d (zkw_trace_create_with_columns)", t->n_cols, SA_COLS);
    if (w->capacity > SAP_WALK_MAX) return fail(ZKW_ERR_INVALID, "capacity %u: at most %u walks per instance", w->capacity, SAP_WALK_MAX);
    if (n_instances == 0) return ZKW_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    if (!w->cf_pi) ZKW_TRY(closed_form_public_inputs<CfStorageApplication>(ctx, w->instances, w->n_instances, &w->cf_pi));
    std::vector<zkw_storage_application_instance> rec(n_instances);
    ZKW_TRY(ctx->read_small(rec.data(), w->instances + first_instance, n_instances * sizeof rec[0]));
    std::vector<NlInstance> inst(n_instances);
    for (size_t k = 0; k < n_instances; k++)
        inst[k] = NlInstance{rec[k].first_item, (u32)rec[k].num_items, w->cf_pi + COMPACT_FORM_LEN * w->n_instances + 4 * (first_instance + k), t, (first_slot + k) % t->n_slots, true};
    const u32 capacity = w->capacity;
    NlcfCall cf;
    SlotClaims claims;  // the slots' tags: committed after the closed-form section's last launch
    ZKW_TRY(nl_synthesize_with(ctx, 10, [&](std::vector<NlPrepJob>& prep) {
        std::vector<SapWalkJob> jobs(prep.size());
        for (size_t k = 0; k < prep.size(); k++)
            jobs[k] = SapWalkJob{w->items, w->keys, w->paths, w->walk_hashes, w->n ? prep[k].first_round : 0, w->n ? prep[k].n_active : 0, prep[k].hdr_bits, prep[k].free_elems, prep[k].state_before,
                                 reinterpret_cast<const uint8_t*>(w->instances + first_instance + k) + offsetof(zkw_storage_application_instance, hidden_fsm_output) +
                                     offsetof(zkw_storage_application_fsm, current_root_hash)};
        SapWalkJob* d_jobs = nullptr;
        ZKW_TRY(ctx->upload("sap_walk_jobs", jobs, &d_jobs));
        { Prof _p(ctx, "k_sap_walk_cycles"); ZKW_LAUNCH_2D(ctx, k_sap_walk_cycles, (capacity * SAP_WALK_CYCLES + 256) / 256, (unsigned)jobs.size(), 256, d_jobs, capacity); }
        return launch_check("k_sap_walk_cycles");
    }, inst, capacity * SA_CYCLES_PER_WALK, t->n_rows, [&] {
        return nlcf_begin<CfStorageApplication>(ctx, 10, w->instances, first_instance, inst, capacity * SA_CYCLES_PER_WALK, t->n_rows, &cf);
    }, &claims));
    return claims.commit_if(nlcf_end(ctx, 10, cf, capacity * SA_CYCLES_PER_WALK, t->n_rows));
}
extern "C" int zkw_storage_application_check_satisfied(zkw_ctx* ctx, const zkw_trace* t, size_t slot, uint32_t capacity, uint64_t* n_violations, uint64_t* first_bad) {
    if (!ctx || !t || t->ctx->device != ctx->device || slot >= t->n_slots || !n_violations || capacity == 0)
        return fail(ZKW_ERR_INVALID, "zkw_storage_application_check_satisfied: bad argument");
    return nl_check(ctx, 10, t, slot, capacity * SA_CYCLES_PER_WALK, n_violations, first_bad);
}

// LinearHasher (type 13): the Keccak-f netlist over the sponge of the serialized L2 -> L1 messages (compute_linear_keccak256,
// data_hasher_and_merklizer.rs:8-67; wrapper base_layer/linear_hasher.rs:28-138). One instance per block.
// message_tails: [total][4], the state of a queue after each of its messages was pushed (what the events sorter keeps as
// ZKW_EVT_RESULT_NEW_TAILS) — the heads the circuit's pops run through; NULL: hashed here from queue_states[b].head (one serial
// Poseidon2 chain per queue)
extern "C" int zkw_linear_hasher_synthesize_batch_with_tails(zkw_ctx* ctx, const zkw_log_query* messages, const uint64_t* message_offsets, size_t n_queues,
                                                  const zkw_queue_state4* queue_states, const uint64_t* message_tails, uint32_t capacity, zkw_trace* t, size_t first_slot,
                                                  zkw_linear_hasher_instance* records_out, uint64_t* public_inputs_out) {
    if (!ctx || !t || !message_offsets || !queue_states || !records_out || t->ctx->device != ctx->device || capacity == 0)
        return fail(ZKW_ERR_INVALID, "zkw_linear_hasher_synthesize_batch: bad argument");
    if (n_queues == 0) return ZKW_OK;
    if (first_slot + n_queues > t->n_slots) return fail(ZKW_ERR_INVALID, "slots [%zu, %zu) of a trace with %zu", first_slot, first_slot + n_queues, t->n_slots);
    if (t->n_cols < LH_COLS) return fail(ZKW_ERR_INVALID, "trace has %zu columns, the LinearHasher circuit needs %d", t->n_cols, LH_COLS);
    const size_t total = message_offsets[n_queues];
    if (message_offsets[0] != 0 || (total && !messages)) return fail(ZKW_ERR_INVALID, "zkw_linear_hasher_synthesize_batch: offsets must start at 0");
    std::vector<u64> moff(message_offsets, message_offsets + n_queues + 1), roff(n_queues + 1, 0);
    for (size_t b = 0; b < n_queues; b++) {
        if (moff[b + 1] < moff[b]) return fail(ZKW_ERR_INVALID, "zkw_linear_hasher_synthesize_batch: offsets decrease at %zu", b);
        const size_t n = moff[b + 1] - moff[b];
        if (n > capacity) return fail(ZKW_ERR_INVALID, "queue %zu: %zu messages, the circuit hashes at most %u", b, n, capacity);
        roff[b + 1] = roff[b] + n * 88 / 136 + 1;
    }
    const u32 cycles = ZKW_LINEAR_HASHER_CYCLES(capacity);
    const size_t n_rows = t->n_rows;
    HIP_TRY(hipSetDevice(ctx->device));
    const zkw_log_query* d_q = nullptr;
    ZKW_TRY(ctx->in("lh_q", messages, total, &d_q));
    zkw_keccak_round_record* d_rounds = nullptr;
    uint8_t* d_hash = nullptr;
    u64 *d_moff = nullptr, *d_roff = nullptr;
    ZKW_TRY(ctx->scratch_t<zkw_keccak_round_record>("lh_rounds", roff[n_queues], &d_rounds));
    ZKW_TRY(ctx->scratch_t<uint8_t>("lh_hash", 32 * n_queues, &d_hash));
    ZKW_TRY(ctx->upload("lh_moff", moff, &d_moff));
    ZKW_TRY(ctx->upload("lh_roff", roff, &d_roff));
    { Prof _p(ctx, "k_linear_blocks"); ZKW_LAUNCH(ctx, k_linear_blocks, blocks_for(roff[n_queues] * 18, 256), 256, d_q, d_moff, d_roff, (u32)n_queues, d_rounds); }
    ZKW_TRY(launch_check("k_linear_blocks"));
    { Prof _p(ctx, "k_linear_keccak256"); ZKW_LAUNCH(ctx, k_linear_keccak256, (unsigned)n_queues, 64, d_q, (size_t)0, d_hash, d_rounds, d_moff, d_roff, true); }
    ZKW_TRY(launch_check("k_linear_keccak256"));
    std::vector<zkw_linear_hasher_instance> recv(n_queues);
    std::vector<uint8_t> hashes(32 * n_queues);
    ZKW_TRY(ctx->read_small(hashes.data(), d_hash, hashes.size()));
    for (size_t b = 0; b < n_queues; b++) {
        memset(&recv[b], 0, sizeof recv[b]);
        recv[b].start_flag = recv[b].completion_flag = 1;
        recv[b].queue_state = queue_states[b];
        memcpy(recv[b].keccak256_hash, &hashes[32 * b], 32);
    }
    zkw_linear_hasher_instance* d_rec = nullptr;
    ZKW_TRY(ctx->upload("lh_record", recv, &d_rec));
    u64 *d_cf = nullptr, *d_pi = nullptr;
    ZKW_TRY(ctx->scratch_t<u64>("lh_cf", COMPACT_FORM_LEN * n_queues, &d_cf));
    ZKW_TRY(ctx->scratch_t<u64>("lh_pi", 4 * n_queues, &d_pi));
    {
        Prof _p(ctx, "k_closed_form_commitments");
        ZKW_LAUNCH_T(ctx, (k_closed_form_commitments<CfLinearHasher>), "k_closed_form_commitments", (unsigned)n_queues, 64, d_rec, n_queues, d_cf);
    }
    ZKW_TRY(launch_check("k_closed_form_commitments"));
    { Prof _p(ctx, "k_commit_encodings"); ZKW_LAUNCH(ctx, k_commit_encodings, blocks_for(n_queues, 64), 64, d_cf, n_queues, (u32)COMPACT_FORM_LEN, d_pi); }
    ZKW_TRY(launch_check("k_commit_encodings"));
    std::vector<NlInstance> inst(n_queues);
    for (size_t b = 0; b < n_queues; b++) inst[b] = NlInstance{roff[b], (u32)(roff[b + 1] - roff[b]), d_pi + 4 * b, t, first_slot + b, true};
    NlcfCall cf;
    SlotClaims claims;  // the slots' tags: committed after the closed-form section's last launch
    ZKW_TRY(nl_synthesize(ctx, 13, false, d_rounds, inst, cycles, n_rows, [&] { return nlcf_begin<CfLinearHasher>(ctx, 13, d_rec, 0, inst, cycles, n_rows, &cf); }, &claims));
    {   // the queue section: every message is popped (Poseidon2 rows below the netlist, include/zkw_netlist_queue.h)
        const u64* d_tails = nullptr;
        if (total && message_tails) ZKW_TRY(ctx->in("lh_tails", message_tails, total * 4, &d_tails));
        else if (total) {
            u64 *d_enc = nullptr, *d_new = nullptr, *d_heads = nullptr;
            ZKW_TRY(ctx->scratch_t<u64>("lh_enc", total * 20, &d_enc));
            ZKW_TRY(ctx->scratch_t<u64>("lh_new_tails", total * 4, &d_new));
            std::vector<u64> heads(4 * n_queues);
            for (size_t b = 0; b < n_queues; b++) memcpy(&heads[4 * b], queue_states[b].head, 32);
            ZKW_TRY(ctx->upload("lh_heads", heads, &d_heads));
            { Prof _p(ctx, "k_encode_log"); ZKW_LAUNCH(ctx, k_encode_log, blocks_for(total, 256), 256, d_q, total, (const u32*)nullptr, d_enc); }
            ZKW_TRY(launch_check("k_encode_log"));
            std::vector<LogChainJob> chains;
            for (size_t b = 0; b < n_queues; b++)
                if (moff[b + 1] > moff[b]) chains.push_back(LogChainJob{d_enc + 20 * moff[b], nullptr, nullptr, d_new + 4 * moff[b], d_heads + 4 * b, moff[b + 1] - moff[b]});
            ZKW_TRY(dev_log_chains(ctx, d_enc, total, chains));
            d_tails = d_new;
        }
        std::vector<NlqQueues> per(n_queues);
        for (size_t b = 0; b < n_queues; b++) {
            per[b] = NlqQueues{};
            per[b].q[0] = NlqQueueIn{d_q ? d_q + moff[b] : nullptr, d_tails ? d_tails + 4 * moff[b] : nullptr, {0}, moff[b + 1] - moff[b]};
            memcpy(per[b].q[0].init, queue_states[b].head, 32);
        }
        ZKW_TRY(nlq_synthesize(ctx, 13, per[0], inst, cycles, n_rows, &per));
    }
    ZKW_TRY(claims.commit_if(nlcf_end(ctx, 13, cf, cycles, n_rows)));
    memcpy(records_out, recv.data(), n_queues * sizeof recv[0]);
    if (public_inputs_out) ZKW_TRY(ctx->read_small(public_inputs_out, d_pi, 32 * n_queues));
    return ZKW_OK;
}

extern "C" int zkw_linear_hasher_synthesize_batch(zkw_ctx* ctx, const zkw_log_query* messages, const uint64_t* message_offsets, size_t n_queues,
                                                  const zkw_queue_state4* queue_states, uint32_t capacity, zkw_trace* t, size_t first_slot,
                                                  zkw_linear_hasher_instance* records_out, uint64_t* public_inputs_out) {
    return zkw_linear_hasher_synthesize_batch_with_tails(ctx, messages, message_offsets, n_queues, queue_states, nullptr, capacity, t, first_slot, records_out, public_inputs_out);
}

extern "C" int zkw_linear_hasher_synthesize(zkw_ctx* ctx, const zkw_log_query* messages, size_t n, const zkw_queue_state4* queue_state,
                                            uint32_t capacity, zkw_trace* t, size_t slot, zkw_linear_hasher_instance* record_out,
                                            uint64_t* public_input_out) {
    if (!queue_state || !record_out) return fail(ZKW_ERR_INVALID, "zkw_linear_hasher_synthesize: bad argument");
    const uint64_t offsets[2] = {0, n};
    return zkw_linear_hasher_synthesize_batch(ctx, messages, offsets, 1, queue_state, capacity, t, slot, record_out, public_input_out);
}

extern "C" int zkw_linear_hasher_check_satisfied(zkw_ctx* ctx, const zkw_trace* t, size_t slot, uint32_t capacity, uint64_t* n_violations, uint64_t* first_bad) {
    if (!ctx || !t || t->ctx->device != ctx->device || slot >= t->n_slots || !n_violations || capacity == 0)
        return fail(ZKW_ERR_INVALID, "zkw_linear_hasher_check_satisfied: bad argument");
    return nl_check(ctx, 13, t, slot, ZKW_LINEAR_HASHER_CYCLES(capacity), n_violations, first_bad);
}

