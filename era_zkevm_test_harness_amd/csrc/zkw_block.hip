// zkw_block.hip — one block's witness generation as a dependency graph of builders (C++ host code; no kernels here).
//
// Counterpart of the post-VM half of `create_artifacts_from_tracer` (src/witness/oracle.rs:928-1130: the builders in
// order, the memory queue threaded VM -> code decommitter -> keccak256 -> sha256 -> ecrecover -> RAM permutation, the
// demuxed log queues feeding the sorters and the precompiles) and of `CircuitMaker::process / into_results`
// (src/witness/postprocessing/mod.rs:353-405: public inputs, one recursion queue per circuit type).
//
// Written against include/zkw.h only (a Rust host could do the same over the FFI). What it adds over calling the
// builders in the reference's order is scheduling: a Poseidon2 queue chain is serial (~10 us per item on one wave)
// while independent chains cost nothing extra, so
//   * every branch of the graph has its own zkw_ctx (HIP stream + scratch) and host thread,
//   * contents (sorts, routing, deduplication) are computed before anything is hashed,
//   * the block's memory queue is assembled up front and hashed once, by the RAM-permutation builder; the decommitter
//     and the precompile builders receive their slices of its states (zkw_*_build_with_tails).
// Wall time of the builders ~ the longest chain of the block instead of the sum over builders.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <future>
#include <thread>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/zkw.h"
#include "zkw_batch.h"     // K blocks as fibers of one thread (zkw_blocks_run)
#include "zkw_internal.h"  // a context inside a batch

namespace {

using Clock = std::chrono::steady_clock;

enum { T_VM = 1, T_DEC = 2, T_DCM = 3, T_DMX = 4, T_KEC = 5, T_SHA = 6, T_ECR = 7, T_RAM = 8, T_STO = 9, T_SAP = 10, T_EVT = 11, T_L1 = 12, T_HSH = 13 };
enum { C_DEC = 0, C_RAM, C_DMX, C_STO, C_EVT, C_L1, C_PRE, N_CTX };

struct Status {
    int rc = ZKW_OK;
    std::string msg;
    bool ok() const { return rc == ZKW_OK; }
};

Status from_rc(int rc) {
    Status s;
    s.rc = rc;
    if (rc != ZKW_OK) s.msg = zkw_last_error();
    return s;
}
Status hip_status(hipError_t e, const char* what) {
    Status s;
    if (e != hipSuccess) {
        s.rc = e == hipErrorOutOfMemory ? ZKW_ERR_OOM : ZKW_ERR_HIP;
        s.msg = std::string(what) + ": " + hipGetErrorString(e);
    }
    return s;
}
#define ST_TRY(expr)              \
    do {                          \
        Status _s = (expr);       \
        if (!_s.ok()) return _s;  \
    } while (0)
#define ST_ZKW(expr) ST_TRY(from_rc(expr))
#define ST_HIP(expr) ST_TRY(hip_status((expr), #expr))

struct Span {
    std::string name;
    double a, b;
};

struct PerType {
    std::vector<uint64_t> pi, enc, states, compact;  // [n][4], [n][8], [n][12], [n][18]
};

// One transfer lane per host thread of the graph: a private non-blocking stream + a grow-only pinned staging buffer,
// created before anything long runs. Copies through pageable memory (hipMemcpy, hipMemcpyAsync with a pageable side) and
// hipHostFree / hipStreamDestroy wait for EVERY stream of the device, i.e. for the other branches' queue chains
// (measured: +0.9 s per builder), so nothing of the kind happens while the graph runs.
enum { X_MAIN = 0, X_LOG, X_STO, X_EVT, X_L1, X_RAM, X_DEC, N_XFER };
struct Xfer {
    zkw_ctx* c = nullptr;  // any context of the block: names the device for the library's buffer / stream caches
    zkw_batch* batch = nullptr;  // the block runs as fibers of a batch: copies are queued with it (small) or go straight onto its stream (large)
    hipStream_t st = nullptr;
    void* pin = nullptr;
    size_t cap = 0;
    std::vector<void*> retired;
    Status reserve(size_t bytes) {
        if (cap >= bytes) return Status();
        if (pin) retired.push_back(pin);
        pin = nullptr;
        cap = 0;
        const size_t want = bytes + bytes / 4 + 4096;
        ST_ZKW(zkw_buffer_alloc(c, 1, want, &pin));
        cap = want;
        return Status();
    }
    Status d2h(void* dst, const void* src, size_t bytes) {
        if (!bytes) return Status();
        if (batch) {
            zkw_batch_copy_d2h(batch, dst, src, bytes);
            return from_rc(zkw_batch_sync(batch));
        }
        ST_TRY(reserve(bytes));
        ST_HIP(hipMemcpyAsync(pin, src, bytes, hipMemcpyDeviceToHost, st));
        ST_HIP(hipStreamSynchronize(st));
        memcpy(dst, pin, bytes);
        return Status();
    }
    Status h2d(void* dst, const void* src, size_t bytes) {
        if (!bytes) return Status();
        if (batch) {
            // small: captured into the batch's upload arena and copied on the device; large (a block's queues): straight onto the batch's
            // stream — the destination is a fresh buffer nothing queued before touches, and everything queued after this call runs after it
            if (bytes <= ((size_t)64 << 10)) zkw_batch_copy_h2d(batch, dst, src, bytes);
            else ST_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, zkw_batch_stream(batch)));
            return Status();
        }
        ST_TRY(reserve(bytes));
        memcpy(pin, src, bytes);
        ST_HIP(hipMemcpyAsync(dst, pin, bytes, hipMemcpyHostToDevice, st));
        ST_HIP(hipStreamSynchronize(st));
        return Status();
    }
    void destroy() {
        if (st) zkw_stream_release(c, st);
        st = nullptr;
        if (pin) zkw_buffer_free(1, pin);
        for (void* q : retired) zkw_buffer_free(1, q);
    }
};

}  // namespace

struct zkw_block {
    int device = 0;
    zkw_batch* batch = nullptr;  // set while the block's builders run as fibers of zkw_blocks_run's batch
    bool from_batch = false;     // its contexts have no stream of their own: they sit on the device's shared stream between calls
    zkw_ctx* ctx[N_CTX] = {};
    Xfer xf[N_XFER];
    uint32_t cap[14] = {};
    // The block's chains go through the device's chain service (zkw_set_chain_service): with zkw_blocks_run they travel in
    // launches shared with the other blocks; for a single block the point is that the service's high-priority streams have
    // hardware queues of their own — on the contexts' own streams a branch's chain can land on the hardware queue of another
    // branch (8 queues, ~14 streams) and the two 1 s chains of the critical path then run one after the other (seen: 2.08 s).
    bool use_chain_service = true;
    Clock::time_point t0;
    std::mutex mu;
    std::vector<Span> spans;
    // device buffers owned by the block
    std::vector<void*> dev;
    zkw_mem_query* d_all_mem = nullptr;
    size_t n_mem = 0;
    size_t mem_off[6] = {};  // VM | code words | keccak256 | sha256 | ecrecover | end
    // witnesses
    zkw_decommit_witness* dec = nullptr;
    zkw_decommitter_witness* dcm = nullptr;
    zkw_demux_witness* dmx = nullptr;
    zkw_precompile_witness* pre[3] = {};
    zkw_ram_witness* ram = nullptr;
    zkw_storage_witness* sto = nullptr;
    zkw_storage_application_witness* sap = nullptr;
    zkw_events_witness *evt = nullptr, *l1 = nullptr;
    uint64_t dmx_off[7] = {};
    zkw_queue_state12 mem_state;
    uint8_t l1_hash[32] = {};
    zkw_linear_hasher_instance linear_hasher = {};
    PerType per[14];
    std::vector<zkw_vm_instance> vm_instances;
    // synthesis ring (created by the first zkw_block_synthesize)
    zkw_trace* ring = nullptr;  // 153 columns: the widest of the synthesized types (a type with fewer uses the first of them)
    size_t ring_rows = 0, ring_slots = 0;

    double ms_now() const { return std::chrono::duration<double, std::milli>(Clock::now() - t0).count(); }
    void span(const char* name, double a) {
        const double b = ms_now();
        std::lock_guard<std::mutex> g(mu);
        spans.push_back(Span{name, a, b});
    }
    template <class T>
    Status alloc(T** p, size_t count) {
        void* q = nullptr;
        ST_ZKW(zkw_buffer_alloc(ctx[0], 0, count * sizeof(T) + 64, &q));
        {
            std::lock_guard<std::mutex> g(mu);
            dev.push_back(q);
        }
        *p = static_cast<T*>(q);
        return Status();
    }
    template <class T>
    Status upload(int lane, T** p, const T* host, size_t count) {
        ST_TRY(alloc(p, count ? count : 1));
        if (count) ST_TRY(xf[lane].h2d(*p, host, count * sizeof(T)));
        return Status();
    }
    // one of the block's input queues: copied to the device, or borrowed when the host has left it there (zkw_block_inputs::queues_on_device)
    template <class T>
    Status queue_in(int lane, const T** p, const T* src, size_t count, bool on_device) {
        if (on_device && count) { *p = src; return Status(); }
        T* d = nullptr;
        ST_TRY(upload(lane, &d, src, count));
        *p = d;
        return Status();
    }
};

namespace {

struct Timed {
    zkw_block* b;
    const char* name;
    double a;
    Timed(zkw_block* blk, const char* n) : b(blk), name(n), a(blk->ms_now()) {}
    ~Timed() { b->span(name, a); }
};

// A branch of the builder graph: a host thread (zkw_block_run) or — the block being one of a batch — a fiber of the batch's thread.
struct Branch {
    std::future<Status> fut;
    std::shared_ptr<Status> st;
    zkw_batch* batch = nullptr;
    int fiber = -1;
    Status get() {
        if (!batch) return fut.get();
        if (fiber < 0) return Status{ZKW_ERR_OOM, "zkw_batch: a builder branch could not be started"};
        (void)zkw_batch_join(batch, fiber);
        return *st;
    }
};
template <class F>
Branch start_branch(zkw_block* B, F f) {
    Branch b;
    if (B->batch) {
        b.batch = B->batch;
        b.st = std::make_shared<Status>();
        std::shared_ptr<Status> st = b.st;
        b.fiber = zkw_batch_spawn(B->batch, [st, f]() { *st = f(); return st->rc; });  // (what the caller has queued so far runs first)
    } else {
        b.fut = std::async(std::launch::async, f);
    }
    return b;
}

// ---- branch: log demuxer, then the three sorters (each on its own thread) --------------------------------------
Status storage_branch(zkw_block* B, const zkw_block_inputs* in, const zkw_log_query* d_q, size_t n) {
    ST_HIP(hipSetDevice(B->device));
    {
        Timed t(B, "storage_sorter");
        ST_ZKW(zkw_storage_sorter_build(B->ctx[C_STO], d_q, n, B->cap[T_STO], &B->sto));
        ST_ZKW(zkw_synchronize(B->ctx[C_STO]));
    }
    if (!in->storage_tree) return Status();
    Timed t(B, "storage_application");
    const size_t nr = zkw_storage_witness_num_results(B->sto);
    const zkw_log_query* d_rq = static_cast<const zkw_log_query*>(zkw_storage_witness_device_ptr(B->sto, ZKW_STO_RESULT_QUERIES));
    const uint64_t* d_rt = static_cast<const uint64_t*>(zkw_storage_witness_device_ptr(B->sto, ZKW_STO_RESULT_NEW_TAILS));
    std::vector<zkw_log_query> hq(nr);
    std::vector<uint64_t> idx(nr);
    std::vector<uint8_t> paths(nr * 256 * 32);
    uint64_t* d_idx = nullptr;
    uint8_t* d_paths = nullptr;
    if (nr) {
        ST_TRY(B->xf[X_STO].d2h(hq.data(), d_rq, nr * sizeof(zkw_log_query)));
        if (in->storage_tree(in->storage_tree_user, hq.data(), nr, idx.data(), paths.data()) != 0) {
            Status s;
            s.rc = ZKW_ERR_INVALID;
            s.msg = "the storage_tree callback failed";
            return s;
        }
        ST_TRY(B->upload(X_STO, &d_idx, idx.data(), nr));
        ST_TRY(B->upload(X_STO, &d_paths, paths.data(), paths.size()));
    }
    ST_ZKW(zkw_storage_application_build(B->ctx[C_STO], d_rq, d_rt, nr, d_idx, d_paths, in->storage_initial_root,
                                         in->storage_initial_next_enumeration_index, B->cap[T_SAP], &B->sap));
    ST_ZKW(zkw_synchronize(B->ctx[C_STO]));
    return Status();
}

Status events_branch(zkw_block* B, int which, const zkw_log_query* d_q, size_t n) {
    ST_HIP(hipSetDevice(B->device));
    zkw_ctx* c = B->ctx[which ? C_L1 : C_EVT];
    {
        Timed t(B, which ? "l1_messages_sorter" : "events_sorter");
        ST_ZKW(zkw_events_sorter_build(c, d_q, n, B->cap[which ? T_L1 : T_EVT], nullptr, which ? &B->l1 : &B->evt));
        ST_ZKW(zkw_synchronize(c));
    }
    if (which) {  // pubdata hash of the net L2 -> L1 messages (oracle.rs:1102-1112)
        Timed t(B, "l1_messages_hasher");
        uint8_t* d_hash = nullptr;
        ST_TRY(B->alloc(&d_hash, 32));
        const zkw_log_query* d_res = static_cast<const zkw_log_query*>(zkw_events_witness_device_ptr(B->l1, ZKW_EVT_RESULT_QUERIES));
        ST_ZKW(zkw_linear_keccak256(c, d_res, zkw_events_witness_num_results(B->l1), d_hash));
        ST_ZKW(zkw_synchronize(c));
        ST_TRY(B->xf[X_L1].d2h(B->l1_hash, d_hash, 32));
    }
    return Status();
}

Status log_branch(zkw_block* B, const zkw_block_inputs* in) {
    ST_HIP(hipSetDevice(B->device));
    const zkw_log_query* d_logs = nullptr;
    {
        Timed t(B, "log_demuxer");
        ST_TRY(B->queue_in(X_LOG, &d_logs, in->log_queries, in->n_log_queries, in->queues_on_device != 0));
        ST_ZKW(zkw_log_demux_build(B->ctx[C_DMX], d_logs, in->n_log_queries, B->cap[T_DMX], nullptr, &B->dmx));
        ST_ZKW(zkw_synchronize(B->ctx[C_DMX]));
        ST_TRY(B->xf[X_LOG].d2h(B->dmx_off, zkw_demux_witness_device_ptr(B->dmx, ZKW_DMX_OUT_OFFSETS), sizeof B->dmx_off));
    }
    const zkw_log_query* d_out = static_cast<const zkw_log_query*>(zkw_demux_witness_device_ptr(B->dmx, ZKW_DMX_OUT_QUERIES));
    auto q = [&](int k) { return d_out + B->dmx_off[k]; };
    auto nq = [&](int k) { return (size_t)(B->dmx_off[k + 1] - B->dmx_off[k]); };
    const zkw_log_query *q0 = q(0), *q1 = q(1), *q2 = q(2);
    const size_t n0 = nq(0), n1 = nq(1), n2 = nq(2);
    auto f_sto = start_branch(B, [=] { return storage_branch(B, in, q0, n0); });
    auto f_evt = start_branch(B, [=] { return events_branch(B, 0, q1, n1); });
    auto f_l1 = start_branch(B, [=] { return events_branch(B, 1, q2, n2); });
    Status s0 = f_sto.get(), s1 = f_evt.get(), s2 = f_l1.get();
    if (!s0.ok()) return s0;
    if (!s1.ok()) return s1;
    return s2;
}

// ---- branch: the RAM permutation over the whole memory queue (the one place the memory queue is hashed) ----------
Status ram_branch(zkw_block* B, const zkw_block_inputs* in, const uint64_t** d_tails) {
    ST_HIP(hipSetDevice(B->device));
    Timed t(B, "ram_permutation");
    ST_ZKW(zkw_ram_build_instances(B->ctx[C_RAM], B->d_all_mem, B->n_mem, B->cap[T_RAM], in->num_non_deterministic_heap_queries, &B->ram));
    *d_tails = static_cast<const uint64_t*>(zkw_ram_witness_device_ptr(B->ram, ZKW_RAM_UNSORTED_TAILS));
    if (!*d_tails) return from_rc(ZKW_ERR_OOM);
    ST_ZKW(zkw_synchronize(B->ctx[C_RAM]));
    return Status();
}

Status dec_finish_branch(zkw_block* B) {
    ST_HIP(hipSetDevice(B->device));
    Timed t(B, "decommit_sorter.finish");
    ST_ZKW(zkw_decommit_sorter_finish(B->ctx[C_DEC], B->dec));
    ST_ZKW(zkw_synchronize(B->ctx[C_DEC]));
    return Status();
}

std::string key32(const uint32_t* h) { return std::string(reinterpret_cast<const char*>(h), 32); }

Status run(zkw_block* B, const zkw_block_inputs* in) {
    ST_HIP(hipSetDevice(B->device));
    for (int i = 0; i < N_CTX; i++) {
        if (B->batch) {  // no stream, no chain service: what the context launches travels with the batch (device pointers)
            B->ctx[i] = zkw_ctx_create_in_batch(B->device, B->batch);
            continue;
        }
        B->ctx[i] = zkw_create(B->device);
        if (!B->ctx[i]) return from_rc(ZKW_ERR_NO_DEVICE);
        ST_ZKW(zkw_set_pointer_mode(B->ctx[i], ZKW_PTR_DEVICE));
        if (B->use_chain_service) ST_ZKW(zkw_set_chain_service(B->ctx[i], 1));
        ST_ZKW(zkw_set_chain_tag(B->ctx[i], i + 1));  // the chain service batches equal stages of equal branches of all blocks in flight
        if (getenv("ZKW_BLOCK_PROFILE")) ST_ZKW(zkw_profile_enable(B->ctx[i], 1));
    }
    for (int i = 0; i < N_XFER; i++) {
        void* st = nullptr;
        B->xf[i].c = B->ctx[0];
        B->xf[i].batch = B->batch;
        if (B->batch) continue;
        ST_ZKW(zkw_stream_acquire(B->ctx[0], &st));
        B->xf[i].st = static_cast<hipStream_t>(st);
        ST_TRY(B->xf[i].reserve(i == X_MAIN || i == X_LOG ? (size_t)8 << 20 : (size_t)1 << 20));
    }
    for (int t = 1; t <= 13; t++) {
        zkw_circuit_geometry g;
        ST_ZKW(zkw_circuit_geometry_of((uint8_t)t, &g));
        B->cap[t] = in->capacities[t] ? in->capacities[t] : g.capacity;
    }
    // the log branch does not depend on anything else: start it first
    auto f_log = start_branch(B, [=] { return log_branch(B, in); });

    // 1. decommit sorter, contents only: the deduplicated requests fix which code words enter the memory queue
    const zkw_decommit_query* d_dq = nullptr;
    std::vector<zkw_decommit_query> dedup;
    {
        Timed t(B, "decommit_sorter.prepare");
        { Timed t1(B, "decommit_sorter.prepare.upload"); ST_TRY(B->queue_in(X_MAIN, &d_dq, in->decommit_queries, in->n_decommit_queries, in->queues_on_device != 0)); }
        { Timed t2(B, "decommit_sorter.prepare.kernels"); ST_ZKW(zkw_decommit_sorter_prepare(B->ctx[C_DEC], d_dq, in->n_decommit_queries, B->cap[T_DEC], nullptr, &B->dec)); }
        dedup.resize(zkw_decommit_witness_num_dedup(B->dec));
        { Timed t3(B, "decommit_sorter.prepare.readback");
        ST_TRY(B->xf[X_MAIN].d2h(dedup.data(), zkw_decommit_witness_device_ptr(B->dec, ZKW_DEC_DEDUP_QUERIES),
                         dedup.size() * sizeof(zkw_decommit_query))); }
    }
    auto f_dec = start_branch(B, [=] { return dec_finish_branch(B); });  // its three chains run next to everything below

    // 2. the whole memory queue: VM | code words in the order of the deduplicated queue | keccak256 | sha256 | ecrecover
    std::vector<uint32_t> words;
    std::vector<uint64_t> woff(dedup.size() + 1, 0);
    uint32_t* d_words = nullptr;
    const uint64_t* d_tails = nullptr;
    {
        Timed t(B, "memory_queue.assemble");
        std::unordered_map<std::string, size_t> by_hash;
        for (size_t k = 0; k < in->n_bytecodes; k++) by_hash[key32(in->bytecode_hashes + 8 * k)] = k;
        for (size_t i = 0; i < dedup.size(); i++) {
            auto it = by_hash.find(key32(dedup[i].hash));
            if (it == by_hash.end()) {
                Status s;
                s.rc = ZKW_ERR_INVALID;
                s.msg = "a decommit request names a bytecode hash that is not among the block's bytecodes";
                return s;
            }
            const uint64_t lo = in->bytecode_word_offsets[it->second], hi = in->bytecode_word_offsets[it->second + 1];
            words.insert(words.end(), in->bytecode_words + 8 * lo, in->bytecode_words + 8 * hi);
            woff[i + 1] = woff[i] + (hi - lo);
        }
        B->mem_off[0] = 0;
        B->mem_off[1] = in->n_vm_memory_queries;
        B->mem_off[2] = B->mem_off[1] + woff.back();
        for (int k = 0; k < 3; k++) B->mem_off[3 + k] = B->mem_off[2 + k] + in->n_precompile_memory_queries[k];
        B->n_mem = B->mem_off[5];
        if (B->n_mem == 0) {
            Status s;
            s.rc = ZKW_ERR_INVALID;
            s.msg = "the block has no memory queries (ram_permutation.rs:43-46)";
            return s;
        }
        ST_TRY(B->alloc(&B->d_all_mem, B->n_mem));
        // the parts the host supplies: host -> device, or (queues_on_device) device -> device on the precompile context's stream, which
        // zkw_decommitter_memory_queries below continues on and the zkw_synchronize after it waits for
        auto part_in = [&](zkw_mem_query* dst, const zkw_mem_query* src, size_t n) -> Status {
            if (!n) return Status();
            if (in->queues_on_device) return from_rc(zkw_copy_device(B->ctx[C_PRE], dst, src, n * sizeof(zkw_mem_query)));
            return B->xf[X_MAIN].h2d(dst, src, n * sizeof(zkw_mem_query));
        };
        ST_TRY(part_in(B->d_all_mem, in->vm_memory_queries, in->n_vm_memory_queries));
        for (int k = 0; k < 3; k++) ST_TRY(part_in(B->d_all_mem + B->mem_off[2 + k], in->precompile_memory_queries[k], in->n_precompile_memory_queries[k]));
        ST_TRY(B->upload(X_MAIN, &d_words, words.data(), words.size()));
        const zkw_decommit_query* d_dedup = static_cast<const zkw_decommit_query*>(zkw_decommit_witness_device_ptr(B->dec, ZKW_DEC_DEDUP_QUERIES));
        // on the precompile context: the decommit context is busy hashing
        ST_ZKW(zkw_decommitter_memory_queries(B->ctx[C_PRE], d_dedup, dedup.size(), d_words, woff.data(), B->d_all_mem + B->mem_off[1]));
        ST_ZKW(zkw_synchronize(B->ctx[C_PRE]));
    }
    const uint64_t** pp_tails = &d_tails;
    auto f_ram = start_branch(B, [=] { return ram_branch(B, in, pp_tails); });

    Status s_ram = f_ram.get(), s_dec = f_dec.get(), s_log = f_log.get();
    ST_TRY(s_ram);
    ST_TRY(s_dec);
    ST_TRY(s_log);

    // 3. the builders that own a slice of the memory queue: states given, nothing left to hash
    auto mem_in_at = [&](size_t start, zkw_queue_state12* st) -> Status {
        memset(st, 0, sizeof *st);
        st->length = (uint32_t)start;
        if (start) ST_TRY(B->xf[X_MAIN].d2h(st->tail, d_tails + 12 * (start - 1), 96));
        return Status();
    };
    {
        Timed t(B, "code_decommitter");
        zkw_queue_state12 st;
        ST_TRY(mem_in_at(B->mem_off[1], &st));
        const zkw_decommit_query* d_dedup = static_cast<const zkw_decommit_query*>(zkw_decommit_witness_device_ptr(B->dec, ZKW_DEC_DEDUP_QUERIES));
        const uint64_t* d_dt = static_cast<const uint64_t*>(zkw_decommit_witness_device_ptr(B->dec, ZKW_DEC_DEDUP_TAILS));
        ST_ZKW(zkw_decommitter_build_with_tails(B->ctx[C_PRE], d_dedup, d_dt, dedup.size(), d_words, woff.data(), B->cap[T_DCM], &st,
                                                d_tails + 12 * B->mem_off[1], &B->dcm));
    }
    const zkw_log_query* d_out = static_cast<const zkw_log_query*>(zkw_demux_witness_device_ptr(B->dmx, ZKW_DMX_OUT_QUERIES));
    const uint64_t* d_out_tails = static_cast<const uint64_t*>(zkw_demux_witness_device_ptr(B->dmx, ZKW_DMX_OUT_NEW_TAILS));
    static const char* pre_names[3] = {"keccak256_round_function", "sha256_round_function", "ecrecover"};
    for (int k = 0; k < 3; k++) {
        Timed t(B, pre_names[k]);
        zkw_queue_state12 st;
        ST_TRY(mem_in_at(B->mem_off[2 + k], &st));
        const size_t nreq = (size_t)(B->dmx_off[4 + k] - B->dmx_off[3 + k]);
        const size_t nq = in->n_precompile_memory_queries[k];
        ST_ZKW(zkw_precompile_build_with_tails(B->ctx[C_PRE], k, nreq ? d_out + B->dmx_off[3 + k] : nullptr,
                                               nreq ? d_out_tails + 4 * B->dmx_off[3 + k] : nullptr, nreq,
                                               nq ? B->d_all_mem + B->mem_off[2 + k] : nullptr, nq, B->cap[T_KEC + k], &st,
                                               nq ? d_tails + 12 * B->mem_off[2 + k] : nullptr, &B->pre[k]));
    }
    ST_ZKW(zkw_synchronize(B->ctx[C_PRE]));
    ST_TRY(mem_in_at(B->n_mem, &B->mem_state));

    // 3b. MainVM instances (oracle.rs:1229-1469): the tracer's vectors cut at the snapshots, entry states from the queues
    //     hashed above (memory queue: the VM's prefix of the RAM permutation's unsorted states; decommit queue: the
    //     decommit sorter's unsorted states)
    if (in->vm_tracer) {
        Timed t(B, "main_vm.slicing");
        const zkw_vm_tracer_streams& v = *in->vm_tracer;
        if (v.stream_len[ZKW_VMS_MEMORY] != in->n_vm_memory_queries || v.n_decommit_states != in->n_decommit_queries || v.n_snapshots < 2) {
            Status s;
            s.rc = ZKW_ERR_INVALID;
            s.msg = "vm_tracer: the memory stream / decommit states must be as long as the block's VM memory queue / decommit queue";
            return s;
        }
        zkw_vm_tracer_streams d = v;
        uint32_t* p32 = nullptr;
        ST_TRY(B->upload(X_MAIN, &p32, v.snapshot_cycles, v.n_snapshots)); d.snapshot_cycles = p32;
        for (int k = 0; k < ZKW_VM_NUM_STREAMS; k++) { ST_TRY(B->upload(X_MAIN, &p32, v.stream_cycles[k], v.stream_len[k])); d.stream_cycles[k] = p32; }
        ST_TRY(B->upload(X_MAIN, &p32, v.decommit_state_cycles, v.n_decommit_states)); d.decommit_state_cycles = p32;
        ST_TRY(B->upload(X_MAIN, &p32, v.callstack_sponge_cycles, v.n_callstack_sponges)); d.callstack_sponge_cycles = p32;
        ST_TRY(B->upload(X_MAIN, &p32, v.storage_log_state_cycles, v.n_storage_log_states)); d.storage_log_state_cycles = p32;
        uint64_t* p64 = nullptr;
        ST_TRY(B->upload(X_MAIN, &p64, v.callstack_sponge_states, v.n_callstack_sponges * 12)); d.callstack_sponge_states = p64;
        zkw_storage_log_detailed_state* psl = nullptr;
        ST_TRY(B->upload(X_MAIN, &psl, v.storage_log_states, v.n_storage_log_states)); d.storage_log_states = psl;
        d.vm_memory_queries = B->d_all_mem;   // the VM's queries are the first n_vm of the memory queue
        d.memory_queue_tails = d_tails;
        d.decommit_queue_tails = static_cast<const uint64_t*>(zkw_decommit_witness_device_ptr(B->dec, ZKW_DEC_UNSORTED_TAILS));
        const size_t ni = v.n_snapshots - 1;
        zkw_vm_instance* d_inst = nullptr;
        ST_TRY(B->alloc(&d_inst, ni));
        ST_ZKW(zkw_vm_slice_instances(B->ctx[C_PRE], &d, d_inst, nullptr, nullptr, nullptr, nullptr));
        ST_ZKW(zkw_synchronize(B->ctx[C_PRE]));
        B->vm_instances.resize(ni);
        ST_TRY(B->xf[X_MAIN].d2h(B->vm_instances.data(), d_inst, ni * sizeof(zkw_vm_instance)));
    }

    // 4. public inputs and one recursion queue per circuit type (postprocessing/mod.rs:353-405), all queues in one launch
    {
        Timed t(B, "recursion_queues");
        struct Src { int type; const void* d_pi; const void* d_cf; size_t n; };
        std::vector<Src> src;
        src.push_back({T_DEC, zkw_decommit_witness_device_ptr(B->dec, ZKW_DEC_PUBLIC_INPUTS), zkw_decommit_witness_device_ptr(B->dec, ZKW_DEC_COMPACT_FORMS), zkw_decommit_witness_num_instances(B->dec)});
        src.push_back({T_DMX, zkw_demux_witness_device_ptr(B->dmx, ZKW_DMX_PUBLIC_INPUTS), zkw_demux_witness_device_ptr(B->dmx, ZKW_DMX_COMPACT_FORMS), zkw_demux_witness_num_instances(B->dmx)});
        src.push_back({T_RAM, zkw_ram_witness_device_ptr(B->ram, ZKW_RAM_PUBLIC_INPUTS), zkw_ram_witness_device_ptr(B->ram, ZKW_RAM_COMPACT_FORMS), zkw_ram_witness_num_instances(B->ram)});
        src.push_back({T_STO, zkw_storage_witness_device_ptr(B->sto, ZKW_STO_PUBLIC_INPUTS), zkw_storage_witness_device_ptr(B->sto, ZKW_STO_COMPACT_FORMS), zkw_storage_witness_num_instances(B->sto)});
        src.push_back({T_EVT, zkw_events_witness_device_ptr(B->evt, ZKW_EVT_PUBLIC_INPUTS), zkw_events_witness_device_ptr(B->evt, ZKW_EVT_COMPACT_FORMS), zkw_events_witness_num_instances(B->evt)});
        src.push_back({T_L1, zkw_events_witness_device_ptr(B->l1, ZKW_EVT_PUBLIC_INPUTS), zkw_events_witness_device_ptr(B->l1, ZKW_EVT_COMPACT_FORMS), zkw_events_witness_num_instances(B->l1)});
        // the circuits whose builders return instance records without compact forms (3, 5, 6, 7, 10, 13): commitments from
        // the records (zkw_closed_form_public_inputs); the MainVM closed form needs the VM's local state and is the host's
        zkw_ctx* cfc = B->ctx[C_PRE];
        auto closed_form = [&](int type, const void* d_inst, size_t n) -> Status {
            uint64_t *d_cf = nullptr, *d_pi = nullptr;
            ST_TRY(B->alloc(&d_cf, n * 18));
            ST_TRY(B->alloc(&d_pi, n * 4));
            ST_ZKW(zkw_closed_form_public_inputs(cfc, (uint8_t)type, d_inst, n, d_cf, d_pi));
            src.push_back({type, d_pi, d_cf, n});
            return Status();
        };
        ST_TRY(closed_form(T_DCM, zkw_decommitter_witness_device_ptr(B->dcm, ZKW_DCM_INSTANCES), zkw_decommitter_witness_num_instances(B->dcm)));
        for (int k = 0; k < 3; k++) {  // kept with the witness: the type-5 synthesis needs them again
            const uint64_t *d_cf = nullptr, *d_pi = nullptr;
            ST_ZKW(zkw_precompile_closed_forms(cfc, B->pre[k], &d_cf, &d_pi));
            src.push_back({T_KEC + k, d_pi, d_cf, zkw_precompile_witness_num_instances(B->pre[k])});
        }
        if (B->sap)
            ST_TRY(closed_form(T_SAP, zkw_storage_application_witness_device_ptr(B->sap, ZKW_SAP_INSTANCES),
                               zkw_storage_application_witness_num_instances(B->sap)));
        {   // LinearHasher: one instance over the net L2 -> L1 messages queue (data_hasher_and_merklizer.rs:34-60)
            zkw_linear_hasher_instance h;
            memset(&h, 0, sizeof h);
            h.start_flag = h.completion_flag = 1;
            const size_t nl = zkw_events_witness_num_instances(B->l1);
            if (nl == 0) return Status{ZKW_ERR_CHECK_FAILED, "L1-messages sorter produced no instance (an empty queue still yields one)"};
            zkw_events_sorter_instance last;
            ST_TRY(B->xf[X_MAIN].d2h(&last, static_cast<const zkw_events_sorter_instance*>(zkw_events_witness_device_ptr(B->l1, ZKW_EVT_INSTANCES)) + (nl - 1),
                                     sizeof last));
            h.queue_state = last.final_queue_state;  // take_queue_state_from_simulator of the deduplicated queue
            memcpy(h.keccak256_hash, B->l1_hash, 32);
            B->linear_hasher = h;
            zkw_linear_hasher_instance* d_h = nullptr;
            ST_TRY(B->upload(X_MAIN, &d_h, &h, 1));
            ST_TRY(closed_form(T_HSH, d_h, 1));
        }
        ST_ZKW(zkw_synchronize(cfc));
        size_t total = 0;
        for (auto& s : src) total += s.n;
        uint64_t *d_enc = nullptr, *d_states = nullptr;
        ST_TRY(B->alloc(&d_enc, total * 8));
        ST_TRY(B->alloc(&d_states, total * 12));
        std::vector<uint64_t> offs(1, 0);
        zkw_ctx* c = B->ctx[C_PRE];
        for (auto& s : src) {
            const size_t o = offs.back();
            // straight from the witness (its branch has synchronised its stream before joining)
            ST_ZKW(zkw_encode_recursion_requests(c, (uint64_t)s.type, static_cast<const uint64_t*>(s.d_pi), s.n, d_enc + 8 * o));
            offs.push_back(o + s.n);
        }
        ST_ZKW(zkw_queue_push_chain_full_batch(c, d_enc, offs.data(), src.size(), nullptr, d_states));
        ST_ZKW(zkw_synchronize(c));
        for (size_t k = 0; k < src.size(); k++) {
            PerType& p = B->per[src[k].type];
            const size_t o = offs[k], n = src[k].n;
            p.pi.resize(n * 4);
            p.enc.resize(n * 8);
            p.states.resize(n * 12);
            ST_TRY(B->xf[X_MAIN].d2h(p.pi.data(), src[k].d_pi, n * 32));
            ST_TRY(B->xf[X_MAIN].d2h(p.enc.data(), d_enc + 8 * o, n * 64));
            ST_TRY(B->xf[X_MAIN].d2h(p.states.data(), d_states + 12 * o, n * 96));
            p.compact.resize(n * 18);
            ST_TRY(B->xf[X_MAIN].d2h(p.compact.data(), src[k].d_cf, n * 18 * 8));
        }
    }
    return Status();
}

thread_local std::string g_block_error;

// HIP multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues PER PRIORITY CLASS (default 4) and work on one hardware
// queue runs in order: with the default, a 50 us kernel of one branch can sit behind another branch's 1.4 s queue chain
// (measured: builders 2.6 s instead of 1.43 s). More is not better: the chip maps about 24 user queues at once, and a
// process that has touched more (16 normal + 16 high-priority + the host framework's) is time-sliced by the hardware
// scheduler from then on, idle queues included (measured: every HBM-bound kernel 40 % slower, 1390 instead of 1870
// circuits/s in bench.py). 8 per class keeps the total below that with the chain service's 8 high-priority streams on
// queues of their own. The variable is read when the HIP runtime initialises the device, so the host calls
// zkw_process_init() before its first HIP call (an explicit call, not a load-time side effect), unless it has chosen a
// value itself.

}  // namespace

extern "C" int zkw_process_init(void) {
    setenv("GPU_MAX_HW_QUEUES", "8", 0);  // 0: a value the host exported wins
    return ZKW_OK;
}

static bool inputs_valid(const zkw_block_inputs* in) {
    if (!in || !in->decommit_queries || in->n_decommit_queries == 0 || (in->n_log_queries && !in->log_queries) ||
        (in->n_vm_memory_queries && !in->vm_memory_queries) || (in->n_bytecodes && (!in->bytecode_hashes || !in->bytecode_words || !in->bytecode_word_offsets)))
        return false;
    for (int k = 0; k < 3; k++)
        if (in->n_precompile_memory_queries[k] && !in->precompile_memory_queries[k]) return false;
    return true;
}

extern "C" int zkw_block_run(int device_id, const zkw_block_inputs* in, zkw_block** out) {
    if (!out || !inputs_valid(in)) return ZKW_ERR_INVALID;
    zkw_block* B = new zkw_block();
    B->device = device_id;
    B->t0 = Clock::now();
    Status s;
    {
        Timed t(B, "builders");
        s = run(B, in);
    }
    if (getenv("ZKW_BLOCK_PROFILE") && s.ok()) {  // per-kernel HIP-event times of every branch's context (debugging aid)
        static const char* cn[N_CTX] = {"dec", "ram", "dmx", "sto", "evt", "l1", "pre"};
        for (int i = 0; i < N_CTX; i++) {
            char names[4096];
            if (zkw_profile_names(B->ctx[i], names, sizeof names) != ZKW_OK) continue;
            char* save = nullptr;
            for (char* k = strtok_r(names, ",", &save); k; k = strtok_r(nullptr, ",", &save)) {
                double ms = 0;
                uint64_t cnt = 0;
                if (zkw_profile_get(B->ctx[i], k, &ms, &cnt) == ZKW_OK) fprintf(stderr, "[zkw_block] %-4s %-28s %10.3f ms %6llu launches\n", cn[i], k, ms, (unsigned long long)cnt);
            }
        }
    }
    if (!s.ok()) {
        zkw_block_free(B);
        // re-raise through the library's thread-local message: zkw_last_error() belongs to zkw_api; a failing zkw call on
        // THIS thread has already set it, failures on worker threads are reported through zkw_block_last_error()
        g_block_error = s.msg;
        return s.rc;
    }
    *out = B;
    return ZKW_OK;
}

extern "C" const char* zkw_block_last_error(void) { return g_block_error.c_str(); }

// K blocks at once. The blocks' builder graphs run as FIBERS of the calling thread (zkw_batch.h): no thread and no stream per block; a
// context of a block launches nothing itself, and whenever no fiber can go on the launches they left travel merged — one launch per kernel
// and stage over all K blocks, the queue chains of a stage as one chain launch on a high-priority stream. The builder code is
// zkw_block_run's, the kernels' bodies are the same (zkw_launch.h), so every block's results are what zkw_block_run gives for it alone.
// Throughput of whole blocks, not latency of one. (ZKW_BLOCKS_THREADS=1: round 5's schedule — a host thread per block and branch, the
// chain service batching the chains — kept as the baseline profiles/r06 compares against.)
static int blocks_run_threads(int device_id, const zkw_block_inputs* const* inputs, size_t n_blocks, zkw_block** out);
extern "C" int zkw_blocks_run(int device_id, const zkw_block_inputs* const* inputs, size_t n_blocks, zkw_block** out) {
    if (!inputs || !out || n_blocks == 0) return ZKW_ERR_INVALID;
    for (size_t k = 0; k < n_blocks; k++) {
        const zkw_block_inputs* in = inputs[k];
        if (!inputs_valid(in)) return ZKW_ERR_INVALID;
        out[k] = nullptr;
    }
    static const bool threads = [] { const char* e = getenv("ZKW_BLOCKS_THREADS"); return e && e[0] == '1'; }();
    if (threads) return blocks_run_threads(device_id, inputs, n_blocks, out);
    if (hipSetDevice(device_id) != hipSuccess) return ZKW_ERR_HIP;
    zkw_batch* batch = zkw_batch_create(device_id);
    if (!batch) { g_block_error = zkw_last_error(); return ZKW_ERR_HIP; }
    std::vector<zkw_block*> blocks(n_blocks, nullptr);
    std::vector<std::function<int()>> roots;
    const Clock::time_point t0 = Clock::now();
    for (size_t k = 0; k < n_blocks; k++) {
        zkw_block* B = new zkw_block();
        B->device = device_id;
        B->t0 = t0;
        B->batch = batch;
        B->use_chain_service = false;
        blocks[k] = B;
        const zkw_block_inputs* in = inputs[k];
        roots.push_back([B, in]() -> int {
            Timed t(B, "builders");
            Status s = run(B, in);
            if (!s.ok()) (void)zkw_fail(s.rc, "%s", s.msg.c_str());
            return s.rc;
        });
    }
    const int rc = zkw_batch_run(batch, roots);
    // the contexts leave the batch: ordinary contexts on the device's shared stream from here on (synthesis moves them to its workers' streams)
    void* shared = zkw_device_shared_stream(device_id);
    for (zkw_block* B : blocks) {
        B->batch = nullptr;
        B->from_batch = true;
        for (int i = 0; i < N_XFER; i++) B->xf[i].batch = nullptr;
        for (int i = 0; i < N_CTX; i++)
            if (B->ctx[i]) zkw_ctx_leave_batch(B->ctx[i], shared);
    }
    if (rc != ZKW_OK) g_block_error = zkw_last_error();
    if (getenv("ZKW_BLOCK_MEM_LOG")) {
        size_t scratch = 0, owned = 0, live = 0, idle = 0;
        static const char* cn[N_CTX] = {"dec", "ram", "dmx", "sto", "evt", "l1", "pre"};
        size_t per_ctx[N_CTX] = {};
        for (zkw_block* B : blocks)
            for (int i = 0; i < N_CTX; i++)
                if (B->ctx[i]) { per_ctx[i] += zkw_ctx_scratch_bytes(B->ctx[i]); scratch += zkw_ctx_scratch_bytes(B->ctx[i]); }
        (void)owned;
        zkw_cache_stats(&live, &idle);
        fprintf(stderr, "[zkw blocks] %zu blocks: %.1f MB of device buffers per block handed out by the cache (%.1f GB; %.1f GB idle in the cache), of which context scratch %.1f MB per block:",
                n_blocks, live / 1e6 / n_blocks, live / 1e9, idle / 1e9, scratch / 1e6 / n_blocks);
        for (int i = 0; i < N_CTX; i++) fprintf(stderr, " %s %.1f", cn[i], per_ctx[i] / 1e6 / n_blocks);
        fprintf(stderr, "\n");
    }
    zkw_batch_destroy(batch);  // (waits for the batch's streams; its arenas go back to the allocation cache)
    if (rc != ZKW_OK || !shared) {
        for (zkw_block* B : blocks) zkw_block_free(B);
        return rc != ZKW_OK ? rc : ZKW_ERR_HIP;
    }
    for (size_t k = 0; k < n_blocks; k++) out[k] = blocks[k];
    return ZKW_OK;
}

static int blocks_run_threads(int device_id, const zkw_block_inputs* const* inputs, size_t n_blocks, zkw_block** out) {
    std::vector<zkw_block*> blocks(n_blocks, nullptr);
    std::vector<std::future<Status>> futs;
    const Clock::time_point t0 = Clock::now();
    (void)zkw_chain_service_expect(device_id, (int)n_blocks);  // a stage of all these blocks is one launch (zkw_api.hip, ChainService)
    struct Unexpect { int dev, n; ~Unexpect() { (void)zkw_chain_service_expect(dev, -n); } } unexpect{device_id, (int)n_blocks};
    for (size_t k = 0; k < n_blocks; k++) {
        zkw_block* B = new zkw_block();
        B->device = device_id;
        B->t0 = t0;
        B->use_chain_service = true;
        blocks[k] = B;
        futs.push_back(std::async(std::launch::async, [B, in = inputs[k]] {
            Timed t(B, "builders");
            return run(B, in);
        }));
    }
    Status first;
    for (auto& f : futs) {
        Status s = f.get();
        if (!s.ok() && first.ok()) first = s;
    }
    if (!first.ok()) {
        for (zkw_block* B : blocks) zkw_block_free(B);
        g_block_error = first.msg;
        return first.rc;
    }
    for (size_t k = 0; k < n_blocks; k++) out[k] = blocks[k];
    return ZKW_OK;
}

// Blocks sharded over the ranks of a multi-GPU job: the mode that scales (a block's builders are bound by one serial chain, so
// splitting ONE block over GPUs only divides its synthesis; whole blocks divide everything). Round-robin: a stream of blocks of
// uneven size spreads evenly, and every rank computes the same assignment without talking.
extern "C" int zkw_blocks_owner(size_t block, int world) { return world > 0 ? (int)(block % (size_t)world) : -1; }

extern "C" int zkw_blocks_run_sharded(int device_id, const zkw_block_inputs* const* inputs, size_t n_blocks, int rank, int world, zkw_block** out) {
    if (!inputs || !out || world < 1 || rank < 0 || rank >= world) return ZKW_ERR_INVALID;
    std::vector<const zkw_block_inputs*> mine;
    std::vector<size_t> where;
    for (size_t k = 0; k < n_blocks; k++) {
        out[k] = nullptr;
        if (!inputs_valid(inputs[k])) return ZKW_ERR_INVALID;  // every rank checks every block: all ranks fail or none
        if (zkw_blocks_owner(k, world) == rank) { mine.push_back(inputs[k]); where.push_back(k); }
    }
    if (mine.empty()) return ZKW_OK;
    std::vector<zkw_block*> got(mine.size(), nullptr);
    const int rc = zkw_blocks_run(device_id, mine.data(), mine.size(), got.data());
    if (rc != ZKW_OK) return rc;
    for (size_t i = 0; i < mine.size(); i++) out[where[i]] = got[i];
    return ZKW_OK;
}

extern "C" void zkw_block_free(zkw_block* B) {
    if (!B) return;
    (void)hipSetDevice(B->device);
    // buffers go back to the library's caches, not to hipFree (which used to wait for the whole device): nothing of this
    // block may still be in flight on any of its contexts
    for (int i = 0; i < N_CTX; i++)
        if (B->ctx[i]) (void)zkw_synchronize(B->ctx[i]);
    if (B->ring) zkw_trace_free(B->ring);
    if (B->dec) zkw_decommit_witness_free(B->dec);
    if (B->dcm) zkw_decommitter_witness_free(B->dcm);
    if (B->dmx) zkw_demux_witness_free(B->dmx);
    for (int k = 0; k < 3; k++)
        if (B->pre[k]) zkw_precompile_witness_free(B->pre[k]);
    if (B->ram) zkw_ram_witness_free(B->ram);
    if (B->sto) zkw_storage_witness_free(B->sto);
    if (B->sap) zkw_storage_application_witness_free(B->sap);
    if (B->evt) zkw_events_witness_free(B->evt);
    if (B->l1) zkw_events_witness_free(B->l1);
    for (void* p : B->dev) zkw_buffer_free(0, p);
    for (int i = 0; i < N_XFER; i++) B->xf[i].destroy();  // before the contexts: the lanes name one of them
    for (int i = 0; i < N_CTX; i++)
        if (B->ctx[i]) zkw_destroy(B->ctx[i]);
    delete B;
}

// K blocks released on a few threads of the library (a block's release is ~300 buffers going back to the allocation cache and seven
// contexts torn down: ~1.2 ms of host time, which at 512 blocks per batch was 0.65 s of every 5.5 s cycle)
extern "C" void zkw_blocks_free(zkw_block* const* blocks, size_t n_blocks) {
    if (!blocks || n_blocks == 0) return;
    static const size_t max_threads = [] { const char* e = getenv("ZKW_FREE_THREADS"); const long v = e ? atol(e) : 0; return (size_t)(v > 0 && v <= 64 ? v : 8); }();
    const size_t n_threads = std::min<size_t>(max_threads, (n_blocks + 15) / 16);
    if (n_threads <= 1) {
        for (size_t k = 0; k < n_blocks; k++) zkw_block_free(blocks[k]);
        return;
    }
    std::atomic<size_t> next{0};
    std::vector<std::thread> pool;
    for (size_t th = 0; th < n_threads; th++)
        pool.emplace_back([&] {
            for (;;) {
                const size_t k = next.fetch_add(1);
                if (k >= n_blocks) return;
                zkw_block_free(blocks[k]);
            }
        });
    for (auto& t : pool) t.join();
}

extern "C" void* zkw_block_witness(const zkw_block* B, uint8_t t) {
    if (!B) return nullptr;
    switch (t) {
        case T_DEC: return B->dec;
        case T_DCM: return B->dcm;
        case T_DMX: return B->dmx;
        case T_KEC: case T_SHA: case T_ECR: return B->pre[t - T_KEC];
        case T_RAM: return B->ram;
        case T_STO: return B->sto;
        case T_SAP: return B->sap;
        case T_EVT: return B->evt;
        case T_L1: return B->l1;
        default: return nullptr;
    }
}

extern "C" zkw_ctx* zkw_block_context(const zkw_block* B, uint8_t t) {
    if (!B) return nullptr;
    switch (t) {
        case T_DEC: return B->ctx[C_DEC];
        case T_DCM: case T_KEC: case T_SHA: case T_ECR: return B->ctx[C_PRE];
        case T_DMX: return B->ctx[C_DMX];
        case T_RAM: return B->ctx[C_RAM];
        case T_STO: case T_SAP: return B->ctx[C_STO];
        case T_EVT: return B->ctx[C_EVT];
        case T_L1: case T_HSH: return B->ctx[C_L1];
        default: return nullptr;
    }
}

extern "C" size_t zkw_block_num_instances(const zkw_block* B, uint8_t t) {
    if (!B) return 0;
    switch (t) {
        case T_VM: return B->vm_instances.size();
        case T_DEC: return zkw_decommit_witness_num_instances(B->dec);
        case T_DCM: return zkw_decommitter_witness_num_instances(B->dcm);
        case T_DMX: return zkw_demux_witness_num_instances(B->dmx);
        case T_KEC: case T_SHA: case T_ECR: return zkw_precompile_witness_num_instances(B->pre[t - T_KEC]);
        case T_RAM: return zkw_ram_witness_num_instances(B->ram);
        case T_STO: return zkw_storage_witness_num_instances(B->sto);
        case T_SAP: return B->sap ? zkw_storage_application_witness_num_instances(B->sap) : 0;
        case T_EVT: return zkw_events_witness_num_instances(B->evt);
        case T_L1: return zkw_events_witness_num_instances(B->l1);
        case T_HSH: return 1;  // compute_linear_keccak256 feeds a single instance (data_hasher_and_merklizer.rs:8-67)
        default: return 0;
    }
}

extern "C" const uint64_t* zkw_block_public_inputs(const zkw_block* B, uint8_t t) {
    return B && t < 14 && !B->per[t].pi.empty() ? B->per[t].pi.data() : nullptr;
}
extern "C" const uint64_t* zkw_block_recursion_encodings(const zkw_block* B, uint8_t t) {
    return B && t < 14 && !B->per[t].enc.empty() ? B->per[t].enc.data() : nullptr;
}
extern "C" const uint64_t* zkw_block_recursion_states(const zkw_block* B, uint8_t t) {
    return B && t < 14 && !B->per[t].states.empty() ? B->per[t].states.data() : nullptr;
}
extern "C" const zkw_vm_instance* zkw_block_vm_instances(const zkw_block* B) {
    return B && !B->vm_instances.empty() ? B->vm_instances.data() : nullptr;
}
extern "C" size_t zkw_block_memory_queue_length(const zkw_block* B) { return B ? B->n_mem : 0; }
extern "C" const zkw_mem_query* zkw_block_memory_queue_device_ptr(const zkw_block* B) { return B ? B->d_all_mem : nullptr; }
extern "C" int zkw_block_memory_queue_state(const zkw_block* B, zkw_queue_state12* out) {
    if (!B || !out) return ZKW_ERR_INVALID;
    *out = B->mem_state;
    return ZKW_OK;
}
extern "C" int zkw_block_demuxed_offsets(const zkw_block* B, uint64_t out[7]) {
    if (!B || !out) return ZKW_ERR_INVALID;
    memcpy(out, B->dmx_off, sizeof B->dmx_off);
    return ZKW_OK;
}
extern "C" int zkw_block_linear_hasher_instance(const zkw_block* B, zkw_linear_hasher_instance* out) {
    if (!B || !out) return ZKW_ERR_INVALID;
    *out = B->linear_hasher;
    return ZKW_OK;
}
extern "C" int zkw_block_l1_messages_hash(const zkw_block* B, uint8_t out[32]) {
    if (!B || !out) return ZKW_ERR_INVALID;
    memcpy(out, B->l1_hash, 32);
    return ZKW_OK;
}

extern "C" int zkw_block_timings(const zkw_block* B, char* names, size_t names_bytes, double* start_ms, double* end_ms,
                                 size_t max_spans, size_t* n_spans) {
    if (!B || !n_spans) return ZKW_ERR_INVALID;
    std::string all;
    size_t n = 0;
    for (const Span& s : B->spans) {
        if (n >= max_spans) break;
        if (n) all += ",";
        all += s.name;
        if (start_ms) start_ms[n] = s.a;
        if (end_ms) end_ms[n] = s.b;
        n++;
    }
    if (names && names_bytes) {
        strncpy(names, all.c_str(), names_bytes - 1);
        names[names_bytes - 1] = 0;
    }
    *n_spans = n;
    return ZKW_OK;
}

// ---- synthesis of every instance, in the reference's emission order ----------------------------------------------
namespace {
// emission order of the synthesized types: the builders that hand their circuits to the callback themselves — log demuxer
// (oracle.rs:975-984), RAM permutation (:1039-1049), storage application (:1115-1130) —, then CircuitMaker order (:1494-1732:
// decommits sorter, code decommitter, keccak, sha256, ecrecover, storage sorter, events sorter, L1-messages sorter,
// L1-messages hasher)
const int kOrder[12] = {T_DMX, T_RAM, T_SAP, T_DEC, T_DCM, T_KEC, T_SHA, T_ECR, T_STO, T_EVT, T_L1, T_HSH};
// the block's synthesizable instances in emission order and their LPT owners
int shard_plan(const zkw_block* B, int world, std::vector<uint8_t>* types, std::vector<uint32_t>* index, std::vector<uint32_t>* owner) {
    for (int t : kOrder)
        for (size_t i = 0; i < zkw_block_num_instances(B, (uint8_t)t); i++) {
            types->push_back((uint8_t)t);
            index->push_back((uint32_t)i);
        }
    owner->assign(types->size(), 0);
    return zkw_shard_lpt(types->data(), types->size(), world, owner->data());
}
}  // namespace

extern "C" int zkw_block_synthesize(zkw_block* B, size_t n_rows, size_t ring_slots, zkw_circuit_fn cb, void* user, size_t* n_done) {
    return zkw_block_synthesize_sharded(B, n_rows, ring_slots, 0, 1, cb, user, n_done);
}

static int block_synthesize_impl(zkw_block* B, size_t n_rows, size_t ring_slots, int rank, int world, zkw_circuit_fn cb, void* user, size_t* n_done, int skip_type,
                                 zkw_trace* callers_ring, int only_type = -1, size_t slot0 = 0);
extern "C" int zkw_block_synthesize_sharded(zkw_block* B, size_t n_rows, size_t ring_slots, int rank, int world, zkw_circuit_fn cb,
                                            void* user, size_t* n_done) {
    return block_synthesize_impl(B, n_rows, ring_slots, rank, world, cb, user, n_done, -1, nullptr);
}
// skip_type: a circuit type whose instances somebody else synthesizes (zkw_blocks_synthesize: ECRecover of all blocks in joint calls);
// callers_ring: a ring of `ring_slots` slots of 153 columns the caller owns (zkw_blocks_synthesize: one per worker, not one per block —
// at 1.28 GB a slot, a ring per block capped the blocks in flight at ~150), else the block's own, created on first use
static int block_synthesize_impl(zkw_block* B, size_t n_rows, size_t ring_slots, int rank, int world, zkw_circuit_fn cb, void* user, size_t* n_done, int skip_type,
                                 zkw_trace* callers_ring, int only_type, size_t slot0) {
    if (!B || n_rows == 0 || ring_slots == 0 || world < 1 || rank < 0 || rank >= world) return ZKW_ERR_INVALID;
    std::vector<uint8_t> plan_types;
    std::vector<uint32_t> plan_index, plan_owner;
    {
        const int prc = shard_plan(B, world, &plan_types, &plan_index, &plan_owner);
        if (prc != ZKW_OK) return prc;
    }
    // owner of (type, instance) in O(1): one table per type, filled from the plan once
    std::vector<std::vector<int>> owner_of(14);
    for (size_t k = 0; k < plan_types.size(); k++) {
        std::vector<int>& o = owner_of[plan_types[k]];
        if (o.size() <= plan_index[k]) o.resize(plan_index[k] + 1, -1);
        o[plan_index[k]] = (int)plan_owner[k];
    }
    auto owned = [&](int t, size_t inst) { return inst < owner_of[t].size() && owner_of[t][inst] == rank; };
    if (hipSetDevice(B->device) != hipSuccess) return ZKW_ERR_HIP;
    if (!callers_ring && B->ring && (B->ring_rows != n_rows || B->ring_slots != ring_slots)) {
        zkw_trace_free(B->ring);
        B->ring = nullptr;
    }
    int rc = ZKW_OK;
    if (!callers_ring && !B->ring) {
        if ((rc = zkw_trace_create_with_columns(B->ctx[C_RAM], n_rows, 153, ring_slots, &B->ring)) != ZKW_OK) return rc;
        B->ring_rows = n_rows;
        B->ring_slots = ring_slots;
    }
    const double a = B->ms_now();
    size_t done = 0;
    for (int t : kOrder) {
        if (t == skip_type || (only_type >= 0 && t != only_type)) continue;
        const size_t ni = zkw_block_num_instances(B, (uint8_t)t);
        zkw_ctx* c = zkw_block_context(B, (uint8_t)t);
        zkw_trace* ring = callers_ring ? callers_ring : B->ring;

        for (size_t first = 0; first < ni;) {
            // a maximal run of consecutive instances this rank owns (world == 1: all of them), at most one ring
            if (!owned(t, first)) { first++; continue; }
            size_t cnt = 1;
            while (cnt < ring_slots && first + cnt < ni && owned(t, first + cnt)) cnt++;
            switch (t) {
                default:  // one entry point for every witness-handle type (ZkSyncBaseLayerCircuit::synthesis, base_layer/mod.rs:286-323)
                    rc = zkw_synthesize(c, (uint8_t)t, zkw_block_witness(B, (uint8_t)t), first, cnt, ring, slot0);
                    break;
                case T_HSH: {  // one instance over the net L2 -> L1 messages (the L1 sorter's result queue)
                    zkw_linear_hasher_instance rec;
                    // (the result queue's states are the events sorter's: the pops of the trace's queue section run through them)
                    const uint64_t lh_off[2] = {0, zkw_events_witness_num_results(B->l1)};
                    rc = zkw_linear_hasher_synthesize_batch_with_tails(c, static_cast<const zkw_log_query*>(zkw_events_witness_device_ptr(B->l1, ZKW_EVT_RESULT_QUERIES)), lh_off, 1,
                                                                       &B->linear_hasher.queue_state, static_cast<const uint64_t*>(zkw_events_witness_device_ptr(B->l1, ZKW_EVT_RESULT_NEW_TAILS)),
                                                                       B->cap[T_HSH], ring, slot0, &rec, nullptr);
                    break;
                }
            }
            if (rc != ZKW_OK) return rc;
            // the ring is shared by contexts with different streams: a slot is complete before the next type touches it
            if ((rc = zkw_synchronize(c)) != ZKW_OK) return rc;
            for (size_t k = 0; k < cnt; k++) {
                if (cb) {
                    const uint64_t* pi = B->per[t].pi.data() + 4 * (first + k);
                    // the callee may use the block's contexts (a check, a read): while it runs they are ordinary contexts on the batch's stream —
                    // a fiber must not park inside the caller's frames (everything this fiber queued has run: the synchronise above)
                    zkw_batch* held[N_CTX];
                    for (int i = 0; i < N_CTX; i++) held[i] = zkw_ctx_swap_batch(B->ctx[i], nullptr);
                    const int crc = cb(user, (uint8_t)t, first + k, ring, slot0 + k, pi);
                    for (int i = 0; i < N_CTX; i++) (void)zkw_ctx_swap_batch(B->ctx[i], held[i]);
                    if (crc != 0) return ZKW_ERR_CHECK_FAILED;
                }
                done++;
            }
            first += cnt;
        }
    }
    B->span("synthesis", a);
    if (n_done) *n_done = done;
    return ZKW_OK;
}

// high-priority streams for the ECRecover threads: kept for the life of the process (hipStreamDestroy waits for the whole device)
static std::mutex g_prio_mu;
static std::unordered_map<int, std::vector<hipStream_t>>& g_prio_idle = *new std::unordered_map<int, std::vector<hipStream_t>>();  // by device
static hipStream_t priority_stream_acquire(int device) {
    {
        std::lock_guard<std::mutex> g(g_prio_mu);
        std::vector<hipStream_t>& idle = g_prio_idle[device];
        if (!idle.empty()) { hipStream_t s = idle.back(); idle.pop_back(); return s; }
    }
    int lo = 0, hi = 0;
    hipStream_t s = nullptr;
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess || hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi) != hipSuccess) return nullptr;
    return s;
}
static void priority_stream_release(int device, hipStream_t s) {
    std::lock_guard<std::mutex> g(g_prio_mu);
    g_prio_idle[device].push_back(s);
}

// ---- K blocks: every instance of every block. Two things differ from K calls of zkw_block_synthesize: (1) the ECRecover instances of ALL
// blocks go through joint calls (zkw_ecrecover_synthesize_multi, at most `ec_chunk` instances each, a ring of their own): the accumulator
// chain of a request was one lane and ~13 ms per call whatever the batch (round 6: a wave, 2.5 ms), so 48 blocks' calls one after the other were 0.6 of the 0.75 s
// their synthesis took; (2) the other types run block by block on up to 8 host threads of the library (each block on its own ring of
// `ring_slots` slots and its own contexts), next to the ECRecover thread. cb may be called from several threads at once; a slot is the
// callee's until it returns (external_calls::run's circuit_callback, per block in the reference's emission order except that ECRecover
// instances arrive on their own).
extern "C" int zkw_blocks_synthesize(zkw_block* const* blocks, size_t n_blocks, size_t n_rows, size_t ring_slots, size_t ec_chunk, zkw_blocks_circuit_fn cb,
                                     void* user, size_t* n_done) {
    if (!blocks || n_blocks == 0 || n_rows == 0 || ring_slots == 0) return ZKW_ERR_INVALID;
    for (size_t k = 0; k < n_blocks; k++)
        if (!blocks[k] || blocks[k]->device != blocks[0]->device) return ZKW_ERR_INVALID;
    // the ring of the joint ECRecover calls is ec_chunk slots of 129 columns (1.08 GB each at 2^20 rows): 16 by default, at most 64
    if (ec_chunk == 0) ec_chunk = 16;
    if (ec_chunk > 64) ec_chunk = 64;
    std::atomic<size_t> done{0};
    std::atomic<int> first_rc{ZKW_OK};
    const Clock::time_point t_start = Clock::now();
    static const bool synth_log = getenv("ZKW_SYNTH_LOG") != nullptr;  // one line per thread: when it was done
    auto log_done = [&](const char* who, size_t idx) {
        if (synth_log) fprintf(stderr, "[zkw synth] %s %zu done at %.0f ms\n", who, idx, std::chrono::duration<double, std::milli>(Clock::now() - t_start).count());
    };
    auto note = [&](int rc) { int ok = ZKW_OK; if (rc != ZKW_OK) first_rc.compare_exchange_strong(ok, rc); };
    struct Fwd { size_t block; zkw_blocks_circuit_fn cb; void* user; };
    auto fwd = [](void* u, uint8_t type, size_t inst, const zkw_trace* tr, size_t slot, const uint64_t* pi) -> int {
        const Fwd* f = static_cast<const Fwd*>(u);
        return f->cb ? f->cb(f->user, f->block, type, inst, tr, slot, pi) : 0;
    };
    // (1) ECRecover of all blocks: joint calls of up to ec_chunk instances, on ZKW_EC_THREADS threads (default 2), each with a ring and a context of
    // its own on a HIGH-PRIORITY stream. A call is latency — the accumulator chain of a request is one wave, ~2.5 ms whatever the batch (13 ms when this was written), the
    // segment evaluator's dependent loads another ~9 ms — and next to the other types' fills a single thread on an ordinary stream was the
    // long pole of the whole synthesis (its kernels queued behind the fills: 2.2 - 3.1 s for the 1 024 instances of 512 blocks, the workers done
    // after 1.6 - 1.9 s); priority streams have hardware queues of their own, and two calls in flight hide each other's serial kernels.
    static const size_t ec_threads_max = [] { const char* e = getenv("ZKW_EC_THREADS"); const long v = e ? atol(e) : 0; return (size_t)(v > 0 && v <= 8 ? v : 2); }();
    size_t ec_most = 0, ec_total = 0;
    for (size_t k = 0; k < n_blocks; k++) { const size_t n = zkw_block_num_instances(blocks[k], T_ECR); ec_most = std::max(ec_most, n); ec_total += n; }
    const size_t ec_slots = std::max(ec_most, std::min(ec_chunk, ec_total));
    const size_t n_ec_threads = ec_total == 0 ? 0 : std::max<size_t>(1, std::min(ec_threads_max, (ec_total + ec_slots - 1) / ec_slots));
    std::mutex ec_mu;
    size_t ec_cursor = 0;
    std::vector<std::thread> ec_pool;
    for (size_t et = 0; et < n_ec_threads; et++)
        ec_pool.emplace_back([&, et] {
            if (hipSetDevice(blocks[0]->device) != hipSuccess) { note(ZKW_ERR_HIP); return; }
            // a context of its own: the blocks' precompile contexts are busy with their Keccak / SHA-256 / decommitter instances on the other threads
            zkw_ctx* c = zkw_create(blocks[0]->device);
            if (!c) { note(ZKW_ERR_NO_DEVICE); return; }
            hipStream_t hp = priority_stream_acquire(blocks[0]->device);
            int rc = hp ? zkw_set_stream(c, hp) : ZKW_OK;  // (no priority stream: the context's own)
            zkw_trace* ring = nullptr;
            if (rc == ZKW_OK) rc = zkw_trace_create_with_columns(c, n_rows, 129 /* 80 + 3 x 16 + 1 columns: include/zkw_ecrecover_circuit_spec.h EK_COLS */, ec_slots, &ring);
            while (rc == ZKW_OK && first_rc.load() == ZKW_OK) {
                std::vector<zkw_precompile_witness*> ws;
                size_t cnt = 0, b0, b1;
                {
                    std::lock_guard<std::mutex> g(ec_mu);
                    b0 = b1 = ec_cursor;
                    while (b1 < n_blocks && cnt + zkw_block_num_instances(blocks[b1], T_ECR) <= ec_slots) {
                        ws.push_back(static_cast<zkw_precompile_witness*>(zkw_block_witness(blocks[b1], T_ECR)));
                        cnt += zkw_block_num_instances(blocks[b1], T_ECR);
                        b1++;
                    }
                    ec_cursor = b1;
                }
                if (b0 == b1) break;
                rc = zkw_ecrecover_synthesize_multi(c, ws.data(), ws.size(), ring, 0);
                if (rc == ZKW_OK) rc = zkw_synchronize(c);
                if (rc != ZKW_OK) break;
                size_t slot = 0;
                for (size_t b = b0; b < b1 && first_rc.load() == ZKW_OK; b++) {
                    const size_t ni = zkw_block_num_instances(blocks[b], T_ECR);
                    for (size_t i = 0; i < ni; i++, slot++) {
                        if (cb && cb(user, b, (uint8_t)T_ECR, i, ring, slot, blocks[b]->per[T_ECR].pi.data() + 4 * i) != 0) { note(ZKW_ERR_CHECK_FAILED); break; }
                        done++;
                    }
                }
            }
            if (rc != ZKW_OK) note(rc);
            log_done("ecrecover thread", et);
            if (ring) zkw_trace_free(ring);
            if (hp) { (void)zkw_set_stream(c, ZKW_STREAM_OWN); priority_stream_release(blocks[0]->device, hp); }
            zkw_destroy(c);
        });
    // (2) everything else: a few workers, each with a contiguous share of the blocks, a ring of G x ring_slots slots and a batch (zkw_batch.h)
    // of G fibers. Fiber s owns slot(s) s of the ring and every G-th block of the worker's share, and goes through them TYPE BY TYPE — the
    // LogDemuxer instances of its blocks, then their RAMPermutation instances, ... — running the code zkw_block_synthesize runs. The G fibers
    // move in step, so (a) the fills of a type leave as ONE launch per kernel over G instances instead of one small launch per block (a
    // Keccak instance is 293 waves; the serial sponges — 3 - 4.5 ms of one wave for the L1-messages hasher — cost their latency once per G
    // blocks), and (b) a slot holds the same layout call after call, so a fill only rewrites the cells it owns (zkw_trace::slot_tag): no
    // zeroing pass over the ~1 GB of general columns of a netlist circuit, about half the bytes of a queue circuit. Round 5 ran the blocks one
    // by one on eight threads with a ring per BLOCK (1.28 GB a slot: ~150 blocks in flight at most, every call into a slot a cold one).
    // ZKW_SYNTH_THREADS workers (default 3: while one worker's fibers wait for a serial kernel the others' fills have the chip; 2 and 4 measured about the same),
    // ZKW_SYNTH_GROUP fibers per worker (default 16). cb is called on the workers' threads, one call at a time per worker.
    std::vector<std::thread> pool;
    static const size_t max_threads = [] { const char* e = getenv("ZKW_SYNTH_THREADS"); const long v = e ? atol(e) : 0; return (size_t)(v > 0 && v <= 64 ? v : 3); }();
    static const size_t group_size = [] { const char* e = getenv("ZKW_SYNTH_GROUP"); const long v = e ? atol(e) : 0; return (size_t)(v > 0 && v <= 256 ? v : 16); }();
    const size_t n_threads = std::max<size_t>(1, std::min<size_t>(max_threads, (n_blocks + group_size - 1) / group_size));
    for (size_t th = 0; th < n_threads; th++)
        pool.emplace_back([&, th] {
            const size_t b0 = n_blocks * th / n_threads, b1 = n_blocks * (th + 1) / n_threads;
            if (b0 == b1) return;
            const size_t G = std::min(group_size, b1 - b0);
            if (hipSetDevice(blocks[0]->device) != hipSuccess) { note(ZKW_ERR_HIP); return; }
            zkw_ctx* wc = zkw_create(blocks[0]->device);
            if (!wc) { note(ZKW_ERR_NO_DEVICE); return; }
            zkw_trace* ring = nullptr;
            int rc = zkw_trace_create_with_columns(wc, n_rows, 153, G * ring_slots, &ring);
            void* shared = zkw_device_shared_stream(blocks[0]->device);
            zkw_batch* batch = rc == ZKW_OK ? zkw_batch_create(blocks[0]->device) : nullptr;
            log_done("worker's ring and batch ready:", th);
            if (rc != ZKW_OK || !shared || !batch) {
                note(rc != ZKW_OK ? rc : ZKW_ERR_HIP);
                if (batch) zkw_batch_destroy(batch);
                if (ring) zkw_trace_free(ring);
                zkw_destroy(wc);
                return;
            }
            std::vector<size_t> counts(b1 - b0, 0);
            for (size_t b = b0; b < b1; b++)
                for (int i = 0; i < N_CTX; i++) {
                    (void)zkw_synchronize(blocks[b]->ctx[i]);  // (idle already: the builders are done)
                    zkw_ctx_enter_batch(blocks[b]->ctx[i], batch);
                }
            std::vector<std::function<int()>> roots;
            for (size_t s = 0; s < G; s++)
                roots.push_back([&, s]() -> int {
                    for (int t : kOrder) {
                        if (t == T_ECR) continue;
                        for (size_t b = b0 + s; b < b1; b += G) {
                            if (first_rc.load() != ZKW_OK) return ZKW_OK;
                            Fwd f{b, cb, user};
                            size_t n = 0;
                            // what the synthesis of a type adds to its context's scratch (windows, gathers: up to ~50 MB) goes back as soon as
                            // its instances are done: kept until the block's release it was 100 MB x every block of the batch
                            zkw_ctx* c = zkw_block_context(blocks[b], (uint8_t)t);
                            std::vector<std::string> mark;
                            zkw_ctx_scratch_mark(c, &mark);
                            const int r = block_synthesize_impl(blocks[b], n_rows, ring_slots, 0, 1, fwd, &f, &n, T_ECR, ring, t, s * ring_slots);
                            counts[b - b0] += n;
                            if (r != ZKW_OK) return r;
                            zkw_ctx_scratch_release_since(c, mark);
                        }
                    }
                    return ZKW_OK;
                });
            rc = zkw_batch_run(batch, roots);
            for (size_t b = b0; b < b1; b++) {
                zkw_block* B = blocks[b];
                for (int i = 0; i < N_CTX; i++) {
                    // a block built by a batch has no stream of its own and returns to the device's shared one, any other to its own
                    zkw_ctx_leave_batch(B->ctx[i], shared);
                    if (!B->from_batch) (void)zkw_set_stream(B->ctx[i], ZKW_STREAM_OWN);
                }
                done += counts[b - b0];
            }
            if (rc != ZKW_OK) note(rc);
            log_done("worker", th);
            zkw_batch_destroy(batch);
            zkw_trace_free(ring);
            zkw_destroy(wc);
        });
    for (auto& t : pool) t.join();
    for (auto& t : ec_pool) t.join();
    if (getenv("ZKW_BLOCK_MEM_LOG")) {
        size_t scratch = 0, live = 0, idle = 0, per_ctx[N_CTX] = {};
        static const char* cn[N_CTX] = {"dec", "ram", "dmx", "sto", "evt", "l1", "pre"};
        for (size_t k = 0; k < n_blocks; k++)
            for (int i = 0; i < N_CTX; i++)
                if (blocks[k]->ctx[i]) { per_ctx[i] += zkw_ctx_scratch_bytes(blocks[k]->ctx[i]); scratch += zkw_ctx_scratch_bytes(blocks[k]->ctx[i]); }
        zkw_cache_stats(&live, &idle);
        fprintf(stderr, "[zkw blocks] after the synthesis of %zu blocks: %.1f GB of device buffers handed out (%.1f GB idle in the cache), context scratch %.1f MB per block:", n_blocks, live / 1e9,
                idle / 1e9, scratch / 1e6 / n_blocks);
        for (int i = 0; i < N_CTX; i++) fprintf(stderr, " %s %.1f", cn[i], per_ctx[i] / 1e6 / n_blocks);
        fprintf(stderr, "\n");
    }
    if (n_done) *n_done = done.load();
    return first_rc.load();
}

// ---- multi-GPU: every rank has run the builders (they are deterministic and bounded by one serial chain, so replicating
// them costs nothing on the critical path), synthesizes the instances the LPT plan gives it, and the per-instance
// closed-form records travel to the root in ONE gather: record = [circuit_type, instance, compact form (18), public
// input (4)] = 24 words = 192 bytes. The root receives them in emission order, ready for the recursion-queue replay.
extern "C" int zkw_block_gather_closed_form_inputs(zkw_block* B, zkw_comm* comm, int rank, int world, int root, uint64_t* out,
                                                   size_t max_records, size_t* n_records) {
    if (!B || !comm || world < 1 || rank < 0 || rank >= world || root < 0 || root >= world) return ZKW_ERR_INVALID;
    if (hipSetDevice(B->device) != hipSuccess) return ZKW_ERR_HIP;
    std::vector<uint8_t> types;
    std::vector<uint32_t> index, owner;
    int rc = shard_plan(B, world, &types, &index, &owner);
    if (rc != ZKW_OK) return rc;
    const size_t n = types.size();
    std::vector<uint64_t> mine;
    for (size_t k = 0; k < n; k++) {
        if ((int)owner[k] != rank) continue;
        const PerType& p = B->per[types[k]];
        mine.push_back(types[k]);
        mine.push_back(index[k]);
        mine.insert(mine.end(), p.compact.begin() + 18 * index[k], p.compact.begin() + 18 * (index[k] + 1));
        mine.insert(mine.end(), p.pi.begin() + 4 * index[k], p.pi.begin() + 4 * (index[k] + 1));
    }
    // host records in, host records out: the communicator stages them through its own (reused) device buffers, synchronises
    // on every rank, and hands the root the records in emission order
    if (rank == root && n && (!out || max_records < n)) return ZKW_ERR_INVALID;
    if ((rc = zkw_gather_records(comm, owner.data(), n, mine.data(), 24 * 8, root, out)) != ZKW_OK) return rc;
    if (n_records) *n_records = n;
    return ZKW_OK;
}

// The gather of the sharded-blocks mode: block k's records [n_instances, then per instance (circuit_type, instance, compact form
// (18), public input (4))] travel from the block's owner to the root as ONE fixed-size record of 1 + 24 * max_per_block words,
// so that no rank needs another rank's instance counts to take part. out[n_blocks][1 + 24 * max_per_block] (host, root only).
extern "C" int zkw_blocks_gather_closed_form_inputs(zkw_block* const* blocks, size_t n_blocks, zkw_comm* comm, int rank, int world, int root,
                                                    size_t max_per_block, uint64_t* out) {
    if (!comm || (n_blocks && !blocks) || world < 1 || rank < 0 || rank >= world || root < 0 || root >= world || max_per_block == 0) return ZKW_ERR_INVALID;
    const size_t words = 1 + 24 * max_per_block;
    // A failure that only this rank can see (a block that was not built, a block with too many instances) must not keep the
    // rank out of the collective — the others would wait for it forever. The block's record travels with an error sentinel
    // in its count word instead; the failing rank AND the root return the error after the gather.
    const uint64_t BAD = ~0ull;
    int local_rc = ZKW_OK;
    std::vector<uint32_t> owner(n_blocks);
    std::vector<uint64_t> mine;
    for (size_t k = 0; k < n_blocks; k++) {
        owner[k] = (uint32_t)zkw_blocks_owner(k, world);
        if ((int)owner[k] != rank) continue;
        const size_t base = mine.size();
        mine.resize(base + words, 0);
        zkw_block* B = blocks[k];
        std::vector<uint8_t> types;
        std::vector<uint32_t> index, one;
        int rc = B ? shard_plan(B, 1, &types, &index, &one) : ZKW_ERR_INVALID;
        if (rc == ZKW_OK && types.size() > max_per_block) {
            g_block_error = "zkw_blocks_gather_closed_form_inputs: block has more instances than max_per_block";
            rc = ZKW_ERR_INVALID;
        } else if (!B) {
            g_block_error = "zkw_blocks_gather_closed_form_inputs: a block this rank owns is NULL";
        }
        if (rc != ZKW_OK) {
            mine[base] = BAD;
            if (local_rc == ZKW_OK) local_rc = rc;
            continue;
        }
        mine[base] = types.size();
        for (size_t i = 0; i < types.size(); i++) {
            const PerType& p = B->per[types[i]];
            uint64_t* r = &mine[base + 1 + 24 * i];
            r[0] = types[i];
            r[1] = index[i];
            std::copy(p.compact.begin() + 18 * index[i], p.compact.begin() + 18 * (index[i] + 1), r + 2);
            std::copy(p.pi.begin() + 4 * index[i], p.pi.begin() + 4 * (index[i] + 1), r + 20);
        }
    }
    std::vector<uint64_t> sink;  // a root without an output array still takes part, then reports the bad argument
    const bool no_out = rank == root && n_blocks && !out;
    if (no_out) sink.resize(n_blocks * words);
    int rc = zkw_gather_records(comm, owner.data(), n_blocks, mine.data(), words * 8, root, no_out ? sink.data() : out);
    if (rc != ZKW_OK) return rc;
    if (local_rc != ZKW_OK) return local_rc;
    if (no_out) return ZKW_ERR_INVALID;
    if (rank == root)
        for (size_t k = 0; k < n_blocks; k++)
            if (out[k * words] == BAD) {
                g_block_error = "zkw_blocks_gather_closed_form_inputs: the owner of block " + std::to_string(k) + " (rank " + std::to_string(owner[k]) + ") reported a failure";
                return ZKW_ERR_INVALID;
            }
    return ZKW_OK;
}
