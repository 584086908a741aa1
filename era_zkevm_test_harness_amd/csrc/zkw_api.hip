// zkw_api.hip — host side of libzkw above the kernels and the extern "C" boundary (include/zkw.h).
//
// Host logic mirrors the reference's builders (Rust, compiled code => C++ here):
//   RamBuilder::run  <->  compute_ram_circuit_snapshots, src/witness/individual_circuits/ram_permutation.rs:26-470
// One process drives one GPU; everything is enqueued on the context's stream, scratch lives in a
// grow-only per-context pool so that steady-state calls do no hipMalloc.
#include "zkw_ctx.h"
#include "circuit_check_host.h"
#include "closed_forms_host.h"
#include "callstack_kernels.cuh"
#include "vm_kernels.cuh"
#include "radix_sort.cuh"

// ------------------------------------------------------------------------------------------------ errors
static thread_local std::string g_last_error;

#undef fail
int zkw_fail(int code, const char* fmt, ...) {  // the same for the library's other translation units (zkw_internal.h)
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define fail zkw_fail

// ------------------------------------------------------------------------------------------------ allocation cache
// hipFree / hipHostFree wait for EVERY stream of the device and hipMalloc takes the runtime's global lock: with many
// blocks in flight (zkw_blocks_run) the ~300 buffers of each block cost more than its kernels (measured: 65 ms of
// frees per block, and a 1.25 GB trace ring costs up to 20 ms to map). Freed buffers therefore go to
// a per-device list of size classes and are handed out again: 8 classes per octave up to SMALL_MAX bytes (at most
// 12.5 % slack), the exact size rounded to 2 MiB above it (the traces and the benchmark-sized witness arrays recur with
// the same sizes step after step; no slack where HBM is tight). Nothing here zeroes memory: as with hipMalloc, the
// contents of a new buffer are unspecified. An allocation failure empties the cache and retries once;
// zkw_trim_caches() empties it on request; ZKW_ALLOC_CACHE=0 turns it off.
struct AllocCache {
    static constexpr size_t SMALL_MAX = size_t(256) << 20;
    static constexpr int MAX_DEV = 16;
    std::mutex mu;
    std::map<const void*, std::pair<int, size_t>> live[2];  // [pinned host?] pointer -> (device, class bytes)
    std::map<size_t, std::vector<void*>> idle[2][MAX_DEV];
    const bool enabled = [] { const char* e = getenv("ZKW_ALLOC_CACHE"); return !(e && e[0] == '0'); }();

    static size_t size_class(size_t b) {
        if (b <= 4096) return 4096;
        if (b > SMALL_MAX) return (b + ((size_t(2) << 20) - 1)) & ~((size_t(2) << 20) - 1);
        int top = 63 - __builtin_clzll(b);
        const size_t step = size_t(1) << (top - 3);
        return (b + step - 1) & ~(step - 1);
    }
    static hipError_t raw_alloc(int host, void** p, size_t bytes) {
        return host ? hipHostMalloc(p, bytes, hipHostMallocDefault) : hipMalloc(p, bytes);
    }
    hipError_t alloc(int host, void** p, size_t bytes) {
        int dev = 0;
        if (!enabled || hipGetDevice(&dev) != hipSuccess || dev >= MAX_DEV) return raw_alloc(host, p, bytes);
        const size_t cls = size_class(bytes);
        {
            std::lock_guard<std::mutex> g(mu);
            auto it = idle[host][dev].find(cls);
            if (it != idle[host][dev].end() && !it->second.empty()) {
                *p = it->second.back();
                it->second.pop_back();
                live[host][*p] = {dev, cls};
                return hipSuccess;
            }
        }
        hipError_t e = raw_alloc(host, p, cls);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            // the device (or the pinned pool) is full of buffers the cache holds idle: give them all back and try once more. Every hipFree
            // waits for the device, so this costs seconds when thousands of buffers are cached — worth a line when it happens
            static std::atomic<int> n_trims{0};
            if (n_trims.fetch_add(1) < 8)
                fprintf(stderr, "[zkw] out of %s memory at a request of %zu bytes: the allocation cache is emptied and the request retried (expect a stall)\n", host ? "pinned host" : "device", cls);
            trim();
            e = raw_alloc(host, p, cls);
        }
        if (e == hipSuccess) {
            std::lock_guard<std::mutex> g(mu);
            live[host][*p] = {dev, cls};
        }
        return e;
    }
    // the caller has synchronised whatever used the buffer (every zkw_*_free / context destruction does)
    void release(int host, void* p) {
        if (!p) return;
        {
            std::lock_guard<std::mutex> g(mu);
            auto it = live[host].find(p);
            if (it != live[host].end()) {
                idle[host][it->second.first][it->second.second].push_back(p);
                live[host].erase(it);
                return;
            }
        }
        if (host) (void)hipHostFree(p); else (void)hipFree(p);
    }
    void trim() {
        std::vector<void*> d, h;
        {
            std::lock_guard<std::mutex> g(mu);
            for (int dev = 0; dev < MAX_DEV; ++dev) {
                for (auto& kv : idle[0][dev]) { d.insert(d.end(), kv.second.begin(), kv.second.end()); kv.second.clear(); }
                for (auto& kv : idle[1][dev]) { h.insert(h.end(), kv.second.begin(), kv.second.end()); kv.second.clear(); }
            }
        }
        for (void* q : d) (void)hipFree(q);
        for (void* q : h) (void)hipHostFree(q);
    }
};
static AllocCache& alloc_cache() {
    static AllocCache* c = new AllocCache();  // never destroyed: the HIP runtime may be gone by static destruction time
    return *c;
}
hipError_t zkw_cache_alloc(int pinned_host, void** p, size_t bytes) { return alloc_cache().alloc(pinned_host ? 1 : 0, p, bytes); }
void zkw_cache_release(int pinned_host, void* p) { alloc_cache().release(pinned_host ? 1 : 0, p); }


// Streams are pooled for the same reason: hipStreamDestroy waits for the whole device. A released stream has been
// synchronised by its owner; it keeps its hardware queue.
struct StreamPool {
    std::mutex mu;
    std::vector<hipStream_t> idle[AllocCache::MAX_DEV];
    hipError_t acquire(hipStream_t* s) {
        int dev = 0;
        if (alloc_cache().enabled && hipGetDevice(&dev) == hipSuccess && dev < AllocCache::MAX_DEV) {
            std::lock_guard<std::mutex> g(mu);
            if (!idle[dev].empty()) {
                *s = idle[dev].back();
                idle[dev].pop_back();
                return hipSuccess;
            }
        }
        return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
    }
    void release(hipStream_t s) {
        if (!s) return;
        int dev = 0;
        if (alloc_cache().enabled && hipGetDevice(&dev) == hipSuccess && dev < AllocCache::MAX_DEV) {
            std::lock_guard<std::mutex> g(mu);
            idle[dev].push_back(s);
            return;
        }
        (void)hipStreamDestroy(s);
    }
    void trim() {
        std::vector<hipStream_t> all;
        {
            std::lock_guard<std::mutex> g(mu);
            for (auto& v : idle) { all.insert(all.end(), v.begin(), v.end()); v.clear(); }
        }
        for (hipStream_t s : all) (void)hipStreamDestroy(s);
    }
};
static StreamPool& stream_pool() {
    static StreamPool* p = new StreamPool();
    return *p;
}

hipError_t zkw_pool_stream_acquire(hipStream_t* s) { return stream_pool().acquire(s); }
void zkw_pool_stream_release(hipStream_t s) { stream_pool().release(s); }

extern "C" void zkw_trim_caches(void) {
    alloc_cache().trim();
    stream_pool().trim();
}

extern "C" const char* zkw_last_error(void) { return g_last_error.c_str(); }
extern "C" const char* zkw_version(void) { return "zkw 0.2 (gfx950)"; }

extern "C" zkw_ctx* zkw_create(int device_id) {
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0) {
        fail(ZKW_ERR_NO_DEVICE, "no HIP device available (%s); libzkw has no CPU fallback",
             e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
        return nullptr;
    }
    if (device_id < 0 || device_id >= count) {
        fail(ZKW_ERR_INVALID, "device_id %d out of range [0, %d)", device_id, count);
        return nullptr;
    }
    if (hipSetDevice(device_id) != hipSuccess) {
        fail(ZKW_ERR_HIP, "hipSetDevice(%d) failed", device_id);
        return nullptr;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess) {
        if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
            fail(ZKW_ERR_NO_DEVICE, "device %d is %s; libzkw kernels are built for gfx950 only", device_id,
                 prop.gcnArchName);
            return nullptr;
        }
    }
    zkw_ctx* ctx = new zkw_ctx();
    ctx->device = device_id;
    if (stream_pool().acquire(&ctx->own_stream) != hipSuccess) {
        fail(ZKW_ERR_HIP, "hipStreamCreate failed");
        delete ctx;
        return nullptr;
    }
    ctx->stream = ctx->own_stream;
    return ctx;
}

static void ctx_destroy_now(zkw_ctx* ctx) {
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& kv : ctx->pool)
        if (kv.second.p) dev_free(kv.second.p);
    for (auto& sp : ctx->spans) { (void)hipEventDestroy(sp.a); (void)hipEventDestroy(sp.b); }
    for (auto e : ctx->free_events) (void)hipEventDestroy(e);
    for (auto& kv : ctx->stages) {
        if (kv.second.p) pin_free(kv.second.p);
        if (kv.second.ev) (void)hipEventDestroy(kv.second.ev);
    }
    for (void* q : ctx->retired_dev) dev_free(q);
    for (void* q : ctx->retired_host) pin_free(q);
    if (ctx->pinned_rb) pin_free(ctx->pinned_rb);
    if (ctx->chain_ev_a) (void)hipEventDestroy(ctx->chain_ev_a);
    if (ctx->chain_ev_b) (void)hipEventDestroy(ctx->chain_ev_b);
    if (ctx->side_stream) {  // a fork nobody joined
        (void)hipStreamSynchronize(ctx->side_stream);
        stream_pool().release(ctx->side_stream);
    }
    if (ctx->side_ev_fork) {
        (void)hipEventDestroy(ctx->side_ev_fork);
        (void)hipEventDestroy(ctx->side_ev_join);
    }
    if (ctx->own_stream) {
        (void)hipStreamSynchronize(ctx->own_stream);
        stream_pool().release(ctx->own_stream);
    }
    delete ctx;
}
void ctx_retain(zkw_ctx* ctx) { ctx->children.fetch_add(1); }
// the context's internals the library's other translation units need (zkw_internal.h)
int zkw_ctx_device(const zkw_ctx* ctx) { return ctx->device; }
void* zkw_ctx_stream(const zkw_ctx* ctx) { return ctx->stream; }
void zkw_ctx_retain(zkw_ctx* ctx) { ctx_retain(ctx); }
void zkw_ctx_release(zkw_ctx* ctx) { ctx_release(ctx); }
// zkw_destroy and the last child's release may race: whoever flips `destroying` first destroys, exactly once
static void ctx_try_destroy(zkw_ctx* ctx) {
    bool expected = false;
    if (ctx->destroying.compare_exchange_strong(expected, true)) ctx_destroy_now(ctx);
}
void ctx_release(zkw_ctx* ctx) {
    if (ctx->children.fetch_sub(1) == 1 && ctx->destroy_requested.load()) ctx_try_destroy(ctx);
}

// Witnesses and traces dereference their context when they are read or freed, so a context with outstanding
// children is only marked: the last zkw_*_free destroys it (zkw.h "Lifetimes").
extern "C" void zkw_destroy(zkw_ctx* ctx) {
    if (!ctx) return;
    if (ctx->children.load() > 0) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
    }
    // request first, THEN look at the children: a release that drops the last child after this store sees the request; one
    // that dropped it before is seen by the load below. Either way exactly one side wins ctx_try_destroy.
    ctx->destroy_requested.store(true);
    if (ctx->children.load() == 0) ctx_try_destroy(ctx);
}

extern "C" int zkw_set_stream(zkw_ctx* ctx, void* s) {
    if (!ctx) return fail(ZKW_ERR_INVALID, "null context");
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    ctx->stream = (s == ZKW_STREAM_OWN) ? ctx->own_stream : static_cast<hipStream_t>(s);
    return ZKW_OK;
}

extern "C" int zkw_buffer_alloc(zkw_ctx* ctx, int pinned_host, size_t bytes, void** out) {
    if (!ctx || !out) return fail(ZKW_ERR_INVALID, "zkw_buffer_alloc: null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    *out = nullptr;
    hipError_t e = pinned_host ? pin_malloc(out, bytes ? bytes : 1) : dev_malloc(out, bytes ? bytes : 1);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(ZKW_ERR_OOM, "zkw_buffer_alloc: %zu bytes of %s memory: %s", bytes, pinned_host ? "pinned host" : "device", hipGetErrorString(e));
    }
    return ZKW_OK;
}
extern "C" void zkw_buffer_free(int pinned_host, void* p) {
    if (pinned_host) pin_free(p); else dev_free(p);
}
extern "C" int zkw_stream_acquire(zkw_ctx* ctx, void** stream) {
    if (!ctx || !stream) return fail(ZKW_ERR_INVALID, "zkw_stream_acquire: null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = nullptr;
    HIP_TRY(stream_pool().acquire(&s));
    *stream = s;
    return ZKW_OK;
}
extern "C" void zkw_stream_release(zkw_ctx* ctx, void* stream) {
    if (!ctx || !stream) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(static_cast<hipStream_t>(stream));
    stream_pool().release(static_cast<hipStream_t>(stream));
}

extern "C" int zkw_set_chain_stream(zkw_ctx* ctx, void* s) {
    if (!ctx) return fail(ZKW_ERR_INVALID, "null context");
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (s && !(ctx->chain_ev_a && ctx->chain_ev_b)) {  // both events or neither
        hipEvent_t a = nullptr, b = nullptr;
        HIP_TRY(hipEventCreateWithFlags(&a, hipEventDisableTiming));
        hipError_t e = hipEventCreateWithFlags(&b, hipEventDisableTiming);
        if (e != hipSuccess) {
            (void)hipEventDestroy(a);
            return fail(ZKW_ERR_HIP, "hipEventCreate failed: %s", hipGetErrorString(e));
        }
        if (ctx->chain_ev_a) (void)hipEventDestroy(ctx->chain_ev_a);
        if (ctx->chain_ev_b) (void)hipEventDestroy(ctx->chain_ev_b);
        ctx->chain_ev_a = a;
        ctx->chain_ev_b = b;
    }
    ctx->chain_stream = static_cast<hipStream_t>(s);
    return ZKW_OK;
}

extern "C" int zkw_set_pointer_mode(zkw_ctx* ctx, int mode) {
    if (!ctx || (mode != ZKW_PTR_HOST && mode != ZKW_PTR_DEVICE)) return fail(ZKW_ERR_INVALID, "bad pointer mode");
    ctx->ptr_mode = mode;
    return ZKW_OK;
}

extern "C" int zkw_set_chain_form(zkw_ctx* ctx, int lanes_per_state) {
    if (!ctx || (lanes_per_state != 0 && lanes_per_state != 1 && lanes_per_state != 2 && lanes_per_state != 4 && lanes_per_state != 16))
        return fail(ZKW_ERR_INVALID, "chain form must be 0 (auto), 1, 2, 4 or 16");
    ctx->chain_form = lanes_per_state;
    return ZKW_OK;
}

extern "C" int zkw_set_netlist_fill_form(zkw_ctx* ctx, int form) {
    if (!ctx || (form != 0 && form != 1)) return fail(ZKW_ERR_INVALID, "netlist fill form must be 0 (a wave per cycle) or 1 (a lane per cycle)");
    ctx->netlist_fill_form = form;
    return ZKW_OK;
}

extern "C" int zkw_synchronize(zkw_ctx* ctx) {
    if (!ctx) return fail(ZKW_ERR_INVALID, "null context");
    if (ctx->batched()) return zkw_batch_sync(ctx->batch);  // parks the calling fiber until what it queued has run
    HIP_TRY(ctx->sync_stream());  // (and whatever a failed call left on the side stream)
    return ZKW_OK;
}

// A context of a block that runs inside a batch (zkw_batch.h): no stream of its own — what it launches travels on the batch's stream,
// merged with the other blocks' launches. zkw_ctx_leave_batch hands it over to ordinary use (synthesis, getters) on that stream.
zkw_ctx* zkw_ctx_create_in_batch(int device_id, zkw_batch* b) {
    zkw_ctx* ctx = new zkw_ctx();
    ctx->device = device_id;
    ctx->batch = b;
    ctx->stream = zkw_batch_stream(b);
    ctx->ptr_mode = ZKW_PTR_DEVICE;
    return ctx;
}
// (diagnostics, ZKW_BLOCK_MEM_LOG) bytes of a context's named scratch; device bytes handed out / idle in the allocation cache
size_t zkw_ctx_scratch_bytes(const zkw_ctx* ctx) {
    size_t b = 0;
    for (auto& kv : ctx->pool) b += kv.second.cap;
    for (void* q : ctx->retired_dev) { (void)q; }
    return b;
}
// Scratch a context has acquired since `mark` (a set of names, zkw_ctx_scratch_mark) goes back to the allocation cache: the synthesis of
// a block's instances leaves ~100 MB of windows in the block's contexts that nothing reads again. The stream must be idle.
void zkw_ctx_scratch_mark(const zkw_ctx* ctx, std::vector<std::string>* names) {
    names->clear();
    for (auto& kv : ctx->pool) names->push_back(kv.first);
}
void zkw_ctx_scratch_release_since(zkw_ctx* ctx, const std::vector<std::string>& names) {
    for (auto it = ctx->pool.begin(); it != ctx->pool.end();) {
        if (std::find(names.begin(), names.end(), it->first) == names.end()) {
            if (it->second.p) dev_free(it->second.p);
            it = ctx->pool.erase(it);
        } else {
            ++it;
        }
    }
}
void zkw_cache_stats(size_t* live_bytes, size_t* idle_bytes) {
    AllocCache& c = alloc_cache();
    std::lock_guard<std::mutex> g(c.mu);
    size_t l = 0, i = 0;
    for (auto& kv : c.live[0]) l += kv.second.second;
    for (int d = 0; d < AllocCache::MAX_DEV; d++)
        for (auto& kv : c.idle[0][d]) i += kv.first * kv.second.size();
    *live_bytes = l;
    *idle_bytes = i;
}
int zkw_copy_device(zkw_ctx* ctx, void* dst, const void* src, size_t bytes) {
    if (!ctx) return fail(ZKW_ERR_INVALID, "null context");
    if (bytes) HIP_TRY(ctx->copy_async(dst, src, bytes, hipMemcpyDeviceToDevice));
    return ZKW_OK;
}
void* zkw_device_shared_stream(int device_id) {
    static std::mutex mu;
    static std::map<int, hipStream_t>& m = *new std::map<int, hipStream_t>();
    std::lock_guard<std::mutex> g(mu);
    auto it = m.find(device_id);
    if (it != m.end()) return it->second;
    hipStream_t s = nullptr;
    if (hipSetDevice(device_id) != hipSuccess || hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return nullptr;
    m[device_id] = s;
    return s;
}
// an existing context joins a batch for the time its block's instances are synthesized as one of the batch's fibers (its stream must be idle)
void zkw_ctx_enter_batch(zkw_ctx* ctx, zkw_batch* b) {
    ctx->batch = b;
    ctx->stream = zkw_batch_stream(b);
}
// only the membership, not the stream: around a host callback that may use the context (a fiber must not park inside foreign frames)
zkw_batch* zkw_ctx_swap_batch(zkw_ctx* ctx, zkw_batch* b) {
    zkw_batch* old = ctx->batch;
    ctx->batch = b;
    return old;
}
void zkw_ctx_leave_batch(zkw_ctx* ctx, void* stream) {
    ctx->batch = nullptr;
    ctx->stream = static_cast<hipStream_t>(stream);
}

extern "C" int zkw_profile_enable(zkw_ctx* ctx, int on) {
    if (!ctx) return fail(ZKW_ERR_INVALID, "null context");
    ZKW_TRY(ctx->prof_collect());
    ctx->profiling = on != 0;
    return ZKW_OK;
}

extern "C" int zkw_profile_reset(zkw_ctx* ctx) {
    if (!ctx) return fail(ZKW_ERR_INVALID, "null context");
    ZKW_TRY(ctx->prof_collect());
    ctx->prof_totals.clear();
    return ZKW_OK;
}

extern "C" int zkw_profile_get(zkw_ctx* ctx, const char* kernel, double* total_ms, uint64_t* launches) {
    if (!ctx || !kernel) return fail(ZKW_ERR_INVALID, "null argument");
    ZKW_TRY(ctx->prof_collect());
    auto it = ctx->prof_totals.find(kernel);
    if (total_ms) *total_ms = it == ctx->prof_totals.end() ? 0.0 : it->second.first;
    if (launches) *launches = it == ctx->prof_totals.end() ? 0 : it->second.second;
    return ZKW_OK;
}

extern "C" int zkw_profile_names(zkw_ctx* ctx, char* buf, size_t buf_bytes) {
    if (!ctx || !buf || !buf_bytes) return fail(ZKW_ERR_INVALID, "null argument");
    ZKW_TRY(ctx->prof_collect());
    std::string all;
    for (auto& kv : ctx->prof_totals) { if (!all.empty()) all += ","; all += kv.first; }
    if (all.size() + 1 > buf_bytes) return fail(ZKW_ERR_INVALID, "buffer too small: need %zu bytes", all.size() + 1);
    memcpy(buf, all.c_str(), all.size() + 1);
    return ZKW_OK;
}

// The quad chain kernel's launch. Up to 16 384 chains (= 256 CUs x 4 SIMDs x 16 chains per wave) go out as 4-wave workgroups with an
// LDS request of more than half a CU's 160 KB: one workgroup per CU, its waves on the CU's four SIMDs — no two chain waves of the launch
// share a SIMD even when the launch arrives on a chip that another stream's fill kernels keep full (one-wave workgroups then land
// wherever a slot is free, some SIMDs get two or three chain waves, and the pass takes 1.5x or 2x as long: the slowest chain sets it).
// A second such launch cannot start on a CU that still holds the first one's workgroup, so overlapping chain passes queue up instead of
// doubling up. ZKW_CHAIN_WG4=0 restores one-wave workgroups.
// the LDS request that lets a CU take ONE 4-wave workgroup of `func` (more than half of the CU's LDS, from the device's own figure); -1 when
// the device refuses it: a device or partition mode without that much LDS per workgroup runs the one-wave form, which is correct everywhere
static int one_workgroup_per_cu_lds(const void* func, int slot /* 0 .. 3: which kernel */) {
    static std::atomic<int> lds_of[4][64];  // per kernel and device: 0 not tried, > 0 granted, -1 refused
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -1;
    int lds = lds_of[slot][dev].load();
    if (lds == 0) {
        int per_cu = 0;
        if (hipDeviceGetAttribute(&per_cu, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, dev) != hipSuccess || per_cu <= 0) per_cu = 160 * 1024;
        const int want = (per_cu / 2 + 4096) & ~1023;  // 84 KB of 160: one such workgroup per CU
        lds = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, want) == hipSuccess ? want : -1;
        if (lds < 0) (void)hipGetLastError();
        lds_of[slot][dev].store(lds);
    }
    return lds;
}
static int launch_chain_q4(hipStream_t st, const ChainJob* d_jobs, int n_jobs) {
    static const bool wg4 = [] { const char* e = getenv("ZKW_CHAIN_WG4"); return !(e && e[0] == '0'); }();
    if (wg4 && n_jobs > 64 && n_jobs <= 256 * 64) {
        const int lds = one_workgroup_per_cu_lds(reinterpret_cast<const void*>(&k_chain_full_q4x4), 0);
        if (lds > 0) {
            hipLaunchKernelGGL(k_chain_full_q4x4, dim3((n_jobs + 63) / 64), dim3(256), (size_t)lds, st, d_jobs, n_jobs);
            return ZKW_OK;
        }
    }
    hipLaunchKernelGGL(k_chain_full_q4, dim3((n_jobs + 15) / 16), dim3(64), 0, st, d_jobs, n_jobs);
    return ZKW_OK;
}
// the row forms as the chain service launches them: 16 chains per 4-wave workgroup, one workgroup per CU (up to 4 096 chains), so that the
// waves of launches that run side by side — a memory-queue stage next to a decommit stage next to a log-queue stage — never share a SIMD
static void launch_chain_full_rows(hipStream_t st, const ChainJob* d_jobs, int n_jobs) {
    static const bool wg4 = [] { const char* e = getenv("ZKW_CHAIN_WG4"); return !(e && e[0] == '0'); }();
    const int lds = wg4 && n_jobs > 4 && n_jobs <= 256 * 16 ? one_workgroup_per_cu_lds(reinterpret_cast<const void*>(&k_chain_full_x4), 1) : -1;
    if (lds > 0) hipLaunchKernelGGL(k_chain_full_x4, dim3((n_jobs + 15) / 16), dim3(256), (size_t)lds, st, d_jobs, n_jobs);
    else hipLaunchKernelGGL(k_chain_full, dim3((n_jobs + 3) / 4), dim3(64), 0, st, d_jobs, n_jobs);
}
static void launch_chain_log_rows(hipStream_t st, const LogChainJob* d_jobs, int n_jobs) {
    static const bool wg4 = [] { const char* e = getenv("ZKW_CHAIN_WG4"); const char* l = getenv("ZKW_CHAIN_LOG_WG4"); return !(e && e[0] == '0') && !(l && l[0] == '0'); }();
    const int lds = wg4 && n_jobs > 4 && n_jobs <= 256 * 16 ? one_workgroup_per_cu_lds(reinterpret_cast<const void*>(&k_chain_log_x4), 2) : -1;
    if (lds > 0) hipLaunchKernelGGL(k_chain_log_x4, dim3((n_jobs + 15) / 16), dim3(256), (size_t)lds, st, d_jobs, n_jobs);
    else hipLaunchKernelGGL(k_chain_log, dim3((n_jobs + 3) / 4), dim3(64), 0, st, d_jobs, n_jobs);
}

int zkw_launch_chain_full(hipStream_t st, const ChainJob* d_jobs, int n_jobs) {
    if (n_jobs >= 4096) return launch_chain_q4(st, d_jobs, n_jobs);
    launch_chain_full_rows(st, d_jobs, n_jobs);
    return ZKW_OK;
}
int zkw_launch_chain_log(hipStream_t st, const LogChainJob* d_jobs, int n_jobs) {
    // more than 4 096 log-queue chains (the three sorters' stages of 512 blocks arrive together: 4 608): several launches of the form that
    // takes a CU per 16 chains, rather than one launch of one-wave workgroups that settle on the SIMDs of other stages' chains (the jobs are
    // sorted by length: the first launch holds the long ones, the others are gone in milliseconds)
    for (int at = 0; at < n_jobs; at += 4096) launch_chain_log_rows(st, d_jobs + at, std::min(4096, n_jobs - at));
    return ZKW_OK;
}

// ------------------------------------------------------------------------------------------------ chain service
// Many contexts, few launches. A queue chain is serial — microseconds per item on ONE wave — and a launch of n chains costs
// what its longest chain costs as long as every wave has a SIMD to itself (4 096 chains in the row form). When many blocks
// are in flight (zkw_blocks_run: K blocks x ~6 builder threads, each with its own context and stream), their chain jobs
// would be K x 6 long-running one-wave kernels on as many streams: HIP multiplexes streams onto a handful of hardware
// queues, a queue runs in order, and the launches serialise (measured: 8 concurrent blocks = 1.8 blocks/s, hardly more than
// one). The service turns them into one launch: a context that opted in (zkw_set_chain_service) synchronises its stream,
// hands its jobs over and waits; a worker collects whatever arrives within a short window from ALL contexts of the device
// and launches it as one kernel on a HIGH-PRIORITY stream (HIP gives priority levels their own hardware queues — measured
// with tools/probe_hw_queues — so the long kernel never sits in front of anybody's short ones). Results are identical:
// the jobs are the same, only the launch they travel in differs.
struct ChainService {
    struct Batch {
        std::vector<ChainJob> full;
        std::vector<LogChainJob> log;
        int waiters = 0;
        bool done[2] = {false, false};  // [0] full-width chains, [1] log-queue chains
        int rc = ZKW_OK;
        std::string err;
        int key = 0;
        std::chrono::steady_clock::time_point open_since, last_arrival;
    };
    int device = 0;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    // the batches that still accept jobs, by KEY: 0 = whatever arrives (contexts without a tag), else one stage of one builder branch
    // (zkw_ctx::next_chain_key) — the jobs of that stage of ALL blocks in flight, which are equally long, travel in one launch
    std::map<int, std::shared_ptr<Batch>> open;
    int expected = 0;  // blocks in flight under zkw_blocks_run: a keyed batch with that many submitters leaves at once
    std::vector<std::thread> workers;
    bool stop = false;

    // ZKW_CHAIN_WORKERS (default 8): launches in flight at once — a worker stays with its batch until the batch's longest chain is done; each has ONE
    // high-priority stream (8 = the high-priority hardware queues under GPU_MAX_HW_QUEUES=8: two streams per worker, or more workers,
    // put two long launches on one queue — 12.0 / 9.8 / 8.3 blocks/s with 8 / 12 / 16 two-stream workers in round 5's first half).
    // ZKW_CHAIN_WINDOW_US (default 4000, the maximal age of a batch is ten times that): the silence that closes a batch. Round 5, second
    // half (ZKW_CHAIN_LOG=1 at 96 blocks in flight): the first few blocks to reach a stage left in four small batches, took the four
    // workers for the 1.1 s of a memory-queue chain, and the other ~90 blocks' chains waited for a worker — three rounds of 1.1 s for work
    // that fits one. Hence the keys and `expected`: a keyed batch waits for ALL blocks in flight (or ZKW_CHAIN_LONG_WINDOW_US of silence,
    // default 100 ms: a block may skip a stage or fail), so a stage is ONE launch.
    int n_workers = 8;
    long quiet_us = 4000, max_us = 40000, quiet_long_us = 100000;
    explicit ChainService(int dev) : device(dev) {
        if (const char* e = getenv("ZKW_CHAIN_WORKERS")) n_workers = std::max(1, std::min(32, atoi(e)));
        if (const char* e = getenv("ZKW_CHAIN_WINDOW_US")) { quiet_us = std::max(50L, atol(e)); max_us = 10 * quiet_us; }
        if (const char* e = getenv("ZKW_CHAIN_LONG_WINDOW_US")) quiet_long_us = std::max(50L, atol(e));
        for (int i = 0; i < n_workers; i++) workers.emplace_back([this] { run(); });
    }
    ~ChainService() {
        { std::lock_guard<std::mutex> g(mu); stop = true; }
        cv_work.notify_all();
        for (auto& t : workers) t.join();
    }
    void expect(int delta) {
        { std::lock_guard<std::mutex> g(mu); expected = std::max(0, expected + delta); }
        cv_work.notify_all();
    }
    // (under the lock) may this batch leave?
    bool ready(const Batch& b, std::chrono::steady_clock::time_point now) const {
        if (b.key != 0 && expected > 1) {
            if (b.waiters >= expected) return true;
            return now - b.last_arrival >= std::chrono::microseconds(quiet_long_us) || now - b.open_since >= std::chrono::microseconds(10 * quiet_long_us);
        }
        return now - b.last_arrival >= std::chrono::microseconds(quiet_us) || now - b.open_since >= std::chrono::microseconds(max_us);
    }
    int submit(const std::vector<ChainJob>* full, const std::vector<LogChainJob>* log, std::string* err, int key) {
        std::shared_ptr<Batch> b;
        {
            std::unique_lock<std::mutex> lk(mu);
            std::shared_ptr<Batch>& slot = open[key];
            const auto now = std::chrono::steady_clock::now();
            if (!slot) { slot = std::make_shared<Batch>(); slot->key = key; slot->open_since = now; }
            b = slot;
            if (full) b->full.insert(b->full.end(), full->begin(), full->end());
            if (log) b->log.insert(b->log.end(), log->begin(), log->end());
            b->waiters++;
            b->last_arrival = now;
            cv_work.notify_one();
            const int kind = full ? 0 : 1;
            cv_done.wait(lk, [&] { return b->done[kind]; });
            // read under the lock: a log-queue waiter is released before the batch's full-width chains have finished, and the
            // worker may still record their failure
            const int rc = b->rc;
            if (rc != ZKW_OK && err) *err = b->err;
            return rc;
        }
    }
    // a context that has nothing to hand over for stage `key` (no chains there, or chains short enough to run on its own stream): it still counts
    // as one of the `expected` submitters, or the blocks that do have chains would wait out the long window (measured: 100 ms per block in the
    // decommit sorter's prepare stage, where three of four synthetic blocks took the short path)
    void skip(int key) {
        if (key == 0) return;
        {
            std::lock_guard<std::mutex> g(mu);
            if (expected <= 1) return;
            std::shared_ptr<Batch>& slot = open[key];
            const auto now = std::chrono::steady_clock::now();
            if (!slot) { slot = std::make_shared<Batch>(); slot->key = key; slot->open_since = now; }
            slot->waiters++;
            slot->last_arrival = now;
        }
        cv_work.notify_one();
    }
    void run() {
        (void)hipSetDevice(device);
        hipStream_t st = nullptr, st_log = nullptr;  // st_log: only for a batch that carries both kinds (unkeyed jobs): they run side by side
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (hipStreamCreateWithPriority(&st, hipStreamNonBlocking, hi) != hipSuccess) st = nullptr;
        void* pin = nullptr;
        void* dev = nullptr;
        size_t cap = 0;
        for (;;) {
            std::shared_ptr<Batch> b;
            {
                std::unique_lock<std::mutex> lk(mu);
                for (;;) {
                    if (stop) break;
                    const auto now = std::chrono::steady_clock::now();
                    auto best = open.end();  // the ready batch that has been open longest
                    for (auto it = open.begin(); it != open.end(); ++it)
                        if (it->second && ready(*it->second, now) && (best == open.end() || it->second->open_since < best->second->open_since)) best = it;
                    if (best != open.end()) { b = best->second; open.erase(best); break; }
                    if (open.empty()) cv_work.wait(lk, [&] { return stop || !open.empty(); });
                    else cv_work.wait_for(lk, std::chrono::microseconds(200));
                }
                if (stop) break;
            }
            int rc = ZKW_OK;
            std::string err;
            auto fail_hip = [&](hipError_t e, const char* what) { if (e != hipSuccess && rc == ZKW_OK) { rc = ZKW_ERR_HIP; err = std::string(what) + ": " + hipGetErrorString(e); } };
            static const bool log_batches = getenv("ZKW_CHAIN_LOG") != nullptr;  // debugging aid: one line per batch (when it left, what it carries)
            const auto t_launch = std::chrono::steady_clock::now();
            size_t longest = 0;
            if (log_batches) for (const ChainJob& j : b->full) longest = std::max(longest, (size_t)j.n);
            const size_t bytes = b->full.size() * sizeof(ChainJob) + b->log.size() * sizeof(LogChainJob) + 256;
            const bool mixed = !b->full.empty() && !b->log.empty();
            if (mixed && !st_log && hipStreamCreateWithPriority(&st_log, hipStreamNonBlocking, hi) != hipSuccess) st_log = nullptr;
            hipStream_t sl = mixed ? st_log : st;  // the log-queue chains' stream
            if (!st || !sl) { rc = ZKW_ERR_HIP; err = "chain service: no stream"; }
            const bool nothing = b->full.empty() && b->log.empty();  // every submitter of the stage skipped it
            if (rc == ZKW_OK && !nothing && cap < bytes) {  // grow-only; the outgrown pair goes back to the allocation cache (no hipFree stall)
                const size_t want = bytes * 2;
                void *np = nullptr, *nd = nullptr;
                fail_hip(pin_malloc(&np, want), "hipHostMalloc");
                if (rc == ZKW_OK) fail_hip(dev_malloc(&nd, want), "hipMalloc");
                if (rc == ZKW_OK) {
                    if (pin) pin_free(pin);
                    if (dev) dev_free(dev);
                    pin = np; dev = nd; cap = want;
                } else if (np) {
                    pin_free(np);
                }
            }
            if (rc == ZKW_OK && !nothing) {
                char* hp = static_cast<char*>(pin);
                char* dp = static_cast<char*>(dev);
                const size_t off_log = (b->full.size() * sizeof(ChainJob) + 127) & ~(size_t)127;
                if (!b->full.empty()) memcpy(hp, b->full.data(), b->full.size() * sizeof(ChainJob));
                if (!b->log.empty()) memcpy(hp + off_log, b->log.data(), b->log.size() * sizeof(LogChainJob));
                fail_hip(hipMemcpyAsync(dp, hp, off_log + b->log.size() * sizeof(LogChainJob), hipMemcpyHostToDevice, st), "job upload");
                fail_hip(hipStreamSynchronize(st), "job upload");
                const int nf = (int)b->full.size(), nl = (int)b->log.size();
                if (rc == ZKW_OK && nl) launch_chain_log_rows(sl, reinterpret_cast<const LogChainJob*>(dp + off_log), nl);
                if (rc == ZKW_OK && nf) {
                    if (nf >= 4096) { const int lrc = launch_chain_q4(st, reinterpret_cast<const ChainJob*>(dp), nf); if (lrc != ZKW_OK && rc == ZKW_OK) { rc = lrc; err = "chain launch (quad form)"; } }
                    else launch_chain_full_rows(st, reinterpret_cast<const ChainJob*>(dp), nf);
                }
                fail_hip(hipGetLastError(), "chain launch");
                fail_hip(hipStreamSynchronize(sl), "chain batch (log queues)");
                {
                    std::lock_guard<std::mutex> g(mu);
                    if (rc != ZKW_OK) { b->rc = rc; b->err = err; }
                    b->done[1] = true;
                }
                cv_done.notify_all();
                fail_hip(hipStreamSynchronize(st), "chain batch");
            }
            {
                std::lock_guard<std::mutex> g(mu);
                if (rc != ZKW_OK) { b->rc = rc; b->err = err; }
                b->done[0] = b->done[1] = true;
            }
            cv_done.notify_all();
            if (log_batches) {
                static const auto t_zero = std::chrono::steady_clock::now();
                const auto us = [&](std::chrono::steady_clock::time_point t) { return (long)std::chrono::duration_cast<std::chrono::microseconds>(t - t_zero).count(); };
                fprintf(stderr, "[zkw chain] batch of %d waiters: %zu full-width chains (longest %zu items), %zu log chains; left at %ld us, done at %ld us\n", b->waiters, b->full.size(), longest,
                        b->log.size(), us(t_launch), us(std::chrono::steady_clock::now()));
            }
        }
        if (pin) pin_free(pin);
        if (dev) dev_free(dev);
        if (st) (void)hipStreamDestroy(st);
        if (st_log) (void)hipStreamDestroy(st_log);
    }
};
static std::mutex g_chain_services_mu;
// leaked on purpose, like the allocation cache: a static destructor would join the workers and destroy HIP streams while the
// HIP runtime itself is being torn down at process exit
static std::map<int, ChainService*>& g_chain_services = *new std::map<int, ChainService*>();
// One service per device. A batch carries both kinds of chains (full-width queues, log queues) as two kernels on two streams
// and completes per kind: a short log-queue chain does not wait for a long memory-queue chain that arrived in the same
// window. Keeping the kinds in ONE batch (rather than one service per kind) matters: fewer concurrent one-wave kernels,
// fewer chances that the dispatcher parks two of them on the same SIMD (each then runs 1.4x slower; measured).
static ChainService* chain_service_of(int device) {
    std::lock_guard<std::mutex> g(g_chain_services_mu);
    ChainService*& p = g_chain_services[device];
    if (!p) p = new ChainService(device);
    return p;
}

// zkw_blocks_run tells the device's service how many blocks are in flight (delta = +K before, -K after)
extern "C" int zkw_chain_service_expect(int device_id, int delta) {
    if (device_id < 0) return fail(ZKW_ERR_INVALID, "zkw_chain_service_expect: bad device");
    chain_service_of(device_id)->expect(delta);
    return ZKW_OK;
}
extern "C" int zkw_set_chain_tag(zkw_ctx* ctx, int tag) {
    if (!ctx || tag < 0 || tag > 1000) return fail(ZKW_ERR_INVALID, "zkw_set_chain_tag: bad argument");
    ctx->chain_tag = tag;
    ctx->chain_seq = 0;
    return ZKW_OK;
}

extern "C" int zkw_set_chain_service(zkw_ctx* ctx, int on) {
    if (!ctx) return fail(ZKW_ERR_INVALID, "null context");
    ctx->chain_service = on != 0;
    if (on) (void)chain_service_of(ctx->device);
    return ZKW_OK;
}

// hand the chains of one builder call over to the service and wait for them
static int chain_service_run(zkw_ctx* ctx, const std::vector<ChainJob>* full, const std::vector<LogChainJob>* log, const char* name, int key) {
    HIP_TRY(hipStreamSynchronize(ctx->stream));  // the jobs' inputs are produced on this context's stream
    const auto t0 = std::chrono::steady_clock::now();
    std::string err;
    const int rc = chain_service_of(ctx->device)->submit(full, log, &err, key);
    if (ctx->profiling) {  // wall time spent waiting for the shared launch (no HIP events: it runs on the service's stream)
        auto& t = ctx->prof_totals[std::string(name) + "(service)"];
        t.first += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        t.second += 1;
    }
    if (rc != ZKW_OK) return fail(rc, "chain service: %s", err.c_str());
    return ZKW_OK;
}

// ------------------------------------------------------------------------------------------------ device-level steps
// (all pointers are device pointers here)

int dev_encode(zkw_ctx* ctx, const zkw_mem_query* q, size_t n, u64* enc) {
    if (n == 0) return ZKW_OK;
    unsigned grid = blocks_for(n, 256);
    if (grid > 256 * 16) grid = 256 * 16;
    { Prof _p(ctx, "k_encode_mem"); ZKW_LAUNCH(ctx, k_encode_mem, grid, 256, q, n, enc); }
    return launch_check("k_encode_mem");
}

// Chains: one chain per 16-lane DPP row, 4 chains per wave, one wave per block. A single wave already
// issues a VALU instruction every ~2 cycles (measured 3.7 us per permutation step, flat from 1 to 4096
// concurrent chains), so throughput comes from giving each wave its own SIMD: up to 1024 waves.
int dev_chains(zkw_ctx* ctx, const std::vector<ChainJob>& jobs) {
    if (ctx->batched()) return jobs.empty() ? ZKW_OK : zkw_batch_chains(ctx->batch, jobs.data(), jobs.size(), nullptr, 0);  // one launch per stage of all blocks
    const int key = ctx->chain_service ? ctx->next_chain_key() : 0;  // (counted even when there is nothing to hash: equal stages of all blocks keep equal keys)
    if (jobs.empty()) { if (ctx->chain_service) chain_service_of(ctx->device)->skip(key); return ZKW_OK; }
    // the service is for chains whose LATENCY matters (a launch costs what its longest chain costs): a handful of items per chain — the
    // recursion queues of a block, eleven chains of a few records — is cheaper on the context's own stream than a rendezvous with every
    // other block in flight (measured at 96 blocks: that last stage waited 0.2 - 0.3 s for the slowest block)
    u64 longest = 0;
    for (const ChainJob& j : jobs) longest = std::max<u64>(longest, j.n);
    if (ctx->chain_service && longest > 64) return chain_service_run(ctx, &jobs, nullptr, "k_chain_full", key);
    if (ctx->chain_service) chain_service_of(ctx->device)->skip(key);
    ChainJob* d_jobs = nullptr;
    ZKW_TRY(ctx->upload("chain_jobs", jobs, &d_jobs));
    int n_jobs = (int)jobs.size();
    // auto: the row form has the lowest latency (10 us per step) and wins while every wave can have a SIMD to itself
    // (<= 4096 chains); the quad form packs 16 chains per wave (14.3 us up to 16 384 chains = one wave per SIMD, 21.3 us
    // with two waves per SIMD). The lane form (64 chains per wave, ~2.4x fewer VALU instructions per permutation, 36 us
    // per step) is never chosen automatically: measured on the bench's batch it loses to the quad form both alone
    // (1 418 vs 1 493 circuits/s) and next to another pipeline's fills (1 240 vs 1 790), DESIGN.md 3.2. The pair form (32 chains
    // per wave, round 3) is in between and not chosen either: 31 us per step on the bench's batch with half the quad form's waves
    // — the same pass time alone (1 506 vs 1 481 circuits/s), and next to the fills its longer pass costs more than the issue
    // slots it frees (1 738 vs 1 812)
    const int form = ctx->chain_form ? ctx->chain_form : (n_jobs >= 4096 ? 4 : 16);
    // the chain kernel may run on its own stream (e.g. one created with a CU mask): ordered after everything queued on
    // the context's stream so far, and the context's stream continues after it
    hipStream_t st = ctx->stream;
    if (ctx->chain_stream) {
        HIP_TRY(hipEventRecord(ctx->chain_ev_a, ctx->stream));
        HIP_TRY(hipStreamWaitEvent(ctx->chain_stream, ctx->chain_ev_a, 0));
        st = ctx->chain_stream;
    }
    const char* name = form == 16 ? "k_chain_full" : form == 4 ? "k_chain_full_q4" : form == 2 ? "k_chain_full_p2" : "k_chain_full_lane";
    {
        Prof _p(ctx, name);
        if (form == 16) hipLaunchKernelGGL(k_chain_full, dim3((n_jobs + 3) / 4), dim3(64), 0, st, d_jobs, n_jobs);
        else if (form == 4) ZKW_TRY(launch_chain_q4(st, d_jobs, n_jobs));
        else if (form == 2) hipLaunchKernelGGL(k_chain_full_p2, dim3((n_jobs + 31) / 32), dim3(64), 0, st, d_jobs, n_jobs);
        else hipLaunchKernelGGL(k_chain_full_lane, dim3((n_jobs + 63) / 64), dim3(64), 0, st, d_jobs, n_jobs);
        if (ctx->chain_stream) {  // the profiling events live on the context's stream: bring the kernel's end onto it first
            HIP_TRY(hipEventRecord(ctx->chain_ev_b, ctx->chain_stream));
            HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->chain_ev_b, 0));
        }
    }
    return launch_check(name);
}

int dev_fs(zkw_ctx* ctx, const std::vector<FsJob>& jobs, int state_w, int n_chal) {
    if (jobs.empty()) return ZKW_OK;
    FsJob* d_jobs = nullptr;
    ZKW_TRY(ctx->upload("fs_jobs", jobs, &d_jobs));
    int n = (int)jobs.size();
    { Prof _p(ctx, "k_fs_challenges"); ZKW_LAUNCH(ctx, k_fs_challenges, (n + 63) / 64, 64, d_jobs, n, state_w, n_chal); }
    return launch_check("k_fs_challenges");
}

template <int W, int REPS>
static int gp_launch(zkw_ctx* ctx, const GpSeg* d_segs, int n_segs, const GpTile* d_tiles, unsigned n_tiles,
                     u64* d_aggr) {
    { Prof _p(ctx, "k_gp_local"); ZKW_LAUNCH_T(ctx, (k_gp_local<W, REPS>), "k_gp_local", n_tiles, GP_BLOCK, d_segs, d_tiles, d_aggr); }
    ZKW_TRY(launch_check("k_gp_local"));
    { Prof _p(ctx, "k_gp_tiles"); ZKW_LAUNCH_T(ctx, (k_gp_tiles<REPS>), "k_gp_tiles", (n_segs * REPS + 63) / 64, 64, d_segs, n_segs, d_aggr); }
    ZKW_TRY(launch_check("k_gp_tiles"));
    { Prof _p(ctx, "k_gp_apply"); ZKW_LAUNCH_T(ctx, (k_gp_apply<REPS>), "k_gp_apply", n_tiles, GP_BLOCK, d_segs, d_tiles, d_aggr); }
    return launch_check("k_gp_apply");
}

// segs: rows/z/chal/n filled by the caller; first_tile/n_tiles filled here
int dev_grand_products(zkw_ctx* ctx, std::vector<GpSeg>& segs, int width, int n_reps) {
    std::vector<GpTile> tiles;
    for (size_t s = 0; s < segs.size(); s++) {
        segs[s].first_tile = (u32)tiles.size();
        segs[s].n_tiles = (u32)((segs[s].n + GP_TILE - 1) / GP_TILE);
        for (u32 t = 0; t < segs[s].n_tiles; t++) tiles.push_back(GpTile{(u32)s, t});
    }
    if (tiles.empty()) return ZKW_OK;
    GpSeg* d_segs = nullptr;
    GpTile* d_tiles = nullptr;
    u64* d_aggr = nullptr;
    ZKW_TRY(ctx->upload("gp_segs", segs, &d_segs));
    ZKW_TRY(ctx->upload("gp_tiles", tiles, &d_tiles));
    ZKW_TRY(ctx->scratch_t<u64>("gp_aggr", tiles.size() * 2, &d_aggr));
    const int n_segs = (int)segs.size();
    const unsigned n_tiles = (unsigned)tiles.size();
    if (width == 8 && n_reps == 2) return gp_launch<8, 2>(ctx, d_segs, n_segs, d_tiles, n_tiles, d_aggr);
    if (width == 8 && n_reps == 1) return gp_launch<8, 1>(ctx, d_segs, n_segs, d_tiles, n_tiles, d_aggr);
    if (width == 20 && n_reps == 2) return gp_launch<20, 2>(ctx, d_segs, n_segs, d_tiles, n_tiles, d_aggr);
    if (width == 20 && n_reps == 1) return gp_launch<20, 1>(ctx, d_segs, n_segs, d_tiles, n_tiles, d_aggr);
    return fail(ZKW_ERR_INVALID, "grand product: unsupported width %d / repetitions %d", width, n_reps);
}

// ------------------------------------------------------------------------------------------------ L1 entry points
extern "C" int zkw_encode_memory_queries(zkw_ctx* ctx, const zkw_mem_query* q, size_t n, uint64_t* enc) {
    if (!ctx || (n && (!q || !enc))) return fail(ZKW_ERR_INVALID, "zkw_encode_memory_queries: null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    const zkw_mem_query* d_q = nullptr;
    u64* d_enc = nullptr;
    ZKW_TRY(ctx->in("enc_q", q, n, &d_q));
    ZKW_TRY(ctx->out("enc_out", enc, n * 8, &d_enc));
    ZKW_TRY(dev_encode(ctx, d_q, n, d_enc));
    ZKW_TRY(ctx->finish_out(enc, d_enc, n * 8));
    return ctx->sync_if_host();
}

extern "C" int zkw_queue_push_chain_full_batch(zkw_ctx* ctx, const uint64_t* enc, const uint64_t* offsets,
                                               size_t n_queues, const uint64_t* tails_in, uint64_t* tails) {
    if (!ctx || !offsets) return fail(ZKW_ERR_INVALID, "zkw_queue_push_chain_full_batch: null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    for (size_t k = 0; k < n_queues; k++)
        if (offsets[k + 1] < offsets[k]) return fail(ZKW_ERR_INVALID, "offsets must be non-decreasing");
    const size_t total = n_queues ? offsets[n_queues] - offsets[0] : 0;
    if (total && (!enc || !tails)) return fail(ZKW_ERR_INVALID, "null enc/tails");
    const size_t base = n_queues ? offsets[0] : 0;
    const u64* d_enc = nullptr;
    const u64* d_tin = nullptr;
    u64* d_tails = nullptr;
    ZKW_TRY(ctx->in("chain_enc", enc + base * 8, total * 8, &d_enc));
    if (tails_in) ZKW_TRY(ctx->in("chain_tin", tails_in, n_queues * 12, &d_tin));
    ZKW_TRY(ctx->out("chain_tails", tails + base * 12, total * 12, &d_tails));
    std::vector<ChainJob> jobs(n_queues);
    for (size_t k = 0; k < n_queues; k++) {
        size_t lo = offsets[k] - base;
        jobs[k].enc = d_enc + lo * 8;
        jobs[k].tails = d_tails + lo * 12;
        jobs[k].tail_in = d_tin ? d_tin + 12 * k : nullptr;
        jobs[k].n = offsets[k + 1] - offsets[k];
    }
    ZKW_TRY(dev_chains(ctx, jobs));
    ZKW_TRY(ctx->finish_out(tails + base * 12, d_tails, total * 12));
    return ctx->sync_if_host();
}

extern "C" int zkw_queue_push_chain_full(zkw_ctx* ctx, const uint64_t* enc, size_t n, const uint64_t tail_in[12],
                                         uint64_t* tails) {
    uint64_t offsets[2] = {0, n};
    return zkw_queue_push_chain_full_batch(ctx, enc, offsets, 1, tail_in, tails);
}

extern "C" int zkw_fs_challenges(zkw_ctx* ctx, const uint64_t* tail_u, uint32_t len_u, const uint64_t* tail_s,
                                 uint32_t len_s, int state_w, int n_chal, uint64_t* out) {
    if (!ctx || !tail_u || !tail_s || !out) return fail(ZKW_ERR_INVALID, "zkw_fs_challenges: null argument");
    if ((state_w != 4 && state_w != 12) || n_chal < 1 || n_chal > 64)
        return fail(ZKW_ERR_INVALID, "zkw_fs_challenges: state_w must be 4 or 12, 1 <= n_chal <= 64");
    HIP_TRY(hipSetDevice(ctx->device));
    const u64 *d_u = nullptr, *d_s = nullptr;
    u64* d_out = nullptr;
    ZKW_TRY(ctx->in("fs_u", tail_u, (size_t)state_w, &d_u));
    ZKW_TRY(ctx->in("fs_s", tail_s, (size_t)state_w, &d_s));
    ZKW_TRY(ctx->out("fs_out", out, (size_t)2 * n_chal, &d_out));
    std::vector<FsJob> jobs(1);
    jobs[0] = FsJob{d_u, d_s, len_u, len_s, d_out};
    ZKW_TRY(dev_fs(ctx, jobs, state_w, n_chal));
    ZKW_TRY(ctx->finish_out(out, d_out, (size_t)2 * n_chal));
    return ctx->sync_if_host();
}

extern "C" int zkw_grand_product_chains(zkw_ctx* ctx, const uint64_t* lhs, const uint64_t* rhs, size_t n, int width,
                                        const uint64_t* challenges, int n_reps, uint64_t* lhs_z, uint64_t* rhs_z) {
    if (!ctx || !challenges || (n && (!lhs || !rhs || !lhs_z || !rhs_z)))
        return fail(ZKW_ERR_INVALID, "zkw_grand_product_chains: null argument");
    if ((width != 8 && width != 20) || (n_reps != 1 && n_reps != 2))
        return fail(ZKW_ERR_INVALID, "zkw_grand_product_chains: width must be 8 or 20, n_reps 1 or 2");
    HIP_TRY(hipSetDevice(ctx->device));
    if (n == 0) return ZKW_OK;
    const u64 *d_l = nullptr, *d_r = nullptr, *d_c = nullptr;
    u64 *d_lz = nullptr, *d_rz = nullptr;
    ZKW_TRY(ctx->in("gp_lhs", lhs, n * width, &d_l));
    ZKW_TRY(ctx->in("gp_rhs", rhs, n * width, &d_r));
    ZKW_TRY(ctx->in("gp_chal", challenges, (size_t)n_reps * (width + 1), &d_c));
    ZKW_TRY(ctx->out("gp_lz", lhs_z, n * n_reps, &d_lz));
    ZKW_TRY(ctx->out("gp_rz", rhs_z, n * n_reps, &d_rz));
    std::vector<GpSeg> segs(2);
    segs[0] = GpSeg{d_l, d_lz, d_c, n, 0, 0};
    segs[1] = GpSeg{d_r, d_rz, d_c, n, 0, 0};
    ZKW_TRY(dev_grand_products(ctx, segs, width, n_reps));
    ZKW_TRY(ctx->finish_out(lhs_z, d_lz, n * n_reps));
    ZKW_TRY(ctx->finish_out(rhs_z, d_rz, n * n_reps));
    ZKW_TRY(ctx->sync_if_host());
    if (ctx->ptr_mode == ZKW_PTR_HOST)
        for (int r = 0; r < n_reps; r++)
            if (lhs_z[(size_t)r * n + n - 1] != rhs_z[(size_t)r * n + n - 1])
                return fail(ZKW_ERR_CHECK_FAILED, "grand products differ in repetition %d (utils.rs:685-696)", r);
    return ZKW_OK;
}

extern "C" int zkw_encode_log_queries(zkw_ctx* ctx, const zkw_log_query* q, size_t n, const uint32_t* ext_ts,
                                      uint64_t* enc) {
    if (!ctx || (n && (!q || !enc))) return fail(ZKW_ERR_INVALID, "zkw_encode_log_queries: null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    if (n == 0) return ZKW_OK;
    const zkw_log_query* d_q = nullptr;
    const u32* d_e = nullptr;
    u64* d_enc = nullptr;
    ZKW_TRY(ctx->in("lenc_q", q, n, &d_q));
    if (ext_ts) ZKW_TRY(ctx->in("lenc_ts", ext_ts, n, &d_e));
    ZKW_TRY(ctx->out("lenc_out", enc, n * 20, &d_enc));
    { Prof _p(ctx, "k_encode_log"); ZKW_LAUNCH(ctx, k_encode_log, blocks_for(n, 256), 256, d_q, n, d_e, d_enc); }
    ZKW_TRY(launch_check("k_encode_log"));
    ZKW_TRY(ctx->finish_out(enc, d_enc, n * 20));
    return ctx->sync_if_host();
}

extern "C" int zkw_encode_decommit_queries(zkw_ctx* ctx, const zkw_decommit_query* q, size_t n, uint64_t* enc) {
    if (!ctx || (n && (!q || !enc))) return fail(ZKW_ERR_INVALID, "zkw_encode_decommit_queries: null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    if (n == 0) return ZKW_OK;
    const zkw_decommit_query* d_q = nullptr;
    u64* d_enc = nullptr;
    ZKW_TRY(ctx->in("denc_q", q, n, &d_q));
    ZKW_TRY(ctx->out("denc_out", enc, n * 8, &d_enc));
    { Prof _p(ctx, "k_encode_decommit"); ZKW_LAUNCH(ctx, k_encode_decommit, blocks_for(n, 256), 256, d_q, n, d_enc); }
    ZKW_TRY(launch_check("k_encode_decommit"));
    ZKW_TRY(ctx->finish_out(enc, d_enc, n * 8));
    return ctx->sync_if_host();
}

// device-level: rounds 1-2 of every item in parallel, then one serial permutation per item and queue
int dev_log_chains(zkw_ctx* ctx, const u64* d_enc, size_t total, std::vector<LogChainJob>& jobs) {
    const int key = ctx->chain_service && !ctx->batched() ? ctx->next_chain_key() : 0;
    if (total == 0 || jobs.empty()) { if (ctx->chain_service && !ctx->batched()) chain_service_of(ctx->device)->skip(key); return ZKW_OK; }
    u64* d_pre = nullptr;
    ZKW_TRY(ctx->scratch_t<u64>("log_pre", total * 4, &d_pre));
    { Prof _p(ctx, "k_log_prehash"); ZKW_LAUNCH(ctx, k_log_prehash, blocks_for(total, 128), 128, d_enc, total, d_pre); }
    ZKW_TRY(launch_check("k_log_prehash"));
    for (auto& j : jobs) j.pre = d_pre + (j.enc - d_enc) / 20 * 4;
    if (ctx->batched()) return zkw_batch_chains(ctx->batch, nullptr, 0, jobs.data(), jobs.size());
    if (ctx->chain_service) return chain_service_run(ctx, nullptr, &jobs, "k_chain_log", key);
    LogChainJob* d_jobs = nullptr;
    ZKW_TRY(ctx->upload("log_chain_jobs", jobs, &d_jobs));
    const int n_jobs = (int)jobs.size();
    { Prof _p(ctx, "k_chain_log"); hipLaunchKernelGGL(k_chain_log, dim3((n_jobs + 3) / 4), dim3(64), 0, ctx->stream, d_jobs, n_jobs); }
    return launch_check("k_chain_log");
}

extern "C" int zkw_queue_push_chain_log_batch(zkw_ctx* ctx, const uint64_t* enc, const uint64_t* offsets,
                                              size_t n_queues, const uint64_t* tails_in, uint64_t* old_tails,
                                              uint64_t* new_tails) {
    if (!ctx || !offsets) return fail(ZKW_ERR_INVALID, "zkw_queue_push_chain_log_batch: null argument");
    HIP_TRY(hipSetDevice(ctx->device));
    for (size_t k = 0; k < n_queues; k++)
        if (offsets[k + 1] < offsets[k]) return fail(ZKW_ERR_INVALID, "offsets must be non-decreasing");
    const size_t base = n_queues ? offsets[0] : 0, total = n_queues ? offsets[n_queues] - base : 0;
    if (total && (!enc || !new_tails)) return fail(ZKW_ERR_INVALID, "null enc/new_tails");
    const u64 *d_enc = nullptr, *d_tin = nullptr;
    u64 *d_old = nullptr, *d_new = nullptr;
    ZKW_TRY(ctx->in("lchain_enc", enc + base * 20, total * 20, &d_enc));
    if (tails_in) ZKW_TRY(ctx->in("lchain_tin", tails_in, n_queues * 4, &d_tin));
    if (old_tails) ZKW_TRY(ctx->out("lchain_old", old_tails + base * 4, total * 4, &d_old));
    ZKW_TRY(ctx->out("lchain_new", new_tails + base * 4, total * 4, &d_new));
    std::vector<LogChainJob> jobs(n_queues);
    for (size_t k = 0; k < n_queues; k++) {
        const size_t lo = offsets[k] - base;
        jobs[k] = LogChainJob{d_enc + lo * 20, nullptr, d_old ? d_old + lo * 4 : nullptr, d_new + lo * 4,
                              d_tin ? d_tin + 4 * k : nullptr, offsets[k + 1] - offsets[k]};
    }
    ZKW_TRY(dev_log_chains(ctx, d_enc, total, jobs));
    if (old_tails) ZKW_TRY(ctx->finish_out(old_tails + base * 4, d_old, total * 4));
    ZKW_TRY(ctx->finish_out(new_tails + base * 4, d_new, total * 4));
    return ctx->sync_if_host();
}

extern "C" int zkw_queue_push_chain_log(zkw_ctx* ctx, const uint64_t* enc, size_t n, const uint64_t tail_in[4],
                                        uint64_t* old_tails, uint64_t* new_tails) {
    uint64_t offsets[2] = {0, n};
    return zkw_queue_push_chain_log_batch(ctx, enc, offsets, 1, tail_in, old_tails, new_tails);
}

// ------------------------------------------------------------------------------------------------ RAM builder
struct zkw_ram_witness {
    zkw_ctx* ctx = nullptr;
    std::vector<uint64_t> offsets;       // n_blocks + 1, rebased to 0
    std::vector<uint64_t> inst_offsets;  // n_blocks + 1
    uint32_t capacity = 0;
    size_t total = 0, n_instances = 0;
    // owned device arrays
    // The sorted queue is kept as the sorting permutation (4 B per query instead of a 48 B copy): sorted item i =
    // unsorted_q[perm[i]], perm indexes the whole batch. ZKW_RAM_SORTED_QUERIES is gathered on first access.
    zkw_mem_query* sorted_q = nullptr;
    bool sorted_valid = false;
    u32* perm = nullptr;
    // The queue in its original order. Device-pointer mode: the CALLER's array (it must stay valid and unchanged
    // while the witness is synthesized or read); host-pointer mode: a copy owned by the witness. The builder keeps no
    // encodings (64 B per item and side): the chain, grand-product and fill kernels encode the 48-byte queries on the
    // fly; the [total][8] arrays of the C ABI are materialised on first access (ram_encodings).
    const zkw_mem_query* unsorted_q = nullptr;
    zkw_mem_query* owned_q = nullptr;
    u64 *unsorted_enc = nullptr, *sorted_enc = nullptr;
    bool enc_valid = false;
    // queue tails, compact: capacity words of every tail [total][4] + full tails at instance ends [n_instances][12];
    // the full [total][12] arrays of the C ABI are expanded on first access (ram_full_tails)
    u64 *unsorted_caps = nullptr, *sorted_caps = nullptr, *unsorted_marks = nullptr, *sorted_marks = nullptr;
    u64 *unsorted_tails = nullptr, *sorted_tails = nullptr;
    bool tails_valid = false;
    // grand-product chains: the builder only needs them at instance boundaries and the fills only inside the block
    // being filled, so they live in a window (zbuf_*, zcap items) that is recomputed per group of blocks / per synthesis
    // call (16 B per query and side instead of 32 B resident); the [total] arrays of the C ABI are computed on first access
    u64 *challenges = nullptr, *lhs_z = nullptr, *rhs_z = nullptr;
    bool z_valid = false;
    u64 *zbuf_l = nullptr, *zbuf_r = nullptr;
    zkw_mem_query* sq_win = nullptr;  // the sorted queries of the blocks being synthesized, gathered per synthesis call
    size_t zcap = 0, sqcap = 0;       // capacity of the chain windows / of the sorted window, in queue items
    zkw_ram_instance* instances = nullptr;
    u32* nondet_counts = nullptr;
    u64 *compact_forms = nullptr, *public_inputs = nullptr;  // [n_instances][18], [n_instances][4]

    void release() {
        void* ptrs[] = {sorted_q, perm, owned_q, unsorted_enc, sorted_enc, unsorted_caps, sorted_caps, unsorted_marks, sorted_marks,
                        unsorted_tails, sorted_tails, challenges,
                        lhs_z,    rhs_z,        instances,  nondet_counts, compact_forms, public_inputs, zbuf_l, zbuf_r, sq_win};
        for (void* p : ptrs)
            if (p) dev_free(p);
        sorted_q = nullptr;
        sorted_valid = false;
        perm = nullptr;
        owned_q = nullptr;
        unsorted_q = nullptr;
        enc_valid = false;
        unsorted_enc = sorted_enc = unsorted_tails = sorted_tails = challenges = lhs_z = rhs_z = nullptr;
        unsorted_caps = sorted_caps = unsorted_marks = sorted_marks = nullptr;
        tails_valid = false;
        z_valid = false;
        zbuf_l = zbuf_r = nullptr;
        sq_win = nullptr;
        zcap = sqcap = 0;
        instances = nullptr;
        nondet_counts = nullptr;
        compact_forms = public_inputs = nullptr;
    }
};

extern "C" void zkw_ram_witness_free(zkw_ram_witness* w);
static int ram_alloc(zkw_ram_witness* w, size_t n_blocks) {
    const size_t t = w->total, ni = w->n_instances;
    HIP_TRY(dev_malloc((void**)&w->perm, (t + 1) * sizeof(u32)));
    HIP_TRY(dev_malloc((void**)&w->unsorted_caps, (t + 1) * 4 * sizeof(u64)));
    HIP_TRY(dev_malloc((void**)&w->sorted_caps, (t + 1) * 4 * sizeof(u64)));
    HIP_TRY(dev_malloc((void**)&w->unsorted_marks, (ni + 1) * 12 * sizeof(u64)));
    HIP_TRY(dev_malloc((void**)&w->sorted_marks, (ni + 1) * 12 * sizeof(u64)));
    HIP_TRY(dev_malloc((void**)&w->challenges, (n_blocks + 1) * 18 * sizeof(u64)));
    {   // window of the grand-product chains: whole blocks, about 1 GB per side, at least the largest block
        size_t max_block = 0;
        for (size_t b = 0; b + 1 < w->offsets.size(); b++) max_block = std::max(max_block, (size_t)(w->offsets[b + 1] - w->offsets[b]));
        size_t window = (size_t)1 << 26, sq_window = (size_t)1 << 22;  // 2 x 1 GB of chains (few builder groups), 192 MB of sorted queries
        if (const char* e = getenv("ZKW_Z_WINDOW_ITEMS")) window = sq_window = (size_t)strtoull(e, nullptr, 10);  // tests: force small groups
        w->zcap = std::max(max_block, std::min(t, window));
        w->sqcap = std::max(max_block, std::min(t, sq_window));
        HIP_TRY(dev_malloc((void**)&w->zbuf_l, (w->zcap + 1) * 2 * sizeof(u64)));
        HIP_TRY(dev_malloc((void**)&w->zbuf_r, (w->zcap + 1) * 2 * sizeof(u64)));
        HIP_TRY(dev_malloc((void**)&w->sq_win, (w->sqcap + 1) * sizeof(zkw_mem_query)));
    }
    HIP_TRY(dev_malloc((void**)&w->instances, (ni + 1) * sizeof(zkw_ram_instance)));
    HIP_TRY(dev_malloc((void**)&w->nondet_counts, (ni + 1) * sizeof(u32)));
    HIP_TRY(dev_malloc((void**)&w->compact_forms, (ni + 1) * COMPACT_FORM_LEN * sizeof(u64)));
    HIP_TRY(dev_malloc((void**)&w->public_inputs, (ni + 1) * 4 * sizeof(u64)));
    return ZKW_OK;
}

// the block a query position belongs to: largest b with offsets[b] <= i
__device__ __forceinline__ u32 block_of_position(const u64* __restrict__ offsets, int n_blocks, size_t i) {
    int lo = 0, hi = n_blocks;
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (offsets[mid] <= i) lo = mid; else hi = mid;
    }
    return (u32)lo;
}

// Widths of the sort key's fields in this batch: max timestamp / page / index, so that the radix sort only walks
// the bits that are in use.
static __device__ __forceinline__ void k_ram_key_ranges(const VB& vb, const zkw_mem_query* __restrict__ q, size_t n, u32* __restrict__ maxima) {
    u32 t = 0, p = 0, x = 0;
    for (size_t i = (size_t)vb.x * blockDim.x + threadIdx.x; i < n; i += (size_t)vb.nx * blockDim.x) {
        t = max(t, q[i].timestamp);
        p = max(p, q[i].page);
        x = max(x, q[i].index);
    }
    for (int o = 32; o > 0; o >>= 1) {
        t = max(t, (u32)__shfl_xor((int)t, o));
        p = max(p, (u32)__shfl_xor((int)p, o));
        x = max(x, (u32)__shfl_xor((int)x, o));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMax(maxima + 0, t);
        atomicMax(maxima + 1, p);
        atomicMax(maxima + 2, x);
    }
}

// (block, page, index, timestamp) packed into one word, most significant first
static __device__ __forceinline__ void k_ram_packed_keys(const VB& vb, const zkw_mem_query* __restrict__ q, size_t n, const u64* __restrict__ offsets,
                                  int n_blocks, unsigned bits_p, unsigned bits_i, unsigned bits_t,
                                  u64* __restrict__ key, u32* __restrict__ iota) {
    size_t i = (size_t)vb.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 k = n_blocks > 1 ? block_of_position(offsets, n_blocks, i) : 0;
    k = (k << bits_p) | q[i].page;
    k = (k << bits_i) | q[i].index;
    k = (k << bits_t) | q[i].timestamp;
    key[i] = k;
    iota[i] = (u32)i;
}

// (block, page, index) of the items in their current order `perm`, packed into one word
static __device__ __forceinline__ void k_ram_packed_cells(const VB& vb, const u64* __restrict__ cell, const u32* __restrict__ perm, size_t n,
                                   const u64* __restrict__ offsets, int n_blocks, unsigned bits_p, unsigned bits_i,
                                   u64* __restrict__ key) {
    size_t i = (size_t)vb.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u32 src = perm[i];
    const u64 c = cell[src];
    u64 k = n_blocks > 1 ? block_of_position(offsets, n_blocks, src) : 0;
    k = (k << bits_p) | (c >> 32);
    k = (k << bits_i) | (c & 0xFFFFFFFFu);
    key[i] = k;
}

static __device__ __forceinline__ void k_block_ids(const VB& vb, const u64* __restrict__ offsets, int n_blocks, size_t n, u32* __restrict__ ids) {
    size_t i = (size_t)vb.x * blockDim.x + threadIdx.x;
    if (i < n) ids[i] = block_of_position(offsets, n_blocks, i);
}

static unsigned bits_in_use(u32 x) { return x ? 32u - (unsigned)__builtin_clz(x) : 0u; }

// The sorting permutation for all blocks at once: order (block, page, index, timestamp), W/ram_permutation.rs:48-53
// per block. The key is as wide as this batch makes it, and the sort takes the cheapest of three routes:
//   one   - all four fields fit 64 bits: one radix sort of the packed word over the bits in use;
//   two   - they do not, but (block, page, index) do: timestamps first, then the packed cell word (stable);
//   three - full-width pages and indices in a multi-block batch: timestamp, cell, block id, each stable.
// `work_a` / `work_b`: two device areas of 32 bytes per query that nothing else uses until the sort is over (the
// builder passes the capacity-word arrays, which the chains fill afterwards); the radix temporary falls back to
// the context scratch when it does not fit (tiny batches: its histograms dominate).
static int ram_sort(zkw_ctx* ctx, const zkw_mem_query* d_q, size_t total, const std::vector<uint64_t>& offsets,
                    void* work_a, void* work_b, u32** perm_out) {
    const size_t n_blocks = offsets.size() - 1;
    size_t tmp_bytes = radix_temp_bytes(total);
    // work_a: ts | k32 | v0 | v1 (4 x u32) | cell | k64a (2 x u64) = 32 bytes per query
    u32* ts = static_cast<u32*>(work_a);
    u32 *k32 = ts + total, *v0 = k32 + total, *v1 = v0 + total;
    u64* cell = reinterpret_cast<u64*>(v1 + total);
    u64* k64a = cell + total;
    // work_b: k64b | radix temporary
    u64* k64b = static_cast<u64*>(work_b);
    void* tmp = nullptr;
    const size_t tmp_off = (total * 8 + 255) & ~(size_t)255;
    if (tmp_off + tmp_bytes <= total * 32) tmp = static_cast<char*>(work_b) + tmp_off;
    else ZKW_TRY(ctx->scratch("sort_tmp", tmp_bytes + 256, &tmp));
    const unsigned grid = blocks_for(total, 256);

    void* d_max_v = nullptr;
    ZKW_TRY(ctx->scratch("sort_max", 256, &d_max_v));
    u32* d_max = static_cast<u32*>(d_max_v);
    HIP_TRY(ctx->memset_async(d_max, 0, 16));
    { Prof _p(ctx, "k_ram_key_ranges");
      const unsigned g = grid < 4096 ? grid : 4096;
      ZKW_LAUNCH(ctx, k_ram_key_ranges, g, 256, d_q, total, d_max); }
    ZKW_TRY(launch_check("k_ram_key_ranges"));
    u32 maxima[4];
    ZKW_TRY(ctx->read_small(maxima, d_max, 16));
    const unsigned bits_t = bits_in_use(maxima[0]), bits_p = bits_in_use(maxima[1]), bits_i = bits_in_use(maxima[2]);
    unsigned bits_b = 0;
    while ((1ull << bits_b) < n_blocks) bits_b++;
    u64* d_off = nullptr;
    if (n_blocks > 1) ZKW_TRY(ctx->upload("sort_off", offsets, &d_off));

    if (bits_b + bits_p + bits_i + bits_t <= 64) {
        const unsigned bits = bits_b + bits_p + bits_i + bits_t;
        { Prof _p(ctx, "k_ram_packed_keys");
          ZKW_LAUNCH(ctx, k_ram_packed_keys, grid, 256, d_q, total, d_off, (int)n_blocks, bits_p, bits_i, bits_t, k64a, v0); }
        ZKW_TRY(launch_check("k_ram_packed_keys"));
        { Prof _p(ctx, "radix_sort"); ZKW_TRY(radix_sort_pairs<u64>(ctx, tmp, tmp_bytes, k64a, k64b, v0, v1, total, bits ? bits : 1)); }
        *perm_out = v1;
        return ZKW_OK;
    }

    { Prof _p(ctx, "k_ram_sort_keys"); ZKW_LAUNCH(ctx, k_ram_sort_keys, grid, 256, d_q, total, ts, cell, v0, (const u64*)nullptr, 0); }
    ZKW_TRY(launch_check("k_ram_sort_keys"));
    // pass 1: timestamp
    { Prof _p(ctx, "radix_sort"); ZKW_TRY(radix_sort_pairs<u32>(ctx, tmp, tmp_bytes, ts, k32, v0, v1, total, bits_t ? bits_t : 1)); }
    if (bits_b + bits_p + bits_i <= 64) {
        // pass 2: (block, page, index) of the ts-sorted items
        const unsigned bits = bits_b + bits_p + bits_i;
        { Prof _p(ctx, "k_ram_packed_cells");
          ZKW_LAUNCH(ctx, k_ram_packed_cells, grid, 256, cell, v1, total, d_off, (int)n_blocks, bits_p, bits_i, k64a); }
        ZKW_TRY(launch_check("k_ram_packed_cells"));
        { Prof _p(ctx, "radix_sort"); ZKW_TRY(radix_sort_pairs<u64>(ctx, tmp, tmp_bytes, k64a, k64b, v1, v0, total, bits ? bits : 1)); }
        *perm_out = v0;
        return ZKW_OK;
    }
    // pass 2: cell of the ts-sorted items
    { Prof _p(ctx, "k_gather_u64_by_u32"); ZKW_LAUNCH(ctx, k_gather_u64_by_u32, grid, 256, cell, v1, total, k64a); }
    ZKW_TRY(launch_check("k_gather_u64_by_u32"));
    { Prof _p(ctx, "radix_sort"); ZKW_TRY(radix_sort_pairs<u64>(ctx, tmp, tmp_bytes, k64a, k64b, v1, v0, total, 64)); }
    // pass 3: block id, so that each block's items end up contiguous again (only multi-block batches get here)
    { Prof _p(ctx, "k_block_ids"); ZKW_LAUNCH(ctx, k_block_ids, grid, 256, d_off, (int)n_blocks, total, ts); }
    ZKW_TRY(launch_check("k_block_ids"));
    // ts[] now holds block ids in ORIGINAL order; gather them through the current permutation
    { Prof _p(ctx, "k_gather_u32_by_u32"); ZKW_LAUNCH(ctx, k_gather_u32_by_u32, grid, 256, ts, v0, total, k32); }
    ZKW_TRY(launch_check("k_gather_u32_by_u32"));
    { Prof _p(ctx, "radix_sort"); ZKW_TRY(radix_sort_pairs<u32>(ctx, tmp, tmp_bytes, k32, ts, v0, v1, total, bits_b)); }
    *perm_out = v1;
    return ZKW_OK;
}

// grand-product chains of blocks [b0, b1): block b's [2][n_b] chains at dst + 2 * (offsets[b] - offsets[b0])
static int ram_gp_blocks(zkw_ctx* ctx, const zkw_ram_witness* w, size_t b0, size_t b1, u64* dst_l, u64* dst_r) {
    std::vector<GpSeg> segs;
    segs.reserve(2 * (b1 - b0));
    const size_t base = w->offsets[b0];
    for (size_t b = b0; b < b1; b++) {
        const size_t lo = w->offsets[b], n = w->offsets[b + 1] - lo;
        segs.push_back(GpSeg{nullptr, dst_l + 2 * (lo - base), w->challenges + 18 * b, n, 0, 0, w->unsorted_q + lo});
        segs.push_back(GpSeg{nullptr, dst_r + 2 * (lo - base), w->challenges + 18 * b, n, 0, 0, w->unsorted_q, w->perm + lo});
    }
    return dev_grand_products(ctx, segs, 8, 2);
}

static int ram_run(zkw_ctx* ctx, zkw_ram_witness* w, const zkw_mem_query* d_q, const uint32_t* n_nondet) {
    const size_t n_blocks = w->offsets.size() - 1, total = w->total;
    // K1 — src/witness/oracle.rs:894-903 encodes each query as it is pushed; here every consumer encodes on the fly
    if (ctx->ptr_mode == ZKW_PTR_DEVICE) {
        w->unsorted_q = d_q;
    } else {  // d_q is the context's staging copy, which the next call overwrites
        if (!w->owned_q) HIP_TRY(dev_malloc((void**)&w->owned_q, (total + 1) * sizeof(zkw_mem_query)));
        HIP_TRY(ctx->copy_async(w->owned_q, d_q, total * sizeof(zkw_mem_query), hipMemcpyDeviceToDevice));
        w->unsorted_q = w->owned_q;
    }
    // K7 (sorted side)
    u32* perm = nullptr;
    ZKW_TRY(ram_sort(ctx, d_q, total, w->offsets, w->unsorted_caps, w->sorted_caps, &perm));
    w->tails_valid = false;
    w->enc_valid = false;
    w->sorted_valid = false;
    // the permutation lives in the sort's scratch (the capacity-word arrays the chains are about to fill): keep a copy
    HIP_TRY(ctx->copy_async(w->perm, perm, total * sizeof(u32), hipMemcpyDeviceToDevice));
    // K2: 2 chains per block, all in one launch
    std::vector<ChainJob> chains;
    chains.reserve(2 * n_blocks);
    for (size_t b = 0; b < n_blocks; b++) {
        const size_t lo = w->offsets[b], n = w->offsets[b + 1] - lo;
        const size_t io = w->inst_offsets[b];
        chains.push_back(ChainJob{nullptr, nullptr, nullptr, n, w->unsorted_caps + 4 * lo, w->unsorted_marks + 12 * io, w->capacity, w->unsorted_q + lo});
        chains.push_back(ChainJob{nullptr, nullptr, nullptr, n, w->sorted_caps + 4 * lo, w->sorted_marks + 12 * io, w->capacity, w->unsorted_q, w->perm + lo});
    }
    ZKW_TRY(dev_chains(ctx, chains));
    // K5: challenges from the two final tails (W/ram_permutation.rs:80-90)
    std::vector<FsJob> fs(n_blocks);
    for (size_t b = 0; b < n_blocks; b++) {
        const size_t lo = w->offsets[b], n = w->offsets[b + 1] - lo;
        const size_t last_inst = w->inst_offsets[b + 1] - 1;  // the block's last instance ends on its last item
        fs[b] = FsJob{w->unsorted_marks + 12 * last_inst, w->sorted_marks + 12 * last_inst, (u32)n, (u32)n,
                      w->challenges + 18 * b};
    }
    ZKW_TRY(dev_fs(ctx, fs, 12, 9));
    // K6 + a10 in groups of whole blocks that fit the chain window: both repetitions and both sides of a group in one
    // launch set (W/ram_permutation.rs:115-138), then the per-instance records that read the chains at instance ends
    w->z_valid = false;
    for (size_t b0 = 0; b0 < n_blocks;) {
        size_t b1 = b0 + 1;
        while (b1 < n_blocks && w->offsets[b1 + 1] - w->offsets[b0] <= w->zcap) b1++;
        ZKW_TRY(ram_gp_blocks(ctx, w, b0, b1, w->zbuf_l, w->zbuf_r));
        const size_t base = w->offsets[b0];
        std::vector<RamBlock> blocks(b1 - b0);
        size_t max_inst = 0;
        for (size_t b = b0; b < b1; b++) {
            const size_t lo = w->offsets[b], n = w->offsets[b + 1] - lo;
            const size_t n_inst = w->inst_offsets[b + 1] - w->inst_offsets[b];
            if (n_inst > max_inst) max_inst = n_inst;
            blocks[b - b0] = RamBlock{w->unsorted_q,
                                      w->perm + lo,
                                      w->unsorted_marks + 12 * w->inst_offsets[b],
                                      w->sorted_marks + 12 * w->inst_offsets[b],
                                      w->zbuf_l + 2 * (lo - base),
                                      w->zbuf_r + 2 * (lo - base),
                                      w->instances + w->inst_offsets[b],
                                      w->nondet_counts + w->inst_offsets[b],
                                      n,
                                      w->capacity,
                                      n_nondet ? n_nondet[b] : 0u};
        }
        RamBlock* d_blocks = nullptr;
        ZKW_TRY(ctx->upload("ram_blocks", blocks, &d_blocks));
        const unsigned gx = (unsigned)(max_inst < 64 ? max_inst : 64), gy = (unsigned)(b1 - b0);
        { Prof _p(ctx, "k_ram_count_nondet"); ZKW_LAUNCH_2D(ctx, k_ram_count_nondet, gx, gy, 256, d_blocks); }
        ZKW_TRY(launch_check("k_ram_count_nondet"));
        { Prof _p(ctx, "k_ram_instances"); ZKW_LAUNCH_2D(ctx, k_ram_instances, blocks_for(max_inst, 64), gy, 64, d_blocks); }
        ZKW_TRY(launch_check("k_ram_instances"));
        b0 = b1;
    }
    // a20: compact forms and public inputs of every instance (postprocessing/mod.rs:353-369)
    const size_t ni = w->n_instances;
    { Prof _p(ctx, "k_ram_commitments"); ZKW_LAUNCH(ctx, k_ram_commitments, blocks_for(4 * ni, 64), 64, w->instances, ni, w->compact_forms); }
    ZKW_TRY(launch_check("k_ram_commitments"));
    { Prof _p(ctx, "k_commit_encodings"); ZKW_LAUNCH(ctx, k_commit_encodings, blocks_for(ni, 64), 64, w->compact_forms, ni, (u32)COMPACT_FORM_LEN, w->public_inputs); }
    return launch_check("k_commit_encodings");
}

extern "C" int zkw_ram_build_instances_batch(zkw_ctx* ctx, const zkw_mem_query* q, const uint64_t* block_offsets,
                                             size_t n_blocks, uint32_t capacity, const uint32_t* n_nondet,
                                             zkw_ram_witness** out) {
    if (!ctx || !q || !block_offsets || !out || n_blocks == 0 || capacity == 0)
        return fail(ZKW_ERR_INVALID, "zkw_ram_build_instances_batch: bad argument");
    HIP_TRY(hipSetDevice(ctx->device));
    const uint64_t base = block_offsets[0];
    std::vector<uint64_t> offs(n_blocks + 1), ioffs(n_blocks + 1, 0);
    for (size_t b = 0; b <= n_blocks; b++) offs[b] = block_offsets[b] - base;
    for (size_t b = 0; b < n_blocks; b++) {
        if (block_offsets[b + 1] <= block_offsets[b])
            return fail(ZKW_ERR_INVALID, "block %zu is empty: the VM must have made memory requests "
                                         "(W/ram_permutation.rs:43-46)", b);
        ioffs[b + 1] = ioffs[b] + (offs[b + 1] - offs[b] + capacity - 1) / capacity;
    }
    const size_t total = offs[n_blocks];
    if (total >= (1ull << 32)) return fail(ZKW_ERR_INVALID, "more than 2^32-1 queries in one batch");
    zkw_ram_witness* w = *out;
    if (w && (w->ctx != ctx || w->offsets != offs || w->capacity != capacity)) {
        zkw_ram_witness_free(w);
        w = nullptr;
        *out = nullptr;
    }
    if (!w) {
        w = new zkw_ram_witness();
        w->ctx = ctx;
        w->offsets = offs;
        w->inst_offsets = ioffs;
        w->capacity = capacity;
        w->total = total;
        w->n_instances = ioffs[n_blocks];
        int rc = ram_alloc(w, n_blocks);
        if (rc != ZKW_OK) {
            w->release();
            delete w;
            return rc;
        }
    }
    const zkw_mem_query* d_q = nullptr;
    int rc = ctx->in("ram_q", q + base, total, &d_q);
    if (rc == ZKW_OK) rc = ram_run(ctx, w, d_q, n_nondet);
    if (rc == ZKW_OK) rc = ctx->sync_if_host();
    if (rc != ZKW_OK) {
        if (!*out) {
            w->release();
            delete w;
        }
        return rc;
    }
    if (!*out) ctx_retain(ctx);  // a reused witness already holds its reference
    *out = w;
    return ZKW_OK;
}

extern "C" int zkw_ram_build_instances(zkw_ctx* ctx, const zkw_mem_query* q, size_t n, uint32_t capacity,
                                       uint32_t num_non_deterministic_heap_queries, zkw_ram_witness** out) {
    uint64_t offsets[2] = {0, n};
    return zkw_ram_build_instances_batch(ctx, q, offsets, 1, capacity, &num_non_deterministic_heap_queries, out);
}

extern "C" size_t zkw_ram_witness_num_instances(const zkw_ram_witness* w) { return w ? w->n_instances : 0; }
extern "C" size_t zkw_ram_witness_num_items(const zkw_ram_witness* w) { return w ? w->total : 0; }

// The sorted queries of the ABI: gathered on first access.
static int ram_sorted_queries(const zkw_ram_witness* cw) {
    zkw_ram_witness* w = const_cast<zkw_ram_witness*>(cw);
    if (w->sorted_valid) return ZKW_OK;
    zkw_ctx* ctx = w->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t t = w->total;
    if (!w->sorted_q && dev_malloc((void**)&w->sorted_q, (t + 1) * sizeof(zkw_mem_query)) != hipSuccess)
        return fail(ZKW_ERR_OOM, "no room for the sorted queries (%zu bytes): read ZKW_RAM_SORTED_QUERIES from a smaller batch", t * sizeof(zkw_mem_query));
    { Prof _p(ctx, "k_gather_encode"); ZKW_LAUNCH(ctx, k_gather_encode, blocks_for(t, 256), 256, w->unsorted_q, w->perm, t, w->sorted_q, (u64*)nullptr); }
    ZKW_TRY(launch_check("k_gather_encode"));
    w->sorted_valid = true;
    return ZKW_OK;
}

// The grand-product chains of the ABI ([2][n_b] per block at element offset 2 * block_offsets[b]): on first access.
static int ram_full_chains(const zkw_ram_witness* cw) {
    zkw_ram_witness* w = const_cast<zkw_ram_witness*>(cw);
    if (w->z_valid) return ZKW_OK;
    zkw_ctx* ctx = w->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t t = w->total;
    if (!w->lhs_z) {
        if (dev_malloc((void**)&w->lhs_z, (t + 1) * 16) != hipSuccess || dev_malloc((void**)&w->rhs_z, (t + 1) * 16) != hipSuccess)
            return fail(ZKW_ERR_OOM, "no room for the grand-product chains (%zu bytes): read ZKW_RAM_*_Z from a smaller batch", 2 * t * 16);
    }
    ZKW_TRY(ram_gp_blocks(ctx, w, 0, w->offsets.size() - 1, w->lhs_z, w->rhs_z));
    w->z_valid = true;
    return ZKW_OK;
}

// The [total][8] encoding arrays of the ABI: materialised on first access from the queries.
static int ram_encodings(const zkw_ram_witness* cw) {
    zkw_ram_witness* w = const_cast<zkw_ram_witness*>(cw);
    if (w->enc_valid) return ZKW_OK;
    zkw_ctx* ctx = w->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t t = w->total;
    if (!w->unsorted_enc) {
        if (dev_malloc((void**)&w->unsorted_enc, (t + 1) * 64) != hipSuccess || dev_malloc((void**)&w->sorted_enc, (t + 1) * 64) != hipSuccess)
            return fail(ZKW_ERR_OOM, "no room for the materialised encodings (%zu bytes): read ZKW_RAM_*_ENC from a smaller batch", 2 * t * 64);
    }
    ZKW_TRY(ram_sorted_queries(cw));
    ZKW_TRY(dev_encode(ctx, w->unsorted_q, t, w->unsorted_enc));
    ZKW_TRY(dev_encode(ctx, w->sorted_q, t, w->sorted_enc));
    w->enc_valid = true;
    return ZKW_OK;
}

// The [total][12] tail arrays of the ABI. The builder keeps tails compact (capacity words + instance ends); the full
// arrays are expanded on first access: one independent permutation per item from (enc[i], caps[i-1]).
static int ram_full_tails(const zkw_ram_witness* cw) {
    zkw_ram_witness* w = const_cast<zkw_ram_witness*>(cw);
    if (w->tails_valid) return ZKW_OK;
    ZKW_TRY(ram_encodings(cw));
    zkw_ctx* ctx = w->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t t = w->total;
    if (!w->unsorted_tails) {
        if (dev_malloc((void**)&w->unsorted_tails, (t + 1) * 96) != hipSuccess || dev_malloc((void**)&w->sorted_tails, (t + 1) * 96) != hipSuccess)
            return fail(ZKW_ERR_OOM, "no room for the expanded queue tails (%zu bytes): read ZKW_RAM_*_TAILS from a smaller batch", 2 * t * 96);
    }
    u64* d_off = nullptr;
    ZKW_TRY(ctx->upload("tails_off", w->offsets, &d_off));
    const int nq = (int)(w->offsets.size() - 1);
    { Prof _p(ctx, "k_tails_expand"); ZKW_LAUNCH(ctx, k_tails_expand, blocks_for(t, 64), 64, w->unsorted_enc, w->unsorted_caps, d_off, nq, t, w->unsorted_tails); }
    ZKW_TRY(launch_check("k_tails_expand"));
    { Prof _p(ctx, "k_tails_expand"); ZKW_LAUNCH(ctx, k_tails_expand, blocks_for(t, 64), 64, w->sorted_enc, w->sorted_caps, d_off, nq, t, w->sorted_tails); }
    ZKW_TRY(launch_check("k_tails_expand"));
    w->tails_valid = true;
    return ZKW_OK;
}

// rc (optional): status of the lazy materialisation, so that an OOM / HIP failure is reported as such
static const void* ram_array(const zkw_ram_witness* w, int what, size_t* bytes, bool materialize = true, int* rc = nullptr) {
    const size_t t = w->total, nb = w->offsets.size() - 1;
    int st = ZKW_OK;
    const void* p = nullptr;
#define RAM_LAZY(fn, field) do { if (materialize) { st = fn(w); if (st == ZKW_OK) p = w->field; } } while (0)
    switch (what) {
        case ZKW_RAM_SORTED_QUERIES: *bytes = t * sizeof(zkw_mem_query); RAM_LAZY(ram_sorted_queries, sorted_q); break;
        case ZKW_RAM_UNSORTED_ENC: *bytes = t * 64; RAM_LAZY(ram_encodings, unsorted_enc); break;
        case ZKW_RAM_SORTED_ENC: *bytes = t * 64; RAM_LAZY(ram_encodings, sorted_enc); break;
        case ZKW_RAM_UNSORTED_TAILS: *bytes = t * 96; RAM_LAZY(ram_full_tails, unsorted_tails); break;
        case ZKW_RAM_SORTED_TAILS: *bytes = t * 96; RAM_LAZY(ram_full_tails, sorted_tails); break;
        case ZKW_RAM_CHALLENGES: *bytes = nb * 18 * 8; p = w->challenges; break;
        case ZKW_RAM_LHS_Z: *bytes = t * 16; RAM_LAZY(ram_full_chains, lhs_z); break;
        case ZKW_RAM_RHS_Z: *bytes = t * 16; RAM_LAZY(ram_full_chains, rhs_z); break;
        case ZKW_RAM_INSTANCES: *bytes = w->n_instances * sizeof(zkw_ram_instance); p = w->instances; break;
        case ZKW_RAM_COMPACT_FORMS: *bytes = w->n_instances * COMPACT_FORM_LEN * 8; p = w->compact_forms; break;
        case ZKW_RAM_PUBLIC_INPUTS: *bytes = w->n_instances * 32; p = w->public_inputs; break;
        default: *bytes = 0; st = ZKW_ERR_INVALID; break;
    }
#undef RAM_LAZY
    if (rc) *rc = st;
    return p;
}

extern "C" size_t zkw_ram_witness_bytes(const zkw_ram_witness* w, int what) {
    size_t b = 0;
    if (w) (void)ram_array(w, what, &b, false);
    return b;
}

extern "C" const void* zkw_ram_witness_device_ptr(const zkw_ram_witness* w, int what) {
    size_t b = 0;
    return w ? ram_array(w, what, &b) : nullptr;
}

extern "C" int zkw_ram_witness_get(const zkw_ram_witness* w, int what, void* dst, size_t dst_bytes) {
    if (!w || !dst) return fail(ZKW_ERR_INVALID, "zkw_ram_witness_get: null argument");
    size_t bytes = 0;
    int st = ZKW_OK;
    const void* src = ram_array(w, what, &bytes, true, &st);
    if (st == ZKW_ERR_INVALID) return fail(ZKW_ERR_INVALID, "zkw_ram_witness_get: unknown array %d", what);
    if (st != ZKW_OK) return st;  // the lazy materialisation failed: its own code and message (OOM, HIP)
    if (!src) return fail(ZKW_ERR_INVALID, "zkw_ram_witness_get: array %d is empty", what);
    if (dst_bytes < bytes) return fail(ZKW_ERR_INVALID, "zkw_ram_witness_get: need %zu bytes, got %zu", bytes, dst_bytes);
    zkw_ctx* ctx = w->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(ctx->copy_async(dst, src, bytes, ctx->ptr_mode == ZKW_PTR_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost));
    return ctx->sync_if_host();
}

extern "C" void zkw_ram_witness_free(zkw_ram_witness* w) {
    if (!w) return;
    (void)hipSetDevice(w->ctx->device);
    (void)hipStreamSynchronize(w->ctx->stream);
    w->release();
    zkw_ctx* owner = w->ctx;
    delete w;
    ctx_release(owner);
}

// ------------------------------------------------------------------------------------------------ traces / synthesis

extern "C" int zkw_trace_create_with_columns(zkw_ctx* ctx, size_t n_rows, size_t n_cols, size_t n_slots, zkw_trace** out) {
    if (!ctx || !out || n_rows < 256 || n_slots == 0 || n_cols == 0 || n_cols > 4096)
        return fail(ZKW_ERR_INVALID, "zkw_trace_create: bad argument");
    HIP_TRY(hipSetDevice(ctx->device));
    zkw_trace* t = new zkw_trace();
    t->ctx = ctx;
    t->n_rows = n_rows;
    t->n_cols = n_cols;
    t->n_slots = n_slots;
    t->slot_tag.reset(new std::atomic<uint64_t>[n_slots]);
    for (size_t k = 0; k < n_slots; k++) t->slot_tag[k].store(0, std::memory_order_relaxed);
    hipError_t e = dev_malloc((void**)&t->data, t->slot_elems() * n_slots * sizeof(u64));
    if (e != hipSuccess) {
        delete t;
        return fail(ZKW_ERR_OOM, "zkw_trace_create: hipMalloc of %zu bytes failed: %s",
                    n_cols * n_rows * n_slots * 8, hipGetErrorString(e));
    }
    ctx_retain(ctx);
    *out = t;
    return ZKW_OK;
}

extern "C" int zkw_trace_create(zkw_ctx* ctx, size_t n_rows, size_t n_slots, zkw_trace** out) {
    return zkw_trace_create_with_columns(ctx, n_rows, RC_COLS, n_slots, out);
}

extern "C" void zkw_trace_free(zkw_trace* t) {
    if (!t) return;
    (void)hipSetDevice(t->ctx->device);
    (void)hipStreamSynchronize(t->ctx->stream);
    if (t->data) dev_free(t->data);
    zkw_ctx* owner = t->ctx;
    delete t;
    ctx_release(owner);
}

extern "C" size_t zkw_trace_num_rows(const zkw_trace* t) { return t ? t->n_rows : 0; }
extern "C" size_t zkw_trace_num_cols(const zkw_trace* t) { return t ? t->n_cols : 0; }
extern "C" size_t zkw_trace_num_slots(const zkw_trace* t) { return t ? t->n_slots : 0; }
extern "C" const uint64_t* zkw_trace_device_ptr(const zkw_trace* t, size_t slot) {
    if (!t || slot >= t->n_slots) return nullptr;
    return t->slot_for_write(slot, 0);  // the caller may write through it: the slot's contents are unknown from here on
}

extern "C" int zkw_trace_get(const zkw_trace* t, size_t slot, uint32_t first_col, uint32_t n_cols, uint64_t* dst) {
    if (!t || !dst || slot >= t->n_slots || (size_t)first_col + n_cols > t->n_cols) return fail(ZKW_ERR_INVALID, "zkw_trace_get: bad argument");
    zkw_ctx* ctx = t->ctx;
    HIP_TRY(hipSetDevice(ctx->device));
    const u64* src = t->data + slot * t->slot_elems() + (size_t)first_col * t->n_rows;
    HIP_TRY(ctx->copy_async(dst, src, (size_t)n_cols * t->n_rows * sizeof(u64), ctx->ptr_mode == ZKW_PTR_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost));
    return ctx->sync_if_host();
}

extern "C" int zkw_ram_synthesize(zkw_ctx* ctx, const zkw_ram_witness* w, size_t first_instance, size_t n_instances,
                                  zkw_trace* t, size_t first_slot) {
    if (!ctx || !w || !t || w->ctx != ctx || t->ctx->device != ctx->device) return fail(ZKW_ERR_INVALID, "zkw_ram_synthesize: bad argument");
    if (first_instance + n_instances > w->n_instances) return fail(ZKW_ERR_INVALID, "instance range out of bounds");
    if (n_instances > t->n_slots) return fail(ZKW_ERR_INVALID, "more instances (%zu) than trace slots (%zu)", n_instances, t->n_slots);
    const u32 capacity = w->capacity;
    const size_t n_rows = t->n_rows;
    if (RC_MIN_ROWS(capacity) > n_rows)
        return fail(ZKW_ERR_INVALID, "capacity %u needs %llu rows, trace has %zu", capacity,
                    (unsigned long long)RC_MIN_ROWS(capacity), n_rows);
    if (n_rows & 1) return fail(ZKW_ERR_INVALID, "trace length must be even (it is a power of two in every circuit)");
    if (n_instances == 0) return ZKW_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    // the grand-product chains of the blocks these instances belong to, recomputed into the witness's window
    const size_t n_blocks = w->offsets.size() - 1;
    size_t b_first = 0;
    while (b_first + 1 < n_blocks && w->inst_offsets[b_first + 1] <= first_instance) b_first++;
    size_t b_end = b_first + 1;  // one past the last block touched
    while (b_end < n_blocks && w->inst_offsets[b_end] < first_instance + n_instances) b_end++;
    const size_t win = std::min(w->zcap, w->sqcap);
    if (w->offsets[b_end] - w->offsets[b_first] > win) {  // too many blocks for one window: split at a block boundary
        size_t b_mid = b_first + 1;
        while (b_mid + 1 < b_end && w->offsets[b_mid + 1] - w->offsets[b_first] <= win) b_mid++;
        const size_t n_head = w->inst_offsets[b_mid] - first_instance;
        ZKW_TRY(zkw_ram_synthesize(ctx, w, first_instance, n_head, t, first_slot));
        return zkw_ram_synthesize(ctx, w, first_instance + n_head, n_instances - n_head, t, first_slot + n_head);
    }
    ZKW_TRY(ram_gp_blocks(ctx, w, b_first, b_end, w->zbuf_l, w->zbuf_r));
    const size_t z_base = w->offsets[b_first];
    {   // the fills read the sorted queue contiguously: gather the touched blocks once per call
        const size_t cnt = w->offsets[b_end] - z_base;
        Prof _p(ctx, "k_gather_encode");
        ZKW_LAUNCH(ctx, k_gather_encode, blocks_for(cnt, 256), 256, w->unsorted_q, w->perm + z_base, cnt, w->sq_win, (u64*)nullptr);
    }
    ZKW_TRY(launch_check("k_gather_encode"));
    const u32 rstride = (u32)RC_REGION_STRIDE(capacity);  // rows per region incl. the alignment gap
    const u32 n_tiles = (rstride + 255) / 256;
    u32 *d_hist = nullptr, *d_nd = nullptr;
    ZKW_TRY(ctx->scratch_t<u32>("synth_hist", n_instances * 256, &d_hist));
    ZKW_TRY(ctx->scratch_t<u32>("synth_nd", n_instances * n_tiles, &d_nd));
    HIP_TRY(ctx->memset_async(d_hist, 0, n_instances * 256 * sizeof(u32)));
    SlotClaims claims(t);
    std::vector<SynthJob> jobs(n_instances);
    size_t b = b_first;
    for (size_t k = 0; k < n_instances; k++) {
        const size_t idx = first_instance + k;
        while (b + 1 < n_blocks && w->inst_offsets[b + 1] <= idx) b++;
        const size_t lo = w->offsets[b], nb = w->offsets[b + 1] - lo;
        SynthJob& j = jobs[k];
        j.inst = w->instances + idx;
        j.sorted_q = w->sq_win + (lo - z_base);
        j.unsorted_q = w->unsorted_q + lo;
        j.unsorted_caps = w->unsorted_caps + 4 * lo;
        j.sorted_caps = w->sorted_caps + 4 * lo;
        j.u_mark = w->unsorted_marks + 12 * idx;
        j.s_mark = w->sorted_marks + 12 * idx;
        j.challenges = w->challenges + 18 * b;
        j.lhs_z = w->zbuf_l + 2 * (lo - z_base);
        j.rhs_z = w->zbuf_r + 2 * (lo - z_base);
        j.n_block = nb;
        {   // a slot whose previous tenant was this layout keeps its zero padding rows (zkw_ctx.h slot_tag; every other writer resets the tag)
            const uint64_t tag = ((uint64_t)ZKW_CIRCUIT_RAM_PERMUTATION << 56) ^ ((uint64_t)capacity << 24) ^ (uint64_t)n_rows ^ 0x5A00000000000000ull;
            const size_t slot = (first_slot + k) % t->n_slots;
            bool clean = false;
            j.trace = claims.claim(slot, tag, &clean);
            j.tail_clean = clean;
        }
        j.hist = d_hist + 256 * k;
        j.nd_tiles = d_nd + (size_t)n_tiles * k;
        j.public_input = w->public_inputs + 4 * idx;
        j.first_inst = w->instances + w->inst_offsets[b];
    }
    SynthJob* d_jobs = nullptr;
    ZKW_TRY(ctx->upload("synth_jobs", jobs, &d_jobs));
    const unsigned nj = (unsigned)n_instances;
    const dim3 g64((rstride + 63) / 64, nj), g256(n_tiles, nj);
    { Prof _p(ctx, "k_ram_nd_tiles"); ZKW_LAUNCH_D(ctx, (k_ram_nd_tiles), "k_ram_nd_tiles", g256, 256, 0, d_jobs, capacity); }
    ZKW_TRY(launch_check("k_ram_nd_tiles"));
    { Prof _p(ctx, "k_ram_nd_scan"); ZKW_LAUNCH(ctx, k_ram_nd_scan, nj, 64, d_jobs, (int)nj, n_tiles); }
    ZKW_TRY(launch_check("k_ram_nd_scan"));
    { Prof _p(ctx, "k_ram_fill_poseidon"); ZKW_LAUNCH_D(ctx, (k_ram_fill_poseidon<0>), "k_ram_fill_poseidon", g64, 64, 0, d_jobs, capacity, n_rows); }
    ZKW_TRY(launch_check("k_ram_fill_poseidon<0>"));
    { Prof _p(ctx, "k_ram_fill_poseidon"); ZKW_LAUNCH_D(ctx, (k_ram_fill_poseidon<1>), "k_ram_fill_poseidon", g64, 64, 0, d_jobs, capacity, n_rows); }
    ZKW_TRY(launch_check("k_ram_fill_poseidon<1>"));
    { Prof _p(ctx, "k_ram_fill_A"); ZKW_LAUNCH_D(ctx, (k_ram_fill_A), "k_ram_fill_A", g256, 256, 0, d_jobs, capacity, n_rows); }
    ZKW_TRY(launch_check("k_ram_fill_A"));
    { Prof _p(ctx, "k_ram_fill_B"); ZKW_LAUNCH_D(ctx, (k_ram_fill_B), "k_ram_fill_B", g256, 256, 0, d_jobs, capacity, n_rows); }
    ZKW_TRY(launch_check("k_ram_fill_B"));
    { Prof _p(ctx, "k_ram_fill_C"); ZKW_LAUNCH_D(ctx, (k_ram_fill_C), "k_ram_fill_C", g256, 256, 0, d_jobs, capacity, n_rows); }
    ZKW_TRY(launch_check("k_ram_fill_C"));
    { Prof _p(ctx, "k_ram_fill_D"); const unsigned d_tiles = (unsigned)((rstride + RC_D_TILES * 256 - 1) / (RC_D_TILES * 256));  // row D: RC_D_TILES tiles per block (they share one inversion per lane)
      ZKW_LAUNCH(ctx, k_ram_fill_D, nj * (d_tiles + 1), 256, d_jobs, nj, d_tiles, capacity, n_rows); }
    ZKW_TRY(launch_check("k_ram_fill_D"));
    { Prof _p(ctx, "k_ram_fill_tail"); ZKW_LAUNCH(ctx, k_ram_fill_tail, nj * ((RC_G + RC_L + 1) * TAIL_CHUNKS), 256, d_jobs, capacity, n_rows); }
    ZKW_TRY(launch_check("k_ram_fill_tail"));
    return claims.commit_if(ctx->sync_if_host());
}


extern "C" int zkw_ram_check_satisfied(zkw_ctx* ctx, const zkw_trace* t, size_t slot, uint32_t capacity,
                                       uint64_t* n_violations, uint64_t* first_bad) {
    if (!ctx || !t || t->ctx->device != ctx->device || slot >= t->n_slots || !n_violations)
        return fail(ZKW_ERR_INVALID, "zkw_ram_check_satisfied: bad argument");
    if (RC_MIN_ROWS(capacity) > t->n_rows) return fail(ZKW_ERR_INVALID, "capacity does not fit the trace");
    return check_satisfied<SpecRam>(ctx, t, slot, capacity, n_violations, first_bad);
}

extern "C" int zkw_decommit_sorter_check_satisfied(zkw_ctx* ctx, const zkw_trace* t, size_t slot, uint32_t capacity,
                                                   uint64_t* n_violations, uint64_t* first_bad) {
    if (!ctx || !t || t->ctx->device != ctx->device || slot >= t->n_slots || !n_violations)
        return fail(ZKW_ERR_INVALID, "zkw_decommit_sorter_check_satisfied: bad argument");
    if (DS_MIN_ROWS(capacity) > t->n_rows) return fail(ZKW_ERR_INVALID, "capacity does not fit the trace");
    return check_satisfied<SpecDecommitSorter>(ctx, t, slot, capacity, n_violations, first_bad);
}

// ------------------------------------------------------------------------------------------------ closed forms from records (a20)
extern "C" int zkw_closed_form_public_inputs(zkw_ctx* ctx, uint8_t circuit_type, const void* instances, size_t n, uint64_t* compact,
                                             uint64_t* public_inputs) {
    if (!ctx || !compact || !public_inputs || (n && !instances)) return fail(ZKW_ERR_INVALID, "zkw_closed_form_public_inputs: null argument");
    if (n == 0) return ZKW_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    switch (circuit_type) {
        case 3: return closed_form_from_records<CfDecommitter>(ctx, instances, n, compact, public_inputs);
        case 5: return closed_form_from_records<CfPrecompile<ZKW_PRECOMPILE_KECCAK256>>(ctx, instances, n, compact, public_inputs);
        case 6: return closed_form_from_records<CfPrecompile<ZKW_PRECOMPILE_SHA256>>(ctx, instances, n, compact, public_inputs);
        case 7: return closed_form_from_records<CfPrecompile<ZKW_PRECOMPILE_ECRECOVER>>(ctx, instances, n, compact, public_inputs);
        case 10: return closed_form_from_records<CfStorageApplication>(ctx, instances, n, compact, public_inputs);
        case 13: return closed_form_from_records<CfLinearHasher>(ctx, instances, n, compact, public_inputs);
        default: break;
    }
    return fail(ZKW_ERR_INVALID, "zkw_closed_form_public_inputs: circuit type %u keeps its compact forms in its witness (2, 4, 8, 9, 11, 12) "
                                 "or is not a base-layer circuit with a closed form here", (unsigned)circuit_type);
}

// ------------------------------------------------------------------------------------------------ public inputs (a20)
extern "C" int zkw_commit_encodings(zkw_ctx* ctx, const uint64_t* enc, size_t n_items, uint32_t item_len, uint64_t* out) {
    if (!ctx || !out || (n_items && item_len && !enc)) return fail(ZKW_ERR_INVALID, "zkw_commit_encodings: null argument");
    if (n_items == 0) return ZKW_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    const u64* d_enc = nullptr;
    u64* d_out = nullptr;
    ZKW_TRY(ctx->in("ce_enc", enc, n_items * item_len + 1, &d_enc));
    ZKW_TRY(ctx->out("ce_out", out, n_items * 4, &d_out));
    { Prof _p(ctx, "k_commit_encodings"); ZKW_LAUNCH(ctx, k_commit_encodings, blocks_for(n_items, 64), 64, d_enc, n_items, item_len, d_out); }
    ZKW_TRY(launch_check("k_commit_encodings"));
    ZKW_TRY(ctx->finish_out(out, d_out, n_items * 4));
    return ctx->sync_if_host();
}

extern "C" int zkw_encode_recursion_requests(zkw_ctx* ctx, uint64_t circuit_type, const uint64_t* public_inputs, size_t n,
                                             uint64_t* enc) {
    if (!ctx || !enc || !public_inputs) return fail(ZKW_ERR_INVALID, "zkw_encode_recursion_requests: null argument");
    if (n == 0) return ZKW_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    const u64* d_pi = nullptr;
    u64* d_enc = nullptr;
    ZKW_TRY(ctx->in("rr_pi", public_inputs, n * 4, &d_pi));
    ZKW_TRY(ctx->out("rr_enc", enc, n * 8, &d_enc));
    { Prof _p(ctx, "k_encode_recursion"); ZKW_LAUNCH(ctx, k_encode_recursion, blocks_for(n, 64), 64, circuit_type, d_pi, n, d_enc); }
    ZKW_TRY(launch_check("k_encode_recursion"));
    ZKW_TRY(ctx->finish_out(enc, d_enc, n * 8));
    return ctx->sync_if_host();
}

// ------------------------------------------------------------------------------------------------ callstack (a3, a6)
extern "C" int zkw_encode_callstack_entries(zkw_ctx* ctx, const zkw_callstack_entry* entries, size_t n, uint64_t* enc) {
    if (!ctx || !enc || !entries) return fail(ZKW_ERR_INVALID, "zkw_encode_callstack_entries: null argument");
    if (n == 0) return ZKW_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    const zkw_callstack_entry* d_e = nullptr;
    u64* d_enc = nullptr;
    ZKW_TRY(ctx->in("cs_e", entries, n, &d_e));
    ZKW_TRY(ctx->out("cs_enc", enc, n * 32, &d_enc));
    { Prof _p(ctx, "k_encode_callstack"); hipLaunchKernelGGL(k_encode_callstack, dim3(blocks_for(n, 64)), dim3(64), 0, ctx->stream, d_e, n, d_enc); }
    ZKW_TRY(launch_check("k_encode_callstack"));
    ZKW_TRY(ctx->finish_out(enc, d_enc, n * 32));
    return ctx->sync_if_host();
}

extern "C" int zkw_callstack_simulate(zkw_ctx* ctx, const uint8_t* is_push, size_t n_ops, const zkw_callstack_entry* pushed,
                                      size_t n_pushed, uint64_t* previous_state, uint64_t* new_state, uint32_t* depth,
                                      uint64_t* round_states, uint32_t* entry_index) {
    if (!ctx || !is_push || !previous_state || !new_state || !depth || !round_states || !entry_index || (n_pushed && !pushed))
        return fail(ZKW_ERR_INVALID, "zkw_callstack_simulate: null argument");
    if (n_ops == 0) return ZKW_OK;
    if (n_ops >= (1ull << 31)) return fail(ZKW_ERR_INVALID, "too many stack operations");
    HIP_TRY(hipSetDevice(ctx->device));
    const uint8_t* d_ops = nullptr;
    const zkw_callstack_entry* d_e = nullptr;
    ZKW_TRY(ctx->in("st_ops", is_push, n_ops, &d_ops));
    if (n_pushed) ZKW_TRY(ctx->in("st_e", pushed, n_pushed, &d_e));
    u32 *d_depth = nullptr, *d_rank = nullptr, *d_meta = nullptr, *d_pd = nullptr, *d_pid = nullptr, *d_sd = nullptr, *d_sid = nullptr,
        *d_parent = nullptr, *d_node = nullptr;
    u64* d_rounds = nullptr;
    void* tmp = nullptr;
    const size_t tmp_bytes = radix_temp_bytes(n_ops);
    ZKW_TRY(ctx->scratch_t<u32>("st_depth", n_ops, &d_depth));
    ZKW_TRY(ctx->scratch_t<u32>("st_rank", n_ops, &d_rank));
    ZKW_TRY(ctx->scratch_t<u32>("st_meta", 4, &d_meta));
    ZKW_TRY(ctx->scratch_t<u32>("st_pd", n_ops, &d_pd));
    ZKW_TRY(ctx->scratch_t<u32>("st_pid", n_ops, &d_pid));
    ZKW_TRY(ctx->scratch_t<u32>("st_sd", n_ops, &d_sd));
    ZKW_TRY(ctx->scratch_t<u32>("st_sid", n_ops, &d_sid));
    ZKW_TRY(ctx->scratch_t<u32>("st_parent", n_ops, &d_parent));
    ZKW_TRY(ctx->scratch_t<u32>("st_node", n_ops, &d_node));
    ZKW_TRY(ctx->scratch_t<u64>("st_rounds", n_ops * 48, &d_rounds));
    ZKW_TRY(ctx->scratch("st_tmp", tmp_bytes + 256, &tmp));
    u64 *d_prefix = nullptr, *d_totals = nullptr;
    ZKW_TRY(ctx->scratch_t<u64>("st_prefix", 2 * (n_ops + 1), &d_prefix));
    ZKW_TRY(ctx->scratch_t<u64>("st_totals", 2, &d_totals));
    HIP_TRY(ctx->memset_async(d_meta, 0, 4 * sizeof(u32)));
    ZKW_TRY((sum_prefix<2>(ctx, "k_stack_prefix", StackDelta{d_ops}, n_ops, d_prefix, d_totals)));
    { Prof _p(ctx, "k_stack_depth"); hipLaunchKernelGGL(k_stack_depth, dim3(blocks_for(n_ops, 256)), dim3(256), 0, ctx->stream, d_ops, n_ops, d_prefix, d_depth, d_rank, d_meta); }
    ZKW_TRY(launch_check("k_stack_depth"));
    u32 meta[3];
    ZKW_TRY(ctx->read_small(meta, d_meta, sizeof meta));
    if (meta[2]) return fail(ZKW_ERR_INVALID, "pop from the empty callstack (circuit_encodings/src/lib.rs:619)");
    const u32 n_push = meta[0], max_depth = meta[1];
    if (n_push > n_pushed) return fail(ZKW_ERR_INVALID, "%u pushes but only %zu entries", n_push, n_pushed);
    const unsigned grid = blocks_for(n_ops, 256);
    if (n_push) {
        { Prof _p(ctx, "k_stack_push_keys"); hipLaunchKernelGGL(k_stack_push_keys, dim3(grid), dim3(256), 0, ctx->stream, d_ops, n_ops, d_depth, d_rank, d_pd, d_pid); }
        ZKW_TRY(launch_check("k_stack_push_keys"));
        unsigned bits = 1;
        while ((1ull << bits) <= max_depth) bits++;
        { Prof _p(ctx, "radix_sort"); ZKW_TRY(radix_sort_pairs<u32>(ctx, tmp, tmp_bytes, d_pd, d_sd, d_pid, d_sid, n_push, bits)); }
    }
    { Prof _p(ctx, "k_stack_links"); hipLaunchKernelGGL(k_stack_links, dim3(grid), dim3(256), 0, ctx->stream, d_ops, n_ops, d_depth, d_rank, d_sd, d_sid, d_meta, d_parent, d_node); }
    ZKW_TRY(launch_check("k_stack_links"));
    for (u32 d = 1; d <= max_depth; d++) {
        Prof _p(ctx, "k_stack_level");
        hipLaunchKernelGGL(k_stack_level, dim3(blocks_for(n_push, 64)), dim3(64), 0, ctx->stream, d_e, d_pd, d_parent, d_meta, d, d_rounds);
    }
    ZKW_TRY(launch_check("k_stack_level"));
    u64 *d_prev = nullptr, *d_new = nullptr, *d_rs = nullptr;
    u32 *d_dep = nullptr, *d_idx = nullptr;
    ZKW_TRY(ctx->out("st_o_prev", previous_state, n_ops * 12, &d_prev));
    ZKW_TRY(ctx->out("st_o_new", new_state, n_ops * 12, &d_new));
    ZKW_TRY(ctx->out("st_o_rs", round_states, n_ops * 48, &d_rs));
    ZKW_TRY(ctx->out("st_o_dep", depth, n_ops, &d_dep));
    ZKW_TRY(ctx->out("st_o_idx", entry_index, n_ops, &d_idx));
    { Prof _p(ctx, "k_stack_emit"); hipLaunchKernelGGL(k_stack_emit, dim3(grid), dim3(256), 0, ctx->stream, d_ops, n_ops, d_depth, d_parent, d_node, d_rounds, d_meta, d_prev, d_new, d_dep, d_rs, d_idx); }
    ZKW_TRY(launch_check("k_stack_emit"));
    ZKW_TRY(ctx->finish_out(previous_state, d_prev, n_ops * 12));
    ZKW_TRY(ctx->finish_out(new_state, d_new, n_ops * 12));
    ZKW_TRY(ctx->finish_out(round_states, d_rs, n_ops * 48));
    ZKW_TRY(ctx->finish_out(depth, d_dep, n_ops));
    ZKW_TRY(ctx->finish_out(entry_index, d_idx, n_ops));
    return ctx->sync_if_host();
}

// ------------------------------------------------------------------------------------------------ MainVM instance slicing (a19)
extern "C" int zkw_vm_slice_instances(zkw_ctx* ctx, const zkw_vm_tracer_streams* in, zkw_vm_instance* instances,
                                      uint32_t* memory_read_index, uint32_t* memory_write_index, uint64_t* n_reads, uint64_t* n_writes) {
    if (!ctx || !in || !instances || !in->snapshot_cycles || in->n_snapshots < 2 || ((memory_read_index == nullptr) != (memory_write_index == nullptr)))
        return fail(ZKW_ERR_INVALID, "zkw_vm_slice_instances: bad argument");
    for (int k = 0; k < ZKW_VM_NUM_STREAMS; k++)
        if (in->stream_len[k] && !in->stream_cycles[k]) return fail(ZKW_ERR_INVALID, "zkw_vm_slice_instances: stream %d has no cycle stamps", k);
    const size_t n_mem = in->stream_len[ZKW_VMS_MEMORY];
    if (n_mem >= (1ull << 32)) return fail(ZKW_ERR_INVALID, "too many memory queries");
    if ((n_mem && (!in->vm_memory_queries || !in->memory_queue_tails)) || (in->n_decommit_states && (!in->decommit_state_cycles || !in->decommit_queue_tails)) ||
        (in->n_callstack_sponges && (!in->callstack_sponge_cycles || !in->callstack_sponge_states)) ||
        (in->n_storage_log_states && (!in->storage_log_state_cycles || !in->storage_log_states)))
        return fail(ZKW_ERR_INVALID, "zkw_vm_slice_instances: a stream's payload is missing");
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t n_inst = in->n_snapshots - 1;
    VmSliceJob job;
    job.s = *in;
    ZKW_TRY(ctx->in("vm_snap", in->snapshot_cycles, in->n_snapshots, &job.s.snapshot_cycles));
    static const char* names[ZKW_VM_NUM_STREAMS] = {"vm_c0", "vm_c1", "vm_c2", "vm_c3", "vm_c4", "vm_c5", "vm_c6", "vm_c7"};
    for (int k = 0; k < ZKW_VM_NUM_STREAMS; k++) ZKW_TRY(ctx->in(names[k], in->stream_cycles[k], in->stream_len[k], &job.s.stream_cycles[k]));
    ZKW_TRY(ctx->in("vm_memq", in->vm_memory_queries, n_mem, &job.s.vm_memory_queries));
    ZKW_TRY(ctx->in("vm_memt", in->memory_queue_tails, n_mem * 12, &job.s.memory_queue_tails));
    ZKW_TRY(ctx->in("vm_decc", in->decommit_state_cycles, in->n_decommit_states, &job.s.decommit_state_cycles));
    ZKW_TRY(ctx->in("vm_dect", in->decommit_queue_tails, in->n_decommit_states * 12, &job.s.decommit_queue_tails));
    ZKW_TRY(ctx->in("vm_csc", in->callstack_sponge_cycles, in->n_callstack_sponges, &job.s.callstack_sponge_cycles));
    ZKW_TRY(ctx->in("vm_css", in->callstack_sponge_states, in->n_callstack_sponges * 12, &job.s.callstack_sponge_states));
    ZKW_TRY(ctx->in("vm_slc", in->storage_log_state_cycles, in->n_storage_log_states, &job.s.storage_log_state_cycles));
    ZKW_TRY(ctx->in("vm_sls", in->storage_log_states, in->n_storage_log_states, &job.s.storage_log_states));
    // the read / write split of the memory stream: one stable partition for all instances
    u32 *d_tiles = nullptr, *d_prefix = nullptr, *d_ri = nullptr, *d_wi = nullptr;
    u64* d_total = nullptr;
    const u32 n_tiles = (u32)((n_mem + VM_TILE) / VM_TILE);  // covers index n_mem itself (the total)
    ZKW_TRY(ctx->scratch_t<u32>("vm_tiles", n_tiles, &d_tiles));
    ZKW_TRY(ctx->scratch_t<u32>("vm_prefix", n_mem + 1, &d_prefix));
    ZKW_TRY(ctx->scratch_t<u64>("vm_total", 1, &d_total));
    if (memory_read_index) {
        ZKW_TRY(ctx->out("vm_ri", memory_read_index, n_mem, &d_ri));
        ZKW_TRY(ctx->out("vm_wi", memory_write_index, n_mem, &d_wi));
    }
    { Prof _p(ctx, "k_vm_rw_tile_counts"); ZKW_LAUNCH(ctx, k_vm_rw_tile_counts, n_tiles, 256, job.s.vm_memory_queries, (u64)n_mem, d_tiles); }
    ZKW_TRY(launch_check("k_vm_rw_tile_counts"));
    { Prof _p(ctx, "k_vm_rw_scan_tiles"); ZKW_LAUNCH(ctx, k_vm_rw_scan_tiles, 1, 1024, d_tiles, n_tiles, d_total); }
    ZKW_TRY(launch_check("k_vm_rw_scan_tiles"));
    { Prof _p(ctx, "k_vm_rw_scatter"); ZKW_LAUNCH(ctx, k_vm_rw_scatter, n_tiles, 256, job.s.vm_memory_queries, (u64)n_mem, d_tiles, d_prefix, d_ri, d_wi); }
    ZKW_TRY(launch_check("k_vm_rw_scatter"));
    job.read_prefix = d_prefix;
    ZKW_TRY(ctx->out("vm_inst", instances, n_inst, &job.out));
    { Prof _p(ctx, "k_vm_slice"); ZKW_LAUNCH(ctx, k_vm_slice, blocks_for(n_inst, 64), 64, job); }
    ZKW_TRY(launch_check("k_vm_slice"));
    ZKW_TRY(ctx->finish_out(instances, job.out, n_inst));
    if (memory_read_index) {
        ZKW_TRY(ctx->finish_out(memory_read_index, d_ri, n_mem));
        ZKW_TRY(ctx->finish_out(memory_write_index, d_wi, n_mem));
    }
    if (n_reads || n_writes) {
        u64 total = 0;
        ZKW_TRY(ctx->read_small(&total, d_total, 8));
        if (n_reads) *n_reads = total;
        if (n_writes) *n_writes = n_mem - total;
    }
    return ctx->sync_if_host();
}


