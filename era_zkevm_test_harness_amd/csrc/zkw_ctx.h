// zkw_ctx.h — the context, its buffer helpers and the device-level steps shared by the library's translation units that
// launch kernels (zkw_api.hip: core + RAM path; zkw_sorters.hip: the four sorter / demuxer circuits; zkw_precompiles.hip: code
// decommitter, precompiles, storage application and the netlist circuits; zkw_setup.hip: layouts, selectors, sigma). The other
// units (zkw_block.hip, zkw_comm.hip, zkw_recursion.hip, zkw_vm_trace.hip) are written against include/zkw.h and zkw_internal.h.
// Kernels live in the *.cuh headers with internal linkage: a unit includes the ones it launches.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/zkw.h"
#include "zkw_internal.h"
#include "zkw_batch.h"
#include "ram_kernels.cuh"
#include "log_kernels.cuh"
#include "../../include/zkw_ram_circuit_spec.h"  // RC_COLS: the default width of a trace

using namespace zkw;

#define fail zkw_fail  // sets the calling thread's zkw_last_error() text and returns the code

// the library's allocation caches (zkw_api.hip): device / pinned host buffers by size class, never hipFree'd while cached
hipError_t zkw_cache_alloc(int pinned_host, void** p, size_t bytes);
void zkw_cache_release(int pinned_host, void* p);
static inline hipError_t dev_malloc(void** p, size_t bytes) { return zkw_cache_alloc(0, p, bytes); }
template <class T> static inline hipError_t dev_malloc(T** p, size_t bytes) { return zkw_cache_alloc(0, (void**)p, bytes); }
static inline void dev_free(void* p) { zkw_cache_release(0, p); }
static inline hipError_t pin_malloc(void** p, size_t bytes) { return zkw_cache_alloc(1, p, bytes); }
static inline void pin_free(void* p) { zkw_cache_release(1, p); }
// the library's stream pool (zkw_api.hip): streams are never destroyed while pooled; a released stream must be idle or ordered by events
hipError_t zkw_pool_stream_acquire(hipStream_t* s);
void zkw_pool_stream_release(hipStream_t s);

#define HIP_TRY(expr)                                                                               \
    do {                                                                                            \
        hipError_t _e = (expr);                                                                     \
        if (_e != hipSuccess)                                                                       \
            return fail(_e == hipErrorOutOfMemory ? ZKW_ERR_OOM : ZKW_ERR_HIP, "%s failed: %s (%s:%d)", \
                        #expr, hipGetErrorString(_e), __FILE__, __LINE__);                          \
    } while (0)

#define ZKW_TRY(expr)            \
    do {                         \
        int _rc = (expr);        \
        if (_rc != ZKW_OK) return _rc; \
    } while (0)

// ------------------------------------------------------------------------------------------------ context
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

// pinned host staging for descriptor uploads; `ev` marks the last copy that read it
struct HostStage {
    void* p = nullptr;
    size_t cap = 0;
    hipEvent_t ev = nullptr;
    bool pending = false;
};

struct zkw_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    int ptr_mode = ZKW_PTR_HOST;
    // a context of a block that runs inside a batch (zkw_blocks_run, zkw_batch.h): launches, memsets, copies and waits on `stream` are
    // queued with the batch / park the calling fiber instead of reaching HIP; `stream` is then the batch's stream (not owned)
    zkw_batch* batch = nullptr;
    bool batched() const { return batch && zkw_batch_in_fiber(batch); }
    hipError_t memset_async(void* p, int value, size_t bytes) {
        if (batched()) { zkw_batch_memset(batch, p, value, bytes); return hipSuccess; }
        return hipMemsetAsync(p, value, bytes, stream);
    }
    hipError_t copy_async(void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
        if (batched()) {
            if (kind == hipMemcpyDeviceToDevice) zkw_batch_copy_d2d(batch, dst, src, bytes);
            else if (kind == hipMemcpyHostToDevice) zkw_batch_copy_h2d(batch, dst, src, bytes);
            else if (kind == hipMemcpyDeviceToHost) zkw_batch_copy_d2h(batch, dst, src, bytes);
            else return hipErrorInvalidValue;
            return hipSuccess;
        }
        return hipMemcpyAsync(dst, src, bytes, kind, stream);
    }
    hipError_t sync_stream() {
        if (batched()) return zkw_batch_sync(batch) == ZKW_OK ? hipSuccess : hipErrorUnknown;
        if (side_stream && side_join() != ZKW_OK) return hipErrorUnknown;  // a fork left open by a failed call
        return hipStreamSynchronize(stream);
    }
    hipStream_t chain_stream = nullptr;  // optional second stream for the queue-chain kernels (zkw_set_chain_stream)
    hipEvent_t chain_ev_a = nullptr, chain_ev_b = nullptr;
    // a side stream for work that depends on nothing the main stream is about to write (the closed-form sponges of the netlist circuits):
    // fork = it waits for everything queued on `stream` so far, join = `stream` waits for it. Borrowed from the library's stream pool for the
    // time between the two (round 6; it used to be created per context and destroyed with it: two streams per block in flight, and
    // hipStreamDestroy waits for the whole device). A fork that was never joined — a failure in between — is joined by sync_stream().
    hipStream_t side_stream = nullptr;
    hipEvent_t side_ev_fork = nullptr, side_ev_join = nullptr;
    int side_fork(hipStream_t* out) {
        if (!side_ev_fork) {
            HIP_TRY(hipEventCreateWithFlags(&side_ev_fork, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&side_ev_join, hipEventDisableTiming));
        }
        if (!side_stream) HIP_TRY(zkw_pool_stream_acquire(&side_stream));
        HIP_TRY(hipEventRecord(side_ev_fork, stream));
        HIP_TRY(hipStreamWaitEvent(side_stream, side_ev_fork, 0));
        *out = side_stream;
        return ZKW_OK;
    }
    int side_join() {
        if (!side_stream) return ZKW_OK;
        hipStream_t s = side_stream;
        side_stream = nullptr;
        const hipError_t e1 = hipEventRecord(side_ev_join, s);
        const hipError_t e2 = e1 == hipSuccess ? hipStreamWaitEvent(stream, side_ev_join, 0) : e1;
        if (e2 != hipSuccess) (void)hipStreamSynchronize(s);  // the stream goes back to the pool idle either way
        zkw_pool_stream_release(s);
        HIP_TRY(e2);
        return ZKW_OK;
    }
    // witnesses and traces created from this context keep it alive: zkw_destroy defers while any is outstanding
    std::atomic<long> children{0};
    std::atomic<bool> destroy_requested{false};
    std::atomic<bool> destroying{false};
    bool chain_service = false;  // queue chains go to the device's chain service (batched with other contexts' chains)
    // which stage of which builder branch a chain submission is (zkw_block sets the tag of a branch's context; the sequence number counts
    // the context's submissions): with many blocks in flight the service batches EQUAL stages of all blocks into one launch (zkw_api.hip)
    int chain_tag = 0, chain_seq = 0;
    int next_chain_key() { return chain_tag ? chain_tag * 4096 + (chain_seq++ & 4095) : 0; }
    int chain_form = 0;  // lanes per Poseidon2 state in the queue-chain kernel: 4 (quad), 16 (row), 0 = auto
    int netlist_fill_form = 0;  // 0: a wave per cycle (k_nl_fill), 1: a lane per cycle (k_nl_walk + k_nl_expand)
    std::map<std::string, DevBuf> pool;  // named grow-only scratch
    std::map<std::string, HostStage> stages;
    // optional per-kernel timing with HIP events on the context's stream (zkw_profile_*)
    bool profiling = false;
    struct ProfSpan { const char* name; hipEvent_t a, b; };
    std::vector<ProfSpan> spans;
    std::vector<hipEvent_t> free_events;
    std::map<std::string, std::pair<double, uint64_t>> prof_totals;  // name -> (ms, launches)

    hipEvent_t prof_event() {
        if (!free_events.empty()) { hipEvent_t e = free_events.back(); free_events.pop_back(); return e; }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }
    int prof_collect() {
        if (spans.empty()) return ZKW_OK;
        HIP_TRY(hipStreamSynchronize(stream));
        for (auto& sp : spans) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, sp.a, sp.b) == hipSuccess) {
                auto& t = prof_totals[sp.name];
                t.first += ms;
                t.second += 1;
            }
            free_events.push_back(sp.a);
            free_events.push_back(sp.b);
        }
        spans.clear();
        return ZKW_OK;
    }

    int scratch(const char* name, size_t bytes, void** out) {
        DevBuf& b = pool[name];
        if (b.cap < bytes) {
            if (b.p) {
                retired_dev.push_back(b.p);
                b.p = nullptr;
                b.cap = 0;
            }
            size_t want = bytes + bytes / 8 + 256;
            HIP_TRY(dev_malloc(&b.p, want));
            b.cap = want;
        }
        *out = b.p;
        return ZKW_OK;
    }
    template <class T>
    int scratch_t(const char* name, size_t count, T** out) {
        void* p = nullptr;
        ZKW_TRY(scratch(name, count * sizeof(T) + 16, &p));
        *out = static_cast<T*>(p);
        return ZKW_OK;
    }
    // descriptor upload: host vector -> named device scratch (async, pageable source is copied by the
    // runtime before return)
    template <class T>
    int upload(const char* name, const std::vector<T>& h, T** out) {
        if (batched()) {  // the batch's upload arena: captured now, on the device before anything queued after this call runs
            *out = static_cast<T*>(zkw_batch_upload(batch, h.data(), h.size() * sizeof(T), alignof(T) > 16 ? alignof(T) : 16));
            return *out ? ZKW_OK : ZKW_ERR_OOM;
        }
        ZKW_TRY(scratch_t<T>(name, h.size() ? h.size() : 1, out));
        if (h.empty()) return ZKW_OK;
        const size_t bytes = h.size() * sizeof(T);
        HostStage& st = stages[name];
        if (st.pending) {  // the previous upload from this staging buffer must have been consumed
            HIP_TRY(hipEventSynchronize(st.ev));
            st.pending = false;
        }
        if (st.cap < bytes) {
            if (st.p) retired_host.push_back(st.p);
            st.p = nullptr;
            st.cap = 0;
            HIP_TRY(pin_malloc(&st.p, bytes + bytes / 2 + 256));
            st.cap = bytes + bytes / 2 + 256;
        }
        if (!st.ev) HIP_TRY(hipEventCreateWithFlags(&st.ev, hipEventDisableTiming));
        memcpy(st.p, h.data(), bytes);
        HIP_TRY(hipMemcpyAsync(*out, st.p, bytes, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipEventRecord(st.ev, stream));
        st.pending = true;
        return ZKW_OK;
    }
    // stage an input: returns a device pointer for `src` (copying when src is a host pointer)
    template <class T>
    int in(const char* name, const T* src, size_t count, const T** out) {
        if (ptr_mode == ZKW_PTR_DEVICE || count == 0) {
            *out = src;
            return ZKW_OK;
        }
        T* d = nullptr;
        ZKW_TRY(scratch_t<T>(name, count, &d));
        HIP_TRY(copy_async(d, src, count * sizeof(T), hipMemcpyHostToDevice));
        *out = d;
        return ZKW_OK;
    }
    // reserve an output: device pointer to write to (dst itself in device mode)
    template <class T>
    int out(const char* name, T* dst, size_t count, T** dev) {
        if (ptr_mode == ZKW_PTR_DEVICE) {
            *dev = dst;
            return ZKW_OK;
        }
        return scratch_t<T>(name, count ? count : 1, dev);
    }
    template <class T>
    int finish_out(T* dst, const T* dev, size_t count) {
        if (ptr_mode == ZKW_PTR_DEVICE || count == 0) return ZKW_OK;
        HIP_TRY(copy_async(dst, dev, count * sizeof(T), hipMemcpyDeviceToHost));
        return ZKW_OK;
    }
    int sync_if_host() {
        if (ptr_mode == ZKW_PTR_HOST) HIP_TRY(sync_stream());
        return ZKW_OK;
    }
    // Small device -> host readback (counts, violation flags) THROUGH PINNED MEMORY, then a sync of this stream only.
    // A hipMemcpyAsync into pageable memory waits for every stream of the device (measured: 0.9 s behind another
    // context's queue chain), which serialises the builders that zkw_block_run runs side by side.
    void* pinned_rb = nullptr;
    size_t pinned_rb_cap = 0;
    int read_small(void* dst, const void* src, size_t bytes) {
        if (batched()) {  // one gather + one copy for all the blocks of the batch that read something back at this point
            zkw_batch_copy_d2h(batch, dst, src, bytes);
            return zkw_batch_sync(batch);
        }
        if (pinned_rb_cap < bytes) {
            if (pinned_rb) retired_host.push_back(pinned_rb);
            pinned_rb = nullptr;
            pinned_rb_cap = 0;
            const size_t want = bytes < 4096 ? 4096 : bytes;
            HIP_TRY(pin_malloc(&pinned_rb, want));
            pinned_rb_cap = want;
        }
        HIP_TRY(hipMemcpyAsync(pinned_rb, src, bytes, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        memcpy(dst, pinned_rb, bytes);
        return ZKW_OK;
    }
    // buffers replaced by a bigger one: hipFree / hipHostFree wait for the whole device, so they are kept until the
    // context is destroyed (growth is geometric: bounded waste)
    std::vector<void*> retired_dev, retired_host;
};

// RAII span around one kernel launch (or a library sort); free when profiling is off
struct Prof {
    zkw_ctx* c;
    hipEvent_t b = nullptr;
    Prof(zkw_ctx* ctx, const char* name) : c(ctx) {
        if (!c->profiling || c->batched()) return;
        hipEvent_t a = c->prof_event();
        b = c->prof_event();
        (void)hipEventRecord(a, c->stream);
        c->spans.push_back(zkw_ctx::ProfSpan{name, a, b});
    }
    ~Prof() {
        if (b) (void)hipEventRecord(b, c->stream);
    }
};

static inline int launch_check(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(ZKW_ERR_HIP, "launch of %s failed: %s", what, hipGetErrorString(e));
    return ZKW_OK;
}

// Launch of a kernel BODY (zkw_launch.h) with a grid of gx x gy workgroups of BS threads: on the context's stream, or — a context of a
// batch — left with the batch to travel with the other blocks' launches of the same kernel. Arguments convert to the body's parameter types.
template <auto Body, int BS>
struct Launcher {
    using S = LaunchSig<decltype(Body)>;
    template <class... A>
    static int go(zkw_ctx* ctx, const char* name, dim3 grid, size_t lds_bytes, const A&... a) {
        if (grid.x == 0 || grid.y == 0 || grid.z == 0) return ZKW_OK;
        if (ctx->batched()) {
            typename S::T t;
            S::pack(t, a...);
            zkw_batch_launch(ctx->batch, S::template desc<Body, BS>(name), grid, lds_bytes, &t);
            return ZKW_OK;
        }
        S::template single<Body, BS>(ctx->stream, grid, lds_bytes, a...);  // (timed by the caller's Prof, if any)
        return launch_check(name);
    }
    // more than the default 64 KB of dynamic LDS for both launch forms of the kernel (once per device is enough; cheap to repeat)
    static int allow_dynamic_lds(int bytes) {
        HIP_TRY(hipFuncSetAttribute(S::template single_fn<Body, BS>(), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        HIP_TRY(hipFuncSetAttribute(S::template desc<Body, BS>("")->multi_fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        return ZKW_OK;
    }
};
#define ZKW_LAUNCH(ctx, kernel, gx, bs, ...) ZKW_TRY((Launcher<&kernel, bs>::go(ctx, #kernel, dim3((unsigned)(gx)), 0, __VA_ARGS__)))
#define ZKW_LAUNCH_2D(ctx, kernel, gx, gy, bs, ...) ZKW_TRY((Launcher<&kernel, bs>::go(ctx, #kernel, dim3((unsigned)(gx), (unsigned)(gy)), 0, __VA_ARGS__)))
// the same for a template kernel (the name has commas): ZKW_LAUNCH_T(ctx, (k_foo<A, B>), "k_foo", ...)
#define ZKW_LAUNCH_T(ctx, kernel, name, gx, bs, ...) ZKW_TRY((Launcher<&kernel, bs>::go(ctx, name, dim3((unsigned)(gx)), 0, __VA_ARGS__)))
// the general form: any dim3 grid, dynamic LDS
#define ZKW_LAUNCH_D(ctx, kernel, name, grid, bs, lds, ...) ZKW_TRY((Launcher<&kernel, bs>::go(ctx, name, grid, lds, __VA_ARGS__)))

// rows [0, width) of blockIdx.y's column of a column-major strip
static __device__ __forceinline__ void k_zero_strip(const VB& vb, u64* __restrict__ base, size_t pitch, size_t width) {
    const size_t i = (size_t)vb.x * blockDim.x + threadIdx.x;
    if (i < width) base[(size_t)vb.y * pitch + i] = 0;
}
// Zeroes what the fill of a "zkw trace v3" netlist circuit does NOT write itself: the general-purpose columns [0, g) (the
// fill then overwrites its header / gate cells), the lookup columns [g, g + lookup_cols) below the last cycle only (the fill
// writes every lookup cell of the cycles' rows, padding and header rows included), and the multiplicity columns. Zeroing
// the whole slot first wrote the lookup columns twice: a third of the memset.
static inline int zero_netlist_slot(zkw_ctx* ctx, u64* trace, size_t n_rows, size_t g, size_t lookup_cols, size_t n_cols, size_t used_rows) {
    HIP_TRY(hipMemsetAsync(trace, 0, g * n_rows * sizeof(u64), ctx->stream));
    if (used_rows < n_rows) {  // (hipMemset2DAsync ran this strip at 0.8 TB/s: 0.2 ms per Keccak slot)
        ZKW_LAUNCH_2D(ctx, k_zero_strip, (unsigned)((n_rows - used_rows + 255) / 256), (unsigned)lookup_cols, 256,
                           trace + g * n_rows + used_rows, n_rows, n_rows - used_rows);
        ZKW_TRY(launch_check("k_zero_strip"));
    }
    HIP_TRY(hipMemsetAsync(trace + (g + lookup_cols) * n_rows, 0, (n_cols - g - lookup_cols) * n_rows * sizeof(u64), ctx->stream));
    return ZKW_OK;
}

static inline unsigned blocks_for(size_t n, unsigned bs) { return (unsigned)((n + bs - 1) / bs); }

void ctx_retain(zkw_ctx* ctx);
zkw_ctx* zkw_ctx_create_in_batch(int device_id, zkw_batch* b);
void zkw_ctx_leave_batch(zkw_ctx* ctx, void* stream);

void ctx_release(zkw_ctx* ctx);

struct zkw_trace {
    zkw_ctx* ctx = nullptr;
    size_t n_rows = 0, n_cols = RC_COLS, n_slots = 0;
    u64* data = nullptr;
    size_t slot_elems() const { return n_cols * n_rows; }
    // What a slot held last (0 = unknown): a synthesis into a slot whose tag is its own layout only rewrites the cells its fill writes,
    // everything else is still zero from the same layout's previous tenant. Every other writer and zkw_trace_device_ptr reset the tag.
    // Sized at creation (zkw_trace_create_with_columns), relaxed atomics: a consumer thread may take a slot's pointer while a
    // producer synthesizes into another slot of the same trace.
    std::unique_ptr<std::atomic<uint64_t>[]> slot_tag;
    u64* slot_for_write(size_t slot, uint64_t tag) const {
        slot_tag[slot].store(tag, std::memory_order_relaxed);
        return data + slot * slot_elems();
    }
    uint64_t tag_of(size_t slot) const { return slot_tag[slot].load(std::memory_order_relaxed); }
};

// The slots one synthesis call writes. claim() reads a slot's tag and RESETS it; commit() — after the call's last launch has been
// enqueued — writes the new tags. A call that fails in between (upload, scratch, a launch) leaves its slots tagged "unknown", so the
// next synthesis into them is a cold one (the tag used to be set while the job list was built, and a failed call left a
// slot of the previous tenant's data under the new layout's tag).
struct SlotClaims {
    struct Claim { const zkw_trace* t; size_t slot; uint64_t tag; };
    const zkw_trace* t;
    std::vector<Claim> pending;
    explicit SlotClaims(const zkw_trace* tr = nullptr) : t(tr) {}
    u64* claim(size_t slot, uint64_t tag, bool* clean) { return claim(t, slot, tag, clean); }
    u64* claim(const zkw_trace* tr, size_t slot, uint64_t tag, bool* clean) {
        *clean = tr->tag_of(slot) == tag;
        pending.push_back(Claim{tr, slot, tag});
        return tr->slot_for_write(slot, 0);
    }
    void commit() {
        for (auto& p : pending) p.t->slot_for_write(p.slot, p.tag);
        pending.clear();
    }
    int commit_if(int rc) {
        if (rc == ZKW_OK) commit();
        return rc;
    }
};


// device-level steps (zkw_api.hip): every builder's queue chains, challenges and grand products go through these
int dev_encode(zkw_ctx* ctx, const zkw_mem_query* q, size_t n, u64* enc);
int dev_chains(zkw_ctx* ctx, const std::vector<ChainJob>& jobs);
int dev_fs(zkw_ctx* ctx, const std::vector<FsJob>& jobs, int state_w, int n_chal);
int dev_grand_products(zkw_ctx* ctx, std::vector<GpSeg>& segs, int width, int n_reps);
int dev_log_chains(zkw_ctx* ctx, const u64* d_enc, size_t total, std::vector<LogChainJob>& jobs);
// the chain launches as the chain service and the batch make them (row forms up to 4 096 chains, the quad form above)
int zkw_launch_chain_full(hipStream_t st, const ChainJob* d_jobs, int n_jobs);
int zkw_launch_chain_log(hipStream_t st, const LogChainJob* d_jobs, int n_jobs);
