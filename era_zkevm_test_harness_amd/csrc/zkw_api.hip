// zkw_api.hip — host side of libzkw above the kernels and the extern "C" boundary (include/zkw.h).
//
// Host logic mirrors the reference's builders (Rust, compiled code => C++ here):
//   RamBuilder::run  <->  compute_ram_circuit_snapshots, src/witness/individual_circuits/ram_permutation.rs:26-470
// One process drives one GPU; everything is enqueued on the context's stream, scratch lives in a
// grow-only per-context pool so that steady-state calls do no hipMalloc.
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
#include "ram_kernels.cuh"
#include "ram_circuit_kernels.cuh"
#include "log_kernels.cuh"
#include "decommit_kernels.cuh"
#include "events_kernels.cuh"
#include "demux_kernels.cuh"
#include "storage_kernels.cuh"
#include "decommitter_kernels.cuh"
#include "public_input_kernels.cuh"
#include "callstack_kernels.cuh"
#include "precompile_kernels.cuh"
#include "storage_application_kernels.cuh"
#include "decommit_sorter_circuit_kernels.cuh"
#include "events_sorter_circuit_kernels.cuh"
#include "log_demux_circuit_kernels.cuh"
#include "storage_sorter_circuit_kernels.cuh"
#include "vm_kernels.cuh"
#include "netlist_kernels.cuh"
#include "sort.h"

using namespace zkw;

// ------------------------------------------------------------------------------------------------ errors
static thread_local std::string g_last_error;

static int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}
int zkw_fail(int code, const char* fmt, ...) {  // the same for the library's other translation units (zkw_internal.h)
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

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
static inline hipError_t dev_malloc(void** p, size_t bytes) { return alloc_cache().alloc(0, p, bytes); }
template <class T> static inline hipError_t dev_malloc(T** p, size_t bytes) { return alloc_cache().alloc(0, (void**)p, bytes); }
static inline void dev_free(void* p) { alloc_cache().release(0, p); }
static inline hipError_t pin_malloc(void** p, size_t bytes) { return alloc_cache().alloc(1, p, bytes); }
static inline void pin_free(void* p) { alloc_cache().release(1, p); }


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

extern "C" void zkw_trim_caches(void) {
    alloc_cache().trim();
    stream_pool().trim();
}

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
    hipStream_t chain_stream = nullptr;  // optional second stream for the queue-chain kernels (zkw_set_chain_stream)
    hipEvent_t chain_ev_a = nullptr, chain_ev_b = nullptr;
    // witnesses and traces created from this context keep it alive: zkw_destroy defers while any is outstanding
    std::atomic<long> children{0};
    std::atomic<bool> destroy_requested{false};
    std::atomic<bool> destroying{false};
    bool chain_service = false;  // queue chains go to the device's chain service (batched with other contexts' chains)
    int chain_form = 0;  // lanes per Poseidon2 state in the queue-chain kernel: 4 (quad), 16 (row), 0 = auto
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
        HIP_TRY(hipMemcpyAsync(d, src, count * sizeof(T), hipMemcpyHostToDevice, stream));
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
        HIP_TRY(hipMemcpyAsync(dst, dev, count * sizeof(T), hipMemcpyDeviceToHost, stream));
        return ZKW_OK;
    }
    int sync_if_host() {
        if (ptr_mode == ZKW_PTR_HOST) HIP_TRY(hipStreamSynchronize(stream));
        return ZKW_OK;
    }
    // Small device -> host readback (counts, violation flags) THROUGH PINNED MEMORY, then a sync of this stream only.
    // A hipMemcpyAsync into pageable memory waits for every stream of the device (measured: 0.9 s behind another
    // context's queue chain), which serialises the builders that zkw_block_run runs side by side.
    void* pinned_rb = nullptr;
    size_t pinned_rb_cap = 0;
    int read_small(void* dst, const void* src, size_t bytes) {
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
        if (!c->profiling) return;
        hipEvent_t a = c->prof_event();
        b = c->prof_event();
        (void)hipEventRecord(a, c->stream);
        c->spans.push_back(zkw_ctx::ProfSpan{name, a, b});
    }
    ~Prof() {
        if (b) (void)hipEventRecord(b, c->stream);
    }
};

static int launch_check(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(ZKW_ERR_HIP, "launch of %s failed: %s", what, hipGetErrorString(e));
    return ZKW_OK;
}

// rows [0, width) of blockIdx.y's column of a column-major strip
__global__ __launch_bounds__(256) void k_zero_strip(u64* __restrict__ base, size_t pitch, size_t width) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < width) base[(size_t)blockIdx.y * pitch + i] = 0;
}
// Zeroes what the fill of a "zkw trace v3" netlist circuit does NOT write itself: the general-purpose columns [0, g) (the
// fill then overwrites its header / gate cells), the lookup columns [g, g + lookup_cols) below the last cycle only (the fill
// writes every lookup cell of the cycles' rows, padding and header rows included), and the multiplicity columns. Zeroing
// the whole slot first wrote the lookup columns twice: a third of the memset.
static int zero_netlist_slot(zkw_ctx* ctx, u64* trace, size_t n_rows, size_t g, size_t lookup_cols, size_t n_cols, size_t used_rows) {
    HIP_TRY(hipMemsetAsync(trace, 0, g * n_rows * sizeof(u64), ctx->stream));
    if (used_rows < n_rows) {  // (hipMemset2DAsync ran this strip at 0.8 TB/s: 0.2 ms per Keccak slot)
        hipLaunchKernelGGL(k_zero_strip, dim3((unsigned)((n_rows - used_rows + 255) / 256), (unsigned)lookup_cols), dim3(256), 0, ctx->stream,
                           trace + g * n_rows + used_rows, n_rows, n_rows - used_rows);
        ZKW_TRY(launch_check("k_zero_strip"));
    }
    HIP_TRY(hipMemsetAsync(trace + (g + lookup_cols) * n_rows, 0, (n_cols - g - lookup_cols) * n_rows * sizeof(u64), ctx->stream));
    return ZKW_OK;
}

static inline unsigned blocks_for(size_t n, unsigned bs) { return (unsigned)((n + bs - 1) / bs); }

extern "C" const char* zkw_last_error(void) { return g_last_error.c_str(); }
extern "C" const char* zkw_version(void) { return "zkw 0.2 (gfx950)"; }

extern "C" int zkw_circuit_geometry_of(uint8_t circuit_type, zkw_circuit_geometry* out) {
    // {copy columns, lookup width, repetitions, max degree, capacity, big size hint}: vm_main.rs:29-44,
    // sort_code_decommits.rs:28-39, code_decommitter.rs:28-39, log_demux.rs:36-47, keccak256_round_function.rs:28-39,
    // sha256_round_function.rs:28-39, ecrecover.rs:30-41, ram_permutation.rs:26-41,117-122, storage_sort_dedup.rs:29-40,
    // storage_apply.rs:28-39, events_sort_dedup.rs:28-39 (x2), linear_hasher.rs:28-39; geometry_config.rs:5-20
    static const struct { u32 c, lw, lr, deg, cap; bool big; } T[14] = {
        {0, 0, 0, 0, 0, false},
        {130, 3, 8, 8, 5585, false},    // 1 MainVM
        {130, 1, 18, 8, 117500, true},  // 2 CodeDecommittmentsSorter
        {108, 4, 11, 8, 2845, false},   // 3 CodeDecommitter
        {136, 1, 14, 8, 58750, true},   // 4 LogDemuxer
        {86, 3, 14, 8, 293, false},     // 5 KeccakRoundFunction
        {116, 4, 9, 8, 2206, false},    // 6 Sha256RoundFunction
        {80, 3, 16, 8, 7, false},       // 7 ECRecover
        {133, 1, 15, 8, 136714, true},  // 8 RAMPermutation
        {132, 1, 16, 8, 46921, true},   // 9 StorageSorter
        {60, 3, 26, 8, 33, false},      // 10 StorageApplication
        {130, 1, 8, 18, 31287, true},   // 11 EventsSorter
        {130, 1, 8, 18, 31287, true},   // 12 L1MessagesSorter
        {66, 3, 26, 8, 774, false},     // 13 L1MessagesHasher
    };
    if (!out || circuit_type < 1 || circuit_type > 13) return fail(ZKW_ERR_INVALID, "unknown base-layer circuit type %u", circuit_type);
    const auto& g = T[circuit_type];
    out->num_columns_under_copy_permutation = g.c;
    out->num_witness_columns = 0;
    out->num_constant_columns = 4;
    out->max_allowed_constraint_degree = g.deg;
    out->lookup_width = g.lw;
    out->lookup_repetitions = g.lr;
    out->capacity = g.cap;
    out->trace_len_log2 = 20;
    out->size_hint_variables = g.big ? (1ull << 26) + (1ull << 25) : (1ull << 26);
    return ZKW_OK;
}

// Where this library's own trace layout ("zkw trace v2") of a circuit type puts things: what the reference keeps in
// FinalizationHintsForProver / VerificationKey.fixed_parameters for ITS layout (setup/base_layer/finalization_hint_N.json:
// `public_inputs` = (column, row) of the four PI cells, `nop_gates_to_add`, `final_trace_len`). No GPU needed.
extern "C" int zkw_circuit_layout_of(uint8_t circuit_type, uint32_t capacity, zkw_circuit_layout* out) {
    if (!out) return fail(ZKW_ERR_INVALID, "zkw_circuit_layout_of: null argument");
    zkw_circuit_geometry g;
    ZKW_TRY(zkw_circuit_geometry_of(circuit_type, &g));
    memset(out, 0, sizeof *out);
    if (capacity == 0) capacity = g.capacity;
    out->capacity = capacity;
    out->trace_len = 1ull << g.trace_len_log2;
    uint64_t boundary = 0, min_rows = 0, pi_off = 0;
    switch (circuit_type) {
        case 8: out->num_columns = RC_COLS; out->rows_per_cycle = RC_ROWS_PER_CYCLE; out->region_stride = RC_REGION_STRIDE(capacity); boundary = RC_BOUNDARY_ROW(capacity); min_rows = RC_MIN_ROWS(capacity); pi_off = RC_ROWOFF_PI; break;
        case 2: out->num_columns = DS_COLS; out->rows_per_cycle = DS_ROWS_PER_CYCLE; out->region_stride = DS_REGION_STRIDE(capacity); boundary = DS_BOUNDARY_ROW(capacity); min_rows = DS_MIN_ROWS(capacity); pi_off = DS_ROWOFF_PI; break;
        case 4: out->num_columns = LD_COLS; out->rows_per_cycle = LD_ROWS_PER_CYCLE; out->region_stride = LD_REGION_STRIDE(capacity); boundary = LD_BOUNDARY_ROW(capacity); min_rows = LD_MIN_ROWS(capacity); pi_off = LD_ROWOFF_PI; break;
        case 9: out->num_columns = SS_COLS; out->rows_per_cycle = SS_ROWS_PER_CYCLE; out->region_stride = SS_REGION_STRIDE(capacity); boundary = SS_BOUNDARY_ROW(capacity); min_rows = SS_MIN_ROWS(capacity); pi_off = SS_ROWOFF_PI; break;
        case 11: case 12: out->num_columns = ES_COLS; out->rows_per_cycle = ES_ROWS_PER_CYCLE; out->region_stride = ES_REGION_STRIDE(capacity); boundary = ES_BOUNDARY_ROW(capacity); min_rows = ES_MIN_ROWS(capacity); pi_off = ES_ROWOFF_PI; break;
        // the netlist circuits ("zkw trace v4") are cycle-major: region_stride = 0, cycle i starts at row i * rows_per_cycle
        case 3: case 5: case 6: case 13: {
            const nl_spec* sp = nl_host_spec(circuit_type);
            const uint32_t cycles = circuit_type == 13 ? ZKW_LINEAR_HASHER_CYCLES(capacity) : capacity;
            out->num_columns = sp->cols; out->rows_per_cycle = sp->rows_per_cycle; boundary = NL_BOUNDARY_ROW(sp, cycles);
            min_rows = NL_USED_ROWS(sp, cycles); pi_off = 2 * NL_BND_ROWS(sp);
            out->total_table_rows = sp->total_table_rows;
            break;
        }
        default: return ZKW_OK;  // a known circuit type this library does not synthesize yet: synthesizable = 0
    }
    out->synthesizable = 1;
    if (out->region_stride) out->total_table_rows = 256;  // the queue circuits' one table: RangeCheckTable<8>
    out->rows_used = min_rows;
    out->fits = min_rows <= out->trace_len;
    out->nop_rows = out->fits ? out->trace_len - min_rows : 0;
    for (int k = 0; k < 4; k++) {
        out->public_input_column[k] = (uint32_t)k;
        out->public_input_row[k] = boundary + pi_off;
    }
    return ZKW_OK;
}

// Setup side, selectors: which gate set applies to each row of a trace of this library's layout — what the
// reference's setup keeps in its constant columns (gate selectors, the lookup table id of a row). Host arithmetic over the specs.
//   queue circuits (2, 4, 8, 9, 11, 12; "zkw trace v2", region-major): selector = row type of the spec (0 .. NUM_ROW_TYPES - 1:
//     the per-cycle row types, then the boundary rows), ZKW_ROW_PADDING elsewhere (gaps of a region, rows after the boundary)
//   netlist circuits (3, 5, 6, 13; "zkw trace v3", cycle-major): selector = lookup table id of the row (0: none) |
//     ZKW_ROW_HAS_GATES when ADD gates sit in its general-purpose columns | ZKW_ROW_HEADER for a cycle's first row;
//     boundary rows ZKW_ROW_BOUNDARY + k; ZKW_ROW_PADDING elsewhere
extern "C" int zkw_setup_row_selectors(uint8_t circuit_type, uint32_t capacity, size_t n_rows, uint8_t* out) {
    if (!out || n_rows == 0) return fail(ZKW_ERR_INVALID, "zkw_setup_row_selectors: null argument");
    zkw_circuit_layout lay;
    ZKW_TRY(zkw_circuit_layout_of(circuit_type, capacity, &lay));
    if (!lay.synthesizable) return fail(ZKW_ERR_INVALID, "circuit type %u has no layout in this library", (unsigned)circuit_type);
    if (lay.rows_used > n_rows) return fail(ZKW_ERR_INVALID, "capacity %u needs %llu rows, %zu given", lay.capacity, (unsigned long long)lay.rows_used, n_rows);
    memset(out, ZKW_ROW_PADDING, n_rows);
    const uint32_t cap = lay.capacity;
    if (lay.region_stride) {  // region-major
        const uint64_t stride = lay.region_stride, rpc = lay.rows_per_cycle;
        for (uint64_t r = 0; r < rpc; r++) memset(out + r * stride, (int)r, cap);
        const uint64_t bnd = rpc * stride;
        for (uint64_t k = 0; bnd + k < lay.rows_used; k++) out[bnd + k] = (uint8_t)(rpc + k);
        return ZKW_OK;
    }
    const uint32_t cycles = circuit_type == 13 ? ZKW_LINEAR_HASHER_CYCLES(cap) : cap;
    const uint64_t rpc = lay.rows_per_cycle;
    const nl_spec* sp = nl_host_spec(circuit_type);
    std::vector<uint8_t> one(rpc);  // every cycle has the same selectors
    for (uint32_t st = 0; st < sp->steps_per_cycle; st++) {
        const nl_step_type& T = sp->step_types[sp->cycle[st].type];
        uint8_t* row = one.data() + sp->cycle[st].row0;
        row[0] = ZKW_ROW_HEADER;
        for (uint32_t r = 1; r < T.rows; r++)
            row[r] = (uint8_t)((r <= T.lookup_rows ? sp->ops[T.op0 + (r - 1) * sp->r].table : 0) | (sp->gate_row_end[T.rowend0 + r] ? ZKW_ROW_HAS_GATES : 0));
    }
    for (uint32_t c = 0; c < cycles; c++) memcpy(out + (uint64_t)c * rpc, one.data(), rpc);
    for (uint64_t k = 0; (uint64_t)cycles * rpc + k < lay.rows_used; k++) out[(uint64_t)cycles * rpc + k] = (uint8_t)(ZKW_ROW_BOUNDARY + k);
    return ZKW_OK;
}

// Setup side, copy permutation of the queue circuits: sigma[c][r] = the cell (c' * n_rows + r') that follows
// cell (c, r) in its copy cycle; a cell under no copy constraint maps to itself. Built on the host from the spec's link table
// (the same table the satisfiability checker walks, ram_circuit_kernels.cuh k_check_links) with a union-find over the
// general-purpose cells; cycles run through their cells in increasing cell order. Seconds at production size (1.1 GB of output).
namespace {
struct LinkSpec { int G /* general-purpose + lookup columns: links reach both */, rows_per_cycle, num_links, off_bin, off_bout; const rc_link* links; };
static const rc_link h_rc_links[RC_NUM_LINKS] = RC_LINKS_INIT;
static const rc_link h_ds_links[DS_NUM_LINKS] = DS_LINKS_INIT;
static const rc_link h_es_links[ES_NUM_LINKS] = ES_LINKS_INIT;
static const rc_link h_ld_links[LD_NUM_LINKS] = LD_LINKS_INIT;
static const rc_link h_ss_links[SS_NUM_LINKS] = SS_LINKS_INIT;
bool link_spec_of(uint8_t t, LinkSpec* o) {
    switch (t) {
        case 8: *o = {RC_G + RC_L, RC_ROWS_PER_CYCLE, RC_NUM_LINKS, RC_ROWOFF_BND_IN, RC_ROWOFF_BND_OUT, h_rc_links}; return true;
        case 2: *o = {DS_G + DS_L, DS_ROWS_PER_CYCLE, DS_NUM_LINKS, DS_ROWOFF_BND_IN, DS_ROWOFF_BND_OUT, h_ds_links}; return true;
        case 11: case 12: *o = {ES_G + ES_L, ES_ROWS_PER_CYCLE, ES_NUM_LINKS, ES_ROWOFF_BND_IN, ES_ROWOFF_BND_OUT, h_es_links}; return true;
        case 4: *o = {LD_G + LD_L, LD_ROWS_PER_CYCLE, LD_NUM_LINKS, LD_ROWOFF_BND_IN, LD_ROWOFF_BND_OUT, h_ld_links}; return true;
        case 9: *o = {SS_G + SS_L, SS_ROWS_PER_CYCLE, SS_NUM_LINKS, SS_ROWOFF_BND_IN, SS_ROWOFF_BND_OUT, h_ss_links}; return true;
        default: return false;
    }
}
}  // namespace

extern "C" int zkw_setup_copy_permutation(uint8_t circuit_type, uint32_t capacity, size_t n_rows, uint64_t* sigma, uint32_t* n_columns) {
    LinkSpec sp;
    const bool netlist = circuit_type == 3 || circuit_type == 5 || circuit_type == 6 || circuit_type == 13;
    if (netlist) sp = {(int)nl_host_spec(circuit_type)->mult_col, 0, 0, 0, 0, nullptr};  // all but the multiplicity column
    else if (!link_spec_of(circuit_type, &sp))
        return fail(ZKW_ERR_INVALID, "zkw_setup_copy_permutation: circuit type %u has no layout in this library", (unsigned)circuit_type);
    zkw_circuit_layout lay;
    ZKW_TRY(zkw_circuit_layout_of(circuit_type, capacity, &lay));
    if (n_columns) *n_columns = (uint32_t)sp.G;
    if (!sigma) return ZKW_OK;  // size query
    if (lay.rows_used > n_rows || n_rows >= (1ull << 32) / (size_t)sp.G)
        return fail(ZKW_ERR_INVALID, "capacity %u needs %llu rows, %zu given", lay.capacity, (unsigned long long)lay.rows_used, n_rows);
    const uint32_t cap = lay.capacity;
    const uint64_t rs = lay.region_stride, bnd = (uint64_t)sp.rows_per_cycle * rs;
    const size_t n_cells = (size_t)sp.G * n_rows;
    std::vector<uint32_t> parent(n_cells);
    for (size_t i = 0; i < n_cells; i++) parent[i] = (uint32_t)i;
    auto find = [&](uint32_t x) {
        while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; }
        return x;
    };
    bool out_of_range = false;
    auto unite = [&](uint64_t col_a, uint64_t row_a, uint64_t col_b, uint64_t row_b) {
        if (col_a >= (uint64_t)sp.G || col_b >= (uint64_t)sp.G || row_a >= n_rows || row_b >= n_rows) { out_of_range = true; return; }
        uint32_t a = find((uint32_t)(col_a * n_rows + row_a)), b = find((uint32_t)(col_b * n_rows + row_b));
        if (a != b) parent[a > b ? a : b] = a > b ? b : a;  // the smallest cell of a class is its root
    };
    auto brow = [&](int rt) { return bnd + (uint64_t)(rt - sp.rows_per_cycle); };  // a boundary row type
    if (netlist) {
        // the netlist circuits: every operand cell of a lookup / gate is a copy of the cell that produced it (the references of
        // the spec, resolved exactly as the checkers do: k_kc_check_rows, k_sc_check_cycle); constants and free witness bytes
        // are under no copy constraint
        const uint32_t cycles = circuit_type == 13 ? ZKW_LINEAR_HASHER_CYCLES(cap) : cap;
        const nl_spec* ns = nl_host_spec(circuit_type);
        const uint64_t nb = NL_BOUNDARY_ROW(ns, cycles), brows = NL_BND_ROWS(ns);
        // the cell a reference names, seen from step st of cycle c (nl_home of netlist_kernels.cuh as coordinates); false: a constant
        auto home = [&](uint32_t c, uint32_t st, uint32_t ref, uint64_t* hc, uint64_t* hr) {
            for (;;) {
                const nl_cycle_step& cs = ns->cycle[st];
                const nl_step_type& T = ns->step_types[cs.type];
                const uint64_t base = (uint64_t)c * ns->rows_per_cycle + cs.row0;
                if (ref < NL_REF_HDR) {
                    const nl_home h = ns->homes[T.home0 + ref];
                    if (h.kind == 1) { const nl_gate& g = ns->gates[T.gate0 + h.item]; *hc = g.col + h.cell; *hr = base + g.row; }
                    else { *hc = ns->g + ns->w * (h.item % ns->r) + h.cell; *hr = base + 1 + h.item / ns->r; }
                    return true;
                }
                if (ref < NL_REF_PREV) { *hc = ref - NL_REF_HDR; *hr = base; return true; }
                if (ref >= NL_REF_FREE && ref < NL_REF_RC) return false;
                if (ref >= NL_REF_RC) return false;
                uint32_t k;
                if (ref >= NL_REF_CYC || st == 0) {
                    k = ref >= NL_REF_CYC ? ref - NL_REF_CYC : ref - NL_REF_PREV;
                    if (c == 0) { *hc = k % ns->g; *hr = nb + k / ns->g; return true; }
                    c--;
                    st = ns->steps_per_cycle - 1;
                } else {
                    k = ref - NL_REF_PREV;
                    st--;
                }
                ref = ns->out[(size_t)ns->cycle[st].type * ns->state + k];
            }
        };
        uint64_t hc, hr;
        for (uint32_t c = 0; c < cycles; c++)
            for (uint32_t st = 0; st < ns->steps_per_cycle; st++) {
                const nl_cycle_step& cs = ns->cycle[st];
                const nl_step_type& T = ns->step_types[cs.type];
                const uint64_t base = (uint64_t)c * ns->rows_per_cycle + cs.row0;
                if (st)  // a step's header is a copy of the cycle's
                    for (int f = 0; f < NL_HDR_FIELDS; f++) unite((uint64_t)f, base, (uint64_t)f, (uint64_t)c * ns->rows_per_cycle);
                for (uint32_t j = 0; j < T.n_ops; j++) {
                    const nl_op& op = ns->ops[T.op0 + j];
                    const nl_table& tb = ns->tables[op.table - 1];
                    for (uint32_t i = 0; i < tb.n_in; i++) {
                        if (op.in[i] < NL_REF_HDR) {
                            const nl_home h = ns->homes[T.home0 + op.in[i]];
                            if (h.kind == 2 && h.item == j && h.cell == i) continue;  // a hint's own cell
                        }
                        if (home(c, st, op.in[i], &hc, &hr)) unite((uint64_t)(ns->g + ns->w * (j % ns->r) + i), base + 1 + j / ns->r, hc, hr);
                    }
                }
                for (uint32_t gi = 0; gi < T.n_gates; gi++) {
                    const nl_gate& g = ns->gates[T.gate0 + gi];
                    for (uint32_t i = 0; i < g.n_known; i++)
                        if (home(c, st, ns->terms[T.term0 + g.first_term + i].ref, &hc, &hr)) unite((uint64_t)(g.col + i), base + g.row, hc, hr);
                }
            }
        if (cycles)
            for (uint32_t k = 0; k < ns->state; k++)
                if (home(cycles, 0, NL_REF_CYC + k, &hc, &hr)) unite((uint64_t)(k % ns->g), nb + brows + k / ns->g, hc, hr);
    }
    for (int l = 0; l < sp.num_links; l++) {
        const rc_link k = sp.links[l];
        if (k.kind == 3) { unite(k.col_a, bnd + sp.off_bout, k.col_b, (uint64_t)k.row_b * rs + cap - 1); continue; }
        if (k.kind == 4) { unite(k.col_a, brow(k.row_a), k.col_b, bnd + sp.off_bout); continue; }
        if (k.kind == 5) { unite(k.col_a, brow(k.row_a), k.col_b, brow(k.row_b)); continue; }
        for (uint32_t i = 0; i < cap; i++) {
            const uint64_t ra = (uint64_t)k.row_a * rs + i;
            if (k.kind == 0) unite(k.col_a, ra, k.col_b, (uint64_t)k.row_b * rs + i);
            else if (k.kind == 1) { if (i) unite(k.col_a, ra, k.col_b, (uint64_t)k.row_b * rs + i - 1); else unite(k.col_a, ra, k.bin_col, bnd + sp.off_bin); }
            else unite(k.col_a, ra, k.col_b, bnd + sp.off_bin);
        }
    }
    if (out_of_range) return fail(ZKW_ERR_INVALID, "zkw_setup_copy_permutation: a link of the spec leaves the copy-permutation columns");
    // cycles: the cells of a class in increasing order, the last one back to the root
    std::vector<uint32_t> last(n_cells);
    for (size_t i = 0; i < n_cells; i++) { sigma[i] = i; last[i] = (uint32_t)i; }
    for (size_t i = 0; i < n_cells; i++) {
        const uint32_t r = find((uint32_t)i);
        if (r == i) continue;
        sigma[last[r]] = i;  // i > last[r]: cells are visited in increasing order
        last[r] = (uint32_t)i;
        sigma[i] = r;
    }
    return ZKW_OK;
}

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
    if (ctx->own_stream) {
        (void)hipStreamSynchronize(ctx->own_stream);
        stream_pool().release(ctx->own_stream);
    }
    delete ctx;
}
static void ctx_retain(zkw_ctx* ctx) { ctx->children.fetch_add(1); }
static void ctx_release(zkw_ctx* ctx);
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
static void ctx_release(zkw_ctx* ctx) {
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

extern "C" int zkw_synchronize(zkw_ctx* ctx) {
    if (!ctx) return fail(ZKW_ERR_INVALID, "null context");
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return ZKW_OK;
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
    };
    int device = 0;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::shared_ptr<Batch> open;   // the batch that still accepts jobs
    std::chrono::steady_clock::time_point open_since, last_arrival;
    std::vector<std::thread> workers;
    bool stop = false;

    explicit ChainService(int dev) : device(dev) {
        for (int i = 0; i < 4; i++) workers.emplace_back([this] { run(); });
    }
    ~ChainService() {
        { std::lock_guard<std::mutex> g(mu); stop = true; }
        cv_work.notify_all();
        for (auto& t : workers) t.join();
    }
    int submit(const std::vector<ChainJob>* full, const std::vector<LogChainJob>* log, std::string* err) {
        std::shared_ptr<Batch> b;
        {
            std::unique_lock<std::mutex> lk(mu);
            if (!open) { open = std::make_shared<Batch>(); open_since = std::chrono::steady_clock::now(); }
            b = open;
            if (full) b->full.insert(b->full.end(), full->begin(), full->end());
            if (log) b->log.insert(b->log.end(), log->begin(), log->end());
            b->waiters++;
            last_arrival = std::chrono::steady_clock::now();
            cv_work.notify_one();
            const int kind = full ? 0 : 1;
            cv_done.wait(lk, [&] { return b->done[kind]; });
        }
        if (b->rc != ZKW_OK && err) *err = b->err;
        return b->rc;
    }
    void run() {
        (void)hipSetDevice(device);
        hipStream_t st = nullptr, st_log = nullptr;  // the two kinds of chains of a batch run side by side, not one after the other
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (hipStreamCreateWithPriority(&st, hipStreamNonBlocking, hi) != hipSuccess) st = nullptr;
        if (hipStreamCreateWithPriority(&st_log, hipStreamNonBlocking, hi) != hipSuccess) st_log = nullptr;
        void* pin = nullptr;
        void* dev = nullptr;
        size_t cap = 0;
        for (;;) {
            std::shared_ptr<Batch> b;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&] { return stop || open; });
                if (stop) break;
                // batching window: launch when nothing has arrived for 400 us, or 4 ms after the first job
                for (;;) {
                    const auto now = std::chrono::steady_clock::now();
                    if (!open) break;  // another worker took it
                    if (now - last_arrival >= std::chrono::microseconds(400) || now - open_since >= std::chrono::milliseconds(4)) break;
                    cv_work.wait_for(lk, std::chrono::microseconds(200));
                    if (stop) break;
                }
                if (stop) break;
                if (!open) continue;
                b = open;
                open.reset();
            }
            int rc = ZKW_OK;
            std::string err;
            auto fail_hip = [&](hipError_t e, const char* what) { if (e != hipSuccess && rc == ZKW_OK) { rc = ZKW_ERR_HIP; err = std::string(what) + ": " + hipGetErrorString(e); } };
            const size_t bytes = b->full.size() * sizeof(ChainJob) + b->log.size() * sizeof(LogChainJob) + 256;
            if (!st || !st_log) { rc = ZKW_ERR_HIP; err = "chain service: no stream"; }
            if (rc == ZKW_OK && cap < bytes) {  // grow-only; the outgrown pair goes back to the allocation cache (no hipFree stall)
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
            if (rc == ZKW_OK) {
                char* hp = static_cast<char*>(pin);
                char* dp = static_cast<char*>(dev);
                const size_t off_log = (b->full.size() * sizeof(ChainJob) + 127) & ~(size_t)127;
                if (!b->full.empty()) memcpy(hp, b->full.data(), b->full.size() * sizeof(ChainJob));
                if (!b->log.empty()) memcpy(hp + off_log, b->log.data(), b->log.size() * sizeof(LogChainJob));
                fail_hip(hipMemcpyAsync(dp, hp, off_log + b->log.size() * sizeof(LogChainJob), hipMemcpyHostToDevice, st), "job upload");
                fail_hip(hipStreamSynchronize(st), "job upload");
                const int nf = (int)b->full.size(), nl = (int)b->log.size();
                if (rc == ZKW_OK && nl) hipLaunchKernelGGL(k_chain_log, dim3((nl + 3) / 4), dim3(64), 0, st_log, reinterpret_cast<const LogChainJob*>(dp + off_log), nl);
                if (rc == ZKW_OK && nf) {
                    if (nf >= 4096) hipLaunchKernelGGL(k_chain_full_q4, dim3((nf + 15) / 16), dim3(64), 0, st, reinterpret_cast<const ChainJob*>(dp), nf);
                    else hipLaunchKernelGGL(k_chain_full, dim3((nf + 3) / 4), dim3(64), 0, st, reinterpret_cast<const ChainJob*>(dp), nf);
                }
                fail_hip(hipGetLastError(), "chain launch");
                fail_hip(hipStreamSynchronize(st_log), "chain batch (log queues)");
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

extern "C" int zkw_set_chain_service(zkw_ctx* ctx, int on) {
    if (!ctx) return fail(ZKW_ERR_INVALID, "null context");
    ctx->chain_service = on != 0;
    if (on) (void)chain_service_of(ctx->device);
    return ZKW_OK;
}

// hand the chains of one builder call over to the service and wait for them
static int chain_service_run(zkw_ctx* ctx, const std::vector<ChainJob>* full, const std::vector<LogChainJob>* log, const char* name) {
    HIP_TRY(hipStreamSynchronize(ctx->stream));  // the jobs' inputs are produced on this context's stream
    const auto t0 = std::chrono::steady_clock::now();
    std::string err;
    const int rc = chain_service_of(ctx->device)->submit(full, log, &err);
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

static int dev_encode(zkw_ctx* ctx, const zkw_mem_query* q, size_t n, u64* enc) {
    if (n == 0) return ZKW_OK;
    unsigned grid = blocks_for(n, 256);
    if (grid > 256 * 16) grid = 256 * 16;
    { Prof _p(ctx, "k_encode_mem"); hipLaunchKernelGGL(k_encode_mem, dim3(grid), dim3(256), 0, ctx->stream, q, n, enc); }
    return launch_check("k_encode_mem");
}

// Chains: one chain per 16-lane DPP row, 4 chains per wave, one wave per block. A single wave already
// issues a VALU instruction every ~2 cycles (measured 3.7 us per permutation step, flat from 1 to 4096
// concurrent chains), so throughput comes from giving each wave its own SIMD: up to 1024 waves.
static int dev_chains(zkw_ctx* ctx, const std::vector<ChainJob>& jobs) {
    if (jobs.empty()) return ZKW_OK;
    if (ctx->chain_service) return chain_service_run(ctx, &jobs, nullptr, "k_chain_full");
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
        else if (form == 4) hipLaunchKernelGGL(k_chain_full_q4, dim3((n_jobs + 15) / 16), dim3(64), 0, st, d_jobs, n_jobs);
        else if (form == 2) hipLaunchKernelGGL(k_chain_full_p2, dim3((n_jobs + 31) / 32), dim3(64), 0, st, d_jobs, n_jobs);
        else hipLaunchKernelGGL(k_chain_full_lane, dim3((n_jobs + 63) / 64), dim3(64), 0, st, d_jobs, n_jobs);
        if (ctx->chain_stream) {  // the profiling events live on the context's stream: bring the kernel's end onto it first
            HIP_TRY(hipEventRecord(ctx->chain_ev_b, ctx->chain_stream));
            HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->chain_ev_b, 0));
        }
    }
    return launch_check(name);
}

static int dev_fs(zkw_ctx* ctx, const std::vector<FsJob>& jobs, int state_w, int n_chal) {
    if (jobs.empty()) return ZKW_OK;
    FsJob* d_jobs = nullptr;
    ZKW_TRY(ctx->upload("fs_jobs", jobs, &d_jobs));
    int n = (int)jobs.size();
    { Prof _p(ctx, "k_fs_challenges"); hipLaunchKernelGGL(k_fs_challenges, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, d_jobs, n, state_w, n_chal); }
    return launch_check("k_fs_challenges");
}

template <int W, int REPS>
static int gp_launch(zkw_ctx* ctx, const GpSeg* d_segs, int n_segs, const GpTile* d_tiles, unsigned n_tiles,
                     u64* d_aggr) {
    { Prof _p(ctx, "k_gp_local"); hipLaunchKernelGGL((k_gp_local<W, REPS>), dim3(n_tiles), dim3(GP_BLOCK), 0, ctx->stream, d_segs, d_tiles, d_aggr); }
    ZKW_TRY(launch_check("k_gp_local"));
    { Prof _p(ctx, "k_gp_tiles"); hipLaunchKernelGGL((k_gp_tiles<REPS>), dim3((n_segs * REPS + 63) / 64), dim3(64), 0, ctx->stream, d_segs, n_segs,
                       d_aggr); }
    ZKW_TRY(launch_check("k_gp_tiles"));
    { Prof _p(ctx, "k_gp_apply"); hipLaunchKernelGGL((k_gp_apply<REPS>), dim3(n_tiles), dim3(GP_BLOCK), 0, ctx->stream, d_segs, d_tiles, d_aggr); }
    return launch_check("k_gp_apply");
}

// segs: rows/z/chal/n filled by the caller; first_tile/n_tiles filled here
static int dev_grand_products(zkw_ctx* ctx, std::vector<GpSeg>& segs, int width, int n_reps) {
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
    { Prof _p(ctx, "k_encode_log"); hipLaunchKernelGGL(k_encode_log, dim3(blocks_for(n, 256)), dim3(256), 0, ctx->stream, d_q, n, d_e, d_enc); }
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
    { Prof _p(ctx, "k_encode_decommit"); hipLaunchKernelGGL(k_encode_decommit, dim3(blocks_for(n, 256)), dim3(256), 0, ctx->stream, d_q, n, d_enc); }
    ZKW_TRY(launch_check("k_encode_decommit"));
    ZKW_TRY(ctx->finish_out(enc, d_enc, n * 8));
    return ctx->sync_if_host();
}

// device-level: rounds 1-2 of every item in parallel, then one serial permutation per item and queue
static int dev_log_chains(zkw_ctx* ctx, const u64* d_enc, size_t total, std::vector<LogChainJob>& jobs) {
    if (total == 0 || jobs.empty()) return ZKW_OK;
    u64* d_pre = nullptr;
    ZKW_TRY(ctx->scratch_t<u64>("log_pre", total * 4, &d_pre));
    { Prof _p(ctx, "k_log_prehash"); hipLaunchKernelGGL(k_log_prehash, dim3(blocks_for(total, 128)), dim3(128), 0, ctx->stream, d_enc, total, d_pre); }
    ZKW_TRY(launch_check("k_log_prehash"));
    for (auto& j : jobs) j.pre = d_pre + (j.enc - d_enc) / 20 * 4;
    if (ctx->chain_service) return chain_service_run(ctx, nullptr, &jobs, "k_chain_log");
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

__global__ void k_block_ids(const u64* __restrict__ offsets, int n_blocks, size_t n, u32* __restrict__ ids) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int lo = 0, hi = n_blocks;  // largest b with offsets[b] <= i
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (offsets[mid] <= i) lo = mid; else hi = mid;
    }
    ids[i] = (u32)lo;
}

// the sorting permutation for all blocks at once (see sort.hip)
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
    { Prof _p(ctx, "k_ram_sort_keys"); hipLaunchKernelGGL(k_ram_sort_keys, dim3(grid), dim3(256), 0, ctx->stream, d_q, total, ts, cell, v0,
                       (const u64*)nullptr, 0); }
    ZKW_TRY(launch_check("k_ram_sort_keys"));
    // pass 1: timestamp
    { Prof _p(ctx, "radix_sort"); HIP_TRY(radix_sort_pairs_u32(tmp, tmp_bytes, ts, k32, v0, v1, total, 32, ctx->stream)); }
    // pass 2: cell of the ts-sorted items
    { Prof _p(ctx, "k_gather_u64_by_u32"); hipLaunchKernelGGL(k_gather_u64_by_u32, dim3(grid), dim3(256), 0, ctx->stream, cell, v1, total, k64a); }
    ZKW_TRY(launch_check("k_gather_u64_by_u32"));
    { Prof _p(ctx, "radix_sort"); HIP_TRY(radix_sort_pairs_u64(tmp, tmp_bytes, k64a, k64b, v1, v0, total, 64, ctx->stream)); }
    u32* perm = v0;
    if (n_blocks > 1) {
        // pass 3: block id, so that each block's items end up contiguous again
        u64* d_off = nullptr;
        ZKW_TRY(ctx->upload("sort_off", offsets, &d_off));
        unsigned bits = 1;
        while ((1ull << bits) < n_blocks) bits++;
        { Prof _p(ctx, "k_block_ids"); hipLaunchKernelGGL(k_block_ids, dim3(grid), dim3(256), 0, ctx->stream, d_off, (int)n_blocks, total, ts); }
        ZKW_TRY(launch_check("k_block_ids"));
        // ts[] now holds block ids in ORIGINAL order; gather them through the current permutation
        { Prof _p(ctx, "k_gather_u32_by_u32"); hipLaunchKernelGGL(k_gather_u32_by_u32, dim3(grid), dim3(256), 0, ctx->stream, ts, v0, total, k32); }
        ZKW_TRY(launch_check("k_gather_u32_by_u32"));
        { Prof _p(ctx, "radix_sort"); HIP_TRY(radix_sort_pairs_u32(tmp, tmp_bytes, k32, ts, v0, v1, total, bits, ctx->stream)); }
        perm = v1;
    }
    *perm_out = perm;
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
        HIP_TRY(hipMemcpyAsync(w->owned_q, d_q, total * sizeof(zkw_mem_query), hipMemcpyDeviceToDevice, ctx->stream));
        w->unsorted_q = w->owned_q;
    }
    // K7 (sorted side)
    u32* perm = nullptr;
    ZKW_TRY(ram_sort(ctx, d_q, total, w->offsets, w->unsorted_caps, w->sorted_caps, &perm));
    w->tails_valid = false;
    w->enc_valid = false;
    w->sorted_valid = false;
    // the permutation lives in the sort's scratch (the capacity-word arrays the chains are about to fill): keep a copy
    HIP_TRY(hipMemcpyAsync(w->perm, perm, total * sizeof(u32), hipMemcpyDeviceToDevice, ctx->stream));
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
        { Prof _p(ctx, "k_ram_count_nondet"); hipLaunchKernelGGL(k_ram_count_nondet, dim3(gx, gy), dim3(256), 0, ctx->stream, d_blocks); }
        ZKW_TRY(launch_check("k_ram_count_nondet"));
        { Prof _p(ctx, "k_ram_instances"); hipLaunchKernelGGL(k_ram_instances, dim3(blocks_for(max_inst, 64), gy), dim3(64), 0, ctx->stream, d_blocks); }
        ZKW_TRY(launch_check("k_ram_instances"));
        b0 = b1;
    }
    // a20: compact forms and public inputs of every instance (postprocessing/mod.rs:353-369)
    const size_t ni = w->n_instances;
    { Prof _p(ctx, "k_ram_commitments"); hipLaunchKernelGGL(k_ram_commitments, dim3(blocks_for(4 * ni, 64)), dim3(64), 0, ctx->stream, w->instances, ni, w->compact_forms); }
    ZKW_TRY(launch_check("k_ram_commitments"));
    { Prof _p(ctx, "k_commit_encodings"); hipLaunchKernelGGL(k_commit_encodings, dim3(blocks_for(ni, 64)), dim3(64), 0, ctx->stream, w->compact_forms, ni, (u32)COMPACT_FORM_LEN, w->public_inputs); }
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
    { Prof _p(ctx, "k_gather_encode"); hipLaunchKernelGGL(k_gather_encode, dim3(blocks_for(t, 256)), dim3(256), 0, ctx->stream, w->unsorted_q, w->perm, t, w->sorted_q, (u64*)nullptr); }
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
    { Prof _p(ctx, "k_tails_expand"); hipLaunchKernelGGL(k_tails_expand, dim3(blocks_for(t, 64)), dim3(64), 0, ctx->stream, w->unsorted_enc, w->unsorted_caps, d_off, nq, t, w->unsorted_tails); }
    ZKW_TRY(launch_check("k_tails_expand"));
    { Prof _p(ctx, "k_tails_expand"); hipLaunchKernelGGL(k_tails_expand, dim3(blocks_for(t, 64)), dim3(64), 0, ctx->stream, w->sorted_enc, w->sorted_caps, d_off, nq, t, w->sorted_tails); }
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
    HIP_TRY(hipMemcpyAsync(dst, src, bytes,
                           ctx->ptr_mode == ZKW_PTR_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost,
                           ctx->stream));
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
struct zkw_trace {
    zkw_ctx* ctx = nullptr;
    size_t n_rows = 0, n_cols = RC_COLS, n_slots = 0;
    u64* data = nullptr;
    size_t slot_elems() const { return n_cols * n_rows; }
    // What a slot held last, for the netlist circuits (which then only rewrite the cells their fill writes: everything else is
    // still zero from the same layout's previous tenant). 0 = unknown; every other writer and zkw_trace_device_ptr reset it.
    mutable std::vector<uint64_t> slot_tag;
    u64* slot_for_write(size_t slot, uint64_t tag) const {
        if (slot_tag.size() != n_slots) slot_tag.assign(n_slots, 0);
        slot_tag[slot] = tag;
        return data + slot * slot_elems();
    }
    uint64_t tag_of(size_t slot) const { return slot_tag.size() == n_slots ? slot_tag[slot] : 0; }
};

extern "C" int zkw_trace_create_with_columns(zkw_ctx* ctx, size_t n_rows, size_t n_cols, size_t n_slots, zkw_trace** out) {
    if (!ctx || !out || n_rows < 256 || n_slots == 0 || n_cols == 0 || n_cols > 4096)
        return fail(ZKW_ERR_INVALID, "zkw_trace_create: bad argument");
    HIP_TRY(hipSetDevice(ctx->device));
    zkw_trace* t = new zkw_trace();
    t->ctx = ctx;
    t->n_rows = n_rows;
    t->n_cols = n_cols;
    t->n_slots = n_slots;
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
    HIP_TRY(hipMemcpyAsync(dst, src, (size_t)n_cols * t->n_rows * sizeof(u64),
                           ctx->ptr_mode == ZKW_PTR_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, ctx->stream));
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
        hipLaunchKernelGGL(k_gather_encode, dim3(blocks_for(cnt, 256)), dim3(256), 0, ctx->stream, w->unsorted_q, w->perm + z_base, cnt, w->sq_win, (u64*)nullptr);
    }
    ZKW_TRY(launch_check("k_gather_encode"));
    const u32 rstride = (u32)RC_REGION_STRIDE(capacity);  // rows per region incl. the alignment gap
    const u32 n_tiles = (rstride + 255) / 256;
    u32 *d_hist = nullptr, *d_nd = nullptr;
    ZKW_TRY(ctx->scratch_t<u32>("synth_hist", n_instances * 256, &d_hist));
    ZKW_TRY(ctx->scratch_t<u32>("synth_nd", n_instances * n_tiles, &d_nd));
    HIP_TRY(hipMemsetAsync(d_hist, 0, n_instances * 256 * sizeof(u32), ctx->stream));
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
        j.trace = t->slot_for_write((first_slot + k) % t->n_slots, 0);
        j.hist = d_hist + 256 * k;
        j.nd_tiles = d_nd + (size_t)n_tiles * k;
        j.public_input = w->public_inputs + 4 * idx;
    }
    SynthJob* d_jobs = nullptr;
    ZKW_TRY(ctx->upload("synth_jobs", jobs, &d_jobs));
    const unsigned nj = (unsigned)n_instances;
    const dim3 g64((rstride + 63) / 64, nj), g256(n_tiles, nj);
    { Prof _p(ctx, "k_ram_nd_tiles"); hipLaunchKernelGGL(k_ram_nd_tiles, g256, dim3(256), 0, ctx->stream, d_jobs, capacity); }
    ZKW_TRY(launch_check("k_ram_nd_tiles"));
    { Prof _p(ctx, "k_ram_nd_scan"); hipLaunchKernelGGL(k_ram_nd_scan, dim3(nj), dim3(64), 0, ctx->stream, d_jobs, (int)nj, n_tiles); }
    ZKW_TRY(launch_check("k_ram_nd_scan"));
    { Prof _p(ctx, "k_ram_fill_poseidon"); hipLaunchKernelGGL((k_ram_fill_poseidon<0>), g64, dim3(64), 0, ctx->stream, d_jobs, capacity, n_rows); }
    ZKW_TRY(launch_check("k_ram_fill_poseidon<0>"));
    { Prof _p(ctx, "k_ram_fill_poseidon"); hipLaunchKernelGGL((k_ram_fill_poseidon<1>), g64, dim3(64), 0, ctx->stream, d_jobs, capacity, n_rows); }
    ZKW_TRY(launch_check("k_ram_fill_poseidon<1>"));
    { Prof _p(ctx, "k_ram_fill_A"); hipLaunchKernelGGL(k_ram_fill_A, g256, dim3(256), 0, ctx->stream, d_jobs, capacity, n_rows); }
    ZKW_TRY(launch_check("k_ram_fill_A"));
    { Prof _p(ctx, "k_ram_fill_B"); hipLaunchKernelGGL(k_ram_fill_B, g256, dim3(256), 0, ctx->stream, d_jobs, capacity, n_rows); }
    ZKW_TRY(launch_check("k_ram_fill_B"));
    { Prof _p(ctx, "k_ram_fill_C"); hipLaunchKernelGGL(k_ram_fill_C, g256, dim3(256), 0, ctx->stream, d_jobs, capacity, n_rows); }
    ZKW_TRY(launch_check("k_ram_fill_C"));
    { Prof _p(ctx, "k_ram_fill_D"); hipLaunchKernelGGL(k_ram_fill_D, g256, dim3(256), 0, ctx->stream, d_jobs, capacity, n_rows); }
    ZKW_TRY(launch_check("k_ram_fill_D"));
    { Prof _p(ctx, "k_ram_fill_tail"); hipLaunchKernelGGL(k_ram_fill_tail, dim3((RC_G + RC_L + 1) * TAIL_CHUNKS, nj), dim3(256), 0, ctx->stream, d_jobs, capacity, n_rows); }
    ZKW_TRY(launch_check("k_ram_fill_tail"));
    { Prof _p(ctx, "k_ram_fill_boundary"); hipLaunchKernelGGL(k_ram_fill_boundary, dim3(nj), dim3(64), 0, ctx->stream, d_jobs, capacity, n_rows); }
    ZKW_TRY(launch_check("k_ram_fill_boundary"));
    return ctx->sync_if_host();
}

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
    { Prof _p(ctx, "k_encode_decommit"); hipLaunchKernelGGL(k_encode_decommit, dim3(grid), dim3(256), 0, ctx->stream, d_q, n, w->unsorted_enc); }
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
    { Prof _p(ctx, "k_decommit_sort_keys"); hipLaunchKernelGGL(k_decommit_sort_keys, dim3(grid), dim3(256), 0, ctx->stream, d_q, n, ts, hk[0], hk[1], hk[2], hk[3], v0); }
    ZKW_TRY(launch_check("k_decommit_sort_keys"));
    { Prof _p(ctx, "radix_sort"); HIP_TRY(radix_sort_pairs_u32(tmp, tmp_bytes, ts, k32, v0, v1, n, 32, ctx->stream)); }
    u32 *cur = v1, *nxt = v0;
    for (int k = 0; k < 4; k++) {
        { Prof _p(ctx, "k_gather_u64_by_u32"); hipLaunchKernelGGL(k_gather_u64_by_u32, dim3(grid), dim3(256), 0, ctx->stream, hk[k], cur, n, k64a); }
        ZKW_TRY(launch_check("k_gather_u64_by_u32"));
        { Prof _p(ctx, "radix_sort"); HIP_TRY(radix_sort_pairs_u64(tmp, tmp_bytes, k64a, k64b, cur, nxt, n, 64, ctx->stream)); }
        u32* t = cur; cur = nxt; nxt = t;
    }
    { Prof _p(ctx, "k_decommit_gather_encode"); hipLaunchKernelGGL(k_decommit_gather_encode, dim3(grid), dim3(256), 0, ctx->stream, d_q, cur, n, w->sorted_q, w->sorted_enc); }
    ZKW_TRY(launch_check("k_decommit_gather_encode"));
    // deduplicated queue = the fresh requests in sorted order
    u32 *fresh_count = nullptr, *last_fresh = nullptr, *totals = nullptr;
    ZKW_TRY(ctx->scratch_t<u32>("dec_fresh", n, &fresh_count));
    ZKW_TRY(ctx->scratch_t<u32>("dec_lastf", n, &last_fresh));
    ZKW_TRY(ctx->scratch_t<u32>("dec_totals", 2, &totals));
    { Prof _p(ctx, "k_decommit_dedup"); hipLaunchKernelGGL(k_decommit_dedup, dim3(1), dim3(1024), 0, ctx->stream, w->sorted_q, w->sorted_enc, n, fresh_count, last_fresh, w->dedup_q, w->dedup_enc, totals); }
    ZKW_TRY(launch_check("k_decommit_dedup"));
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
    { Prof _p(ctx, "k_decommit_instances"); hipLaunchKernelGGL(k_decommit_instances, dim3(blocks_for(w->n_instances, 64)), dim3(64), 0, ctx->stream, d_blk); }
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
        { Prof _p(ctx, "k_ds_commitments"); hipLaunchKernelGGL(k_ds_commitments, dim3(blocks_for(4 * ni, 64)), dim3(64), 0, ctx->stream, w->instances, ni, w->compact_forms); }
        rc = launch_check("k_ds_commitments");
        if (rc == ZKW_OK) {
            { Prof _p(ctx, "k_commit_encodings"); hipLaunchKernelGGL(k_commit_encodings, dim3(blocks_for(ni, 64)), dim3(64), 0, ctx->stream, w->compact_forms, ni, (u32)COMPACT_FORM_LEN, w->public_inputs); }
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
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, ctx->ptr_mode == ZKW_PTR_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, ctx->stream));
    return ctx->sync_if_host();
}
extern "C" void zkw_decommit_witness_free(zkw_decommit_witness* w) {
    if (!w) return;
    (void)hipSetDevice(w->ctx->device);
    (void)hipStreamSynchronize(w->ctx->stream);
    w->release();
    zkw_ctx* owner = w->ctx;
    delete w;
    ctx_release(owner);
}

// ------------------------------------------------------------------------------------------------ events sorter
// a20 for the 4-wide log-queue circuits: compact closed-form inputs [ni][18] followed by the public inputs [ni][4]
// in one allocation (ClosedFormInputCompactForm::from_full_form + commit, postprocessing/mod.rs:353-369)
template <class T>
static int closed_form_public_inputs(zkw_ctx* ctx, const typename T::Inst* d_inst, size_t ni, u64** cf_pi) {
    if (!*cf_pi) HIP_TRY(dev_malloc((void**)cf_pi, ni * (COMPACT_FORM_LEN + 4) * sizeof(u64)));
    u64 *compact = *cf_pi, *pis = *cf_pi + COMPACT_FORM_LEN * ni;
    { Prof _p(ctx, "k_closed_form_commitments"); hipLaunchKernelGGL((k_closed_form_commitments<T>), dim3(blocks_for(4 * ni, CfLanes<T>::value)), dim3(CfLanes<T>::value), 0, ctx->stream, d_inst, ni, compact); }
    ZKW_TRY(launch_check("k_closed_form_commitments"));
    { Prof _p(ctx, "k_commit_encodings"); hipLaunchKernelGGL(k_commit_encodings, dim3(blocks_for(ni, 64)), dim3(64), 0, ctx->stream, compact, ni, (u32)COMPACT_FORM_LEN, pis); }
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
    { Prof _p(ctx, "k_closed_form_commitments"); hipLaunchKernelGGL((k_closed_form_commitments<T>), dim3(blocks_for(4 * n, CfLanes<T>::value)), dim3(CfLanes<T>::value), 0, ctx->stream, d_inst, n, d_cf); }
    ZKW_TRY(launch_check("k_closed_form_commitments"));
    { Prof _p(ctx, "k_commit_encodings"); hipLaunchKernelGGL(k_commit_encodings, dim3(blocks_for(n, 64)), dim3(64), 0, ctx->stream, d_cf, n, (u32)COMPACT_FORM_LEN, d_pi); }
    ZKW_TRY(launch_check("k_commit_encodings"));
    ZKW_TRY(ctx->finish_out(reinterpret_cast<u64*>(compact), d_cf, n * COMPACT_FORM_LEN));
    ZKW_TRY(ctx->finish_out(reinterpret_cast<u64*>(public_inputs), d_pi, n * 4));
    return ctx->sync_if_host();
}

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
    { Prof _p(ctx, "k_encode_log"); hipLaunchKernelGGL(k_encode_log, dim3(grid), dim3(256), 0, ctx->stream, d_q, n, (const u32*)nullptr, u_enc); }
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
    { Prof _p(ctx, "k_events_sort_keys"); hipLaunchKernelGGL(k_events_sort_keys, dim3(grid), dim3(256), 0, ctx->stream, d_q, n, key, v0); }
    ZKW_TRY(launch_check("k_events_sort_keys"));
    { Prof _p(ctx, "radix_sort"); HIP_TRY(radix_sort_pairs_u64(tmp, tmp_bytes, key, key_out, v0, v1, n, 33, ctx->stream)); }
    { Prof _p(ctx, "k_log_gather_encode"); hipLaunchKernelGGL(k_log_gather_encode, dim3(grid), dim3(256), 0, ctx->stream, d_q, v1, n, w->sorted_q, s_enc); }
    ZKW_TRY(launch_check("k_log_gather_encode"));
    ZKW_TRY(ctx->scratch_t<u32>("evt_kept", n, &kept));
    ZKW_TRY(ctx->scratch_t<u32>("evt_totals", 2, &totals));
    { Prof _p(ctx, "k_events_dedup"); hipLaunchKernelGGL(k_events_dedup, dim3(1), dim3(1024), 0, ctx->stream, w->sorted_q, n, kept, w->result_q, r_enc, totals); }
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
    { Prof _p(ctx, "k_events_instances"); hipLaunchKernelGGL(k_events_instances, dim3(blocks_for(w->n_instances, 64)), dim3(64), 0, ctx->stream, d_blk); }
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
        if (hipMemcpy(w->instances, &inst, sizeof inst, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemset(w->challenges, 0, 42 * 8) != hipSuccess)
            rc = fail(ZKW_ERR_HIP, "copy failed");
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
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, ctx->ptr_mode == ZKW_PTR_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, ctx->stream));
    return ctx->sync_if_host();
}
extern "C" void zkw_events_witness_free(zkw_events_witness* w) {
    if (!w) return;
    (void)hipSetDevice(w->ctx->device);
    (void)hipStreamSynchronize(w->ctx->stream);
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
    { Prof _p(ctx, "k_encode_log"); hipLaunchKernelGGL(k_encode_log, dim3(blocks_for(n, 256)), dim3(256), 0, ctx->stream, d_q, n, (const u32*)nullptr, in_enc); }
    ZKW_TRY(launch_check("k_encode_log"));
    u32* route_count = w->route_count;
    { Prof _p(ctx, "k_demux_route"); hipLaunchKernelGGL(k_demux_route, dim3(1), dim3(1024), 0, ctx->stream, d_q, in_enc, n, params, route_count, w->out_q, out_enc, w->d_offsets); }
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
    { Prof _p(ctx, "k_demux_instances"); hipLaunchKernelGGL(k_demux_instances, dim3(blocks_for(w->n_instances, 64)), dim3(64), 0, ctx->stream, d_blk); }
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
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, ctx->ptr_mode == ZKW_PTR_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, ctx->stream));
    return ctx->sync_if_host();
}
extern "C" void zkw_demux_witness_free(zkw_demux_witness* w) {
    if (!w) return;
    (void)hipSetDevice(w->ctx->device);
    (void)hipStreamSynchronize(w->ctx->stream);
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
    u32* scans = nullptr;  // [4][n]: D, S, R, E of k_storage_cells (kept for synthesis)
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
    { Prof _p(ctx, "k_storage_sort_keys"); hipLaunchKernelGGL(k_storage_sort_keys, dim3(grid), dim3(256), 0, ctx->stream, d_q, n, kk[0], kk[1], kk[2], kk[3], kk[4], kk[5], a2, iota); }
    ZKW_TRY(launch_check("k_storage_sort_keys"));
    // plain and extended encodings of the unsorted side
    { Prof _p(ctx, "k_encode_log"); hipLaunchKernelGGL(k_encode_log, dim3(grid), dim3(256), 0, ctx->stream, d_q, n, (const u32*)nullptr, u_enc); }
    ZKW_TRY(launch_check("k_encode_log"));
    { Prof _p(ctx, "k_encode_log"); hipLaunchKernelGGL(k_encode_log, dim3(grid), dim3(256), 0, ctx->stream, d_q, n, (const u32*)iota, w->lhs_enc); }
    ZKW_TRY(launch_check("k_encode_log"));
    HIP_TRY(hipMemcpyAsync(v0, iota, n * sizeof(u32), hipMemcpyDeviceToDevice, ctx->stream));
    u32 *cur = v0, *nxt = v1;
    for (int k = 0; k < 6; k++) {
        { Prof _p(ctx, "k_gather_u64_by_u32"); hipLaunchKernelGGL(k_gather_u64_by_u32, dim3(grid), dim3(256), 0, ctx->stream, kk[k], cur, n, k64a); }
        ZKW_TRY(launch_check("k_gather_u64_by_u32"));
        { Prof _p(ctx, "radix_sort"); HIP_TRY(radix_sort_pairs_u64(tmp, tmp_bytes, k64a, k64b, cur, nxt, n, 64, ctx->stream)); }
        u32* t = cur; cur = nxt; nxt = t;
    }
    { Prof _p(ctx, "k_gather_u32_by_u32"); hipLaunchKernelGGL(k_gather_u32_by_u32, dim3(grid), dim3(256), 0, ctx->stream, a2, cur, n, k32a); }
    ZKW_TRY(launch_check("k_gather_u32_by_u32"));
    { Prof _p(ctx, "radix_sort"); HIP_TRY(radix_sort_pairs_u32(tmp, tmp_bytes, k32a, k32b, cur, nxt, n, 32, ctx->stream)); }
    { u32* t = cur; cur = nxt; nxt = t; }
    { Prof _p(ctx, "k_storage_gather_encode"); hipLaunchKernelGGL(k_storage_gather_encode, dim3(grid), dim3(256), 0, ctx->stream, d_q, cur, n, w->sorted_q, w->sorted_ext, s_enc); }
    ZKW_TRY(launch_check("k_storage_gather_encode"));
    // per-cell registers and the deduplicated queue
    StorageScan sc;
    u32* totals = nullptr;
    sc.D = reinterpret_cast<int*>(w->scans);
    sc.S = w->scans + n;
    sc.R = w->scans + 2 * n;
    sc.E = w->scans + 3 * n;
    ZKW_TRY(ctx->scratch_t<u32>("sto_totals", 2, &totals));
    { Prof _p(ctx, "k_storage_cells"); hipLaunchKernelGGL(k_storage_cells, dim3(1), dim3(1024), 0, ctx->stream, w->sorted_q, n, sc, w->result_q, r_enc, totals); }
    ZKW_TRY(launch_check("k_storage_cells"));
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
    { Prof _p(ctx, "k_storage_instances"); hipLaunchKernelGGL(k_storage_instances, dim3(blocks_for(w->n_instances, 64)), dim3(64), 0, ctx->stream, d_blk); }
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
        if (hipMemcpy(w->instances, &inst, sizeof inst, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemset(w->challenges, 0, 42 * 8) != hipSuccess)
            rc = fail(ZKW_ERR_HIP, "copy failed");
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
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, ctx->ptr_mode == ZKW_PTR_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, ctx->stream));
    return ctx->sync_if_host();
}
extern "C" void zkw_storage_witness_free(zkw_storage_witness* w) {
    if (!w) return;
    (void)hipSetDevice(w->ctx->device);
    (void)hipStreamSynchronize(w->ctx->stream);
    w->release();
    zkw_ctx* owner = w->ctx;
    delete w;
    ctx_release(owner);
}

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
    void release() {
        void* ptrs[] = {mem_q, mem_enc, mem_tails, round_states, instances, sha256_rounds, cf_pi};
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
        { Prof _p(ctx, "k_decommitter_mem_queries"); hipLaunchKernelGGL(k_decommitter_mem_queries, dim3(blocks_for(total, 256)), dim3(256), 0, ctx->stream, job, (u64)total); }
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
    if (hipMemsetAsync(d_viol, 0, 4, ctx->stream) != hipSuccess) return bail(fail(ZKW_ERR_HIP, "memset failed"));
    DecommitterJob job{d_req, d_words, d_woff, d_roff, w->round_states, w->mem_q, w->mem_enc, d_viol, n_requests, w->sha256_rounds};
    { Prof _p(ctx, "k_decommitter_sha"); hipLaunchKernelGGL(k_decommitter_sha, dim3(blocks_for(n_requests, 64)), dim3(64), 0, ctx->stream, job); }
    if ((rc = launch_check("k_decommitter_sha")) != ZKW_OK) return bail(rc);
    { Prof _p(ctx, "k_decommitter_mem_queries"); hipLaunchKernelGGL(k_decommitter_mem_queries, dim3(blocks_for(w->total_words, 256)), dim3(256), 0, ctx->stream, job, (u64)w->total_words); }
    if ((rc = launch_check("k_decommitter_mem_queries")) != ZKW_OK) return bail(rc);
    zkw_queue_state12* d_min = nullptr;
    std::vector<zkw_queue_state12> minv(1, *mem_in);
    if ((rc = ctx->upload("dcm_mem_in", minv, &d_min)) != ZKW_OK) return bail(rc);
    if (given_mem_tails) {  // the caller has already hashed the memory queue this slice belongs to (zkw_block_run)
        if (hipMemcpyAsync(w->mem_tails, given_mem_tails, w->total_words * 96,
                           ctx->ptr_mode == ZKW_PTR_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
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
    { Prof _p(ctx, "k_decommitter_instances"); hipLaunchKernelGGL(k_decommitter_instances, dim3(blocks_for(w->n_instances, 64)), dim3(64), 0, ctx->stream, d_blk); }
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
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, ctx->ptr_mode == ZKW_PTR_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, ctx->stream));
    return ctx->sync_if_host();
}
extern "C" void zkw_decommitter_witness_free(zkw_decommitter_witness* w) {
    if (!w) return;
    (void)hipSetDevice(w->ctx->device);
    (void)hipStreamSynchronize(w->ctx->stream);
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
    { Prof _p(ctx, "k_linear_keccak256"); hipLaunchKernelGGL(k_linear_keccak256, dim3(1), dim3(64), 0, ctx->stream, d_q, n, d_out, (zkw_keccak_round_record*)nullptr, (const u64*)nullptr, (const u64*)nullptr); }
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

// ------------------------------------------------------------------------------------------------ public inputs (a20)
extern "C" int zkw_commit_encodings(zkw_ctx* ctx, const uint64_t* enc, size_t n_items, uint32_t item_len, uint64_t* out) {
    if (!ctx || !out || (n_items && item_len && !enc)) return fail(ZKW_ERR_INVALID, "zkw_commit_encodings: null argument");
    if (n_items == 0) return ZKW_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    const u64* d_enc = nullptr;
    u64* d_out = nullptr;
    ZKW_TRY(ctx->in("ce_enc", enc, n_items * item_len + 1, &d_enc));
    ZKW_TRY(ctx->out("ce_out", out, n_items * 4, &d_out));
    { Prof _p(ctx, "k_commit_encodings"); hipLaunchKernelGGL(k_commit_encodings, dim3(blocks_for(n_items, 64)), dim3(64), 0, ctx->stream, d_enc, n_items, item_len, d_out); }
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
    { Prof _p(ctx, "k_encode_recursion"); hipLaunchKernelGGL(k_encode_recursion, dim3(blocks_for(n, 64)), dim3(64), 0, ctx->stream, circuit_type, d_pi, n, d_enc); }
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
    { Prof _p(ctx, "k_stack_depth"); hipLaunchKernelGGL(k_stack_depth, dim3(1), dim3(1024), 0, ctx->stream, d_ops, n_ops, d_depth, d_rank, d_meta); }
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
        { Prof _p(ctx, "radix_sort"); HIP_TRY(radix_sort_pairs_u32(tmp, tmp_bytes, d_pd, d_sd, d_pid, d_sid, n_push, bits, ctx->stream)); }
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
    void release() {
        void* ptrs[] = {mem_enc, mem_tails, instances, keccak_rounds, sha256_rounds, cf_pi};
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
        ZKW_TRY(ctx->scratch_t<u64>("pc_roff", n_requests + 1, &d_roff));
        ZKW_TRY(ctx->scratch_t<u64>("pc_qoff", n_requests + 1, &d_qoff));
        ZKW_TRY(ctx->scratch_t<u64>("pc_rdoff", n_requests + 1, &d_rdoff));
        ZKW_TRY(ctx->scratch_t<u64>("pc_meta", 4, &d_meta));
        { Prof _p(ctx, "k_precompile_counts"); hipLaunchKernelGGL(k_precompile_counts, dim3(1), dim3(1024), 0, ctx->stream, kind, d_req, n_requests, d_roff, d_qoff, d_rdoff, d_meta); }
        ZKW_TRY(launch_check("k_precompile_counts"));
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
    auto bail = [&](int rc) { w->release(); delete w; return rc; };
    if (e != hipSuccess) return bail(fail(ZKW_ERR_OOM, "zkw_precompile_build: hipMalloc failed: %s", hipGetErrorString(e)));
    int rc = ZKW_OK;
    PrecompileSnap* d_snaps = nullptr;
    u32* d_viol = nullptr;
    if ((rc = ctx->scratch_t<u32>("pc_viol", 1, &d_viol)) != ZKW_OK) return bail(rc);
    if (hipMemsetAsync(d_viol, 0, 4, ctx->stream) != hipSuccess) return bail(fail(ZKW_ERR_HIP, "memset failed"));
    if (n_requests) {
        if ((rc = ctx->scratch_t<PrecompileSnap>("pc_snaps", w->n_instances, &d_snaps)) != ZKW_OK) return bail(rc);
        if (n_queries) {
            if ((rc = dev_encode(ctx, d_mq, n_queries, w->mem_enc)) != ZKW_OK) return bail(rc);
            zkw_queue_state12* d_min = nullptr;
            std::vector<zkw_queue_state12> minv(1, *mem_in);
            if ((rc = ctx->upload("pc_mem_in", minv, &d_min)) != ZKW_OK) return bail(rc);
            if (given_mem_tails) {  // already hashed by the caller as part of the whole memory queue (zkw_block_run)
                if (hipMemcpyAsync(w->mem_tails, given_mem_tails, n_queries * 96,
                                   ctx->ptr_mode == ZKW_PTR_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
                    return bail(fail(ZKW_ERR_HIP, "copy of the given memory-queue states failed"));
            } else {
                std::vector<ChainJob> chains(1, ChainJob{w->mem_enc, w->mem_tails, d_min->tail, n_queries});
                if ((rc = dev_chains(ctx, chains)) != ZKW_OK) return bail(rc);
            }
        }
        PrecompileJob job{kind, d_req, d_mq, d_roff, d_qoff, d_rdoff, d_snaps, d_viol, n_requests, w->total_rounds, capacity, w->keccak_rounds, w->sha256_rounds};
        { Prof _p(ctx, "k_precompile_walk"); hipLaunchKernelGGL(k_precompile_walk, dim3(blocks_for(n_requests, 64)), dim3(64), 0, ctx->stream, job); }
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
    { Prof _p(ctx, "k_precompile_instances"); hipLaunchKernelGGL(k_precompile_instances, dim3(blocks_for(w->n_instances, 64)), dim3(64), 0, ctx->stream, d_blk); }
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
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, ctx->ptr_mode == ZKW_PTR_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, ctx->stream));
    return ctx->sync_if_host();
}
extern "C" void zkw_precompile_witness_free(zkw_precompile_witness* w) {
    if (!w) return;
    (void)hipSetDevice(w->ctx->device);
    (void)hipStreamSynchronize(w->ctx->stream);
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
    void release() {
        void* ptrs[] = {keys, paths, roots, leaf_indexes, instances};
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
    job.keys = w->keys; job.paths = w->paths; job.roots = w->roots;
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
    if (hipMemsetAsync(d_viol, 0, 4, ctx->stream) != hipSuccess) return bail(fail(ZKW_ERR_HIP, "memset failed"));
    u64 meta[2] = {1, initial_next_enumeration_index};
    if (n) {
        const unsigned g64 = blocks_for(n, 64);
        { Prof _p(ctx, "k_sap_keys"); hipLaunchKernelGGL(k_sap_keys, dim3(g64), dim3(64), 0, ctx->stream, job); }
        TRY(launch_check("k_sap_keys"));
        { Prof _p(ctx, "k_sap_scan"); hipLaunchKernelGGL(k_sap_scan, dim3(1), dim3(1024), 0, ctx->stream, job); }
        TRY(launch_check("k_sap_scan"));
        { Prof _p(ctx, "k_sap_pairs"); hipLaunchKernelGGL(k_sap_pairs, dim3(g64), dim3(64), 0, ctx->stream, job); }
        TRY(launch_check("k_sap_pairs"));
        { Prof _p(ctx, "k_sap_leaves"); hipLaunchKernelGGL(k_sap_leaves, dim3(g64), dim3(64), 0, ctx->stream, job); }
        TRY(launch_check("k_sap_leaves"));
        static_assert(ZKW_STORAGE_TREE_DEPTH == 256, "k_sap_levels walks 256 levels");
        if (n <= SAP_PERSISTENT_MAX) {
            Prof _p(ctx, "k_sap_levels");
            hipLaunchKernelGGL(k_sap_levels, dim3(1), dim3(SAP_PERSISTENT_THREADS), 0, ctx->stream, job);
        } else {
            for (int L = 0; L < ZKW_STORAGE_TREE_DEPTH && rc == ZKW_OK; L++) {
                Prof _p(ctx, "k_sap_level");
                hipLaunchKernelGGL(k_sap_level, dim3(g64), dim3(64), 0, ctx->stream, job, L);
            }
        }
        TRY(launch_check("k_sap_level"));
        { Prof _p(ctx, "k_sap_roots"); hipLaunchKernelGGL(k_sap_roots, dim3(g64), dim3(64), 0, ctx->stream, job); }
        TRY(launch_check("k_sap_roots"));
        if (rc != ZKW_OK) return bail(rc);
        if (hipMemcpyAsync(w->leaf_indexes, job.init_index, n * 8, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess ||
            ctx->read_small(meta, d_meta, sizeof meta) != ZKW_OK)
            return bail(fail(ZKW_ERR_HIP, "readback failed"));
    }
    w->n_instances = n ? (size_t)meta[0] : 1;
    if (dev_malloc((void**)&w->instances, w->n_instances * sizeof(zkw_storage_application_instance) + 64) != hipSuccess)
        return bail(fail(ZKW_ERR_OOM, "zkw_storage_application_build: hipMalloc failed"));
    SapKeccakOut ko{d_snap, d_hash};
    { Prof _p(ctx, "k_sap_keccak"); hipLaunchKernelGGL(k_sap_keccak, dim3(1), dim3(64), 0, ctx->stream, job, ko); }
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
    { Prof _p(ctx, "k_sap_instances"); hipLaunchKernelGGL(k_sap_instances, dim3(blocks_for(w->n_instances, 64)), dim3(64), 0, ctx->stream, d_blk); }
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
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, ctx->ptr_mode == ZKW_PTR_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, ctx->stream));
    return ctx->sync_if_host();
}
extern "C" void zkw_storage_application_witness_free(zkw_storage_application_witness* w) {
    if (!w) return;
    (void)hipSetDevice(w->ctx->device);
    (void)hipStreamSynchronize(w->ctx->stream);
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
        { Prof _p(ctx, "k_ds_fresh_prefix"); hipLaunchKernelGGL(k_ds_fresh_prefix, dim3(1), dim3(1024), 0, ctx->stream, w->sorted_q, w->n, w->fresh_prefix); }
        ZKW_TRY(launch_check("k_ds_fresh_prefix"));
    }
    u32* d_hist = nullptr;
    ZKW_TRY(ctx->scratch_t<u32>("ds_hist", n_instances * 256, &d_hist));
    HIP_TRY(hipMemsetAsync(d_hist, 0, n_instances * 256 * sizeof(u32), ctx->stream));
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
        j.trace = t->slot_for_write((first_slot + k) % t->n_slots, 0);
        j.hist = d_hist + 256 * k;
        j.public_input = w->public_inputs + 4 * (first_instance + k);
    }
    DsSynthJob* d_jobs = nullptr;
    ZKW_TRY(ctx->upload("ds_jobs", jobs, &d_jobs));
    const unsigned nj = (unsigned)n_instances;
    const u32 rstride = (u32)DS_REGION_STRIDE(capacity);
    const dim3 g64((rstride + 63) / 64, nj), g256((rstride + 255) / 256, nj);
    { Prof _p(ctx, "k_ds_fill_poseidon"); hipLaunchKernelGGL((k_ds_fill_poseidon<0>), g64, dim3(64), 0, ctx->stream, d_jobs, capacity, n_rows); }
    ZKW_TRY(launch_check("k_ds_fill_poseidon<0>"));
    { Prof _p(ctx, "k_ds_fill_poseidon"); hipLaunchKernelGGL((k_ds_fill_poseidon<1>), g64, dim3(64), 0, ctx->stream, d_jobs, capacity, n_rows); }
    ZKW_TRY(launch_check("k_ds_fill_poseidon<1>"));
    { Prof _p(ctx, "k_ds_fill_poseidon"); hipLaunchKernelGGL((k_ds_fill_poseidon<2>), g64, dim3(64), 0, ctx->stream, d_jobs, capacity, n_rows); }
    ZKW_TRY(launch_check("k_ds_fill_poseidon<2>"));
    { Prof _p(ctx, "k_ds_fill_row_A"); hipLaunchKernelGGL((k_ds_fill_row<DS_ROW_A>), g256, dim3(256), 0, ctx->stream, d_jobs, capacity, n_rows); }
    ZKW_TRY(launch_check("k_ds_fill_row<A>"));
    { Prof _p(ctx, "k_ds_fill_row_B"); hipLaunchKernelGGL((k_ds_fill_row<DS_ROW_B>), g256, dim3(256), 0, ctx->stream, d_jobs, capacity, n_rows); }
    ZKW_TRY(launch_check("k_ds_fill_row<B>"));
    { Prof _p(ctx, "k_ds_fill_row_C"); hipLaunchKernelGGL((k_ds_fill_row<DS_ROW_C>), g256, dim3(256), 0, ctx->stream, d_jobs, capacity, n_rows); }
    ZKW_TRY(launch_check("k_ds_fill_row<C>"));
    { Prof _p(ctx, "k_ds_fill_row_D"); hipLaunchKernelGGL((k_ds_fill_row<DS_ROW_D>), g256, dim3(256), 0, ctx->stream, d_jobs, capacity, n_rows); }
    ZKW_TRY(launch_check("k_ds_fill_row<D>"));
    { Prof _p(ctx, "k_ds_fill_tail"); hipLaunchKernelGGL(k_ds_fill_tail, dim3((DS_G + DS_L + 1) * TAIL_CHUNKS, nj), dim3(256), 0, ctx->stream, d_jobs, capacity, n_rows); }
    ZKW_TRY(launch_check("k_ds_fill_tail"));
    { Prof _p(ctx, "k_ds_fill_boundary"); hipLaunchKernelGGL(k_ds_fill_boundary, dim3(nj), dim3(64), 0, ctx->stream, d_jobs, capacity, n_rows); }
    ZKW_TRY(launch_check("k_ds_fill_boundary"));
    return ctx->sync_if_host();
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
        { Prof _p(ctx, "k_es_kept_prefix"); hipLaunchKernelGGL(k_es_kept_prefix, dim3(1), dim3(1024), 0, ctx->stream, w->sorted_q, n, w->kept_prefix); }
        ZKW_TRY(launch_check("k_es_kept_prefix"));
    }
    u32* d_hist = nullptr;
    ZKW_TRY(ctx->scratch_t<u32>("es_hist", n_instances * 256, &d_hist));
    HIP_TRY(hipMemsetAsync(d_hist, 0, n_instances * 256 * sizeof(u32), ctx->stream));
    const size_t m = n ? n : 1;
    u64 *u_enc = w->enc_all, *s_enc = w->enc_all + 20 * m;
    u64 *u_new = w->tails_all + 4 * m, *s_new = w->tails_all + 12 * m, *r_new = w->tails_all + 16 * m;
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
        j.trace = t->slot_for_write((first_slot + k) % t->n_slots, 0);
        j.hist = d_hist + 256 * k;
    }
    EsSynthJob* d_jobs = nullptr;
    ZKW_TRY(ctx->upload("es_jobs", jobs, &d_jobs));
    const unsigned nj = (unsigned)n_instances;
    const u32 rstride = (u32)ES_REGION_STRIDE(capacity);
    const dim3 g64((rstride + 63) / 64, nj), g256((rstride + 255) / 256, nj);
    { Prof _p(ctx, "k_es_fill_queue"); hipLaunchKernelGGL((k_es_fill_queue<0>), g64, dim3(64), 0, ctx->stream, d_jobs, capacity, n_rows); }
    ZKW_TRY(launch_check("k_es_fill_queue<0>"));
    { Prof _p(ctx, "k_es_fill_queue"); hipLaunchKernelGGL((k_es_fill_queue<1>), g64, dim3(64), 0, ctx->stream, d_jobs, capacity, n_rows); }
    ZKW_TRY(launch_check("k_es_fill_queue<1>"));
    { Prof _p(ctx, "k_es_fill_queue"); hipLaunchKernelGGL((k_es_fill_queue<2>), g64, dim3(64), 0, ctx->stream, d_jobs, capacity, n_rows); }
    ZKW_TRY(launch_check("k_es_fill_queue<2>"));
#define ES_LAUNCH_ROW(R) { Prof _p(ctx, "k_es_fill_row"); hipLaunchKernelGGL((k_es_fill_row<ES_ROW_##R>), g256, dim3(256), 0, ctx->stream, d_jobs, capacity, n_rows); } \
    ZKW_TRY(launch_check("k_es_fill_row<" #R ">"));
    ES_LAUNCH_ROW(A) ES_LAUNCH_ROW(N0) ES_LAUNCH_ROW(N1) ES_LAUNCH_ROW(N2) ES_LAUNCH_ROW(N3) ES_LAUNCH_ROW(N4) ES_LAUNCH_ROW(N5)
    ES_LAUNCH_ROW(N6) ES_LAUNCH_ROW(N7) ES_LAUNCH_ROW(T) ES_LAUNCH_ROW(V) ES_LAUNCH_ROW(W) ES_LAUNCH_ROW(Q)
#undef ES_LAUNCH_ROW
    { Prof _p(ctx, "k_es_fill_tail"); hipLaunchKernelGGL(k_es_fill_tail, dim3((ES_G + ES_L + 1) * TAIL_CHUNKS, nj), dim3(256), 0, ctx->stream, d_jobs, capacity, n_rows); }
    ZKW_TRY(launch_check("k_es_fill_tail"));
    { Prof _p(ctx, "k_es_fill_boundary"); hipLaunchKernelGGL(k_es_fill_boundary, dim3(nj), dim3(64), 0, ctx->stream, d_jobs, capacity, n_rows); }
    ZKW_TRY(launch_check("k_es_fill_boundary"));
    return ctx->sync_if_host();
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
    HIP_TRY(hipMemsetAsync(d_hist, 0, n_instances * 256 * sizeof(u32), ctx->stream));
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
        j.trace = t->slot_for_write((first_slot + k) % t->n_slots, 0);
        j.hist = d_hist + 256 * k;
    }
    LdSynthJob* d_jobs = nullptr;
    ZKW_TRY(ctx->upload("ld_jobs", jobs, &d_jobs));
    const unsigned nj = (unsigned)n_instances;
    const u32 rstride = (u32)LD_REGION_STRIDE(capacity);
    const dim3 g64((rstride + 63) / 64, nj), g256((rstride + 255) / 256, nj);
    { Prof _p(ctx, "k_ld_fill_queue"); hipLaunchKernelGGL((k_ld_fill_queue<0>), g64, dim3(64), 0, ctx->stream, d_jobs, capacity, n_rows); }
    ZKW_TRY(launch_check("k_ld_fill_queue<0>"));
    { Prof _p(ctx, "k_ld_fill_queue"); hipLaunchKernelGGL((k_ld_fill_queue<1>), g64, dim3(64), 0, ctx->stream, d_jobs, capacity, n_rows); }
    ZKW_TRY(launch_check("k_ld_fill_queue<1>"));
#define LD_LAUNCH_ROW(R) { Prof _p(ctx, "k_ld_fill_row"); hipLaunchKernelGGL((k_ld_fill_row<LD_ROW_##R>), g256, dim3(256), 0, ctx->stream, d_jobs, capacity, n_rows); } \
    ZKW_TRY(launch_check("k_ld_fill_row<" #R ">"));
    LD_LAUNCH_ROW(X0) LD_LAUNCH_ROW(X1) LD_LAUNCH_ROW(X2) LD_LAUNCH_ROW(X3) LD_LAUNCH_ROW(R) LD_LAUNCH_ROW(Q)
#undef LD_LAUNCH_ROW
    { Prof _p(ctx, "k_ld_fill_tail"); hipLaunchKernelGGL(k_ld_fill_tail, dim3((LD_G + LD_L + 1) * TAIL_CHUNKS, nj), dim3(256), 0, ctx->stream, d_jobs, capacity, n_rows); }
    ZKW_TRY(launch_check("k_ld_fill_tail"));
    { Prof _p(ctx, "k_ld_fill_boundary"); hipLaunchKernelGGL(k_ld_fill_boundary, dim3(nj), dim3(64), 0, ctx->stream, d_jobs, capacity, n_rows); }
    ZKW_TRY(launch_check("k_ld_fill_boundary"));
    return ctx->sync_if_host();
}

extern "C" int zkw_log_demux_check_satisfied(zkw_ctx* ctx, const zkw_trace* t, size_t slot, uint32_t capacity,
                                             uint64_t* n_violations, uint64_t* first_bad) {
    if (!ctx || !t || t->ctx->device != ctx->device || slot >= t->n_slots || !n_violations)
        return fail(ZKW_ERR_INVALID, "zkw_log_demux_check_satisfied: bad argument");
    if (LD_MIN_ROWS(capacity) > t->n_rows) return fail(ZKW_ERR_INVALID, "capacity does not fit the trace");
    return check_satisfied<SpecLogDemux>(ctx, t, slot, capacity, n_violations, first_bad);
}

// compact closed-form inputs and public inputs of a precompile witness's instances, computed once and kept with the witness
extern "C" int zkw_precompile_closed_forms(zkw_ctx* ctx, zkw_precompile_witness* w, const uint64_t** compact, const uint64_t** public_inputs) {
    if (!ctx || !w || w->ctx != ctx) return fail(ZKW_ERR_INVALID, "zkw_precompile_closed_forms: bad argument");
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
struct NlCached { NlDev host; NlDev* dev = nullptr; };
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
    for (int left = 64 - (int)hs->n_tables; left > 0; left--) {  // the next slice goes to the table with the most lookups per slice
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
    // 16 waves per workgroup where two such workgroups still fit a CU (the Keccak family: 32 cycles in flight per CU), else 8
    d.lds_bytes = NlLds(*hs, V.size, d.n_pk_terms, 8).total;
    d.lds_bytes16 = NlLds(*hs, V.size, d.n_pk_terms, 16).total;
    d.fill_waves = d.lds_bytes16 <= 80 * 1024 ? 16 : 8;
    if (getenv("ZKW_NL_VERBOSE")) fprintf(stderr, "[zkw] netlist circuit %d: %u terms packed into %u, LDS %u bytes for %u waves\n", circuit_type, hs->n_terms, d.n_pk_terms, d.fill_waves == 16 ? d.lds_bytes16 : d.lds_bytes, d.fill_waves);
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
    static bool attr_set[16] = {};
    if (!attr_set[ctx->device & 15]) {  // more than the default 64 KB of dynamic LDS
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_nl_fill<W, R, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set[ctx->device & 15] = true;
    }
    static const u32 probe = [] { const char* e = getenv("ZKW_NL_PROBE"); return e ? (u32)atoi(e) : 0u; }();  // measurement only: 1 = no level walk, 2 = no streaming
    // as many workgroups as the LDS lets a CU hold: the write phase is a stream of stores and wants waves in flight
    const unsigned lds = WAVES == 16 ? nc->host.lds_bytes16 : nc->host.lds_bytes;
    const unsigned per_cu = std::max<unsigned>(1, std::min<unsigned>(4, (160u * 1024u) / std::max<unsigned>(1, lds)));
    const unsigned blocks = std::min<unsigned>((capacity + WAVES - 1) / WAVES, std::max<unsigned>(1, 256 * per_cu / nj));
    { Prof _p(ctx, "k_nl_fill"); hipLaunchKernelGGL((k_nl_fill<W, R, WAVES>), dim3(blocks, nj), dim3(64 * WAVES), lds, ctx->stream, nc->dev, d_jobs, capacity, n_rows, probe); }
    return launch_check("k_nl_fill");
}
template <int W, int R>
int nl_launch_fill(zkw_ctx* ctx, const NlCached* nc, const NlJob* d_jobs, unsigned nj, u32 capacity, size_t n_rows) {
    // (a call with few cycles keeps 8 waves per workgroup: twice the workgroups, so that every CU has one)
    if (nc->host.fill_waves == 16 && (size_t)((capacity + 15) / 16) * nj >= 128) ZKW_TRY((nl_launch_fill_w<W, R, 16>(ctx, nc, d_jobs, nj, capacity, n_rows)));
    else ZKW_TRY((nl_launch_fill_w<W, R, 8>(ctx, nc, d_jobs, nj, capacity, n_rows)));
    { Prof _p(ctx, "k_nl_hist"); hipLaunchKernelGGL((k_nl_hist<R>), dim3(nc->host.n_hist_slices, nc->host.s.total_table_rows > NL_HIST_HALF ? 2 : 1, nj), dim3(NL_HIST_THREADS), 0, ctx->stream, nc->dev, d_jobs, capacity, n_rows); }
    return launch_check("k_nl_hist");
}

// synthesis of instances of one netlist circuit from the block's round records (`sha_like`: zkw_sha256_round_record, else keccak)
int nl_synthesize(zkw_ctx* ctx, int circuit_type, bool sha_like, const void* d_rounds, const std::vector<NlInstance>& inst, u32 capacity, size_t n_rows) {
    const NlCached* nc = nullptr;
    ZKW_TRY(nl_get(ctx, circuit_type, &nc));
    const nl_spec& S = nc->host.s;
    if (nc->host.lds_bytes > 160 * 1024) return fail(ZKW_ERR_INVALID, "netlist of circuit %d needs %u bytes of LDS", circuit_type, nc->host.lds_bytes);
    const size_t used = NL_USED_ROWS(&S, capacity);
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
    const size_t bnd = NL_BOUNDARY_ROW(&S, capacity);
    for (size_t k = 0; k < ni; k++) {
        const size_t rec_bytes = sha_like ? sizeof(zkw_sha256_round_record) : sizeof(zkw_keccak_round_record);
        prep[k] = inst[k].fresh ? NlPrepJob{static_cast<const char*>(d_rounds) + inst[k].first_round * rec_bytes, 0, inst[k].n_active, d_hdr + k * hdr_n, d_free + k * free_n, d_state + k * state_n}
                                : NlPrepJob{d_rounds, inst[k].first_round, inst[k].n_active, d_hdr + k * hdr_n, d_free + k * free_n, d_state + k * state_n};
        // The fill writes the lookup cells of every row above the boundary and the general-purpose cells of the header / gate rows;
        // everything else is zero. A slot whose previous tenant was the same layout (circuit, capacity, rows) already has those
        // zeros: nothing to clear (the multiplicity column is rewritten over the tables' rows). Otherwise: clear it.
        const uint64_t tag = ((uint64_t)circuit_type << 56) ^ ((uint64_t)capacity << 24) ^ (uint64_t)n_rows ^ 0x5A00000000000000ull;
        const bool clean = inst[k].t->tag_of(inst[k].slot) == tag;
        u64* tr = inst[k].t->slot_for_write(inst[k].slot, tag);
        jobs[k] = NlJob{prep[k].hdr_bits, prep[k].free_elems, prep[k].state_before, inst[k].public_input, tr, d_keys + k * keys_n, d_hist + k * hist_n};
        if (!clean) {
            HIP_TRY(hipMemsetAsync(tr, 0, (size_t)S.g * n_rows * sizeof(u64), ctx->stream));  // general-purpose columns
            hipLaunchKernelGGL(k_zero_strip, dim3((unsigned)((n_rows - bnd + 255) / 256), S.mult_col - S.g), dim3(256), 0, ctx->stream, tr + (size_t)S.g * n_rows + bnd, n_rows, n_rows - bnd);
            ZKW_TRY(launch_check("k_zero_strip"));
            HIP_TRY(hipMemsetAsync(tr + (size_t)S.mult_col * n_rows, 0, n_rows * sizeof(u64), ctx->stream));
        }
    }
    NlPrepJob* d_prep = nullptr;
    NlJob* d_jobs = nullptr;
    ZKW_TRY(ctx->upload("nl_prep", prep, &d_prep));
    ZKW_TRY(ctx->upload("nl_jobs", jobs, &d_jobs));
    const unsigned nj = (unsigned)ni;
    if (sha_like) { Prof _p(ctx, "k_nl_prepare"); hipLaunchKernelGGL(k_nl_prepare_sha, dim3(capacity + 1, nj), dim3(128), 0, ctx->stream, d_prep, capacity); }
    else { Prof _p(ctx, "k_nl_prepare"); hipLaunchKernelGGL(k_nl_prepare_keccak, dim3(capacity + 1, nj), dim3(256), 0, ctx->stream, d_prep, capacity); }
    ZKW_TRY(launch_check("k_nl_prepare"));
    switch (circuit_type) {
        case 6: ZKW_TRY((nl_launch_fill<SC_W, SC_R>(ctx, nc, d_jobs, nj, capacity, n_rows))); break;
        case 3: ZKW_TRY((nl_launch_fill<DC_W, DC_R>(ctx, nc, d_jobs, nj, capacity, n_rows))); break;
        case 5: ZKW_TRY((nl_launch_fill<KC_W, KC_R>(ctx, nc, d_jobs, nj, capacity, n_rows))); break;
        default: ZKW_TRY((nl_launch_fill<LH_W, LH_R>(ctx, nc, d_jobs, nj, capacity, n_rows))); break;
    }
    { Prof _p(ctx, "k_nl_finish"); hipLaunchKernelGGL(k_nl_finish, dim3((std::max(S.state, S.total_table_rows) + 255) / 256, nj), dim3(256), 0, ctx->stream, nc->dev, d_jobs, capacity, n_rows); }
    return launch_check("k_nl_finish");
}

int nl_check(zkw_ctx* ctx, int circuit_type, const zkw_trace* t, size_t slot, u32 capacity, uint64_t* n_violations, uint64_t* first_bad) {
    const NlCached* nc = nullptr;
    ZKW_TRY(nl_get(ctx, circuit_type, &nc));
    const nl_spec& S = nc->host.s;
    if (t->n_cols < S.cols) return fail(ZKW_ERR_INVALID, "trace has %zu columns, the circuit needs %u", t->n_cols, S.cols);
    if (NL_USED_ROWS(&S, capacity) > t->n_rows) return fail(ZKW_ERR_INVALID, "capacity does not fit the trace");
    HIP_TRY(hipSetDevice(ctx->device));
    const u64* trace = t->data + slot * t->slot_elems();
    const size_t n_rows = t->n_rows;
    CheckResult* d_res = nullptr;
    u32* d_hist = nullptr;
    ZKW_TRY(ctx->scratch_t<CheckResult>("check_res", 1, &d_res));
    ZKW_TRY(ctx->scratch_t<u32>("nl_check_hist", S.total_table_rows, &d_hist));
    CheckResult init{0ull, ~0ull};
    HIP_TRY(hipMemcpyAsync(d_res, &init, sizeof init, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemsetAsync(d_hist, 0, S.total_table_rows * sizeof(u32), ctx->stream));
    { Prof _p(ctx, "k_nl_check_steps"); hipLaunchKernelGGL(k_nl_check_steps, dim3((nc->host.max_items + 255) / 256, capacity * S.steps_per_cycle), dim3(256), 0, ctx->stream, nc->dev, trace, capacity, n_rows, d_hist, d_res); }
    ZKW_TRY(launch_check("k_nl_check_steps"));
    { Prof _p(ctx, "k_nl_check_tail"); hipLaunchKernelGGL(k_nl_check_tail, dim3(1024), dim3(256), 0, ctx->stream, nc->dev, trace, capacity, n_rows, d_hist, d_res); }
    ZKW_TRY(launch_check("k_nl_check_tail"));
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
    return nl_synthesize(ctx, 5, false, w->keccak_rounds, nl_instances(first_instance, n_instances, w->capacity, w->n_requests != 0, w->total_rounds, w->cf_pi, w->n_instances, t, first_slot),
                         w->capacity, t->n_rows);
}
extern "C" int zkw_keccak_round_check_satisfied(zkw_ctx* ctx, const zkw_trace* t, size_t slot, uint32_t capacity, uint64_t* n_violations, uint64_t* first_bad) {
    if (!ctx || !t || t->ctx->device != ctx->device || slot >= t->n_slots || !n_violations || capacity == 0)
        return fail(ZKW_ERR_INVALID, "zkw_keccak_round_check_satisfied: bad argument");
    return nl_check(ctx, 5, t, slot, capacity, n_violations, first_bad);
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
    return nl_synthesize(ctx, 6, true, w->sha256_rounds, nl_instances(first_instance, n_instances, w->capacity, w->n_requests != 0, w->total_rounds, w->cf_pi, w->n_instances, t, first_slot),
                         w->capacity, t->n_rows);
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
    return nl_synthesize(ctx, 3, true, w->sha256_rounds, nl_instances(first_instance, n_instances, w->capacity, true, w->total_rounds, w->cf_pi, w->n_instances, t, first_slot),
                         w->capacity, t->n_rows);
}
extern "C" int zkw_code_decommitter_check_satisfied(zkw_ctx* ctx, const zkw_trace* t, size_t slot, uint32_t capacity, uint64_t* n_violations, uint64_t* first_bad) {
    if (!ctx || !t || t->ctx->device != ctx->device || slot >= t->n_slots || !n_violations || capacity == 0)
        return fail(ZKW_ERR_INVALID, "zkw_code_decommitter_check_satisfied: bad argument");
    return nl_check(ctx, 3, t, slot, capacity, n_violations, first_bad);
}

// copy-permutation check through sigma columns (zkw_setup_copy_permutation): trace[cell] == trace[sigma[cell]] for every cell
__global__ __launch_bounds__(256) void k_check_sigma(const u64* __restrict__ trace, const u64* __restrict__ sigma, size_t n_cells, CheckResult* res,
                                                     size_t n_rows) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_cells; i += stride) {
        const u64 j = sigma[i];
        if (j >= n_cells || trace[i] != trace[j]) flag_bad(res, 4, i / n_rows, i % n_rows);
    }
}
extern "C" int zkw_check_copy_permutation(zkw_ctx* ctx, const zkw_trace* t, size_t slot, const uint64_t* sigma, uint32_t n_columns,
                                          uint64_t* n_violations, uint64_t* first_bad) {
    if (!ctx || !t || !sigma || !n_violations || t->ctx->device != ctx->device || slot >= t->n_slots || n_columns == 0 || n_columns > t->n_cols)
        return fail(ZKW_ERR_INVALID, "zkw_check_copy_permutation: bad argument");
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t n_cells = (size_t)n_columns * t->n_rows;
    const u64* d_sigma = nullptr;
    ZKW_TRY(ctx->in("sigma", sigma, n_cells, &d_sigma));
    CheckResult* d_res = nullptr;
    ZKW_TRY(ctx->scratch_t<CheckResult>("check_res", 1, &d_res));
    CheckResult init{0ull, ~0ull};
    HIP_TRY(hipMemcpyAsync(d_res, &init, sizeof init, hipMemcpyHostToDevice, ctx->stream));
    { Prof _p(ctx, "k_check_sigma"); hipLaunchKernelGGL(k_check_sigma, dim3(2048), dim3(256), 0, ctx->stream, t->data + slot * t->slot_elems(), d_sigma, n_cells, d_res, t->n_rows); }
    ZKW_TRY(launch_check("k_check_sigma"));
    CheckResult res;
    ZKW_TRY(ctx->read_small(&res, d_res, sizeof res));
    *n_violations = res.violations;
    if (first_bad) *first_bad = res.violations ? res.first_bad : 0;
    return ZKW_OK;
}

// LinearHasher (type 13): the Keccak-f netlist over the sponge of the serialized L2 -> L1 messages (compute_linear_keccak256,
// data_hasher_and_merklizer.rs:8-67; wrapper base_layer/linear_hasher.rs:28-138). One instance per block.
extern "C" int zkw_linear_hasher_synthesize_batch(zkw_ctx* ctx, const zkw_log_query* messages, const uint64_t* message_offsets, size_t n_queues,
                                                  const zkw_queue_state4* queue_states, uint32_t capacity, zkw_trace* t, size_t first_slot,
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
    { Prof _p(ctx, "k_linear_keccak256"); hipLaunchKernelGGL(k_linear_keccak256, dim3((unsigned)n_queues), dim3(64), 0, ctx->stream, d_q, (size_t)0, d_hash, d_rounds, d_moff, d_roff); }
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
        constexpr int lanes = CfLanes<CfLinearHasher>::value;
        Prof _p(ctx, "k_closed_form_commitments");
        hipLaunchKernelGGL((k_closed_form_commitments<CfLinearHasher>), dim3(blocks_for(4 * n_queues, lanes)), dim3(lanes), 0, ctx->stream, d_rec, n_queues, d_cf);
    }
    ZKW_TRY(launch_check("k_closed_form_commitments"));
    { Prof _p(ctx, "k_commit_encodings"); hipLaunchKernelGGL(k_commit_encodings, dim3(blocks_for(n_queues, 64)), dim3(64), 0, ctx->stream, d_cf, n_queues, (u32)COMPACT_FORM_LEN, d_pi); }
    ZKW_TRY(launch_check("k_commit_encodings"));
    std::vector<NlInstance> inst(n_queues);
    for (size_t b = 0; b < n_queues; b++) inst[b] = NlInstance{roff[b], (u32)(roff[b + 1] - roff[b]), d_pi + 4 * b, t, first_slot + b, true};
    ZKW_TRY(nl_synthesize(ctx, 13, false, d_rounds, inst, cycles, n_rows));
    memcpy(records_out, recv.data(), n_queues * sizeof recv[0]);
    if (public_inputs_out) ZKW_TRY(ctx->read_small(public_inputs_out, d_pi, 32 * n_queues));
    return ZKW_OK;
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
    HIP_TRY(hipMemsetAsync(d_hist, 0, n_instances * 256 * sizeof(u32), ctx->stream));
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
        j.trace = t->slot_for_write((first_slot + k) % t->n_slots, 0);
        j.hist = d_hist + 256 * k;
    }
    SsSynthJob* d_jobs = nullptr;
    ZKW_TRY(ctx->upload("ss_jobs", jobs, &d_jobs));
    const unsigned nj = (unsigned)n_instances;
    const u32 rstride = (u32)SS_REGION_STRIDE(capacity);
    const dim3 g64((rstride + 63) / 64, nj), g256((rstride + 255) / 256, nj);
    { Prof _p(ctx, "k_ss_fill_queue"); hipLaunchKernelGGL((k_ss_fill_queue<0>), g64, dim3(64), 0, ctx->stream, d_jobs, capacity, n_rows); }
    ZKW_TRY(launch_check("k_ss_fill_queue<0>"));
    { Prof _p(ctx, "k_ss_fill_queue"); hipLaunchKernelGGL((k_ss_fill_queue<1>), g64, dim3(64), 0, ctx->stream, d_jobs, capacity, n_rows); }
    ZKW_TRY(launch_check("k_ss_fill_queue<1>"));
    { Prof _p(ctx, "k_ss_fill_queue"); hipLaunchKernelGGL((k_ss_fill_queue<2>), g64, dim3(64), 0, ctx->stream, d_jobs, capacity, n_rows); }
    ZKW_TRY(launch_check("k_ss_fill_queue<2>"));
#define SS_LAUNCH_ROW(R) { Prof _p(ctx, "k_ss_fill_row"); hipLaunchKernelGGL((k_ss_fill_row<SS_ROW_##R>), g256, dim3(256), 0, ctx->stream, d_jobs, capacity, n_rows); } \
    ZKW_TRY(launch_check("k_ss_fill_row<" #R ">"));
    SS_LAUNCH_ROW(A) SS_LAUNCH_ROW(X0) SS_LAUNCH_ROW(X1) SS_LAUNCH_ROW(X2) SS_LAUNCH_ROW(X3) SS_LAUNCH_ROW(X4) SS_LAUNCH_ROW(X5)
    SS_LAUNCH_ROW(X6) SS_LAUNCH_ROW(X7) SS_LAUNCH_ROW(K) SS_LAUNCH_ROW(C1) SS_LAUNCH_ROW(C2) SS_LAUNCH_ROW(Q)
#undef SS_LAUNCH_ROW
    { Prof _p(ctx, "k_ss_fill_tail"); hipLaunchKernelGGL(k_ss_fill_tail, dim3((SS_G + SS_L + 1) * TAIL_CHUNKS, nj), dim3(256), 0, ctx->stream, d_jobs, capacity, n_rows); }
    ZKW_TRY(launch_check("k_ss_fill_tail"));
    { Prof _p(ctx, "k_ss_fill_boundary"); hipLaunchKernelGGL(k_ss_fill_boundary, dim3(nj), dim3(64), 0, ctx->stream, d_jobs, capacity, n_rows); }
    ZKW_TRY(launch_check("k_ss_fill_boundary"));
    return ctx->sync_if_host();
}

extern "C" int zkw_storage_sorter_check_satisfied(zkw_ctx* ctx, const zkw_trace* t, size_t slot, uint32_t capacity,
                                                  uint64_t* n_violations, uint64_t* first_bad) {
    if (!ctx || !t || t->ctx->device != ctx->device || slot >= t->n_slots || !n_violations)
        return fail(ZKW_ERR_INVALID, "zkw_storage_sorter_check_satisfied: bad argument");
    if (SS_MIN_ROWS(capacity) > t->n_rows) return fail(ZKW_ERR_INVALID, "capacity does not fit the trace");
    return check_satisfied<SpecStorageSorter>(ctx, t, slot, capacity, n_violations, first_bad);
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
    { Prof _p(ctx, "k_vm_rw_tile_counts"); hipLaunchKernelGGL(k_vm_rw_tile_counts, dim3(n_tiles), dim3(256), 0, ctx->stream, job.s.vm_memory_queries, (u64)n_mem, d_tiles); }
    ZKW_TRY(launch_check("k_vm_rw_tile_counts"));
    { Prof _p(ctx, "k_vm_rw_scan_tiles"); hipLaunchKernelGGL(k_vm_rw_scan_tiles, dim3(1), dim3(1024), 0, ctx->stream, d_tiles, n_tiles, d_total); }
    ZKW_TRY(launch_check("k_vm_rw_scan_tiles"));
    { Prof _p(ctx, "k_vm_rw_scatter"); hipLaunchKernelGGL(k_vm_rw_scatter, dim3(n_tiles), dim3(256), 0, ctx->stream, job.s.vm_memory_queries, (u64)n_mem, d_tiles, d_prefix, d_ri, d_wi); }
    ZKW_TRY(launch_check("k_vm_rw_scatter"));
    job.read_prefix = d_prefix;
    ZKW_TRY(ctx->out("vm_inst", instances, n_inst, &job.out));
    { Prof _p(ctx, "k_vm_slice"); hipLaunchKernelGGL(k_vm_slice, dim3(blocks_for(n_inst, 64)), dim3(64), 0, ctx->stream, job); }
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

