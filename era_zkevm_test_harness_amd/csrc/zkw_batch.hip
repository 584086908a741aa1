// zkw_batch.hip — fibers + merged launches (zkw_batch.h). Host code; its only kernels are the byte fill / copy the merged memsets and
// small copies travel in.
#include "zkw_ctx.h"

#include <sys/mman.h>
#include <ucontext.h>
#include <unistd.h>

#include <deque>

namespace {

using Clock = std::chrono::steady_clock;

// ---- the two kernels of the batch itself -------------------------------------------------------------------------------------------
static __device__ __forceinline__ void k_batch_fill(const VB& vb, void* dst, int value, size_t bytes) {
    const size_t i = (size_t)vb.x * blockDim.x + threadIdx.x, stride = (size_t)vb.nx * blockDim.x;
    unsigned char* d = static_cast<unsigned char*>(dst);
    const unsigned char v = (unsigned char)value;
    if ((reinterpret_cast<uintptr_t>(d) & 15) == 0 && (bytes & 15) == 0) {
        const unsigned w = 0x01010101u * v;
        const uint4 q = make_uint4(w, w, w, w);
        uint4* d4 = reinterpret_cast<uint4*>(d);
        for (size_t k = i; k < bytes / 16; k += stride) d4[k] = q;
    } else {
        for (size_t k = i; k < bytes; k += stride) d[k] = v;
    }
}
static __device__ __forceinline__ void k_batch_copy(const VB& vb, void* dst, const void* src, size_t bytes) {
    const size_t i = (size_t)vb.x * blockDim.x + threadIdx.x, stride = (size_t)vb.nx * blockDim.x;
    unsigned char* d = static_cast<unsigned char*>(dst);
    const unsigned char* s = static_cast<const unsigned char*>(src);
    if (((reinterpret_cast<uintptr_t>(d) | reinterpret_cast<uintptr_t>(s)) & 15) == 0) {
        const size_t n16 = bytes / 16;
        const uint4* s4 = reinterpret_cast<const uint4*>(s);
        uint4* d4 = reinterpret_cast<uint4*>(d);
        for (size_t k = i; k < n16; k += stride) d4[k] = s4[k];
        for (size_t k = n16 * 16 + i; k < bytes; k += stride) d[k] = s[k];
    } else if (((reinterpret_cast<uintptr_t>(d) | reinterpret_cast<uintptr_t>(s)) & 3) == 0) {
        const size_t n4 = bytes / 4;
        const unsigned* s1 = reinterpret_cast<const unsigned*>(s);
        unsigned* d1 = reinterpret_cast<unsigned*>(d);
        for (size_t k = i; k < n4; k += stride) d1[k] = s1[k];
        for (size_t k = n4 * 4 + i; k < bytes; k += stride) d[k] = s[k];
    } else {
        for (size_t k = i; k < bytes; k += stride) d[k] = s[k];
    }
}
static unsigned byte_grid(size_t bytes) {
    const size_t per_wg = 256 * 16 * 4;  // four 16-byte words per lane
    size_t g = (bytes + per_wg - 1) / per_wg;
    return (unsigned)std::min<size_t>(std::max<size_t>(g, 1), 4096);
}

// ---- a mirrored bump arena: pinned host chunk + device chunk at equal offsets --------------------------------------------------------
struct Arena {
    struct Chunk {
        char *host = nullptr, *dev = nullptr;
        size_t cap = 0, used = 0, moved = 0;  // [moved, used) has not crossed the bus yet
    };
    std::vector<Chunk> chunks;
    size_t cur = 0;  // the chunk allocations come from; the ones before it are full, the ones after it empty (kept from earlier runs)
    int alloc(size_t bytes, size_t align, char** host, char** dev) {
        if (align < 16) align = 16;
        for (; cur < chunks.size(); cur++) {
            Chunk& c = chunks[cur];
            const size_t at = (c.used + align - 1) & ~(align - 1);
            if (at + bytes <= c.cap) {
                *host = c.host + at;
                *dev = c.dev + at;
                c.used = at + bytes;
                return ZKW_OK;
            }
        }
        Chunk c;
        c.cap = std::max<size_t>(bytes + align, (size_t)8 << 20);
        void *h = nullptr, *d = nullptr;
        if (pin_malloc(&h, c.cap) != hipSuccess) return fail(ZKW_ERR_OOM, "zkw_batch: hipHostMalloc of %zu bytes failed", c.cap);
        if (dev_malloc(&d, c.cap) != hipSuccess) {
            pin_free(h);
            return fail(ZKW_ERR_OOM, "zkw_batch: hipMalloc of %zu bytes failed", c.cap);
        }
        c.host = static_cast<char*>(h);
        c.dev = static_cast<char*>(d);
        chunks.push_back(c);
        cur = chunks.size() - 1;
        return alloc(bytes, align, host, dev);
    }
    void reset() {  // between runs of a batch: nothing in flight refers to the arena any more
        for (Chunk& c : chunks) c.used = c.moved = 0;
        cur = 0;
    }
    void release() {
        for (Chunk& c : chunks) {
            pin_free(c.host);
            dev_free(c.dev);
        }
        chunks.clear();
    }
};

struct Op {
    enum Kind : uint8_t { LAUNCH, CHAIN } kind;
    const BatchKernel* k;
    unsigned gx, gy, gz, lds;
    size_t blob;  // LAUNCH: offset of the packed arguments in the fiber's blob; CHAIN: index into the fiber's chain list
};
struct ChainReq {
    std::vector<ChainJob> full;
    std::vector<LogChainJob> log;
};
struct Landing {
    void* dst;
    const char* src;
    size_t bytes;
};

struct Fiber {
    ucontext_t uc;
    void* stack = nullptr;
    size_t stack_bytes = 0;
    std::function<int()> fn;
    int rc = ZKW_OK;
    std::string err;
    enum State { READY, WAIT_GPU, WAIT_JOIN, DONE } state = READY;
    std::vector<Op> ops;
    std::vector<char> blob;
    std::deque<ChainReq> chains;
    std::vector<Landing> landings;
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};  // what a parked fiber waits for: the flush's end on the main stream, its chains
    int n_ev = 0;
    bool wants_gpu = false;  // parked in sync() / chains(): wake when its queued work has run
    int wait_rc = ZKW_OK;
    int join_on = -1;
};

}  // namespace

struct zkw_batch {
    int device = 0;
    hipStream_t main = nullptr;
    std::vector<hipStream_t> chain_streams;
    size_t next_chain_stream = 0;
    std::vector<hipEvent_t> free_events, used_events;
    Arena up, down;
    std::vector<std::unique_ptr<Fiber>> fibers;
    Fiber* cur = nullptr;
    ucontext_t sched;
    int flush_rc = ZKW_OK;
    std::string flush_err;
    // statistics (ZKW_BATCH_LOG=1)
    size_t n_flushes = 0, n_launches = 0, n_jobs = 0, n_chain_launches = 0, n_switches = 0;
    double flush_host_ms = 0, wait_ms = 0;

    hipEvent_t event() {
        hipEvent_t e = nullptr;
        if (!free_events.empty()) {
            e = free_events.back();
            free_events.pop_back();
        } else if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
            return nullptr;
        }
        used_events.push_back(e);
        return e;
    }
    void fail_flush(hipError_t e, const char* what) {
        if (e != hipSuccess && flush_rc == ZKW_OK) {
            flush_rc = e == hipErrorOutOfMemory ? ZKW_ERR_OOM : ZKW_ERR_HIP;
            flush_err = std::string("zkw_batch: ") + what + ": " + hipGetErrorString(e);
        }
    }
    void park() {  // from a fiber: back to the scheduler
        n_switches++;
        Fiber* f = cur;
        swapcontext(&f->uc, &sched);
    }
    bool flush();
    int run(const std::vector<std::function<int()>>& roots);
};

namespace {

thread_local zkw_batch* tl_batch = nullptr;

// per device: streams are expensive to create and hipStreamDestroy waits for the whole device, so a batch borrows them
struct BatchStreams {
    std::mutex mu;
    std::vector<hipStream_t> idle_main, idle_chain;
};
BatchStreams& batch_streams(int device) {
    static std::mutex mu;
    static std::map<int, BatchStreams*>& m = *new std::map<int, BatchStreams*>();
    std::lock_guard<std::mutex> g(mu);
    BatchStreams*& p = m[device];
    if (!p) p = new BatchStreams();
    return *p;
}

void fiber_entry(unsigned lo, unsigned hi) {
    Fiber* f = reinterpret_cast<Fiber*>(((uintptr_t)hi << 32) | (uintptr_t)lo);
    f->rc = f->fn();
    if (f->rc != ZKW_OK) f->err = zkw_last_error();
    f->state = Fiber::DONE;
    // uc_link returns to the scheduler
}

int make_fiber(zkw_batch* b, std::function<int()> fn) {
    std::unique_ptr<Fiber> f(new Fiber());
    f->stack_bytes = (size_t)512 << 10;
    f->stack = mmap(nullptr, f->stack_bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (f->stack == MAP_FAILED) return -1;
    f->fn = std::move(fn);
    getcontext(&f->uc);
    f->uc.uc_stack.ss_sp = f->stack;
    f->uc.uc_stack.ss_size = f->stack_bytes;
    f->uc.uc_link = &b->sched;
    const uintptr_t p = reinterpret_cast<uintptr_t>(f.get());
    makecontext(&f->uc, reinterpret_cast<void (*)()>(fiber_entry), 2, (unsigned)(p & 0xffffffffu), (unsigned)(p >> 32));
    b->fibers.push_back(std::move(f));
    return (int)b->fibers.size() - 1;
}

}  // namespace

static bool ensure_chain_streams(zkw_batch* b);
// One flush: every pending launch of every fiber, position by position. Returns whether anything was sent.
bool zkw_batch::flush() {
    const auto t0 = Clock::now();
    size_t rounds = 0;
    for (auto& f : fibers) rounds = std::max(rounds, f->ops.size());
    bool any_wants = false;
    for (auto& f : fibers) any_wants = any_wants || (f->state == Fiber::WAIT_GPU && f->wants_gpu && f->n_ev == 0);
    if (rounds == 0 && !any_wants) return false;
    n_flushes++;
    struct Merged {
        const BatchKernel* k;
        unsigned lds = 0;
        std::vector<std::pair<Fiber*, const Op*>> jobs;
        char* d_table = nullptr;
        unsigned *d_prefix = nullptr, *d_gxy = nullptr;
        unsigned total = 0;
    };
    struct ChainGroup {
        std::vector<ChainJob> full;
        std::vector<LogChainJob> log;
        std::vector<Fiber*> owners;
        ChainJob* d_full = nullptr;
        LogChainJob* d_log = nullptr;
    };
    struct Round {
        std::vector<Merged> merged;
        ChainGroup chain;
    };
    std::vector<Round> plan(rounds);
    for (size_t r = 0; r < rounds; r++) {
        Round& R = plan[r];
        for (auto& fp : fibers) {
            Fiber* f = fp.get();
            if (r >= f->ops.size()) continue;
            const Op& op = f->ops[r];
            if (op.kind == Op::CHAIN) {
                ChainReq& c = f->chains[op.blob];
                R.chain.full.insert(R.chain.full.end(), c.full.begin(), c.full.end());
                R.chain.log.insert(R.chain.log.end(), c.log.begin(), c.log.end());
                R.chain.owners.push_back(f);
                continue;
            }
            Merged* m = nullptr;
            for (Merged& x : R.merged)
                if (x.k == op.k && x.lds == op.lds) { m = &x; break; }
            if (!m) {
                R.merged.push_back(Merged());
                m = &R.merged.back();
                m->k = op.k;
                m->lds = op.lds;
            }
            m->jobs.emplace_back(f, &op);
        }
        // the job tables of this round, in the upload arena
        for (Merged& m : R.merged) {
            const size_t n = m.jobs.size(), stride = m.k->tup_bytes;
            char *h_table = nullptr, *d_table = nullptr, *h_pre = nullptr, *d_pre = nullptr;
            if (up.alloc(n * stride, m.k->tup_align, &h_table, &d_table) != ZKW_OK || up.alloc((3 * n + 1) * sizeof(unsigned), 16, &h_pre, &d_pre) != ZKW_OK) {
                if (flush_rc == ZKW_OK) { flush_rc = ZKW_ERR_OOM; flush_err = zkw_last_error(); }
                continue;
            }
            unsigned* pre = reinterpret_cast<unsigned*>(h_pre);
            unsigned* gxy = pre + n + 1;
            unsigned long long acc = 0;
            for (size_t j = 0; j < n; j++) {
                const Op* op = m.jobs[j].second;
                memcpy(h_table + j * stride, m.jobs[j].first->blob.data() + op->blob, stride);
                pre[j] = (unsigned)acc;
                gxy[2 * j] = op->gx;
                gxy[2 * j + 1] = op->gy;
                acc += (unsigned long long)op->gx * op->gy * op->gz;
            }
            pre[n] = (unsigned)acc;
            if (acc >= (1ull << 31) && flush_rc == ZKW_OK) { flush_rc = ZKW_ERR_INVALID; flush_err = std::string("zkw_batch: merged grid of ") + m.k->name + " is too large"; }
            m.d_table = d_table;
            m.d_prefix = reinterpret_cast<unsigned*>(d_pre);
            m.d_gxy = m.d_prefix + n + 1;
            m.total = (unsigned)acc;
        }
        ChainGroup& c = R.chain;
        // Longest chains first: a workgroup of the row form holds 16 chains and lives as long as its longest one. In the order the blocks
        // submitted them — a demuxer's 58 750-item input queue next to its six output queues of 35 000 ... 0 items — every workgroup of a
        // stage holds a long chain and the stage keeps all its CUs for its whole length (512 blocks' demuxers: 224 CUs for 0.4 s, and the
        // memory-queue chains that arrive 0.1 s later wait for them). Sorted, the long chains share workgroups and the others leave early.
        std::stable_sort(c.full.begin(), c.full.end(), [](const ChainJob& a, const ChainJob& b) { return a.n > b.n; });
        std::stable_sort(c.log.begin(), c.log.end(), [](const LogChainJob& a, const LogChainJob& b) { return a.n > b.n; });
        if (!c.full.empty()) {
            char *h = nullptr, *d = nullptr;
            if (up.alloc(c.full.size() * sizeof(ChainJob), alignof(ChainJob), &h, &d) == ZKW_OK) {
                memcpy(h, c.full.data(), c.full.size() * sizeof(ChainJob));
                c.d_full = reinterpret_cast<ChainJob*>(d);
            } else if (flush_rc == ZKW_OK) { flush_rc = ZKW_ERR_OOM; flush_err = zkw_last_error(); }
        }
        if (!c.log.empty()) {
            char *h = nullptr, *d = nullptr;
            if (up.alloc(c.log.size() * sizeof(LogChainJob), alignof(LogChainJob), &h, &d) == ZKW_OK) {
                memcpy(h, c.log.data(), c.log.size() * sizeof(LogChainJob));
                c.d_log = reinterpret_cast<LogChainJob*>(d);
            } else if (flush_rc == ZKW_OK) { flush_rc = ZKW_ERR_OOM; flush_err = zkw_last_error(); }
        }
    }
    // 1. everything the fibers uploaded and the job tables cross the bus
    for (Arena::Chunk& c : up.chunks)
        if (c.used > c.moved) {
            if (flush_rc == ZKW_OK) fail_flush(hipMemcpyAsync(c.dev + c.moved, c.host + c.moved, c.used - c.moved, hipMemcpyHostToDevice, main), "upload");
            c.moved = c.used;
        }
    // 2. the rounds, in order, on the one stream
    for (size_t r = 0; r < rounds && flush_rc == ZKW_OK; r++) {
        Round& R = plan[r];
        for (Merged& m : R.merged) {
            if (m.total == 0) continue;
            int n = (int)m.jobs.size();
            void* args[] = {&m.d_table, &m.d_prefix, &m.d_gxy, &n};
            fail_flush(hipLaunchKernel(m.k->multi_fn, dim3(m.total), dim3(m.k->bs), args, m.lds, main), m.k->name);
            n_launches++;
            n_jobs += m.jobs.size();
        }
        ChainGroup& c = R.chain;
        if (c.owners.empty()) continue;
        hipEvent_t done_full = nullptr, done_log = nullptr;
        if (c.d_full || c.d_log) {
            hipEvent_t before = event();
            if (!before) { fail_flush(hipErrorOutOfMemory, "hipEventCreate"); break; }
            fail_flush(hipEventRecord(before, main), "hipEventRecord");
            for (int kind = 0; kind < 2 && flush_rc == ZKW_OK; kind++) {
                if (kind == 0 ? !c.d_full : !c.d_log) continue;
                if (!ensure_chain_streams(this)) { fail_flush(hipErrorOutOfMemory, "hipStreamCreateWithPriority"); break; }
                hipStream_t st = chain_streams[next_chain_stream++ % chain_streams.size()];
                fail_flush(hipStreamWaitEvent(st, before, 0), "hipStreamWaitEvent");
                int rc = kind == 0 ? zkw_launch_chain_full(st, c.d_full, (int)c.full.size()) : zkw_launch_chain_log(st, c.d_log, (int)c.log.size());
                if (rc != ZKW_OK && flush_rc == ZKW_OK) { flush_rc = rc; flush_err = zkw_last_error(); }
                fail_flush(hipGetLastError(), "chain launch");
                hipEvent_t e = event();
                if (!e) { fail_flush(hipErrorOutOfMemory, "hipEventCreate"); break; }
                fail_flush(hipEventRecord(e, st), "hipEventRecord");
                (kind == 0 ? done_full : done_log) = e;
                n_chain_launches++;
            }
        }
        for (Fiber* f : c.owners) {
            if (done_full) f->ev[f->n_ev++] = done_full;
            if (done_log) f->ev[f->n_ev++] = done_log;
        }
    }
    // 3. the read-backs gathered by this flush's copy jobs come home
    for (Arena::Chunk& c : down.chunks)
        if (c.used > c.moved) {
            if (flush_rc == ZKW_OK) fail_flush(hipMemcpyAsync(c.host + c.moved, c.dev + c.moved, c.used - c.moved, hipMemcpyDeviceToHost, main), "read-back");
            c.moved = c.used;
        }
    hipEvent_t end = event();
    if (end) fail_flush(hipEventRecord(end, main), "hipEventRecord");
    else fail_flush(hipErrorOutOfMemory, "hipEventCreate");
    for (auto& fp : fibers) {
        Fiber* f = fp.get();
        const bool had = !f->ops.empty();
        f->ops.clear();
        f->blob.clear();
        f->chains.clear();
        if ((had || (f->state == Fiber::WAIT_GPU && f->wants_gpu && f->n_ev == 0)) && f->state == Fiber::WAIT_GPU) f->ev[f->n_ev++] = end;
    }
    flush_host_ms += std::chrono::duration<double, std::milli>(Clock::now() - t0).count();
    return true;
}

int zkw_batch::run(const std::vector<std::function<int()>>& roots) {
    if (hipSetDevice(device) != hipSuccess) return fail(ZKW_ERR_HIP, "hipSetDevice failed");
    zkw_batch* outer = tl_batch;
    tl_batch = this;
    n_flushes = n_launches = n_jobs = n_chain_launches = n_switches = 0;
    flush_host_ms = wait_ms = 0;
    for (auto& r : roots)
        if (make_fiber(this, r) < 0) {
            for (auto& f : fibers)
                if (f->stack) munmap(f->stack, f->stack_bytes);
            fibers.clear();  // (none of them has run)
            tl_batch = outer;
            return fail(ZKW_ERR_OOM, "zkw_batch: no stack for a fiber");
        }
    static const int verbose = [] { const char* e = getenv("ZKW_BATCH_LOG"); return e ? atoi(e) : 0; }();  // 2: one line per flush
    const auto t_run = Clock::now();
    auto ms_since = [&](Clock::time_point t) { return std::chrono::duration<double, std::milli>(Clock::now() - t).count(); };
    auto t_sweep = Clock::now();
    int first_rc = ZKW_OK;
    std::string first_err;
    auto note = [&](int rc, const std::string& e) { if (rc != ZKW_OK && first_rc == ZKW_OK) { first_rc = rc; first_err = e; } };
    for (;;) {
        // 1. run whatever can run (fibers spawned meanwhile are at the end of the list and run in the same sweep)
        bool ran = false;
        size_t live = 0;
        // (Measured and rejected, round 6: sending what N parked fibers have left while the sweep still runs later fibers — to overlap the host's
        // part of a stage with the GPU's. The blocks fall out of step, a stage stops being one launch, and the builders of 512 blocks take 1.93 s
        // at N = 256, 2.08 s at 128, 3.8 s at 64 against 1.77 s: the flush happens when nobody can run, and only then.)
        for (size_t i = 0; i < fibers.size(); i++) {
            Fiber* f = fibers[i].get();
            if (f->state == Fiber::WAIT_JOIN && fibers[f->join_on]->state == Fiber::DONE) f->state = Fiber::READY;
            if (f->state == Fiber::READY) {
                cur = f;
                n_switches++;
                swapcontext(&sched, &f->uc);
                cur = nullptr;
                ran = true;
                if (f->state == Fiber::DONE) {
                    note(f->rc, f->err);
                    if (f->stack) { munmap(f->stack, f->stack_bytes); f->stack = nullptr; }
                }
            }
            if (f->state != Fiber::DONE) live++;
        }
        if (live == 0) break;
        if (ran) continue;  // a sweep may have made joiners ready
        // 2. nobody can run: send what they left
        const double host_ms = ms_since(t_sweep);
        const size_t l0 = n_launches, c0 = n_chain_launches;
        const auto t_fl = Clock::now();
        flush();
        const double fl_ms = ms_since(t_fl);
        // 3. wait until some parked fiber has everything it waits for
        const auto tw = Clock::now();
        bool woke = false;
        for (int spins = 0; !woke; spins++) {
            bool waiting = false;
            for (auto& fp : fibers) {
                Fiber* f = fp.get();
                if (f->state != Fiber::WAIT_GPU) continue;
                if (f->n_ev == 0 && flush_rc == ZKW_OK) continue;  // (its work has not been sent: cannot happen after a flush)
                waiting = true;
                bool done = true;
                if (flush_rc == ZKW_OK)
                    for (int k = 0; k < f->n_ev && done; k++) {
                        const hipError_t q = hipEventQuery(f->ev[k]);
                        if (q == hipErrorNotReady) done = false;
                        else if (q != hipSuccess) fail_flush(q, "hipEventQuery");
                    }
                if (!done) continue;
                for (const Landing& l : f->landings) memcpy(l.dst, l.src, l.bytes);
                f->landings.clear();
                f->n_ev = 0;
                f->wants_gpu = false;
                f->wait_rc = flush_rc;
                f->state = Fiber::READY;
                woke = true;
            }
            if (woke) break;
            if (!waiting) {  // every live fiber waits for a join that cannot come
                tl_batch = outer;
                return fail(ZKW_ERR_INVALID, "zkw_batch: the fibers wait for each other");
            }
            if (spins > 64) usleep(spins > 1024 ? 200 : 20);
        }
        wait_ms += std::chrono::duration<double, std::milli>(Clock::now() - tw).count();
        if (verbose >= 2) {
            size_t woken = 0;
            for (auto& fp : fibers) woken += fp->state == Fiber::READY;
            fprintf(stderr, "[zkw batch] t=%8.1f ms: fibers ran %.1f ms, flush %.1f ms (%zu merged launches, %zu chain launches), waited %.1f ms, %zu fibers woken\n", ms_since(t_run), host_ms, fl_ms,
                    n_launches - l0, n_chain_launches - c0, ms_since(tw), woken);
        }
        t_sweep = Clock::now();
    }
    tl_batch = outer;
    if (flush_rc != ZKW_OK && first_rc == ZKW_OK) { first_rc = flush_rc; first_err = flush_err; }
    (void)hipStreamSynchronize(main);
    for (hipStream_t s : chain_streams) (void)hipStreamSynchronize(s);
    for (hipEvent_t e : used_events) free_events.push_back(e);
    used_events.clear();
    up.reset();
    down.reset();
    flush_rc = ZKW_OK;
    flush_err.clear();
    if (getenv("ZKW_BATCH_LOG"))
        fprintf(stderr, "[zkw batch] %zu fibers, %zu flushes, %zu merged launches carrying %zu jobs, %zu chain launches, %zu switches; host %.1f ms in flushes, %.1f ms waiting\n",
                fibers.size(), n_flushes, n_launches, n_jobs, n_chain_launches, n_switches, flush_host_ms, wait_ms);
    fibers.clear();
    if (first_rc != ZKW_OK) return fail(first_rc, "%s", first_err.c_str());
    return ZKW_OK;
}

// ------------------------------------------------------------------------------------------------ the interface
zkw_batch* zkw_batch_create(int device) {
    if (hipSetDevice(device) != hipSuccess) { fail(ZKW_ERR_HIP, "hipSetDevice failed"); return nullptr; }
    zkw_batch* b = new zkw_batch();
    b->device = device;
    BatchStreams& S = batch_streams(device);
    {
        std::lock_guard<std::mutex> g(S.mu);
        if (!S.idle_main.empty()) { b->main = S.idle_main.back(); S.idle_main.pop_back(); }
    }
    if (!b->main && hipStreamCreateWithFlags(&b->main, hipStreamNonBlocking) != hipSuccess) {
        fail(ZKW_ERR_HIP, "zkw_batch: hipStreamCreate failed");
        zkw_batch_destroy(b);
        return nullptr;
    }
    return b;  // (the chain streams are borrowed when the first chain launch needs them: a batch that synthesizes has none)
}

// the batch's high-priority streams for queue chains, borrowed from the device's pool on first use
static bool ensure_chain_streams(zkw_batch* b) {
    if (!b->chain_streams.empty()) return true;
    static const int n_chain = [] { const char* e = getenv("ZKW_BATCH_CHAIN_STREAMS"); const int v = e ? atoi(e) : 0; return v > 0 && v <= 32 ? v : 8; }();
    BatchStreams& S = batch_streams(b->device);
    {
        std::lock_guard<std::mutex> g(S.mu);
        while ((int)b->chain_streams.size() < n_chain && !S.idle_chain.empty()) { b->chain_streams.push_back(S.idle_chain.back()); S.idle_chain.pop_back(); }
    }
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    while ((int)b->chain_streams.size() < n_chain) {
        hipStream_t s = nullptr;
        if (hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi) != hipSuccess) break;
        b->chain_streams.push_back(s);
    }
    return !b->chain_streams.empty();
}

void zkw_batch_destroy(zkw_batch* b) {
    if (!b) return;
    (void)hipSetDevice(b->device);
    if (b->main) (void)hipStreamSynchronize(b->main);
    for (hipStream_t s : b->chain_streams) (void)hipStreamSynchronize(s);
    b->up.release();
    b->down.release();
    for (hipEvent_t e : b->used_events) (void)hipEventDestroy(e);
    for (hipEvent_t e : b->free_events) (void)hipEventDestroy(e);
    BatchStreams& S = batch_streams(b->device);
    {
        std::lock_guard<std::mutex> g(S.mu);
        if (b->main) S.idle_main.push_back(b->main);
        for (hipStream_t s : b->chain_streams) S.idle_chain.push_back(s);
    }
    delete b;
}

int zkw_batch_run(zkw_batch* b, const std::vector<std::function<int()>>& roots) { return b->run(roots); }
hipStream_t zkw_batch_stream(zkw_batch* b) { return b->main; }
bool zkw_batch_in_fiber(const zkw_batch* b) { return b && tl_batch == b && b->cur != nullptr; }

int zkw_batch_spawn(zkw_batch* b, std::function<int()> fn) {
    // the new fiber's launches start at round 0 of the next flush: what the caller has queued and the branch may depend on goes first
    if (b->cur && !b->cur->ops.empty() && zkw_batch_sync(b) != ZKW_OK) return -1;
    return make_fiber(b, std::move(fn));
}

int zkw_batch_join(zkw_batch* b, int fiber) {
    if (fiber < 0 || fiber >= (int)b->fibers.size()) return fail(ZKW_ERR_INVALID, "zkw_batch_join: no such fiber");
    Fiber* t = b->fibers[fiber].get();
    if (t->state != Fiber::DONE) {
        b->cur->state = Fiber::WAIT_JOIN;
        b->cur->join_on = fiber;
        b->park();
    }
    if (t->rc != ZKW_OK) return fail(t->rc, "%s", t->err.c_str());
    return ZKW_OK;
}

void zkw_batch_launch(zkw_batch* b, const BatchKernel* k, dim3 grid, size_t lds_bytes, const void* tup) {
    Fiber* f = b->cur;
    const size_t at = (f->blob.size() + 15) & ~(size_t)15;
    f->blob.resize(at + k->tup_bytes);
    memcpy(f->blob.data() + at, tup, k->tup_bytes);
    f->ops.push_back(Op{Op::LAUNCH, k, grid.x, grid.y, grid.z, (unsigned)lds_bytes, at});
}

void* zkw_batch_upload(zkw_batch* b, const void* host, size_t bytes, size_t align) {
    char *h = nullptr, *d = nullptr;
    if (b->up.alloc(bytes ? bytes : 1, align, &h, &d) != ZKW_OK) return nullptr;
    if (bytes) memcpy(h, host, bytes);
    return d;
}

void zkw_batch_memset(zkw_batch* b, void* dev, int value, size_t bytes) {
    if (!bytes) return;
    using S = LaunchSig<decltype(&k_batch_fill)>;
    S::T t;
    S::pack(t, dev, value, bytes);
    zkw_batch_launch(b, S::desc<&k_batch_fill, 256>("k_batch_fill"), dim3(byte_grid(bytes)), 0, &t);
}

void zkw_batch_copy_d2d(zkw_batch* b, void* dst, const void* src, size_t bytes) {
    if (!bytes) return;
    using S = LaunchSig<decltype(&k_batch_copy)>;
    S::T t;
    S::pack(t, dst, src, bytes);
    zkw_batch_launch(b, S::desc<&k_batch_copy, 256>("k_batch_copy"), dim3(byte_grid(bytes)), 0, &t);
}

void zkw_batch_copy_h2d(zkw_batch* b, void* dst, const void* host, size_t bytes) {
    if (!bytes) return;
    void* staged = zkw_batch_upload(b, host, bytes, 16);
    if (!staged) { if (b->flush_rc == ZKW_OK) { b->flush_rc = ZKW_ERR_OOM; b->flush_err = zkw_last_error(); } return; }
    zkw_batch_copy_d2d(b, dst, staged, bytes);
}

void zkw_batch_copy_d2h(zkw_batch* b, void* host, const void* src, size_t bytes) {
    if (!bytes) return;
    char *h = nullptr, *d = nullptr;
    if (b->down.alloc(bytes, 16, &h, &d) != ZKW_OK) { if (b->flush_rc == ZKW_OK) { b->flush_rc = ZKW_ERR_OOM; b->flush_err = zkw_last_error(); } return; }
    zkw_batch_copy_d2d(b, d, src, bytes);
    b->cur->landings.push_back(Landing{host, h, bytes});
}

int zkw_batch_sync(zkw_batch* b) {
    Fiber* f = b->cur;
    f->state = Fiber::WAIT_GPU;
    f->wants_gpu = true;
    f->wait_rc = ZKW_OK;
    b->park();
    if (f->wait_rc != ZKW_OK) return fail(f->wait_rc, "%s", b->flush_err.c_str());
    return ZKW_OK;
}

int zkw_batch_chains(zkw_batch* b, const ChainJob* full, size_t n_full, const LogChainJob* log, size_t n_log) {
    Fiber* f = b->cur;
    if (n_full || n_log) {
        f->chains.emplace_back();
        ChainReq& c = f->chains.back();
        if (n_full) c.full.assign(full, full + n_full);
        if (n_log) c.log.assign(log, log + n_log);
        f->ops.push_back(Op{Op::CHAIN, nullptr, 0, 0, 0, 0, f->chains.size() - 1});
    }
    return zkw_batch_sync(b);
}
