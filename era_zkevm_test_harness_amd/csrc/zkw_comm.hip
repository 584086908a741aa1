// zkw_comm.hip — multi-GPU: the shard plan and the one collective. C++ host code; no kernels.
//
// Circuit instances are independent once the builders have fixed their hidden FSM inputs, so they are sharded across the
// GPUs of a node with no data-path collective; the only exchange is the gather of the per-instance closed-form records to
// the root, which replays the order-sensitive RecursionQueueSimulator pushes (src/witness/postprocessing/mod.rs:396-402)
// and assembles the scheduler witness (src/external_calls.rs:354-537).
//
// The gather is written once over a small transport table (send / recv / local copy / group begin-end / sync):
//   * RCCL  — device pointers, enqueued on the context's stream, xGMI inside a node. Resolved with dlopen when a
//             communicator of more than one rank is created: libzkw does not link against it, a single-GPU host never
//             loads it, and a failure to find it is an error code, not a load failure.
//   * TCP   — a full mesh of sockets on ONE host (every rank binds and connects on the same `address`, 127.0.0.1 unless
//             told otherwise: rank j connects to every i < j at port + i). Host pointers when the communicator has no
//             context, device pointers staged through host memory when it has one. It exists so that the collective's
//             logic (unequal counts, empty ranks, root != 0, rank order) is exercised by multi-process tests on machines
//             without GPUs, and as the transport of last resort where RCCL cannot start. The hello carries a job id
//             (ZKW_COMM_JOB_ID, 64 bits, 0 when unset) next to the rank: a process of another job that reaches the port
//             is turned away. It is a test / fallback transport for one trusted host, not an authenticated channel.
#include <arpa/inet.h>
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <rccl/rccl.h>
#include <sys/socket.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/zkw.h"
#include "zkw_internal.h"

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
    bool load() {
        if (handle) return true;
        // a process that already holds an RCCL (torch's) keeps using that copy: one runtime per process
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
            if (handle) break;
        }
        for (const char* n : names) {
            if (handle) break;
            handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        }
        if (!handle) { error = std::string("librccl.so not found: ") + dlerror(); return false; }
        auto sym = [&](const char* s) { void* p = dlsym(handle, s); if (!p) error = std::string("RCCL symbol missing: ") + s; return p; };
        GetUniqueId = reinterpret_cast<decltype(GetUniqueId)>(sym("ncclGetUniqueId"));
        CommInitRank = reinterpret_cast<decltype(CommInitRank)>(sym("ncclCommInitRank"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
        Send = reinterpret_cast<decltype(Send)>(sym("ncclSend"));
        Recv = reinterpret_cast<decltype(Recv)>(sym("ncclRecv"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
        if (!GetUniqueId || !CommInitRank || !CommDestroy || !Send || !Recv || !GroupStart || !GroupEnd || !GetErrorString) {
            handle = nullptr;
            return false;
        }
        return true;
    }
};
Rccl g_rccl;
std::mutex g_rccl_mu;

// ---- the transport table ------------------------------------------------------------------------------------------
struct Transport {
    virtual ~Transport() {}
    virtual int begin() { return ZKW_OK; }  // a group of sends / receives that may be posted in any order
    virtual int end() { return ZKW_OK; }
    virtual int send(const void* p, size_t bytes, int peer) = 0;
    virtual int recv(void* p, size_t bytes, int peer) = 0;
    virtual int copy(void* dst, const void* src, size_t bytes) = 0;  // this rank's own share
    virtual int sync() = 0;                                          // everything posted so far has landed
};

struct RcclTransport : Transport {
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    bool in_group = false;
    ~RcclTransport() override {
        if (comm) (void)g_rccl.CommDestroy(comm);
    }
    int nccl(ncclResult_t r, const char* what) {
        if (r == ncclSuccess) return ZKW_OK;
        if (in_group) { in_group = false; (void)g_rccl.GroupEnd(); }  // never leave the group open on an error path
        return zkw_fail(ZKW_ERR_HIP, "%s failed: %s", what, g_rccl.GetErrorString(r));
    }
    int begin() override { int rc = nccl(g_rccl.GroupStart(), "ncclGroupStart"); in_group = rc == ZKW_OK; return rc; }
    int end() override { in_group = false; return nccl(g_rccl.GroupEnd(), "ncclGroupEnd"); }
    int send(const void* p, size_t bytes, int peer) override { return nccl(g_rccl.Send(p, bytes, ncclUint8, peer, comm, stream), "ncclSend"); }
    int recv(void* p, size_t bytes, int peer) override { return nccl(g_rccl.Recv(p, bytes, ncclUint8, peer, comm, stream), "ncclRecv"); }
    int copy(void* dst, const void* src, size_t bytes) override {
        hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, stream);
        if (e != hipSuccess) {
            if (in_group) { in_group = false; (void)g_rccl.GroupEnd(); }
            return zkw_fail(ZKW_ERR_HIP, "hipMemcpyAsync failed: %s", hipGetErrorString(e));
        }
        return ZKW_OK;
    }
    int sync() override {
        hipError_t e = hipStreamSynchronize(stream);
        return e == hipSuccess ? ZKW_OK : zkw_fail(ZKW_ERR_HIP, "hipStreamSynchronize failed: %s", hipGetErrorString(e));
    }
};

// one rank alone: the gather is a copy
struct LocalTransport : Transport {
    hipStream_t stream = nullptr;
    bool device = false;
    int send(const void*, size_t, int) override { return zkw_fail(ZKW_ERR_INVALID, "single-rank communicator: no peers"); }
    int recv(void*, size_t, int) override { return zkw_fail(ZKW_ERR_INVALID, "single-rank communicator: no peers"); }
    int copy(void* dst, const void* src, size_t bytes) override {
        if (!device) { memcpy(dst, src, bytes); return ZKW_OK; }
        hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, stream);
        return e == hipSuccess ? ZKW_OK : zkw_fail(ZKW_ERR_HIP, "hipMemcpyAsync failed: %s", hipGetErrorString(e));
    }
    int sync() override {
        if (!device) return ZKW_OK;
        hipError_t e = hipStreamSynchronize(stream);
        return e == hipSuccess ? ZKW_OK : zkw_fail(ZKW_ERR_HIP, "hipStreamSynchronize failed: %s", hipGetErrorString(e));
    }
};

struct TcpTransport : Transport {
    std::vector<int> fd;  // fd[peer], -1 for self
    hipStream_t stream = nullptr;
    bool device = false;  // buffers are device pointers: stage through host memory
    ~TcpTransport() override {
        for (int f : fd)
            if (f >= 0) close(f);
    }
    static int write_all(int f, const void* p, size_t n) {
        const char* c = static_cast<const char*>(p);
        while (n) {
            ssize_t k = ::send(f, c, n, MSG_NOSIGNAL);
            if (k < 0 && (errno == EINTR || errno == EAGAIN || errno == EWOULDBLOCK)) continue;
            if (k <= 0) return -1;
            c += k;
            n -= (size_t)k;
        }
        return 0;
    }
    static int read_all(int f, void* p, size_t n) {
        char* c = static_cast<char*>(p);
        while (n) {
            ssize_t k = ::recv(f, c, n, 0);
            if (k < 0 && (errno == EINTR || errno == EAGAIN || errno == EWOULDBLOCK)) continue;
            if (k <= 0) return -1;
            c += k;
            n -= (size_t)k;
        }
        return 0;
    }
    int send(const void* p, size_t bytes, int peer) override {
        std::vector<char> stage;
        if (device) {
            stage.resize(bytes);
            if (hipStreamSynchronize(stream) != hipSuccess || hipMemcpy(stage.data(), p, bytes, hipMemcpyDeviceToHost) != hipSuccess)
                return zkw_fail(ZKW_ERR_HIP, "tcp transport: staging the records failed");
            p = stage.data();
        }
        if (write_all(fd[peer], p, bytes) != 0) return zkw_fail(ZKW_ERR_HIP, "tcp transport: send to rank %d failed", peer);
        return ZKW_OK;
    }
    int recv(void* p, size_t bytes, int peer) override {
        if (!device) {
            if (read_all(fd[peer], p, bytes) != 0) return zkw_fail(ZKW_ERR_HIP, "tcp transport: receive from rank %d failed", peer);
            return ZKW_OK;
        }
        std::vector<char> stage(bytes);
        if (read_all(fd[peer], stage.data(), bytes) != 0) return zkw_fail(ZKW_ERR_HIP, "tcp transport: receive from rank %d failed", peer);
        if (hipMemcpy(p, stage.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) return zkw_fail(ZKW_ERR_HIP, "tcp transport: upload failed");
        return ZKW_OK;
    }
    int copy(void* dst, const void* src, size_t bytes) override {
        if (!device) { memmove(dst, src, bytes); return ZKW_OK; }
        hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, stream);
        return e == hipSuccess ? ZKW_OK : zkw_fail(ZKW_ERR_HIP, "hipMemcpyAsync failed: %s", hipGetErrorString(e));
    }
    int sync() override {
        if (!device) return ZKW_OK;
        return hipStreamSynchronize(stream) == hipSuccess ? ZKW_OK : zkw_fail(ZKW_ERR_HIP, "hipStreamSynchronize failed");
    }
};

}  // namespace

struct zkw_comm {
    zkw_ctx* ctx = nullptr;  // may be NULL for a host-memory TCP communicator
    int rank = 0, world = 1;
    Transport* tp = nullptr;
    // device staging of zkw_gather_records, grown on demand and reused by every call (freed with the communicator)
    void *d_send = nullptr, *d_recv = nullptr;
    size_t send_cap = 0, recv_cap = 0;
};

extern "C" int zkw_comm_unique_id(uint8_t id[ZKW_COMM_ID_BYTES]) {
    if (!id) return zkw_fail(ZKW_ERR_INVALID, "zkw_comm_unique_id: null argument");
    std::lock_guard<std::mutex> g(g_rccl_mu);
    if (!g_rccl.load()) return zkw_fail(ZKW_ERR_NO_DEVICE, "%s", g_rccl.error.c_str());
    static_assert(ZKW_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    ncclUniqueId u;
    ncclResult_t r = g_rccl.GetUniqueId(&u);
    if (r != ncclSuccess) return zkw_fail(ZKW_ERR_HIP, "ncclGetUniqueId failed: %s", g_rccl.GetErrorString(r));
    memcpy(id, u.internal, ZKW_COMM_ID_BYTES);
    return ZKW_OK;
}

static int comm_init(zkw_ctx* ctx, const uint8_t id[ZKW_COMM_ID_BYTES], int rank, int world, bool force_rccl, zkw_comm** out) {
    if (!ctx || !out || world < 1 || rank < 0 || rank >= world || ((world > 1 || force_rccl) && !id)) return zkw_fail(ZKW_ERR_INVALID, "zkw_comm_init: bad argument");
    if (hipSetDevice(zkw_ctx_device(ctx)) != hipSuccess) return zkw_fail(ZKW_ERR_HIP, "zkw_comm_init: hipSetDevice failed");
    zkw_comm* c = new zkw_comm();
    c->ctx = ctx;
    c->rank = rank;
    c->world = world;
    if (world == 1 && !force_rccl) {  // a single rank needs no transport at all
        LocalTransport* t = new LocalTransport();
        t->stream = static_cast<hipStream_t>(zkw_ctx_stream(ctx));
        t->device = true;
        c->tp = t;
    } else {
        std::lock_guard<std::mutex> g(g_rccl_mu);
        if (!g_rccl.load()) { delete c; return zkw_fail(ZKW_ERR_NO_DEVICE, "%s", g_rccl.error.c_str()); }
        RcclTransport* t = new RcclTransport();
        t->stream = static_cast<hipStream_t>(zkw_ctx_stream(ctx));
        ncclUniqueId u;
        memcpy(u.internal, id, ZKW_COMM_ID_BYTES);
        ncclResult_t r = g_rccl.CommInitRank(&t->comm, world, u, rank);
        if (r != ncclSuccess) { t->comm = nullptr; delete t; delete c; return zkw_fail(ZKW_ERR_HIP, "ncclCommInitRank failed: %s", g_rccl.GetErrorString(r)); }
        c->tp = t;
    }
    zkw_ctx_retain(ctx);
    *out = c;
    return ZKW_OK;
}

extern "C" int zkw_comm_init(zkw_ctx* ctx, const uint8_t id[ZKW_COMM_ID_BYTES], int rank, int world, zkw_comm** out) {
    return comm_init(ctx, id, rank, world, false, out);
}

// The RCCL transport whatever the world size: with world == 1 this is ncclCommInitRank over one rank, so that the dlopen,
// the symbol table, the stream ordering and the error paths of the transport run on a single-GPU host (tests).
extern "C" int zkw_comm_init_rccl(zkw_ctx* ctx, const uint8_t id[ZKW_COMM_ID_BYTES], int rank, int world, zkw_comm** out) {
    return comm_init(ctx, id, rank, world, true, out);
}

// One grouped send + receive of `bytes` from this rank to `peer` and back from it through the communicator's transport table
// (peer == own rank: a send to self matched by a receive from self inside one group, which RCCL supports). src / dst as in
// zkw_gather_closed_form_inputs (device pointers; host pointers on a context-less TCP communicator; they must not overlap).
// Enqueued on the stream (zkw_comm_synchronize). The local transport has no peers and refuses.
extern "C" int zkw_comm_exchange(zkw_comm* c, const void* src, void* dst, size_t bytes, int peer) {
    if (!c || !src || !dst || bytes == 0 || peer < 0 || peer >= c->world) return zkw_fail(ZKW_ERR_INVALID, "zkw_comm_exchange: bad argument");
    if (c->ctx && hipSetDevice(zkw_ctx_device(c->ctx)) != hipSuccess) return zkw_fail(ZKW_ERR_HIP, "hipSetDevice failed");
    Transport& t = *c->tp;
    int rc = t.begin();
    if (rc != ZKW_OK) return rc;
    if ((rc = t.send(src, bytes, peer)) != ZKW_OK) return rc;
    if ((rc = t.recv(dst, bytes, peer)) != ZKW_OK) return rc;
    return t.end();
}

extern "C" int zkw_comm_init_tcp(zkw_ctx* ctx, const char* address, int port, int rank, int world, int timeout_ms, zkw_comm** out) {
    if (!out || !address || world < 1 || rank < 0 || rank >= world || port <= 0 || port + world > 65535) return zkw_fail(ZKW_ERR_INVALID, "zkw_comm_init_tcp: bad argument");
    if (ctx && hipSetDevice(zkw_ctx_device(ctx)) != hipSuccess) return zkw_fail(ZKW_ERR_HIP, "zkw_comm_init_tcp: hipSetDevice failed");
    TcpTransport* t = new TcpTransport();
    t->fd.assign((size_t)world, -1);
    t->device = ctx != nullptr;
    if (ctx) t->stream = static_cast<hipStream_t>(zkw_ctx_stream(ctx));
    auto fail_with = [&](const char* what) {
        const int err = errno;  // before the cleanup's close() calls can change it
        delete t;
        return zkw_fail(ZKW_ERR_HIP, "zkw_comm_init_tcp (rank %d): %s: %s", rank, what, strerror(err));
    };
    struct Hello { uint64_t job; int32_t rank; int32_t world; } hello{0, rank, world};
    if (const char* j = getenv("ZKW_COMM_JOB_ID")) hello.job = strtoull(j, nullptr, 0);
    sockaddr_in sa{};
    sa.sin_family = AF_INET;
    if (inet_pton(AF_INET, address, &sa.sin_addr) != 1) { delete t; return zkw_fail(ZKW_ERR_INVALID, "zkw_comm_init_tcp: '%s' is not an IPv4 address", address); }
    const int one = 1;
    // listen for the higher ranks first, then connect to the lower ones (whose listeners exist or appear within the timeout)
    int lfd = -1;
    if (rank + 1 < world) {
        lfd = socket(AF_INET, SOCK_STREAM, 0);
        if (lfd < 0) return fail_with("socket");
        setsockopt(lfd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
        sa.sin_port = htons((uint16_t)(port + rank));
        if (bind(lfd, reinterpret_cast<sockaddr*>(&sa), sizeof sa) != 0 || listen(lfd, world) != 0) { close(lfd); return fail_with("bind/listen"); }
    }
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(timeout_ms > 0 ? timeout_ms : 30000);
    for (int peer = 0; peer < rank; peer++) {
        for (;;) {
            int f = socket(AF_INET, SOCK_STREAM, 0);
            if (f < 0) { if (lfd >= 0) close(lfd); return fail_with("socket"); }
            sa.sin_port = htons((uint16_t)(port + peer));
            if (connect(f, reinterpret_cast<sockaddr*>(&sa), sizeof sa) == 0) {
                setsockopt(f, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
                if (TcpTransport::write_all(f, &hello, sizeof hello) != 0) { close(f); if (lfd >= 0) close(lfd); return fail_with("hello"); }
                // the listener answers one byte: 1 = taken, 0 = turned away (another job id / a bad rank). A listener that is not one of this
                // job's (or that closes the socket) is an error HERE, not at the first collective
                unsigned char ack = 0;
                bool got = false;
                for (;;) {
                    const auto left = std::chrono::duration_cast<std::chrono::milliseconds>(deadline - std::chrono::steady_clock::now()).count();
                    if (left <= 0) { errno = ETIMEDOUT; break; }
                    pollfd pf{f, POLLIN, 0};
                    const int k = poll(&pf, 1, (int)std::min<long long>(left, 1000));
                    if (k > 0) { const ssize_t n = recv(f, &ack, 1, 0); if (n == 1) got = true; else if (n == 0) errno = ECONNRESET; break; }
                    if (k < 0 && errno != EINTR) break;
                }
                if (!got || ack != 1) { if (got) errno = ECONNREFUSED; close(f); if (lfd >= 0) close(lfd); return fail_with(got ? "hello turned away by the listener (job id or rank mismatch)" : "hello acknowledgement"); }
                t->fd[peer] = f;
                break;
            }
            close(f);
            if (std::chrono::steady_clock::now() > deadline) { if (lfd >= 0) close(lfd); return fail_with("connect (timed out)"); }
            std::this_thread::sleep_for(std::chrono::milliseconds(20));
        }
    }
    // accept() and the hello are bounded by poll() on the descriptors: a receive timeout set on the listening socket would
    // be inherited by every accepted data socket and fail a later gather whose peer is still computing
    auto wait_readable = [&](int f) {
        for (;;) {
            const auto left = std::chrono::duration_cast<std::chrono::milliseconds>(deadline - std::chrono::steady_clock::now()).count();
            if (left <= 0) { errno = ETIMEDOUT; return false; }
            pollfd pf{f, POLLIN, 0};
            const int k = poll(&pf, 1, (int)std::min<long long>(left, 1000));
            if (k > 0) return true;
            if (k < 0 && errno != EINTR) return false;
        }
    };
    for (int k = rank + 1; k < world;) {
        if (!wait_readable(lfd)) { close(lfd); return fail_with("accept (timed out)"); }
        int f = accept(lfd, nullptr, nullptr);
        if (f < 0) {
            if (errno == EINTR || errno == EAGAIN || errno == ECONNABORTED) continue;
            close(lfd);
            return fail_with("accept");
        }
        setsockopt(f, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
        Hello h{};
        {   // the WHOLE hello under the deadline: a peer that sends part of it must not hang the init past timeout_ms
            size_t have = 0;
            bool ok = true;
            while (have < sizeof h) {
                if (!wait_readable(f)) { ok = false; break; }
                const ssize_t n = recv(f, reinterpret_cast<char*>(&h) + have, sizeof h - have, 0);
                if (n > 0) have += (size_t)n;
                else if (n == 0 || (errno != EINTR && errno != EAGAIN)) { ok = false; if (n == 0) errno = ECONNRESET; break; }
            }
            if (!ok) { close(f); close(lfd); return fail_with("hello (timed out or cut short)"); }
        }
        const unsigned char no = 0, yes = 1;
        if (h.job != hello.job) { (void)!write(f, &no, 1); close(f); continue; }  // a process of another job: told so, keep waiting for ours
        if (h.world != world || h.rank <= rank || h.rank >= world || t->fd[h.rank] >= 0) { (void)!write(f, &no, 1); close(f); close(lfd); delete t; return zkw_fail(ZKW_ERR_HIP, "zkw_comm_init_tcp: bad hello (rank %d of %d)", h.rank, h.world); }
        if (TcpTransport::write_all(f, &yes, 1) != 0) { close(f); close(lfd); return fail_with("hello acknowledgement"); }
        t->fd[h.rank] = f;
        k++;
    }
    if (lfd >= 0) close(lfd);
    zkw_comm* c = new zkw_comm();
    c->ctx = ctx;
    c->rank = rank;
    c->world = world;
    c->tp = t;
    if (ctx) zkw_ctx_retain(ctx);
    *out = c;
    return ZKW_OK;
}

extern "C" void zkw_comm_destroy(zkw_comm* c) {
    if (!c) return;
    if (c->ctx) (void)hipSetDevice(zkw_ctx_device(c->ctx));
    if (c->tp) (void)c->tp->sync();
    delete c->tp;
    if (c->d_send) (void)hipFree(c->d_send);
    if (c->d_recv) (void)hipFree(c->d_recv);
    zkw_ctx* owner = c->ctx;
    delete c;
    if (owner) zkw_ctx_release(owner);
}

extern "C" int zkw_comm_synchronize(zkw_comm* c) {
    if (!c) return zkw_fail(ZKW_ERR_INVALID, "zkw_comm_synchronize: null communicator");
    if (c->ctx && hipSetDevice(zkw_ctx_device(c->ctx)) != hipSuccess) return zkw_fail(ZKW_ERR_HIP, "hipSetDevice failed");
    return c->tp->sync();
}

// counts[r] records of record_bytes each from rank r, concatenated in rank order into recv on root.
extern "C" int zkw_gather_closed_form_inputs(zkw_comm* c, const void* records, const uint64_t* counts, size_t record_bytes,
                                             int root, void* recv) {
    if (!c || !counts || record_bytes == 0 || root < 0 || root >= c->world) return zkw_fail(ZKW_ERR_INVALID, "zkw_gather_closed_form_inputs: bad argument");
    if (c->ctx && hipSetDevice(zkw_ctx_device(c->ctx)) != hipSuccess) return zkw_fail(ZKW_ERR_HIP, "hipSetDevice failed");
    const size_t mine = (size_t)counts[c->rank] * record_bytes;
    if (mine && !records) return zkw_fail(ZKW_ERR_INVALID, "zkw_gather_closed_form_inputs: no records given");
    uint64_t total = 0;
    for (int r = 0; r < c->world; r++) total += counts[r];
    if (c->rank == root && total && !recv) return zkw_fail(ZKW_ERR_INVALID, "zkw_gather_closed_form_inputs: the root needs a receive buffer");
    Transport& t = *c->tp;
    int rc = t.begin();
    if (rc != ZKW_OK) return rc;
    if (c->rank == root) {
        size_t off = 0;
        for (int r = 0; r < c->world && rc == ZKW_OK; r++) {
            const size_t bytes = (size_t)counts[r] * record_bytes;
            if (bytes) rc = r == root ? t.copy(static_cast<char*>(recv) + off, records, bytes) : t.recv(static_cast<char*>(recv) + off, bytes, r);
            off += bytes;
        }
    } else if (mine) {
        rc = t.send(records, mine, root);
    }
    if (rc != ZKW_OK) return rc;  // (the transport has closed its group)
    return t.end();
}

// The gather as the block sequencer and the bench use it: every rank knows the ordered instance list and its owners
// (zkw_shard_lpt), holds the records of ITS instances in list order (host memory), and the root receives all n records in
// list (= emission) order, ready for the recursion-queue replay. Synchronous on every rank: when it returns, the rank's send
// has left its buffers and the staging buffers may be reused (they are, by the next call).
extern "C" int zkw_gather_records(zkw_comm* c, const uint32_t* owner, size_t n, const void* mine, size_t record_bytes, int root, void* out) {
    if (!c || (n && !owner) || record_bytes == 0 || root < 0 || root >= c->world) return zkw_fail(ZKW_ERR_INVALID, "zkw_gather_records: bad argument");
    std::vector<uint64_t> counts((size_t)c->world, 0);
    for (size_t k = 0; k < n; k++) {
        if (owner[k] >= (uint32_t)c->world) return zkw_fail(ZKW_ERR_INVALID, "zkw_gather_records: owner[%zu] = %u of %d ranks", k, owner[k], c->world);
        counts[owner[k]]++;
    }
    const size_t mine_bytes = (size_t)counts[c->rank] * record_bytes, all_bytes = n * record_bytes;
    if (mine_bytes && !mine) return zkw_fail(ZKW_ERR_INVALID, "zkw_gather_records: this rank owns %llu records and passed none", (unsigned long long)counts[c->rank]);
    if (c->rank == root && n && !out) return zkw_fail(ZKW_ERR_INVALID, "zkw_gather_records: the root needs an output array");
    std::vector<char> got(c->rank == root ? all_bytes : 0);  // rank order
    int rc;
    if (!c->ctx) {
        if ((rc = zkw_gather_closed_form_inputs(c, mine, counts.data(), record_bytes, root, got.data())) != ZKW_OK) return rc;
    } else {
        if (hipSetDevice(zkw_ctx_device(c->ctx)) != hipSuccess) return zkw_fail(ZKW_ERR_HIP, "hipSetDevice failed");
        hipStream_t st = static_cast<hipStream_t>(zkw_ctx_stream(c->ctx));
        auto grow = [&](void** p, size_t* cap, size_t need) {
            if (need <= *cap) return hipSuccess;
            if (*p) (void)hipFree(*p);
            *p = nullptr;
            *cap = 0;
            hipError_t e = hipMalloc(p, need);
            if (e == hipSuccess) *cap = need;
            return e;
        };
        if (grow(&c->d_send, &c->send_cap, mine_bytes ? mine_bytes : 1) != hipSuccess || (c->rank == root && grow(&c->d_recv, &c->recv_cap, all_bytes ? all_bytes : 1) != hipSuccess))
            return zkw_fail(ZKW_ERR_OOM, "zkw_gather_records: staging allocation failed");
        if (mine_bytes && hipMemcpyAsync(c->d_send, mine, mine_bytes, hipMemcpyHostToDevice, st) != hipSuccess) return zkw_fail(ZKW_ERR_HIP, "zkw_gather_records: upload failed");
        if ((rc = zkw_gather_closed_form_inputs(c, c->d_send, counts.data(), record_bytes, root, c->rank == root ? c->d_recv : nullptr)) != ZKW_OK) return rc;
        if (c->rank == root && all_bytes && hipMemcpyAsync(got.data(), c->d_recv, all_bytes, hipMemcpyDeviceToHost, st) != hipSuccess) return zkw_fail(ZKW_ERR_HIP, "zkw_gather_records: download failed");
    }
    if ((rc = c->tp->sync()) != ZKW_OK) return rc;  // EVERY rank: the send has completed before its buffer is reused
    if (c->rank != root) return ZKW_OK;
    std::vector<size_t> next((size_t)c->world, 0);
    for (int r = 1; r < c->world; r++) next[r] = next[r - 1] + counts[r - 1];
    for (size_t k = 0; k < n; k++) memcpy(static_cast<char*>(out) + k * record_bytes, got.data() + (next[owner[k]]++) * record_bytes, record_bytes);
    return ZKW_OK;
}

// Longest-processing-time assignment of an ordered instance list to ranks, weight = rows the reference's synthesis of
// that circuit type uses (setup/base_layer/finalization_hint_N.json): the shard plan of SURVEY 8(e).
// Deterministic (ties: lower instance index first, lowest rank first), so every rank computes the same plan. No GPU needed.
extern "C" int zkw_shard_lpt(const uint8_t* circuit_types, size_t n, int world, uint32_t* owner) {
    static const uint32_t ROWS_USED[14] = {0, 1033358, 1021855, 1045894, 770857, 957656, 1039794, 938955, 1044096, 1046318, 1027359, 590817, 590817, 1038150};
    if ((n && (!circuit_types || !owner)) || world < 1) return zkw_fail(ZKW_ERR_INVALID, "zkw_shard_lpt: bad argument");
    std::vector<size_t> order(n);
    for (size_t i = 0; i < n; i++) {
        if (circuit_types[i] < 1 || circuit_types[i] > 13) return zkw_fail(ZKW_ERR_INVALID, "zkw_shard_lpt: circuit type %u", circuit_types[i]);
        order[i] = i;
    }
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return ROWS_USED[circuit_types[a]] > ROWS_USED[circuit_types[b]]; });
    std::vector<uint64_t> load((size_t)world, 0);
    for (size_t i : order) {
        int best = 0;
        for (int r = 1; r < world; r++)
            if (load[r] < load[best]) best = r;
        owner[i] = (uint32_t)best;
        load[best] += ROWS_USED[circuit_types[i]];
    }
    return ZKW_OK;
}
