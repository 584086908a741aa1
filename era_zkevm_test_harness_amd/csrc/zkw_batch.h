// zkw_batch.h — many blocks in flight on ONE host thread and a handful of streams: builders as fibers, launches merged by stage.
//
// zkw_blocks_run used to give every block a host thread per branch of its builder graph and a HIP stream per context: at 96 blocks in
// flight that was ~700 threads serialising inside the HIP runtime and 1 344 streams on 8 hardware queues, and the builders of a batch
// took 1.7 x as long as one block's (profiles/r05/blocks_in_flight.txt). Here a block's builder branches are FIBERS (ucontext) of the
// calling thread: the builder code is the code zkw_block_run runs, unchanged — but a context that belongs to a batch launches nothing
// itself (zkw_launch.h). A fiber runs until it needs a result on the host (a count read back, a queue chain, a join) and parks; when no
// fiber can run, the batch FLUSHES: the launches the fibers left are walked position by position (a fiber's k-th pending launch in round
// k, so a fiber's own order is kept on the one in-order stream), the launches of a round that name the same kernel leave as ONE k_multi
// launch over a job table, the Poseidon2 queue chains of a round as one chain launch on a high-priority stream of their own (their
// latency — a second for a memory queue — must not sit in front of the short kernels), the small read-backs as one device-side gather and
// one copy. Then the batch waits for what it sent and wakes the fibers whose results have arrived. Equal stages of equal blocks park at
// the same place, so a stage of all K blocks is one launch: K x ~110 launches become ~110 + what differing blocks add.
//
// Everything here runs on the thread that called zkw_batch_run; nothing is thread-safe and nothing needs to be.
#pragma once
#include <functional>
#include <vector>

#include "zkw_launch.h"

namespace zkw {
struct ChainJob;
struct LogChainJob;
}  // namespace zkw

struct zkw_batch;

zkw_batch* zkw_batch_create(int device);
void zkw_batch_destroy(zkw_batch* b);
// runs `roots` as fibers (and whatever they spawn) to completion; the first failure's code, its text in zkw_last_error()
int zkw_batch_run(zkw_batch* b, const std::vector<std::function<int()>>& roots);
hipStream_t zkw_batch_stream(zkw_batch* b);  // the in-order stream every merged launch travels on

// ---- from inside a fiber -----------------------------------------------------------------------------------------------------------
bool zkw_batch_in_fiber(const zkw_batch* b);
int zkw_batch_spawn(zkw_batch* b, std::function<int()> fn);  // fiber id
int zkw_batch_join(zkw_batch* b, int fiber);                 // parks until that fiber has returned; its return code
// (lds_bytes: dynamic LDS of the launch; launches merge when kernel AND lds_bytes agree)
void zkw_batch_launch(zkw_batch* b, const zkw::BatchKernel* k, dim3 grid, size_t lds_bytes, const void* tup);
// `bytes` of host memory captured NOW into the batch's upload arena; the device address is valid until the batch is destroyed and holds
// the bytes for every launch queued after this call
void* zkw_batch_upload(zkw_batch* b, const void* host, size_t bytes, size_t align);
void zkw_batch_memset(zkw_batch* b, void* dev, int value, size_t bytes);
void zkw_batch_copy_d2d(zkw_batch* b, void* dst, const void* src, size_t bytes);
void zkw_batch_copy_h2d(zkw_batch* b, void* dst, const void* host, size_t bytes);  // host bytes captured now
void zkw_batch_copy_d2h(zkw_batch* b, void* host, const void* src, size_t bytes);  // lands in `host` before the fiber's next sync returns
int zkw_batch_sync(zkw_batch* b);  // parks until everything this fiber queued has run
// queue chains (device pointers): parks until they have run
int zkw_batch_chains(zkw_batch* b, const zkw::ChainJob* full, size_t n_full, const zkw::LogChainJob* log, size_t n_log);
