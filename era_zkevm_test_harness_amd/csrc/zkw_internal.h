// zkw_internal.h — the little that the library's translation units share besides include/zkw.h.
#pragma once
#include <string>
#include <vector>
// sets the calling thread's zkw_last_error() text and returns `code` (defined in zkw_api.hip)
int zkw_fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
// a context's device / HIP stream, and the reference count its witnesses, traces and communicators hold on it
struct zkw_ctx;
int zkw_ctx_device(const zkw_ctx* ctx);
void* zkw_ctx_stream(const zkw_ctx* ctx);
void zkw_ctx_retain(zkw_ctx* ctx);
void zkw_ctx_release(zkw_ctx* ctx);
// a context that belongs to a batch of blocks (zkw_batch.h): created without a stream, handed over to ordinary use afterwards
struct zkw_batch;
zkw_ctx* zkw_ctx_create_in_batch(int device_id, zkw_batch* b);
void zkw_ctx_enter_batch(zkw_ctx* ctx, zkw_batch* b);
zkw_batch* zkw_ctx_swap_batch(zkw_ctx* ctx, zkw_batch* b);
void zkw_ctx_leave_batch(zkw_ctx* ctx, void* stream);
void* zkw_device_shared_stream(int device_id);  // one stream per device for contexts that have none of their own (never destroyed)
int zkw_copy_device(zkw_ctx* ctx, void* dst, const void* src, size_t bytes);  // device -> device, ordered on the context's stream (or queued with its batch)
size_t zkw_ctx_scratch_bytes(const zkw_ctx* ctx);
void zkw_cache_stats(size_t* live_bytes, size_t* idle_bytes);
void zkw_ctx_scratch_mark(const zkw_ctx* ctx, std::vector<std::string>* names);
void zkw_ctx_scratch_release_since(zkw_ctx* ctx, const std::vector<std::string>& names);
