// zkw_vm_trace.hip — the pre-builder half of `create_artifacts_from_tracer`: from the tracer's raw record of one
// block (log queries, frame pushes and pops, stamped with VM cycles) to everything the MainVM instances and the log-based
// builders consume. C++ host code over include/zkw.h; the hashing runs in the library's kernels (k_encode_log, k_log_prehash +
// k_chain_log for the 3-round log-queue pushes, the level-synchronous callstack kernels behind zkw_callstack_simulate).
//
// Reference:
//   CallstackWithAuxData::{push_entry, pop_entry, add_log_query}  src/witness/callstack_handler.rs:174-460
//       a frame's forward and rollback log queues with frame markers; on `ret` they are glued to the parent's, on a panic the
//       rollback queue is appended, reversed, to the parent's FORWARD queue (the reverted writes are applied in reverse)
//   log-queue simulation                                          src/witness/oracle.rs:233-499
//       flat queue = forward ++ reverse(rollback) of the root frame, pushed through ONE LogQueueSimulator: chain_of_states,
//       the original (applied) queue = its prefix, marker positions, cycle -> (forward pointer, rollback pointer)
//   callstack replay                                              src/witness/oracle.rs:501-843
//       rollback tails of new frames, rollback head segments, the storage-log state per cycle, the callstack entries with
//       their rollback-queue segments pushed / popped through the CallstackSimulator, its sponge state per cycle
// The reference keeps this on the host too (it is bookkeeping over a few thousand frames); what is data-parallel — the
// encodings, two of the three rounds of every push, the callstack forest — is on the GPU, the serial third round is one chain.
// Reference panics (`assert!`, `expect`) become ZKW_ERR_CHECK_FAILED with the reference's message.
#include <algorithm>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/zkw.h"
#include "zkw_internal.h"

namespace {

enum ItemKind : uint8_t { Q_FWD = 0, Q_RB = 1, M_FWD_HEAD = 2, M_FWD_TAIL = 3, M_RB_HEAD = 4, M_RB_TAIL = 5 };
struct Item {
    uint8_t kind;
    uint32_t ref;    // log index for queries, frame index for markers
    uint32_t cycle;  // queries
    uint32_t frame;  // queries: the frame that issued it
};
enum Action : uint8_t { A_PUSH = 0, A_POP = 1, A_FRESH = 2, A_EXITED = 3 };
struct History {
    uint8_t action;
    bool panic;
    bool has_end;
    uint32_t frame, begin, end;
    int64_t affected_entry;  // index into `entries`, -1 = the empty root context
};
struct Frame {
    uint32_t index;
    History rec;
    std::vector<Item> fwd, rb;
};
struct LogState {
    uint64_t forward_tail[4] = {}, rollback_head[4] = {}, rollback_tail[4] = {};
    uint32_t forward_length = 0, rollback_length = 0, frame = 0;
    bool same_as(const LogState& o) const {
        return !memcmp(forward_tail, o.forward_tail, 32) && !memcmp(rollback_head, o.rollback_head, 32) && !memcmp(rollback_tail, o.rollback_tail, 32) &&
               forward_length == o.forward_length && rollback_length == o.rollback_length && frame == o.frame;
    }
};

}  // namespace

struct zkw_vm_trace {
    std::vector<zkw_log_query> flat_queries;
    std::vector<uint32_t> flat_cycles, flat_frames;
    std::vector<uint64_t> old_tails, new_tails;  // [n_flat][4]
    size_t original_len = 0;
    uint64_t global_end[4] = {};
    std::vector<uint32_t> frame_tail_cycles;
    std::vector<uint64_t> frame_tails;
    std::vector<uint32_t> head_cycles;
    std::vector<uint64_t> heads;
    std::vector<uint32_t> hist_cycles, hist_frames;
    std::vector<zkw_storage_log_detailed_state> hist;
    std::vector<uint32_t> cw_cycles, cw_depth;
    std::vector<uint8_t> cw_is_push;
    std::vector<zkw_callstack_entry> cw_entries;
    std::vector<uint64_t> cw_prev, cw_new, cw_rounds;
    std::vector<uint32_t> sponge_cycles;
    std::vector<uint64_t> sponge_states;
    std::vector<uint32_t> new_frame_cycles;
    std::vector<zkw_callstack_entry> new_frame_entries;
};

#define CHECK(cond, ...)                                         \
    do {                                                         \
        if (!(cond)) {                                           \
            delete T;                                            \
            return zkw_fail(ZKW_ERR_CHECK_FAILED, __VA_ARGS__);  \
        }                                                        \
    } while (0)
#define TRY(expr)                 \
    do {                          \
        int _rc = (expr);         \
        if (_rc != ZKW_OK) {      \
            delete T;             \
            return _rc;           \
        }                         \
    } while (0)

extern "C" int zkw_vm_trace_build(zkw_ctx* ctx, const zkw_vm_event* events, size_t n_events, const zkw_log_query* log_queries, size_t n_logs,
                                  const zkw_callstack_entry* entries, size_t n_entries, zkw_vm_trace** out) {
    if (!ctx || !out || !events || n_events == 0 || (n_logs && !log_queries) || (n_entries && !entries)) return zkw_fail(ZKW_ERR_INVALID, "zkw_vm_trace_build: null argument");
    if (events[0].kind != ZKW_VME_PUSH) return zkw_fail(ZKW_ERR_INVALID, "zkw_vm_trace_build: the trace starts with the initial frame's push (from_initial_callstack)");
    zkw_vm_trace* T = new zkw_vm_trace();

    // ---- A. callstack_handler.rs:174-460: replay the events
    std::vector<History> history;
    std::vector<Frame> stack;
    Frame cur;
    cur.index = 0;
    cur.rec = History{A_FRESH, false, false, 0, 0, 0, -1};
    cur.fwd.push_back(Item{M_FWD_HEAD, 0, 0, 0});
    cur.rb.push_back(Item{M_RB_TAIL, 0, 0, 0});
    history.push_back(cur.rec);
    uint32_t frame_counter = 1;
    for (size_t i = 0; i < n_events; i++) {
        const zkw_vm_event& e = events[i];
        if (e.kind == ZKW_VME_LOG) {
            CHECK(e.index < n_logs, "event %zu: log query %u of %zu", i, e.index, n_logs);
            const zkw_log_query& q = log_queries[e.index];
            CHECK(!q.rollback, "event %zu: a traced log query carries the rollback flag", i);
            cur.fwd.push_back(Item{Q_FWD, e.index, e.cycle, cur.index});
            if (q.rw_flag) cur.rb.push_back(Item{Q_RB, e.index, e.cycle, cur.index});
        } else if (e.kind == ZKW_VME_PUSH) {
            CHECK((size_t)2 * e.index + 1 < n_entries, "event %zu: frame entries %u of %zu", i, 2 * e.index + 1, n_entries);
            T->new_frame_cycles.push_back(e.cycle);
            T->new_frame_entries.push_back(entries[2 * e.index + 1]);
            Frame fresh;
            fresh.index = frame_counter++;
            fresh.rec = History{A_FRESH, false, false, fresh.index, e.cycle, 0, (int64_t)(2 * e.index + 1)};
            fresh.fwd.push_back(Item{M_FWD_HEAD, fresh.index, 0, 0});
            fresh.rb.push_back(Item{M_RB_TAIL, fresh.index, 0, 0});
            cur.rec.affected_entry = (int64_t)(2 * e.index);
            cur.rec.end = e.cycle;
            cur.rec.has_end = true;
            History pushed = cur.rec;
            pushed.action = A_PUSH;
            history.push_back(pushed);
            history.push_back(fresh.rec);
            stack.push_back(std::move(cur));
            cur = std::move(fresh);
        } else if (e.kind == ZKW_VME_POP) {
            CHECK(!stack.empty(), "event %zu: pop from the empty callstack", i);
            Frame parent = std::move(stack.back());
            stack.pop_back();
            parent.rec.begin = e.cycle;
            parent.rec.has_end = false;
            History popped = parent.rec;
            popped.action = A_POP;
            popped.panic = e.panicked != 0;
            Frame child = std::move(cur);
            cur = std::move(parent);
            cur.fwd.insert(cur.fwd.end(), child.fwd.begin(), child.fwd.end());
            cur.fwd.push_back(Item{M_FWD_TAIL, child.index, 0, 0});
            child.rb.push_back(Item{M_RB_HEAD, child.index, 0, 0});
            if (e.panicked) cur.fwd.insert(cur.fwd.end(), child.rb.rbegin(), child.rb.rend());  // reverted: applied, in reverse
            else cur.rb.insert(cur.rb.end(), child.rb.begin(), child.rb.end());
            History exited = child.rec;
            exited.action = A_EXITED;
            exited.panic = e.panicked != 0;
            exited.end = e.cycle;
            exited.has_end = true;
            history.push_back(exited);
            history.push_back(popped);
        } else {
            CHECK(false, "event %zu: unknown kind %u", i, e.kind);
        }
    }
    CHECK(stack.empty(), "parent frame didn't exit");  // oracle.rs:236-239

    // ---- B. oracle.rs:308-350: the flat queue and the marker positions
    const size_t n_fwd_items = cur.fwd.size();
    std::vector<Item> flat(cur.fwd);
    flat.insert(flat.end(), cur.rb.rbegin(), cur.rb.rend());
    std::vector<int64_t> rb_tail_pos(frame_counter, -2);
    bool orig_set = false;
    for (size_t k = 0; k < flat.size(); k++) {
        const Item& it = flat[k];
        if (k >= n_fwd_items && !orig_set) { T->original_len = T->flat_queries.size(); orig_set = true; }
        if (it.kind == M_RB_TAIL) rb_tail_pos[it.ref] = (int64_t)T->flat_queries.size() - 1;
        if (it.kind > Q_RB) continue;
        zkw_log_query q = log_queries[it.ref];
        q.rollback = it.kind == Q_RB;
        T->flat_queries.push_back(q);
        T->flat_cycles.push_back(it.cycle);
        T->flat_frames.push_back(it.frame);
    }
    if (!orig_set) T->original_len = T->flat_queries.size();
    const size_t n_flat = T->flat_queries.size();

    // ---- C. hash it: one LogQueueSimulator over the flat queue (QueueSimulator::push, circuit_encodings/src/lib.rs:179-221)
    T->old_tails.assign(4 * n_flat, 0);
    T->new_tails.assign(4 * n_flat, 0);
    if (n_flat) {
        std::vector<uint64_t> enc(20 * n_flat);
        TRY(zkw_encode_log_queries(ctx, T->flat_queries.data(), n_flat, nullptr, enc.data()));
        TRY(zkw_queue_push_chain_log(ctx, enc.data(), n_flat, nullptr, T->old_tails.data(), T->new_tails.data()));
        TRY(zkw_synchronize(ctx));
        memcpy(T->global_end, T->new_tails.data() + 4 * (n_flat - 1), 32);
    }
    const uint64_t* OT = T->old_tails.data();
    const uint64_t* NT = T->new_tails.data();

    // ---- oracle.rs:352-432: cycle -> (forward pointer, rollback pointer); rollbacks follow their forward twin
    struct Ptrs { size_t fwd = 0; int64_t rb = -1; };
    std::map<uint32_t, Ptrs> by_cycle;
    {
        std::map<uint32_t, size_t> first_forward;  // timestamp -> position of the forward item (sponges_data keys, :364-405)
        for (size_t p = 0; p < n_flat; p++) {
            const zkw_log_query& q = T->flat_queries[p];
            if (q.rollback) {
                auto f = first_forward.find(q.timestamp);
                CHECK(f != first_forward.end(), "rollbacks always happen after forward case (timestamp %u)", q.timestamp);
                auto c = by_cycle.find(T->flat_cycles[p]);
                CHECK(c != by_cycle.end(), "rollbacks always happen after forward case (cycle %u)", T->flat_cycles[p]);
                c->second.rb = (int64_t)p;
            } else {
                first_forward.emplace(q.timestamp, p);
                by_cycle[T->flat_cycles[p]].fwd = p;
            }
            if (p < T->original_len) {  // :445-494 the demultiplexing asserts
                CHECK(q.aux_byte <= 3, "unreachable aux byte %u", q.aux_byte);
                CHECK(q.aux_byte != 0 || q.shard_id <= 1, "unreachable shard %u", q.shard_id);
                CHECK(q.aux_byte != 3 || !q.rollback, "a precompile call cannot be rolled back");
            }
        }
    }

    // ---- oracle.rs:276-299, 526-563: beginnings of frames, rollback tails of new frames
    std::vector<uint32_t> begin_of_frame(frame_counter, 0);
    for (const History& h : history)
        if (h.action == A_FRESH) begin_of_frame[h.frame] = h.begin;
    begin_of_frame[0] = 0;
    std::vector<uint64_t> frame_tail(4 * (size_t)frame_counter);
    for (uint32_t f = 0; f < frame_counter; f++) {
        const uint64_t* tail = T->global_end;
        if (f) {
            CHECK(rb_tail_pos[f] != -2, "frame %u has no rollback tail marker", f);
            if (rb_tail_pos[f] >= 0) tail = NT + 4 * rb_tail_pos[f];
        }
        memcpy(frame_tail.data() + 4 * f, tail, 32);
        T->frame_tail_cycles.push_back(begin_of_frame[f]);
    }
    T->frame_tails = frame_tail;

    // ---- oracle.rs:571-578: rollback head segments in cycle order
    for (const auto& kv : by_cycle)
        if (kv.second.rb >= 0) {
            T->head_cycles.push_back(kv.first);
            T->heads.insert(T->heads.end(), OT + 4 * kv.second.rb, OT + 4 * kv.second.rb + 4);
        }

    // ---- oracle.rs:580-843: the callstack replay
    std::map<uint32_t, LogState> hist;
    LogState st;
    memcpy(st.rollback_head, T->global_end, 32);
    memcpy(st.rollback_tail, T->global_end, 32);
    std::vector<LogState> saved;
    bool have_merge = false, merge_panic = false;
    LogState merge;
    std::vector<uint8_t> ops;
    std::vector<zkw_callstack_entry> pushed;
    auto record = [&](uint32_t cycle, bool fresh, std::string* err) {
        auto it = hist.find(cycle);
        if (it != hist.end()) {
            if (fresh) {
                if (!(it->second.frame < st.frame && !memcmp(it->second.forward_tail, st.forward_tail, 32) && it->second.forward_length == st.forward_length))
                    *err = "frame divergence for cycle " + std::to_string(cycle);
            } else if (!it->second.same_as(st)) {
                *err = "duplicate divergence for cycle " + std::to_string(cycle);
            }
        }
        hist[cycle] = st;
    };
    auto walk_span = [&](uint32_t begin, uint32_t end, std::string* err) {  // (begin, end]: begin is bound to the previous span
        for (auto it = by_cycle.upper_bound(begin); it != by_cycle.end() && it->first <= end; ++it) {
            const uint64_t* nf = NT + 4 * it->second.fwd;
            if (memcmp(nf, st.forward_tail, 32)) {
                memcpy(st.forward_tail, nf, 32);
                st.forward_length++;
            }
            if (it->second.rb >= 0) {
                memcpy(st.rollback_head, OT + 4 * it->second.rb, 32);
                st.rollback_length++;
            }
            record(it->first, false, err);
        }
    };
    for (const History& h : history) {
        std::string err;
        switch (h.action) {
            case A_PUSH: {
                CHECK(h.has_end, "frame must end");
                walk_span(h.begin, h.end, &err);
                zkw_callstack_entry e;
                if (h.affected_entry >= 0) e = entries[h.affected_entry];
                else memset(&e, 0, sizeof e);
                memcpy(e.rollback_queue_head, st.rollback_head, 32);
                memcpy(e.rollback_queue_tail, st.rollback_tail, 32);
                e.rollback_queue_segment_length = st.rollback_length;
                saved.push_back(st);
                CHECK(T->cw_cycles.empty() || T->cw_cycles.back() != h.end, "trying to add callstack witness for cycle %u, but previous one is on the same cycle", h.end);
                ops.push_back(1);
                pushed.push_back(e);
                T->cw_cycles.push_back(h.end);
                break;
            }
            case A_POP: {
                CHECK(have_merge && merge_panic == h.panic && !saved.empty(), "callstack pop without a finished frame to merge (frame %u)", h.frame);
                have_merge = false;
                st = saved.back();  // == the popped entry's rollback head / tail / length (:690-704)
                saved.pop_back();
                st.frame = h.frame;
                CHECK(st.forward_length <= merge.forward_length, "divergence at frame %u", h.frame);
                memcpy(st.forward_tail, merge.forward_tail, 32);
                st.forward_length = merge.forward_length;
                if (h.panic) {
                    CHECK(!memcmp(st.forward_tail, merge.rollback_head, 32), "divergence at frame %u with panic", h.frame);
                    memcpy(st.forward_tail, merge.rollback_tail, 32);
                    st.forward_length += merge.rollback_length;
                } else {
                    CHECK(!memcmp(st.rollback_head, merge.rollback_tail, 32), "divergence at frame %u without panic", h.frame);
                    memcpy(st.rollback_head, merge.rollback_head, 32);
                    st.rollback_length += merge.rollback_length;
                }
                record(h.begin, false, &err);
                CHECK(T->cw_cycles.empty() || T->cw_cycles.back() != h.begin, "trying to add callstack witness for cycle %u, but previous one is on the same cycle", h.begin);
                ops.push_back(0);
                T->cw_cycles.push_back(h.begin);
                break;
            }
            case A_FRESH:
                st.frame = h.frame;
                st.rollback_length = 0;
                memcpy(st.rollback_head, frame_tail.data() + 4 * h.frame, 32);
                memcpy(st.rollback_tail, frame_tail.data() + 4 * h.frame, 32);
                record(h.begin, true, &err);
                break;
            case A_EXITED:
                CHECK(!have_merge, "two frames finished without a pop in between (frame %u)", h.frame);
                CHECK(h.has_end, "frame must end");
                walk_span(h.begin, h.end, &err);
                have_merge = true;
                merge_panic = h.panic;
                merge = st;
                break;
        }
        CHECK(err.empty(), "%s", err.c_str());
    }
    for (const auto& kv : hist) {
        T->hist_cycles.push_back(kv.first);
        T->hist_frames.push_back(kv.second.frame);
        zkw_storage_log_detailed_state s;
        memcpy(s.forward_tail, kv.second.forward_tail, 32);
        memcpy(s.rollback_head, kv.second.rollback_head, 32);
        memcpy(s.rollback_tail, kv.second.rollback_tail, 32);
        s.forward_length = kv.second.forward_length;
        s.rollback_length = kv.second.rollback_length;
        T->hist.push_back(s);
    }

    // ---- the CallstackSimulator over the whole sequence (FullWidthStackSimulator, lib.rs:558-644)
    const size_t n_ops = ops.size();
    T->cw_is_push = ops;
    T->cw_prev.assign(12 * n_ops, 0);
    T->cw_new.assign(12 * n_ops, 0);
    T->cw_rounds.assign(48 * n_ops, 0);
    T->cw_depth.assign(n_ops, 0);
    T->cw_entries.resize(n_ops);
    T->sponge_cycles.push_back(0);
    T->sponge_states.assign(12, 0);
    if (n_ops) {
        std::vector<uint32_t> entry_index(n_ops);
        TRY(zkw_callstack_simulate(ctx, ops.data(), n_ops, pushed.data(), pushed.size(), T->cw_prev.data(), T->cw_new.data(), T->cw_depth.data(),
                                   T->cw_rounds.data(), entry_index.data()));
        TRY(zkw_synchronize(ctx));
        for (size_t k = 0; k < n_ops; k++) T->cw_entries[k] = pushed[entry_index[k]];
        T->sponge_cycles.insert(T->sponge_cycles.end(), T->cw_cycles.begin(), T->cw_cycles.end());
        T->sponge_states.insert(T->sponge_states.end(), T->cw_new.begin(), T->cw_new.end());
    }
    *out = T;
    return ZKW_OK;
}

extern "C" void zkw_vm_trace_free(zkw_vm_trace* t) { delete t; }

namespace {
struct View { const void* p; size_t n, item; };
View view_of(const zkw_vm_trace* t, int what) {
    switch (what) {
        case ZKW_VMT_FLAT_QUERIES: return {t->flat_queries.data(), t->flat_queries.size(), sizeof(zkw_log_query)};
        case ZKW_VMT_FLAT_CYCLES: return {t->flat_cycles.data(), t->flat_cycles.size(), 4};
        case ZKW_VMT_FLAT_FRAMES: return {t->flat_frames.data(), t->flat_frames.size(), 4};
        case ZKW_VMT_FLAT_OLD_TAILS: return {t->old_tails.data(), t->old_tails.size() / 4, 32};
        case ZKW_VMT_FLAT_NEW_TAILS: return {t->new_tails.data(), t->new_tails.size() / 4, 32};
        case ZKW_VMT_NEW_FRAME_TAIL_CYCLES: return {t->frame_tail_cycles.data(), t->frame_tail_cycles.size(), 4};
        case ZKW_VMT_NEW_FRAME_TAILS: return {t->frame_tails.data(), t->frame_tails.size() / 4, 32};
        case ZKW_VMT_HEAD_SEGMENT_CYCLES: return {t->head_cycles.data(), t->head_cycles.size(), 4};
        case ZKW_VMT_HEAD_SEGMENTS: return {t->heads.data(), t->heads.size() / 4, 32};
        case ZKW_VMT_STORAGE_LOG_STATE_CYCLES: return {t->hist_cycles.data(), t->hist_cycles.size(), 4};
        case ZKW_VMT_STORAGE_LOG_STATE_FRAMES: return {t->hist_frames.data(), t->hist_frames.size(), 4};
        case ZKW_VMT_STORAGE_LOG_STATES: return {t->hist.data(), t->hist.size(), sizeof(zkw_storage_log_detailed_state)};
        case ZKW_VMT_CALLSTACK_WITNESS_CYCLES: return {t->cw_cycles.data(), t->cw_cycles.size(), 4};
        case ZKW_VMT_CALLSTACK_WITNESS_IS_PUSH: return {t->cw_is_push.data(), t->cw_is_push.size(), 1};
        case ZKW_VMT_CALLSTACK_WITNESS_ENTRIES: return {t->cw_entries.data(), t->cw_entries.size(), sizeof(zkw_callstack_entry)};
        case ZKW_VMT_CALLSTACK_WITNESS_PREVIOUS_STATES: return {t->cw_prev.data(), t->cw_prev.size() / 12, 96};
        case ZKW_VMT_CALLSTACK_WITNESS_NEW_STATES: return {t->cw_new.data(), t->cw_new.size() / 12, 96};
        case ZKW_VMT_CALLSTACK_WITNESS_DEPTHS: return {t->cw_depth.data(), t->cw_depth.size(), 4};
        case ZKW_VMT_CALLSTACK_WITNESS_ROUND_STATES: return {t->cw_rounds.data(), t->cw_rounds.size() / 48, 384};
        case ZKW_VMT_CALLSTACK_SPONGE_CYCLES: return {t->sponge_cycles.data(), t->sponge_cycles.size(), 4};
        case ZKW_VMT_CALLSTACK_SPONGE_STATES: return {t->sponge_states.data(), t->sponge_states.size() / 12, 96};
        case ZKW_VMT_NEW_FRAME_CYCLES: return {t->new_frame_cycles.data(), t->new_frame_cycles.size(), 4};
        case ZKW_VMT_NEW_FRAME_ENTRIES: return {t->new_frame_entries.data(), t->new_frame_entries.size(), sizeof(zkw_callstack_entry)};
    }
    return {nullptr, 0, 0};
}
}  // namespace

extern "C" size_t zkw_vm_trace_count(const zkw_vm_trace* t, int what) { return t ? view_of(t, what).n : 0; }
extern "C" const void* zkw_vm_trace_ptr(const zkw_vm_trace* t, int what) { return t ? view_of(t, what).p : nullptr; }
extern "C" int zkw_vm_trace_get(const zkw_vm_trace* t, int what, void* dst, size_t dst_bytes) {
    if (!t) return zkw_fail(ZKW_ERR_INVALID, "zkw_vm_trace_get: null trace");
    const View v = view_of(t, what);
    if (v.item == 0) return zkw_fail(ZKW_ERR_INVALID, "zkw_vm_trace_get: unknown array %d", what);
    if (dst_bytes < v.n * v.item || (v.n && !dst)) return zkw_fail(ZKW_ERR_INVALID, "zkw_vm_trace_get: %zu bytes needed, %zu given", v.n * v.item, dst_bytes);
    if (v.n) memcpy(dst, v.p, v.n * v.item);
    return ZKW_OK;
}
extern "C" int zkw_vm_trace_info(const zkw_vm_trace* t, zkw_vm_trace_summary* out) {
    if (!t || !out) return zkw_fail(ZKW_ERR_INVALID, "zkw_vm_trace_info: null argument");
    out->n_flat = t->flat_queries.size();
    out->original_log_queue_length = t->original_len;
    out->n_frames = t->frame_tail_cycles.size();
    memcpy(out->global_end_of_storage_log, t->global_end, 32);
    if (t->original_len) memcpy(out->original_log_queue_tail, t->new_tails.data() + 4 * (t->original_len - 1), 32);
    else memset(out->original_log_queue_tail, 0, 32);
    return ZKW_OK;
}

// the four FIFOs and three entry-state histories this half produces, as HOST arrays inside `s` (valid while the trace lives);
// the caller adds the streams the VM itself records (memory, storage queries, refunds, decommit requests) and slices
extern "C" int zkw_vm_trace_streams(const zkw_vm_trace* t, zkw_vm_tracer_streams* s) {
    if (!t || !s) return zkw_fail(ZKW_ERR_INVALID, "zkw_vm_trace_streams: null argument");
    s->stream_cycles[ZKW_VMS_ROLLBACK_TAILS_FOR_NEW_FRAMES] = t->frame_tail_cycles.data();
    s->stream_len[ZKW_VMS_ROLLBACK_TAILS_FOR_NEW_FRAMES] = t->frame_tail_cycles.size();
    s->stream_cycles[ZKW_VMS_CALLSTACK_VALUES] = t->cw_cycles.data();
    s->stream_len[ZKW_VMS_CALLSTACK_VALUES] = t->cw_cycles.size();
    s->stream_cycles[ZKW_VMS_ROLLBACK_HEAD_SEGMENTS] = t->head_cycles.data();
    s->stream_len[ZKW_VMS_ROLLBACK_HEAD_SEGMENTS] = t->head_cycles.size();
    s->stream_cycles[ZKW_VMS_NEW_FRAMES] = t->new_frame_cycles.data();
    s->stream_len[ZKW_VMS_NEW_FRAMES] = t->new_frame_cycles.size();
    s->callstack_sponge_cycles = t->sponge_cycles.data();
    s->callstack_sponge_states = t->sponge_states.data();
    s->n_callstack_sponges = t->sponge_cycles.size();
    s->storage_log_state_cycles = t->hist_cycles.data();
    s->storage_log_states = t->hist.data();
    s->n_storage_log_states = t->hist_cycles.size();
    memcpy(s->global_end_of_storage_log, t->global_end, 32);
    return ZKW_OK;
}
