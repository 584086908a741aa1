// netlist_queue_kernels.cuh — the QUEUE SECTION of the netlist circuits (format, reference citations: include/zkw_netlist_queue.h):
// the request-queue pops and memory-queue pushes of Sha256RoundFunction (6) and CodeDecommitter (3) as Poseidon2 rows below the
// hash netlist of a "zkw trace v4" trace, tied to it by copy constraints (value nibbles of the hashed words / of the digest).
// Queue arithmetic: circuit_encodings/src/lib.rs:180-203, 391-429; encodings: memory_query.rs:24-118, log_query.rs:102-396,
// decommittment_request.rs:9-74. Placement is this library's own.
//
// Fill (k_nlq_fill): a ROW of 16 lanes owns one operation of one cycle and runs its permutations cooperatively (p2::Coop); the four
// rows of a wave take four consecutive cycles of the same operation (uniform control flow); the section is region-major (consecutive
// cycles = consecutive rows). The linked cells are read back from the netlist rows that k_nl_fill wrote earlier on the stream.
#pragma once
#include "../../include/zkw_netlist_queue.h"
#include "netlist_kernels.cuh"

namespace zkw {

// what a round of the builder's walk did to the queues (written by k_precompile_walk / k_decommitter_sha): the request it belongs to,
// the index of the first memory query it pushes, how many it pushes, flags: 1 = the round pops its request, 2 = it pushes one more
// query than the reads (sha256: the digest write)
struct NlqQueueIn { const void* items; const u64* states; u64 init[12]; u64 n_items; };
struct NlqJob { const nlq_feed* feed; u64* trace; NlqQueueIn queues[NLQ_MAX_QUEUES]; };
struct NlqFeedJob { const RoundOps* round_ops; u64 first_round; u32 n_active; nlq_feed* feed; u64 n_items; /* L1MessagesHasher: messages of the queue */ };
struct NlqFreeHome { uint16_t row, col; };  // the one cell of a cycle that holds FREE element i

#define NLQ_TR(col, row) trace[(size_t)(col) * n_rows + (size_t)(row)]

// feed of a cycle from the builder's per-round record (the oracle walks the rounds instead: orc_sha256_queue_feed)
static __device__ __forceinline__ void k_nlq_feed(const VB& vb, int circuit_type, const NlqFeedJob* __restrict__ jobs, u32 capacity, u32 n_ops) {
    const NlqFeedJob j = jobs[vb.y];
    const u32 c = vb.x * blockDim.x + threadIdx.x;
    if (c >= capacity) return;
    nlq_feed* f = j.feed + (size_t)c * n_ops;
    if (circuit_type == 13) {  // message m is popped in the cycle that absorbs its first byte (at most two per cycle)
        u64 m0 = NLQ_LH_FIRST(c), m1 = NLQ_LH_FIRST(c + 1);
        if (m0 > j.n_items) m0 = j.n_items;
        if (m1 > j.n_items) m1 = j.n_items;
        for (u32 k = 0; k < 2; k++) f[k] = m0 + k < m1 ? nlq_feed{1, (u32)(m0 + k)} : nlq_feed{0, (u32)m1};
        return;
    }
    if (c >= j.n_active) {  // idle: everything disabled, the queues stay where the last active round left them
        u32 nreq = 0, nq = 0;
        if (j.n_active || j.first_round) {
            const RoundOps ro = j.round_ops[j.first_round + j.n_active - 1];
            nreq = ro.request + 1; nq = ro.first_query + ro.n_push;
        }
        f[0] = nlq_feed{0, nreq};
        for (u32 k = 1; k < n_ops; k++) f[k] = nlq_feed{0, nq};
        return;
    }
    const RoundOps ro = j.round_ops[j.first_round + c];
    const u32 pop = ro.flags & 1;
    f[0] = nlq_feed{pop, pop ? ro.request : ro.request + 1};
    if (circuit_type == 7) {  // one request per cycle: four reads, two writes
        for (u32 k = 0; k < 6; k++) f[1 + k] = nlq_feed{1, ro.first_query + k};
    } else if (circuit_type == 5) {  // up to six reads, then the digest write
        const u32 write = (ro.flags >> 1) & 1, n_reads = ro.n_push - write;
        for (u32 k = 0; k < 6; k++) f[1 + k] = nlq_feed{k < n_reads ? 1u : 0u, ro.first_query + (k < n_reads ? k : n_reads)};
        f[7] = nlq_feed{write, ro.first_query + n_reads};
    } else if (circuit_type == 6) {
        f[1] = nlq_feed{1, ro.first_query};
        f[2] = nlq_feed{1, ro.first_query + 1};
        f[3] = nlq_feed{(ro.flags >> 1) & 1, ro.first_query + 2, ro.flags >> 8};  // aux: rounds left after this one
    } else {
        f[1] = nlq_feed{1, ro.first_query};
        f[2] = nlq_feed{ro.n_push > 1 ? 1u : 0u, ro.first_query + 1};
    }
}

struct NlqCellStore {  // cell k of a block whose first row (within the cycle's operations) is r0, for cycle c
    u64* trace; size_t n_rows; size_t row00; u32 capacity, g;
    __device__ __forceinline__ u64& at(u32 r0, u32 k) const {
        const u32 r = k >= 2 * g ? 2 : k >= g ? 1 : 0;  // (blocks have at most 130 cells, g >= 60)
        return trace[(size_t)(k - r * g) * n_rows + row00 + (size_t)(r0 + r) * capacity];
    }
};

// the netlist cell a linked component copies; has = false when the cell has no link in this cycle
__device__ __forceinline__ u64 nlq_linked_value(const nl_spec& S, const NlqFreeHome* __restrict__ fh, const u64* __restrict__ trace, size_t n_rows, u32 capacity,
                                                u32 c, const nlq_op& op, u32 cell, bool& has) {
    uint32_t cyc = 0, ref = 0;
    has = nlq_link_target(&op, c, capacity, cell, &cyc, &ref) != 0;
    if (!has) return 0;
    if (ref >= NL_REF_CYC && ref < NL_REF_FREE) return nl_home_cell(S, trace, n_rows, capacity, cyc, 0, ref);
    const NlqFreeHome h = fh[ref - NL_REF_FREE];
    return NLQ_TR(h.col, (size_t)cyc * S.rows_per_cycle + h.row);
}

// enc element e of an operation from a cell reader
template <class F>
__device__ __forceinline__ u64 nlq_enc_value(u32 item, u32 e, F&& cell) {
    const u32 n = nlq_enc_n_terms(item, e);
    u64 acc = 0;
    for (u32 i = 0; i < n; i++) {
        const nlq_term tm = nlq_enc_term(item, e, i);
        acc = gl::add(acc, gl::mul(gl::canon(cell(tm.cell)), 1ull << tm.shift));
    }
    return gl::canon(acc);
}

// the flattened Poseidon2 gate of the queue rows: p2::coop_flattened (poseidon2.cuh)
template <class Put>
__device__ __forceinline__ u64 nlq_coop_p2(const p2::Coop& co, u64 x, u32 g, Put&& put) { return p2::coop_flattened(co, x, g, put); }

// grid (ceil(capacity / 4), n_ops, instances), 64 lanes: a ROW of 16 lanes owns one operation of one cycle, the four rows of a wave
// four consecutive cycles of the same operation (uniform control flow; a store instruction writes 32-byte segments of <= 16 columns).
// Lane g computes what it needs itself — its cells of the ENC block (stride 16), the encoding elements its permutation inputs take —
// from the item record / the linked netlist cells; only the permutation crosses lanes. (The first version gave an operation to a LANE:
// its three dependent lane-serial permutations of ~65 us each were the whole kernel time.)
static __device__ __forceinline__ void k_nlq_fill(const VB& vb, const NlDev* __restrict__ devp, const NlqFreeHome* __restrict__ fh, const NlqFreeHome* __restrict__ lh, const nlq_desc& d, const NlqJob* __restrict__ jobs,
                                                        u32 capacity, size_t n_rows) {
    const nl_spec& S = devp->s;
    const NlqJob& job = jobs[vb.z];
    const u32 g = threadIdx.x & 15, c_raw = vb.x * 4 + (threadIdx.x >> 4), j = vb.y;
    const bool valid = c_raw < capacity;
    const u32 c = valid ? c_raw : capacity - 1;  // a row beyond the capacity recomputes the last cycle and stores nothing (DPP wants whole rows)
    u64* trace = job.trace;
    p2::Coop co;
    co.init((int)g);
    const nlq_op op = d.ops[j];
    const nlq_feed f = job.feed[(size_t)c * d.n_ops + j];
    const NlqQueueIn& Q = job.queues[op.queue];
    const u32 G = S.g, w = nlq_kind_width(op.kind), ncomp = nlq_item_comps(op.item), nenc = nlq_item_enc(op.item), r0 = nlq_op_row0(&d, G, j);
    const NlqCellStore st{trace, n_rows, (size_t)NLQ_BASE(&S, capacity) + 1 + c, capacity, G};
    const void* rec = f.en ? static_cast<const char*>(Q.items) + (size_t)f.idx * nlq_item_bytes(op.item) : nullptr;
    // the linked cells of the operation (64 nibbles / 32 bytes of a memory word's value) are fetched once, four independent loads per
    // lane, and staged in LDS: the encodings read every one of them, and a dependent table + cell load per term was the kernel's time
    __shared__ u64 sh_link[4][112];  // by cell index (the linkable cells of every item lie below 112)
    u64* my_link = sh_link[threadIdx.x >> 4];
    if (op.link != NLQ_LINK_NONE) {
#pragma unroll
        for (u32 q = 0; q < 7; q++) {
            const u32 k = g + 16 * q;
            if (k == 0 || k >= ncomp || !nlq_comp_linked(&op, k)) continue;
            // (the source cell inside this cycle, where the host could resolve it: the links that do not depend on the cycle)
            const NlqFreeHome h = k >= NLQ_MEM_NIBBLE0 && k < NLQ_MEM_NIBBLE0 + 64 ? lh[j * 64 + (k - NLQ_MEM_NIBBLE0)] : NlqFreeHome{0xFFFF, 0xFFFF};
            if (h.row != 0xFFFF) { my_link[k] = NLQ_TR(h.col, (size_t)c * S.rows_per_cycle + h.row); continue; }
            bool has = false;
            const u64 v = nlq_linked_value(S, fh, trace, n_rows, capacity, c, op, k, has);
            my_link[k] = has ? v : nlq_item_component(op.item, rec, k);  // no link in this cycle: the item's own field
        }
    }
    __syncthreads();
    auto comp0 = [&](u32 k) -> u64 { return nlq_comp_linked(&op, k) ? my_link[k] : nlq_item_component(op.item, rec, k); };
    auto comp = [&](u32 k) -> u64 {  // cell k >= 1 of the ENC block: a field of the item, the netlist cell it copies, or the recomposition of byte cells
        const int r = nlq_aux_of_cell(op.item, k);
        if (r < 0) return comp0(k);
        u64 acc = 0;
        for (u32 i = 0; i < nlq_aux_n_terms(op.item, (u32)r); i++) {
            const nlq_term tm = nlq_aux_term(op.item, (u32)r, i);
            acc = gl::add(acc, gl::mul(gl::canon(comp0(tm.cell)), 1ull << tm.shift));
        }
        return gl::canon(acc);
    };
    auto enc_val = [&](u32 e) -> u64 { return e < nenc ? nlq_enc_value(op.item, e, comp) : 0; };
    for (u32 k = g; k < ncomp; k += 16) {
        const u64 v = k ? comp(k) : (u64)(f.en ? 1 : 0);
        if (valid) st.at(r0, k) = v;
    }
    const u64 e_lo = enc_val(g), e_hi = enc_val(16 + g);  // enc[g], enc[16 + g]
    if (valid && g < nenc) st.at(r0, ncomp + g) = e_lo;
    if (valid && 16 + g < nenc) st.at(r0, ncomp + 16 + g) = e_hi;
    const u64* src = f.idx ? Q.states + (f.idx - 1) * w : Q.init;
    const u64 old = g < w ? src[g] : 0;
    auto put_at = [&](u32 pr0) { return [&st, pr0, valid](u32 k, u64 v) { if (valid) st.at(pr0, k) = v; }; };
    u64 out;
    if (op.kind != NLQ_POP4) {  // enc[0..8] ++ old[8..12]
        out = nlq_coop_p2(co, g < 8 ? e_lo : g < 12 ? old : 0, g, put_at(nlq_p2_row0(&d, G, j, 0)));
    } else {  // enc[0..8] ++ 0000 | enc[8..16] ++ capacity | enc[16..20] ++ old[0..4] ++ capacity
        out = nlq_coop_p2(co, g < 8 ? e_lo : 0, g, put_at(nlq_p2_row0(&d, G, j, 0)));
        const u64 e_mid = enc_val(8 + g);
        out = nlq_coop_p2(co, g < 8 ? e_mid : g < 12 ? out : 0, g, put_at(nlq_p2_row0(&d, G, j, 1)));
        const u64 old_up = (u64)__shfl((unsigned long long)old, (int)((threadIdx.x & 48) | ((g - 4) & 15)), 64);  // lane 4 + k takes old[k]
        out = nlq_coop_p2(co, g < 4 ? e_hi : g < 8 ? old_up : g < 12 ? out : 0, g, put_at(nlq_p2_row0(&d, G, j, 2)));
    }
    const u64 nw = f.en ? out : old;
    if (valid && g < w) {
        st.at(r0, ncomp + nenc + g) = old;
        st.at(r0, ncomp + nenc + w + g) = nw;
    }
    if (g < op.extra) {  // registers: a limb of the request the cycle is working on (the last one operation 0 popped)
        const nlq_feed f0 = job.feed[(size_t)c * d.n_ops];
        const NlqQueueIn& Q0 = job.queues[d.ops[0].queue];
        const long long cur = (long long)f0.idx - (f0.en ? 0 : 1);
        const void* rec0 = cur >= 0 && (u64)cur < Q0.n_items ? static_cast<const char*>(Q0.items) + (size_t)cur * nlq_item_bytes(d.ops[0].item) : nullptr;
        u64 v = 0;
        if (g >= 2) v = f.aux;  // a counter the feed supplies
        else for (u32 i = 0; i < 4; i++) v |= nlq_item_component(d.ops[0].item, rec0, op.reg_cell[g] + i) << (8 * i);
        if (valid) st.at(r0, ncomp + nenc + 2 * w + g) = v;
    }
    // QBND: the queue states before cycle 0 (written by the first operation on the queue) and after the last cycle (by the last)
    const size_t q0 = NLQ_BASE(&S, capacity);
    bool first = true, last = true;
    for (u32 i = 0; i < d.n_ops; i++)
        if (d.ops[i].queue == op.queue) { if (i < j) first = false; if (i > j) last = false; }
    if (valid && g < w) {
        if (c == 0 && first) NLQ_TR(nlq_bnd_col(&d, op.queue, 0, g), q0) = old;
        if (c + 1 == capacity && last) NLQ_TR(nlq_bnd_col(&d, op.queue, 1, g), q0) = nw;
    }
}

// ------------------------------------------------------------------------------------------------ checker (codes of oracle/netlist_queue.c)
__device__ __forceinline__ u64 nlq_cell_at(const nl_spec& S, const u64* __restrict__ trace, size_t n_rows, u32 capacity, u32 c, u32 r0, u32 k) {
    return NLQ_TR(k % S.g, NLQ_ROW(&S, capacity, r0 + k / S.g, c));
}
__device__ u64 nlq_prev_new(const nl_spec& S, const nlq_desc& d, const u64* __restrict__ trace, size_t n_rows, u32 capacity, u32 c, u32 j, u32 queue, u32 k) {
    for (;;) {
        while (j > 0) {
            j--;
            if (d.ops[j].queue == queue) return nlq_cell_at(S, trace, n_rows, capacity, c, nlq_op_row0(&d, S.g, j), nlq_new0(&d.ops[j]) + k);
        }
        if (c == 0) return NLQ_TR(nlq_bnd_col(&d, queue, 0, k), NLQ_BASE(&S, capacity));
        c--;
        j = d.n_ops;
    }
}

// grid (ceil(capacity / 64), n_ops): one lane per operation of a cycle; lane (0, 0, 0) also checks the QBND row
static __global__ __launch_bounds__(64) void k_nlq_check(const NlDev* __restrict__ devp, const NlqFreeHome* __restrict__ fh, nlq_desc d, nlq_rels rels, const u64* __restrict__ trace,
                                                         u32 capacity, size_t n_rows, CheckResult* res) {
    const nl_spec& S = devp->s;
    const u32 c = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
    const u32 G = S.g, p2rows = nlq_rows_for(NLQ_P2_CELLS, G);
    const size_t q0 = NLQ_BASE(&S, capacity);
    if (c == 0 && j == 0) {
        for (u32 q = 0; q < d.n_queues; q++) {
            bool ok = true;
            for (u32 k = 0; k < d.width[q]; k++)
                if (NLQ_TR(nlq_bnd_col(&d, q, 1, k), q0) != nlq_prev_new(S, d, trace, n_rows, capacity, capacity, 0, q, k)) ok = false;
            if (!ok) flag_bad(res, 4, 0x100 + q, q0);
        }
        for (u32 col = nlq_bnd_cells(&d); col < G; col++)
            if (NLQ_TR(col, q0)) { flag_bad(res, 6, col, q0); break; }
    }
    if (c >= capacity) return;
    if (j == 0)  // the relations between the operations of the cycle (include/zkw_netlist_queue.h nlq_rel)
        for (u32 i = 0; i < rels.n; i++) {
            const nlq_rel r = rels.r[i];
            if (r.prev && c == 0) continue;
            u64 en = r.gate == NLQ_REL_ACTIVE ? gl::canon(gl::sub(1, gl::canon(NLQ_TR(NL_HDR_IDLE, (size_t)c * S.rows_per_cycle)))) : gl::canon(nlq_cell_at(S, trace, n_rows, capacity, c, nlq_op_row0(&d, G, r.gate), 0));
            if (r.gate2 != NLQ_REL_CONST) en = gl::canon(gl::sub(en, gl::canon(nlq_cell_at(S, trace, n_rows, capacity, c, nlq_op_row0(&d, G, r.gate2), 0))));
            const u64 b = gl::canon(nlq_cell_at(S, trace, n_rows, capacity, c, nlq_op_row0(&d, G, r.op_b), r.cell_b));
            u64 a = 0;
            for (u32 k = 0; r.op_a != NLQ_REL_CONST && k < (r.span ? r.span : 1u); k++)  // (span: little-endian recomposition of byte cells)
                a = gl::canon(gl::add(a, gl::mul(gl::canon(nlq_cell_at(S, trace, n_rows, capacity, r.prev ? c - 1 : c, nlq_op_row0(&d, G, r.op_a), r.cell_a + k)), 1ull << (8 * k))));
            const u64 diff = gl::canon(gl::sub(gl::canon(r.prev == 3 ? gl::add(b, a) : gl::sub(b, a)), r.add >= 0 ? (u64)r.add : gl::P - (u64)(-r.add)));
            if (gl::canon(gl::mul(en, diff)) != 0) flag_bad(res, 7, 0x1000 + i, NLQ_ROW(&S, capacity, nlq_op_row0(&d, G, r.gate == NLQ_REL_ACTIVE ? r.op_b : r.gate), c));
        }
    const nlq_op op = d.ops[j];
    const u32 w = nlq_kind_width(op.kind), ncomp = nlq_item_comps(op.item), nenc = nlq_item_enc(op.item), r0 = nlq_op_row0(&d, G, j);
    const u32 ncells = nlq_enc_cells(&op), erows = nlq_rows_for(ncells, G);
    const u64 row_e = NLQ_ROW(&S, capacity, r0, c);
    auto cell = [&](u32 k) { return nlq_cell_at(S, trace, n_rows, capacity, c, r0, k); };
    const u64 en = cell(0);
    if (en > 1) flag_bad(res, 3, j, row_e);
    const size_t hdr_row = (size_t)c * S.rows_per_cycle;
    if (op.en_rule == NLQ_EN_RESET && en != NLQ_TR(NL_HDR_RESET, hdr_row)) flag_bad(res, 3, 0x100 + j, row_e);
    if (op.en_rule == NLQ_EN_ACTIVE && gl::canon(en) != gl::canon(gl::sub(1, gl::canon(NLQ_TR(NL_HDR_IDLE, hdr_row))))) flag_bad(res, 3, 0x100 + j, row_e);
    bool ok = true;
    for (u32 k = 1; k < ncomp; k++)
    {
        bool has = false;
        const u64 v = nlq_linked_value(S, fh, trace, n_rows, capacity, c, op, k, has);
        if (has && cell(k) != v) ok = false;
    }
    if (!ok) flag_bad(res, 2, j, row_e);
    // (no per-lane arrays with run-time indices — they would live in scratch memory: every value is read from its cell where it is used)
    for (u32 e = 0; e < nenc; e++)
        if (nlq_enc_value(op.item, e, cell) != gl::canon(cell(ncomp + e))) flag_bad(res, 7, 32 * j + e, row_e);
    for (u32 r = 0; r < nlq_aux_n(op.item); r++) {  // recomposition gates (a limb = its bytes)
        u64 acc = 0;
        for (u32 i = 0; i < nlq_aux_n_terms(op.item, r); i++) {
            const nlq_term tm = nlq_aux_term(op.item, r, i);
            acc = gl::add(acc, gl::mul(gl::canon(cell(tm.cell)), 1ull << tm.shift));
        }
        if (gl::canon(acc) != gl::canon(cell(nlq_aux_result(op.item, r)))) flag_bad(res, 7, 0x400 + 16 * j + r, row_e);
    }
    const u32 o0 = ncomp + nenc, n_perms = nlq_kind_perms(op.kind);
    auto want_in = [&](u32 p, u32 k) -> u64 {  // what input k of permutation p copies
        if (op.kind != NLQ_POP4) return k < 8 ? cell(ncomp + k) : cell(o0 + k);
        if (k >= 8) return p == 0 ? 0 : nlq_cell_at(S, trace, n_rows, capacity, c, nlq_p2_row0(&d, G, j, p - 1), NLQ_P2_CELLS - 12 + k);
        if (p < 2) return cell(ncomp + 8 * p + k);
        return k < 4 ? cell(ncomp + 16 + k) : cell(o0 + (k - 4));
    };
    for (u32 p = 0; p < n_perms; p++) {
        const u32 pr0 = nlq_p2_row0(&d, G, j, p);
        const u64 row_p = NLQ_ROW(&S, capacity, pr0, c);
        u64 s[12];
        ok = true;
#pragma unroll
        for (int k = 0; k < 12; k++) {
            s[k] = nlq_cell_at(S, trace, n_rows, capacity, c, pr0, k);
            if (s[k] != want_in(p, k)) ok = false;
            s[k] = gl::canon(s[k]);
        }
        if (!ok) flag_bad(res, 2, 0x1000 + 4 * j + p, row_p);
        // the flattened-gate relation: recompute and compare every variable
        ok = true;
        u32 pos = 12;
        p2::external(s);
        int r = 0;
        for (int k = 0; k < P2_HALF_FULL_ROUNDS; k++, r++) {
            p2::full_round(s, r);
#pragma unroll
            for (int i = 0; i < 12; i++) ok &= nlq_cell_at(S, trace, n_rows, capacity, c, pr0, pos + i) == gl::canon(s[i]);
            pos += 12;
        }
        for (int k = 0; k < P2_PARTIAL_ROUNDS; k++, r++) {
            s[0] = gl::pow7(gl::add_canon(s[0], p2::rc_at(12 * r)));
            ok &= nlq_cell_at(S, trace, n_rows, capacity, c, pr0, pos++) == gl::canon(s[0]);
            p2::internal(s);
        }
        for (int k = 0; k < P2_HALF_FULL_ROUNDS; k++, r++) {
            p2::full_round(s, r);
#pragma unroll
            for (int i = 0; i < 12; i++) ok &= nlq_cell_at(S, trace, n_rows, capacity, c, pr0, pos + i) == gl::canon(s[i]);
            pos += 12;
        }
        if (!ok) flag_bad(res, 8, 4 * j + p, row_p);
        for (u32 rr = 0; rr < p2rows; rr++)
            for (u32 col = (rr + 1 == p2rows ? NLQ_P2_CELLS - rr * G : G); col < G; col++)
                if (NLQ_TR(col, NLQ_ROW(&S, capacity, pr0 + rr, c))) { flag_bad(res, 6, col, NLQ_ROW(&S, capacity, pr0 + rr, c)); break; }
    }
    const u32 out_r0 = nlq_p2_row0(&d, G, j, n_perms - 1);
    ok = true;
    for (u32 k = 0; k < w; k++) {
        const u64 o = gl::canon(cell(o0 + k)), out_k = gl::canon(nlq_cell_at(S, trace, n_rows, capacity, c, out_r0, NLQ_P2_CELLS - 12 + k));
        const u64 want = gl::canon(gl::add(o, gl::mul(gl::canon(en), gl::canon(gl::sub(out_k, o)))));
        if (want != gl::canon(cell(o0 + w + k))) ok = false;
    }
    if (!ok) flag_bad(res, 7, 0x800 + j, row_e);
    ok = true;
    for (u32 k = 0; k < w; k++)
        if (cell(o0 + k) != nlq_prev_new(S, d, trace, n_rows, capacity, c, j, op.queue, k)) ok = false;
    if (!ok) flag_bad(res, 2, 0x2000 + j, row_e);
    for (u32 rr = 0; rr < erows; rr++)
        for (u32 col = (rr + 1 == erows ? ncells - rr * G : G); col < G; col++)
            if (NLQ_TR(col, NLQ_ROW(&S, capacity, r0 + rr, c))) { flag_bad(res, 6, col, NLQ_ROW(&S, capacity, r0 + rr, c)); break; }
}
#undef NLQ_TR

}  // namespace zkw
