// netlist_kernels.cuh — synthesis and satisfiability check of the netlist circuits in "zkw trace v4" (include/zkw_netlist.h,
// tools/netlist.py): Sha256RoundFunction (6), CodeDecommitter (3), Keccak256RoundFunction (5), L1MessagesHasher (13) on the
// REFERENCE's geometry and lookup-table sets (wrappers circuit_definitions/src/circuit_definitions/base_layer/
// {sha256_round_function,code_decommitter,keccak256_round_function,linear_hasher}.rs:28-39 + add_tables), ONE multiplicity
// column. One set of kernels for the four circuits, driven by the generated specs (include/zkw_*_circuit_spec.h); the
// circuit bodies are in the absent era-zkevm_circuits crate, so the placement is this library's own.
//
// Fill (k_nl_fill): a WAVE owns a cycle. The workgroup stages the netlists of all step types in LDS once (lanes index them
// with different values); a wave then walks its cycle's steps: the items of a step level by level with no workgroup barrier
// (a wave's LDS operations execute in order), every value a reference can name in ONE byte array per wave, and streams the
// step's rows out lane <-> row — all general-purpose and lookup cells of the rows, so nothing above the boundary needs a
// memset. Multiplicities are a pass of their own over 16-bit keys the fill leaves behind (k_nl_hist: a table's bins in LDS).
#pragma once
#include "../../include/zkw_netlist.h"
#include "netlist_eval.cuh"
#include "../../include/zkw_sha256_circuit_spec.h"
#include "../../include/zkw_code_decommitter_circuit_spec.h"
#include "../../include/zkw_keccak_circuit_spec.h"
#include "../../include/zkw_linear_hasher_circuit_spec.h"
#include "../../include/zkw_storage_application_circuit_spec.h"
#include "../../include/zkw_ecrecover_circuit_spec.h"
#include "../../include/zkw_types.h"
#include "ram_circuit_kernels.cuh"  // CheckResult, flag_bad

namespace zkw {

__device__ __forceinline__ void store_streaming(u64* cell, u64 v) { *cell = v; }

// host copies of the four specs (launch sizing, setup side) ...
NL_DEFINE_SPEC(h_sc, SC);
NL_DEFINE_SPEC(h_dc, DC);
NL_DEFINE_SPEC(h_kc, KC);
NL_DEFINE_SPEC(h_lh, LH);
NL_DEFINE_SPEC(h_sa, SA);
NL_DEFINE_SPEC(h_ek, EK);  // ECRecover (7): Keccak-f over the recovered public key; its EC section: ecrecover_kernels.cuh
static inline const nl_spec* nl_host_spec(int circuit_type) {
    switch (circuit_type) {
        case 6: return &h_sc_spec;
        case 3: return &h_dc_spec;
        case 5: return &h_kc_spec;
        case 13: return &h_lh_spec;
        case 10: return &h_sa_spec;
        case 7: return &h_ek_spec;
        default: return nullptr;
    }
}

// cycles of a trace of `capacity` units of the circuit's own counting (LinearHasher: messages; StorageApplication: tree queries)
static inline uint32_t nl_cycles_of(int circuit_type, uint32_t capacity) {
    return circuit_type == 13 ? ZKW_LINEAR_HASHER_CYCLES(capacity) : circuit_type == 10 ? capacity * SA_CYCLES_PER_WALK : capacity;
}
static inline bool nl_is_netlist(int circuit_type) { return nl_host_spec(circuit_type) != nullptr; }

// ---- what the kernels get: the spec with DEVICE pointers plus tables derived from it on the host
struct NlHistEntry { u32 step, r0, r1, key0, lookup_rows; };  // rows [r0, r1) of the lookup rows of cycle step `step` hold this table
struct NlDev {
    nl_spec s;                   // pointers are device pointers
    const uint16_t* cellmap;     // [step type: cell0 + col * rows + row] dense reference of the general-purpose cell, 0xFFFF = empty
    const u32* cell0;            // [step type]
    const u32* step_key0;        // [cycle step]: offset of the step's keys inside a cycle's keys
    u32 keys_per_cycle;
    const NlHistEntry* hist_entries;
    const u32* hist_first;       // [n_tables + 1]: table t's entries
    const u32* hist_slice0;      // [n_tables + 1]: table t owns histogram slices [slice0[t], slice0[t + 1])
    u32 n_hist_slices;
    // the gates' KNOWN cells run-length packed for the fill (a word = 8 consecutive values at shifts 4 apart is ONE entry):
    // {dense first reference, count - 1, shift of the first | 0x80 if negative, shift step}; per gate: [pk_first[g], pk_first[g + 1])
    const uint32_t* pk_terms;    // ref | (count - 1) << 16 | code << 20 | step << 28
    const uint16_t* pk_first;    // [n_gates + 1] (indices relative to the step type's pk0)
    const u32* pk0;              // [step type]
    u32 n_pk_terms;
    u32 max_items;               // max over step types of n_ops + n_gates + rows (the checker's grid)
    u32 vsize;                   // bytes of a wave's value array
    u32 fill_waves;              // waves per workgroup of k_nl_fill (8 or 16)
    u32 lds_bytes, lds_bytes16;  // dynamic LDS of k_nl_fill with 8 / 16 waves per workgroup
    // the lane-per-cycle path (k_nl_walk + k_nl_expand, circuits with thousands of short cycles): a resolved instruction stream
    // per step type (operands as (kind, index) with the value's LDS slot already looked up: slots are re-used once a value is dead),
    // where a step's state comes from, and what pass 2 needs to know about a row of a cycle
    const u32* prog;             // NL_I_* words
    const u32* prog0;            // [step type]
    const uint16_t* out_src;     // [step type][state]: NL_SRC_* encoded
    const struct NlRowMeta* rowmeta;  // [rows_per_cycle]
    u32 max_slots, walk_lds;     // LDS of k_nl_walk = (max_slots + 3 * state + max_free) * 64
};
struct NlRowMeta { u32 key_base; uint16_t lookup_rows; uint8_t rowend, flags /* 1: general cells written, 2: lookup row */, keyfmt /* n_in | in_bits << 4 */, _pad[3]; };
// operand sources of the instruction stream: kind << 13 | index
enum { NL_SRC_VAL = 0, NL_SRC_HDR = 1, NL_SRC_PREV = 2, NL_SRC_CYC = 3, NL_SRC_FREE = 4, NL_SRC_RC = 5, NL_SRC_IMM = 6 };
// instructions: LOOKUP w0 = 1 | fn << 4 | param << 8 | n_in << 12 | n_out << 14, w1 = src0 | src1 << 16, w2 = src2 | dst0 << 16,
//   w3 = dst1 | dst2 << 16, w4 = row << 16 | first column;  HINT w0 = 2 | lo_a << 4 | n_a << 8 | lo_b << 12 | n_b << 16,
//   w1 = src_a | src_b << 16, w2 = dst;  GATE w0 = 3 | n_known << 4 | n_new << 12 | mask_last << 20 | new_step << 24, w1 = constant,
//   w2 = row << 16 | first column, w3 = shift of the first NEW cell, then n_known x (src | code << 16), n_new x dst;
//   LATE w0 = 4, w1 = src, w2 = row << 16 | column (a gate's late cell, written once its producer has run);  END 0. dst 0xFFFF: not kept
enum { NL_I_END = 0, NL_I_LOOKUP = 1, NL_I_HINT = 2, NL_I_GATE = 3, NL_I_LATE = 4 };
struct NlJob {
    const uint8_t* hdr_bits;      // [capacity]: bit 0 reset, bit 1 idle
    const uint8_t* free_elems;    // [capacity][free_per_cycle]
    const uint8_t* state_before;  // [capacity + 1][state]
    const u64* public_input;      // [4]
    u64* trace;                   // [cols][n_rows]
    uint16_t* keys;               // [capacity][keys_per_cycle]
    u32* hist;                    // [n_hist_slices][2 halves][NL_HIST_HALF]: k_nl_hist's bins (every bin stored), summed by k_nl_finish
};

constexpr int NL_FILL_WAVES_MAX = 16;  // waves (= cycles in flight) per workgroup: 16 where two such workgroups fit a CU's LDS, else 8 (nl_get)

// layout of a wave's value array: [values | header 4 | prev state | cycle state | free | rc 8 | constants 256]
struct NlV {
    u32 hdr, prev, cyc, fre, rc, con, size;
    __host__ __device__ NlV(const nl_spec& s) {
        hdr = s.max_values;
        prev = hdr + 4;
        cyc = prev + s.state;
        fre = cyc + s.state;
        rc = fre + s.max_free;
        con = rc + 8;
        size = (con + 256 + 15) & ~15u;
    }
    __host__ __device__ uint16_t dense(u32 ref) const {
        return (uint16_t)(ref < NL_REF_HDR ? ref : ref < NL_REF_PREV ? hdr + (ref - NL_REF_HDR) : ref < NL_REF_CYC ? prev + (ref - NL_REF_PREV)
                          : ref < NL_REF_FREE ? cyc + (ref - NL_REF_CYC) : ref < NL_REF_RC ? fre + (ref - NL_REF_FREE)
                          : ref < NL_REF_CONST ? rc + (ref - NL_REF_RC) : con + (ref - NL_REF_CONST));
    }
};

struct NlLdsGate { u32 constant; uint16_t new_ref; uint8_t n_new, new_sh0, new_step, mask_last, _pad[2]; };  // NEW cells: consecutive values at shifts sh0 + i * step; mask_last: the gate has late cells, its last NEW cell is a digit like the others
// LDS carve of k_nl_fill, the same arithmetic on the host (lds_bytes) and in the kernel
struct NlLds {
    u32 tab, types, cyc, op_table, op_in, op_out, gates, term_ref, term_code, pk, pk_first, hints, order, level, out, waves, total;
    __host__ __device__ NlLds(const nl_spec& s, u32 vsize, u32 n_pk, u32 n_waves) {
        u32 at = 0;
        auto take = [&](u32 bytes) { u32 r = at; at = (at + bytes + 15) & ~15u; return r; };
        tab = take(s.n_tables * sizeof(nl_table));
        types = take(s.n_step_types * sizeof(nl_step_type));
        cyc = take(s.steps_per_cycle * sizeof(nl_cycle_step));
        op_table = take(s.n_ops * 2);  // (16-bit: the ECRecover table set has 262 tables)
        op_in = take(s.n_ops * 6);
        op_out = take(s.n_ops * 2);
        gates = take(s.n_gates * sizeof(NlLdsGate));
        term_ref = term_code = at;  // (the NEW cells of a gate are described by its record)
        pk = take(n_pk * 4);
        pk_first = take((s.n_gates + s.n_step_types) * 2);
        hints = take((s.n_hints ? s.n_hints : 1) * sizeof(nl_hint));
        order = take(s.n_order * 2);
        level = take(s.n_level_starts * 2);
        out = take(s.n_step_types * s.state * 2);
        waves = take(n_waves * vsize);
        total = at + 16;  // (gate operands are read 8 bytes at a time: the last wave's last cells have something behind them)
    }
};

#define NL_TR(col, row) job.trace[(size_t)(col) * n_rows + (size_t)(row)]
#define NL_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

__device__ __forceinline__ void nl_eval_sel(u32 fn, u32 k, u32 a0, u32 a1, u32 a2, u32& o0, u32& o1, u32& o2) {
    // selects, not branches: the lanes of a level hold lookups of several tables
    const u32 lo = a0 & ((1u << k) - 1), hi = a0 >> k;
    u32 r0 = a0 ^ a1;                                       // XOR8
    r0 = fn == NL_FN_AND8 ? (a0 & a1) : r0;
    r0 = fn == NL_FN_TRIXOR4 ? (a0 ^ a1 ^ a2) : r0;
    r0 = fn == NL_FN_CH4 ? ((a0 & a1) ^ (~a0 & a2 & 15u)) : r0;
    r0 = fn == NL_FN_MAJ4 ? ((a0 & a1) ^ (a0 & a2) ^ (a1 & a2)) : r0;
    const bool split = fn == NL_FN_BYTESPLIT || fn == NL_FN_SPLIT4;
    o0 = split ? lo : r0;
    o1 = split ? hi : 0;
    o2 = fn == NL_FN_SPLIT4 ? ((lo << (4 - k)) | hi) : 0;
}

// WAVES = waves per workgroup, a wave owns a cycle. (Two cycles per wave — each half of a wave walking its own cycle, since a
// level of the SHA-256 netlist holds 15 items on average and most lanes idle through the walk — was measured: 4.19 ms against
// 3.35 ms per 8 SHA-256 instances. The walk is bound by the latency of its dependent LDS reads, which two waves per SIMD hide
// from each other and one wave per SIMD does not; CPW below is what is left of that experiment.)
template <int W, int R, int WAVES>
static __device__ __forceinline__ void k_nl_fill(const VB& vb, const NlDev* __restrict__ devp, const NlJob* __restrict__ jobs, u32 capacity, size_t n_rows, u32 probe) {
    constexpr int CPW = 1, NL_FILL_WAVES = WAVES, NL_FILL_THREADS = 64 * WAVES;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    constexpr u32 LW = 64 / CPW;                             // lanes that share a cycle
    const NlDev& D = *devp;
    const nl_spec& S = D.s;
    const NlJob job = jobs[vb.y];
    const int t = threadIdx.x, wv = t >> 6;
    const u32 lane = (u32)(t & 63) % LW, sub = (u32)(t & 63) / LW;
    const NlV V(S);
    const NlLds L(S, V.size, D.n_pk_terms, WAVES);
    nl_table* const s_tab = reinterpret_cast<nl_table*>(lds + L.tab);
    nl_step_type* const s_types = reinterpret_cast<nl_step_type*>(lds + L.types);
    nl_cycle_step* const s_cyc = reinterpret_cast<nl_cycle_step*>(lds + L.cyc);
    uint16_t* const s_op_table = reinterpret_cast<uint16_t*>(lds + L.op_table);
    uint16_t* const s_op_in = reinterpret_cast<uint16_t*>(lds + L.op_in);  // [3][n_ops]
    uint16_t* const s_op_out = reinterpret_cast<uint16_t*>(lds + L.op_out);
    NlLdsGate* const s_gates = reinterpret_cast<NlLdsGate*>(lds + L.gates);
    u32* const s_pk = reinterpret_cast<u32*>(lds + L.pk);
    uint16_t* const s_pk_first = reinterpret_cast<uint16_t*>(lds + L.pk_first);   // [step type: gate0 + type index + gate .. + n_gates]
    nl_hint* const s_hints = reinterpret_cast<nl_hint*>(lds + L.hints);
    uint16_t* const s_order = reinterpret_cast<uint16_t*>(lds + L.order);
    uint16_t* const s_level = reinterpret_cast<uint16_t*>(lds + L.level);
    uint16_t* const s_out = reinterpret_cast<uint16_t*>(lds + L.out);
    uint8_t* const val = lds + L.waves + (wv * CPW + sub) * V.size;
    // ---- stage the netlists (references made dense: one LDS read resolves any of them)
    for (u32 i = t; i < S.n_tables; i += NL_FILL_THREADS) s_tab[i] = S.tables[i];
    for (u32 i = t; i < S.n_step_types; i += NL_FILL_THREADS) s_types[i] = S.step_types[i];
    for (u32 i = t; i < S.steps_per_cycle; i += NL_FILL_THREADS) s_cyc[i] = S.cycle[i];
    for (u32 i = t; i < S.n_ops; i += NL_FILL_THREADS) {
        const nl_op op = S.ops[i];
        s_op_table[i] = op.table;
        for (int k = 0; k < 3; k++) s_op_in[k * S.n_ops + i] = V.dense(op.in[k]);
        s_op_out[i] = op.out;
    }
    for (u32 i = t; i < S.n_gates; i += NL_FILL_THREADS) {
        const nl_gate g = S.gates[i];
        u32 ty = 0;
        while (ty + 1 < S.n_step_types && S.step_types[ty + 1].gate0 <= i) ty++;
        const nl_term* tm = S.terms + S.step_types[ty].term0 + g.first_term + g.n_known;
        NlLdsGate lg;
        lg.constant = g.constant;
        lg.n_new = (uint8_t)g.n_new;
        lg.new_ref = g.n_new ? V.dense(tm[0].ref) : 0;
        lg.new_sh0 = g.n_new ? (uint8_t)(tm[0].code & 0x7F) : 0;
        lg.new_step = g.n_new > 1 ? (uint8_t)((tm[1].code & 0x7F) - (tm[0].code & 0x7F)) : 0;  // (nl_get checked that the NEW cells are evenly spaced)
        lg.mask_last = 0;
        for (u32 k = 0; k < g.n_known; k++) lg.mask_last |= (tm[(int)k - (int)g.n_known].code & NL_TERM_LATE) ? 1 : 0;
        lg._pad[0] = lg._pad[1] = 0;
        s_gates[i] = lg;
    }
    for (u32 i = t; i < D.n_pk_terms; i += NL_FILL_THREADS) s_pk[i] = D.pk_terms[i];
    for (u32 i = t; i < S.n_gates + S.n_step_types; i += NL_FILL_THREADS) s_pk_first[i] = D.pk_first[i];
    for (u32 i = t; i < S.n_hints; i += NL_FILL_THREADS) {
        nl_hint h = S.hints[i];
        h.ref_a = V.dense(h.ref_a); h.ref_b = V.dense(h.ref_b);
        s_hints[i] = h;
    }
    for (u32 i = t; i < S.n_order; i += NL_FILL_THREADS) s_order[i] = S.order[i];
    for (u32 i = t; i < S.n_level_starts; i += NL_FILL_THREADS) s_level[i] = S.level_start[i];
    for (u32 i = t; i < S.n_step_types * S.state; i += NL_FILL_THREADS) s_out[i] = V.dense(S.out[i]);
    for (u32 i = t; i < NL_FILL_WAVES * 256; i += NL_FILL_THREADS) lds[L.waves + (i >> 8) * V.size + V.con + (i & 255)] = (uint8_t)i;
    __syncthreads();  // the only workgroup barrier: from here on every wave is on its own
    // everything the cycle loop needs from the spec, in registers: the trace stores below go through a pointer the compiler cannot
    // prove distinct from *devp, so every `S.field` inside the loops would be re-read from global memory after each store
    const u32 m_n_ops = S.n_ops, G = S.g, STATE = S.state, STEPS = S.steps_per_cycle, RPC = S.rows_per_cycle, FPC = S.free_per_cycle;
    const int m0a = S.masks[0], m0b = S.masks[1], m1a = S.masks[2], m1b = S.masks[3];
    const uint16_t* const g_cellmap = D.cellmap;
    const u32* const g_cell0 = D.cell0;
    const u32* const g_key0 = D.step_key0;
    const u32* const g_pk0 = D.pk0;
    const u32 keys_per_cycle = D.keys_per_cycle;
    for (u32 c0 = (vb.x * (NL_FILL_WAVES / CPW) + wv) * CPW; c0 < capacity; c0 += vb.nx * NL_FILL_WAVES) {
        NL_WAVE_SYNC();  // the previous cycle's write phase has read everything it needs
        const bool live = c0 + sub < capacity;         // (the last pair of a trace may have one cycle only: its half walks along, stores nothing)
        const u32 c = live ? c0 + sub : c0;
        const u32 bits = job.hdr_bits[c], reset = bits & 1, idle = (bits >> 1) & 1;
        if (lane == 0) {
            val[V.hdr + 0] = (uint8_t)reset; val[V.hdr + 1] = (uint8_t)idle;
            val[V.hdr + 2] = (uint8_t)(m0a + m0b * (int)reset);
            val[V.hdr + 3] = (uint8_t)(m1a + m1b * (int)idle);
        }
        for (u32 k = lane; k < STATE; k += LW) {
            const uint8_t x = job.state_before[(size_t)c * STATE + k];
            val[V.cyc + k] = x;
            val[V.prev + k] = x;
        }
        u32 free_at = 0;
        for (u32 s = 0; s < STEPS; s++) {
            const u32 type = s_cyc[s].type, row0 = s_cyc[s].row0;
            const nl_step_type T = s_types[type];
            const size_t base = (size_t)c * RPC + row0;
            const u32 pk0_t = g_pk0[type];  // (a global read: once per step, not once per gate)
            for (u32 k = lane; k < T.n_free; k += LW) val[V.fre + k] = job.free_elems[(size_t)c * FPC + free_at + k];
            if (lane < 8) val[V.rc + lane] = s_cyc[s].rc[lane];  // (indexing a register copy by lane would put it in scratch memory)
            free_at += T.n_free;
            NL_WAVE_SYNC();
            const uint16_t* const lvl = s_level + T.level0;
            u32 lv0 = lvl[0];
            for (u32 l = 0; l < ((probe & 1) ? 0u : T.n_levels); l++) {  // probe bit 0: skip the level walk (measurement only)
                const u32 lv1 = lvl[l + 1];
                for (u32 e = lv0 + lane; e < lv1; e += LW) {
                    const u32 it = s_order[T.order0 + e];
                    if (it < NL_ORDER_GATE) {
                        u32 j = T.op0 + it, a0;
                        if (it >= NL_ORDER_FUSED) {  // a hint and the lookup it keys: one item, one level
                            const nl_hint h = s_hints[T.hint0 + (it - NL_ORDER_FUSED)];
                            const u32 a = val[h.ref_a], b = val[h.ref_b];
                            a0 = ((a >> h.lo_a) & ((1u << h.n_a) - 1)) | (((b >> h.lo_b) & ((1u << h.n_b) - 1)) << h.n_a);
                            val[h.value] = (uint8_t)a0;
                            j = T.op0 + h.fused_slot;
                        } else {
                            a0 = val[s_op_in[j]];
                        }
                        const nl_table tb = s_tab[s_op_table[j] - 1];
                        const u32 out = s_op_out[j];
                        u32 o0, o1, o2;
                        nl_eval_sel(tb.fn, tb.param, a0, val[s_op_in[m_n_ops + j]], val[s_op_in[2 * m_n_ops + j]], o0, o1, o2);
                        if (out != 0xFFFF) {
                            val[out] = (uint8_t)o0;
                            if (tb.n_out > 1) val[out + 1] = (uint8_t)o1;
                            if (tb.n_out > 2) val[out + 2] = (uint8_t)o2;
                        }
                    } else if (it < NL_ORDER_HINT) {
                        const u32 gi = T.gate0 + (it - NL_ORDER_GATE);
                        const NlLdsGate g = s_gates[gi];
                        const uint16_t* const pf = s_pk_first + gi + type;  // (one extra entry per step type closes its last gate)
                        const u32 p0 = pk0_t + pf[0], p1 = pk0_t + pf[1];
                        long long sum = g.constant;
                        // the known cells, run-length packed: a word operand (8 nibbles) is one entry
                        for (u32 p = p0; p < p1; p++) {
                            const u32 w = s_pk[p], ref = w & 0xFFFF, cnt = (w >> 16) & 15, code = (w >> 20) & 0xFF, step = w >> 28;
                            // the run's cells in ONE 8-byte LDS read (any alignment), then packed: nibble runs (step 4) by three
                            // shift-or-mask stages, byte runs as they are (cells of a run are < 2^step, or the run is one cell)
                            u64 raw;
                            __builtin_memcpy(&raw, val + ref, 8);
                            raw = cnt >= 7 ? raw : raw & ((1ull << (8 * (cnt + 1))) - 1);
                            u64 x4 = (raw | (raw >> 4)) & 0x00FF00FF00FF00FFull;
                            x4 = (x4 | (x4 >> 8)) & 0x0000FFFF0000FFFFull;
                            x4 = (x4 | (x4 >> 16)) & 0xFFFFFFFFull;
                            u64 part = step == 4 ? x4 : raw;
                            if (step != 4 && step != 8 && cnt != 0) {  // (no circuit has such runs today)
                                part = 0;
                                for (u32 k = 0; k <= cnt; k++) part |= ((raw >> (8 * k)) & 0xFF) << (k * step);
                            }
                            const long long v = (long long)(part << (code & 0x7F));
                            sum += (code & 0x80) ? -v : v;
                        }
                        if (g.new_step == 4 && g.n_new >= 7 && g.n_new <= 9) {
                            // the NEW cells of a word: 7 - 9 consecutive values, nibble digits and a last digit that takes what is left
                            const u64 v = (u64)sum >> g.new_sh0;
                            u64 x = v & 0xFFFFFFFFull;
                            x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
                            x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
                            x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
                            const u32 last = (u32)(uint8_t)(v >> (4 * (g.n_new - 1))) & (g.mask_last ? 15u : 255u);
                            if (g.n_new == 9) {
                                __builtin_memcpy(val + g.new_ref, &x, 8);
                                val[g.new_ref + 8] = (uint8_t)last;
                            } else if (g.n_new == 8) {
                                x = (x & 0x00FFFFFFFFFFFFFFull) | ((u64)last << 56);
                                __builtin_memcpy(val + g.new_ref, &x, 8);
                            } else {
                                x = (x & 0x0000FFFFFFFFFFFFull) | ((u64)last << 48);
                                const u32 lo = (u32)x;
                                const uint16_t mid = (uint16_t)(x >> 32);
                                __builtin_memcpy(val + g.new_ref, &lo, 4);
                                __builtin_memcpy(val + g.new_ref + 4, &mid, 2);
                                val[g.new_ref + 6] = (uint8_t)(x >> 48);
                            }
                        } else {
                            for (u32 i = 0; i < g.n_new; i++) {
                                u64 x = (u64)sum >> (g.new_sh0 + i * g.new_step);
                                if (i + 1 < g.n_new || g.mask_last) x &= (1ull << g.new_step) - 1;
                                val[g.new_ref + i] = (uint8_t)x;
                            }
                        }
                    } else {
                        const nl_hint h = s_hints[T.hint0 + (it - NL_ORDER_HINT)];
                        const u32 a = val[h.ref_a], b = val[h.ref_b];
                        val[h.value] = (uint8_t)(((a >> h.lo_a) & ((1u << h.n_a) - 1)) | (((b >> h.lo_b) & ((1u << h.n_b) - 1)) << h.n_a));
                    }
                }
                lv0 = lv1;
                NL_WAVE_SYNC();
            }
            // ---- stream the step's rows out, lane <-> row: every general-purpose and lookup cell
            const uint16_t* const cmap = g_cellmap + g_cell0[type];
            uint16_t* const keys = job.keys + (size_t)c * keys_per_cycle + g_key0[s];
            for (u32 r = lane; r < ((probe & 2) || !live ? 0u : T.rows); r += LW) {  // probe bit 1: skip the streaming (measurement only)
                const size_t row = base + r;
                if (r <= T.gate_rows && !(probe & 4)) {  // rows below the gate rows hold no general-purpose cell: zero already (nl_synthesize)
                    // the cell map is read in batches of independent loads
                    for (u32 col0 = 0; col0 < G; col0 += 8) {
                        uint16_t ref[8];
#pragma unroll
                        for (int k = 0; k < 8; k++) ref[k] = col0 + k < G ? cmap[(size_t)(col0 + k) * T.rows + r] : (uint16_t)0xFFFF;
#pragma unroll
                        for (int k = 0; k < 8; k++)
                            if (col0 + k < G) store_streaming(&NL_TR(col0 + k, row), ref[k] == 0xFFFF ? 0 : (u64)val[ref[k]]);
                    }
                }
                if (probe & 8) continue;
                if (r == 0 || r > T.lookup_rows) {
                    for (int k = 0; k < W * R; k++) store_streaming(&NL_TR(G + k, row), 0);
                    continue;
                }
#pragma unroll 1
                for (int sl = 0; sl < R; sl++) {
                    const u32 j = T.op0 + (r - 1) * R + sl;
                    const nl_table tb = s_tab[s_op_table[j] - 1];
                    const u32 a0 = val[s_op_in[j]], a1 = val[s_op_in[m_n_ops + j]], a2 = val[s_op_in[2 * m_n_ops + j]];
                    u32 o0, o1, o2;
                    nl_eval_sel(tb.fn, tb.param, a0, a1, a2, o0, o1, o2);
                    // cell k of the slot: input k, or output k - n_in, or zero padding — selects only (a local array indexed by a
                    // runtime value would live in scratch memory)
#pragma unroll
                    for (int k = 0; k < W; k++) {
                        const int jj = k - (int)tb.n_in;
                        const u32 vin = k == 0 ? a0 : k == 1 ? a1 : a2;
                        const u32 vout = jj == 0 ? o0 : jj == 1 ? o1 : o2;
                        const u64 v = jj < 0 ? vin : (jj < (int)tb.n_out ? vout : 0u);
                        store_streaming(&NL_TR(G + W * sl + k, row), v);
                    }
                    const u32 key = a0 | (tb.n_in > 1 ? a1 << tb.in_bits : 0u) | (tb.n_in > 2 ? a2 << (2 * tb.in_bits) : 0u);
                    keys[(size_t)sl * T.lookup_rows + (r - 1)] = (uint16_t)key;  // lanes <-> rows: contiguous runs per slot
                }
            }
            // the state this step leaves becomes the next step's PREV bank (through registers: the banks overlap in time)
            const uint16_t* const so = s_out + type * STATE;
            uint8_t nx[256 / LW];  // state <= 256 elements (constant indices after unrolling: registers)
#pragma unroll
            for (u32 k = 0; k < 256 / LW; k++) nx[k] = lane + k * LW < STATE ? val[so[lane + k * LW]] : (uint8_t)0;
            NL_WAVE_SYNC();
#pragma unroll
            for (u32 k = 0; k < 256 / LW; k++)
                if (lane + k * LW < STATE) val[V.prev + lane + k * LW] = nx[k];
        }
    }
}

// ------------------------------------------------------------------------------------------------ lane-per-cycle fill
// For circuits with thousands of short cycles (SHA-256 family, StorageApplication) the level walk of k_nl_fill — a wave per
// cycle, ~15 items per level, an LDS round trip per level — is latency-bound. Here a LANE owns a cycle: the 64 lanes of a wave
// run the same resolved instruction stream on 64 cycles, no level structure, no cross-lane traffic; values live in LDS as
// [slot][lane] (64 consecutive bytes per access: conflict-free), slots re-used once a value is dead. The cells of 64 cycles
// cannot be stored coalesced from that shape (a trace is cycle-major: lane <-> cycle is a stride of rows_per_cycle rows), so pass 1
// writes them as BYTES into a scratch tile [column][row of the cycle][64 cycles] (coalesced, an eighth of the trace), and pass 2
// (k_nl_expand) transposes 64 rows x 64 cycles of a column through LDS and stores u64 cells lane <-> row, 512 contiguous bytes
// per cycle — plus the 16-bit keys of the lookups in the layout k_nl_hist reads.
// The instruction stream and the state sources are KERNEL ARGUMENTS (known global address space, uniform index: scalar loads,
// which wait on lgkmcnt only). A vector load inside the item loop would make the wave wait for every outstanding global store
// of the loop (gfx9 returns loads and stores through one in-order counter): measured 1.5 us per item with the pointers taken
// from the NlDev struct (generic address space: flat loads).
// (every LDS operand is an OFFSET into the kernel's one shared array: with pointers the compiler cannot tell the address space
// and emits flat loads, which wait for every outstanding global store — measured: 1.5 us per item)
#define NL_WALK_GET(src) nl_walk_get(lds, (src), o_vals, o_prev, o_cyc, o_fr, rc_lo, rc_hi, h0, h1, h2, h3, lane)
__device__ __forceinline__ u32 nl_walk_get(const uint8_t* lds, u32 src, u32 o_vals, u32 o_prev, u32 o_cyc, u32 o_fr, u32 rc_lo, u32 rc_hi,
                                           u32 h0, u32 h1, u32 h2, u32 h3, u32 lane) {
    const u32 kind = src >> 13, idx = src & 0x1FFF;  // uniform across the wave: scalar branches
    switch (kind) {
        case NL_SRC_VAL: return lds[o_vals + idx * 64 + lane];
        case NL_SRC_HDR: return idx == 0 ? h0 : idx == 1 ? h1 : idx == 2 ? h2 : h3;
        case NL_SRC_PREV: return lds[o_prev + idx * 64 + lane];
        case NL_SRC_CYC: return lds[o_cyc + idx * 64 + lane];
        case NL_SRC_FREE: return lds[o_fr + idx * 64 + lane];
        case NL_SRC_RC: return ((idx < 4 ? rc_lo : rc_hi) >> (8 * (idx & 3))) & 255u;
        default: return idx;
    }
}

template <int W, int R>
static __device__ __forceinline__ void k_nl_walk(const VB& vb, const NlDev* __restrict__ devp, const NlJob* __restrict__ jobs, u32 capacity,
                                                       uint8_t* __restrict__ scratch, size_t scratch_per_job, const u32* __restrict__ prog_g,
                                                       const u32* __restrict__ prog0, const uint16_t* __restrict__ out_src_g) {
    // the constant address space: never clobbered, so a uniform index is a scalar load whatever the loop stores
    typedef const __attribute__((address_space(4))) u32* c_u32;
    typedef const __attribute__((address_space(4))) uint16_t* c_u16;
    const c_u32 prog = (c_u32)prog_g;
    const c_u16 out_src = (c_u16)out_src_g;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const NlDev& D = *devp;
    const nl_spec& S = D.s;
    const NlJob job = jobs[vb.y];
    const u32 lane = threadIdx.x, tile = vb.x;
    const u32 STATE = S.state, RPC = S.rows_per_cycle, STEPS = S.steps_per_cycle, FPC = S.free_per_cycle, G = S.g;
    const u32 c = min(tile * 64 + lane, capacity - 1);  // (the lanes beyond the last cycle walk it again; pass 2 ignores them)
    const u32 o_vals = 0, o_cyc = D.max_slots * 64 + 2 * STATE * 64, o_fr = o_cyc + STATE * 64;  // [slot][lane] | prev | next | cyc | free
    u32 o_prev = D.max_slots * 64, o_next = o_prev + STATE * 64;
    uint8_t* const sc = scratch + vb.y * scratch_per_job + (size_t)tile * ((size_t)S.mult_col * RPC * 64) + lane;
#define NL_SC(col, row) sc[((size_t)(col) * RPC + (row)) * 64]
    const u32 bits = job.hdr_bits[c], h0 = bits & 1, h1 = (bits >> 1) & 1;
    const u32 h2 = (u32)(uint8_t)(S.masks[0] + S.masks[1] * (int)h0), h3 = (u32)(uint8_t)(S.masks[2] + S.masks[3] * (int)h1);
    for (u32 k0 = 0; k0 < STATE; k0 += 16) {  // 16 independent loads in flight
        uint8_t x[16];
#pragma unroll
        for (int k = 0; k < 16; k++) x[k] = k0 + k < STATE ? job.state_before[(size_t)c * STATE + k0 + k] : (uint8_t)0;
#pragma unroll
        for (int k = 0; k < 16; k++)
            if (k0 + k < STATE) { lds[o_cyc + (k0 + k) * 64 + lane] = x[k]; lds[o_prev + (k0 + k) * 64 + lane] = x[k]; }
    }
    u32 free_at = 0;
    for (u32 s = 0; s < STEPS; s++) {
        // (loaded through generic pointers of the spec struct, i.e. into vector registers: made scalars again, or everything derived
        // from them — the program counter first of all — would count as divergent and be fetched with vector loads)
        const nl_cycle_step cs = S.cycle[s];
        const u32 type = (u32)__builtin_amdgcn_readfirstlane((int)cs.type), row0 = (u32)__builtin_amdgcn_readfirstlane((int)cs.row0);
        const u32 n_free = (u32)__builtin_amdgcn_readfirstlane((int)S.step_types[type].n_free);
        const uint8_t* const frg = job.free_elems + (size_t)c * FPC + free_at;
        free_at += n_free;
        for (u32 f0 = 0; f0 < n_free; f0 += 16) {  // 16 independent loads in flight (a lane's elements are consecutive bytes)
            uint8_t x[16];
#pragma unroll
            for (int k = 0; k < 16; k++) x[k] = f0 + k < n_free ? frg[f0 + k] : (uint8_t)0;
#pragma unroll
            for (int k = 0; k < 16; k++)
                if (f0 + k < n_free) lds[o_fr + (f0 + k) * 64 + lane] = x[k];
        }
        u32 rc_lo, rc_hi;
        __builtin_memcpy(&rc_lo, cs.rc, 4);
        __builtin_memcpy(&rc_hi, cs.rc + 4, 4);
        rc_lo = (u32)__builtin_amdgcn_readfirstlane((int)rc_lo);
        rc_hi = (u32)__builtin_amdgcn_readfirstlane((int)rc_hi);
        NL_SC(0, row0) = (uint8_t)h0; NL_SC(1, row0) = (uint8_t)h1; NL_SC(2, row0) = (uint8_t)h2; NL_SC(3, row0) = (uint8_t)h3;
        u32 pc = (u32)__builtin_amdgcn_readfirstlane((int)prog0[type]);
        for (;;) {
            const u32 w0 = prog[pc];
            const u32 opc = w0 & 15;
            if (opc == NL_I_END) break;
            if (opc == NL_I_LOOKUP) {
                const u32 w1 = prog[pc + 1], w2 = prog[pc + 2], w3 = prog[pc + 3], w4 = prog[pc + 4];
                pc += 5;
                const u32 fn = (w0 >> 4) & 15, param = (w0 >> 8) & 15, n_in = (w0 >> 12) & 3, n_out = (w0 >> 14) & 3;
                const u32 a0 = NL_WALK_GET(w1 & 0xFFFF);
                const u32 a1 = n_in > 1 ? NL_WALK_GET(w1 >> 16) : 0u;
                const u32 a2 = n_in > 2 ? NL_WALK_GET(w2 & 0xFFFF) : 0u;
                u32 o0, o1, o2;
                nl_eval_sel(fn, param, a0, a1, a2, o0, o1, o2);
                const u32 row = row0 + (w4 >> 16), col = w4 & 0xFFFF;
#pragma unroll
                for (int k = 0; k < W; k++) {
                    const int jj = k - (int)n_in;
                    const u32 vin = k == 0 ? a0 : k == 1 ? a1 : a2;
                    const u32 vout = jj == 0 ? o0 : jj == 1 ? o1 : o2;
                    NL_SC(col + k, row) = (uint8_t)(jj < 0 ? vin : (jj < (int)n_out ? vout : 0u));
                }
                const u32 d0 = w2 >> 16, d1 = w3 & 0xFFFF, d2 = w3 >> 16;
                if (d0 != 0xFFFF) lds[o_vals + d0 * 64 + lane] = (uint8_t)o0;
                if (d1 != 0xFFFF) lds[o_vals + d1 * 64 + lane] = (uint8_t)o1;
                if (d2 != 0xFFFF) lds[o_vals + d2 * 64 + lane] = (uint8_t)o2;
            } else if (opc == NL_I_HINT) {
                const u32 w1 = prog[pc + 1], w2 = prog[pc + 2];
                pc += 3;
                const u32 a = NL_WALK_GET(w1 & 0xFFFF);
                const u32 b = NL_WALK_GET(w1 >> 16);
                const u32 lo_a = (w0 >> 4) & 15, n_a = (w0 >> 8) & 15, lo_b = (w0 >> 12) & 15, n_b = (w0 >> 16) & 15;
                if ((w2 & 0xFFFF) != 0xFFFF) lds[o_vals + (w2 & 0xFFFF) * 64 + lane] = (uint8_t)(((a >> lo_a) & ((1u << n_a) - 1)) | (((b >> lo_b) & ((1u << n_b) - 1)) << n_a));
            } else if (opc == NL_I_GATE) {
                const u32 n_known = (w0 >> 4) & 255, n_new = (w0 >> 12) & 255, mask_last = (w0 >> 20) & 1, new_step = w0 >> 24;
                const u32 constant = prog[pc + 1], w2 = prog[pc + 2], sh0 = prog[pc + 3];
                pc += 4;
                const u32 row = row0 + (w2 >> 16), col = w2 & 0xFFFF;
                long long sum = constant;
                for (u32 i = 0; i < n_known; i++) {
                    const u32 t = prog[pc + i], code = t >> 16;
                    if (code & NL_TERM_LATE) continue;  // in the constraint, not in the evaluation: an NL_I_LATE writes the cell
                    const long long x = (long long)NL_WALK_GET(t & 0xFFFF);
                    NL_SC(col + i, row) = (uint8_t)x;
                    sum += (code & 0x80) ? -(x << (code & 0x7F)) : (x << (code & 0x7F));
                }
                pc += n_known;
                for (u32 i = 0; i < n_new; i++) {
                    u64 x = (u64)sum >> (sh0 + i * new_step);
                    if (i + 1 < n_new || mask_last) x &= (1ull << new_step) - 1;
                    const u32 d = prog[pc + i] & 0xFFFF;
                    NL_SC(col + n_known + i, row) = (uint8_t)x;
                    if (d != 0xFFFF) lds[o_vals + d * 64 + lane] = (uint8_t)x;
                }
                pc += n_new;
            } else {  // NL_I_LATE
                const u32 w1 = prog[pc + 1], w2 = prog[pc + 2];
                pc += 3;
                NL_SC(w2 & 0xFFFF, row0 + (w2 >> 16)) = (uint8_t)NL_WALK_GET(w1 & 0xFFFF);
            }
        }
        // the state this step leaves: the next step's PREV bank
        const c_u16 so = out_src + (size_t)type * STATE;
        for (u32 k = 0; k < STATE; k++) lds[o_next + k * 64 + lane] = (uint8_t)NL_WALK_GET(so[k]);
        const u32 tmp = o_prev; o_prev = o_next; o_next = tmp;
    }
#undef NL_SC
}

// pass 2. grid (units, tiles of 64 cycles, instances), one wave per unit: unit < G * row_blocks: a general-purpose column x 64
// rows of the cycle; the others: a lookup slot (W columns + its keys) x 64 rows. 64 x 64 bytes of the scratch tile per column
// through LDS (row pitch 68 bytes: lanes read one byte each at distinct banks), then per cycle one 512-byte store per column.
template <int W, int R>
static __device__ __forceinline__ void k_nl_expand(const VB& vb, const NlDev* __restrict__ devp, const NlJob* __restrict__ jobs, u32 capacity, size_t n_rows,
                                                         const uint8_t* __restrict__ scratch, size_t scratch_per_job) {
    __shared__ __attribute__((aligned(16))) uint8_t tileb[W][64 * 68];
    const NlDev& D = *devp;
    const nl_spec& S = D.s;
    const NlJob job = jobs[vb.z];
    const u32 lane = threadIdx.x, tile = vb.y, RPC = S.rows_per_cycle, G = S.g;
    const u32 row_blocks = (RPC + 63) / 64;
    const bool general = vb.x < G * row_blocks;
    const u32 u = general ? vb.x : vb.x - G * row_blocks;
    const u32 rb = u % row_blocks, which = u / row_blocks;      // which: the column / the lookup slot
    const u32 col0 = general ? which : G + W * which, ncols = general ? 1 : W;
    const u32 r = rb * 64 + lane;                                // row of the cycle this lane stores
    const bool row_ok = r < RPC;
    const NlRowMeta rm = D.rowmeta[row_ok ? r : 0];
    if (general) {  // nothing of this column in these rows: leave (uniform)
        bool any = row_ok && (rm.flags & 1);
        if (!__any(any)) return;
    }
    const uint8_t* const sc = scratch + vb.z * scratch_per_job + (size_t)tile * ((size_t)S.mult_col * RPC * 64);
    const u32 rows_here = min(64u, RPC - rb * 64);
    for (u32 cc = 0; cc < ncols; cc++) {
        const u32* const src = reinterpret_cast<const u32*>(sc + ((size_t)(col0 + cc) * RPC + rb * 64) * 64);
        for (u32 d = lane; d < rows_here * 16; d += 64) *reinterpret_cast<u32*>(&tileb[cc][(d >> 4) * 68 + (d & 15) * 4]) = src[d];
    }
    __syncthreads();
    const u32 cyc0 = tile * 64, ncyc = min(64u, capacity - cyc0);
    u64* const trace = job.trace;
    if (general) {
        const bool write = row_ok && (rm.flags & 1), used = which < rm.rowend;
        if (!write) return;
        for (u32 k = 0; k < ncyc; k++)
            trace[(size_t)col0 * n_rows + (size_t)(cyc0 + k) * RPC + r] = used ? (u64)tileb[0][lane * 68 + k] : 0ull;
        return;
    }
    if (!row_ok) return;
    const bool lk = (rm.flags & 2) != 0;
    const u32 n_in = rm.keyfmt & 15, in_bits = rm.keyfmt >> 4;
    uint16_t* const keys = job.keys + rm.key_base + (size_t)which * rm.lookup_rows;
    for (u32 k = 0; k < ncyc; k++) {
        const size_t row = (size_t)(cyc0 + k) * RPC + r;
        u32 a[W];
#pragma unroll
        for (int cc = 0; cc < W; cc++) {
            a[cc] = lk ? tileb[cc][lane * 68 + k] : 0u;
            trace[(size_t)(col0 + cc) * n_rows + row] = a[cc];
        }
        if (lk) keys[(size_t)(cyc0 + k) * D.keys_per_cycle] = (uint16_t)(a[0] | (n_in > 1 ? a[1] << in_bits : 0u) | (n_in > 2 ? a[2] << (2 * in_bits) : 0u));
    }
}

// ---- multiplicities: grid (histogram slices, 2 halves of a table's rows, instances). A workgroup counts the keys of ITS table in
// its share of the cycles, the half's bins (at most 32768) in LDS, and stores them to its slice; k_nl_finish adds a table's slices
// up into the ONE multiplicity column (33 M global atomics per call, the first version, cost more than the counting).
constexpr int NL_HIST_THREADS = 1024;
constexpr int NL_HIST_HALF = 32768;
template <int R>
static __device__ __forceinline__ void k_nl_hist(const VB& vb, const NlDev* __restrict__ devp, const NlJob* __restrict__ jobs, u32 capacity, size_t n_rows) {
    const NlDev& D = *devp;
    const nl_spec& S = D.s;
    const NlJob job = jobs[vb.z];
    const u32 half = vb.y, t = threadIdx.x;
    u32 tb = 0;
    while (D.hist_slice0[tb + 1] <= vb.x) tb++;
    const nl_table T = S.tables[tb];
    if (half * NL_HIST_HALF >= T.rows) return;
    const u32 split = vb.x - D.hist_slice0[tb], n_splits = D.hist_slice0[tb + 1] - D.hist_slice0[tb];
    __shared__ u32 s_bins[NL_HIST_HALF];
    const u32 nbins = T.rows < NL_HIST_HALF ? T.rows : NL_HIST_HALF;
    for (u32 i = t; i < nbins; i += NL_HIST_THREADS) s_bins[i] = 0;
    __syncthreads();
    // the table's row runs of a cycle, cached with their running key counts: item i of a cycle -> (run, slot, row)
    constexpr int MAX_RUNS = 64;
    __shared__ NlHistEntry s_run[MAX_RUNS];
    __shared__ u32 s_run_first[MAX_RUNS + 1];
    const u32 e0 = D.hist_first[tb], n_runs = min(D.hist_first[tb + 1] - e0, (u32)MAX_RUNS);
    if (t < n_runs) s_run[t] = D.hist_entries[e0 + t];
    __syncthreads();
    if (t == 0) {
        u32 acc = 0;
        for (u32 e = 0; e < n_runs; e++) { s_run_first[e] = acc; acc += (s_run[e].r1 - s_run[e].r0) * R; }
        s_run_first[n_runs] = acc;
    }
    __syncthreads();
    const u32 per_cycle = s_run_first[n_runs], keys_per_cycle = D.keys_per_cycle;
    const u32 ncyc = capacity > split ? (capacity - split + n_splits - 1) / n_splits : 0;
    // flat item index over the workgroup's cycles -> (cycle, run, slot, row) without a hardware division: cycle and slot are
    // multiplications by reciprocals, the runs' first items are cached and a thread's run pointer only advances within a cycle.
    // The first version decomposed a 64-bit index with two divisions (~300 instructions per key: the kernel was bound by them,
    // not by the LDS atomics); four independent 2-byte loads are in flight before their counts.
    __shared__ u32 s_run_inv[MAX_RUNS];
    if (t < n_runs) s_run_inv[t] = (u32)(0xFFFFFFFFull / (s_run[t].r1 - s_run[t].r0)) + 1;  // ceil(2^32 / rows of the run) (1 row: wraps to 0, handled below)
    __syncthreads();
    const uint16_t* const keys = job.keys;
    const u32 total = ncyc * per_cycle;  // < 2^32: at most 2^20 rows x 26 lookups
    const u32 inv_pc = per_cycle > 1 ? (u32)(0xFFFFFFFFull / per_cycle) + 1 : 0;
    u32 e = 0, k_prev = 0;
    for (u32 i0 = t; i0 < total; i0 += 4 * NL_HIST_THREADS) {
        u32 key[4];
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const u32 i = i0 + b * NL_HIST_THREADS;
            key[b] = 0xFFFFFFFFu;
            if (i < total) {
                u32 ci = per_cycle > 1 ? (u32)(((u64)i * inv_pc) >> 32) : i;
                if (ci * per_cycle > i) ci--;              // (a reciprocal rounded up gives at most one too many)
                const u32 k = i - ci * per_cycle;
                if (k < k_prev) e = 0;
                k_prev = k;
                while (s_run_first[e + 1] <= k) e++;
                const NlHistEntry he = s_run[e];
                const u32 q = k - s_run_first[e], nr = he.r1 - he.r0;
                u32 sl = nr == 1 ? q : (u32)(((u64)q * s_run_inv[e]) >> 32);
                if (sl * nr > q) sl--;
                key[b] = keys[(size_t)(split + ci * n_splits) * keys_per_cycle + he.key0 + (size_t)sl * he.lookup_rows + he.r0 + (q - sl * nr)];
            }
        }
#pragma unroll
        for (int b = 0; b < 4; b++)
            if (key[b] != 0xFFFFFFFFu && key[b] / NL_HIST_HALF == half) atomicAdd(&s_bins[key[b] % NL_HIST_HALF], 1u);
    }
    __syncthreads();
    u32* const out = job.hist + ((size_t)vb.x * 2 + half) * NL_HIST_HALF;
    for (u32 i = t; i < nbins; i += NL_HIST_THREADS) out[i] = s_bins[i];
}

// the multiplicity column (sum of a table's slices), boundary rows (BND_IN, BND_OUT) and the public input row
static __device__ __forceinline__ void k_nl_finish(const VB& vb, const NlDev* __restrict__ devp, const NlJob* __restrict__ jobs, u32 capacity, size_t n_rows) {
    const NlDev& D = *devp;
    const nl_spec& S = D.s;
    const NlJob job = jobs[vb.y];
    const u32 i = vb.x * blockDim.x + threadIdx.x;
    if (i < S.total_table_rows) {
        u32 tb = 0;
        while (tb + 1 < S.n_tables && S.tables[tb + 1].offset <= i) tb++;
        const u32 e = i - S.tables[tb].offset, half = e / NL_HIST_HALF, bin = e % NL_HIST_HALF;
        u64 n = 0;
        for (u32 x = D.hist_slice0[tb]; x < D.hist_slice0[tb + 1]; x++) n += job.hist[((size_t)x * 2 + half) * NL_HIST_HALF + bin];
        NL_TR(S.mult_col, i) = n;
    }
    const size_t bnd = NL_BOUNDARY_ROW(&S, capacity), brows = NL_BND_ROWS(&S);
    if (i < S.state) {
        NL_TR(i % S.g, bnd + i / S.g) = job.state_before[i];
        NL_TR(i % S.g, bnd + brows + i / S.g) = job.state_before[(size_t)capacity * S.state + i];
    }
    if (i < 4) NL_TR(i, bnd + 2 * brows) = job.public_input[i];
}
#undef NL_TR

// ---- round records -> the engine's inputs. SHA-like: 64-byte blocks, state 8 words as 64 nibbles; Keccak-like: 136 / 200 bytes
struct NlPrepJob { const void* rounds; u64 first_round; u32 n_active; uint8_t* hdr_bits; uint8_t* free_elems; uint8_t* state_before; u32 state; /* elements of a cycle state (SHA-256: 64 nibbles of the chaining value, then zeros) */ };
static __device__ __forceinline__ void k_nl_prepare_sha(const VB& vb, const NlPrepJob* __restrict__ jobs, u32 capacity) {
    const NlPrepJob j = jobs[vb.y];
    const zkw_sha256_round_record* rounds = static_cast<const zkw_sha256_round_record*>(j.rounds);
    const u32 c = vb.x, t = threadIdx.x;  // c in [0, capacity]: the state BEFORE cycle c
    const u64 idx = j.first_round + (c < j.n_active ? c : j.n_active);  // records before the cycle
    for (u32 k = t; k < j.state; k += 128)
        j.state_before[(size_t)c * j.state + k] = (idx && k < 64) ? (uint8_t)((rounds[idx - 1].state_after[k >> 3] >> (4 * (k & 7))) & 15) : 0;
    if (c == capacity) return;
    const bool active = c < j.n_active;
    const uint8_t b = active ? rounds[j.first_round + c].block[t >> 1] : 0;
    j.free_elems[(size_t)c * 128 + t] = (uint8_t)((t & 1) ? b >> 4 : b & 15);
    if (t == 0) j.hdr_bits[c] = active ? (rounds[j.first_round + c].reset ? 1 : 0) : 2;
}
static __device__ __forceinline__ void k_nl_prepare_keccak(const VB& vb, const NlPrepJob* __restrict__ jobs, u32 capacity) {
    const NlPrepJob j = jobs[vb.y];
    const zkw_keccak_round_record* rounds = static_cast<const zkw_keccak_round_record*>(j.rounds);
    const u32 c = vb.x, t = threadIdx.x;
    const u64 idx = j.first_round + (c < j.n_active ? c : j.n_active);
    if (t < 200) j.state_before[(size_t)c * 200 + t] = idx ? rounds[idx - 1].state_after[t] : 0;
    if (c == capacity) return;
    const bool active = c < j.n_active;
    if (t < 136) j.free_elems[(size_t)c * 136 + t] = active ? rounds[j.first_round + c].block[t] : 0;
    if (t == 0) j.hdr_bits[c] = active ? (rounds[j.first_round + c].reset ? 1 : 0) : 2;
}

// ------------------------------------------------------------------------------------------------ checker
// violation kinds: 1 lookup relation / range, 2 copy constraint, 3 header, 4 boundary, 5 multiplicity, 6 non-zero unused cell,
// 7 gate arithmetic (the codes of oracle/netlist_circuit.c)
#define NL_TR(col, row) trace[(size_t)(col) * n_rows + (size_t)(row)]
__device__ u64 nl_home_cell(const nl_spec& S, const u64* __restrict__ trace, size_t n_rows, u32 capacity, u32 c, u32 s, u32 ref) {
    for (;;) {
        const nl_cycle_step& cs = S.cycle[s];
        const nl_step_type& T = S.step_types[cs.type];
        const size_t base = (size_t)c * S.rows_per_cycle + cs.row0;
        if (ref < NL_REF_HDR) {
            const nl_home h = S.homes[T.home0 + ref];
            if (h.kind == 1) {
                const nl_gate& g = S.gates[T.gate0 + h.item];
                return NL_TR(g.col + h.cell, base + g.row);
            }
            return NL_TR(S.g + S.w * (h.item % S.r) + h.cell, base + 1 + h.item / S.r);
        }
        if (ref < NL_REF_PREV) return NL_TR(ref - NL_REF_HDR, base);
        if (ref >= NL_REF_CONST) return ref - NL_REF_CONST;
        if (ref >= NL_REF_RC) return cs.rc[ref - NL_REF_RC];
        if (ref >= NL_REF_FREE) return 0;
        u32 k;
        if (ref >= NL_REF_CYC || s == 0) {
            k = ref >= NL_REF_CYC ? ref - NL_REF_CYC : ref - NL_REF_PREV;
            if (c == 0) {
                const size_t bnd = NL_BOUNDARY_ROW(&S, capacity);
                return NL_TR(k % S.g, bnd + k / S.g);
            }
            c--;
            s = S.steps_per_cycle - 1;
        } else {
            k = ref - NL_REF_PREV;
            s--;
        }
        ref = S.out[(size_t)S.cycle[s].type * S.state + k];
    }
}

// grid (chunks of items, capacity * steps_per_cycle): one lane per lookup / gate / row of a step instance
static __global__ __launch_bounds__(256) void k_nl_check_steps(const NlDev* __restrict__ devp, const u64* __restrict__ trace, u32 capacity, size_t n_rows,
                                                        u32* __restrict__ hist, CheckResult* res) {
    const nl_spec& S = devp->s;
    const u32 c = blockIdx.y / S.steps_per_cycle, s = blockIdx.y % S.steps_per_cycle;
    const nl_cycle_step& cs = S.cycle[s];
    const nl_step_type& T = S.step_types[cs.type];
    const size_t base = (size_t)c * S.rows_per_cycle + cs.row0;
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < T.n_ops) {
        const u32 j = i;
        const nl_op& op = S.ops[T.op0 + j];
        const nl_table& t = S.tables[op.table - 1];
        const size_t row = base + 1 + j / S.r;
        const u32 slot = j % S.r, col = S.g + S.w * slot;
        u32 a[3] = {0, 0, 0}, o[3];
        bool ok = true;
        for (u32 k = 0; k < t.n_in; k++) {
            const u64 x = NL_TR(col + k, row);
            if (x >> t.in_bits) ok = false;
            a[k] = (u32)x;
        }
        if (ok) {
            nl_table_eval(t.fn, t.param, a, o);
            for (u32 k = 0; k < t.n_out; k++) ok &= NL_TR(col + t.n_in + k, row) == o[k];
            for (u32 k = t.n_in + t.n_out; k < S.w; k++) ok &= NL_TR(col + k, row) == 0;
        }
        if (!ok) { flag_bad(res, 1, slot, row); return; }
        bool copies = true;
        for (u32 k = 0; k < t.n_in; k++) {
            const u32 ref = op.in[k];
            if (ref >= NL_REF_FREE && ref < NL_REF_RC) continue;
            if (ref < NL_REF_HDR) {
                const nl_home h = S.homes[T.home0 + ref];
                if (h.kind == 2 && h.item == j && h.cell == k) continue;  // the hint's own cell
            }
            if (a[k] != nl_home_cell(S, trace, n_rows, capacity, c, s, ref)) copies = false;
        }
        if (!copies) flag_bad(res, 2, slot, row);
        atomicAdd(&hist[nl_table_key(&t, a)], 1u);  // padding lookups hit entry 0 of their table like any other
    } else if (i < T.n_ops + T.n_gates) {
        const u32 gi = i - T.n_ops;
        const nl_gate& g = S.gates[T.gate0 + gi];
        const nl_term* tm = S.terms + T.term0 + g.first_term;
        const size_t row = base + g.row;
        u64 acc = gl::canon(g.constant);
        bool copies = true;
        for (u32 k = 0; k < (u32)g.n_known + g.n_new; k++) {
            const u64 x = NL_TR(g.col + k, row);
            if (k < g.n_known && !(tm[k].ref >= NL_REF_FREE && tm[k].ref < NL_REF_RC) && x != nl_home_cell(S, trace, n_rows, capacity, c, s, tm[k].ref)) copies = false;
            const u64 term = gl::canon(gl::mul(x, 1ull << (tm[k].code & 0x7F)));
            acc = gl::canon((tm[k].code & 0x80) ? gl::sub(acc, term) : gl::add(acc, term));
        }
        if (!copies) flag_bad(res, 2, 0x10000 + gi, row);
        if (acc != 0) flag_bad(res, 7, gi, row);
    } else if (i < T.n_ops + T.n_gates + T.rows) {
        const u32 r = i - T.n_ops - T.n_gates;
        const size_t row = base + r;
        if (r == 0) {
            if (s == 0) {
                const u64 reset = NL_TR(NL_HDR_RESET, base), idle = NL_TR(NL_HDR_IDLE, base);
                if (reset > 1 || idle > 1) flag_bad(res, 3, 0, base);
                else if (NL_TR(NL_HDR_M0, base) != (u64)(S.masks[0] + S.masks[1] * (long long)reset) ||
                         NL_TR(NL_HDR_M1, base) != (u64)(S.masks[2] + S.masks[3] * (long long)idle)) flag_bad(res, 3, 1, base);
            } else {
                const size_t b0 = (size_t)c * S.rows_per_cycle;
                for (int f = 0; f < NL_HDR_FIELDS; f++)
                    if (NL_TR(f, base) != NL_TR(f, b0)) { flag_bad(res, 3, 2, base); break; }
            }
        }
        for (u32 col = S.gate_row_end[T.rowend0 + r]; col < S.g; col++)
            if (NL_TR(col, row)) { flag_bad(res, 6, col, row); break; }
        if (r == 0 || r > T.lookup_rows)
            for (u32 col = S.g; col < S.mult_col; col++)
                if (NL_TR(col, row)) { flag_bad(res, 6, col, row); break; }
    }
}

static __global__ __launch_bounds__(256) void k_nl_check_tail(const NlDev* __restrict__ devp, const u64* __restrict__ trace, u32 capacity, size_t n_rows,
                                                       const u32* __restrict__ hist, CheckResult* res, u64 q_begin, u64 q_end, u64 e_begin, u64 e_end, u64 c_begin, u64 c_end) {
    // rows [q_begin, q_end): the queue section (netlist_queue_kernels.cuh checks its general-purpose cells; its lookup cells are zero);
    // rows [e_begin, e_end): the EC section of the ECRecover circuit (ecrecover_kernels.cuh checks every cell of it);
    // rows [c_begin, c_end): the closed-form section (netlist_closed_form_kernels.cuh: its general-purpose cells; lookup cells zero)
    const nl_spec& S = devp->s;
    const size_t bnd = NL_BOUNDARY_ROW(&S, capacity), brows = NL_BND_ROWS(&S);
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x; row < n_rows; row += stride) {
        if (NL_TR(S.mult_col, row) != (row < S.total_table_rows ? (u64)hist[row] : 0)) flag_bad(res, 5, 0, row);
        if (row < bnd || (row >= e_begin && row < e_end)) continue;
        const size_t off = row - bnd;
        for (u32 col = ((row >= q_begin && row < q_end) || (row >= c_begin && row < c_end)) ? S.g : 0; col < S.mult_col; col++) {
            bool allowed = false;
            if (off < 2 * brows) allowed = col < S.g && (off % brows) * S.g + col < S.state;
            else if (off == 2 * brows) allowed = col < 4;
            const u64 x = NL_TR(col, row);
            if (!allowed && x) { flag_bad(res, 6, col, row); break; }
            if (allowed && off < brows && x > 255) { flag_bad(res, 4, col, row); break; }
        }
        if (off >= brows && off < 2 * brows && capacity)
            for (u32 col = 0; col < S.g; col++) {
                const u32 k = (u32)(off - brows) * S.g + col;
                if (k < S.state && NL_TR(col, row) != nl_home_cell(S, trace, n_rows, capacity, capacity, 0, NL_REF_CYC + k)) flag_bad(res, 4, k, row);
            }
    }
}
#undef NL_TR

}  // namespace zkw
