// netlist_closed_form_kernels.cuh — the CLOSED-FORM SECTION of the netlist circuits on gfx950 (format, reference citations:
// include/zkw_netlist_closed_form.h): the flat encodings of an instance's observable input / output and hidden FSM input / output as
// cells of the trace, their commitments, the compact form and the public input as flattened Poseidon2 rows
// (ClosedFormInputCompactForm::from_full_form, src/witness/utils.rs:269-306), and the ties of those words to the registers of the trace.
//
//   k_nlcf_sponges<Cf>   one 256-lane workgroup per instance: the four encoders (public_input_kernels.cuh: CfPrecompile ...) run on one
//                        lane of a wave each into LDS, all lanes store the flags and words, then ONE wave walks the four sponges side by
//                        side — a 16-lane DPP row per sponge, p2::coop_flattened (the form of csrc/closed_form_kernels.cuh) — and the
//                        three permutations of the compact form. Reads nothing of the trace: the host launches it on a side stream,
//                        next to the netlist fill (a Keccak FSM is 55 dependent permutations: latency, not work).
//   k_nlcf_ties          one lane per tie cell, after the fills: a / b = the words (read back from the header cells), the digits =
//                        copies of the registers (QBND of the queue section, BND_IN / BND_OUT of the netlist).
//   k_nlcf_check         every relation of the section from the cells alone, the verdicts of oracle/netlist_closed_form.c.
#pragma once
#include "../../include/zkw_netlist_closed_form.h"
#include "netlist_kernels.cuh"
#include "poseidon2.cuh"
#include "public_input_kernels.cuh"

namespace zkw {

struct NlcfJob { u64* trace; u64 index; };  // the slot; the instance's index in the witness's records

#define NLCF_TR(col, row) trace[(size_t)(col) * n_rows + (size_t)(row)]
// cell k of the header block / variable v of P2 block `perm`
#define NLCF_H(k) NLCF_TR((k) % G, c0 + (k) / G)
#define NLCF_P(perm, v) NLCF_TR((v) % G, c0 + nlcf_perm_row0(&d, G, (perm)) + (v) / G)

template <class T>
static __device__ __forceinline__ void k_nlcf_sponges(const VB& vb, const nlcf_desc& d, const typename T::Inst* __restrict__ inst, const NlcfJob* __restrict__ jobs, u32 G,
                                                             size_t n_rows, u64 c0) {
    const NlcfJob job = jobs[vb.x];
    u64* __restrict__ trace = job.trace;
    __shared__ u64 sh_w[4][T::MAXLEN];
    __shared__ u64 sh_c[4][4];
    __shared__ u64 sh_flags[2];
    const u32 t = threadIdx.x, wave = t >> 6, lane = t & 63;
    if (lane == 0) {  // one lane of each wave: the four encoders side by side
        const typename T::Inst& me = inst[job.index];
        int m = 0;
        if (wave == 0) {  // the observable input is the one of the block's first instance (postprocessing/mod.rs:358-364)
            size_t j = job.index;
            while (j > 0 && !inst[j].start_flag) j--;
            m = T::input(inst[j], sh_w[NLCF_OI]);
            sh_flags[0] = me.start_flag ? 1 : 0;
            sh_flags[1] = me.completion_flag ? 1 : 0;
        } else if (wave == 1) m = T::output(me, sh_w[NLCF_OO]);
        else if (wave == 2) m = T::fsm(T::fsm_in(me), sh_w[NLCF_FI]);
        else m = T::fsm(T::fsm_out(me), sh_w[NLCF_FO]);
        if (m != (int)d.n[wave] || m > T::MAXLEN) __builtin_trap();  // the layout's word counts are the encoders'
    }
    __syncthreads();
    const u32 w1 = 2 + d.n[0], w2 = w1 + d.n[1], w3 = w2 + d.n[2], w4 = w3 + d.n[3];
    for (u32 k = t; k < w4; k += 256) {
        const u64 v = k < 2 ? sh_flags[k] : k < w1 ? sh_w[0][k - 2] : k < w2 ? sh_w[1][k - w1] : k < w3 ? sh_w[2][k - w2] : sh_w[3][k - w3];
        NLCF_H(k) = v;
    }
    // the sponges: row p of wave 0 = part p, a uniform number of rounds (a row that has run out permutes zeros and stores nothing).
    // A permutation's 130 stores go to (slot % G, row0 + slot / G): the division by the run-time G is a multiplication by its reciprocal
    // (exact for slot < 2^16), the block's first row is computed once per permutation — the chain is latency, every instruction of it counts
    const u32 g = lane & 15, part = lane >> 4;
    const u32 g_inv = (u32)((0x100000000ull + G - 1) / G), p2_rows = nlq_rows_for(NLQ_P2_CELLS, G), hdr_rows = nlcf_header_rows(&d, G);
    auto store_p2 = [&](u64* base /* (column 0, first row of the block) */, u32 slot, u64 v) {
        const u32 r = (u32)(((u64)slot * g_inv) >> 32), c = slot - r * G;
        base[(size_t)c * n_rows + r] = v;
    };
    p2::Coop co;
    co.init((int)g);
    if (wave == 0) {
        const u32 n = d.n[part], perms = (n + 7) / 8, perm0 = nlcf_perm0(&d, part);
        u32 most = 0;
        for (u32 p = 0; p < 4; p++) most = max(most, (u32)((d.n[p] + 7) / 8));
        u64 out = g == 11 ? (u64)n : 0;  // overwrite mode from (0, .., 0, n): apply_length_specialization
        u64* base = trace + c0 + hdr_rows + (size_t)perm0 * p2_rows;
        for (u32 q = 0; q < most; q++, base += p2_rows) {
            const bool on = q < perms;
            u64 x = 0;
            if (on && g < 8) x = 8 * q + g < n ? sh_w[part][8 * q + g] : 0;
            else if (on && g < 12) x = out;
            const u64 y = p2::coop_flattened(co, x, g, [&](u32 slot, u64 v) { if (on) store_p2(base, slot, v); });
            if (on) out = y;
        }
        if (g < 4) sh_c[part][g] = n ? out : 0;  // an empty encoding commits to zero
    }
    __syncthreads();
    if (wave == 0) {  // the compact form [start, completion, c(OI), c(OO), c(FI), c(FO)] -> the public input (row 0 of the wave)
        const bool on = part == 0;
        u64* base = trace + c0 + hdr_rows + (size_t)nlcf_perm0(&d, 4) * p2_rows;
        u64 out = g == 11 ? (u64)NLCF_CP_WORDS : 0;
        for (u32 q = 0; q < NLCF_CP_PERMS; q++, base += p2_rows) {
            const u32 k = 8 * q + g;
            u64 x = 0;
            if (on && g < 8) x = k >= NLCF_CP_WORDS ? 0 : k < 2 ? sh_flags[k] : sh_c[(k - 2) / 4][(k - 2) % 4];
            else if (on && g < 12) x = out;
            const u64 y = p2::coop_flattened(co, x, g, [&](u32 slot, u64 v) { if (on) store_p2(base, slot, v); });
            out = y;
        }
    }
}

// group and tie of tie-cell index i (cells of all groups back to back, 2 + n_cells per tie)
struct NlcfTieAt { u32 gi, j, c; };
__device__ __forceinline__ bool nlcf_tie_at(const nlcf_desc& d, u32 i, NlcfTieAt* at) {
    for (u32 gi = 0; gi < d.n_groups; gi++) {
        const u32 per = 2u + d.g[gi].n_cells, n = d.g[gi].count * per;
        if (i < n) { at->gi = gi; at->j = i / per; at->c = i % per; return true; }
        i -= n;
    }
    return false;
}
__device__ __forceinline__ u64 nlcf_reg(const nlcf_desc& d, const nlq_desc& qd, const nl_spec& S, const u64* __restrict__ trace, size_t n_rows, u32 cycles, u64 c0,
                                        const nlcf_group& gr, u32 j, u32 t) {
    if (gr.reg_kind == NLCF_REG_FO_WORD) {  // a word of the FSM output: the header block's own cell
        const u32 G = S.g, k = nlcf_word_cell(&d, NLCF_FO, gr.reg0 + j);
        return NLCF_H(k);
    }
    if (gr.reg_kind == NLCF_REG_QUEUE_BEFORE || gr.reg_kind == NLCF_REG_QUEUE_AFTER)
        return NLCF_TR(nlq_bnd_col(&qd, gr.queue, gr.reg_kind == NLCF_REG_QUEUE_AFTER, gr.reg0 + j), NLQ_BASE(&S, cycles));
    if (gr.reg_kind == NLCF_REG_OP_FIRST || gr.reg_kind == NLCF_REG_OP_LAST) {  // the GATED kinds: digit 0 the operation's cell, 1 / 2 the gates' enables
        const u32 op = t == 0 ? gr.queue : t == 1 ? gr.gate : gr.gate2, cell = t == 0 ? gr.reg0 : 0;
        const u32 c = gr.reg_kind == NLCF_REG_OP_LAST ? cycles - 1 : 0;
        if (op == NLCF_GATE_ACTIVE) return NLCF_TR(NL_HDR_IDLE, (size_t)c * S.rows_per_cycle);  // the cycle's idle bit
        return NLCF_TR(cell % S.g, NLQ_ROW(&S, cycles, nlq_op_row0(&qd, S.g, op) + cell / S.g, c));
    }
    const u32 e = (gr.reg0 + j) * gr.n_cells + t;
    return NLCF_TR(e % S.g, NL_BOUNDARY_ROW(&S, cycles) + (gr.reg_kind == NLCF_REG_STATE_OUT ? NL_BND_ROWS(&S) : 0) + e / S.g);
}
// the word a tie's a (side 0) / b (side 1) cell copies, read from the header block; no word: the constant 0
__device__ __forceinline__ u64 nlcf_side(const nlcf_desc& d, const u64* __restrict__ trace, size_t n_rows, u64 c0, u32 G, const nlcf_group& gr, u32 j, int side) {
    const int32_t w = nlcf_tie_word(&gr, side ? gr.b_word0 : gr.a_word0, j);
    if (w < 0) return side ? 0 : gr.a_const;
    const u32 k = nlcf_word_cell(&d, side ? nlcf_b_part(&gr) : nlcf_a_part(&gr), (u32)w);
    return NLCF_H(k);
}

// grid (tie cells / 256, instances)
static __device__ __forceinline__ void k_nlcf_ties(const VB& vb, const nlcf_desc& d, const nlq_desc& qd, const NlDev* __restrict__ devp, const NlcfJob* __restrict__ jobs, u32 cycles,
                                                          size_t n_rows, u64 c0) {
    const nl_spec& S = devp->s;
    const u32 G = S.g;
    u64* __restrict__ trace = jobs[vb.y].trace;
    NlcfTieAt at;
    if (!nlcf_tie_at(d, vb.x * blockDim.x + threadIdx.x, &at)) return;
    const nlcf_group& gr = d.g[at.gi];
    const u32 k = nlcf_tie_cell0(&d, at.gi, at.j) + at.c;
    NLCF_H(k) = at.c < 2 ? nlcf_side(d, trace, n_rows, c0, G, gr, at.j, (int)at.c) : nlcf_reg(d, qd, S, trace, n_rows, cycles, c0, gr, at.j, at.c - 2);
}

// the value input v of P2 block `perm` copies (oracle: perm_source)
__device__ __forceinline__ u64 nlcf_perm_source(const nlcf_desc& d, const u64* __restrict__ trace, size_t n_rows, u64 c0, u32 G, u32 perm, u32 v) {
    const u32 cp0 = nlcf_perm0(&d, 4);
    u32 part = 4, q = perm - cp0, n = NLCF_CP_WORDS;
    if (perm < cp0) {
        part = 0;
        while (perm >= nlcf_perm0(&d, part + 1)) part++;
        q = perm - nlcf_perm0(&d, part);
        n = d.n[part];
    }
    if (v >= 8) return q ? NLCF_P(perm - 1, 118 + v) : (v == 11 ? (u64)n : 0);
    const u32 w = 8 * q + v;
    if (w >= n) return 0;
    if (part < 4) return NLCF_H(nlcf_word_cell(&d, part, w));
    if (w < 2) return NLCF_H(w);
    const u32 cpart = (w - 2) / 4;
    return d.n[cpart] ? NLCF_P(nlcf_perm0(&d, cpart + 1) - 1, 118 + (w - 2) % 4) : 0;
}

__device__ __forceinline__ u32 nlcf_row_min(u32 x) {  // minimum over the 16 lanes of a DPP row
    for (int m = 8; m; m >>= 1) x = min(x, (u32)__shfl_xor((int)x, m, 16));
    return x;
}

// blocks [0, tie_blocks): one lane per tie; then blocks of 16 permutations (a DPP row each); the last block: flags, unused header cells, PI row
static __global__ __launch_bounds__(256) void k_nlcf_check(nlcf_desc d, nlq_desc qd, const NlDev* __restrict__ devp, const u64* __restrict__ trace, u32 cycles,
                                                           size_t n_rows, u64 c0, u32 tie_blocks, u32 perm_blocks, CheckResult* res) {
    const nl_spec& S = devp->s;
    const u32 G = S.g, t = threadIdx.x;
    if (blockIdx.x < tie_blocks) {
        u32 i = blockIdx.x * 256 + t, gi = 0;
        for (; gi < d.n_groups && i >= d.g[gi].count; gi++) i -= d.g[gi].count;
        if (gi >= d.n_groups) return;
        const nlcf_group& gr = d.g[gi];
        const u32 c = nlcf_tie_cell0(&d, gi, i);
        const u64 a = NLCF_H(c), b = NLCF_H(c + 1);
        const u64 start = NLCF_H(NLCF_CELL_START), completion = NLCF_H(NLCF_CELL_COMPLETION);
        if (a != nlcf_side(d, trace, n_rows, c0, G, gr, i, 0)) flag_bad(res, 2, c, c0 + c / G);
        if (b != nlcf_side(d, trace, n_rows, c0, G, gr, i, 1)) flag_bad(res, 2, c + 1, c0 + (c + 1) / G);
        u64 R = 0, dig[3] = {0, 0, 0};
        for (u32 k = gr.n_cells; k-- > 0;) {
            const u64 x = NLCF_H(c + 2 + k);
            if (x != nlcf_reg(d, qd, S, trace, n_rows, cycles, c0, gr, i, k)) flag_bad(res, 2, c + 2 + k, c0 + (c + 2 + k) / G);
            R = gl::add(gr.n_cells > 1 ? gl::mul(R, 1ull << gr.bits) : 0, x);
            if (k < 3) dig[k] = x;
        }
        R = gl::canon(R);
        const u64 am = gl::canon(a), bm = gl::canon(b);
        const bool has_b = gr.b_word0 >= 0;
        const u64 addf = gr.add >= 0 ? (u64)gr.add : gl::P - (u64)(-gr.add);
        bool ok;
        switch (gr.kind) {
            case NLCF_IN: ok = R == gl::canon(gl::add(bm, gl::mul(start, gl::sub(am, bm)))); break;
            case NLCF_OUT_LIVE: ok = completion == 1 || R == am; break;
            case NLCF_DONE: ok = gl::canon(gl::mul(completion, gl::sub(R, am))) == 0; break;
            case NLCF_OUT_OO: ok = has_b ? (R == bm && am == gl::canon(gl::mul(completion, bm))) : (R == am && am == gl::canon(gl::mul(completion, R))); break;
            case NLCF_IN_GATED: {  // (g1 - g2) (r - (b + start (a - b)) - add) = 0
                const u64 g1 = gr.gate == NLCF_GATE_ACTIVE ? gl::sub(1, dig[1]) : dig[1];
                const u64 gate = gl::sub(g1, gr.n_cells > 2 ? dig[2] : 0), sel = gl::add(bm, gl::mul(start, gl::sub(am, bm)));
                ok = gl::canon(gl::mul(gate, gl::sub(gl::sub(dig[0], sel), addf))) == 0;
                break;
            }
            case NLCF_OUT_GATED: {  // (1 - completion) (g1 - g2) (a -/+ r - add) = 0
                const u64 lhs = gr.negate ? gl::add(am, dig[0]) : gl::sub(am, dig[0]);
                const u64 gate = gr.n_cells < 2 ? 1 : gl::sub(dig[1], gr.n_cells > 2 ? dig[2] : 0);
                ok = gl::canon(gl::mul(gl::mul(gl::sub(1, completion), gate), gl::sub(lhs, addf))) == 0;
                break;
            }
            default: ok = R == am; break;
        }
        if (!ok) flag_bad(res, 7, c, c0 + c / G);
        return;
    }
    const u32 g = t & 15;
    if (blockIdx.x < tie_blocks + perm_blocks) {
        const u32 perm = (blockIdx.x - tie_blocks) * 16 + (t >> 4), n_perms = nlcf_n_perms(&d);
        const bool on = perm < n_perms;  // (a row beyond the last permutation runs on zeros: the DPP steps are wave-wide)
        p2::Coop co;
        co.init((int)g);
        u64 x = 0;
        if (on && g < 12) {
            x = NLCF_P(perm, g);
            if (x != nlcf_perm_source(d, trace, n_rows, c0, G, perm, g)) flag_bad(res, 2, (1u << 20) + 130 * perm + g, c0 + nlcf_perm_row0(&d, G, perm) + g / G);
        }
        u32 bad = ~0u;  // the first variable this lane finds wrong
        p2::coop_flattened(co, x, g, [&](u32 slot, u64 v) { if (on && slot >= 12 && NLCF_P(perm, slot) != v) bad = min(bad, slot); });
        bad = nlcf_row_min(bad);
        if (on && g == 0 && bad != ~0u) flag_bad(res, 8, (1u << 20) + 130 * perm + bad, c0 + nlcf_perm_row0(&d, G, perm) + bad / G);
        u32 junk = ~0u;
        const u32 cells = nlq_rows_for(NLQ_P2_CELLS, G) * G;
        if (on)
            for (u32 v = NLQ_P2_CELLS + g; v < cells; v += 16)
                if (NLCF_P(perm, v)) { junk = v; break; }
        junk = nlcf_row_min(junk);
        if (on && g == 0 && junk != ~0u) flag_bad(res, 6, junk % G, c0 + nlcf_perm_row0(&d, G, perm) + junk / G);
        return;
    }
    if (t < 2 && NLCF_H(t) > 1) flag_bad(res, 3, t, c0 + t / G);
    if (t < 4 && NLCF_TR(t, NL_PI_ROW(&S, cycles)) != NLCF_P(nlcf_n_perms(&d) - 1, 118 + t)) flag_bad(res, 4, t, NL_PI_ROW(&S, cycles));
    __shared__ u32 sh_junk;
    if (t == 0) sh_junk = ~0u;
    __syncthreads();
    const u32 hc = nlcf_header_cells(&d), hend = nlcf_header_rows(&d, G) * G;
    for (u32 k = hc + t; k < hend; k += 256)
        if (NLCF_H(k)) { atomicMin(&sh_junk, k); break; }
    __syncthreads();
    if (t == 0 && sh_junk != ~0u) flag_bad(res, 6, sh_junk % G, c0 + sh_junk / G);
}
#undef NLCF_P
#undef NLCF_H
#undef NLCF_TR

}  // namespace zkw
