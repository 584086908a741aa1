// closed_form_kernels.cuh — the CLOSED-FORM SECTION of a queue circuit's trace on gfx950: what the reference's circuits derive inside
// the trace — the Fiat-Shamir challenges (produce_fs_challenges, src/witness/utils.rs:498-550), the commitments of the closed-form input
// and the public input (commit_variable_length_encodable_item / ClosedFormInputCompactForm::from_full_form, utils.rs:269-306), the
// start-flag selection of the initial state (W/ram_permutation.rs:355-384) — as boundary rows: flattened Poseidon2 rows chained into
// overwrite-mode sponges, selection rows, copies (tools/gen_ram_circuit.py, class ClosedForm).
//
// A few dozen rows per trace, strictly sequential (every sponge row takes the row before): one 256-lane block per trace walks the generated SCHEDULE
// step by step (rows of one dependency level: the lanes share their copies / constants / FREE cells, every 16-lane DPP row runs one of their
// permutations, p2::coop_flattened) — in the boundary kernel of the circuit, after the fills the copies read. Nothing here is bandwidth: ~40 rows x 148 cells against a 2^20-row trace.
#pragma once
#include "../../include/zkw_ram_circuit_spec.h"
#include "poseidon2.cuh"

namespace zkw {

constexpr int CF_MAX_STEPS = 24, CF_MAX_STEP_ROWS = 96, CF_MAX_COPIES = 1536, CF_MAX_CONSTS = 192, CF_MAX_FREE = 256, CF_MAX_PRODUCTS = 64, CF_MAX_BYTES = 64,
              CF_MAX_LINEARS = 96, CF_MAX_LIN_TERMS = 256;  // LDS copies of the tables
struct CfSpec {
    int n_steps, rows_per_cycle, n_step_rows, n_copies, n_consts, n_free, n_products, n_bytes, n_linears, n_lin_terms;
    const rc_cf_step* steps;      // the parallel schedule of the spec header: rows grouped by dependency level, tables sorted by step
    const uint8_t* step_rows;
    const rc_cf_copy* copies;
    const uint8_t* is_poseidon;
    const rc_cf_const* consts;
    const rc_cf_free* frees;
    const rc_cf_product* products;
    const rc_cf_bytes* bytes;          // lookup cells = the bytes of a limb cell of the row (their multiplicities: the circuit's fill kernels count them from the records)
    const rc_cf_linear* linears;       // cell = constant + sum coef * cell of the row
    const rc_cf_lin_term* lin_terms;
};

// the section's tables of a spec header in constant memory, and the members a checker spec struct (SpecRam ...) exposes them through
#define ZKW_CF_TABLES(PFX, pfx)                                                                                          \
    static __constant__ rc_cf_step c_##pfx##_cf_steps[PFX##_CF_NUM_STEPS] = PFX##_CF_STEPS_INIT;                         \
    static __constant__ uint8_t c_##pfx##_cf_step_rows[PFX##_CF_NUM_STEP_ROWS] = PFX##_CF_STEP_ROWS_INIT;                \
    static __constant__ rc_cf_copy c_##pfx##_cf_copies[PFX##_CF_NUM_COPIES] = PFX##_CF_COPIES_INIT;                      \
    static __constant__ rc_cf_const c_##pfx##_cf_consts[PFX##_CF_NUM_CONSTS] = PFX##_CF_CONSTS_INIT;                     \
    static __constant__ rc_cf_free c_##pfx##_cf_free[PFX##_CF_NUM_FREE] = PFX##_CF_FREE_INIT;                            \
    static __constant__ rc_cf_product c_##pfx##_cf_products[PFX##_CF_NUM_PRODUCTS ? PFX##_CF_NUM_PRODUCTS : 1] = PFX##_CF_PRODUCTS_INIT;       \
    static __constant__ rc_cf_bytes c_##pfx##_cf_bytes[PFX##_CF_NUM_BYTES ? PFX##_CF_NUM_BYTES : 1] = PFX##_CF_BYTES_INIT;                     \
    static __constant__ rc_cf_linear c_##pfx##_cf_linears[PFX##_CF_NUM_LINEARS ? PFX##_CF_NUM_LINEARS : 1] = PFX##_CF_LINEARS_INIT;            \
    static __constant__ rc_cf_lin_term c_##pfx##_cf_lin_terms[PFX##_CF_NUM_LIN_TERMS ? PFX##_CF_NUM_LIN_TERMS : 1] = PFX##_CF_LIN_TERMS_INIT;  \
    static_assert(PFX##_CF_NUM_STEPS <= CF_MAX_STEPS && PFX##_CF_NUM_STEP_ROWS <= CF_MAX_STEP_ROWS && PFX##_CF_NUM_COPIES <= CF_MAX_COPIES && \
                  PFX##_CF_NUM_CONSTS <= CF_MAX_CONSTS && PFX##_CF_NUM_FREE <= CF_MAX_FREE && PFX##_CF_NUM_PRODUCTS <= CF_MAX_PRODUCTS &&     \
                  PFX##_CF_NUM_BYTES <= CF_MAX_BYTES && PFX##_CF_NUM_LINEARS <= CF_MAX_LINEARS && PFX##_CF_NUM_LIN_TERMS <= CF_MAX_LIN_TERMS, \
                  "closed-form tables outgrew their LDS copies");
#define ZKW_CF_SPEC(PFX, pfx, is_poseidon_table)                                                                         \
    CfSpec{PFX##_CF_NUM_STEPS, PFX##_ROWS_PER_CYCLE, PFX##_CF_NUM_STEP_ROWS, PFX##_CF_NUM_COPIES, PFX##_CF_NUM_CONSTS, PFX##_CF_NUM_FREE, PFX##_CF_NUM_PRODUCTS, \
           PFX##_CF_NUM_BYTES, PFX##_CF_NUM_LINEARS, PFX##_CF_NUM_LIN_TERMS, c_##pfx##_cf_steps, c_##pfx##_cf_step_rows, c_##pfx##_cf_copies, is_poseidon_table, \
           c_##pfx##_cf_consts, c_##pfx##_cf_free, c_##pfx##_cf_products, c_##pfx##_cf_bytes, c_##pfx##_cf_linears, c_##pfx##_cf_lin_terms}
#define ZKW_CF_SPEC_MEMBERS(PFX, pfx) \
    __device__ static CfSpec cf_spec() { return ZKW_CF_SPEC(PFX, pfx, is_poseidon()); }

struct CfSources {  // LDS pointers; picked by a select chain (an array indexed at run time would live in scratch memory)
    const u64 *obs_in, *fsm_in, *fsm_out, *flags, *obs_out;
    __device__ __forceinline__ const u64* pick(int k) const { return k == 0 ? obs_in : k == 1 ? fsm_in : k == 2 ? fsm_out : k == 3 ? flags : obs_out; }
};

constexpr int CF_THREADS = 256;  // 16 DPP rows: up to 16 permutations of a step side by side

// The walk, by a CF_THREADS-lane block. src: where the FREE cells come from (encodings staged in LDS by the caller: 0 observable input,
// 1 hidden FSM input, 2 hidden FSM output, 3 flags, 4 observable output); bnd = first boundary row. hook(row_type, row) runs between a
// row's copies / products and its permutation (all lanes call it, for every row of the step). The register rows the section copies from
// must be written and visible (barrier before). A step = rows that depend only on earlier steps: their copies / constants / FREE cells
// are shared by all lanes, their permutations run one per 16-lane DPP row (p2::coop_flattened) — the sponges of the observable input,
// the FSM input and output and the challenges advance side by side, so a section costs its longest chain (12 - 19 steps), not its rows.
template <class Hook>
__device__ __forceinline__ void cf_fill_block(const CfSpec& S, u64* __restrict__ trace, size_t n_rows, size_t bnd, const CfSources& src, Hook&& hook) {
#define CF_CELL(col, row) trace[(size_t)(col) * n_rows + (row)]
#define CF_ROW(rt) (bnd + (size_t)((rt) - S.rows_per_cycle))
    const u32 t = threadIdx.x, g = t & 15, grp = t >> 4;
    // the tables in LDS: every step reads them with per-lane indices (from constant memory that is a vector load per read, in the
    // dependent chain of every step)
    __shared__ rc_cf_step sh_steps[CF_MAX_STEPS];
    __shared__ uint8_t sh_step_rows[CF_MAX_STEP_ROWS];
    __shared__ rc_cf_copy sh_copies[CF_MAX_COPIES];
    __shared__ rc_cf_const sh_consts[CF_MAX_CONSTS];
    __shared__ rc_cf_free sh_free[CF_MAX_FREE];
    __shared__ rc_cf_product sh_products[CF_MAX_PRODUCTS];
    __shared__ rc_cf_bytes sh_bytes[CF_MAX_BYTES];
    __shared__ rc_cf_linear sh_linears[CF_MAX_LINEARS];
    __shared__ rc_cf_lin_term sh_lin_terms[CF_MAX_LIN_TERMS];
    for (int k = (int)t; k < S.n_steps; k += CF_THREADS) sh_steps[k] = S.steps[k];
    for (int k = (int)t; k < S.n_step_rows; k += CF_THREADS) sh_step_rows[k] = S.step_rows[k];
    for (int k = (int)t; k < S.n_copies; k += CF_THREADS) sh_copies[k] = S.copies[k];
    for (int k = (int)t; k < S.n_consts; k += CF_THREADS) sh_consts[k] = S.consts[k];
    for (int k = (int)t; k < S.n_free; k += CF_THREADS) sh_free[k] = S.frees[k];
    for (int k = (int)t; k < S.n_products; k += CF_THREADS) sh_products[k] = S.products[k];
    for (int k = (int)t; k < S.n_bytes; k += CF_THREADS) sh_bytes[k] = S.bytes[k];
    for (int k = (int)t; k < S.n_linears; k += CF_THREADS) sh_linears[k] = S.linears[k];
    for (int k = (int)t; k < S.n_lin_terms; k += CF_THREADS) sh_lin_terms[k] = S.lin_terms[k];
    __syncthreads();
    p2::Coop co;
    co.init((int)g);
    for (int st = 0; st < S.n_steps; st++) {
        const rc_cf_step s = sh_steps[st];
        // the cells a step initialises are independent: all loads first, then the stores
        u64 cv[CF_MAX_COPIES / CF_THREADS];  // (unrolled: registers)
#pragma unroll
        for (int r = 0; r < CF_MAX_COPIES / CF_THREADS; r++) {
            const int k = (int)t + r * CF_THREADS;
            if (k < s.n_copies) { const rc_cf_copy c = sh_copies[s.copy0 + k]; cv[r] = CF_CELL(c.col_b, CF_ROW(c.row_b)); }
        }
        for (int k = (int)t; k < s.n_consts; k += CF_THREADS) {
            const rc_cf_const c = sh_consts[s.const0 + k];
            CF_CELL(c.col, CF_ROW(c.row)) = c.value;
        }
        for (int k = (int)t; k < s.n_free; k += CF_THREADS) {
            const rc_cf_free f = sh_free[s.free0 + k];
            CF_CELL(f.col, CF_ROW(f.row)) = src.pick(f.src)[f.idx];
        }
#pragma unroll
        for (int r = 0; r < CF_MAX_COPIES / CF_THREADS; r++) {
            const int k = (int)t + r * CF_THREADS;
            if (k < s.n_copies) { const rc_cf_copy c = sh_copies[s.copy0 + k]; CF_CELL(c.col_a, CF_ROW(c.row_a)) = cv[r]; }
        }
        __syncthreads();
        if (s.n_bytes) {  // (uniform) the bytes of limb cells the rows copied: one lane per byte
            for (int k = (int)t; k < 4 * s.n_bytes; k += CF_THREADS) {
                const rc_cf_bytes b = sh_bytes[s.byte0 + (k >> 2)];
                const size_t row = CF_ROW(b.row);
                CF_CELL(b.col_b0 + (k & 3), row) = (CF_CELL(b.col_limb, row) >> (8 * (k & 3))) & 0xFF;
            }
            __syncthreads();
        }
        for (int k = (int)t; k < s.n_lins; k += CF_THREADS) {  // linear combinations of the row's cells (copied cells, bytes)
            const rc_cf_linear l = sh_linears[s.lin0 + k];
            const size_t row = CF_ROW(l.row);
            u64 acc = l.constant;
            for (int j = 0; j < l.n_terms; j++) {
                const rc_cf_lin_term tm = sh_lin_terms[l.term0 + j];
                acc = gl::add(acc, gl::mul(tm.coef, CF_CELL(tm.col, row)));
            }
            CF_CELL(l.col, row) = gl::canon(acc);
        }
        for (int k = (int)t; k < s.n_prods; k += CF_THREADS) {  // a product's factors are cells the row copied
            const rc_cf_product p = sh_products[s.prod0 + k];
            const size_t row = CF_ROW(p.row);
            CF_CELL(p.col, row) = gl::canon(gl::mul(CF_CELL(p.col_a, row), CF_CELL(p.col_b, row)));
        }
        for (int k = 0; k < s.n_rows; k++) {
            const int rt = sh_step_rows[s.row0 + k];
            hook(rt, CF_ROW(rt));
        }
        __syncthreads();
        for (int k0 = 0; k0 < s.n_rows; k0 += CF_THREADS / 16) {  // uniform trip count; a group without a Poseidon2 row runs the permutation on zeros and stores nothing
            const int k = k0 + (int)grp;
            const int rt = k < s.n_rows ? sh_step_rows[s.row0 + k] : -1;
            const bool p2row = rt >= 0 && S.is_poseidon[rt];
            const size_t row = p2row ? CF_ROW(rt) : bnd;
            const u64 x = p2row && g < 12 ? CF_CELL(g, row) : 0;
            p2::coop_flattened(co, x, g, [&](u32 slot, u64 v) { if (p2row) CF_CELL(slot, row) = v; });
        }
        __syncthreads();
    }
#undef CF_ROW
#undef CF_CELL
}

// the section of a circuit whose closed form has a record encoder Cf (public_input_kernels.cuh: CfEventsSorter ...; S = its checker spec):
// one lane of waves 1..3 each stages encodings of the records in LDS, then the walk. The caller's lane 0 may write the register rows
// before calling (the barrier in here orders them). Called by every lane of a CF_THREADS-lane block.
template <class Cf, class S, class Hook>
__device__ __forceinline__ void cf_section_from_records(const typename Cf::Inst* first, const typename Cf::Inst* inst, u64* trace, size_t n_rows, size_t bnd,
                                                        Hook&& hook) {
    __shared__ u64 sh_oi[Cf::MAXLEN], sh_oo[Cf::MAXLEN], sh_fi[Cf::MAXLEN], sh_fo[Cf::MAXLEN], sh_flags[2];
    if (threadIdx.x == 64) {  // (one lane of each of the other three waves: the encoders run side by side)
        Cf::input(*first, sh_oi);
        Cf::output(*inst, sh_oo);
    }
    if (threadIdx.x == 128) Cf::fsm(Cf::fsm_in(*inst), sh_fi);
    if (threadIdx.x == 192) {
        Cf::fsm(Cf::fsm_out(*inst), sh_fo);
        sh_flags[0] = inst->start_flag ? 1 : 0;
        sh_flags[1] = inst->completion_flag ? 1 : 0;
    }
    __syncthreads();
    const CfSources src = {sh_oi, sh_fi, sh_fo, sh_flags, sh_oo};
    cf_fill_block(S::cf_spec(), trace, n_rows, bnd, src, hook);
}

// a lookup cell of a section row takes a byte: the multiplicity column counted a zero there (the tail kernels count every lookup cell below
// the cycles as zero). One lane.
__device__ __forceinline__ void cf_put_byte(u64* trace, size_t n_rows, int mult_col, int col, size_t row, u64 b) {
    trace[(size_t)col * n_rows + row] = b;
    trace[(size_t)mult_col * n_rows + b] += 1;
    trace[(size_t)mult_col * n_rows] -= 1;
}

}  // namespace zkw
