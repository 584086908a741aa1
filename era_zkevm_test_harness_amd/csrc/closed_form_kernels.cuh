// closed_form_kernels.cuh — the CLOSED-FORM SECTION of a queue circuit's trace on gfx950: what the reference's circuits derive inside
// the trace — the Fiat-Shamir challenges (produce_fs_challenges, src/witness/utils.rs:498-550), the commitments of the closed-form input
// and the public input (commit_variable_length_encodable_item / ClosedFormInputCompactForm::from_full_form, utils.rs:269-306), the
// start-flag selection of the initial state (W/ram_permutation.rs:355-384) — as boundary rows: flattened Poseidon2 rows chained into
// overwrite-mode sponges, selection rows, copies (tools/gen_ram_circuit.py, class ClosedForm).
//
// A few dozen rows per trace, strictly sequential (every sponge row takes the row before): one wave per trace walks the generated tables
// row type by row type — the lanes share a row's copies / constants / FREE cells, 16 of them its permutation (p2::coop_flattened) — in the
// boundary kernel of the circuit, after the fills the copies read. Nothing here is bandwidth: ~40 rows x 148 cells against a 2^20-row trace.
#pragma once
#include "../../include/zkw_ram_circuit_spec.h"
#include "poseidon2.cuh"

namespace zkw {

struct CfSpec {
    int first, n, n_links, n_consts, n_free, n_products, rows_per_cycle, pi_row;
    const rc_link* links;
    const uint8_t* is_poseidon;
    const rc_cf_const* consts;
    const rc_cf_free* frees;
    const rc_cf_product* products;
};

// the section's tables of a spec header in constant memory, and the members a checker spec struct (SpecRam ...) exposes them through
#define ZKW_CF_TABLES(PFX, pfx)                                                                                          \
    static __constant__ rc_cf_const c_##pfx##_cf_consts[PFX##_CF_NUM_CONSTS] = PFX##_CF_CONSTS_INIT;                     \
    static __constant__ rc_cf_free c_##pfx##_cf_free[PFX##_CF_NUM_FREE] = PFX##_CF_FREE_INIT;                            \
    static __constant__ rc_cf_product c_##pfx##_cf_products[PFX##_CF_NUM_PRODUCTS ? PFX##_CF_NUM_PRODUCTS : 1] = PFX##_CF_PRODUCTS_INIT;
#define ZKW_CF_SPEC_MEMBERS(PFX, pfx)                                                                                    \
    __device__ static CfSpec cf_spec() {                                                                                 \
        return CfSpec{PFX##_CF_FIRST_ROW_TYPE, PFX##_CF_NUM_ROWS, PFX##_NUM_LINKS, PFX##_CF_NUM_CONSTS, PFX##_CF_NUM_FREE, PFX##_CF_NUM_PRODUCTS, \
                      PFX##_ROWS_PER_CYCLE, PFX##_ROW_PI, links(), is_poseidon(), c_##pfx##_cf_consts, c_##pfx##_cf_free, c_##pfx##_cf_products}; \
    }

struct CfSources {  // LDS pointers; picked by a select chain (an array indexed at run time would live in scratch memory)
    const u64 *obs_in, *fsm_in, *fsm_out, *flags, *obs_out;
    __device__ __forceinline__ const u64* pick(int k) const { return k == 0 ? obs_in : k == 1 ? fsm_in : k == 2 ? fsm_out : k == 3 ? flags : obs_out; }
};

// src[k]: where the FREE cells of source k come from (0 observable input, 1 hidden FSM input, 2 hidden FSM output, 3 flags, 4 observable
// output: encodings staged in LDS by the caller); bnd = first boundary row. hook(row_type, row) runs between a row's copies and its
// permutation (all lanes call it). Called by every lane of a 64-lane block; the rows it copies from must be visible (barrier before).
template <class Hook>
__device__ __forceinline__ void cf_fill_wave(const CfSpec& S, u64* __restrict__ trace, size_t n_rows, size_t bnd, const CfSources& src, Hook&& hook) {
#define CF_CELL(col, row) trace[(size_t)(col) * n_rows + (row)]
    const u32 lane = threadIdx.x, g = lane & 15;
    p2::Coop co;
    co.init((int)g);
    for (int r = S.first; r <= S.first + S.n; r++) {
        const int rt = r < S.first + S.n ? r : S.pi_row;  // last: the public-input row takes its copies
        const size_t row = bnd + (size_t)(rt - S.rows_per_cycle);
        for (int l = (int)lane; l < S.n_links; l += 64) {
            const rc_link k = S.links[l];
            if (k.kind == 5 && k.row_a == rt) CF_CELL(k.col_a, row) = CF_CELL(k.col_b, bnd + (size_t)(k.row_b - S.rows_per_cycle));
        }
        if (rt == S.pi_row) break;
        for (int k = (int)lane; k < S.n_consts; k += 64)
            if (S.consts[k].row == rt) CF_CELL(S.consts[k].col, row) = S.consts[k].value;
        for (int k = (int)lane; k < S.n_free; k += 64)
            if (S.frees[k].row == rt) CF_CELL(S.frees[k].col, row) = src.pick(S.frees[k].src)[S.frees[k].idx];
        __syncthreads();
        for (int k = (int)lane; k < S.n_products; k += 64)  // a product's factors are cells the row copied
            if (S.products[k].row == rt) CF_CELL(S.products[k].col, row) = gl::canon(gl::mul(CF_CELL(S.products[k].col_a, row), CF_CELL(S.products[k].col_b, row)));
        hook(rt, row);
        __syncthreads();
        if (S.is_poseidon[rt]) {  // uniform
            const u64 x = g < 12 ? CF_CELL(g, row) : 0;
            const bool writer = lane < 16;  // the other three rows of the wave run the same permutation and store nothing
            p2::coop_flattened(co, x, g, [&](u32 slot, u64 v) { if (writer) CF_CELL(slot, row) = v; });
        }
        __syncthreads();
    }
#undef CF_CELL
}

// the section of a circuit whose closed form has a record encoder Cf (public_input_kernels.cuh: CfEventsSorter ...; S = its checker spec):
// lanes 1..3 stage the encodings of the records in LDS, then the walk. The caller's lane 0 may write the register rows before calling
// (the barrier in here orders them). Called by every lane of a 64-lane block.
template <class Cf, class S, class Hook>
__device__ __forceinline__ void cf_section_from_records(const typename Cf::Inst* first, const typename Cf::Inst* inst, u64* trace, size_t n_rows, size_t bnd,
                                                        Hook&& hook) {
    __shared__ u64 sh_oi[Cf::MAXLEN], sh_oo[Cf::MAXLEN], sh_fi[Cf::MAXLEN], sh_fo[Cf::MAXLEN], sh_flags[2];
    if (threadIdx.x == 1) {
        Cf::input(*first, sh_oi);
        Cf::output(*inst, sh_oo);
    }
    if (threadIdx.x == 2) Cf::fsm(Cf::fsm_in(*inst), sh_fi);
    if (threadIdx.x == 3) {
        Cf::fsm(Cf::fsm_out(*inst), sh_fo);
        sh_flags[0] = inst->start_flag ? 1 : 0;
        sh_flags[1] = inst->completion_flag ? 1 : 0;
    }
    __syncthreads();
    const CfSources src = {sh_oi, sh_fi, sh_fo, sh_flags, sh_oo};
    cf_fill_wave(S::cf_spec(), trace, n_rows, bnd, src, hook);
}

// a lookup cell of a section row takes a byte: the multiplicity column counted a zero there (the tail kernels count every lookup cell below
// the cycles as zero). One lane.
__device__ __forceinline__ void cf_put_byte(u64* trace, size_t n_rows, int mult_col, int col, size_t row, u64 b) {
    trace[(size_t)col * n_rows + row] = b;
    trace[(size_t)mult_col * n_rows + b] += 1;
    trace[(size_t)mult_col * n_rows] -= 1;
}

}  // namespace zkw
