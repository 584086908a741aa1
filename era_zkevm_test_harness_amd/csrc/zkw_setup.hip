// zkw_setup.hip — what a setup needs to know about this library's trace layouts: geometry, layout description (finalization
// hint fields), row selectors, the copy permutation (sigma) and its check on a trace.
#include "zkw_ctx.h"
#include "ram_circuit_kernels.cuh"
#include "decommit_sorter_circuit_kernels.cuh"
#include "events_sorter_circuit_kernels.cuh"
#include "log_demux_circuit_kernels.cuh"
#include "storage_sorter_circuit_kernels.cuh"
#include "netlist_kernels.cuh"
#include "../../include/zkw_netlist_closed_form.h"
#include "../../include/zkw_ecrecover.h"

extern "C" int zkw_circuit_geometry_of(uint8_t circuit_type, zkw_circuit_geometry* out) {
    // {copy columns, lookup width, repetitions, max degree, capacity, big size hint}: vm_main.rs:29-44,
    // sort_code_decommits.rs:28-39, code_decommitter.rs:28-39, log_demux.rs:36-47, keccak256_round_function.rs:28-39,
    // sha256_round_function.rs:28-39, ecrecover.rs:30-41, ram_permutation.rs:26-41,117-122, storage_sort_dedup.rs:29-40,
    // storage_apply.rs:28-39, events_sort_dedup.rs:28-39 (x2), linear_hasher.rs:28-39; geometry_config.rs:5-20
    static const struct { u32 c, lw, lr, deg, cap; bool big; } T[14] = {
        {0, 0, 0, 0, 0, false},
        {130, 3, 8, 8, 5585, false},    // 1 MainVM
        {130, 1, 18, 8, 117500, true},  // 2 CodeDecommittmentsSorter
        {108, 4, 11, 8, 2845, false},   // 3 CodeDecommitter
        {136, 1, 14, 8, 58750, true},   // 4 LogDemuxer
        {86, 3, 14, 8, 293, false},     // 5 KeccakRoundFunction
        {116, 4, 9, 8, 2206, false},    // 6 Sha256RoundFunction
        {80, 3, 16, 8, 7, false},       // 7 ECRecover
        {133, 1, 15, 8, 136714, true},  // 8 RAMPermutation
        {132, 1, 16, 8, 46921, true},   // 9 StorageSorter
        {60, 3, 26, 8, 33, false},      // 10 StorageApplication
        {130, 1, 8, 18, 31287, true},   // 11 EventsSorter
        {130, 1, 8, 18, 31287, true},   // 12 L1MessagesSorter
        {66, 3, 26, 8, 774, false},     // 13 L1MessagesHasher
    };
    if (!out || circuit_type < 1 || circuit_type > 13) return fail(ZKW_ERR_INVALID, "unknown base-layer circuit type %u", circuit_type);
    const auto& g = T[circuit_type];
    out->num_columns_under_copy_permutation = g.c;
    out->num_witness_columns = 0;
    out->num_constant_columns = 4;
    out->max_allowed_constraint_degree = g.deg;
    out->lookup_width = g.lw;
    out->lookup_repetitions = g.lr;
    out->capacity = g.cap;
    out->trace_len_log2 = 20;
    out->size_hint_variables = g.big ? (1ull << 26) + (1ull << 25) : (1ull << 26);
    return ZKW_OK;
}

// Where this library's own trace layout ("zkw trace v2") of a circuit type puts things: what the reference keeps in
// FinalizationHintsForProver / VerificationKey.fixed_parameters for ITS layout (setup/base_layer/finalization_hint_N.json:
// `public_inputs` = (column, row) of the four PI cells, `nop_gates_to_add`, `final_trace_len`). No GPU needed.
extern "C" int zkw_circuit_layout_of(uint8_t circuit_type, uint32_t capacity, zkw_circuit_layout* out) {
    if (!out) return fail(ZKW_ERR_INVALID, "zkw_circuit_layout_of: null argument");
    zkw_circuit_geometry g;
    ZKW_TRY(zkw_circuit_geometry_of(circuit_type, &g));
    memset(out, 0, sizeof *out);
    if (capacity == 0) capacity = g.capacity;
    out->capacity = capacity;
    out->trace_len = 1ull << g.trace_len_log2;
    uint64_t boundary = 0, min_rows = 0, pi_off = 0;
    switch (circuit_type) {
        case 8: out->num_columns = RC_COLS; out->rows_per_cycle = RC_ROWS_PER_CYCLE; out->region_stride = RC_REGION_STRIDE(capacity); boundary = RC_BOUNDARY_ROW(capacity); min_rows = RC_MIN_ROWS(capacity); pi_off = RC_ROWOFF_PI; break;
        case 2: out->num_columns = DS_COLS; out->rows_per_cycle = DS_ROWS_PER_CYCLE; out->region_stride = DS_REGION_STRIDE(capacity); boundary = DS_BOUNDARY_ROW(capacity); min_rows = DS_MIN_ROWS(capacity); pi_off = DS_ROWOFF_PI; break;
        case 4: out->num_columns = LD_COLS; out->rows_per_cycle = LD_ROWS_PER_CYCLE; out->region_stride = LD_REGION_STRIDE(capacity); boundary = LD_BOUNDARY_ROW(capacity); min_rows = LD_MIN_ROWS(capacity); pi_off = LD_ROWOFF_PI; break;
        case 9: out->num_columns = SS_COLS; out->rows_per_cycle = SS_ROWS_PER_CYCLE; out->region_stride = SS_REGION_STRIDE(capacity); boundary = SS_BOUNDARY_ROW(capacity); min_rows = SS_MIN_ROWS(capacity); pi_off = SS_ROWOFF_PI; break;
        case 11: case 12: out->num_columns = ES_COLS; out->rows_per_cycle = ES_ROWS_PER_CYCLE; out->region_stride = ES_REGION_STRIDE(capacity); boundary = ES_BOUNDARY_ROW(capacity); min_rows = ES_MIN_ROWS(capacity); pi_off = ES_ROWOFF_PI; break;
        // the netlist circuits ("zkw trace v4") are cycle-major: region_stride = 0, cycle i starts at row i * rows_per_cycle
        case 3: case 5: case 6: case 7: case 10: case 13: {
            const nl_spec* sp = nl_host_spec(circuit_type);
            const uint32_t cycles = nl_cycles_of(circuit_type, capacity);
            out->num_columns = sp->cols; out->rows_per_cycle = sp->rows_per_cycle; boundary = NL_BOUNDARY_ROW(sp, cycles);
            const nlq_desc* qd = nlq_desc_of(circuit_type);
            min_rows = nlq_used_rows(sp, qd, cycles); pi_off = 2 * NL_BND_ROWS(sp);
            out->total_table_rows = sp->total_table_rows;
            if (qd) { out->queue_first_row = NLQ_BASE(sp, cycles); out->queue_rows_per_cycle = nlq_rows_per_cycle(qd, sp->g); }
            if (circuit_type == 7) {  // the EC section below the queue section, cycle-major (include/zkw_ecrecover.h)
                out->ec_first_row = min_rows;
                out->ec_rows_per_cycle = EC_ROWS_PER_CYCLE;
                min_rows += (uint64_t)cycles * EC_ROWS_PER_CYCLE;
            }
            if (const nlcf_desc* cd = nlcf_desc_of(circuit_type)) {  // the closed-form section: the last rows in use
                out->closed_form_first_row = min_rows;
                out->closed_form_rows = nlcf_rows(cd, sp->g);
                out->closed_form_header_rows = nlcf_header_rows(cd, sp->g);
                min_rows += out->closed_form_rows;
            }
            if (circuit_type == 7 && min_rows < sp->total_table_rows) min_rows = sp->total_table_rows;  // the multiplicity column has a row per table row (197 632 > 7 cycles' rows)
            break;
        }
        default: return ZKW_OK;  // a known circuit type this library does not synthesize yet: synthesizable = 0
    }
    out->synthesizable = 1;
    if (out->region_stride) out->total_table_rows = 256;  // the queue circuits' one table: RangeCheckTable<8>
    out->rows_used = min_rows;
    out->fits = min_rows <= out->trace_len;
    out->nop_rows = out->fits ? out->trace_len - min_rows : 0;
    for (int k = 0; k < 4; k++) {
        out->public_input_column[k] = (uint32_t)k;
        out->public_input_row[k] = boundary + pi_off;
    }
    return ZKW_OK;
}

// Bytes a synthesis call writes into one slot (algorithmic: the cells the fill kernels store, 8 bytes each) — what the measured rates
// of bench.py's `hash_circuits` leg are divided into. `warm`: the slot already holds this layout (its zeros are kept, the netlist
// engine's slot tag); `cold`: any other slot — every cell of the slot's columns is written once (cleared or filled).
extern "C" int zkw_circuit_fill_bytes(uint8_t circuit_type, uint32_t capacity, size_t n_rows, uint64_t* warm, uint64_t* cold) {
    zkw_circuit_layout lay;
    ZKW_TRY(zkw_circuit_layout_of(circuit_type, capacity, &lay));
    if (!lay.synthesizable || !warm || !cold) return fail(ZKW_ERR_INVALID, "zkw_circuit_fill_bytes: bad argument");
    *cold = (uint64_t)lay.num_columns * n_rows * 8;
    if (lay.region_stride) {
        // the queue circuits: a slot that already holds the layout gets the cells a row type uses (its slots + its lookup cells, per cycle),
        // every cell of the boundary rows and the 256 multiplicity rows; the cells that are zero in every trace — unused columns of a row
        // type, the gap rows of a region, the padding below the boundary rows, multiplicity rows >= 256 — keep their zeros
        static const int rc_s[] = RC_ROW_NUM_SLOTS_INIT, rc_l[] = RC_ROW_NUM_LOOKUPS_INIT, ds_s[] = DS_ROW_NUM_SLOTS_INIT, ds_l[] = DS_ROW_NUM_LOOKUPS_INIT,
                         es_s[] = ES_ROW_NUM_SLOTS_INIT, es_l[] = ES_ROW_NUM_LOOKUPS_INIT, ld_s[] = LD_ROW_NUM_SLOTS_INIT, ld_l[] = LD_ROW_NUM_LOOKUPS_INIT,
                         ss_s[] = SS_ROW_NUM_SLOTS_INIT, ss_l[] = SS_ROW_NUM_LOOKUPS_INIT;
        const int *slots = nullptr, *looks = nullptr;
        switch (circuit_type) {
            case ZKW_CIRCUIT_RAM_PERMUTATION: slots = rc_s; looks = rc_l; break;
            case ZKW_CIRCUIT_CODE_DECOMMITTMENTS_SORTER: slots = ds_s; looks = ds_l; break;
            case ZKW_CIRCUIT_LOG_DEMUXER: slots = ld_s; looks = ld_l; break;
            case ZKW_CIRCUIT_STORAGE_SORTER: slots = ss_s; looks = ss_l; break;
            case ZKW_CIRCUIT_EVENTS_SORTER: case ZKW_CIRCUIT_L1_MESSAGES_SORTER: slots = es_s; looks = es_l; break;
            default: return fail(ZKW_ERR_INVALID, "zkw_circuit_fill_bytes: circuit type %u has no region-major spec", (unsigned)circuit_type);
        }
        const uint64_t rows = (lay.rows_used + 1) & ~1ull;
        uint64_t per_cycle = 0;
        for (uint32_t r = 0; r < lay.rows_per_cycle; r++) per_cycle += (uint64_t)slots[r] + looks[r];
        *warm = (per_cycle * lay.capacity + (uint64_t)(lay.num_columns - 1) * (rows - (uint64_t)lay.rows_per_cycle * lay.region_stride) + 256) * 8;
        return ZKW_OK;
    }
    const nl_spec* sp = nl_host_spec(circuit_type);
    const uint32_t cycles = nl_cycles_of(circuit_type, lay.capacity);
    uint64_t per_cycle = 0;
    for (uint32_t st = 0; st < sp->steps_per_cycle; st++) {
        const nl_step_type& T = sp->step_types[sp->cycle[st].type];
        per_cycle += (uint64_t)T.lookup_rows * sp->w * sp->r;
        for (uint32_t r = 0; r < T.rows; r++) per_cycle += sp->gate_row_end[T.rowend0 + r];
    }
    uint64_t cells = per_cycle * cycles + 2ull * sp->state + 4 + sp->total_table_rows;
    if (const nlq_desc* qd = nlq_desc_of(circuit_type)) {
        uint64_t q = 0;
        for (uint32_t j = 0; j < qd->n_ops; j++) q += nlq_enc_cells(&qd->ops[j]) + (uint64_t)nlq_kind_perms(qd->ops[j].kind) * NLQ_P2_CELLS;
        cells += q * cycles + nlq_bnd_cells(qd);
    }
    if (circuit_type == 7) cells += (uint64_t)cycles * EC_ROWS_PER_CYCLE * EC_ROW_CELLS;
    if (const nlcf_desc* cd = nlcf_desc_of(circuit_type)) cells += nlcf_header_cells(cd) + (uint64_t)nlcf_n_perms(cd) * NLQ_P2_CELLS;
    *warm = cells * 8;
    return ZKW_OK;
}

// Setup side, selectors: which gate set applies to each row of a trace of this library's layout — what the
// reference's setup keeps in its constant columns (gate selectors, the lookup table id of a row). Host arithmetic over the specs.
//   queue circuits (2, 4, 8, 9, 11, 12; "zkw trace v2", region-major): selector = row type of the spec (0 .. NUM_ROW_TYPES - 1:
//     the per-cycle row types, then the boundary rows), ZKW_ROW_PADDING elsewhere (gaps of a region, rows after the boundary)
//   netlist circuits (3, 5, 6, 13; "zkw trace v3", cycle-major): selector = lookup table id of the row (0: none) |
//     ZKW_ROW_HAS_GATES when ADD gates sit in its general-purpose columns | ZKW_ROW_HEADER for a cycle's first row;
//     boundary rows ZKW_ROW_BOUNDARY + k; ZKW_ROW_PADDING elsewhere
extern "C" int zkw_setup_row_selectors(uint8_t circuit_type, uint32_t capacity, size_t n_rows, uint8_t* out) {
    if (!out || n_rows == 0) return fail(ZKW_ERR_INVALID, "zkw_setup_row_selectors: null argument");
    zkw_circuit_layout lay;
    ZKW_TRY(zkw_circuit_layout_of(circuit_type, capacity, &lay));
    if (!lay.synthesizable) return fail(ZKW_ERR_INVALID, "circuit type %u has no layout in this library", (unsigned)circuit_type);
    if (lay.rows_used > n_rows) return fail(ZKW_ERR_INVALID, "capacity %u needs %llu rows, %zu given", lay.capacity, (unsigned long long)lay.rows_used, n_rows);
    memset(out, ZKW_ROW_PADDING, n_rows);
    const uint32_t cap = lay.capacity;
    if (lay.region_stride) {  // region-major
        const uint64_t stride = lay.region_stride, rpc = lay.rows_per_cycle;
        for (uint64_t r = 0; r < rpc; r++) memset(out + r * stride, (int)r, cap);
        const uint64_t bnd = rpc * stride;
        for (uint64_t k = 0; bnd + k < lay.rows_used; k++) out[bnd + k] = (uint8_t)(rpc + k);
        return ZKW_OK;
    }
    const uint32_t cycles = nl_cycles_of(circuit_type, cap);
    const uint64_t rpc = lay.rows_per_cycle;
    const nl_spec* sp = nl_host_spec(circuit_type);
    std::vector<uint8_t> one(rpc);  // every cycle has the same selectors
    for (uint32_t st = 0; st < sp->steps_per_cycle; st++) {
        const nl_step_type& T = sp->step_types[sp->cycle[st].type];
        uint8_t* row = one.data() + sp->cycle[st].row0;
        row[0] = ZKW_ROW_HEADER;
        for (uint32_t r = 1; r < T.rows; r++)
        {
            uint32_t tb = r <= T.lookup_rows ? sp->ops[T.op0 + (r - 1) * sp->r].table : 0;
            if (tb > 2 && sp->n_tables > 0x3F) tb -= 256;  // ECRecover's 262 tables: its netlist uses Xor8 (1), And8 (2) and ByteSplit<k> (259..262 -> 3..6)
            row[r] = (uint8_t)(tb | (sp->gate_row_end[T.rowend0 + r] ? ZKW_ROW_HAS_GATES : 0));
        }
    }
    for (uint32_t c = 0; c < cycles; c++) memcpy(out + (uint64_t)c * rpc, one.data(), rpc);
    for (uint64_t k = 0; (uint64_t)cycles * rpc + k < NL_USED_ROWS(sp, cycles); k++) out[(uint64_t)cycles * rpc + k] = (uint8_t)(ZKW_ROW_BOUNDARY + k);
    if (circuit_type == 7) {  // the EC section: per row the table of its lookup slots / whether it is a MUL row / has gates
        EC_DEFINE_SPEC(ecs);
        std::vector<uint8_t> cyc(EC_ROWS_PER_CYCLE, ZKW_ROW_EC_GATES);
        for (uint32_t r = 0; r < EC_NUM_RUNS; r++) {
            const ec_seg_type& T = ecs_types[ecs_runs[r].type];
            std::vector<uint8_t> seg(T.n_rows, ZKW_ROW_EC_GATES);
            for (uint32_t row = 0; row < T.n_rows; row++) {
                const uint32_t tb = ecs_rowtab[T.rowtab0 + row];
                if (tb) seg[row] = (tb & 0x7FFF) == EC_T_XOR8 ? ZKW_ROW_EC_XOR8 : ZKW_ROW_EC_FIXED_BASE;
            }
            const uint32_t* w = ecs_items + T.item0;
            for (uint32_t i = 0; i < T.n_items; i++, w += ec_item_words(w))
                if ((w[0] & 15) == EC_I_MUL) seg[(w[0] >> 4) & 0xFFF] |= ZKW_ROW_EC_MUL;
            for (uint32_t j = 0; j < ecs_runs[r].count; j++) memcpy(cyc.data() + ecs_runs[r].row0 + (size_t)j * T.n_rows, seg.data(), T.n_rows);
        }
        for (uint32_t c = 0; c < cycles; c++) memcpy(out + lay.ec_first_row + (uint64_t)c * EC_ROWS_PER_CYCLE, cyc.data(), EC_ROWS_PER_CYCLE);
    }
    if (const nlq_desc* qd = nlq_desc_of(circuit_type)) {  // the queue section: region-major below the PI row
        out[NLQ_BASE(sp, cycles)] = ZKW_ROW_QUEUE_BOUNDARY;
        for (uint32_t j = 0; j < qd->n_ops; j++) {
            const uint32_t r0 = nlq_op_row0(qd, sp->g, j), erows = nlq_rows_for(nlq_enc_cells(&qd->ops[j]), sp->g), rows = nlq_op_rows(&qd->ops[j], sp->g);
            for (uint32_t r = 0; r < rows; r++) memset(out + NLQ_ROW(sp, cycles, r0 + r, 0), r < erows ? ZKW_ROW_QUEUE_ENCODING : ZKW_ROW_QUEUE_POSEIDON2, cycles);
        }
    }
    if (lay.closed_form_rows) {
        memset(out + lay.closed_form_first_row, ZKW_ROW_CLOSED_FORM_WORDS, lay.closed_form_header_rows);
        memset(out + lay.closed_form_first_row + lay.closed_form_header_rows, ZKW_ROW_CLOSED_FORM_POSEIDON2, lay.closed_form_rows - lay.closed_form_header_rows);
    }
    return ZKW_OK;
}

// Setup side, copy permutation of the queue circuits: sigma[c][r] = the cell (c' * n_rows + r') that follows
// cell (c, r) in its copy cycle; a cell under no copy constraint maps to itself. Built on the host from the spec's link table
// (the same table the satisfiability checker walks, ram_circuit_kernels.cuh k_check_links) with a union-find over the
// general-purpose cells; cycles run through their cells in increasing cell order. Seconds at production size (1.1 GB of output).
namespace {
struct LinkSpec { int G /* general-purpose + lookup columns: links reach both */, rows_per_cycle, num_links, off_bin, off_bout; const rc_link* links; };
static const rc_link h_rc_links[RC_NUM_LINKS] = RC_LINKS_INIT;
static const rc_link h_ds_links[DS_NUM_LINKS] = DS_LINKS_INIT;
static const rc_link h_es_links[ES_NUM_LINKS] = ES_LINKS_INIT;
static const rc_link h_ld_links[LD_NUM_LINKS] = LD_LINKS_INIT;
static const rc_link h_ss_links[SS_NUM_LINKS] = SS_LINKS_INIT;
bool link_spec_of(uint8_t t, LinkSpec* o) {
    switch (t) {
        case 8: *o = {RC_G + RC_L, RC_ROWS_PER_CYCLE, RC_NUM_LINKS, RC_ROWOFF_BND_IN, RC_ROWOFF_BND_OUT, h_rc_links}; return true;
        case 2: *o = {DS_G + DS_L, DS_ROWS_PER_CYCLE, DS_NUM_LINKS, DS_ROWOFF_BND_IN, DS_ROWOFF_BND_OUT, h_ds_links}; return true;
        case 11: case 12: *o = {ES_G + ES_L, ES_ROWS_PER_CYCLE, ES_NUM_LINKS, ES_ROWOFF_BND_IN, ES_ROWOFF_BND_OUT, h_es_links}; return true;
        case 4: *o = {LD_G + LD_L, LD_ROWS_PER_CYCLE, LD_NUM_LINKS, LD_ROWOFF_BND_IN, LD_ROWOFF_BND_OUT, h_ld_links}; return true;
        case 9: *o = {SS_G + SS_L, SS_ROWS_PER_CYCLE, SS_NUM_LINKS, SS_ROWOFF_BND_IN, SS_ROWOFF_BND_OUT, h_ss_links}; return true;
        default: return false;
    }
}
}  // namespace

extern "C" int zkw_setup_copy_permutation(uint8_t circuit_type, uint32_t capacity, size_t n_rows, uint64_t* sigma, uint32_t* n_columns) {
    LinkSpec sp;
    const bool netlist = nl_is_netlist(circuit_type);
    if (netlist) sp = {(int)nl_host_spec(circuit_type)->mult_col, 0, 0, 0, 0, nullptr};  // all but the multiplicity column
    else if (!link_spec_of(circuit_type, &sp))
        return fail(ZKW_ERR_INVALID, "zkw_setup_copy_permutation: circuit type %u has no layout in this library", (unsigned)circuit_type);
    zkw_circuit_layout lay;
    ZKW_TRY(zkw_circuit_layout_of(circuit_type, capacity, &lay));
    if (n_columns) *n_columns = (uint32_t)sp.G;
    if (!sigma) return ZKW_OK;  // size query
    if (lay.rows_used > n_rows || n_rows >= (1ull << 32) / (size_t)sp.G)
        return fail(ZKW_ERR_INVALID, "capacity %u needs %llu rows, %zu given", lay.capacity, (unsigned long long)lay.rows_used, n_rows);
    const uint32_t cap = lay.capacity;
    const uint64_t rs = lay.region_stride, bnd = (uint64_t)sp.rows_per_cycle * rs;
    const size_t n_cells = (size_t)sp.G * n_rows;
    std::vector<uint32_t> parent(n_cells);
    for (size_t i = 0; i < n_cells; i++) parent[i] = (uint32_t)i;
    auto find = [&](uint32_t x) {
        while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; }
        return x;
    };
    bool out_of_range = false;
    auto unite = [&](uint64_t col_a, uint64_t row_a, uint64_t col_b, uint64_t row_b) {
        if (col_a >= (uint64_t)sp.G || col_b >= (uint64_t)sp.G || row_a >= n_rows || row_b >= n_rows) { out_of_range = true; return; }
        uint32_t a = find((uint32_t)(col_a * n_rows + row_a)), b = find((uint32_t)(col_b * n_rows + row_b));
        if (a != b) parent[a > b ? a : b] = a > b ? b : a;  // the smallest cell of a class is its root
    };
    auto brow = [&](int rt) { return bnd + (uint64_t)(rt - sp.rows_per_cycle); };  // a boundary row type
    if (netlist) {
        // the netlist circuits: every operand cell of a lookup / gate is a copy of the cell that produced it (the references of
        // the spec, resolved exactly as the checkers do: k_kc_check_rows, k_sc_check_cycle); constants and free witness bytes
        // are under no copy constraint
        const uint32_t cycles = nl_cycles_of(circuit_type, cap);
        const nl_spec* ns = nl_host_spec(circuit_type);
        const uint64_t nb = NL_BOUNDARY_ROW(ns, cycles), brows = NL_BND_ROWS(ns);
        // the cell a reference names, seen from step st of cycle c (nl_home of netlist_kernels.cuh as coordinates); false: a constant
        auto home = [&](uint32_t c, uint32_t st, uint32_t ref, uint64_t* hc, uint64_t* hr) {
            for (;;) {
                const nl_cycle_step& cs = ns->cycle[st];
                const nl_step_type& T = ns->step_types[cs.type];
                const uint64_t base = (uint64_t)c * ns->rows_per_cycle + cs.row0;
                if (ref < NL_REF_HDR) {
                    const nl_home h = ns->homes[T.home0 + ref];
                    if (h.kind == 1) { const nl_gate& g = ns->gates[T.gate0 + h.item]; *hc = g.col + h.cell; *hr = base + g.row; }
                    else { *hc = ns->g + ns->w * (h.item % ns->r) + h.cell; *hr = base + 1 + h.item / ns->r; }
                    return true;
                }
                if (ref < NL_REF_PREV) { *hc = ref - NL_REF_HDR; *hr = base; return true; }
                if (ref >= NL_REF_FREE && ref < NL_REF_RC) return false;
                if (ref >= NL_REF_RC) return false;
                uint32_t k;
                if (ref >= NL_REF_CYC || st == 0) {
                    k = ref >= NL_REF_CYC ? ref - NL_REF_CYC : ref - NL_REF_PREV;
                    if (c == 0) { *hc = k % ns->g; *hr = nb + k / ns->g; return true; }
                    c--;
                    st = ns->steps_per_cycle - 1;
                } else {
                    k = ref - NL_REF_PREV;
                    st--;
                }
                ref = ns->out[(size_t)ns->cycle[st].type * ns->state + k];
            }
        };
        uint64_t hc, hr;
        for (uint32_t c = 0; c < cycles; c++)
            for (uint32_t st = 0; st < ns->steps_per_cycle; st++) {
                const nl_cycle_step& cs = ns->cycle[st];
                const nl_step_type& T = ns->step_types[cs.type];
                const uint64_t base = (uint64_t)c * ns->rows_per_cycle + cs.row0;
                if (st)  // a step's header is a copy of the cycle's
                    for (int f = 0; f < NL_HDR_FIELDS; f++) unite((uint64_t)f, base, (uint64_t)f, (uint64_t)c * ns->rows_per_cycle);
                for (uint32_t j = 0; j < T.n_ops; j++) {
                    const nl_op& op = ns->ops[T.op0 + j];
                    const nl_table& tb = ns->tables[op.table - 1];
                    for (uint32_t i = 0; i < tb.n_in; i++) {
                        if (op.in[i] < NL_REF_HDR) {
                            const nl_home h = ns->homes[T.home0 + op.in[i]];
                            if (h.kind == 2 && h.item == j && h.cell == i) continue;  // a hint's own cell
                        }
                        if (home(c, st, op.in[i], &hc, &hr)) unite((uint64_t)(ns->g + ns->w * (j % ns->r) + i), base + 1 + j / ns->r, hc, hr);
                    }
                }
                for (uint32_t gi = 0; gi < T.n_gates; gi++) {
                    const nl_gate& g = ns->gates[T.gate0 + gi];
                    for (uint32_t i = 0; i < g.n_known; i++)
                        if (home(c, st, ns->terms[T.term0 + g.first_term + i].ref, &hc, &hr)) unite((uint64_t)(g.col + i), base + g.row, hc, hr);
                }
            }
        if (cycles)
            for (uint32_t k = 0; k < ns->state; k++)
                if (home(cycles, 0, NL_REF_CYC + k, &hc, &hr)) unite((uint64_t)(k % ns->g), nb + brows + k / ns->g, hc, hr);
        // the queue section (include/zkw_netlist_queue.h): a permutation's inputs are copies of the encoding / the old state / the
        // previous permutation's capacity, `old` of the previous `new` on the queue, the linked value nibbles of their netlist cells
        if (const nlq_desc* qd = nlq_desc_of(circuit_type)) {
            const uint32_t G = ns->g;
            const uint64_t q0 = NLQ_BASE(ns, cycles);
            auto qc = [&](uint32_t c, uint32_t r0, uint32_t k, uint64_t* col, uint64_t* row) { *col = k % G; *row = NLQ_ROW(ns, cycles, r0 + k / G, c); };
            auto free_home = [&](uint32_t fi, uint64_t* col, uint64_t* row_in_cycle) {  // the one cell that uses FREE element fi of the (single-step or first-step) cycle
                uint32_t off = 0;
                for (uint32_t st = 0; st < ns->steps_per_cycle; st++) {
                    const nl_cycle_step& cs = ns->cycle[st];
                    const nl_step_type& T = ns->step_types[cs.type];
                    if (fi >= off && fi < off + T.n_free) {
                        const uint32_t ref = NL_REF_FREE + (fi - off);
                        for (uint32_t j = 0; j < T.n_ops; j++) {
                            const nl_op& op = ns->ops[T.op0 + j];
                            if (op.out == 0xFFFF) continue;
                            for (uint32_t i = 0; i < ns->tables[op.table - 1].n_in; i++)
                                if (op.in[i] == ref) { *col = ns->g + ns->w * (j % ns->r) + i; *row_in_cycle = cs.row0 + 1 + j / ns->r; return true; }
                        }
                        for (uint32_t gi = 0; gi < T.n_gates; gi++) {
                            const nl_gate& g = ns->gates[T.gate0 + gi];
                            for (uint32_t i = 0; i < g.n_known; i++)
                                if (ns->terms[T.term0 + g.first_term + i].ref == ref) { *col = g.col + i; *row_in_cycle = cs.row0 + g.row; return true; }
                        }
                    }
                    off += T.n_free;
                }
                return false;
            };
            std::vector<int> last_op(qd->n_queues, -1);  // per queue: the operation whose `new` is the current state (-1: QBND in)
            std::vector<uint32_t> last_cyc(qd->n_queues, 0);
            uint64_t ca, ra, cb, rb;
            for (uint32_t c = 0; c < cycles; c++)
                for (uint32_t j = 0; j < qd->n_ops; j++) {
                    const nlq_op& op = qd->ops[j];
                    const uint32_t w = nlq_kind_width(op.kind), r0 = nlq_op_row0(qd, G, j), e0 = nlq_enc0(&op), o0 = nlq_old0(&op);
                    for (uint32_t k = 1; k < nlq_item_comps(op.item); k++) {
                        uint32_t cyc = 0, ref = 0;
                        if (!nlq_link_target(&op, c, cycles, k, &cyc, &ref)) continue;
                        qc(c, r0, k, &ca, &ra);
                        if (ref >= NL_REF_CYC && ref < NL_REF_FREE) { if (home(cyc, 0, ref, &hc, &hr)) unite(ca, ra, hc, hr); }
                        else if (free_home(ref - NL_REF_FREE, &cb, &rb)) unite(ca, ra, cb, (uint64_t)cyc * ns->rows_per_cycle + rb);
                    }
                    for (uint32_t p = 0; p < nlq_kind_perms(op.kind); p++) {
                        const uint32_t pr0 = nlq_p2_row0(qd, G, j, p);
                        for (uint32_t k = 0; k < 12; k++) {
                            qc(c, pr0, k, &ca, &ra);
                            if (op.kind != NLQ_POP4) { if (k < 8) qc(c, r0, e0 + k, &cb, &rb); else qc(c, r0, o0 + k, &cb, &rb); }
                            else if (k >= 8) { if (p == 0) continue; qc(c, nlq_p2_row0(qd, G, j, p - 1), NLQ_P2_CELLS - 12 + k, &cb, &rb); }
                            else if (p < 2) qc(c, r0, e0 + 8 * p + k, &cb, &rb);
                            else if (k < 4) qc(c, r0, e0 + 16 + k, &cb, &rb);
                            else qc(c, r0, o0 + (k - 4), &cb, &rb);
                            unite(ca, ra, cb, rb);
                        }
                    }
                    for (uint32_t k = 0; k < w; k++) {
                        qc(c, r0, o0 + k, &ca, &ra);
                        if (last_op[op.queue] < 0) { cb = nlq_bnd_col(qd, op.queue, 0, k); rb = q0; }
                        else qc(last_cyc[op.queue], nlq_op_row0(qd, G, (uint32_t)last_op[op.queue]), nlq_new0(&qd->ops[last_op[op.queue]]) + k, &cb, &rb);
                        unite(ca, ra, cb, rb);
                    }
                    last_op[op.queue] = (int)j; last_cyc[op.queue] = c;
                }
            for (uint32_t q = 0; q < qd->n_queues; q++)
                for (uint32_t k = 0; k < qd->width[q] && last_op[q] >= 0; k++) {
                    qc(last_cyc[q], nlq_op_row0(qd, G, (uint32_t)last_op[q]), nlq_new0(&qd->ops[last_op[q]]) + k, &cb, &rb);
                    unite(nlq_bnd_col(qd, q, 1, k), q0, cb, rb);
                }
            // ECRecover's EC section (include/zkw_ecrecover.h): every cell with a tape reference is a copy of the value's home cell; an input
            // byte's home is a copy of the read query's value byte in the queue section, its other occurrences of that home; the netlist's
            // FREE elements (key bytes, mask, ok) are copies of EC home cells — the relations k_ec_check_rows / k_ec_check_links walk
            if (circuit_type == ZKW_CIRCUIT_ECRECOVER) {
                EC_DEFINE_SPEC(ecs);
                const ec_spec S = {ecs_types, ecs_runs, ecs_items, ecs_item_index, ecs_cells, ecs_homes, ecs_outs, ecs_rowtab, ecs_globs, ecs_bigs, ecs_in_home, ecs_key_byte, nullptr};
                for (uint32_t c = 0; c < cycles; c++) {
                    const uint64_t cyc0 = lay.ec_first_row + (uint64_t)c * EC_ROWS_PER_CYCLE;
                    for (uint32_t r = 0; r < EC_ROWS_PER_CYCLE; r++) {
                        uint32_t run, inst, row, prun, pinst, hr2, hc2;
                        ec_locate_row(&S, r, &run, &inst, &row);
                        const ec_seg_type& T = S.types[S.runs[run].type];
                        ec_prev_segment(&S, run, inst, &prun, &pinst);
                        const uint32_t base = S.runs[run].tape0 + inst * T.n_tape, pbase = S.runs[prun].tape0 + pinst * S.types[S.runs[prun].type].n_tape;
                        const uint32_t* cells = S.cells + T.cell0 + (size_t)row * EC_ROW_CELLS;
                        for (uint32_t col = 0; col < EC_ROW_CELLS; col++) {
                            const uint32_t ref = cells[col];
                            if (ref == EC_NONE) continue;
                            const uint32_t t = ec_ref_tape(&S, ref, base, pbase, S.runs[prun].type, inst);
                            if (t != EC_NONE) {
                                ec_home_of_tape(&S, t, &hr2, &hc2);
                                unite(col, cyc0 + r, hc2, cyc0 + hr2);
                            } else if ((ref >> 28) == EC_K_IN) {
                                const uint32_t k = ref & 0xFFFF, h = S.in_home[k];
                                if (run == 0 && row == (h >> 8) && col == (h & 0xFF)) {
                                    const uint32_t op = 1 + k / 32, cell = NLQ_MEM_NIBBLE0 + k % 32;
                                    unite(col, cyc0 + r, cell % G, NLQ_ROW(ns, cycles, nlq_op_row0(qd, G, op) + cell / G, c));
                                } else unite(col, cyc0 + r, h & 0xFF, cyc0 + (h >> 8));
                            }
                        }
                    }
                    for (uint32_t k = 0; k < EK_FREE_PER_CYCLE; k++) {
                        if (!free_home(k, &cb, &rb)) continue;
                        const uint32_t t = k < 64 ? S.runs[EC_NUM_RUNS - 1].tape0 + S.key_byte[k] : S.globs[k == EK_FREE_MASK ? EC_GL_MASK : EC_GL_OK];
                        uint32_t hr2, hc2;
                        ec_home_of_tape(&S, t, &hr2, &hc2);
                        unite(cb, (uint64_t)c * ns->rows_per_cycle + rb, hc2, cyc0 + hr2);
                    }
                }
            }
        }
    }
    if (netlist) {
        // the closed-form section (include/zkw_netlist_closed_form.h): a tie's a / b cells are copies of words, its digits of the registers;
        // a permutation's inputs are copies of words / of the permutation before; the PI row's cells of the last permutation's outputs
        const uint32_t cycles = nl_cycles_of(circuit_type, cap);
        const nl_spec* ns = nl_host_spec(circuit_type);
        const nlcf_desc* cd = nlcf_desc_of(circuit_type);
        const nlq_desc* qd = nlq_desc_of(circuit_type);
        const uint32_t G = ns->g;
        const uint64_t c0 = nlcf_first_row(circuit_type, ns, cycles), nb = NL_BOUNDARY_ROW(ns, cycles), brows = NL_BND_ROWS(ns);
        auto hcell = [&](uint32_t k, uint64_t* col, uint64_t* row) { *col = k % G; *row = c0 + k / G; };
        auto pcell = [&](uint32_t perm, uint32_t v, uint64_t* col, uint64_t* row) { *col = v % G; *row = c0 + nlcf_perm_row0(cd, G, perm) + v / G; };
        uint64_t ca, ra, cb, rb;
        for (uint32_t gi = 0; cd && gi < cd->n_groups; gi++) {
            const nlcf_group& gr = cd->g[gi];
            for (uint32_t j = 0; j < gr.count; j++) {
                const uint32_t c = nlcf_tie_cell0(cd, gi, j);
                const int32_t wa = nlcf_tie_word(&gr, gr.a_word0, j), wb = nlcf_tie_word(&gr, gr.b_word0, j);
                if (wa >= 0) { hcell(c, &ca, &ra); hcell(nlcf_word_cell(cd, nlcf_a_part(&gr), (uint32_t)wa), &cb, &rb); unite(ca, ra, cb, rb); }
                if (wb >= 0) { hcell(c + 1, &ca, &ra); hcell(nlcf_word_cell(cd, nlcf_b_part(&gr), (uint32_t)wb), &cb, &rb); unite(ca, ra, cb, rb); }
                for (uint32_t t = 0; t < gr.n_cells; t++) {
                    hcell(c + 2 + t, &ca, &ra);
                    if (gr.reg_kind == NLCF_REG_FO_WORD) hcell(nlcf_word_cell(cd, NLCF_FO, gr.reg0 + j), &cb, &rb);  // a word of the FSM output: a cell of the header block
                    else if (gr.reg_kind == NLCF_REG_QUEUE_BEFORE || gr.reg_kind == NLCF_REG_QUEUE_AFTER) { cb = nlq_bnd_col(qd, gr.queue, gr.reg_kind == NLCF_REG_QUEUE_AFTER, gr.reg0 + j); rb = NLQ_BASE(ns, cycles); }
                    else if (gr.reg_kind == NLCF_REG_OP_FIRST || gr.reg_kind == NLCF_REG_OP_LAST) {  // a cell / an enable of an operation of the queue section in cycle 0 / the last cycle
                        const uint32_t op = t == 0 ? gr.queue : t == 1 ? gr.gate : gr.gate2, cell = t == 0 ? gr.reg0 : 0;
                        const uint32_t cyc = gr.reg_kind == NLCF_REG_OP_LAST ? cycles - 1 : 0;
                        if (op == NLCF_GATE_ACTIVE) { cb = NL_HDR_IDLE; rb = (uint64_t)cyc * ns->rows_per_cycle; }  // the cycle's idle bit
                        else { cb = cell % G; rb = NLQ_ROW(ns, cycles, nlq_op_row0(qd, G, op) + cell / G, cyc); }
                    } else { const uint32_t e = (gr.reg0 + j) * gr.n_cells + t; cb = e % G; rb = nb + (gr.reg_kind == NLCF_REG_STATE_OUT ? brows : 0) + e / G; }
                    unite(ca, ra, cb, rb);
                }
            }
        }
        const uint32_t cp0 = cd ? nlcf_perm0(cd, 4) : 0;
        for (uint32_t perm = 0; cd && perm < nlcf_n_perms(cd); perm++) {
            uint32_t part = 4, q = perm - cp0, n = NLCF_CP_WORDS;
            if (perm < cp0) { part = 0; while (perm >= nlcf_perm0(cd, part + 1)) part++; q = perm - nlcf_perm0(cd, part); n = cd->n[part]; }
            for (uint32_t v = 0; v < 12; v++) {
                pcell(perm, v, &ca, &ra);
                if (v >= 8) { if (!q) continue; pcell(perm - 1, 118 + v, &cb, &rb); }
                else {
                    const uint32_t w = 8 * q + v;
                    if (w >= n) continue;
                    if (part < 4) hcell(nlcf_word_cell(cd, part, w), &cb, &rb);
                    else if (w < 2) hcell(w, &cb, &rb);
                    else { const uint32_t cpart = (w - 2) / 4; if (!cd->n[cpart]) continue; pcell(nlcf_perm0(cd, cpart + 1) - 1, 118 + (w - 2) % 4, &cb, &rb); }
                }
                unite(ca, ra, cb, rb);
            }
        }
        if (cd)
            for (uint32_t k = 0; k < 4; k++) { pcell(nlcf_n_perms(cd) - 1, 118 + k, &cb, &rb); unite(k, NL_PI_ROW(ns, cycles), cb, rb); }
    }
    for (int l = 0; l < sp.num_links; l++) {
        const rc_link k = sp.links[l];
        if (k.kind == 3) { unite(k.col_a, bnd + sp.off_bout, k.col_b, (uint64_t)k.row_b * rs + cap - 1); continue; }
        if (k.kind == 4) { unite(k.col_a, brow(k.row_a), k.col_b, bnd + sp.off_bout); continue; }
        if (k.kind == 5) { unite(k.col_a, brow(k.row_a), k.col_b, brow(k.row_b)); continue; }
        for (uint32_t i = 0; i < cap; i++) {
            const uint64_t ra = (uint64_t)k.row_a * rs + i;
            if (k.kind == 0) unite(k.col_a, ra, k.col_b, (uint64_t)k.row_b * rs + i);
            else if (k.kind == 1) { if (i) unite(k.col_a, ra, k.col_b, (uint64_t)k.row_b * rs + i - 1); else unite(k.col_a, ra, k.bin_col, bnd + sp.off_bin); }
            else unite(k.col_a, ra, k.col_b, bnd + sp.off_bin);
        }
    }
    if (out_of_range) return fail(ZKW_ERR_INVALID, "zkw_setup_copy_permutation: a link of the spec leaves the copy-permutation columns");
    // cycles: the cells of a class in increasing order, the last one back to the root
    std::vector<uint32_t> last(n_cells);
    for (size_t i = 0; i < n_cells; i++) { sigma[i] = i; last[i] = (uint32_t)i; }
    for (size_t i = 0; i < n_cells; i++) {
        const uint32_t r = find((uint32_t)i);
        if (r == i) continue;
        sigma[last[r]] = i;  // i > last[r]: cells are visited in increasing order
        last[r] = (uint32_t)i;
        sigma[i] = r;
    }
    return ZKW_OK;
}

// copy-permutation check through sigma columns (zkw_setup_copy_permutation): trace[cell] == trace[sigma[cell]] for every cell
__global__ __launch_bounds__(256) void k_check_sigma(const u64* __restrict__ trace, const u64* __restrict__ sigma, size_t n_cells, CheckResult* res,
                                                     size_t n_rows) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_cells; i += stride) {
        const u64 j = sigma[i];
        if (j >= n_cells || trace[i] != trace[j]) flag_bad(res, 4, i / n_rows, i % n_rows);
    }
}
extern "C" int zkw_check_copy_permutation(zkw_ctx* ctx, const zkw_trace* t, size_t slot, const uint64_t* sigma, uint32_t n_columns,
                                          uint64_t* n_violations, uint64_t* first_bad) {
    if (!ctx || !t || !sigma || !n_violations || t->ctx->device != ctx->device || slot >= t->n_slots || n_columns == 0 || n_columns > t->n_cols)
        return fail(ZKW_ERR_INVALID, "zkw_check_copy_permutation: bad argument");
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t n_cells = (size_t)n_columns * t->n_rows;
    const u64* d_sigma = nullptr;
    ZKW_TRY(ctx->in("sigma", sigma, n_cells, &d_sigma));
    CheckResult* d_res = nullptr;
    ZKW_TRY(ctx->scratch_t<CheckResult>("check_res", 1, &d_res));
    CheckResult init{0ull, ~0ull};
    HIP_TRY(hipMemcpyAsync(d_res, &init, sizeof init, hipMemcpyHostToDevice, ctx->stream));
    { Prof _p(ctx, "k_check_sigma"); hipLaunchKernelGGL(k_check_sigma, dim3(2048), dim3(256), 0, ctx->stream, t->data + slot * t->slot_elems(), d_sigma, n_cells, d_res, t->n_rows); }
    ZKW_TRY(launch_check("k_check_sigma"));
    CheckResult res;
    ZKW_TRY(ctx->read_small(&res, d_res, sizeof res));
    *n_violations = res.violations;
    if (first_bad) *first_bad = res.violations ? res.first_bad : 0;
    return ZKW_OK;
}

// Setup side, the lookup tables as columns: what the reference's setup keeps as table polynomials (add_tables of the wrappers, e.g.
// base_layer/sha256_round_function.rs:116-150; boojum's create_*_table, absent). Row t of the stacked table = row t of the multiplicity
// column: the table's cells (inputs, then outputs: `width` columns) and a table-id column (index in the circuit's table list + 1; 0 = no
// table row). Queue circuits: the one 8-bit range table (width 1). Netlist circuits: their table list; ECRecover's FixedBaseMul<i, C>
// tables come from ec_build_fixed_tables. cols = NULL only reports *n_columns (= width + 1).
extern "C" int zkw_setup_lookup_tables(uint8_t circuit_type, size_t n_rows, uint64_t* cols, uint32_t* n_columns) {
    if (!n_columns) return fail(ZKW_ERR_INVALID, "zkw_setup_lookup_tables: null argument");
    const nl_spec* ns = nl_host_spec(circuit_type);
    LinkSpec sp;
    if (!ns && !link_spec_of(circuit_type, &sp)) return fail(ZKW_ERR_INVALID, "zkw_setup_lookup_tables: circuit type %u has no layout in this library", (unsigned)circuit_type);
    const uint32_t width = ns ? ns->w : 1;
    *n_columns = width + 1;
    if (!cols) return ZKW_OK;
    const size_t total = ns ? ns->total_table_rows : 256;
    if (total > n_rows) return fail(ZKW_ERR_INVALID, "zkw_setup_lookup_tables: %zu table rows need more than %zu rows", total, n_rows);
    memset(cols, 0, (size_t)(width + 1) * n_rows * sizeof(uint64_t));
    if (!ns) {
        for (size_t r = 0; r < 256; r++) { cols[r] = r; cols[n_rows + r] = 1; }
        return ZKW_OK;
    }
    std::vector<uint32_t> fixed;
    for (uint32_t k = 0; k < ns->n_tables; k++) {
        const nl_table& t = ns->tables[k];
        if (t.fn > NL_FN_SPLIT4 && fixed.empty()) { fixed.resize(EC_FIXED_WORDS); ec_build_fixed_tables(fixed.data()); }  // (fn 8: FixedBaseMul, param = 8 C + i)
        for (uint32_t key = 0; key < t.rows; key++) {
            uint32_t a[3] = {0, 0, 0}, o[3] = {0, 0, 0};
            for (uint32_t i = 0; i < t.n_in; i++) a[i] = (key >> (t.in_bits * i)) & ((1u << t.in_bits) - 1);
            if (t.fn <= NL_FN_SPLIT4) nl_table_eval(t.fn, t.param, a, o);
            else { o[0] = fixed[((size_t)t.param * 256 + key) * 2]; o[1] = fixed[((size_t)t.param * 256 + key) * 2 + 1]; }
            const size_t row = (size_t)t.offset + key;
            uint32_t c = 0;
            for (uint32_t i = 0; i < t.n_in; i++) cols[(size_t)c++ * n_rows + row] = a[i];
            for (uint32_t i = 0; i < t.n_out && c < width; i++) cols[(size_t)c++ * n_rows + row] = o[i];
            cols[(size_t)width * n_rows + row] = k + 1;
        }
    }
    return ZKW_OK;
}

