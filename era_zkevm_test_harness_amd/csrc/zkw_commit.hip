// zkw_commit.hip — host side of the setup-as-field-elements path (include/zkw.h "Setup side as field elements"): NTT / LDE plans, the
// Merkle tree with a cap, and zkw_setup_commit = columns of a layout -> monomial form -> LDE -> tree -> cap, the shape of
// create_base_layer_setup_data (src/prover_utils.rs:48-197). Kernels: ntt_kernels.cuh.
#include <vector>

#include "zkw_ctx.h"
#include "ntt_kernels.cuh"

using namespace zkw;

namespace {

constexpr u64 GENERATOR = 7;  // multiplicative generator of Goldilocks; 7^((p-1)/2^32) = 0x185629dcda58878c is boojum's / plonky2's 2^32-th root of unity

u64 root_of_unity(u32 log_n) { return gl::canon(gl::pow(gl::pow(GENERATOR, (gl::P - 1) >> 32), 1ull << (32 - log_n))); }

std::vector<u64> powers(u64 base, size_t count) {
    std::vector<u64> v(count);
    u64 x = 1;
    for (size_t i = 0; i < count; i++) { v[i] = gl::canon(x); x = gl::mul(x, base); }
    return v;
}
// lo[i] = base^i, hi[i] = base^(1024 i), i < 1024, back to back
std::vector<u64> split_powers(u64 base) {
    std::vector<u64> v = powers(base, NTT_TW), hi = powers(gl::pow(base, NTT_TW), NTT_TW);
    v.insert(v.end(), hi.begin(), hi.end());
    return v;
}

template <class K>
int allow_lds(K kernel) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    return ZKW_OK;
}

// one transform of n_cols columns: in -> out (device pointers; out may be in; tmp: n_cols * n words when log_n > NTT_MAX_SINGLE).
// shift != 0: input point j is multiplied by shift^j first (evaluation on the coset shift * <w>). `tag` separates the table uploads of
// transforms that are in flight on the stream at the same time.
int run_ntt(zkw_ctx* ctx, const u64* in, u64* out, u64* tmp, u32 log_n, size_t n_cols, bool inverse, u64 shift, const char* tag) {
    const size_t n = (size_t)1 << log_n;
    u64 w = root_of_unity(log_n);
    if (inverse) w = gl::canon(gl::inv(w));
    NttArgs A;
    memset(&A, 0, sizeof A);
    A.log_n = log_n;
    A.post = inverse ? gl::canon(gl::inv((u64)n % gl::P)) : 1;
    const bool single = log_n <= NTT_MAX_SINGLE;
    A.a = single ? log_n : log_n / 2;
    A.b = log_n - A.a;
    std::vector<u64> tab;  // [tw_a | tw_b | tw_lo, tw_hi | pre_lo, pre_hi]
    const size_t na = A.a ? (size_t)1 << (A.a - 1) : 1, nb = A.b ? (size_t)1 << (A.b - 1) : 1;
    {
        std::vector<u64> ta = powers(gl::pow(w, n >> A.a), na), tb = powers(gl::pow(w, n >> A.b), nb), tw = split_powers(w);
        tab.insert(tab.end(), ta.begin(), ta.end());
        tab.insert(tab.end(), tb.begin(), tb.end());
        tab.insert(tab.end(), tw.begin(), tw.end());
        if (shift) { std::vector<u64> pre = split_powers(shift); tab.insert(tab.end(), pre.begin(), pre.end()); }
    }
    u64* d_tab = nullptr;
    ZKW_TRY(ctx->upload(tag, tab, &d_tab));
    A.tw_a = d_tab; A.tw_b = d_tab + na; A.tw_lo = d_tab + na + nb; A.tw_hi = A.tw_lo + NTT_TW;
    if (shift) { A.pre_lo = A.tw_hi + NTT_TW; A.pre_hi = A.pre_lo + NTT_TW; }
    if (single) {
        ZKW_TRY(allow_lds(k_ntt_single));
        A.in = in; A.out = out;
        const size_t lds = (n + 1 + n / 2 + 1) * sizeof(u64);
        Prof _p(ctx, "k_ntt_single");
        hipLaunchKernelGGL(k_ntt_single, dim3(1, (unsigned)n_cols), dim3(NTT_THREADS), lds, ctx->stream, A);
        return launch_check("k_ntt_single");
    }
    ZKW_TRY(allow_lds(k_ntt_pass1));
    ZKW_TRY(allow_lds(k_ntt_pass2));
    const u32 n1 = 1u << A.a, n2 = 1u << A.b, T1 = NTT_TILE / n1, T2 = NTT_TILE / n2;
    {
        NttArgs P1 = A;
        P1.in = in; P1.out = tmp;
        const size_t lds = ((size_t)T1 * (n1 + 1) + n1 / 2) * sizeof(u64);
        Prof _p(ctx, "k_ntt_pass1");
        hipLaunchKernelGGL(k_ntt_pass1, dim3(n2 / T1, (unsigned)n_cols), dim3(NTT_THREADS), lds, ctx->stream, P1);
        ZKW_TRY(launch_check("k_ntt_pass1"));
    }
    {
        NttArgs P2 = A;
        P2.in = tmp; P2.out = out;
        const size_t lds = ((size_t)T2 * (n2 + 1) + n2 / 2) * sizeof(u64);
        Prof _p(ctx, "k_ntt_pass2");
        hipLaunchKernelGGL(k_ntt_pass2, dim3(n1 / T2, (unsigned)n_cols), dim3(NTT_THREADS), lds, ctx->stream, P2);
        ZKW_TRY(launch_check("k_ntt_pass2"));
    }
    return ZKW_OK;
}

bool ntt_size_ok(u32 log_n) { return log_n <= NTT_MAX_SINGLE || (log_n <= 20 && log_n / 2 >= 3); }

// device: values on the domain [n_cols][n] -> out [lde][n_cols][n] (coset c: points GENERATOR * w_(lde n)^c * w_n^i); coeffs / tmp: n_cols * n words each
int run_lde(zkw_ctx* ctx, const u64* values, u32 log_n, size_t n_cols, u32 lde_factor, u64* coeffs, u64* tmp, u64* out) {
    const size_t n = (size_t)1 << log_n;
    ZKW_TRY(run_ntt(ctx, values, coeffs, tmp, log_n, n_cols, true, 0, "ntt_tab_inv"));
    u32 log_lde = 0;
    while ((1u << log_lde) < lde_factor) log_lde++;
    const u64 gamma = root_of_unity(log_n + log_lde);
    static const char* TAGS[8] = {"ntt_tab_c0", "ntt_tab_c1", "ntt_tab_c2", "ntt_tab_c3", "ntt_tab_c4", "ntt_tab_c5", "ntt_tab_c6", "ntt_tab_c7"};
    for (u32 c = 0; c < lde_factor; c++) {
        const u64 shift = gl::canon(gl::mul(GENERATOR, gl::pow(gamma, c)));
        ZKW_TRY(run_ntt(ctx, coeffs, out + (size_t)c * n_cols * n, tmp, log_n, n_cols, false, shift, TAGS[c]));
    }
    return ZKW_OK;
}

// device: leaf columns [n_sets][n_cols][n] -> tree levels (leaves first) in `tree` (4 * (2 * n_sets * n - cap_size) words), the cap is its last level
int run_merkle(zkw_ctx* ctx, const u64* cols, size_t n_sets, size_t n_cols, size_t n, u32 cap_size, u64* tree) {
    const size_t n_leaves = n_sets * n;
    for (size_t s = 0; s < n_sets; s++) {
        Prof _p(ctx, "k_merkle_leaves");
        hipLaunchKernelGGL(k_merkle_leaves, dim3(blocks_for(n, 256)), dim3(256), 0, ctx->stream, cols + s * n_cols * n, n_cols, n, n, tree + 4 * s * n);
        ZKW_TRY(launch_check("k_merkle_leaves"));
    }
    u64* below = tree;
    for (size_t m = n_leaves / 2; m >= cap_size; m /= 2) {
        u64* level = below + 8 * m;
        Prof _p(ctx, "k_merkle_nodes");
        hipLaunchKernelGGL(k_merkle_nodes, dim3(blocks_for(m, 256)), dim3(256), 0, ctx->stream, below, m, level);
        ZKW_TRY(launch_check("k_merkle_nodes"));
        below = level;
        if (m == 1) break;
    }
    return ZKW_OK;
}

bool pow2(size_t x) { return x && !(x & (x - 1)); }
size_t tree_words(size_t n_leaves, u32 cap_size) { return 4 * (2 * n_leaves - cap_size); }

}  // namespace

extern "C" int zkw_ntt(zkw_ctx* ctx, const uint64_t* in, uint64_t* out, uint32_t log_n, size_t n_cols, int inverse) {
    if (!ctx || !in || !out || !ntt_size_ok(log_n)) return fail(ZKW_ERR_INVALID, "zkw_ntt: bad argument (log_n <= 20)");
    if (n_cols == 0) return ZKW_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t total = n_cols << log_n;
    const u64* d_in = nullptr;
    u64 *d_out = nullptr, *d_tmp = nullptr;
    ZKW_TRY(ctx->in("ntt_in", in, total, &d_in));
    ZKW_TRY(ctx->out("ntt_out", out, total, &d_out));
    if (log_n > NTT_MAX_SINGLE) ZKW_TRY(ctx->scratch_t<u64>("ntt_tmp", total, &d_tmp));
    ZKW_TRY(run_ntt(ctx, d_in, d_out, d_tmp, log_n, n_cols, inverse != 0, 0, "ntt_tab"));
    ZKW_TRY(ctx->finish_out(out, d_out, total));
    return ctx->sync_if_host();
}

extern "C" int zkw_lde(zkw_ctx* ctx, const uint64_t* values, uint32_t log_n, size_t n_cols, uint32_t lde_factor, uint64_t* out) {
    if (!ctx || !values || !out || !ntt_size_ok(log_n) || !pow2(lde_factor) || lde_factor > 8)
        return fail(ZKW_ERR_INVALID, "zkw_lde: bad argument (log_n <= 20, lde_factor a power of two <= 8)");
    if (n_cols == 0) return ZKW_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t total = n_cols << log_n;
    const u64* d_in = nullptr;
    u64 *d_out = nullptr, *d_tmp = nullptr, *d_coeffs = nullptr;
    ZKW_TRY(ctx->in("lde_in", values, total, &d_in));
    ZKW_TRY(ctx->out("lde_out", out, total * lde_factor, &d_out));
    ZKW_TRY(ctx->scratch_t<u64>("ntt_tmp", total, &d_tmp));
    ZKW_TRY(ctx->scratch_t<u64>("lde_coeffs", total, &d_coeffs));
    ZKW_TRY(run_lde(ctx, d_in, log_n, n_cols, lde_factor, d_coeffs, d_tmp, d_out));
    ZKW_TRY(ctx->finish_out(out, d_out, total * lde_factor));
    return ctx->sync_if_host();
}

extern "C" size_t zkw_merkle_tree_words(size_t n_leaves, uint32_t cap_size) { return tree_words(n_leaves, cap_size); }

extern "C" int zkw_merkle_tree_with_cap(zkw_ctx* ctx, const uint64_t* leaf_cols, size_t n_sets, size_t n_cols, size_t n, uint32_t cap_size,
                                        uint64_t* cap, uint64_t* tree) {
    if (!ctx || !leaf_cols || !cap || !pow2(n_sets * n) || !pow2(cap_size) || cap_size > n_sets * n || n_cols == 0)
        return fail(ZKW_ERR_INVALID, "zkw_merkle_tree_with_cap: bad argument (leaves and cap powers of two, cap <= leaves)");
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t n_leaves = n_sets * n, words = tree_words(n_leaves, cap_size);
    const u64* d_cols = nullptr;
    u64* d_tree = nullptr;
    ZKW_TRY(ctx->in("merkle_cols", leaf_cols, n_sets * n_cols * n, &d_cols));
    if (tree) ZKW_TRY(ctx->out("merkle_tree", tree, words, &d_tree));
    else ZKW_TRY(ctx->scratch_t<u64>("merkle_tree", words, &d_tree));
    ZKW_TRY(run_merkle(ctx, d_cols, n_sets, n_cols, n, cap_size, d_tree));
    const u64* d_cap = d_tree + words - 4 * (size_t)cap_size;
    if (ctx->ptr_mode == ZKW_PTR_DEVICE) {
        HIP_TRY(hipMemcpyAsync(cap, d_cap, 32 * (size_t)cap_size, hipMemcpyDeviceToDevice, ctx->stream));
    } else {
        if (tree) ZKW_TRY(ctx->finish_out(tree, d_tree, words));
        HIP_TRY(hipMemcpyAsync(cap, d_cap, 32 * (size_t)cap_size, hipMemcpyDeviceToHost, ctx->stream));
    }
    return ctx->sync_if_host();
}

// The setup columns of one of this library's layouts as field elements, on the device: the sigma columns (copy permutation: cell -> k_col' *
// w^row', coset representatives k_j = 7^j), then ONE selector column (zkw_setup_row_selectors: the byte as a field element).
extern "C" int zkw_setup_num_columns(uint8_t circuit_type, uint32_t* n_columns) {
    if (!n_columns) return fail(ZKW_ERR_INVALID, "zkw_setup_num_columns: null argument");
    uint32_t g = 0, tc = 0;
    ZKW_TRY(zkw_setup_copy_permutation(circuit_type, 0, 0, nullptr, &g));
    ZKW_TRY(zkw_setup_lookup_tables(circuit_type, 0, nullptr, &tc));
    *n_columns = g + 1 + tc;
    return ZKW_OK;
}

static int setup_columns_device(zkw_ctx* ctx, uint8_t circuit_type, uint32_t capacity, uint32_t log_n, u64* d_cols, uint32_t n_setup_cols) {
    const size_t n = (size_t)1 << log_n;
    uint32_t tc = 0;
    ZKW_TRY(zkw_setup_lookup_tables(circuit_type, 0, nullptr, &tc));
    const uint32_t G = n_setup_cols - 1 - tc;
    std::vector<uint64_t> tables((size_t)tc * n);
    ZKW_TRY(zkw_setup_lookup_tables(circuit_type, n, tables.data(), &tc));
    HIP_TRY(hipMemcpyAsync(d_cols + (size_t)(G + 1) * n, tables.data(), tables.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    std::vector<uint64_t> sigma((size_t)G * n);
    uint32_t g2 = 0;
    ZKW_TRY(zkw_setup_copy_permutation(circuit_type, capacity, n, sigma.data(), &g2));
    std::vector<uint8_t> sel(n);
    ZKW_TRY(zkw_setup_row_selectors(circuit_type, capacity, n, sel.data()));
    u64 *d_sigma = nullptr, *d_om = nullptr, *d_tab = nullptr;
    uint8_t* d_sel = nullptr;
    ZKW_TRY(ctx->scratch_t<u64>("setup_sigma_idx", sigma.size(), &d_sigma));
    ZKW_TRY(ctx->scratch_t<u64>("setup_omega", n, &d_om));
    ZKW_TRY(ctx->scratch_t<uint8_t>("setup_sel", n, &d_sel));
    HIP_TRY(hipMemcpyAsync(d_sigma, sigma.data(), sigma.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(d_sel, sel.data(), n, hipMemcpyHostToDevice, ctx->stream));
    std::vector<u64> tab = split_powers(root_of_unity(log_n)), reps = powers(GENERATOR, G);
    tab.insert(tab.end(), reps.begin(), reps.end());
    ZKW_TRY(ctx->upload("setup_tab", tab, &d_tab));
    { Prof _p(ctx, "k_powers"); hipLaunchKernelGGL(k_powers, dim3(blocks_for(n, 256)), dim3(256), 0, ctx->stream, d_tab, d_tab + NTT_TW, n, d_om); }
    ZKW_TRY(launch_check("k_powers"));
    { Prof _p(ctx, "k_sigma_to_field"); hipLaunchKernelGGL(k_sigma_to_field, dim3(4096), dim3(256), 0, ctx->stream, d_sigma, (size_t)G * n, log_n, d_om, d_tab + 2 * NTT_TW, d_cols); }
    ZKW_TRY(launch_check("k_sigma_to_field"));
    { Prof _p(ctx, "k_bytes_to_field"); hipLaunchKernelGGL(k_bytes_to_field, dim3(blocks_for(n, 256)), dim3(256), 0, ctx->stream, d_sel, n, d_cols + (size_t)G * n); }
    ZKW_TRY(launch_check("k_bytes_to_field"));
    HIP_TRY(hipStreamSynchronize(ctx->stream));  // the host vectors above are the copies' sources
    return ZKW_OK;
}

extern "C" int zkw_setup_columns(zkw_ctx* ctx, uint8_t circuit_type, uint32_t capacity, uint32_t log_n, uint64_t* columns) {
    uint32_t nc = 0;
    if (!ctx || !columns || log_n > 20) return fail(ZKW_ERR_INVALID, "zkw_setup_columns: bad argument (log_n <= 20: the twiddle tables of k_powers hold 1024 x 1024 exponents)");
    ZKW_TRY(zkw_setup_num_columns(circuit_type, &nc));
    HIP_TRY(hipSetDevice(ctx->device));
    u64* d_cols = nullptr;
    ZKW_TRY(ctx->out("setup_cols", columns, (size_t)nc << log_n, &d_cols));
    ZKW_TRY(setup_columns_device(ctx, circuit_type, capacity, log_n, d_cols, nc));
    ZKW_TRY(ctx->finish_out(columns, d_cols, (size_t)nc << log_n));
    return ctx->sync_if_host();
}

extern "C" int zkw_setup_commit(zkw_ctx* ctx, uint8_t circuit_type, uint32_t capacity, uint32_t log_n, uint32_t lde_factor, uint32_t cap_size,
                                uint64_t* cap) {
    uint32_t nc = 0;
    if (!ctx || !cap || !ntt_size_ok(log_n) || !pow2(lde_factor) || lde_factor > 8 || !pow2(cap_size) || cap_size > ((size_t)lde_factor << log_n))
        return fail(ZKW_ERR_INVALID, "zkw_setup_commit: bad argument");
    ZKW_TRY(zkw_setup_num_columns(circuit_type, &nc));
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t n = (size_t)1 << log_n, total = (size_t)nc * n, words = tree_words(n * lde_factor, cap_size);
    u64 *d_cols = nullptr, *d_coeffs = nullptr, *d_tmp = nullptr, *d_lde = nullptr, *d_tree = nullptr;
    ZKW_TRY(ctx->scratch_t<u64>("setup_cols", total, &d_cols));
    ZKW_TRY(ctx->scratch_t<u64>("lde_coeffs", total, &d_coeffs));
    ZKW_TRY(ctx->scratch_t<u64>("ntt_tmp", total, &d_tmp));
    ZKW_TRY(ctx->scratch_t<u64>("lde_out", total * lde_factor, &d_lde));
    ZKW_TRY(ctx->scratch_t<u64>("merkle_tree", words, &d_tree));
    ZKW_TRY(setup_columns_device(ctx, circuit_type, capacity, log_n, d_cols, nc));
    ZKW_TRY(run_lde(ctx, d_cols, log_n, nc, lde_factor, d_coeffs, d_tmp, d_lde));
    ZKW_TRY(run_merkle(ctx, d_lde, lde_factor, nc, n, cap_size, d_tree));
    HIP_TRY(hipMemcpyAsync(cap, d_tree + words - 4 * (size_t)cap_size, 32 * (size_t)cap_size,
                           ctx->ptr_mode == ZKW_PTR_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, ctx->stream));
    return ctx->sync_if_host();
}
