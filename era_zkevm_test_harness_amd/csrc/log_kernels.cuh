// log_kernels.cuh — encodings and the 4-wide ("log") queue of the sorter / demux circuits (gfx950).
//
// Reference functions replaced:
//   k_encode_log        LogQuery::encoding_witness (+ extended timestamp)  circuit_encodings/src/log_query.rs:102-427
//   k_encode_decommit   DecommittmentQuery::encoding_witness               circuit_encodings/src/decommittment_request.rs:9-74
//   k_log_prehash + k_chain_log   QueueSimulator::push_and_output_intermediate_data   circuit_encodings/src/lib.rs:179-221
#pragma once
#include "ram_kernels.cuh"

namespace zkw {

__device__ __forceinline__ u32 le_byte(const u32* limbs, int i) { return (limbs[i >> 2] >> (8 * (i & 3))) & 0xFF; }

// 128-byte record in, 20 field elements out. Rider bytes: key[0..32] then address[0..19] (both
// little-endian byte order), three per element at bit 32 / 40 / 48 of v0..v16; address byte 19 rides in v17.
__device__ __forceinline__ void encode_log_query(const zkw_log_query& q, bool has_ext, u32 ext_ts, u64 out[20]) {
#pragma unroll
    for (int k = 0; k < 17; k++) {
        u64 base = k < 8 ? q.read_value[k] : (k < 16 ? q.written_value[k - 8] : q.timestamp);
        u64 r = 0;
#pragma unroll
        for (int t = 0; t < 3; t++) {
            const int i = 3 * k + t;  // rider index
            const u32 b = i < 32 ? le_byte(q.key, i) : le_byte(q.address, i - 32);
            r |= (u64)b << (32 + 8 * t);
        }
        out[k] = base | r;
    }
    out[17] = (u64)q.tx_number_in_block | ((u64)le_byte(q.address, 19) << 32) | ((u64)q.aux_byte << 40) | ((u64)q.shard_id << 48);
    out[18] = (u64)(q.rw_flag ? 1 : 0) + 2 * (u64)(q.is_service ? 1 : 0);
    out[19] = (q.rollback ? 1 : 0) + (has_ext ? ((u64)ext_ts << ZKW_EXTENDED_TIMESTAMP_ENCODING_OFFSET) : 0);
    static_assert(ZKW_EXTENDED_TIMESTAMP_ENCODING_ELEMENT == 19, "extended timestamp rides in element 19");
}

static __device__ __forceinline__ void k_encode_log(const VB& vb, const zkw_log_query* __restrict__ q, size_t n,
                                                    const u32* __restrict__ ext_ts, u64* __restrict__ enc) {
    size_t i = (size_t)vb.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    zkw_log_query m;
    const uint4* src = reinterpret_cast<const uint4*>(q + i);
    uint4* dst = reinterpret_cast<uint4*>(&m);
#pragma unroll
    for (int k = 0; k < 8; k++) dst[k] = src[k];
    u64 e[20];
    encode_log_query(m, ext_ts != nullptr, ext_ts ? ext_ts[i] : 0, e);
    ulonglong2* o = reinterpret_cast<ulonglong2*>(enc + 20 * i);
#pragma unroll
    for (int k = 0; k < 10; k++) o[k] = make_ulonglong2(e[2 * k], e[2 * k + 1]);
}

__device__ __forceinline__ void encode_decommit_query(const zkw_decommit_query& d, u64 out[8]) {
    const u64 p = d.memory_page, t = d.timestamp;
    out[0] = (u64)d.hash[0] | ((p & 0xFFFFFF) << 32);
    out[1] = (u64)d.hash[1] | ((p >> 24) << 32) | ((t & 0xFFFF) << 40);
    out[2] = (u64)d.hash[2] | ((t >> 16) << 32) | ((u64)(d.is_fresh ? 1 : 0) << 48);
#pragma unroll
    for (int k = 3; k < 8; k++) out[k] = d.hash[k];
}

static __device__ __forceinline__ void k_encode_decommit(const VB& vb, const zkw_decommit_query* __restrict__ q, size_t n,
                                                         u64* __restrict__ enc) {
    size_t i = (size_t)vb.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    zkw_decommit_query m;
    const uint4* src = reinterpret_cast<const uint4*>(q + i);
    uint4* dst = reinterpret_cast<uint4*>(&m);
    dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];
    u64 e[8];
    encode_decommit_query(m, e);
    ulonglong2* o = reinterpret_cast<ulonglong2*>(enc + 8 * i);
#pragma unroll
    for (int k = 0; k < 4; k++) o[k] = make_ulonglong2(e[2 * k], e[2 * k + 1]);
}

// ------------------------------------------------------------------------------------------------
// 4-wide queue push: new_tail = first 4 words of the 3-round overwrite sponge over enc(20) || old_tail(4)
// started from the zero state. Only the THIRD round sees the previous tail, so rounds 1-2 of every item
// are computed in parallel (k_log_prehash: one item per lane, keeps the 4 capacity words) and the serial
// chain is one permutation per item (k_chain_log, one chain per 16-lane DPP row).
static __device__ __forceinline__ void k_log_prehash(const VB& vb, const u64* __restrict__ enc /* [n][20] */, size_t n,
                                                     u64* __restrict__ pre /* [n][4] */) {
    size_t i = (size_t)vb.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 s[12];
    const ulonglong2* src = reinterpret_cast<const ulonglong2*>(enc + 20 * i);
#pragma unroll
    for (int k = 0; k < 4; k++) { ulonglong2 w = src[k]; s[2 * k] = w.x; s[2 * k + 1] = w.y; }
    s[8] = s[9] = s[10] = s[11] = 0;
    p2::permute(s);
#pragma unroll
    for (int k = 0; k < 4; k++) { ulonglong2 w = src[4 + k]; s[2 * k] = w.x; s[2 * k + 1] = w.y; }
    p2::permute(s);
    ulonglong2* o = reinterpret_cast<ulonglong2*>(pre + 4 * i);
    o[0] = make_ulonglong2(gl::canon(s[8]), gl::canon(s[9]));
    o[1] = make_ulonglong2(gl::canon(s[10]), gl::canon(s[11]));
}

struct LogChainJob {
    const u64* enc;      // [n][20]
    const u64* pre;      // [n][4] capacity after rounds 1-2
    u64* old_tails;      // [n][4] or nullptr — what the reference keeps in `witness` (lib.rs:204)
    u64* new_tails;      // [n][4]
    const u64* tail_in;  // [4] or nullptr
    u64 n;
};

template <int WAVES>  // 4: `k_chain_log_x4`, one workgroup per CU, a wave per SIMD (ram_kernels.cuh, k_chain_full_x4)
static __device__ __forceinline__ void chain_log_body(const LogChainJob* __restrict__ jobs, int n_jobs) {
    const int lane = threadIdx.x & 63, g = lane & 15;
    const int chain = (blockIdx.x * WAVES + (int)(threadIdx.x >> 6)) * 4 + (lane >> 4);
    p2::Coop co;
    co.init(g);
    LogChainJob job;
    job.enc = nullptr; job.pre = nullptr; job.old_tails = nullptr; job.new_tails = nullptr; job.tail_in = nullptr; job.n = 0;
    if (chain < n_jobs) job = jobs[chain];
    // lanes 0-3: enc[16..20]; lanes 4-7: the running tail; lanes 8-11: prehash capacity
    u64 tail = (g >= 4 && g < 8 && job.tail_in) ? job.tail_in[g - 4] : 0;
    auto fetch = [&](u64 i) -> u64 {
        if (g < 4) return job.enc[20 * i + 16 + g];
        if (g >= 8 && g < 12) return job.pre[4 * i + (g - 8)];
        return 0;
    };
    u64 nxt = job.n > 0 ? fetch(0) : 0;
    for (u64 i = 0; __any(i < job.n); i++) {
        const bool live = i < job.n;
        const u64 cur = nxt;
        if (i + 1 < job.n) nxt = fetch(i + 1);
        if (live && job.old_tails && g >= 4 && g < 8) job.old_tails[4 * i + (g - 4)] = gl::canon(tail);
        const u64 x = (g >= 4 && g < 8) ? tail : cur;
        const u64 y = co.permute(x);
        // the new tail is state[0..4] (lanes 0-3); move it to the tail lanes 4-7 for the next item
        const u64 moved = p2::dpp64<0x124>(y);  // row_ror:4 — lane L reads lane L-4 (mod 16)
        if (live) {
            if (g < 4) job.new_tails[4 * i + g] = gl::canon(y);
            if (g >= 4 && g < 8) tail = moved;
        }
    }
}
static __global__ __launch_bounds__(64) void k_chain_log(const LogChainJob* __restrict__ jobs, int n_jobs) { chain_log_body<1>(jobs, n_jobs); }
static __global__ __launch_bounds__(256) void k_chain_log_x4(const LogChainJob* __restrict__ jobs, int n_jobs) { chain_log_body<4>(jobs, n_jobs); }

}  // namespace zkw
