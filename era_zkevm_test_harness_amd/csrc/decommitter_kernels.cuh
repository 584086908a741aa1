// decommitter_kernels.cuh — CodeDecommitter witness builder on gfx950.
// Reference: compute_decommitter_circuit_snapshots, src/witness/individual_circuits/decommit_code.rs:20-439.
// The reference walks SHA-256 rounds one by one across all requests; here each request hashes its bytecode
// independently (one lane per request, the state after every round is kept), and an instance boundary at
// global round e is located by binary search in the prefix sums of rounds per request.
#pragma once
#include "storage_kernels.cuh"
#include "decommit_kernels.cuh"  // qs12

namespace zkw {

static __constant__ u32 c_sha_k[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
    0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

__device__ __forceinline__ u32 rotr(u32 x, int r) { return (x >> r) | (x << (32 - r)); }

// one SHA-256 compression; w[0..16] = the block as 16 big-endian words
__device__ __forceinline__ void sha256_compress(u32 st[8], u32 w[16]) {
    u32 a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
#pragma unroll
    for (int i = 0; i < 64; i++) {
        if (i >= 16) {
            const u32 w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
            const u32 s0 = rotr(w15, 7) ^ rotr(w15, 18) ^ (w15 >> 3), s1 = rotr(w2, 17) ^ rotr(w2, 19) ^ (w2 >> 10);
            w[i & 15] = w[i & 15] + s0 + w[(i - 7) & 15] + s1;
        }
        const u32 S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25), ch = (e & f) ^ (~e & g);
        const u32 t1 = h + S1 + ch + c_sha_k[i] + w[i & 15];
        const u32 S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22), mj = (a & b) ^ (a & c) ^ (b & c);
        const u32 t2 = S0 + mj;
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}

// what a round of a builder's walk does to the queues of its circuit (the queue section of the netlist circuits,
// netlist_queue_kernels.cuh): the request it belongs to, the index of the first memory query it pushes, how many it pushes,
// flags: 1 = the round pops its request, 2 = the last push is the digest write (sha256); bits 8.. = rounds left after this one (sha256)
struct RoundOps { u32 request, first_query, n_push, flags; };

struct DecommitterJob {
    const zkw_decommit_query* requests;  // [n_requests]
    const u32* words;                    // [total_words][8] LE limbs
    const u64* word_offsets;             // [n_requests + 1] (device), rebased to 0
    const u64* round_offsets;            // [n_requests + 1] (device): prefix sums of rounds per request
    u32* round_states;                   // [total_rounds][8]
    zkw_mem_query* mem_q;                // [total_words]
    u64* mem_enc;                        // [total_words][8]
    u32* violations;
    u64 n_requests;
    zkw_sha256_round_record* sha256_rounds;  // may be null: [total_rounds], the cycles of the CodeDecommitter circuit (type 3)
    RoundOps* round_ops;                     // may be null: [total_rounds]
};

// one lane per request: SHA-256 over its bytecode (two big-endian words per block, padding in the last
// block, decommit_code.rs:286-320), every round's state kept; digest compared with the request's hash
static __device__ __forceinline__ void k_decommitter_sha(const VB& vb, DecommitterJob job) {
    const u64 k = (u64)vb.x * blockDim.x + threadIdx.x;
    if (k >= job.n_requests) return;
    const zkw_decommit_query q = job.requests[k];
    const u64 w0 = job.word_offsets[k], nw = job.word_offsets[k + 1] - w0, r0 = job.round_offsets[k];
    const u32 num_words = q.hash[7] & 0xFFFF;
    if (!(num_words & 1) || num_words != nw || !q.is_fresh) { atomicAdd(job.violations, 1u); return; }
    const u64 rounds = (nw + 1) / 2;
    u32 st[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    for (u64 r = 0; r < rounds; r++) {
        u32 w[16];
        const u32* a = job.words + 8 * (w0 + 2 * r);
#pragma unroll
        for (int j = 0; j < 8; j++) w[j] = a[7 - j];  // U256 big-endian = limbs from the most significant
        if (2 * r + 1 < nw) {
#pragma unroll
            for (int j = 0; j < 8; j++) w[8 + j] = a[8 + 7 - j];
        } else {
            w[8] = 0x80000000u;
#pragma unroll
            for (int j = 9; j < 15; j++) w[j] = 0;
            w[15] = num_words * 32 * 8;  // length_in_bits, 32-bit big-endian at bytes 60..64
        }
        // the cycle of the CodeDecommitter circuit: block as hashed (big-endian words, padding included), reset, state after
        u64* rec = job.sha256_rounds ? reinterpret_cast<u64*>(job.sha256_rounds + r0 + r) : nullptr;  // 104 = 8 * 13 bytes
        if (rec) {
#pragma unroll
            for (int m = 0; m < 8; m++) rec[m] = (u64)__builtin_bswap32(w[2 * m]) | ((u64)__builtin_bswap32(w[2 * m + 1]) << 32);
        }
        sha256_compress(st, w);  // expands the schedule in place
        if (rec) {
            rec[8] = (u64)(r == 0 ? 1u : 0u) | ((u64)st[0] << 32);
            rec[9] = (u64)st[1] | ((u64)st[2] << 32);
            rec[10] = (u64)st[3] | ((u64)st[4] << 32);
            rec[11] = (u64)st[5] | ((u64)st[6] << 32);
            rec[12] = (u64)st[7];
        }
        u32* o = job.round_states + 8 * (r0 + r);
#pragma unroll
        for (int j = 0; j < 8; j++) o[j] = st[j];
        if (job.round_ops) job.round_ops[r0 + r] = RoundOps{(u32)k, (u32)(w0 + 2 * r), 2 * r + 1 < nw ? 2u : 1u, r == 0 ? 1u : 0u};
    }
    bool ok = true;
    for (int j = 1; j < 8; j++) ok &= st[j] == q.hash[7 - j];
    if (!ok) atomicAdd(job.violations, 1u);
}

// one lane per code word: the memory write it becomes (decommit_code.rs:47-78) and its encoding
static __device__ __forceinline__ void k_decommitter_mem_queries(const VB& vb, DecommitterJob job, u64 total_words) {
    const u64 i = (u64)vb.x * blockDim.x + threadIdx.x;
    if (i >= total_words) return;
    u64 lo = 0, hi = job.n_requests;  // largest k with word_offsets[k] <= i
    while (hi - lo > 1) {
        const u64 mid = (lo + hi) >> 1;
        if (job.word_offsets[mid] <= i) lo = mid; else hi = mid;
    }
    const zkw_decommit_query* q = job.requests + lo;
    zkw_mem_query m;
    memset(&m, 0, sizeof m);
    m.timestamp = q->timestamp;
    m.page = q->memory_page;
    m.index = (u32)(i - job.word_offsets[lo]);
    m.rw_flag = 1;
#pragma unroll
    for (int k = 0; k < 8; k++) m.value[k] = job.words[8 * i + k];
    uint4* d = reinterpret_cast<uint4*>(job.mem_q + i);
    const uint4* s = reinterpret_cast<const uint4*>(&m);
    d[0] = s[0]; d[1] = s[1]; d[2] = s[2];
    if (!job.mem_enc) return;  // zkw_decommitter_memory_queries: the queries only
    u64 e[8];
    encode_mem_query(m, e);
    ulonglong2* o = reinterpret_cast<ulonglong2*>(job.mem_enc + 8 * i);
    o[0] = make_ulonglong2(e[0], e[1]); o[1] = make_ulonglong2(e[2], e[3]);
    o[2] = make_ulonglong2(e[4], e[5]); o[3] = make_ulonglong2(e[6], e[7]);
}

struct DecommitterBlock {
    DecommitterJob job;
    const u64* dedup_tails;  // [n_requests][12]
    const u64* mem_tails;    // [total_words][12]
    zkw_decommitter_instance* instances;
    zkw_queue_state12 mem_in;
    u64 total_rounds, total_words;
    u32 capacity;
};

static __device__ __forceinline__ void k_decommitter_instances(const VB& vb, const DecommitterBlock* __restrict__ blk) {
    const DecommitterBlock& b = *blk;
    const u64 n_inst = (b.total_rounds + b.capacity - 1) / b.capacity, nreq = b.job.n_requests;
    const u64 idx = (u64)vb.x * blockDim.x + threadIdx.x;
    if (idx >= n_inst) return;
    const u64* dedup_final = b.dedup_tails + 12 * (nreq - 1);
    // FSM after `end` global rounds (end > 0): request k holds round end-1
    auto fill = [&](zkw_decommitter_fsm& f, u64 end, u64& popped, u64& words_done) {
        u64 lo = 0, hi = nreq;
        while (hi - lo > 1) {
            const u64 mid = (lo + hi) >> 1;
            if (b.job.round_offsets[mid] <= end - 1) lo = mid; else hi = mid;
        }
        const u64 k = lo, r = end - b.job.round_offsets[k];
        const u64 nw = b.job.word_offsets[k + 1] - b.job.word_offsets[k], rounds = (nw + 1) / 2;
        const u64 in_req = 2 * r < nw ? 2 * r : nw;  // every round takes two words except the last one
        popped = k + 1;
        words_done = b.job.word_offsets[k] + in_req;
        const zkw_decommit_query* q = b.job.requests + k;
        qs12(f.decommittment_requests_queue_state, b.dedup_tails + 12 * k, dedup_final, (u32)(nreq - popped));
        qs12(f.memory_queue_state, b.mem_in.head, b.mem_tails + 12 * (words_done - 1), b.mem_in.length + (u32)words_done);
        const u32* st = b.job.round_states + 8 * (end - 1);
        for (int j = 0; j < 8; j++) { f.sha256_inner_state[j] = st[j]; f.hash_to_compare_against[j] = j < 7 ? q->hash[j] : 0; }
        f.current_index = (u32)in_req;
        f.current_page = q->memory_page;
        f.timestamp = q->timestamp;
        f.num_rounds_left = (u32)(rounds - r);
        f.length_in_bits = (u32)nw * 32 * 8;
        const bool req_done = r == rounds, all_done = req_done && k + 1 == nreq;
        f.state_get_from_queue = (req_done && !all_done) ? 1 : 0;
        f.state_decommit = req_done ? 0 : 1;
        f.finished = all_done ? 1 : 0;
        f._pad = 0;
    };
    zkw_decommitter_instance& w = b.instances[idx];  // filled in place: a local copy would live in scratch memory
    memset(&w, 0, sizeof w);
    const u64 lo = idx * b.capacity, hi = lo + b.capacity < b.total_rounds ? lo + b.capacity : b.total_rounds;
    u64 p0 = 0, w0 = 0, p1 = 0, w1 = 0;
    w.start_flag = idx == 0;
    if (idx == 0) {
        w.memory_queue_initial_state = b.mem_in;
        qs12(w.sorted_requests_queue_initial_state, nullptr, dedup_final, (u32)nreq);
        w.hidden_fsm_input.memory_queue_state = b.mem_in;  // all_memory_queue_states[start_idx - 1]
    } else {
        fill(w.hidden_fsm_input, lo, p0, w0);
    }
    fill(w.hidden_fsm_output, hi, p1, w1);
    w.first_round = lo; w.num_rounds = hi - lo;
    w.first_request = p0; w.num_requests = p1 - p0;
    w.first_word = w0; w.num_words = w1 - w0;
    if (idx == n_inst - 1) {
        w.completion_flag = 1;
        w.memory_queue_final_state = w.hidden_fsm_output.memory_queue_state;
    }
}

// ------------------------------------------------------------------------------------------------
// L1 messages hasher (compute_linear_keccak256, data_hasher_and_merklizer.rs:8-67): one Keccak-256 sponge
// over n * 88 bytes. The sponge is serial; one wave runs it with lane x+5y holding lane (x, y) of the state
// (theta / rho-pi / chi through ds_bpermute-free LDS-less shuffles), the message bytes are produced on the fly.
static __constant__ u64 c_keccak_rc[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
    0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
    0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
    0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
    0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
static __constant__ int c_keccak_rot[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};

__device__ __forceinline__ u64 rol64(u64 x, int r) { return r ? (x << r) | (x >> (64 - r)) : x; }

// Keccak-f[1600] on one lane, fully unrolled (registers only)
__device__ inline void keccak_f1600(u64 a[25]) {
    for (int round = 0; round < 24; round++) {
        u64 c[5], b[25];
#pragma unroll
        for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];
#pragma unroll
        for (int x = 0; x < 5; x++) {
            const u64 d = c[(x + 4) % 5] ^ rol64(c[(x + 1) % 5], 1);
#pragma unroll
            for (int y = 0; y < 5; y++) a[x + 5 * y] ^= d;
        }
#pragma unroll
        for (int x = 0; x < 5; x++)
#pragma unroll
            for (int y = 0; y < 5; y++) b[y + 5 * ((2 * x + 3 * y) % 5)] = rol64(a[x + 5 * y], c_keccak_rot[x + 5 * y]);
#pragma unroll
        for (int y = 0; y < 5; y++)
#pragma unroll
            for (int x = 0; x < 5; x++) a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
        a[0] ^= c_keccak_rc[round];
    }
}

// byte `pos` of the concatenated serialisations (log_query.rs:503-534)
__device__ __forceinline__ u32 l1_message_byte(const zkw_log_query* q, size_t pos) {
    const zkw_log_query* m = q + pos / 88;
    const int o = (int)(pos % 88);
    if (o == 0) return m->shard_id;
    if (o == 1) return m->is_service ? 1 : 0;
    if (o == 2) return m->tx_number_in_block >> 8;
    if (o == 3) return m->tx_number_in_block & 0xFF;
    if (o < 24) { const int b = 19 - (o - 4); return (m->address[b >> 2] >> (8 * (b & 3))) & 0xFF; }       // big-endian
    if (o < 56) { const int b = 31 - (o - 24); return (m->key[b >> 2] >> (8 * (b & 3))) & 0xFF; }
    const int b = 31 - (o - 56);
    return (m->written_value[b >> 2] >> (8 * (b & 3))) & 0xFF;
}

// lane t (< 17) of the 136-byte block at byte offset `off` of a queue of `len` message bytes, pad10*1 included
__device__ __forceinline__ u64 l1_block_lane(const zkw_log_query* q, size_t len, size_t off, int t) {
    const bool last = len - off < 136;
    u64 lane = 0;
    for (int b = 0; b < 8; b++) {
        const size_t pos = off + 8 * t + b;
        u32 byte = pos < len ? l1_message_byte(q, pos) : 0;
        if (last && pos == len) byte ^= 0x01;
        if (last && 8 * t + b == 135) byte ^= 0x80;
        lane |= (u64)byte << (8 * b);
    }
    return lane;
}

// The absorbed blocks of all queues of a batch, in parallel: thread = (block, word w < 18) -> rounds[block].block words (w = 17:
// the reset flag). The serial sponge then only loads them — building a block's bytes from the 88-byte serialisations inside the
// sponge loop (eight dependent message loads and a division per byte) took more of its time than the permutation.
static __device__ __forceinline__ void k_linear_blocks(const VB& vb, const zkw_log_query* __restrict__ q, const u64* __restrict__ msg_off,
                                                               const u64* __restrict__ round_off, u32 n_queues, zkw_keccak_round_record* __restrict__ rounds) {
    const u64 i = (u64)vb.x * blockDim.x + threadIdx.x, total = round_off[n_queues] * 18;
    if (i >= total) return;
    const u64 blk = i / 18;
    const int w = (int)(i % 18);
    u32 b = 0;
    while (round_off[b + 1] <= blk) b++;
    const u64 local = blk - round_off[b];
    const size_t len = (size_t)(msg_off[b + 1] - msg_off[b]) * 88;
    u64* dst = reinterpret_cast<u64*>(rounds[blk].block);
    dst[w] = w < 17 ? l1_block_lane(q + msg_off[b], len, (size_t)local * 136, w) : (local == 0 ? 1 : 0);
}

// rounds (may be null): one zkw_keccak_round_record per absorbed block — the cycles of the LinearHasher circuit (type 13).
// The sponge is serial: 25 lanes of one wave hold the state, one 64-bit lane each (with the state in LDS and three barriers per
// round a Keccak-f took 9.7 us; lane 0 alone running the unrolled register form of keccak_f1600 was measured at 17 us)
// Batch form: workgroup b hashes the messages [msg_off[b], msg_off[b + 1]) into out + 32 b, its round records start at
// rounds + round_off[b] (msg_off == nullptr: one queue of n messages). The queues of a batch run side by side — the chain of
// one queue stays serial.
static __device__ __forceinline__ void k_linear_keccak256(const VB& vb, const zkw_log_query* __restrict__ q, size_t n, uint8_t* __restrict__ out,
                                                         zkw_keccak_round_record* __restrict__ rounds,
                                                         const u64* __restrict__ msg_off, const u64* __restrict__ round_off, bool blocks_ready) {
    const int t = threadIdx.x;
    if (msg_off) {
        const u64 m0 = msg_off[vb.x];
        q += m0;
        n = msg_off[vb.x + 1] - m0;
        out += 32 * (size_t)vb.x;
        if (rounds) rounds += round_off[vb.x];
    }
    const size_t len = n * 88;
    // lane t = x + 5 y < 25 holds lane (x, y) of the state in a register; a round is four exchanges through the wave's shuffle
    // network (column parities, their neighbours, rho-pi, chi's row neighbours) with no LDS round trip and no barrier: 4.4 ms
    // per queue of 700 messages with the state in LDS and three barriers per round, ~half with this form
    const int x = t % 5, y = t / 5;
    const int src_pi = (x + 3 * y) % 5 + 5 * x;                // B[x + 5y] = rol(A'[x' + 5y'], rot[x' + 5y']) with x = y', y = (2x' + 3y') % 5
    const int rot_pi = t < 25 ? c_keccak_rot[src_pi] : 0;
    u64 a = 0, lane_next = 0;
    if (blocks_ready && t < 17) lane_next = reinterpret_cast<const u64*>(rounds[0].block)[t];
    for (size_t off = 0;; off += 136) {
        const bool last = len - off < 136;  // the final (padded) block; len % 136 == 0 gives a pure padding block
        if (t < 17) {
            u64 lane;
            if (blocks_ready) {  // (k_linear_blocks has written every block: one load, the next block's issued before this block's rounds)
                lane = lane_next;
                if (!last) lane_next = reinterpret_cast<const u64*>(rounds[off / 136 + 1].block)[t];
            } else {
                lane = l1_block_lane(q, len, off, t);
                if (rounds) reinterpret_cast<u64*>(rounds[off / 136].block)[t] = lane;  // records are 8-byte aligned (344 = 8 * 43)
            }
            a ^= lane;
        }
        if (rounds && !blocks_ready && t == 17) reinterpret_cast<u64*>(rounds[off / 136].block)[17] = off == 0 ? 1 : 0;  // reset + padding
        for (int round = 0; round < 24; round++) {
            // theta: column parity (the four other lanes of the column), D from the neighbouring columns' parities
            const u64 c = a ^ __shfl(a, (t + 5) % 25) ^ __shfl(a, (t + 10) % 25) ^ __shfl(a, (t + 15) % 25) ^ __shfl(a, (t + 20) % 25);
            const u64 d = __shfl(c, (x + 4) % 5) ^ rol64(__shfl(c, (x + 1) % 5), 1);
            const u64 ap = a ^ d;
            // rho + pi: lane (x, y) takes the rotated lane it receives
            const u64 bm = rol64(__shfl(ap, src_pi), rot_pi);
            // chi + iota
            const u64 b1 = __shfl(bm, (x + 1) % 5 + 5 * y), b2 = __shfl(bm, (x + 2) % 5 + 5 * y);
            a = bm ^ (~b1 & b2);
            if (t == 0) a ^= c_keccak_rc[round];
        }
        if (rounds && t < 25) reinterpret_cast<u64*>(rounds[off / 136].state_after)[t] = a;
        if (last) break;
    }
    // the first 32 bytes of the state: lanes 0..3
    const u64 w = __shfl(a, (t >> 3) & 3);
    if (t < 32) out[t] = (uint8_t)(w >> (8 * (t & 7)));
}

}  // namespace zkw
