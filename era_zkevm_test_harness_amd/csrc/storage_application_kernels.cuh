// storage_application_kernels.cuh — StorageApplication witness builder.
//
// Reference: decompose_into_storage_application_witnesses, src/witness/individual_circuits/storage_application.rs:31-361,
// over InMemoryStorageTree<256, 32, 8, Blake2s256, ZkSyncStorageLeaf> (src/witness/tree/mod.rs:113-384) and
// StateDiffRecord::encode (circuit_encodings/src/state_diff_record.rs:21-53).
//
// The reference applies the queries to the tree one after another (256 Blake2s per touched leaf, each path seeing
// the writes before it). Here the caller hands over what a tree database returns for the state BEFORE the block
// (leaf index + Merkle path per slot); the sequential semantics are rebuilt in parallel:
//   * the slots of a deduplicated queue are distinct, so the sibling of slot i at level L changed iff an earlier
//     write j < i lies in that sibling subtree, i.e. iff L is the highest bit where key_i and key_j differ; the
//     LATEST such j has recomputed that subtree's hash on its own way up. k_sap_pairs finds j*(i, L) for all
//     levels by a plain all-pairs sweep (lane = i, uniform j, per-lane level table in LDS);
//   * then 256 level-synchronous launches walk all paths upwards together: sibling(i, L) = A(j*, L) or the given
//     pre-state sibling, A(i, L + 1) = Blake2s of the ordered pair (k_sap_level);
//   * enumeration indices / instance cuts are a short serial pass over flags staged in LDS (k_sap_scan), the
//     state-diff Keccak-256 accumulator is one cooperative sponge (k_sap_keccak).
#pragma once
#include "precompile_kernels.cuh"

namespace zkw {

static __constant__ u32 c_b2s_iv[8] = {0x6A09E667, 0xBB67AE85, 0x3C6EF372, 0xA54FF53A, 0x510E527F, 0x9B05688C, 0x1F83D9AB, 0x5BE0CD19};
static __constant__ uint8_t c_b2s_sigma[10][16] = {
    {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
    {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
    {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
    {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
    {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0}};

// Blake2s-256 of a message of `len` <= 64 bytes given as 16 little-endian words (zero padded)
__device__ inline void blake2s_one_block(const u32 m[16], u32 len, u32 out[8]) {
    u32 h[8], v[16];
#pragma unroll
    for (int i = 0; i < 8; i++) h[i] = c_b2s_iv[i];
    h[0] ^= 0x01010000u ^ 32u;
#pragma unroll
    for (int i = 0; i < 8; i++) { v[i] = h[i]; v[i + 8] = c_b2s_iv[i]; }
    v[12] ^= len;
    v[14] = ~v[14];
#define B2S_G(a, b, c, d, x, y)                                                                                  \
    v[a] = v[a] + v[b] + (x); v[d] = rotr(v[d] ^ v[a], 16); v[c] = v[c] + v[d]; v[b] = rotr(v[b] ^ v[c], 12);     \
    v[a] = v[a] + v[b] + (y); v[d] = rotr(v[d] ^ v[a], 8);  v[c] = v[c] + v[d]; v[b] = rotr(v[b] ^ v[c], 7);
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const uint8_t* s = c_b2s_sigma[r];
        B2S_G(0, 4, 8, 12, m[s[0]], m[s[1]]) B2S_G(1, 5, 9, 13, m[s[2]], m[s[3]])
        B2S_G(2, 6, 10, 14, m[s[4]], m[s[5]]) B2S_G(3, 7, 11, 15, m[s[6]], m[s[7]])
        B2S_G(0, 5, 10, 15, m[s[8]], m[s[9]]) B2S_G(1, 6, 11, 12, m[s[10]], m[s[11]])
        B2S_G(2, 7, 8, 13, m[s[12]], m[s[13]]) B2S_G(3, 4, 9, 14, m[s[14]], m[s[15]])
    }
#undef B2S_G
#pragma unroll
    for (int i = 0; i < 8; i++) out[i] = h[i] ^ v[i] ^ v[i + 8];
}

__device__ __forceinline__ u32 bswap32(u32 x) { return __builtin_bswap32(x); }

// leaf_hash(index BE (8 bytes) || value (32 bytes)), tree/mod.rs:322-329; value given as U256 limbs (big-endian bytes
// = limbs from the most significant, byte-swapped)
__device__ inline void sap_leaf_hash(u64 index, const u32 value_limbs[8], u32 out[8]) {
    u32 m[16];
    m[0] = bswap32((u32)(index >> 32));
    m[1] = bswap32((u32)index);
#pragma unroll
    for (int k = 0; k < 8; k++) m[2 + k] = bswap32(value_limbs[7 - k]);
#pragma unroll
    for (int k = 10; k < 16; k++) m[k] = 0;
    blake2s_one_block(m, 40, out);
}

__device__ inline void sap_node_hash(const u32 l[8], const u32 r[8], u32 out[8]) {
    u32 m[16];
#pragma unroll
    for (int k = 0; k < 8; k++) { m[k] = l[k]; m[8 + k] = r[k]; }
    blake2s_one_block(m, 64, out);
}

constexpr u32 SAP_NONE = 0xFFFFFFFFu;

struct SapJob {
    const zkw_log_query* queries;  // [n]
    const u64* init_index;         // [n] pre-state enumeration index of the slot's leaf (0 = empty)
    const u32* init_paths;         // [n][256][8] pre-state Merkle paths (32-byte nodes as 8 LE words)
    u32* keys;                     // [n][8] derived keys
    u32* paths;                    // [n][256][8] out
    u64* new_index;                // [n] enumeration index of the leaf after the query
    u32* prev_write;               // [n] latest write strictly before i (SAP_NONE if none)
    u32* chunk_of;                 // [n] instance that owns query i
    u32* first_writes_upto;        // [n] first writes among queries [0, i]
    u64* chunk_end;                // [n + 1] exclusive end of instance c
    u32* jstar;                    // [256][n]
    u32 *A0, *A1, *C0, *C1;        // [n][8] ping-pong: current-tree / pre-state ancestor hashes
    u32 *R0, *R1;                  // [n][8] ping-pong: ancestors of the leaf AS READ over the current siblings (the first walk of a write)
    u32* walk_hashes;              // [n][2][257][8] kept for synthesis: level 0 = the leaf hash; [0] the leaf as read, [1] as written
    u32* roots;                    // [n][8] root after query i
    u32* violations;
    u64* meta;                     // [0] number of instances, [1] next enumeration index after the block
    u64 n, next_enumeration_index;
    u32 initial_root[8];
    u32 capacity;
};

// derive_final_address: Blake2s-256(0^12 || address BE (20) || key BE (32))
static __device__ __forceinline__ void k_sap_keys(const VB& vb, SapJob job) {
    const u64 i = (u64)vb.x * blockDim.x + threadIdx.x;
    if (i >= job.n) return;
    const zkw_log_query* q = job.queries + i;
    u32 m[16], k[8];
    m[0] = m[1] = m[2] = 0;
#pragma unroll
    for (int j = 0; j < 5; j++) m[3 + j] = bswap32(q->address[4 - j]);
#pragma unroll
    for (int j = 0; j < 8; j++) m[8 + j] = bswap32(q->key[7 - j]);
    blake2s_one_block(m, 64, k);
#pragma unroll
    for (int j = 0; j < 8; j++) job.keys[8 * i + j] = k[j];
}

// One workgroup. Flags are staged in LDS tile by tile; lane 0 runs the two sequential rules (instance cuts,
// storage_application.rs:143-165; enumeration of first writes, tree/mod.rs:305-313), all lanes write the results.
static __device__ __forceinline__ void k_sap_scan(const VB& vb, SapJob job) {
    __shared__ uint8_t s_rw[1024], s_first[1024];
    __shared__ u32 s_chunk[1024], s_prev[1024], s_enum[1024];
    __shared__ u32 total, chunk, first_writes, prev;
    const int t = threadIdx.x;
    if (t == 0) { total = 0; chunk = 0; first_writes = 0; prev = SAP_NONE; }
    __syncthreads();
    for (u64 base = 0; base < job.n; base += 1024) {
        const u64 i = base + t;
        if (i < job.n) {
            const bool rw = job.queries[i].rw_flag != 0;
            s_rw[t] = rw;
            s_first[t] = rw && job.init_index[i] == 0;
        }
        __syncthreads();
        if (t == 0) {
            const int m = (int)(job.n - base < 1024 ? job.n - base : 1024);
            u32 tot = total, ch = chunk, fw = first_writes, pv = prev;
            for (int k = 0; k < m; k++) {
                s_chunk[k] = ch; s_prev[k] = pv; s_enum[k] = fw;
                tot += s_rw[k] ? 2 : 1;
                if (s_first[k]) fw++;
                if (s_rw[k]) pv = (u32)(base + k);
                if (tot >= job.capacity - 1) { job.chunk_end[ch++] = base + k + 1; tot = 0; }
            }
            total = tot; chunk = ch; first_writes = fw; prev = pv;
        }
        __syncthreads();
        if (i < job.n) {
            job.chunk_of[i] = s_chunk[t];
            job.prev_write[i] = s_prev[t];
            job.new_index[i] = s_first[t] ? job.next_enumeration_index + s_enum[t] : job.init_index[i];
            job.first_writes_upto[i] = s_enum[t] + (s_first[t] ? 1 : 0);
        }
        __syncthreads();
    }
    if (t == 0) {
        if (total != 0) job.chunk_end[chunk++] = job.n;
        job.meta[0] = chunk;
        job.meta[1] = job.next_enumeration_index + first_writes;
    }
}

// j*(i, L): the latest write j < i whose key first differs from key_i (from the top) at bit L. One wave per block,
// lane = i; j is uniform across the wave so key_j is a broadcast load. The per-lane level table lives in LDS as
// [level][lane] (bank = lane, conflict free).
static __device__ __forceinline__ void k_sap_pairs(const VB& vb, SapJob job) {
    __shared__ u32 tab[256 * 64];
    const int lane = threadIdx.x;
    const u64 i = (u64)vb.x * 64 + lane;
    for (int L = 0; L < 256; L++) tab[L * 64 + lane] = SAP_NONE;
    u32 ki[8];
    const bool live = i < job.n;
#pragma unroll
    for (int w = 0; w < 8; w++) ki[w] = live ? job.keys[8 * i + w] : 0;
    const u64 last = (u64)vb.x * 64 + 63 < job.n ? (u64)vb.x * 64 + 63 : job.n - 1;
    bool dup = false;
    for (u64 j = 0; j < last; j++) {
        if (!job.queries[j].rw_flag) continue;  // uniform
        u32 x[8];
#pragma unroll
        for (int w = 0; w < 8; w++) x[w] = ki[w] ^ job.keys[8 * j + w];
        int level = -1;
#pragma unroll
        for (int w = 0; w < 8; w++)
            if (x[w]) level = 32 * w + 31 - __clz(x[w]);
        if (live && j < i) {
            if (level < 0) dup = true; else tab[level * 64 + lane] = (u32)j;
        }
    }
    if (dup) atomicAdd(job.violations, 1u);  // the deduplicated queue must not hold a slot twice
    if (live)
        for (int L = 0; L < 256; L++) job.jstar[(u64)L * job.n + i] = tab[L * 64 + lane];
}

// level 0: the leaf after the query (current tree) and the leaf before the block (pre-state check)
static __device__ __forceinline__ void k_sap_leaves(const VB& vb, SapJob job) {
    const u64 i = (u64)vb.x * blockDim.x + threadIdx.x;
    if (i >= job.n) return;
    const zkw_log_query* q = job.queries + i;
    u32 a[8], c[8];
    sap_leaf_hash(job.init_index[i], q->read_value, c);
    if (q->rw_flag) sap_leaf_hash(job.new_index[i], q->written_value, a);
    else {
#pragma unroll
        for (int k = 0; k < 8; k++) a[k] = c[k];
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
        job.A0[8 * i + k] = a[k]; job.C0[8 * i + k] = c[k]; job.R0[8 * i + k] = c[k];
        job.walk_hashes[((i * 2 + 0) * 257) * 8 + k] = c[k];
        job.walk_hashes[((i * 2 + 1) * 257) * 8 + k] = a[k];
    }
}

// one level of every path: A(i, L + 1), C(i, L + 1) from level L (see the header comment)
__device__ __forceinline__ void sap_level_step(const SapJob& job, int L, u64 i) {
    const u32 *Ac = (L & 1) ? job.A1 : job.A0, *Cc = (L & 1) ? job.C1 : job.C0;
    u32 *An = (L & 1) ? job.A0 : job.A1, *Cn = (L & 1) ? job.C0 : job.C1;
    const u32* Rc = (L & 1) ? job.R1 : job.R0;
    u32* Rn = (L & 1) ? job.R0 : job.R1;
    const u32 js = job.jstar[(u64)L * job.n + i];
    const u32* ip = job.init_paths + ((u64)i * 256 + L) * 8;
    u32 init_sib[8], sib[8], a[8], c[8], r[8], o[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        init_sib[k] = ip[k];
        sib[k] = js == SAP_NONE ? init_sib[k] : Ac[8 * (u64)js + k];
        a[k] = Ac[8 * i + k];
        c[k] = Cc[8 * i + k];
        r[k] = Rc[8 * i + k];
    }
    u32* op = job.paths + ((u64)i * 256 + L) * 8;
#pragma unroll
    for (int k = 0; k < 8; k++) op[k] = sib[k];
    const bool right = (job.keys[8 * i + (L >> 5)] >> (L & 31)) & 1;  // is_right_side_node, tree/mod.rs:147-155
    if (right) sap_node_hash(sib, a, o); else sap_node_hash(a, sib, o);
    u32* const wh = job.walk_hashes + ((i * 2) * 257 + (u64)(L + 1)) * 8;
#pragma unroll
    for (int k = 0; k < 8; k++) { An[8 * i + k] = o[k]; wh[257 * 8 + k] = o[k]; }
    if (job.queries[i].rw_flag) {  // the walk of the leaf as read over the same siblings; a read has one walk only (a == r)
        if (right) sap_node_hash(sib, r, o); else sap_node_hash(r, sib, o);
    }
#pragma unroll
    for (int k = 0; k < 8; k++) { Rn[8 * i + k] = o[k]; wh[k] = o[k]; }
    if (right) sap_node_hash(init_sib, c, o); else sap_node_hash(c, init_sib, o);
#pragma unroll
    for (int k = 0; k < 8; k++) Cn[8 * i + k] = o[k];
}

// Level-synchronous launches: for blocks with more storage queries than one workgroup walks (n > SAP_PERSISTENT_MAX)
static __device__ __forceinline__ void k_sap_level(const VB& vb, SapJob job, int L) {
    const u64 i = (u64)vb.x * blockDim.x + threadIdx.x;
    if (i < job.n) sap_level_step(job, L, i);
}

// All 256 levels in ONE launch: a single workgroup walks every path upwards, a workgroup barrier (and a device-scope fence:
// the level's hashes travel through global memory) between levels instead of a kernel boundary. A block's storage
// application sees tens to a few hundred distinct slots (production: 33 per instance), so one CU holds all of them; the walk
// is a chain of 256 dependent Blake2s pairs either way.
constexpr int SAP_PERSISTENT_THREADS = 256;
constexpr u64 SAP_PERSISTENT_MAX = 4 * SAP_PERSISTENT_THREADS;
static __device__ __forceinline__ void k_sap_levels(const VB& vb, SapJob job) {
    for (int L = 0; L < 256; L++) {
        for (u64 i = threadIdx.x; i < job.n; i += SAP_PERSISTENT_THREADS) sap_level_step(job, L, i);
        __threadfence();
        __syncthreads();
    }
}

// after level 255 (A0 / C0 hold the roots): root after every query + the reference's inclusion asserts
static __device__ __forceinline__ void k_sap_roots(const VB& vb, SapJob job) {
    const u64 i = (u64)vb.x * blockDim.x + threadIdx.x;
    if (i >= job.n) return;
    bool bad = false;
    for (int k = 0; k < 8; k++) bad |= job.C0[8 * i + k] != job.initial_root[k];  // the given pre-state proof
    const bool rw = job.queries[i].rw_flag != 0;
    const u32 pw = job.prev_write[i];
    for (int k = 0; k < 8; k++) {
        const u32 before = pw == SAP_NONE ? job.initial_root[k] : job.A0[8 * (u64)pw + k];
        const u32 mine = job.A0[8 * i + k];
        if (!rw) bad |= mine != before;  // verify_inclusion of a read against the current root, :270
        job.roots[8 * i + k] = rw ? mine : before;
    }
    if (bad) atomicAdd(job.violations, 1u);
}

struct SapKeccakOut {
    u64* snapshots;  // [n_instances][25] accumulator state at the end of each instance
    uint8_t* final_hash;  // [32]
};

// byte `pos` (< 272) of the zero-extended StateDiffRecord::encode of query i
__device__ __forceinline__ u32 sap_diff_byte(const SapJob& job, u64 i, int pos) {
    if (pos >= ZKW_STATE_DIFF_RECORD_BYTE_ENCODING_LEN) return 0;
    const zkw_log_query* q = job.queries + i;
    if (pos < 20) { const int b = 19 - pos; return (q->address[b >> 2] >> (8 * (b & 3))) & 0xFF; }
    if (pos < 52) { const int b = 31 - (pos - 20); return (q->key[b >> 2] >> (8 * (b & 3))) & 0xFF; }
    if (pos < 84) { const int b = pos - 52; return (job.keys[8 * i + (b >> 2)] >> (8 * (b & 3))) & 0xFF; }
    if (pos < 92) return (u32)(job.init_index[i] >> (8 * (7 - (pos - 84)))) & 0xFF;  // the index BEFORE the write
    if (pos < 124) { const int b = 31 - (pos - 92); return (q->read_value[b >> 2] >> (8 * (b & 3))) & 0xFF; }
    const int b = 31 - (pos - 124);
    return (q->written_value[b >> 2] >> (8 * (b & 3))) & 0xFF;
}

// one wave: the running Keccak-256 over the writes' state diffs (two rate blocks each, :253-260)
static __device__ __forceinline__ void k_sap_keccak(const VB& vb, SapJob job, SapKeccakOut out) {
    __shared__ u64 A[25], Bm[25], Cc[5];
    const int t = threadIdx.x;
    if (t < 25) A[t] = 0;
    __syncthreads();
    auto permute = [&]() {
        for (int round = 0; round < 24; round++) {
            if (t < 5) Cc[t] = A[t] ^ A[t + 5] ^ A[t + 10] ^ A[t + 15] ^ A[t + 20];
            __syncthreads();
            if (t < 25) {
                const int x = t % 5, y = t / 5;
                const u64 d = Cc[(x + 4) % 5] ^ rol64(Cc[(x + 1) % 5], 1);
                Bm[y + 5 * ((2 * x + 3 * y) % 5)] = rol64(A[t] ^ d, c_keccak_rot[t]);
            }
            __syncthreads();
            if (t < 25) {
                const int x = t % 5, y = t / 5;
                u64 v = Bm[t] ^ (~Bm[(x + 1) % 5 + 5 * y] & Bm[(x + 2) % 5 + 5 * y]);
                if (t == 0) v ^= c_keccak_rc[round];
                A[t] = v;
            }
            __syncthreads();
        }
    };
    u64 c = 0;
    for (u64 i = 0; i < job.n; i++) {
        if (job.queries[i].rw_flag) {
            for (int blk = 0; blk < ZKW_NUM_KECCAK256_ROUNDS_PER_RECORD_ACCUMULATION; blk++) {
                if (t < 17) {
                    u64 lane = 0;
                    for (int b = 0; b < 8; b++) lane |= (u64)sap_diff_byte(job, i, 136 * blk + 8 * t + b) << (8 * b);
                    A[t] ^= lane;
                }
                __syncthreads();
                permute();
            }
        }
        if (i + 1 == job.chunk_end[c]) {
            if (t < 25) out.snapshots[25 * c + t] = A[t];
            c++;
        }
    }
    // finalize() of the (cloned) hasher: an empty padded block
    __syncthreads();
    if (t == 0) A[0] ^= 0x01;
    if (t == 16) A[16] ^= 0x8000000000000000ull;
    __syncthreads();
    permute();
    if (t < 32) out.final_hash[t] = (uint8_t)(A[t >> 3] >> (8 * (t & 7)));
}

struct SapBlock {
    SapJob job;
    const u64* query_tails;  // [n][4]
    const u64* snapshots;
    const uint8_t* final_hash;
    zkw_storage_application_instance* instances;
    u64 n_instances;
};

__device__ __forceinline__ void sap_bytes32(uint8_t* dst, const u32* words) {
    for (int k = 0; k < 8; k++)
        for (int b = 0; b < 4; b++) dst[4 * k + b] = (uint8_t)(words[k] >> (8 * b));
}

static __device__ __forceinline__ void k_sap_instances(const VB& vb, const SapBlock* __restrict__ blk) {
    const SapBlock b = *blk;
    const u64 c = (u64)vb.x * blockDim.x + threadIdx.x;
    if (c >= b.n_instances) return;
    const u64 n = b.job.n;
    const u64* tail_final = n ? b.query_tails + 4 * (n - 1) : nullptr;
    auto fsm = [&](zkw_storage_application_fsm& f, u64 end) {  // state after the first `end` queries (end > 0)
        const u64 last = end - 1;
        const u64 ne = b.job.next_enumeration_index + b.job.first_writes_upto[last];
        f.next_enumeration_counter[0] = (u32)ne; f.next_enumeration_counter[1] = (u32)(ne >> 32);
        sap_bytes32(f.current_root_hash, b.job.roots + 8 * last);
        qs4(f.current_storage_application_log_state, b.query_tails + 4 * last, tail_final, (u32)(n - end));
    };
    auto keccak_state = [&](uint8_t* dst, const u64* st) {
        for (int id = 0; id < 25; id++)
            for (int by = 0; by < 8; by++) dst[((id % 5) * 5 + id / 5) * 8 + by] = st ? (uint8_t)(st[id] >> (8 * by)) : 0;
    };
    zkw_storage_application_instance& w = b.instances[c];  // filled in place: a local copy would live in scratch memory
    memset(&w, 0, sizeof w);
    const u64 ne0 = b.job.next_enumeration_index;
    if (n == 0) {  // the dummy instance, :69-132
        w.start_flag = w.completion_flag = 1;
        w.initial_next_enumeration_counter[0] = (u32)ne0; w.initial_next_enumeration_counter[1] = (u32)(ne0 >> 32);
        sap_bytes32(w.initial_root_hash, b.job.initial_root);
        w.hidden_fsm_output.next_enumeration_counter[0] = (u32)ne0; w.hidden_fsm_output.next_enumeration_counter[1] = (u32)(ne0 >> 32);
        sap_bytes32(w.hidden_fsm_output.current_root_hash, b.job.initial_root);
        w.new_next_enumeration_counter[0] = (u32)ne0; w.new_next_enumeration_counter[1] = (u32)(ne0 >> 32);
        sap_bytes32(w.new_root_hash, b.job.initial_root);
        for (int k = 0; k < 32; k++) w.state_diffs_keccak256_hash[k] = b.final_hash[k];
        return;
    }
    const u64 lo = c ? b.job.chunk_end[c - 1] : 0, hi = b.job.chunk_end[c];
    w.start_flag = c == 0;
    w.completion_flag = c + 1 == b.n_instances;
    if (c == 0) {
        w.initial_next_enumeration_counter[0] = (u32)ne0; w.initial_next_enumeration_counter[1] = (u32)(ne0 >> 32);
        sap_bytes32(w.initial_root_hash, b.job.initial_root);
        qs4(w.storage_application_log_state, nullptr, tail_final, (u32)n);
    } else {
        fsm(w.hidden_fsm_input, lo);
        keccak_state(w.hidden_fsm_input.current_diffs_keccak_accumulator_state, b.snapshots + 25 * (c - 1));
    }
    fsm(w.hidden_fsm_output, hi);
    keccak_state(w.hidden_fsm_output.current_diffs_keccak_accumulator_state, b.snapshots + 25 * c);
    if (w.completion_flag) {
        w.new_next_enumeration_counter[0] = w.hidden_fsm_output.next_enumeration_counter[0];
        w.new_next_enumeration_counter[1] = w.hidden_fsm_output.next_enumeration_counter[1];
        for (int k = 0; k < 32; k++) { w.new_root_hash[k] = w.hidden_fsm_output.current_root_hash[k]; w.state_diffs_keccak256_hash[k] = b.final_hash[k]; }
    }
    w.first_item = lo;
    w.num_items = hi - lo;
}

// ------------------------------------------------------------------------------------------------ synthesis inputs (type 10)
// What zkw_storage_application_synthesize needs of a query after the builder's scratch is gone: the leaf before and after it.
struct SapItem { u64 read_index, write_index; u32 read_value[8], written_value[8]; u32 rw, _pad; };
static __device__ __forceinline__ void k_sap_items(const VB& vb, SapJob job, SapItem* __restrict__ items) {
    const u64 i = (u64)vb.x * blockDim.x + threadIdx.x;
    if (i >= job.n) return;
    const zkw_log_query* q = job.queries + i;
    SapItem& it = items[i];
    it.read_index = job.init_index[i];
    it.write_index = job.new_index[i];
#pragma unroll
    for (int k = 0; k < 8; k++) { it.read_value[k] = q->read_value[k]; it.written_value[k] = q->written_value[k]; }
    it.rw = q->rw_flag != 0;
    it._pad = 0;
}

// The inputs of the netlist engine (include/zkw_storage_application_circuit_spec.h: cycle = one Blake2s compression of a Merkle
// walk, 257 cycles per walk; a read is one walk, a write two — storage_application.rs:141-153) for the instances of a block:
// header bits, free elements (leaf message / sibling, the walk's key << 1), the state (hash 32, key 33) before every cycle.
struct SapWalkJob {
    const SapItem* items;     // block-wide
    const u32* keys;          // [n][8]
    const u32* paths;         // [n][256][8]
    const u32* walk_hashes;   // [n][2][257][8]: the running hash after every cycle of a query's walks (the builder's level walk)
    u64 first_item, num_items;
    uint8_t* hdr_bits;        // [cycles]
    uint8_t* free_elems;      // [cycles][97]
    uint8_t* state_before;    // [cycles + 1][65]
    const uint8_t* idle_root; // [32]: the root an instance WITHOUT walks carries in its idle cycles (its FSM output's current_root_hash: the
                              // closed-form section ties the root an instance hands on to the state after its last cycle); unused when it has walks
};
constexpr u32 SAP_WALK_CYCLES = 257, SAP_WALK_STATE = 65, SAP_WALK_FREE = 97, SAP_WALK_MAX = 1024;

// bits [o, o + 8) of the 256-bit little-endian number kw (bits outside [0, 256) are zero); o >= -8
__device__ __forceinline__ u32 sap_key_bits(const u32* __restrict__ kw, int o) {
    u32 r = 0;
#pragma unroll
    for (int b = 0; b < 8; b++) {
        const int i = o + b;
        if (i >= 0 && i < 256) r |= ((kw[i >> 5] >> (i & 31)) & 1u) << b;
    }
    return r;
}

// the walks of an instance, in order: (item, 0 = the path of the leaf as read / 1 = of the leaf as written)
__device__ __forceinline__ u32 sap_walk_list(const SapWalkJob& j, u32 capacity, u32* item, uint8_t* phase) {
    u32 nw = 0;
    for (u64 k = 0; k < j.num_items && nw < capacity; k++) {
        const u64 i = j.first_item + k;
        item[nw] = (u32)i; phase[nw++] = 0;
        if (j.items[i].rw && nw < capacity) { item[nw] = (u32)i; phase[nw++] = 1; }
    }
    return nw;
}

// A cycle per thread: the header bit, the free elements and the state before the cycle — the key part, and the running hash, which
// the builder's level walk left behind for every level of every walk (walk_hashes: no hashing here; the padding cycles carry the last
// walk's root, the state before cycle 0 is zero). grid = (ceil((cycles + 1) / 256), instances)
static __device__ __forceinline__ void k_sap_walk_cycles(const VB& vb, const SapWalkJob* __restrict__ jobs, u32 capacity) {
    __shared__ u32 s_item[SAP_WALK_MAX];
    __shared__ uint8_t s_phase[SAP_WALK_MAX];
    __shared__ u32 s_nw;
    const SapWalkJob j = jobs[vb.y];
    if (threadIdx.x == 0) s_nw = sap_walk_list(j, capacity, s_item, s_phase);
    __syncthreads();
    const u32 cycles = capacity * SAP_WALK_CYCLES, active = s_nw * SAP_WALK_CYCLES;
    const u32 c = vb.x * blockDim.x + threadIdx.x;
    if (c > cycles) return;
    uint8_t* st = j.state_before + (size_t)c * SAP_WALK_STATE;
    const u32 w = c / SAP_WALK_CYCLES, i = c % SAP_WALK_CYCLES;
    const bool in_walk = c < active;
    u32 key[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (in_walk) {
#pragma unroll
        for (int k = 0; k < 8; k++) key[k] = j.keys[8 * (u64)s_item[w] + k];
    }
    // key part of the state: (key << 1) >> i before cycle i >= 1 of a walk, zero before a leaf cycle and in the padding
    for (u32 k = 0; k < 33; k++) st[32 + k] = (in_walk && i) ? (uint8_t)sap_key_bits(key, (int)i + 8 * (int)k - 1) : 0;
    {   // hash part: what the previous cycle left = level (i - 1) of this walk, the previous walk's root before a leaf cycle
        const u32 cp = c == 0 ? 0 : min(c, active) - 1;  // the cycle whose result this is
        const u32 wp = cp / SAP_WALK_CYCLES, ip = cp % SAP_WALK_CYCLES;
        const bool zero = c == 0 || active == 0;
        const u32* h = j.walk_hashes + (((u64)s_item[zero ? 0 : wp] * 2 + (zero ? 0 : s_phase[wp])) * 257 + ip) * 8;
        for (u32 k = 0; k < 8; k++) {
            const u32 x = zero ? 0u : h[k];
            for (u32 b = 0; b < 4; b++) st[4 * k + b] = active == 0 && j.idle_root ? j.idle_root[4 * k + b] : (uint8_t)(x >> (8 * b));
        }
    }
    if (c == cycles) return;
    j.hdr_bits[c] = in_walk ? (i == 0 ? 1 : 0) : 2;
    uint8_t* fr = j.free_elems + (size_t)c * SAP_WALK_FREE;
    if (!in_walk) {
        for (u32 f = 0; f < SAP_WALK_FREE; f++) fr[f] = 0;
    } else if (i == 0) {  // the leaf cycle: index_be (8) || value_be (32) || zeros, then the key << 1 (33 bytes)
        const SapItem& it = j.items[s_item[w]];
        const u64 idx = s_phase[w] ? it.write_index : it.read_index;
        const u32* val = s_phase[w] ? it.written_value : it.read_value;
        for (u32 f = 0; f < 8; f++) fr[f] = (uint8_t)(idx >> (8 * (7 - f)));
        for (u32 f = 8; f < 40; f++) { const u32 b = 31 - (f - 8); fr[f] = (uint8_t)(val[b >> 2] >> (8 * (b & 3))); }
        for (u32 f = 40; f < 64; f++) fr[f] = 0;
        for (u32 k = 0; k < 33; k++) fr[64 + k] = (uint8_t)sap_key_bits(key, 8 * (int)k - 1);
    } else {  // a level cycle: the sibling
        const u32* sib = j.paths + ((u64)s_item[w] * 256 + (i - 1)) * 8;
        for (u32 f = 0; f < 32; f++) fr[f] = 0;
        for (u32 k = 0; k < 8; k++) { const u32 x = sib[k]; for (u32 b = 0; b < 4; b++) fr[32 + 4 * k + b] = (uint8_t)(x >> (8 * b)); }
        for (u32 f = 64; f < SAP_WALK_FREE; f++) fr[f] = 0;
    }
}

}  // namespace zkw
