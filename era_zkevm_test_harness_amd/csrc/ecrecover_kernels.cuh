// ecrecover_kernels.cuh — the EC SECTION of the ECRecover circuit (type 7) on the GPU: secp256k1 arithmetic over field-element-valued
// rows below the Keccak-f netlist and the queue section of the trace. Geometry and tables are the reference wrapper's
// (circuit_definitions/src/circuit_definitions/base_layer/ecrecover.rs:30-41,138-176); the circuit body is in the absent crate
// era-zkevm_circuits, so the placement is this library's own: format and statement in tools/gen_ecrecover_circuit.py, item semantics in
// include/zkw_ecrecover.h (shared with the test oracle the way nl_table_eval is).
//
// Fill (zkw_precompiles.hip ecrecover_synthesize_many; docs/KERNELS.md 3.19 "Round 6"):
//   k_ec_inputs    the value bytes of the cycle's four reads;
//   k_ec_chain     a WAVE per cycle (= request), a limb of a 256-bit value per lane: the MAIN items of the PRE segment, then the accumulator's
//                  trajectory — 256 double-and-add steps, 32 table additions — in Jacobian coordinates;
//   k_ec_affine    a lane per (cycle, point): the trajectory's points to affine onto the tape, where the segments' `out` states live;
//   k_ec_segments  item lists side by side, a lane per cycle: the MAIN items of the 289 segments after PRE, then the MUL rows of every segment;
//   k_ec_prepare   the tape's outputs -> the inputs of the byte netlist (the 64 key bytes, the mask, ok as FREE elements; the state the
//                  netlist will have after the cycle = the masked address: one Keccak-f per cycle here, for the boundary rows);
//   k_ec_leaves    the remaining items (range checks, byte decompositions, assertions) as lists of ~190, no 256-bit workspace;
//   k_ec_stream    a lane per (row, cycle): every cell of a row is a resolved reference into the tape (or a constant); the Xor8 lookups
//                  leave 16-bit keys behind;
//   k_ec_hist      an instance's Xor8 keys counted in LDS and the FixedBaseMul lookups onto the multiplicity column.
//   (k_ec_tape: the serial form — one lane walks a whole cycle in program order — behind ZKW_EC_SERIAL=1, same tape.)
//   From k_ec_segments' second launch to k_ec_stream the kernels run on the context's side stream, beside the netlist's fill.
// Check: k_ec_check_items (one lane per item instance: the relation from the cells alone), k_ec_check_rows (one lane per row: copies
// against the home cells / constants / the read queries' value bytes, empty cells, the lookups' multiplicities), k_ec_check_links (the
// netlist's FREE elements against the key bytes / mask / ok).
#pragma once
#include "../../include/zkw_ecrecover_circuit_spec.h"
#include "../../include/zkw_ecrecover.h"
#include "netlist_queue_kernels.cuh"
#include "ec_field.cuh"

namespace zkw {

EC_DEFINE_SPEC(h_ecs);

struct EcJob {
    const zkw_mem_query* mem_q;   // the witness's memory queries (6 per request: hash, v, r, s read; ok, address written)
    u64 first_request;            // of the instance
    u32 n_active;                 // requests of the instance; cycles beyond are idle (zero inputs)
    uint8_t* inputs;              // [capacity][128]: the value bytes of the four reads of every cycle (k_ec_inputs)
    u64* tape;                    // [EC_TAPE_PER_CYCLE][ec_tape_stride(capacity)]: value t of cycle c at tape[t * stride + c]
    u64* trace;                   // the slot
    uint8_t* hdr_bits;            // the netlist's inputs (NlPrepJob): [capacity]
    uint8_t* free_elems;          // [capacity][EK_FREE_PER_CYCLE]
    uint8_t* state_before;        // [capacity + 1][200]
};
// the cycles of an instance are interleaved on its tape: the lanes of a wave are cycles (k_ec_chain, k_ec_segments), or rows x cycles
// (k_ec_stream), and value t of the instance's cycles shares one cache line (capacity 7: 56 of 64 bytes) instead of seven 4 MB apart
__host__ __device__ __forceinline__ u32 ec_tape_stride(u32 capacity) { return (capacity + 7u) & ~7u; }

// grid (cycles, jobs) x 128: input byte k of cycle c = value byte k % 32 (little end first) of read k / 32; zeros for an idle cycle
static __device__ __forceinline__ void k_ec_inputs(const VB& vb, const EcJob* __restrict__ jobs) {
    const EcJob j = jobs[vb.y];
    const u32 c = vb.x, k = threadIdx.x;
    u32 v = 0;
    if (c < j.n_active) v = (j.mem_q[6 * (j.first_request + c) + k / 32].value[(k % 32) / 4] >> (8 * (k % 4))) & 0xFF;
    j.inputs[(size_t)c * 128 + k] = (uint8_t)v;
}

// grid (cycles / EC_TAPE_LANES, jobs): a lane per cycle; the workspace of the 256-bit arithmetic (every array with run-time indices) is
// a slice of LDS per lane. status: atomicMax of 1 + (job << 16 | cycle) for a cycle whose inputs have no witness
constexpr int EC_TAPE_LANES = 64;
static __device__ __forceinline__ void k_ec_tape(const VB& vb, const ec_spec* __restrict__ Sp, const EcJob* __restrict__ jobs, u32 capacity, u32* status) {
    __shared__ ec_ws s_ws[EC_TAPE_LANES];
    const EcJob j = jobs[vb.y];
    const u32 c = vb.x * blockDim.x + threadIdx.x;
    if (c >= capacity) return;
    const ec_spec S = *Sp;
    if (ec_eval_cycle_strided(&S, j.inputs + (size_t)c * 128, j.tape + c, ec_tape_stride(capacity), &s_ws[threadIdx.x])) atomicMax(status, 1u + (vb.y << 16 | c));
}


// ---- the fast form of the tape: the accumulator's trajectory first, then every segment on its own lane ------------------------------
// The serial kernel above spends its time in ~550 modular inversions per cycle (every quotient lambda of the affine additions), one after
// the other, because segment k needs the accumulator segment k - 1 leaves. The trajectory does not need the quotients:
//   k_ec_chain    a wave per cycle: the MAIN items of the PRE segment (what the globals — R, the bits of u2, the bytes of u1 — need: 262 of
//                 its 1 445 items, tools/gen_ecrecover_circuit.py split_segment), then the same double-and-add / table additions in
//                 JACOBIAN coordinates (no inversion): 288 points;
//   k_ec_affine   a lane per (cycle, point): the point to affine with an inversion of its own (288 x cycles lanes side by side cost what
//                 one costs; the serial Montgomery batch of round 5 was 2 ms of the chain's lane) onto the tape, where the segments'
//                 `out` states live;
//   k_ec_segments with every segment's input state in place, three launches over item lists (tools/gen_ecrecover_circuit.py split_segment):
//                 the MAIN items of the 289 other segments of every cycle side by side (the hints and what lies between them; their
//                 own out cells are rewritten with the same values); the MUL rows of every segment with what else they need; then
//                 the LEAVES — range checks, byte decompositions, assertions: nine tenths of a segment — in lists of ~190 items that
//                 write no common value, a lane each (1 400 lists per cycle). The walk of an item list is ~3 us an item on a wave
//                 whatever its lanes, so the rate is set by the waves in flight: the first two launches hold the 256-bit workspace in
//                 LDS (476 bytes a lane: five waves on a CU), the leaves need neither it nor the registers of the 256-bit arithmetic
//                 (k_ec_leaves). Whole segments as lists were 3.1 ms at best and 4.1 ms for 32 instances.
// Same tape, bit for bit (tests/test_ec_library_evaluator_host.py walks the library's evaluator in this order on the host).
struct EcChainScratch { ec_jac* pts; };  // [cycles of the call][EC_CHAIN_POINTS]
constexpr u32 EC_CHAIN_POINTS = 288;     // 256 double-and-add steps + 32 table additions
constexpr u32 EC_PART_ITEMS[EC_NUM_TYPES][EC_MAX_PARTS] = EC_PART_ITEMS_INIT;
struct EcTask { u32 run, inst, first, count; };  // items [first, first + count) of segment (run, inst)

// grid (cycles, jobs) x 64: a WAVE per cycle and a SIMD per wave. The chain is bound by the SIMD's issue rate, not by latency: a wave
// instruction costs its 4 (v_mad_u64_u32: 16) cycles whatever the number of active lanes. So
//  - the 256-bit arithmetic runs with a LIMB PER LANE (ec_field.cuh ecl: 8 multiply-adds on 16 lanes per product instead of 64 on one,
//    words moved by DPP row shifts, carries settled by a lookahead over the lanes' ballot): ~600 cycles per multiplication mod p instead
//    of ~2 100 with a value in one lane (the double-and-add loop was 3.1 ms of a 4.75 ms wave);
//  - a request has a wave of its own: lanes that shared a wave ran the mixed addition at every step (one of seven bits of u2 is set 99 % of
//    the time) instead of on half of them;
//  - the wave owns its SIMD: two chain waves on one SIMD take twice as long (eight-wave workgroups, a value in one lane: waves 0-3 end at
//    4.75 ms, waves 4-6, which share their SIMDs, at ~7). Per-wave end times of the other shapes (profiles/r06/ecrecover_steps.txt):
//    one-wave workgroups 4.7 .. 9 ms (the launch takes the slowest), four-wave workgroups with a CU to themselves (an LDS request of more
//    than half a CU) whole workgroups at 4.8 / 9.5 / 14 ms — as if their waves sat on four, two, one SIMD; an idle chip spreads one-wave
//    workgroups one to a SIMD (tools/probe_wave_placement), behind other kernels of the stream it evidently does not. What gives every
//    wave its SIMD: holding the SIMD's whole register file — the two moves below make the kernel's footprint 256 + 256 registers — so
//    that nobody else's wave fits beside it, whoever else is running (4.5 .. 6.6 ms with a value in one lane).
// The MAIN items of PRE (an item interpreter, one value per lane) run on lane 0 first.
static __device__ __forceinline__ void k_ec_chain(const VB& vb, const ec_spec* __restrict__ Sp, const EcJob* __restrict__ jobs, u32 capacity, u32* status, EcChainScratch sc) {
    asm volatile("v_mov_b32 v255, 0\n\tv_accvgpr_write_b32 a255, 0" ::: "v255", "a255");  // (the whole register file of the SIMD: see above)
    __shared__ ec_ws s_ws;
    const EcJob j = jobs[vb.y];
    const u32 c = vb.x, lane = threadIdx.x;
    if (c >= capacity || lane >= 16) return;  // (ecl: lanes 0 .. 15 of the wave)
    const ec_spec S = *Sp;
    const size_t ts = ec_tape_stride(capacity);
    u64* tape = j.tape + c;
    // PRE's MAIN items: an item walk on lane 0, but for the square root y = t^((p + 1) / 4) (EC_H_SQRT of include/zkw_ecrecover.h: ~500
    // multiplications) in between, which the wave computes with a limb per lane
    __shared__ u32 s_sqrt[10];  // t (8 words), the parity bit asked for, the tape index of the hint's 17 values
    ec_eval_ctx E;
    E.S = &S; E.tape = tape; E.ts = (u32)ts; E.in = j.inputs + (size_t)c * 128; E.W = &s_ws;
    E.base = S.runs[0].tape0; E.prev_base = 0; E.prev_type = 0; E.inst = 0;
    u32 bad = 0;
    if (lane == 0) {
        bad = ec_eval_items(&E, S.runs[0].type, 0, EC_PRE_SQRT_ITEM) != 0;
        if (!bad) {
            const ec_seg_type& T0 = S.types[S.runs[0].type];
            const uint32_t* w = S.items + T0.item0 + S.item_index[T0.index0 + EC_PRE_SQRT_ITEM];
            const ec_mod M = ec_modulus(0);
            uint32_t wide = 0;
            const ec_u256 t = ec_get_reduced(&E, w[1], &M, &wide);
            bad = wide != 0 || (w[0] & 15) != EC_I_HINT || (w[0] >> 24) != EC_H_SQRT;
#pragma unroll
            for (int i = 0; i < 8; i++) s_sqrt[i] = t.w[i];
            s_sqrt[8] = (u32)ec_get(&E, w[2]) & 1u;
            s_sqrt[9] = w[3];
        }
        if (bad) atomicMax(status, 1u + (vb.y << 16 | c));
    }
    __threadfence_block();  // (lane 0's values are read by the wave's other lanes below)
    if (__builtin_amdgcn_readlane(bad, 0)) return;
    const bool live = lane < 8;  // a value: limb i in lane i, zero in lanes 8 .. 15
    const u32 li = lane & 7;
    {
        const u32 t = live ? s_sqrt[li] : 0u;
        u32 y = ecl::pow_sqrt(t), e_nr = 0;
        if (__builtin_amdgcn_ballot_w64(ecl::mul(y, y) != t) & 0xFFull) {  // no root: a root of -t proves it (p = 3 mod 4)
            e_nr = 1;
            y = ecl::pow_sqrt(ecl::sub(0u, t));
        } else if (((u32)__builtin_amdgcn_readlane(y, 0) & 1u) != s_sqrt[8]) {
            y = ecl::sub(0u, y);
        }
        u64* out = tape + (size_t)(E.base + s_sqrt[9]) * ts;
        if (live) { out[(size_t)(2 * li) * ts] = y & 0xFFFFu; out[(size_t)(2 * li + 1) * ts] = y >> 16; }
        if (lane == 0) out[(size_t)16 * ts] = e_nr;
    }
    __threadfence_block();
    if (lane == 0) {
        bad = ec_eval_items(&E, S.runs[0].type, EC_PRE_SQRT_ITEM + 1, EC_PART_ITEMS[0][0] - EC_PRE_SQRT_ITEM - 1) != 0;
        if (bad) atomicMax(status, 1u + (vb.y << 16 | c));
    }
    __threadfence_block();
    if (__builtin_amdgcn_readlane(bad, 0)) return;
    const size_t slot = (size_t)vb.y * capacity + c;
    ec_jac* pts = sc.pts + slot * EC_CHAIN_POINTS;
    auto limb_of = [&](const uint32_t* idx) -> u32 { return live ? ((u32)tape[idx[2 * li] * ts] | ((u32)tape[idx[2 * li + 1] * ts] << 16)) : 0u; };
    const u32 rx = limb_of(S.globs + EC_GL_RX), ry = limb_of(S.globs + EC_GL_RY);
    u32 bits[8];  // of u2, bit 255 - k = step k: loaded once (a load per step would be two dependent ones inside the serial loop)
#pragma unroll
    for (int wd = 0; wd < 8; wd++) {
        u32 v = 0;
#pragma unroll
        for (int b = 0; b < 32; b++) v |= (u32)(tape[S.globs[EC_GL_BITS + 32 * wd + b] * ts] & 1) << b;
        bits[wd] = v;
    }
    u32 ax = live ? (S.bigs[EC_BIG_OX * 16 + 2 * li] | (S.bigs[EC_BIG_OX * 16 + 2 * li + 1] << 16)) : 0u;
    u32 ay = live ? (S.bigs[EC_BIG_OY * 16 + 2 * li] | (S.bigs[EC_BIG_OY * 16 + 2 * li + 1] << 16)) : 0u;
    u32 az = lane == 0 ? 1u : 0u;
#pragma unroll 1
    for (int wd = 7; wd >= 0; wd--) {
        const u32 word = (u32)__builtin_amdgcn_readfirstlane(bits[7]);  // (the word of this pass: the array is rotated below so that every index is a constant)
#pragma unroll 1
        for (int b = 31; b >= 0; b--) {
            const u32 k = 255u - (32u * (u32)wd + (u32)b);
            ecl::jdbl(ax, ay, az);
            if ((word >> b) & 1) ecl::jmadd(ax, ay, az, rx, ry);
            if (live) { pts[k].x.w[li] = ax; pts[k].y.w[li] = ay; pts[k].z.w[li] = az; }
        }
#pragma unroll
        for (int i = 7; i > 0; i--) bits[i] = bits[i - 1];
    }
    for (u32 C = 0; C < 32; C++) {
        const u32 b = (u32)__builtin_amdgcn_readfirstlane((u32)tape[S.globs[EC_GL_U1 + C] * ts]);
        if (b) {  // minus byte * 2^(8C) * G: the table point with its y negated
            const u32 tx = live ? S.fixed[((size_t)(8 * C + li) * 256 + b) * 2] : 0u, ty = live ? S.fixed[((size_t)(8 * C + li) * 256 + b) * 2 + 1] : 0u;
            ecl::jmadd(ax, ay, az, tx, ecl::sub(0u, ty));
        }
        if (live) { pts[256 + C].x.w[li] = ax; pts[256 + C].y.w[li] = ay; pts[256 + C].z.w[li] = az; }
    }
}

// grid (points x cycles of the call / 64): a lane per (cycle, point) — the point to affine, onto the tape where the `out` state of its
// segment lives. A point at infinity has no affine form (and the circuit no witness)
static __device__ __forceinline__ void k_ec_affine(const VB& vb, const ec_spec* __restrict__ Sp, const EcJob* __restrict__ jobs, u32 capacity, u32 n_cycles, u32* status, EcChainScratch sc) {
    const u32 lane = vb.x * blockDim.x + threadIdx.x;
    if (lane >= n_cycles * EC_CHAIN_POINTS) return;
    const u32 cyc = lane / EC_CHAIN_POINTS, k = lane % EC_CHAIN_POINTS;
    const u32 job = cyc / capacity, c = cyc % capacity;
    const ec_spec& S = *Sp;
    const ec_jac P = sc.pts[(size_t)cyc * EC_CHAIN_POINTS + k];
    if (ec_is_zero8(&P.z)) { atomicMax(status, 1u + (job << 16 | c)); return; }
    const ec_mod M = ec_modulus(0);
    const ec_u256 zi = ec_invmod(&P.z, &M, nullptr);  // binary extended Euclid, registers only (include/zkw_ecrecover.h)
    const ec_u256 zi2 = ecf::mul(zi, zi), zi3 = ecf::mul(zi2, zi);
    const ec_u256 x = ecf::mul(P.x, zi2), y = ecf::mul(P.y, zi3);
    const u32 run_i = k < 256 ? 1u : 2u, inst = k < 256 ? k : k - 256u;
    const ec_seg_type& T = S.types[S.runs[run_i].type];
    const size_t ts = ec_tape_stride(capacity);
    u64* seg = jobs[job].tape + c + (size_t)(S.runs[run_i].tape0 + inst * T.n_tape) * ts;
    const uint32_t* outs = S.outs + T.out0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        seg[outs[2 * i] * ts] = x.w[i] & 0xFFFFu; seg[outs[2 * i + 1] * ts] = x.w[i] >> 16;
        seg[outs[16 + 2 * i] * ts] = y.w[i] & 0xFFFFu; seg[outs[16 + 2 * i + 1] * ts] = y.w[i] >> 16;
    }
}

// grid (tasks, lanes' chunks of the call's cycles): lane = one cycle of the call (job-major), block = one task — a part of the items of one
// segment — so that a wave runs ONE item list. The workspace is 476 bytes a lane: five workgroups on a CU
static __device__ __forceinline__ void k_ec_segments(const VB& vb, const ec_spec* __restrict__ Sp, const EcJob* __restrict__ jobs, u32 capacity, u32 n_cycles, const EcTask* __restrict__ tasks, u32* status) {
    __shared__ ec_ws s_ws[EC_TAPE_LANES];
    const u32 lane = vb.y * blockDim.x + threadIdx.x;
    if (lane >= n_cycles) return;
    const u32 job = lane / capacity, c = lane % capacity;
    const EcJob j = jobs[job];
    const ec_spec S = *Sp;
    const EcTask T = tasks[vb.x];
    u32 prun, pinst;
    ec_prev_segment(&S, T.run, T.inst, &prun, &pinst);
    ec_eval_ctx E;
    E.S = &S; E.tape = j.tape + c; E.ts = ec_tape_stride(capacity); E.in = j.inputs + (size_t)c * 128; E.W = &s_ws[threadIdx.x];
    E.base = S.runs[T.run].tape0 + T.inst * S.types[S.runs[T.run].type].n_tape;
    E.prev_base = S.runs[prun].tape0 + pinst * S.types[S.runs[prun].type].n_tape;
    E.prev_type = S.runs[prun].type;
    E.inst = T.inst;
    if (ec_eval_items(&E, S.runs[T.run].type, T.first, T.count)) atomicMax(status, 1u + (job << 16 | c));
}
// the same over lists of small items only (a segment's LEAVES): no workspace, a quarter of the registers
static __device__ __forceinline__ void k_ec_leaves(const VB& vb, const ec_spec* __restrict__ Sp, const EcJob* __restrict__ jobs, u32 capacity, u32 n_cycles, const EcTask* __restrict__ tasks, u32* status) {
    const u32 lane = vb.y * blockDim.x + threadIdx.x;
    if (lane >= n_cycles) return;
    const u32 job = lane / capacity, c = lane % capacity;
    const EcJob j = jobs[job];
    const ec_spec S = *Sp;
    const EcTask T = tasks[vb.x];
    u32 prun, pinst;
    ec_prev_segment(&S, T.run, T.inst, &prun, &pinst);
    ec_eval_ctx E;
    E.S = &S; E.tape = j.tape + c; E.ts = ec_tape_stride(capacity); E.in = j.inputs + (size_t)c * 128; E.W = nullptr;
    E.base = S.runs[T.run].tape0 + T.inst * S.types[S.runs[T.run].type].n_tape;
    E.prev_base = S.runs[prun].tape0 + pinst * S.types[S.runs[prun].type].n_tape;
    E.prev_type = S.runs[prun].type;
    E.inst = T.inst;
    if (ec_eval_items_of(&E, S.runs[T.run].type, T.first, T.count, 1)) atomicMax(status, 1u + (job << 16 | c));
}

// grid (cycles / 64, jobs): the netlist's inputs of a cycle from its tape
static __device__ __forceinline__ void k_ec_prepare(const VB& vb, const ec_spec* __restrict__ Sp, const EcJob* __restrict__ jobs, u32 capacity) {
    const EcJob j = jobs[vb.y];
    const u32 c = vb.x * blockDim.x + threadIdx.x;
    if (c > capacity) return;
    if (c == 0)
        for (int k = 0; k < 200; k++) j.state_before[k] = 0;
    if (c == capacity) return;
    const ec_spec& S = *Sp;
    const size_t ts = ec_tape_stride(capacity);
    const u64* tape = j.tape + c;
    const u64 ok = tape[S.globs[EC_GL_OK] * ts], mask = tape[S.globs[EC_GL_MASK] * ts];
    uint8_t* f = j.free_elems + (size_t)c * EK_FREE_PER_CYCLE;
    u64 st[25];
    for (int k = 0; k < 25; k++) st[k] = 0;
    // the key bytes Q.x || Q.y, big end first, off the 16-bit limbs of the state POST leaves (its MAIN items: the byte cells themselves are
    // among POST's leaves, which may still be on their way on the side stream)
    const u32 post0 = S.runs[EC_NUM_RUNS - 1].tape0;
    const uint32_t* post_out = S.outs + S.types[S.runs[EC_NUM_RUNS - 1].type].out0;
#pragma unroll
    for (int k = 0; k < 64; k++) {  // (unrolled: st[] stays in registers)
        const int jb = k < 32 ? 31 - k : 63 - k;  // byte of the coordinate, little end first
        const u64 b = (tape[(post0 + post_out[(k < 32 ? 0 : 16) + jb / 2]) * ts] >> (8 * (jb & 1))) & 0xFF;
        f[k] = (uint8_t)b;
        st[k / 8] |= b << (8 * (k % 8));
    }
    f[EK_FREE_MASK] = (uint8_t)mask;
    f[EK_FREE_OK] = (uint8_t)ok;
    j.hdr_bits[c] = c < j.n_active ? 0 : 2;  // idle: the queue operations of the cycle are disabled
    st[8] ^= 0x01;                           // Keccak padding of a 64-byte message: byte 64 = 0x01, byte 135 = 0x80
    st[16] ^= 0x80ull << 56;
    keccak_f1600(st);
    uint8_t* nx = j.state_before + (size_t)(c + 1) * 200;
    for (int k = 0; k < 200; k++) nx[k] = 0;
    for (int k = 12; k < 32; k++) nx[k] = (uint8_t)(st[k / 8] >> (8 * (k % 8))) & (uint8_t)mask;
    nx[EK_STATE_OK] = (uint8_t)ok;
}

#define EC_TR(col, row) trace[(size_t)(col) * n_rows + (size_t)(row)]

// ---- the rows: every cell of a row is a reference (into the cycle's tape, an input byte) or a constant. The references are RESOLVED once
// per device (zkw_precompiles.hip ec_get: segment instance, previous segment, instance-dependent globals all folded in), so the kernel
// does no decoding: entry = kind << 30 | payload — 0 a constant (empty cells: 0), 1 a tape value of the cycle, 2 an input byte.
struct EcStreamDev {
    const u32* refs;            // [EC_ROW_CELLS][EC_ROWS_PER_CYCLE]
    const uint16_t* row_table;  // [EC_ROWS_PER_CYCLE]: the table the row's 16 slots look up (0: none)
    const uint16_t* xor_index;  // [EC_ROWS_PER_CYCLE]: the row's place among the cycle's Xor8 rows (the others: 0xFFFF)
    u32 n_xor_rows;
    const u32* fix_rows;        // the cycle's rows that look up a FixedBaseMul table: row | table << 16
    u32 n_fix_rows;
};
// grid (rows of a cycle / 64, cycles / 8, jobs) x 512: a lane per (row, cycle), the cycle fastest — eight neighbouring lanes read value t of
// eight cycles from one line of the interleaved tape, and a wave stores eight rows of each of its cycles (64 contiguous bytes per cycle).
// Multiplicities: the Xor8 lookups (16 per row on most rows: 27 M of a 32-instance call) leave 16-bit keys behind for k_ec_hist — as
// global atomics on the multiplicity column they were 1.7 of the kernel's 3.8 ms; the FixedBaseMul lookups (8 rows per FIX segment) are
// counted by k_ec_hist too, from the cells this kernel writes: the kernel itself never touches the multiplicity column, so it may run
// beside the netlist's finish kernel, which stores that column.
constexpr int EC_STREAM_ROWS = 64, EC_STREAM_THREADS = 8 * EC_STREAM_ROWS;
static __device__ __forceinline__ void k_ec_stream(const VB& vb, EcStreamDev sd, const EcJob* __restrict__ jobs, u32 capacity, size_t n_rows, size_t first_row, uint16_t* __restrict__ keybuf) {
    const EcJob j = jobs[vb.z];
    const u32 c = vb.y * 8 + (threadIdx.x & 7), r = vb.x * EC_STREAM_ROWS + (threadIdx.x >> 3);
    if (r >= EC_ROWS_PER_CYCLE || c >= capacity) return;
    u64* trace = j.trace;
    const uint8_t* in = j.inputs + (size_t)c * 128;  // (only the PRE segment's rows name input bytes)
    const size_t ts = ec_tape_stride(capacity);
    const u64* tape = j.tape + c;
    const u32* refs = sd.refs + r;
    const size_t tr = first_row + (size_t)c * EC_ROWS_PER_CYCLE + r;
    auto value = [&](u32 col) -> u64 {
        const u32 e = refs[(size_t)col * EC_ROWS_PER_CYCLE], a = e & 0x3FFFFFFFu;
        return (e >> 30) == 0 ? (u64)a : (e >> 30) == 1 ? tape[a * ts] : (u64)in[a];
    };
#pragma unroll 8
    for (u32 col = 0; col < EC_G; col++) EC_TR(col, tr) = value(col);
    const u32 tb = sd.row_table[r];
    u32 keys[EC_R / 2];
#pragma unroll
    for (u32 slot = 0; slot < EC_R; slot++) {  // the row's lookups: the inputs of a slot are its first two cells
        const u32 col = EC_G + EC_W * slot;
        const u64 a = value(col), b = value(col + 1), o = value(col + 2);
        EC_TR(col, tr) = a;
        EC_TR(col + 1, tr) = b;
        EC_TR(col + 2, tr) = o;
        const u32 key = (u32)a | ((u32)b << 8);
        if (slot & 1) keys[slot / 2] |= key << 16;
        else keys[slot / 2] = key;
    }
    if (tb == EC_T_XOR8) {
        uint4* dst = reinterpret_cast<uint4*>(keybuf + (((size_t)vb.z * capacity + c) * sd.n_xor_rows + sd.xor_index[r]) * EC_R);
        dst[0] = make_uint4(keys[0], keys[1], keys[2], keys[3]);
        dst[1] = make_uint4(keys[4], keys[5], keys[6], keys[7]);
    }
}

// grid (2 halves of the Xor8 table, jobs) x 1024: an instance's Xor8 keys (k_ec_stream) counted in LDS, the half's 32 768 bins added onto the
// instance's multiplicity column — this workgroup's rows of it and nobody else's at that time (the netlist's own multiplicities are
// there already: k_nl_finish, earlier on the stream). The second workgroup also counts the FixedBaseMul lookups, from the rows' cells
// (16 slots a row: the byte's, and 15 padding slots that look up entry 0): a few thousand atomics on rows of the column nobody else adds to.
constexpr int EC_HIST_THREADS = 1024, EC_HIST_HALF = 32768;
static __device__ __forceinline__ void k_ec_hist(const VB& vb, EcStreamDev sd, const EcJob* __restrict__ jobs, u32 capacity, size_t n_rows, size_t first_row, u32 mult_col, const uint16_t* __restrict__ keybuf) {
    __shared__ u32 s_bins[EC_HIST_HALF];
    const u32 half = vb.x, t = threadIdx.x, n_xor_rows = sd.n_xor_rows;
    u64* trace = jobs[vb.y].trace;
    if (half == 1)
        for (u32 i = t; i < capacity * sd.n_fix_rows * EC_R; i += EC_HIST_THREADS) {
            const u32 slot = i % EC_R, fr = sd.fix_rows[(i / EC_R) % sd.n_fix_rows], c = i / (EC_R * sd.n_fix_rows);
            const size_t tr = first_row + (size_t)c * EC_ROWS_PER_CYCLE + (fr & 0xFFFFu);
            atomicAdd(reinterpret_cast<unsigned long long*>(&EC_TR(mult_col, ec_table_key(fr >> 16, EC_TR(EC_G + EC_W * slot, tr), EC_TR(EC_G + EC_W * slot + 1, tr)))), 1ull);
        }
    for (u32 i = t; i < EC_HIST_HALF; i += EC_HIST_THREADS) s_bins[i] = 0;
    __syncthreads();
    const size_t per_job = (size_t)capacity * n_xor_rows * EC_R;  // a multiple of 16 keys
    const uint4* keys = reinterpret_cast<const uint4*>(keybuf + (size_t)vb.y * per_job);
    for (size_t i = t; i < per_job / 8; i += EC_HIST_THREADS) {
        const uint4 v = keys[i];
        const u32 w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const u32 lo = w[k] & 0xFFFFu, hi = w[k] >> 16;
            if ((lo >> 15) == half) atomicAdd(&s_bins[lo & 32767u], 1u);
            if ((hi >> 15) == half) atomicAdd(&s_bins[hi & 32767u], 1u);
        }
    }
    __syncthreads();
    for (u32 i = t; i < EC_HIST_HALF; i += EC_HIST_THREADS)
        if (const u32 n = s_bins[i]) EC_TR(mult_col, (size_t)half * EC_HIST_HALF + i) += n;  // (Xor8 sits at row 0 of the stacked tables: ec_table_key)
}

// ------------------------------------------------------------------------------------------------ checker
// grid (items / 256, segment instances of a cycle, cycles): one lane per item instance
static __global__ __launch_bounds__(256) void k_ec_check_items(const ec_spec* __restrict__ Sp, const u64* __restrict__ trace, u32 capacity, size_t n_rows, size_t first_row,
                                                               CheckResult* res) {
    const ec_spec S = *Sp;
    u32 seg = blockIdx.y, run = 0;
    while (seg >= S.runs[run].count) { seg -= S.runs[run].count; run++; }
    const ec_seg_type& T = S.types[S.runs[run].type];
    const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T.n_items) return;
    const uint32_t* w = S.items + T.item0 + S.item_index[T.index0 + i];
    const size_t seg0 = first_row + (size_t)blockIdx.z * EC_ROWS_PER_CYCLE + S.runs[run].row0 + (size_t)seg * T.n_rows;
    const ec_row_view v = {trace, n_rows, seg0 + ((w[0] >> 4) & 0xFFF)};
    if (ec_check_item(&S, w, &v, seg)) flag_bad(res, (w[0] & 15) == EC_I_LOOKUP ? 1 : 7, i, v.row);
}

// grid (rows of a cycle / 64, cycles): one lane per row — copies, empty cells, the multiplicities of the row's lookups
static __global__ __launch_bounds__(64) void k_ec_check_rows(const ec_spec* __restrict__ Sp, const NlDev* __restrict__ devp, nlq_desc qd, const u64* __restrict__ trace, u32 capacity,
                                                             size_t n_rows, size_t first_row, u32* __restrict__ hist, CheckResult* res) {
    const u32 c = blockIdx.y, r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= EC_ROWS_PER_CYCLE) return;
    const ec_spec& S = *Sp;
    const nl_spec& N = devp->s;
    u32 run, inst, row;
    ec_locate_row(&S, r, &run, &inst, &row);
    const ec_seg_type& T = S.types[S.runs[run].type];
    u32 prun, pinst;
    ec_prev_segment(&S, run, inst, &prun, &pinst);
    const u32 base = S.runs[run].tape0 + inst * T.n_tape, pbase = S.runs[prun].tape0 + pinst * S.types[S.runs[prun].type].n_tape, ptype = S.runs[prun].type;
    const uint32_t* cells = S.cells + T.cell0 + (size_t)row * EC_ROW_CELLS;
    const size_t cyc0 = first_row + (size_t)c * EC_ROWS_PER_CYCLE, tr = cyc0 + r;
    for (u32 col = 0; col < EC_ROW_CELLS; col++) {
        const u32 ref = cells[col];
        const u64 x = EC_TR(col, tr);
        if (ref == EC_NONE) { if (x) flag_bad(res, 6, col, tr); continue; }
        const u32 t = ec_ref_tape(&S, ref, base, pbase, ptype, inst);
        if (t != EC_NONE) {
            u32 hr, hc;
            ec_home_of_tape(&S, t, &hr, &hc);
            if (x != EC_TR(hc, cyc0 + hr)) flag_bad(res, 2, col, tr);
        } else if ((ref >> 28) == EC_K_IN) {
            const u32 k = ref & 0xFFFF, h = S.in_home[k];
            if (run == 0 && row == (h >> 8) && col == (h & 0xFF)) {  // the home: a copy of value byte k % 32 of read k / 32 (queue section)
                const u32 op = 1 + k / 32, cell = NLQ_MEM_NIBBLE0 + k % 32;
                const size_t qrow = NLQ_ROW(&N, capacity, nlq_op_row0(&qd, N.g, op) + cell / N.g, c);
                if (x != EC_TR(cell % N.g, qrow)) flag_bad(res, 2, 0x1000 + k, tr);
            } else if (x != EC_TR(h & 0xFF, cyc0 + (h >> 8))) flag_bad(res, 2, col, tr);
        } else if (x != ec_ref_const(&S, ref, nullptr)) flag_bad(res, 2, col, tr);
    }
    const u32 tb = ec_row_table(&S, run, inst, row);
    if (tb)
        for (u32 slot = 0; slot < EC_R; slot++) {
            const u64 a = EC_TR(EC_G + EC_W * slot, tr), b = EC_TR(EC_G + EC_W * slot + 1, tr);
            if (a < 256 && (tb != EC_T_XOR8 || b < 256)) atomicAdd(&hist[ec_table_key(tb, a, b)], 1u);  // (a bad key: flagged by its item)
        }
}

// grid (cycles): lane k = FREE element k of the cycle's netlist against the EC value it copies (key byte / mask / ok)
static __global__ __launch_bounds__(128) void k_ec_check_links(const ec_spec* __restrict__ Sp, const NlDev* __restrict__ devp, const NlqFreeHome* __restrict__ free_home,
                                                               const u64* __restrict__ trace, size_t n_rows, size_t first_row, CheckResult* res) {
    const u32 c = blockIdx.x, k = threadIdx.x;
    if (k >= EK_FREE_PER_CYCLE) return;
    const ec_spec& S = *Sp;
    const NlqFreeHome fh = free_home[k];
    if (fh.row == 0xFFFF) return;
    const u32 t = k < 64 ? S.runs[EC_NUM_RUNS - 1].tape0 + S.key_byte[k] : S.globs[k == EK_FREE_MASK ? EC_GL_MASK : EC_GL_OK];
    u32 hr, hc;
    ec_home_of_tape(&S, t, &hr, &hc);
    const size_t nrow = (size_t)c * devp->s.rows_per_cycle + fh.row;
    if (EC_TR(fh.col, nrow) != EC_TR(hc, first_row + (size_t)c * EC_ROWS_PER_CYCLE + hr)) flag_bad(res, 2, 0x2000 + k, nrow);
}
#undef EC_TR

}  // namespace zkw
