#!/usr/bin/env python3
"""Generates include/zkw_keccak_circuit_spec.h (Keccak256RoundFunction, type 5) and include/zkw_linear_hasher_circuit_spec.h
(L1MessagesHasher, type 13): ONE set of netlists — a Keccak-f[1600] call as absorb / 24 x round / select steps on bytes — on
the reference's two geometries (keccak256_round_function.rs:28-39,120-140: 86 + 3 x 14; linear_hasher.rs:28-39,125-138:
66 + 3 x 26) and its table set Xor8 / And8 / ByteSplit<1..4> (132 096 rows = `total_tables_len` of vk_5.json / vk_13.json).
Format: tools/netlist.py.

State: 200 bytes, lane (x, y) = bytes 8 (x + 5 y) .., least significant first. Steps of a cycle:
  IN     in = reset ? 0 : prev (And8 with the header mask 255 - 255 * reset), then in ^ block on the first 136 bytes (FREE bytes)
  ROUND  theta, rho, pi, chi, iota; the round constant's bytes are the step's constants (x 24)
  OUT    out = idle ? state before the cycle : f  =  f ^ (mask & (f ^ before)), mask = 255 * idle
  * rotl by 8 q + s, s != 0: every source byte is split at bit 8 - s — one ByteSplit<8 - s> lookup for s = 4..7, ByteSplit<4>
    and ByteSplit<4 - s> of its high nibble for s = 1..3 (the table set stops at <4>) — and a result byte is recomposed by a
    gate from the low part of one byte and the high part of its neighbour (bounded by construction).
  * chi: a ^ (~b & c) = (a ^ c) ^ (b & c): And8 + 2 x Xor8 per byte (the table set has no and-not).
"""
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import netlist as nl  # noqa: E402

RC = [0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000, 0x000000000000808B, 0x0000000080000001,
      0x8000000080008081, 0x8000000000008009, 0x000000000000008A, 0x0000000000000088, 0x0000000080008009, 0x000000008000000A,
      0x000000008000808B, 0x800000000000008B, 0x8000000000008089, 0x8000000000008003, 0x8000000000008002, 0x8000000000000080,
      0x000000000000800A, 0x800000008000000A, 0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008]
ROT = [0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14]  # [x + 5y]
HDR_M0, HDR_M1 = 2, 3


def rotl(st, lane, r):
    q, s = divmod(r, 8)
    if s == 0:
        return [lane[(i - q) % 8] for i in range(8)]
    lo_parts, hi_parts = [], []  # of source byte j: (refs with shifts for x mod 2^(8-s)), ref of x >> (8 - s)
    for j in range(8):
        if s >= 4:
            lo, hi = st.lookup(f"BYTESPLIT_{8 - s}", lane[j])
            lo_parts.append([(lo, 0)])
            hi_parts.append(hi)
        else:
            lo4, hi4 = st.lookup("BYTESPLIT_4", lane[j])
            l2, h2 = st.lookup(f"BYTESPLIT_{4 - s}", hi4)
            lo_parts.append([(lo4, 0), (l2, 4)])
            hi_parts.append(h2)
    out = []
    for i in range(8):
        a, b = (i - q) % 8, (i - q - 1) % 8
        known = [(r_, sh + s, +1) for r_, sh in lo_parts[a]] + [(hi_parts[b], 0, +1)]
        out.append(st.gate(known, [0])[0])
    return out


def build_in(tables):
    st = nl.StepType("in", tables)
    ins = [st.lookup("AND8", nl.hdr(HDR_M0), nl.prev(k)) for k in range(200)]
    st.out = [st.lookup("XOR8", ins[k], nl.free(k)) if k < 136 else ins[k] for k in range(200)]
    return st


def build_round(tables):
    st = nl.StepType("round", tables)
    S = lambda x, y, k: nl.prev(8 * (x + 5 * y) + k)  # noqa: E731
    C = {}
    for x in range(5):
        for k in range(8):
            t = st.lookup("XOR8", S(x, 0, k), S(x, 1, k))
            for y in (2, 3, 4):
                t = st.lookup("XOR8", t, S(x, y, k))
            C[x, k] = t
    R1 = {x: rotl(st, [C[x, k] for k in range(8)], 1) for x in range(5)}
    D = {(x, k): st.lookup("XOR8", C[(x - 1) % 5, k], R1[(x + 1) % 5][k]) for x in range(5) for k in range(8)}
    A = {(x, y): [st.lookup("XOR8", S(x, y, k), D[x, k]) for k in range(8)] for y in range(5) for x in range(5)}
    B = {}
    for x in range(5):
        for y in range(5):
            B[y, (2 * x + 3 * y) % 5] = rotl(st, A[x, y], ROT[x + 5 * y])
    E = {}
    for y in range(5):
        for x in range(5):
            for k in range(8):
                b0, b1, b2 = B[x, y][k], B[(x + 1) % 5, y][k], B[(x + 2) % 5, y][k]
                t = st.lookup("AND8", b1, b2)
                u = st.lookup("XOR8", b0, b2)
                E[x, y, k] = st.lookup("XOR8", u, t)
    for k in range(8):
        E[0, 0, k] = st.lookup("XOR8", E[0, 0, k], nl.rc(k))
    st.out = [E[idx // 8 % 5, idx // 40, idx % 8] for idx in range(200)]
    return st


def build_out(tables):
    st = nl.StepType("out", tables)
    out = []
    for k in range(200):
        d = st.lookup("XOR8", nl.prev(k), nl.cyc(k))
        e = st.lookup("AND8", nl.hdr(HDR_M1), d)
        out.append(st.lookup("XOR8", nl.prev(k), e))
    st.out = out
    return st


def keccak_f(state_bytes):
    a = [int.from_bytes(bytes(state_bytes[8 * i:8 * i + 8]), "little") for i in range(25)]
    rol = lambda v, n: ((v << n) | (v >> (64 - n))) & (2**64 - 1) if n else v  # noqa: E731
    for rnd in range(24):
        c = [a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20] for x in range(5)]
        d = [c[(x - 1) % 5] ^ rol(c[(x + 1) % 5], 1) for x in range(5)]
        a = [a[i] ^ d[i % 5] for i in range(25)]
        b = [0] * 25
        for x in range(5):
            for y in range(5):
                b[y + 5 * ((2 * x + 3 * y) % 5)] = rol(a[x + 5 * y], ROT[x + 5 * y])
        a = [b[i] ^ (~b[(i % 5 + 1) % 5 + 5 * (i // 5)] & b[(i % 5 + 2) % 5 + 5 * (i // 5)]) for i in range(25)]
        a[0] ^= RC[rnd]
    return [byte for v in a for byte in v.to_bytes(8, "little")]


def make_spec(prefix, general_cols, lookups_per_row):
    tables = nl.keccak_tables()
    spec = nl.Spec(prefix, general_cols, 3, lookups_per_row, tables, 200, (255, -255, 0, 255))
    k_in, k_round, k_out = spec.add_step_type(build_in(tables)), spec.add_step_type(build_round(tables)), spec.add_step_type(build_out(tables))
    spec.cycle = [(k_in, [])] + [(k_round, list(RC[r].to_bytes(8, "little"))) for r in range(24)] + [(k_out, [])]
    return spec


def self_check(spec):
    rng = random.Random(1)
    for _ in range(2):
        st = [rng.randrange(256) for _ in range(200)]
        blk = [rng.randrange(256) for _ in range(136)]
        frees = [blk] + [[]] * 25
        absorbed = [st[k] ^ blk[k] if k < 136 else st[k] for k in range(200)]
        assert spec.evaluate_cycle(st, frees, 0, 0) == keccak_f(absorbed), "netlist != Keccak-f[1600]"
        assert spec.evaluate_cycle(st, frees, 1, 0) == keccak_f(blk + [0] * 64)   # reset: absorb into the zero state
        assert spec.evaluate_cycle(st, frees, 0, 1) == st                          # idle carries the state


CIRCUITS = {"KC": (86, 14, "zkw_keccak_circuit_spec.h", "Keccak256RoundFunction: 86 + 3 x 14 columns"),
            "LH": (66, 26, "zkw_linear_hasher_circuit_spec.h", "L1MessagesHasher: the same steps on 66 + 3 x 26 columns")}


def emit(prefix, path=None):
    g, r, header, circuit = CIRCUITS[prefix]
    spec = make_spec(prefix, g, r)
    self_check(spec)
    path = path or os.path.join(nl.root(), "include", header)
    spec.emit(path, f"tools/gen_keccak_circuit.py ({circuit})")
    return spec, path


def main():
    for prefix in CIRCUITS:
        spec, path = emit(prefix)
        print(spec.stats(), path)


if __name__ == "__main__":
    main()
