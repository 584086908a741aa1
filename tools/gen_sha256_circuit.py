#!/usr/bin/env python3
"""Generates include/zkw_sha256_circuit_spec.h — the layout of the Sha256RoundFunction trace that libzkw emits ("zkw trace
v3", circuit type 6): ONE netlist per cycle (= one SHA-256 compression), two kinds of operations.

Geometry of the reference wrapper (circuit_definitions/.../base_layer/sha256_round_function.rs:28-39,52-134): 2^20 rows,
capacity 2206 cycles (geometry_config.rs), lookups next to general-purpose gates (UIntXAddGate<32> among them). The circuit
body lives in the absent crate era-zkevm_circuits: placement and tables are OUR design.

Operations of a cycle:
  * LOOKUP {table, a, b} -> c: width-3 byte lookups, 14 per row, one table per row. Tables (2^16 rows, row a * 256 + b):
    1 XOR8, 2 ANDN8 (~a & b), 3..9 ROT<s> (((a << s) & 0xff) | (b >> (8 - s)), s = 1..7), 10 AND8.
    A 32-bit word is four bytes, least significant first; rotr by n = rotl by 32 - n = 8q + s: byte k of the result is
    ROT<s>(byte k - q, byte k - q - 1) (indices mod 4); shr is the same with zero above the top byte.
  * ADD gates in the general-purpose columns of the same rows (two gates of 43 columns per row): up to seven byte-wise
    operands + a 32-bit constant = out (4 bytes) + 2^32 * carry. One gate per SHA-256 addition chain:
      e' = d + h + S1 + ch + K[i] + W[i],  a' = h + S1 + ch + K[i] + W[i] + S0 + maj,  W[i] = W[i-16] + s0 + W[i-7] + s1,
      H'[j] = H[j] + v[j].
    Output bytes are range-checked by the lookups that consume them; the generator adds XOR(x, 0) lookups for the ones no
    lookup consumes.
References (uint16): 0x0000.. output of lookup j; 0x8000 + 4g + b: byte b of gate g; 0x9000 + f: header field f;
0xA000 + k: byte k of the chaining state after the previous cycle (BND_IN for cycle 0); 0xB000 + i: free witness byte i
(the 64 message bytes of the block); 0xC000 + v: the constant v.
Statement per cycle: in = reset ? IV : prev; H' = compress(in, block); out = idle ? prev : H' (masks as in the type-5 trace).
The generator checks the netlist against hashlib.sha256 before writing the header.
"""
import hashlib
import os
import random
import struct

LOOKUPS_PER_ROW = 14
G = 86
GATE_COLS = 43
T_XOR, T_ANDN, T_AND = 1, 2, 10
T_ROT = lambda s: 2 + s  # noqa: E731
N_TABLES = 10
R_GATE, R_HDR, R_PREV, R_FREE, R_CONST = 0x8000, 0x9000, 0xA000, 0xB000, 0xC000
HDR_RESET, HDR_IDLE, HDR_MASK_R, HDR_MASK_NR, HDR_MASK_I, HDR_MASK_A = range(6)
K = [0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
     0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
     0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
     0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
     0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
     0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
     0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2]
IV = [0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19]


def table_fn(t, a, b):
    if t == T_XOR:
        return a ^ b
    if t == T_ANDN:
        return (~a & 0xFF) & b
    if t == T_AND:
        return a & b
    s = t - 2
    return ((a << s) & 0xFF) | (b >> (8 - s))


class Netlist:
    def __init__(self):
        self.ops = []    # (table, a, b)
        self.gates = []  # (operands: list of 4-byte ref lists, constant)

    def op(self, t, a, b):
        self.ops.append((t, a, b))
        return ("op", len(self.ops) - 1)

    def xor(self, x, y):
        return [self.op(T_XOR, x[k], y[k]) for k in range(4)]

    def word_op(self, t, x, y):
        return [self.op(t, x[k], y[k]) for k in range(4)]

    def rotr(self, x, n, shift=False):
        """rotate (or shift) right by n: bytes least significant first"""
        q, s = divmod((32 - n) % 32, 8)
        out = []
        for k in range(4):
            hi_i, lo_i = k - q, k - q - 1  # byte that supplies the upper part (shifted left by s), the one below it
            if shift:
                # shr n: result byte k = bits of x >> n; as a left rotation by 32 - n with the wrapped-around bytes zeroed:
                # a source byte index that wrapped (index + 4 used) carries no bits
                hi = x[hi_i] if 0 <= hi_i < 4 and False else None
            if not shift:
                a, b = x[hi_i % 4], x[lo_i % 4]
                out.append(a if s == 0 else self.op(T_ROT(s), a, b))
            else:
                # shr by n = 8 * qn + sn: byte k = (x[k + qn] >> sn) | (x[k + qn + 1] << (8 - sn)), zero beyond byte 3
                qn, sn = divmod(n, 8)
                lo = x[k + qn] if k + qn < 4 else ("const", 0)
                hi = x[k + qn + 1] if k + qn + 1 < 4 else ("const", 0)
                if sn == 0:
                    out.append(lo)
                elif lo == ("const", 0) and hi == ("const", 0):
                    out.append(("const", 0))
                else:
                    out.append(self.op(T_ROT(8 - sn), hi, lo))  # ((hi << (8 - sn)) & 0xff) | (lo >> sn)
        return out

    def add(self, operands, constant=0):
        self.gates.append((operands, constant))
        g = len(self.gates) - 1
        return [("gate", g, b) for b in range(4)]


def build_cycle():
    nl = Netlist()
    hdr = lambda f: ("hdr", f)  # noqa: E731
    # chaining state: in = reset ? IV : prev
    inb = []
    for k in range(32):
        p = nl.op(T_ANDN, hdr(HDR_MASK_R), ("prev", k))
        q = nl.op(T_ANDN, hdr(HDR_MASK_NR), ("const", (IV[k // 4] >> (8 * (k % 4))) & 0xFF))
        inb.append(nl.op(T_XOR, p, q))
    H = [inb[4 * j:4 * j + 4] for j in range(8)]
    # message words: block byte 4i + 3 - b is byte b (least significant first) of W[i]; loaded through XOR(x, 0) (range check + home)
    load = [nl.op(T_XOR, ("free", i), ("const", 0)) for i in range(64)]
    W = [[load[4 * i + 3 - b] for b in range(4)] for i in range(16)]
    for i in range(16, 64):
        x, y = W[i - 15], W[i - 2]
        s0 = nl.xor(nl.xor(nl.rotr(x, 7), nl.rotr(x, 18)), nl.rotr(x, 3, shift=True))
        s1 = nl.xor(nl.xor(nl.rotr(y, 17), nl.rotr(y, 19)), nl.rotr(y, 10, shift=True))
        W.append(nl.add([W[i - 16], s0, W[i - 7], s1]))
    a, b, c, d, e, f, g, h = H
    for i in range(64):
        S1 = nl.xor(nl.xor(nl.rotr(e, 6), nl.rotr(e, 11)), nl.rotr(e, 25))
        ch = nl.xor(g, nl.word_op(T_AND, e, nl.xor(f, g)))              # g ^ (e & (f ^ g))
        S0 = nl.xor(nl.xor(nl.rotr(a, 2), nl.rotr(a, 13)), nl.rotr(a, 22))
        maj = nl.xor(nl.word_op(T_AND, a, nl.xor(b, c)), nl.word_op(T_AND, b, c))  # (a & (b ^ c)) ^ (b & c)
        new_e = nl.add([d, h, S1, ch, W[i]], K[i])
        new_a = nl.add([h, S1, ch, W[i], S0, maj], K[i])
        a, b, c, d, e, f, g, h = new_a, a, b, c, new_e, e, f, g
    v = [a, b, c, d, e, f, g, h]
    Hn = [nl.add([H[j], v[j]]) for j in range(8)]
    hn = [Hn[k // 4][k % 4] for k in range(32)]
    # out = idle ? prev : H'
    out = []
    for k in range(32):
        t = nl.op(T_ANDN, hdr(HDR_MASK_I), hn[k])
        u = nl.op(T_ANDN, hdr(HDR_MASK_A), ("prev", k))
        out.append(nl.op(T_XOR, t, u))
    # range checks for gate outputs no lookup consumes
    used = set()
    for t, x, y in nl.ops:
        for r in (x, y):
            if r[0] == "gate":
                used.add((r[1], r[2]))
    for gi in range(len(nl.gates)):
        for bb in range(4):
            if (gi, bb) not in used:
                nl.op(T_XOR, ("gate", gi, bb), ("const", 0))
    return nl, out


def finalize(nl, out):
    """group the lookups by table, pad to rows of LOOKUPS_PER_ROW, encode references"""
    order = sorted(range(len(nl.ops)), key=lambda j: nl.ops[j][0])
    new_index, placed = {}, []
    for j in order:
        t = nl.ops[j][0]
        if placed and placed[-1][0] != t:
            while len(placed) % LOOKUPS_PER_ROW:
                placed.append((placed[-1][0], ("const", 0), ("const", 0)))
        new_index[j] = len(placed)
        placed.append(nl.ops[j])
    while len(placed) % LOOKUPS_PER_ROW:
        placed.append((placed[-1][0], ("const", 0), ("const", 0)))

    def enc(r):
        kind = r[0]
        if kind == "op":
            return new_index[r[1]]
        if kind == "gate":
            return R_GATE + 4 * r[1] + r[2]
        if kind == "hdr":
            return R_HDR + r[1]
        if kind == "prev":
            return R_PREV + r[1]
        if kind == "free":
            return R_FREE + r[1]
        return R_CONST + r[1]

    ops = [(t, enc(a), enc(b)) for t, a, b in placed]
    # k_sc_hist (multiplicities) reads a table's lookups as ONE run of rows per cycle and skips the padding by position:
    # tables ascending, every table starts a row, padding (0, 0) only after a table's last real lookup
    for j in range(1, len(ops)):
        assert ops[j][0] >= ops[j - 1][0]
        if ops[j][0] != ops[j - 1][0]:
            assert j % LOOKUPS_PER_ROW == 0
        elif (ops[j - 1][1], ops[j - 1][2]) == (R_CONST, R_CONST):
            assert (ops[j][1], ops[j][2]) == (R_CONST, R_CONST), "a real lookup after padding"
    gates = [([[enc(r) for r in w] for w in operands], k) for operands, k in nl.gates]
    return ops, gates, [enc(r) for r in out]


def evaluate(ops, gates, out, order, prev, block, reset, idle):
    hdrv = {HDR_RESET: reset, HDR_IDLE: idle, HDR_MASK_R: 255 * reset, HDR_MASK_NR: 255 - 255 * reset, HDR_MASK_I: 255 * idle,
            HDR_MASK_A: 255 - 255 * idle}
    ov, gv = [None] * len(ops), [None] * len(gates)

    def get(r):
        if r < R_GATE:
            return ov[r]
        if r < R_HDR:
            return gv[(r - R_GATE) // 4][(r - R_GATE) % 4]
        if r < R_PREV:
            return hdrv[r - R_HDR]
        if r < R_FREE:
            return prev[r - R_PREV]
        if r < R_CONST:
            return block[r - R_FREE]
        return r - R_CONST

    for it in order:
        if it < R_GATE:
            t, a, b = ops[it]
            ov[it] = table_fn(t, get(a), get(b))
        else:
            operands, k = gates[it - R_GATE]
            total = k + sum(sum(get(w[b]) << (8 * b) for b in range(4)) for w in operands)
            gv[it - R_GATE] = [(total >> (8 * b)) & 0xFF for b in range(4)] + [total >> 32]
            assert total >> 32 <= len(operands)
    return [get(r) for r in out]


def topo_order(ops, gates):
    """evaluation order: items (lookup j or R_GATE + g) by dependency level; returns (order, level starts)"""
    lvl_op, lvl_g = [None] * len(ops), [None] * len(gates)

    def ref_level(r):
        if r < R_GATE:
            return level_op(r)
        if r < R_HDR:
            return level_gate((r - R_GATE) // 4)
        return 0

    def level_op(j):
        if lvl_op[j] is None:
            lvl_op[j] = 1 + max(ref_level(ops[j][1]), ref_level(ops[j][2]))
        return lvl_op[j]

    def level_gate(g):
        if lvl_g[g] is None:
            lvl_g[g] = 1 + max(ref_level(r) for w in gates[g][0] for r in w)
        return lvl_g[g]

    import sys
    sys.setrecursionlimit(100000)
    items = [(level_op(j), j) for j in range(len(ops))] + [(level_gate(g), R_GATE + g) for g in range(len(gates))]
    items.sort()
    order = [it for _, it in items]
    n_levels = items[-1][0]
    starts, cur = [], 0
    for idx, (lv, _) in enumerate(items):
        while cur < lv:
            starts.append(idx)
            cur += 1
    starts.append(len(items))
    return order, starts[0:1] * 0 + starts, n_levels


def sha_compress(state, block):
    w = list(struct.unpack(">16I", bytes(block)))
    rotr = lambda x, n: ((x >> n) | (x << (32 - n))) & 0xFFFFFFFF  # noqa: E731
    for i in range(16, 64):
        s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3)
        s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10)
        w.append((w[i - 16] + s0 + w[i - 7] + s1) & 0xFFFFFFFF)
    a, b, c, d, e, f, g, h = state
    for i in range(64):
        S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)
        ch = (e & f) ^ (~e & g & 0xFFFFFFFF)
        t1 = (h + S1 + ch + K[i] + w[i]) & 0xFFFFFFFF
        S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)
        maj = (a & b) ^ (a & c) ^ (b & c)
        t2 = (S0 + maj) & 0xFFFFFFFF
        a, b, c, d, e, f, g, h = (t1 + t2) & 0xFFFFFFFF, a, b, c, (d + t1) & 0xFFFFFFFF, e, f, g
    return [(x + y) & 0xFFFFFFFF for x, y in zip(state, [a, b, c, d, e, f, g, h])]


def emit(prefix, lookups_per_row, header, title):
    global LOOKUPS_PER_ROW
    LOOKUPS_PER_ROW = lookups_per_row
    nl, out = build_cycle()
    ops, gates, out = finalize(nl, out)
    order, starts, n_levels = topo_order(ops, gates)
    to_bytes = lambda ws: [(w >> (8 * b)) & 0xFF for w in ws for b in range(4)]  # noqa: E731
    rng = random.Random(1)
    # one compression against the plain function, then a two-block message against hashlib
    st = [rng.getrandbits(32) for _ in range(8)]
    blk = [rng.randrange(256) for _ in range(64)]
    assert evaluate(ops, gates, out, order, to_bytes(st), blk, 0, 0) == to_bytes(sha_compress(st, blk))
    assert evaluate(ops, gates, out, order, to_bytes(st), blk, 0, 1) == to_bytes(st)            # idle carries the state
    msg = bytes(rng.randrange(256) for _ in range(100))
    padded = msg + b"\x80" + bytes((55 - len(msg)) % 64) + struct.pack(">Q", 8 * len(msg))
    state = to_bytes([0] * 8)
    for i in range(0, len(padded), 64):
        state = evaluate(ops, gates, out, order, state, list(padded[i:i + 64]), 1 if i == 0 else 0, 0)
    digest = b"".join(struct.pack(">I", sum(state[4 * j + b] << (8 * b) for b in range(4))) for j in range(8))
    assert digest == hashlib.sha256(msg).digest(), "netlist != SHA-256"
    lookup_rows = len(ops) // LOOKUPS_PER_ROW
    gate_rows = -(-len(gates) // 2)
    rows_per_cycle = 1 + max(lookup_rows, gate_rows)
    max_operands = max(len(o) for o, _ in gates)
    o = []

    def w(line):
        o.append(line.replace("SC_", prefix + "_").replace("sc_op", prefix.lower() + "_op").replace("sc_gate", prefix.lower() + "_gate"))

    for t in title:
        o.append(t)
    guard = f"ZKW_{header.upper().replace('.', '_').replace('ZKW_', '')}"
    o.append(f"#ifndef {guard}\n#define {guard}\n#include <stdint.h>")
    w(f"#define SC_G {G}\n#define SC_LOOKUPS_PER_ROW {LOOKUPS_PER_ROW}\n#define SC_LOOKUP_COL0 {G}\n#define SC_NUM_TABLES {N_TABLES}")
    w(f"#define SC_MULT_COL0 {G + 3 * LOOKUPS_PER_ROW}\n#define SC_COLS {G + 3 * LOOKUPS_PER_ROW + N_TABLES}\n#define SC_TABLE_ROWS 65536")
    w("#define SC_T_XOR 1\n#define SC_T_ANDN 2\n#define SC_T_ROT(s) (2 + (s))\n#define SC_T_AND 10")
    w(f"#define SC_NUM_OPS {len(ops)}      /* lookups of a cycle, grouped by table, padded to rows; lookup j: row 1 + j / {LOOKUPS_PER_ROW}, slot j % {LOOKUPS_PER_ROW} */")
    w(f"#define SC_NUM_GATES {len(gates)}   /* ADD gates; gate g: row 1 + g / 2, columns (g % 2) * SC_GATE_COLS .. */")
    w(f"#define SC_GATE_COLS {GATE_COLS}   /* operand o byte b at 4 * o + b, out byte b at SC_GATE_OUT + b, carry at SC_GATE_CARRY */")
    w(f"#define SC_GATE_MAX_OPERANDS {max_operands}\n#define SC_GATE_OUT {4 * max_operands}\n#define SC_GATE_CARRY {4 * max_operands + 4}")
    w(f"#define SC_ROWS_PER_CYCLE {rows_per_cycle}  /* header row + max(lookup rows {lookup_rows}, gate rows {gate_rows}); cycle-major */")
    w("#define SC_HDR_RESET 0\n#define SC_HDR_IDLE 1\n#define SC_HDR_MASK_R 2\n#define SC_HDR_MASK_NR 3\n#define SC_HDR_MASK_I 4\n#define SC_HDR_MASK_A 5\n#define SC_HDR_FIELDS 6")
    w("/* boundary rows after the last cycle: BND_IN (columns 0..31: the chaining state before cycle 0, byte k = byte k % 4 of")
    w("   word k / 4, least significant first), BND_OUT (after the last cycle), PI (columns 0..3) */")
    w("#define SC_BOUNDARY_ROW(capacity) ((uint64_t)(capacity) * SC_ROWS_PER_CYCLE)")
    w("#define SC_MIN_ROWS(capacity) (SC_BOUNDARY_ROW(capacity) + 3 > SC_TABLE_ROWS ? SC_BOUNDARY_ROW(capacity) + 3 : SC_TABLE_ROWS)")
    w(f"#define SC_REF_GATE 0x{R_GATE:X}\n#define SC_REF_HDR 0x{R_HDR:X}\n#define SC_REF_PREV 0x{R_PREV:X}\n#define SC_REF_FREE 0x{R_FREE:X}\n#define SC_REF_CONST 0x{R_CONST:X}")
    w("typedef struct { uint16_t table, a, b; } sc_op;")
    w("typedef struct { uint32_t n_operands, constant; uint16_t in[SC_GATE_MAX_OPERANDS][4]; } sc_gate;")
    w("#define SC_OPS_INIT { \\")
    for t, a, b in ops:
        o.append(f"  {{{t}, {a}, {b}}}, \\")
    o.append("}")
    w("#define SC_GATES_INIT { \\")
    for operands, k in gates:
        rows = [("{" + ", ".join(str(r) for r in wd) + "}") for wd in operands] + ["{0, 0, 0, 0}"] * (max_operands - len(operands))
        o.append(f"  {{{len(operands)}, 0x{k:08X}u, {{{', '.join(rows)}}}}}, \\")
    o.append("}")
    w("/* byte k of the cycle's output state = this reference (a lookup output) */")
    w("#define SC_OUT_INIT {" + ", ".join(str(r) for r in out) + "}")
    w(f"#define SC_NUM_LEVELS {n_levels}")
    w("/* items (lookup j, or SC_REF_GATE + g) in dependency order; level l = [SC_LEVEL_START[l], SC_LEVEL_START[l + 1]) */")
    w("#define SC_EVAL_ORDER_INIT {" + ", ".join(str(it) for it in order) + "}")
    w("#define SC_LEVEL_START_INIT {" + ", ".join(str(v) for v in starts) + "}")
    o.append("#endif")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "include", header)
    open(path, "w").write("\n".join(o) + "\n")
    print(f"{prefix}: {len(ops)} lookups ({lookup_rows} rows of {LOOKUPS_PER_ROW}), {len(gates)} gates ({gate_rows} rows), {n_levels} levels, "
          f"{rows_per_cycle} rows per cycle -> capacity up to {((1 << 20) - 3) // rows_per_cycle} in 2^20 rows; {path}")


def main():
    emit("SC", 14, "zkw_sha256_circuit_spec.h",
         ("/* GENERATED by tools/gen_sha256_circuit.py — do not edit. Layout contract of the Sha256RoundFunction trace emitted by",
          " * zkw_sha256_round_synthesize (\"zkw trace v3\": one netlist per cycle, byte lookups + 32-bit ADD gates). See the generator. */"))
    # CodeDecommitter (type 3): the same compression at 2845 cycles per trace (geometry_config.rs) = 368 rows per cycle: 18
    # lookups per row (54 lookup columns; the reference wrapper has 44, base_layer/code_decommitter.rs:28-39) -> 365 rows
    emit("DC", 18, "zkw_code_decommitter_circuit_spec.h",
         ("/* GENERATED by tools/gen_sha256_circuit.py — do not edit. Layout contract of the CodeDecommitter trace emitted by",
          " * zkw_code_decommitter_synthesize (\"zkw trace v3\": the SHA-256 netlist of zkw_sha256_circuit_spec.h at 18 lookups per row). */"))


if __name__ == "__main__":
    main()
