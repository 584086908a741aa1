#!/usr/bin/env python3
"""Generates include/zkw_keccak_circuit_spec.h — the declarative layout of the Keccak256RoundFunction trace that libzkw
emits ("zkw trace v3", circuit type 5): a NETLIST of byte operations, every one a width-3 lookup.

Geometry of the reference wrapper (circuit_definitions/.../base_layer/keccak256_round_function.rs:28-39,52-142): 86 copy
columns, width-3 lookups x 14 repetitions with ONE table id per row (share_table_id), tables Xor8 / And8 /
ByteSplit<1..4>, 2^20 rows, capacity 293 (geometry_config.rs). The circuit body lives in the absent crate
era-zkevm_circuits, so gate placement — and here also the choice of tables — is OUR design ("parity unpinned" at the
trace-layout level).

Tables (id, 65 536 rows each, row index a * 256 + b, one multiplicity column per table):
    1 XOR8     (a, b, a ^ b)
    2 ANDN8    (a, b, ~a & b)                                   chi:  a ^ (~b & c)
    3..9 ROT<s>, s = 1..7: (a, b, ((a << s) & 0xff) | (b >> (8 - s)))   byte k of a lane rotated left by 8q + s is
                                                                ROT<s>(byte k - q, byte k - q - 1): rho without bit splitting
Statement, per cycle (= one Keccak-f[1600] call of the precompile, keccak256_round_function.rs:214-404):
    in  = reset ? 0 : prev   (prev = state after the previous cycle)   200 ANDN lookups against mask_r = 255 * reset
    abs = in ^ block   on the first 136 bytes                          136 XOR lookups; block = the padded input of the round
    f   = Keccak-f[1600](abs)                                          24 rounds x KC_OPS_PER_ROUND lookups
    out = idle ? prev : f                                              3 x 200 lookups: ANDN(mask_i, f) ^ ANDN(255 - mask_i, prev)
Cycles beyond an instance's rounds are idle: they carry the state to the BND_OUT rows. The block bytes, the reset and the
idle bit come from the witness builder (in the reference's circuit they are tied to the memory
queue and the request queue by Poseidon2 queue gadgets; those ties are not part of this trace yet: DESIGN.md 3.17).

A round is a list of operations {table, a, b}; operand references: 0..199 = byte of the round's input state,
200 + j = output of operation j of the same round, KC_REF_ZERO = the constant 0, KC_REF_RC0 + k = byte k of the round
constant. Operations are grouped by table and padded to whole rows (14 per row) with (0, 0) operands, which every table
maps to 0. The generator checks the netlist against a plain Keccak-f before writing the header.
"""
import os
import random

LOOKUPS_PER_ROW = 14
G = 86
T_NONE, T_XOR, T_ANDN = 0, 1, 2
T_ROT = lambda s: 2 + s  # noqa: E731   3..9
N_TABLES = 9
REF_OP0 = 200
REF_ZERO = 0xFFFF
REF_RC0 = 0xFF00

RC = [0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000, 0x000000000000808B, 0x0000000080000001,
      0x8000000080008081, 0x8000000000008009, 0x000000000000008A, 0x0000000000000088, 0x0000000080008009, 0x000000008000000A,
      0x000000008000808B, 0x800000000000008B, 0x8000000000008089, 0x8000000000008003, 0x8000000000008002, 0x8000000000000080,
      0x000000000000800A, 0x800000008000000A, 0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008]
ROT = [0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14]  # [x + 5y]


def table_fn(t, a, b):
    if t == T_XOR:
        return a ^ b
    if t == T_ANDN:
        return (~a & 0xFF) & b
    s = t - 2
    return ((a << s) & 0xFF) | (b >> (8 - s))


def build_round():
    """one round as (ops, out): ops[j] = (table, a_ref, b_ref); out[idx] = reference that holds byte idx of the next state"""
    ops = []

    def op(t, a, b):
        ops.append((t, a, b))
        return REF_OP0 + len(ops) - 1

    S = lambda x, y, k: 8 * (x + 5 * y) + k  # noqa: E731
    C = {}
    for x in range(5):
        for k in range(8):
            t = op(T_XOR, S(x, 0, k), S(x, 1, k))
            for y in (2, 3, 4):
                t = op(T_XOR, t, S(x, y, k))
            C[x, k] = t
    R1 = {(x, k): op(T_ROT(1), C[x, k], C[x, (k - 1) % 8]) for x in range(5) for k in range(8)}
    D = {(x, k): op(T_XOR, C[(x - 1) % 5, k], R1[(x + 1) % 5, k]) for x in range(5) for k in range(8)}
    A = {(x, y, k): op(T_XOR, S(x, y, k), D[x, k]) for y in range(5) for x in range(5) for k in range(8)}
    B = {}
    for x in range(5):
        for y in range(5):
            q, s = divmod(ROT[x + 5 * y], 8)
            for k in range(8):
                src = A[x, y, (k - q) % 8]
                B[y, (2 * x + 3 * y) % 5, k] = src if s == 0 else op(T_ROT(s), src, A[x, y, (k - q - 1) % 8])
    E = {}
    for y in range(5):
        for x in range(5):
            for k in range(8):
                n = op(T_ANDN, B[(x + 1) % 5, y, k], B[(x + 2) % 5, y, k])
                E[x, y, k] = op(T_XOR, B[x, y, k], n)
    for k in range(8):
        E[0, 0, k] = op(T_XOR, E[0, 0, k], REF_RC0 + k)
    out = [E[idx // 8 % 5, idx // 40, idx % 8] for idx in range(200)]
    return ops, out


def group_and_pad(ops, out):
    """stable sort by table, pad every group to a multiple of LOOKUPS_PER_ROW, renumber the references"""
    order = sorted(range(len(ops)), key=lambda j: ops[j][0])
    new_index, placed = {}, []
    for j in order:
        t = ops[j][0]
        if placed and placed[-1][0] != t:
            while len(placed) % LOOKUPS_PER_ROW:
                placed.append((placed[-1][0], REF_ZERO, REF_ZERO))
        new_index[j] = len(placed)
        placed.append(ops[j])
    while len(placed) % LOOKUPS_PER_ROW:
        placed.append((placed[-1][0], REF_ZERO, REF_ZERO))
    fix = lambda r: REF_OP0 + new_index[r - REF_OP0] if REF_OP0 <= r < REF_RC0 else r  # noqa: E731
    return [(t, fix(a), fix(b)) for t, a, b in placed], [fix(r) for r in out]


def eval_round(ops, out, state, rnd):
    vals = []

    def get(r):
        if r == REF_ZERO:
            return 0
        if r >= REF_RC0:
            return (RC[rnd] >> (8 * (r - REF_RC0))) & 0xFF
        return state[r] if r < REF_OP0 else vals[r - REF_OP0]

    # references may point forward after grouping: evaluate until everything is known
    vals = [None] * len(ops)
    pending = list(range(len(ops)))
    while pending:
        rest = []
        for j in pending:
            t, a, b = ops[j]
            try:
                va, vb = get(a), get(b)
            except TypeError:
                va = None
            if va is None or vb is None:
                rest.append(j)
            else:
                vals[j] = table_fn(t, va, vb)
        assert len(rest) < len(pending), "cyclic netlist"
        pending = rest
    return [get(r) for r in out]


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


def main():
    ops, out = group_and_pad(*build_round())
    assert len(ops) % LOOKUPS_PER_ROW == 0
    rng = random.Random(1)
    for _ in range(3):
        st = [rng.randrange(256) for _ in range(200)]
        cur = st
        for rnd in range(24):
            cur = eval_round(ops, out, cur, rnd)
        assert cur == keccak_f(st), "netlist != Keccak-f[1600]"
    rows_per_round = len(ops) // LOOKUPS_PER_ROW
    mask_rows = -(-200 // LOOKUPS_PER_ROW)
    abs_rows = -(-136 // LOOKUPS_PER_ROW)
    rows_per_cycle = 1 + mask_rows + abs_rows + 24 * rows_per_round + 3 * mask_rows
    o = []
    w = o.append
    w("/* GENERATED by tools/gen_keccak_circuit.py — do not edit. Layout contract of the Keccak256RoundFunction trace emitted")
    w(" * by zkw_keccak_round_synthesize (\"zkw trace v3\": a netlist of width-3 byte lookups). See the generator's docstring. */")
    w("#ifndef ZKW_KECCAK_CIRCUIT_SPEC_H\n#define ZKW_KECCAK_CIRCUIT_SPEC_H\n#include <stdint.h>")
    w(f"#define KC_G {G}                 /* general-purpose columns 0..{G - 1} (used by the cycle header and the boundary rows) */")
    w(f"#define KC_LOOKUPS_PER_ROW {LOOKUPS_PER_ROW}   /* width-3 lookups per row, one table id per row */")
    w(f"#define KC_LOOKUP_COL0 {G}        /* lookup j of a row: columns KC_LOOKUP_COL0 + 3j .. + 2 = (a, b, c) */")
    w(f"#define KC_NUM_TABLES {N_TABLES}         /* ids 1..{N_TABLES}: XOR8, ANDN8, ROT<1..7>; 65 536 rows each, row a * 256 + b */")
    w(f"#define KC_MULT_COL0 {G + 3 * LOOKUPS_PER_ROW}       /* multiplicity column of table t: KC_MULT_COL0 + t - 1 */")
    w(f"#define KC_COLS {G + 3 * LOOKUPS_PER_ROW + N_TABLES}")
    w("#define KC_TABLE_ROWS 65536")
    w("#define KC_T_XOR 1\n#define KC_T_ANDN 2\n#define KC_T_ROT(s) (2 + (s)) /* s = 1..7 */")
    w(f"#define KC_OPS_PER_ROUND {len(ops)}\n#define KC_ROWS_PER_ROUND {rows_per_round}")
    w(f"#define KC_MASK_ROWS {mask_rows}        /* 200 ANDN lookups (mask, previous state byte) -> input state byte, padded */")
    w(f"#define KC_ABSORB_ROWS {abs_rows}      /* 136 XOR lookups (input state byte, block byte) -> absorbed byte, padded */")
    w(f"#define KC_ROWS_PER_CYCLE {rows_per_cycle}  /* header + mask + absorb + 24 rounds + 3 x select; cycle-major: cycle i starts at row i * KC_ROWS_PER_CYCLE */")
    w("#define KC_ROW_MASK0 1\n#define KC_ROW_ABSORB0 (1 + KC_MASK_ROWS)\n#define KC_ROW_ROUND0 (1 + KC_MASK_ROWS + KC_ABSORB_ROWS)")
    w("/* out = idle ? prev : f: T rows ANDN(mask_i, f_k), U rows ANDN(mask_a, prev_k), O rows XOR(t_k, u_k) = the cycle's output state */")
    w("#define KC_ROW_SEL_T0 (KC_ROW_ROUND0 + 24 * KC_ROWS_PER_ROUND)\n#define KC_ROW_SEL_U0 (KC_ROW_SEL_T0 + KC_MASK_ROWS)\n#define KC_ROW_SEL_O0 (KC_ROW_SEL_U0 + KC_MASK_ROWS)")
    w("/* header row of a cycle (general columns): reset and idle bits (boolean), mask_r = 255 * reset, mask_i = 255 * idle,")
    w("   mask_a = 255 - mask_i */")
    w("#define KC_HDR_RESET 0\n#define KC_HDR_IDLE 1\n#define KC_HDR_MASK_R 2\n#define KC_HDR_MASK_I 3\n#define KC_HDR_MASK_A 4")
    w("/* boundary rows after the last cycle: BND_IN (columns 0..199: the state before cycle 0 = hidden_fsm_input),")
    w("   BND_OUT (the state after the last cycle = hidden_fsm_output) — 200 bytes over 3 rows of 86 columns each — then PI */")
    w("#define KC_BND_ROWS_PER_STATE 3\n#define KC_BOUNDARY_ROW(capacity) ((uint64_t)(capacity) * KC_ROWS_PER_CYCLE)")
    w("#define KC_MIN_ROWS(capacity) (KC_BOUNDARY_ROW(capacity) + 2 * KC_BND_ROWS_PER_STATE + 1 > KC_TABLE_ROWS ? KC_BOUNDARY_ROW(capacity) + 2 * KC_BND_ROWS_PER_STATE + 1 : KC_TABLE_ROWS)")
    w("/* operand references of a round's operations */")
    w(f"#define KC_REF_OP0 {REF_OP0}      /* KC_REF_OP0 + j = output (cell c) of operation j of the same round */")
    w(f"#define KC_REF_RC0 0x{REF_RC0:X}   /* + k = byte k of the round constant */")
    w(f"#define KC_REF_ZERO 0x{REF_ZERO:X}  /* the constant 0 (padding operations) */")
    w("typedef struct { uint16_t table, a, b; } kc_op;")
    w("#define KC_ROUND_OPS_INIT { \\")
    for t, a, b in ops:
        w(f"  {{{t}, {a}, {b}}}, \\")
    w("}")
    # evaluation order: operations by dependency level (references may point forward after the grouping by table)
    level = [None] * len(ops)

    def lvl(r):
        return 0 if r < REF_OP0 or r >= REF_RC0 else level_of(r - REF_OP0)

    def level_of(j):
        if level[j] is None:
            level[j] = 1 + max(lvl(ops[j][1]), lvl(ops[j][2]))
        return level[j]

    for j in range(len(ops)):
        level_of(j)
    order = sorted(range(len(ops)), key=lambda j: level[j])
    n_levels = max(level)
    starts = [next(i for i, j in enumerate(order) if level[j] == l) for l in range(1, n_levels + 1)] + [len(ops)]
    w(f"#define KC_ROUND_NUM_LEVELS {n_levels}")
    w("/* operations in dependency order; level l = entries [KC_ROUND_LEVEL_START[l], KC_ROUND_LEVEL_START[l + 1]) depend only on earlier levels */")
    w("#define KC_ROUND_EVAL_ORDER_INIT {" + ", ".join(str(j) for j in order) + "}")
    w("#define KC_ROUND_LEVEL_START_INIT {" + ", ".join(str(v) for v in starts) + "}")
    w("/* byte idx of the next state = this reference (an operation output of the round) */")
    w("#define KC_ROUND_OUT_INIT {" + ", ".join(str(r) for r in out) + "}")
    w("#define KC_RC_INIT {" + ", ".join(f"0x{v:016X}ULL" for v in RC) + "}")
    w("#endif")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "include", "zkw_keccak_circuit_spec.h")
    open(path, "w").write("\n".join(o) + "\n")
    by_table = {}
    for t, _, _ in ops:
        by_table[t] = by_table.get(t, 0) + 1
    print(f"{len(ops)} operations per round ({by_table}), {rows_per_round} rows per round, {rows_per_cycle} rows per cycle -> "
          f"capacity up to {((1 << 20) - 7) // rows_per_cycle} in 2^20 rows; {path}")


if __name__ == "__main__":
    main()
