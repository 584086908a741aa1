#!/usr/bin/env python3
"""Generates include/zkw_ecrecover_circuit_spec.h: the ECRecover base-layer circuit (type 7) on the reference wrapper's geometry and
table set (circuit_definitions/src/circuit_definitions/base_layer/ecrecover.rs:30-41: 80 copy columns, width-3 lookups x 16 per row;
:138-176: Xor8, And8, 8 x 32 FixedBaseMulTable<i, C>, ByteSplit<1..4> = 197 632 table rows = `total_tables_len` of vk_7.json).

The circuit body (`ecrecover_function_entry_point`) lives in the absent crate era-zkevm_circuits, so — like every other circuit of
this library — gate placement is this library's own; what is the reference's: geometry, lookup width / repetitions, one table per row,
the table set with its row counts and contents, one multiplicity column, the capacity unit (one request per cycle, 7 per instance),
4 reads + 2 writes per request (src/witness/individual_circuits/ecrecover.rs:143-178), 16-bit limbs for the non-native fields.

A trace of this circuit has three parts:
  1. a byte netlist ("zkw trace v4", tools/netlist.py; prefix EK): ONE Keccak-f[1600] per cycle over the 64-byte public key (the
     step types of tools/gen_keccak_circuit.py; the absorb step takes the key bytes as FREE elements and the padding as constants,
     the select step also masks the 20 address bytes with 255 * ok);
  2. the queue section (include/zkw_netlist_queue.h): pop of the call, 4 reads, 2 writes as Poseidon2 rows;
  3. the EC SECTION (this file; prefix EC): secp256k1 arithmetic over field-element-valued rows, `EC_ROWS_PER_CYCLE` rows per cycle,
     cycle-major, below the queue section. A cycle is a sequence of SEGMENTS (PRE, 256 x DAA, 32 x FIX, POST), each an instance of a
     segment type = a list of ITEMS over general-purpose cells and lookup slots:
       LIN     sum coef_i * cell_i + const == sum 2^shift_j * new_j      (new cells = digits of the known part: byte / bit / limb
               decompositions, lazy limb-wise sums and differences, borrow chains)
       SEL     o = b ? x : y                  (b * (x - y) + y - o == 0)
       FMA     a * b + c == d
       MUL     a * b + 8 m == q * m + r over 16-bit limbs, m = the secp256k1 base (P) or scalar (N) modulus: ONE row
               [a16 | b16 | q16 | r16 | carry15]; position k (two limbs = 32 bits): D_k + c_{k-1} == 2^32 c_k with carries stored + 2^31
       LOOKUP  XOR8 (a, b, a ^ b) = the range check of two bytes; FIXEDBASE<i, C> (byte, word i of x, word i of y of byte * 2^(8C) * G)
       HINT    a witness the items above constrain (quotients lambda = dy / dx, products, the square root, zero-test inverses, >=)
     Values live on a per-cycle TAPE; a cell is a reference: tape value of this segment / of the previous segment's state / of the
     cycle's globals (fixed by PRE), a constant, a limb of a 256-bit constant, an input byte. Every tape value has a HOME cell (its
     first occurrence); every other occurrence is a copy constraint.

Statement of a cycle (inputs: the 4 x 32 value bytes of the reads h, v, r, s; outputs: ok, the 20 address bytes):
  e_r = (r == 0) | (r >= n), e_s likewise, vbit = v[0] boolean; on e_r | e_s the inputs are replaced by a fixed valid signature;
  x = r*, t = x^3 + 7, y with y^2 == t (y < p, parity vbit) or e_nr and y^2 == -t (p = 3 mod 4: -1 is a non-residue, so a root of
  -t proves that t has none); on e_nr the point is replaced by a fixed valid one; u2 = s* / r*, u1 = h* / r* (mod n);
  acc = O; 256 x (acc = 2 acc; acc += bit ? R : 0) over the bits of u2; 32 x (acc -= byte * 2^(8C) * G) over the bytes of u1 by table;
  Q = acc - 2^256 O, normalised (< p); ok = !(e_r | e_s | e_nr); the netlist hashes Q.x || Q.y and masks. Affine (incomplete)
  addition with an offset point O: inputs that drive the accumulator into x1 == x2 have no witness (the builder rejects them).
"""
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import netlist as nl  # noqa: E402
import gen_keccak_circuit as gk  # noqa: E402
from era_zkevm_test_harness_amd import secp256k1 as ec  # noqa: E402

G_COLS, LW, LR = 80, 3, 16
PGL = 2**64 - 2**32 + 1
P, N = ec.P, ec.N
KMUL = 8  # a * b + KMUL * m == q * m + r keeps q >= 0 for every lazy r < 2^259

# ---- references -------------------------------------------------------------------------------------------------------------
K_TAPE, K_PREV, K_GLOB, K_GLOBJ, K_CONST, K_BIG, K_IN, K_NONE = 0, 1, 2, 3, 4, 5, 6, 15
NONE = 0xFFFFFFFF


class Ref:
    __slots__ = ("kind", "a", "b")

    def __init__(self, kind, a, b=0):
        self.kind, self.a, self.b = kind, a, b

    def enc(self):
        if self.kind == K_GLOBJ:
            return (K_GLOBJ << 28) | ((self.b & 0xFF) << 16) | self.a
        if self.kind == K_BIG:
            return (K_BIG << 28) | (self.a << 4) | self.b
        return (self.kind << 28) | self.a

    def key(self):
        return (self.kind, self.a, self.b)


def const(v):
    assert 0 <= v < (1 << 28)
    return Ref(K_CONST, v)


def big(idx, limb):
    return Ref(K_BIG, idx, limb)


def limbs16(v, n=16):
    return [(v >> (16 * i)) & 0xFFFF for i in range(n)]


def offset_limbs(m):
    """limbs o_k of 4 m with o_k >= 2^16 for k < 15 (so that x_k - y_k + o_k > 0 limb by limb for 16-bit x_k, y_k) and a top limb
    that holds the rest: sum o_k 2^(16 k) == 4 m"""
    v = 4 * m
    L = [(v >> (16 * i)) & 0xFFFF for i in range(15)] + [v >> 240]
    o = [L[0] + 0x10000] + [L[k] + 0x10000 - 1 for k in range(1, 15)] + [L[15] - 1]
    assert sum(x << (16 * k) for k, x in enumerate(o)) == v and all(x >= 0x10000 for x in o)
    return o


# the 256-bit constants cells may reference (limb by limb)
BIGS = {}
BIG_LIST = []


def add_big(name, limbs):
    BIGS[name] = len(BIG_LIST)
    BIG_LIST.append(list(limbs))


def bigv(name):
    return [big(BIGS[name], k) for k in range(16)]


def hash_point(tag):
    """a curve point nobody knows a discrete-log relation of: try-and-increment on keccak256(tag || counter)"""
    ctr = 0
    while True:
        x = int.from_bytes(ec.keccak256(tag + ctr.to_bytes(4, "big")), "big") % P
        pt = ec.lift_x(x, 0)
        if pt is not None:
            return pt
        ctr += 1


OFFSET_POINT = hash_point(b"zkw ecrecover accumulator offset")
NEG_OFFSET_END = ec.neg(ec.mul(pow(2, 256, N), OFFSET_POINT))
# the substitute for a cycle whose r / s are out of range (and for idle cycles, whose inputs are zeros): a valid signature
SUB_H = int.from_bytes(ec.keccak256(b"zkw ecrecover substitute message"), "big")
_sv, SUB_R, SUB_S = ec.sign(SUB_H, 0x7A6B77, 0x1F2E3D4C5B6A79880796A5B4C3D2E1F0)
SUB_POINT = ec.lift_x(SUB_R, _sv)

add_big("P", limbs16(P))
add_big("N", limbs16(N))
add_big("OFFP", offset_limbs(P))
add_big("OFFN", offset_limbs(N))
add_big("PM1", limbs16(P - 1))
add_big("NM1", limbs16(N - 1))
add_big("ONE", limbs16(1))
add_big("OX", limbs16(OFFSET_POINT[0]))
add_big("OY", limbs16(OFFSET_POINT[1]))
add_big("EX", limbs16(NEG_OFFSET_END[0]))
add_big("EY", limbs16(NEG_OFFSET_END[1]))
add_big("SUBH", limbs16(SUB_H))
add_big("SUBR", limbs16(SUB_R))
add_big("SUBS", limbs16(SUB_S))
add_big("SUBX", limbs16(SUB_POINT[0]))
add_big("SUBY", limbs16(SUB_POINT[1]))

# ---- items ------------------------------------------------------------------------------------------------------------------
I_LIN, I_SEL, I_FMA, I_MUL, I_HINT, I_LOOKUP = 1, 2, 3, 4, 5, 6
H_MULSUB, H_DIV, H_SQRT, H_ISZERO, H_GE = 1, 2, 3, 4, 5
T_XOR8, T_AND8, T_FIXED0 = 1, 2, 3  # table ids (EK netlist spec order): FIXEDBASE<i, C> = 3 + 8 C + i, BYTESPLIT<k> = 258 + k
ROWTAB_PER_INSTANCE = 0x8000  # the row's table id grows by 8 per instance of the segment (FIX: C = instance)
MOD_P, MOD_N = 0, 1


class Seg:
    """one segment type under construction. Values are Refs; NEW values are allocated on the segment's tape."""

    def __init__(self, name):
        self.name = name
        self.n_tape = 0
        self.items = []      # dicts
        self.grow, self.gcol = 0, 0          # next free general-purpose cell
        self.xor_slots = []  # [(row, slot)] handed out for XOR8 lookups
        self.xrow, self.xslot = 0, 0
        self.row_table = {}  # row -> table id (| ROWTAB_PER_INSTANCE)
        self.fixed_rows = 0
        self.out = None
        self.globs = []      # PRE: tape indices of the cycle's globals
        self.pending_range = []

    def new(self, n=1):
        t = self.n_tape
        self.n_tape += n
        return [Ref(K_TAPE, t + i) for i in range(n)]

    def _place(self, n):
        assert n <= G_COLS
        if self.gcol + n > G_COLS:
            self.grow, self.gcol = self.grow + 1, 0
        at = (self.grow, self.gcol)
        self.gcol += n
        return at

    # -- general-purpose items
    def lin(self, known, constant=0, new=((0, 0),)):
        """known: [(ref, coef)]; new: [(shift, width)] ascending, width 0 = everything that is left (only the last). Returns NEW refs."""
        outs = self.new(len(new))
        row, col = self._place(len(known) + len(new))
        self.items.append(dict(k=I_LIN, row=row, col=col, known=list(known), const=constant, new=list(new), outs=outs))
        return outs

    def lin_assert(self, known, constant=0):
        row, col = self._place(len(known))
        self.items.append(dict(k=I_LIN, row=row, col=col, known=list(known), const=constant, new=[], outs=[]))

    def sel(self, b, x, y):
        o = self.new(1)[0]
        row, col = self._place(4)
        self.items.append(dict(k=I_SEL, row=row, col=col, b=b, x=x, y=y, out=o))
        return o

    def fma(self, a, b, c):
        d = self.new(1)[0]
        row, col = self._place(4)
        self.items.append(dict(k=I_FMA, row=row, col=col, a=a, b=b, c=c, d=d, new=True))
        return d

    def fma_assert(self, a, b, c, d):
        row, col = self._place(4)
        self.items.append(dict(k=I_FMA, row=row, col=col, a=a, b=b, c=c, d=d, new=False))

    def mulrow(self, mod, a, b, r):
        """a * b + 8 m == q * m + r; a, b, r: 16 refs each (consecutive). Returns (q[16], c[15]) — both still to be range-checked."""
        if self.gcol:
            self.grow, self.gcol = self.grow + 1, 0
        q, c = self.new(16), self.new(15)
        self.items.append(dict(k=I_MUL, row=self.grow, col=0, mod=mod, a=a, b=b, r=r, q=q, c=c))
        self.grow += 1
        return q, c

    def hint(self, kind, n_out, **args):
        outs = self.new(n_out)
        self.items.append(dict(k=I_HINT, row=0, col=0, hint=kind, outs=outs, **args))
        return outs

    # -- lookups
    def _xor_slot(self):
        while self.row_table.get(self.xrow, T_XOR8) != T_XOR8:
            self.xrow, self.xslot = self.xrow + 1, 0
        self.row_table[self.xrow] = T_XOR8
        at = (self.xrow, self.xslot)
        self.xslot += 1
        if self.xslot == LR:
            self.xrow, self.xslot = self.xrow + 1, 0
        return at

    def xor8(self, a, b):
        o = self.new(1)[0]
        row, slot = self._xor_slot()
        self.items.append(dict(k=I_LOOKUP, row=row, col=slot, table=T_XOR8, ins=[a, b], outs=[o]))
        return o

    def fixed(self, i, byte, row):
        """FIXEDBASE<i, C = instance> keyed by `byte` on lookup row `row` (one table per row): (x word i, y word i)"""
        assert row not in self.row_table
        self.row_table[row] = (T_FIXED0 + i) | ROWTAB_PER_INSTANCE
        outs = self.new(2)
        self.items.append(dict(k=I_LOOKUP, row=row, col=0, table=(T_FIXED0 + i) | ROWTAB_PER_INSTANCE, ins=[byte], outs=outs))
        return outs

    # -- composites
    def range_bytes(self, bs):
        """queue bytes for the XOR8 range check (two per lookup); flush_range() pairs what is pending"""
        self.pending_range.extend(bs)
        while len(self.pending_range) >= 2:
            a, b = self.pending_range[:2]
            del self.pending_range[:2]
            self.xor8(a, b)

    def flush_range(self):
        if self.pending_range:
            self.xor8(self.pending_range.pop(), const(0))

    def bytes_of(self, v, n):
        """v == sum b_i 2^(8 i), the n bytes range-checked"""
        bs = self.lin([(v, 1)], 0, [(8 * i, 8) for i in range(n - 1)] + [(8 * (n - 1), 0)])
        self.range_bytes(bs)
        return bs

    def check_vec16(self, limbs, top_bytes=2):
        """16-bit range of every limb (the top one: top_bytes bytes); returns the bytes, little end first"""
        out = []
        for k, l in enumerate(limbs):
            out += self.bytes_of(l, top_bytes if k == 15 else 2)
        return out

    def lazy(self, terms, big_name=None):
        """limb-wise sum: terms [(vec, coef)] (+ the limbs of a constant): 16 LIN items"""
        res = []
        for k in range(16):
            known = [(v[k], c) for v, c in terms]
            if big_name is not None:
                known.append((big(BIGS[big_name], k), 1))
            res.append(self.lin(known)[0])
        return res

    def mul_checked(self, mod, a, b, r):
        q, c = self.mulrow(mod, a, b, r)
        self.check_vec16(q, 3)
        for x in c:
            self.bytes_of(x, 4)

    def nn_mulsub(self, mod, a, b, c=None, d=None):
        """r = a * b - c - d (mod m), canonical, range-checked; the MUL row is the caller's (its r side is a lazy sum)"""
        r = self.hint(H_MULSUB, 16, mod=mod, a=a, b=b, c=c, d=d)
        self.check_vec16(r)
        return r

    def nn_mul(self, mod, a, b):
        r = self.nn_mulsub(mod, a, b)
        self.mul_checked(mod, a, b, r)
        return r

    def nn_div(self, mod, num, den):
        """lam = num / den (mod m): lam * den == num"""
        lam = self.hint(H_DIV, 16, mod=mod, a=num, b=den)
        self.check_vec16(lam)
        self.mul_checked(mod, lam, den, num)
        return lam

    def add_points(self, x1, y1, x2, y2):
        """(x1, y1) + (x2, y2), x1 != x2 (mod p): three MUL rows"""
        dx = self.lazy([(x2, 1), (x1, -1)], "OFFP")
        dy = self.lazy([(y2, 1), (y1, -1)], "OFFP")
        lam = self.nn_div(MOD_P, dy, dx)
        x3 = self.nn_mulsub(MOD_P, lam, lam, x1, x2)
        self.mul_checked(MOD_P, lam, lam, self.lazy([(x3, 1), (x1, 1), (x2, 1)]))
        dxx = self.lazy([(x1, 1), (x3, -1)], "OFFP")
        y3 = self.nn_mulsub(MOD_P, lam, dxx, y1)
        self.mul_checked(MOD_P, lam, dxx, self.lazy([(y3, 1), (y1, 1)]))
        return x3, y3

    def double_point(self, x1, y1):
        sq = self.nn_mul(MOD_P, x1, x1)
        lam = self.nn_div(MOD_P, self.lazy([(sq, 3)]), self.lazy([(y1, 2)]))
        x3 = self.nn_mulsub(MOD_P, lam, lam, x1, x1)
        self.mul_checked(MOD_P, lam, lam, self.lazy([(x3, 1), (x1, 2)]))
        dxx = self.lazy([(x1, 1), (x3, -1)], "OFFP")
        y3 = self.nn_mulsub(MOD_P, lam, dxx, y1)
        self.mul_checked(MOD_P, lam, dxx, self.lazy([(y3, 1), (y1, 1)]))
        return x3, y3

    def boolean(self, b):
        self.fma_assert(b, b, const(0), b)

    def is_zero(self, s):
        """z = (s == 0) for a field element s: s * inv + z == 1, s * z == 0"""
        inv, z = self.hint(H_ISZERO, 2, a=[s])
        self.fma_assert(s, inv, z, const(1))
        self.fma_assert(s, z, const(0), const(0))
        return z

    def or2(self, a, b):
        t = self.fma(a, b, const(0))
        return self.lin([(a, 1), (b, 1), (t, -1)])[0]

    def less_than_const(self, x, cm1_name):
        """x <= C - 1 for a 16-limb x and the constant C - 1: a borrow chain, 16-bit digits as bytes, no borrow out of the top"""
        nb = None
        for k in range(16):
            known = [(big(BIGS[cm1_name], k), 1), (x[k], -1)] + ([(nb, 1)] if nb is not None else [])
            b0, b1, nb = self.lin(known, 0x10000 - (0 if nb is None else 1), [(0, 8), (8, 8), (16, 0)])
            self.range_bytes([b0, b1, nb])
        self.lin_assert([(nb, 1)], -1)

    def out_of_range(self, x, mod_name, m1_name):
        """e = (x == 0) | (x >= m) for a 16-limb x: g = (x >= m) by one borrow chain over (g ? x - m : m - 1 - x)"""
        z = self.is_zero(self.lin([(l, 1) for l in x])[0])
        g = self.hint(H_GE, 1, a=x, big=BIGS[mod_name])[0]
        self.boolean(g)
        nb = None
        for k in range(16):
            hi = self.sel(g, x[k], big(BIGS[m1_name], k))
            lo = self.sel(g, big(BIGS[mod_name], k), x[k])
            known = [(hi, 1), (lo, -1)] + ([(nb, 1)] if nb is not None else [])
            b0, b1, nb = self.lin(known, 0x10000 - (0 if nb is None else 1), [(0, 8), (8, 8), (16, 0)])
            self.range_bytes([b0, b1, nb])
        self.lin_assert([(nb, 1)], -1)
        return self.or2(z, g)

    def finish(self):
        self.flush_range()
        rows = self.grow + (1 if self.gcol else 0)
        lrows = max(list(self.row_table) + [-1]) + 1
        self.n_rows = max(rows, lrows)
        # cells: [row][col] -> Ref; homes of tape values = first occurrence in item order
        self.cells = {}
        self.home = {}

        def put(row, col, ref, is_new=False):
            assert (row, col) not in self.cells, (self.name, row, col)
            self.cells[row, col] = ref
            if ref.kind == K_TAPE and ref.a not in self.home:
                self.home[ref.a] = (row, col)
            if is_new:
                assert self.home[ref.a] == (row, col), "a NEW cell must be its value's first occurrence"

        for it in self.items:
            k, row, col = it["k"], it["row"], it["col"]
            if k == I_LIN:
                for i, (r, _c) in enumerate(it["known"]):
                    put(row, col + i, r)
                for j, o in enumerate(it["outs"]):
                    put(row, col + len(it["known"]) + j, o, True)
            elif k == I_SEL:
                for i, r in enumerate((it["b"], it["x"], it["y"])):
                    put(row, col + i, r)
                put(row, col + 3, it["out"], True)
            elif k == I_FMA:
                for i, r in enumerate((it["a"], it["b"], it["c"])):
                    put(row, col + i, r)
                put(row, col + 3, it["d"], it["new"])
            elif k == I_MUL:
                for i in range(16):
                    put(row, i, it["a"][i])
                    put(row, 16 + i, it["b"][i])
                    put(row, 48 + i, it["r"][i])
                for i in range(16):
                    put(row, 32 + i, it["q"][i], True)
                for i in range(15):
                    put(row, 64 + i, it["c"][i], True)
            elif k == I_LOOKUP:
                c0 = G_COLS + LW * col
                for i, r in enumerate(it["ins"]):
                    put(row, c0 + i, r)
                for j, o in enumerate(it["outs"]):
                    put(row, c0 + len(it["ins"]) + j, o, True)
        for t in range(self.n_tape):
            assert t in self.home, (self.name, "tape value without a cell", t)


# ---- the program -------------------------------------------------------------------------------------------------------------
# globals of a cycle (fixed by PRE): R* (x, y), the bits of u2 (little end first), the bytes of u1, ok, the mask byte
GL_RX, GL_RY, GL_BITS, GL_U1, GL_OK, GL_MASK, GL_COUNT = 0, 16, 32, 288, 320, 321, 322


def build_pre():
    s = Seg("pre")
    inb = [Ref(K_IN, k) for k in range(128)]  # h, v, r, s: value byte x (little end first) of read w = IN 32 w + x
    s.range_bytes(inb)
    word = lambda w: [s.lin([(inb[32 * w + 2 * k], 1), (inb[32 * w + 2 * k + 1], 256)])[0] for k in range(16)]  # noqa: E731
    h, r, sg = word(0), word(2), word(3)
    vbit = inb[32]
    s.boolean(vbit)
    e_r = s.out_of_range(r, "N", "NM1")
    e_s = s.out_of_range(sg, "N", "NM1")
    e1 = s.or2(e_r, e_s)
    rs = [s.sel(e1, big(BIGS["SUBR"], k), r[k]) for k in range(16)]
    ss = [s.sel(e1, big(BIGS["SUBS"], k), sg[k]) for k in range(16)]
    hs = [s.sel(e1, big(BIGS["SUBH"], k), h[k]) for k in range(16)]
    # t = x^3 + 7; y^2 == t, or e_nr and y^2 == -t
    x2 = s.nn_mul(MOD_P, rs, rs)
    c3 = s.nn_mul(MOD_P, x2, rs)
    t = [s.lin([(c3[k], 1)], 7 if k == 0 else 0)[0] for k in range(16)]
    out = s.hint(H_SQRT, 17, a=t, b=[vbit])
    y, e_nr = out[:16], out[16]
    s.boolean(e_nr)
    yb = s.check_vec16(y)
    s.less_than_const(y, "PM1")
    negt = s.lazy([(t, -1)], "OFFP")
    rr = [s.sel(e_nr, negt[k], t[k]) for k in range(16)]
    s.mul_checked(MOD_P, y, y, rr)
    ybit, yrest = s.lin([(yb[0], 1)], 0, [(0, 1), (1, 0)])
    s.boolean(ybit)
    s.range_bytes([yrest])
    dlt = s.lin([(ybit, 1), (vbit, -1)])[0]
    s.fma_assert(e_nr, dlt, const(0), dlt)  # (1 - e_nr) * (ybit - vbit) == 0
    e_any = s.or2(e1, e_nr)
    ok = s.lin([(e_any, -1)], 1)[0]
    mask = s.lin([(ok, 255)])[0]
    xs = [s.sel(e_nr, big(BIGS["SUBX"], k), rs[k]) for k in range(16)]
    ys = [s.sel(e_nr, big(BIGS["SUBY"], k), y[k]) for k in range(16)]
    # scalars: ri = 1 / r*, u2 = s* ri, u1 = h* ri (mod n)
    ri = s.nn_div(MOD_N, bigv("ONE"), rs)
    u2 = s.hint(H_MULSUB, 16, mod=MOD_N, a=ss, b=ri, c=None, d=None)
    u2b = s.check_vec16(u2)
    s.mul_checked(MOD_N, ss, ri, u2)
    u1 = s.hint(H_MULSUB, 16, mod=MOD_N, a=hs, b=ri, c=None, d=None)
    u1b = s.check_vec16(u1)
    s.mul_checked(MOD_N, hs, ri, u1)
    bits = []
    for b in u2b:
        bb = s.lin([(b, 1)], 0, [(i, 1) for i in range(7)] + [(7, 0)])
        for x in bb:
            s.boolean(x)
        bits += bb
    # the accumulator starts at the offset point
    ax = [s.lin([], limb)[0] for limb in limbs16(OFFSET_POINT[0])]
    ay = [s.lin([], limb)[0] for limb in limbs16(OFFSET_POINT[1])]
    s.out = ax + ay
    s.globs = [v.a for v in xs + ys + bits + u1b + [ok, mask]]
    assert len(s.globs) == GL_COUNT
    s.finish()
    s.parts = split_segment(s)
    return s


def item_reads_writes(it):
    """(tape indices of the item's own segment it reads, tape indices it writes)"""
    k = it["k"]
    if k == I_LIN:
        rd, wr = [r for r, _c in it["known"]], it["outs"]
    elif k == I_SEL:
        rd, wr = [it["b"], it["x"], it["y"]], [it["out"]]
    elif k == I_FMA:
        rd, wr = [it["a"], it["b"], it["c"]] + ([] if it["new"] else [it["d"]]), ([it["d"]] if it["new"] else [])
    elif k == I_MUL:
        rd, wr = list(it["a"]) + list(it["b"]) + list(it["r"]), list(it["q"]) + list(it["c"])
    elif k == I_HINT:
        rd = [r for key in ("a", "b", "c", "d") for r in (it.get(key) or [])]
        wr = it["outs"]
    else:
        rd, wr = it["ins"], it["outs"]
    return {r.a for r in rd if r.kind == K_TAPE}, {r.a for r in wr}


PART_ITEMS = 200  # items a list of a segment's LEAVES should hold


def split_segment(s):
    """the segment's items in the order MAIN, MULS, LEAVES 0, LEAVES 1, ...:
      MAIN    the ancestors of what later segments read — the state the segment leaves (`out`) and, for PRE, the cycle's globals: the hints
              and the limb-wise sums between them, a tenth of the items. PRE's MAIN is what the accumulator chain of a request waits for;
              the MAINs of the other segments are one launch (every segment's input state is known from the chain);
      MULS    the MUL rows (quotient and carries of a 256-bit product: the one item kind besides the hints that needs the 256-bit
              workspace) and what they still need beyond MAIN (the lazy sums on their r side);
      LEAVES  everything else — byte decompositions, range-check lookups, assertions: nine tenths of a segment, small items only — in
              lists that share no tape value they write, a lane each.
    Every list keeps the segment's order, so an item still follows what it reads. Cells, homes and tape indices are untouched (they were
    fixed by finish()); only the order the evaluator walks the items in changes. Returns the sizes [MAIN, MULS, LEAVES...]."""
    rw = [item_reads_writes(it) for it in s.items]
    n = len(s.items)
    needed = set(s.globs) | {r.a for r in s.out}
    main = [False] * n
    for i in range(n - 1, -1, -1):
        if rw[i][1] & needed:
            main[i] = True
            needed |= rw[i][0]
    heavy = lambda i: s.items[i]["k"] in (I_MUL, I_HINT)  # noqa: E731
    muls, needed = [False] * n, set()
    for i in range(n - 1, -1, -1):
        if main[i]:
            continue
        if heavy(i) or (rw[i][1] & needed):
            muls[i] = True
            needed |= rw[i][0]
    rest = [i for i in range(n) if not main[i] and not muls[i]]
    assert not any(heavy(i) for i in rest)
    # components of the leaves: items joined by a tape value one of them writes and another reads
    parent = {i: i for i in rest}

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x

    writer = {}
    for i in rest:
        for t in rw[i][1]:
            writer[t] = i
    for i in rest:
        for t in rw[i][0]:
            if t in writer and writer[t] != i:
                parent[find(i)] = find(writer[t])
    comps = {}
    for i in rest:
        comps.setdefault(find(i), []).append(i)
    parts = max(1, round(len(rest) / PART_ITEMS))
    bins = [[] for _ in range(parts)]
    for c in sorted(comps.values(), key=lambda c: -len(c)):
        b = min(range(parts), key=lambda k: len(bins[k]))
        bins[b] += c
    order = [i for i in range(n) if main[i]]
    sizes = [len(order)]
    order += [i for i in range(n) if muls[i]]
    sizes.append(len(order) - sizes[0])
    for b in bins:
        order += sorted(b)
        sizes.append(len(b))
    # an item still follows what it reads
    pos, seen = {i: k for k, i in enumerate(order)}, {}
    for i in order:
        for t in rw[i][0]:
            assert t in seen and pos[seen[t]] < pos[i], (s.name, "an item would run before what it reads", i)
        for t in rw[i][1]:
            seen[t] = i
    s.items = [s.items[i] for i in order]
    return sizes


def glob_vec(base):
    return [Ref(K_GLOB, base + k) for k in range(16)]


def prev_vec(base):
    return [Ref(K_PREV, base + k) for k in range(16)]


def build_daa():
    s = Seg("daa")
    x1, y1 = prev_vec(0), prev_vec(16)
    x3, y3 = s.double_point(x1, y1)
    x4, y4 = s.add_points(x3, y3, glob_vec(GL_RX), glob_vec(GL_RY))
    bit = Ref(K_GLOBJ, GL_BITS + 255, -1)  # instance j takes bit 255 - j
    s.out = [s.sel(bit, x4[k], x3[k]) for k in range(16)] + [s.sel(bit, y4[k], y3[k]) for k in range(16)]
    s.finish()
    s.parts = split_segment(s)
    return s


def build_fix():
    s = Seg("fix")
    byte = Ref(K_GLOBJ, GL_U1, 1)  # instance C takes byte C of u1
    words = [s.fixed(i, byte, i) for i in range(8)]
    xb = [s.bytes_of(words[i][0], 4) for i in range(8)]
    yb = [s.bytes_of(words[i][1], 4) for i in range(8)]
    tx = [s.lin([(xb[k // 2][2 * (k % 2)], 1), (xb[k // 2][2 * (k % 2) + 1], 256)])[0] for k in range(16)]
    ty = [s.lin([(yb[k // 2][2 * (k % 2)], 1), (yb[k // 2][2 * (k % 2) + 1], 256)])[0] for k in range(16)]
    z = s.is_zero(byte)
    x1, y1 = prev_vec(0), prev_vec(16)
    nty = s.lazy([(ty, -1)], "OFFP")  # -T.y, lazy
    # (x1, y1) + (tx, -ty): the add_points formulas with y2 lazy
    dx = s.lazy([(tx, 1), (x1, -1)], "OFFP")
    dy = s.lazy([(nty, 1), (y1, -1)], "OFFP")
    lam = s.nn_div(MOD_P, dy, dx)
    x3 = s.nn_mulsub(MOD_P, lam, lam, x1, tx)
    s.mul_checked(MOD_P, lam, lam, s.lazy([(x3, 1), (x1, 1), (tx, 1)]))
    dxx = s.lazy([(x1, 1), (x3, -1)], "OFFP")
    y3 = s.nn_mulsub(MOD_P, lam, dxx, y1)
    s.mul_checked(MOD_P, lam, dxx, s.lazy([(y3, 1), (y1, 1)]))
    s.out = [s.sel(z, x1[k], x3[k]) for k in range(16)] + [s.sel(z, y1[k], y3[k]) for k in range(16)]
    s.finish()
    s.parts = split_segment(s)
    return s


def build_post():
    s = Seg("post")
    qx, qy = s.add_points(prev_vec(0), prev_vec(16), bigv("EX"), bigv("EY"))
    s.less_than_const(qx, "PM1")
    s.less_than_const(qy, "PM1")
    # the 64 bytes the netlist hashes: Q.x || Q.y, big end first = FREE elements 0..63 of the cycle's absorb step; the bytes exist
    # (check_vec16 inside add_points made them): find them again as the NEW cells of the byte decompositions of qx / qy
    s.out = qx + qy
    s.finish()
    s.parts = split_segment(s)
    return s


# ---- evaluation (Python integers: the reference semantics of the items) -----------------------------------------------------
class Unsat(Exception):
    pass


def to_int(limbs):
    return sum(int(v) << (16 * k) for k, v in enumerate(limbs))


class Cycle:
    def __init__(self, spec, inputs):
        self.spec, self.inputs = spec, list(inputs)
        self.tape = [None] * spec.tape_per_cycle
        self.pre_base = 0

    def run(self):
        sp = self.spec
        prev = None
        for (t, count, row0, tape0) in sp.runs:
            st = sp.types[t]
            for j in range(count):
                base = tape0 + j * st.n_tape
                self.eval_segment(st, j, base, prev)
                prev = (st, base)
        return self.tape

    def get(self, ref, st, j, base, prev):
        k = ref.kind
        if k == K_TAPE:
            v = self.tape[base + ref.a]
        elif k == K_PREV:
            v = self.tape[prev[1] + prev[0].out[ref.a].a]
        elif k == K_GLOB:
            v = self.tape[self.spec.types[0].globs[ref.a]]
        elif k == K_GLOBJ:
            v = self.tape[self.spec.types[0].globs[ref.a + ref.b * j]]
        elif k == K_CONST:
            v = ref.a
        elif k == K_BIG:
            v = BIG_LIST[ref.a][ref.b]
        elif k == K_IN:
            v = self.inputs[ref.a]
        else:
            raise ValueError(k)
        assert v is not None, (st.name, j, ref.key())
        return v

    def eval_segment(self, st, j, base, prev):
        g = lambda r: self.get(r, st, j, base, prev)  # noqa: E731
        gv = lambda vec: [g(r) for r in vec]          # noqa: E731
        tape = self.tape
        for it in st.items:
            k = it["k"]
            if k == I_LIN:
                S = it["const"] + sum(c * g(r) for r, c in it["known"])
                new = it["new"]
                if not new:
                    if S % PGL:
                        raise Unsat((st.name, "assertion", S))
                    continue
                if len(new) == 1 and new[0] == (0, 0):
                    tape[base + it["outs"][0].a] = S % PGL
                    continue
                if S < 0:
                    raise Unsat((st.name, "negative digits", S))
                for n, (sh, w) in enumerate(new):
                    x = S >> sh
                    if w:
                        x &= (1 << w) - 1
                    tape[base + it["outs"][n].a] = x
                assert sum(tape[base + o.a] << sh for o, (sh, w) in zip(it["outs"], new)) == S
            elif k == I_SEL:
                tape[base + it["out"].a] = g(it["x"]) if g(it["b"]) else g(it["y"])
                assert g(it["b"]) in (0, 1)
            elif k == I_FMA:
                v = (g(it["a"]) * g(it["b"]) + g(it["c"])) % PGL
                if it["new"]:
                    tape[base + it["d"].a] = v
                elif v != g(it["d"]) % PGL:
                    raise Unsat((st.name, "fma assertion"))
            elif k == I_MUL:
                m = P if it["mod"] == MOD_P else N
                a, b, r = gv(it["a"]), gv(it["b"]), gv(it["r"])
                num = to_int(a) * to_int(b) + KMUL * m - to_int(r)
                if num % m or num < 0:
                    raise Unsat((st.name, "mul row", num % m))
                q = num // m
                assert q < (1 << 264)
                ql = [(q >> (16 * i)) & 0xFFFF for i in range(15)] + [q >> 240]
                ml = limbs16(m)
                carry, cs = 0, []
                for kk in range(16):
                    d = 0
                    for tpos, w in ((2 * kk, 1), (2 * kk + 1, 1 << 16)):
                        pt = sum(a[i] * b[tpos - i] for i in range(16) if 0 <= tpos - i < 16)
                        qm = sum((ql[i] - (KMUL if i == 0 else 0)) * ml[tpos - i] for i in range(16) if 0 <= tpos - i < 16)
                        d += w * (pt - qm - (r[tpos] if tpos < 16 else 0))
                    tot = d + carry
                    assert tot % (1 << 32) == 0
                    carry = tot >> 32
                    cs.append(carry)
                assert cs[15] == 0 and all(abs(c) < (1 << 31) for c in cs)
                for i in range(16):
                    tape[base + it["q"][i].a] = ql[i]
                for i in range(15):
                    tape[base + it["c"][i].a] = cs[i] + (1 << 31)
            elif k == I_LOOKUP:
                ins = gv(it["ins"])
                tb = it["table"]
                if tb == T_XOR8:
                    assert all(0 <= x < 256 for x in ins), (st.name, "range", ins)
                    tape[base + it["outs"][0].a] = ins[0] ^ ins[1]
                else:
                    i, C = (tb & 0xFF) - T_FIXED0, j
                    xw, yw = fixed_base_entry(i, C, ins[0])
                    tape[base + it["outs"][0].a], tape[base + it["outs"][1].a] = xw, yw
            elif k == I_HINT:
                h = it["hint"]
                outs = [base + o.a for o in it["outs"]]
                if h == H_MULSUB:
                    m = P if it["mod"] == MOD_P else N
                    v = to_int(gv(it["a"])) * to_int(gv(it["b"]))
                    for opt in (it["c"], it["d"]):
                        if opt is not None:
                            v -= to_int(gv(opt))
                    for i, l in enumerate(limbs16(v % m)):
                        tape[outs[i]] = l
                elif h == H_DIV:
                    m = P if it["mod"] == MOD_P else N
                    den = to_int(gv(it["b"])) % m
                    if den == 0:
                        raise Unsat((st.name, j, "division by zero"))
                    for i, l in enumerate(limbs16(to_int(gv(it["a"])) * pow(den, -1, m) % m)):
                        tape[outs[i]] = l
                elif h == H_SQRT:
                    t = to_int(gv(it["a"])) % P
                    vb = g(it["b"][0])
                    y = pow(t, (P + 1) // 4, P)
                    e_nr = 0
                    if y * y % P != t:
                        e_nr = 1
                        y = pow((-t) % P, (P + 1) // 4, P)
                        assert y * y % P == (-t) % P
                    elif (y & 1) != (vb & 1):
                        y = P - y  # t != 0 here is not needed: y == 0 has one parity only and a mismatch then has no witness
                    for i, l in enumerate(limbs16(y)):
                        tape[outs[i]] = l
                    tape[outs[16]] = e_nr
                elif h == H_ISZERO:
                    x = g(it["a"][0]) % PGL
                    tape[outs[0]] = pow(x, -1, PGL) if x else 0
                    tape[outs[1]] = 0 if x else 1
                elif h == H_GE:
                    tape[outs[0]] = 1 if to_int(gv(it["a"])) >= to_int(BIG_LIST[it["big"]]) else 0
            else:
                raise ValueError(k)


_FIXED_CACHE = {}


def fixed_base_entry(i, C, byte):
    """row `byte` of FixedBaseMulTable<i, C>: (word i of x, word i of y) of byte * 2^(8 C) * G, (0, 0) for byte 0 — boojum's
    create_fixed_base_mul_table (gadgets/tables; absent crate, restated: 32-bit word i = bits [32 i, 32 i + 32))"""
    if C not in _FIXED_CACHE:
        base = ec.mul(1 << (8 * C), ec.G)
        pts, cur = [(0, 0)], None
        for _ in range(255):
            cur = ec.add(cur, base)
            pts.append(cur)
        _FIXED_CACHE[C] = pts
    x, y = _FIXED_CACHE[C][byte]
    return (x >> (32 * i)) & 0xFFFFFFFF, (y >> (32 * i)) & 0xFFFFFFFF


class EcSpec:
    def __init__(self):
        self.types = [build_pre(), build_daa(), build_fix(), build_post()]
        self.runs, row0, tape0 = [], 0, 0
        for t, count in ((0, 1), (1, 256), (2, 32), (3, 1)):
            self.runs.append((t, count, row0, tape0))
            row0 += count * self.types[t].n_rows
            tape0 += count * self.types[t].n_tape
        self.rows_per_cycle, self.tape_per_cycle = row0, tape0
        post = self.types[3]
        # where the 64 key bytes live: byte k (little end first) of limb vector out[0..16) / out[16..32): the NEW byte cells of the
        # LIN items that decompose those limbs (check_vec16)
        self.key_byte_tape = []  # POST tape index of Q.x byte 0..31 (little end first), then Q.y
        byte_of = {}
        for it in post.items:
            if it["k"] == I_LIN and len(it["known"]) == 1 and it["known"][0][1] == 1 and len(it["new"]) == 2 and it["new"][0] == (0, 8):
                byte_of[it["known"][0][0].key()] = [o.a for o in it["outs"]]
        for limb in post.out:
            self.key_byte_tape += byte_of[limb.key()]
        assert len(self.key_byte_tape) == 64

    def evaluate(self, inputs):
        return Cycle(self, inputs).run()

    def outputs(self, tape):
        """(ok, mask, key bytes Q.x || Q.y big end first) of an evaluated cycle"""
        pre = self.types[0]
        post_base = self.runs[3][3]
        kb = [tape[post_base + t] for t in self.key_byte_tape]
        key = bytes(reversed(kb[:32])) + bytes(reversed(kb[32:]))
        return tape[pre.globs[GL_OK]], tape[pre.globs[GL_MASK]], key

    def stats(self):
        per = "; ".join(f"{st.name}: {st.n_rows} rows, {st.n_tape} tape values, {len(st.items)} items" for st in self.types)
        return f"EC section: {per}; {self.rows_per_cycle} rows and {self.tape_per_cycle} tape values per cycle"


def inputs_of(h, v, r, s):
    return list(h.to_bytes(32, "little")) + list(v.to_bytes(32, "little")) + list(r.to_bytes(32, "little")) + list(s.to_bytes(32, "little"))


def self_check(spec):
    rng = random.Random(7)
    cases = []
    h0 = 0x456e9aea5e197a1f1af7a3e85a3212fa4049a3ba34c2289b4c860fc0b0c64ef3  # go-ethereum's ecrecover precompile vector
    cases.append((h0, 1, 0x9242685bf161793cc25603c231bc2f568eb630ea16aa137d2664ac8038825608, 0x4f8ae3bd7535248d0bd448298cc2e2071e56992d0774dc340c368ae950852ada))
    v, r, s = ec.sign(h0, 1, 0xC0FFEE)
    cases.append((h0, v, r, s))          # key 1: address 0x7e5f4552091a69125d5dfcb7b8c2659029395bdf
    cases.append((0, 0, 0, 0))           # an idle cycle's inputs
    cases.append((h0, v, N, s))          # r out of range
    cases.append((h0, v, r, N + 5))      # s out of range
    cases.append((0, v, r, s))           # a zero hash is fine
    x = 5
    while ec.lift_x(x, 0) is not None:
        x += 1
    cases.append((h0, 0, x, s))          # x^3 + 7 is not a square
    cases.append((h0, 1 - v, r, s))      # the other root: another key
    for _ in range(2):
        hh, key, k = rng.randrange(1 << 256), rng.randrange(1, N), rng.randrange(1, N)
        v, r, s = ec.sign(hh, key, k)
        cases.append((hh, v, r, s))
    for (h, v, r, s) in cases:
        tape = spec.evaluate(inputs_of(h, v, r, s))
        ok, mask, key = spec.outputs(tape)
        want_ok, want_addr = ec.ecrecover(h, v, r, s)
        assert ok == want_ok and mask == 255 * want_ok, (hex(r), ok, want_ok)
        if ok:
            assert int.from_bytes(ec.keccak256(key)[12:], "big") == want_addr, "recovered address"
    tape = spec.evaluate(inputs_of(*cases[1]))
    assert int.from_bytes(ec.keccak256(spec.outputs(tape)[2])[12:], "big") == 0x7E5F4552091A69125D5DFCB7B8C2659029395BDF
    return len(cases)


# ---- the byte netlist of the circuit (prefix EK): one Keccak-f[1600] over the public key per cycle --------------------------
FN_FIXEDBASE = 8


def ek_tables():
    """ecrecover.rs:138-176 in order: Xor8, And8, FixedBaseMulTable<i, C> (C outer, i inner: seq_macro over C), ByteSplit<1..4>"""
    fixed = []
    for C in range(32):
        for i in range(8):
            fixed.append(nl.Table(f"FIXED_{i}_{C}", FN_FIXEDBASE, 8 * C + i, 1, 8, 2))
    kt = nl.keccak_tables()
    return kt[:2] + fixed + kt[2:]


EK_FREE_MASK, EK_FREE_OK = 0, 1  # FREE elements of the select step
EK_STATE_OK = 32                 # state byte that holds `ok` after a cycle (bytes 12..31: the masked address, the rest zero)


def build_ek_in(tables):
    """the sponge absorbs Q.x || Q.y (FREE bytes 0..63, big end first) into the ZERO state: the block is the key, the Keccak padding
    0x01 .. 0x80 (constants) and the capacity zeros — every request is one fresh Keccak-256 of 64 bytes"""
    st = nl.StepType("in", tables)
    pad = [0] * 200
    pad[64], pad[135] = 0x01, 0x80
    st.out = [st.lookup("XOR8", nl.free(k), nl.const(0)) if k < 64 else nl.const(pad[k]) for k in range(200)]
    return st


def build_ek_out(tables):
    """address = digest[12..32] masked by 255 * ok; `ok` itself is carried in state byte 32 (the first write's value)"""
    st = nl.StepType("out", tables)
    mask = st.lookup("XOR8", nl.free(EK_FREE_MASK), nl.const(0))
    okb = st.lookup("XOR8", nl.free(EK_FREE_OK), nl.const(0))
    st.out = [st.lookup("AND8", nl.prev(k), mask) if 12 <= k < 32 else okb if k == EK_STATE_OK else nl.const(0) for k in range(200)]
    return st


def make_ek_spec():
    tables = ek_tables()
    spec = nl.Spec("EK", G_COLS, LW, LR, tables, 200, (255, -255, 0, 255))
    assert spec.total_table_rows == 197632, spec.total_table_rows  # `total_tables_len` of setup/base_layer/vk_7.json
    k_in, k_round, k_out = spec.add_step_type(build_ek_in(tables)), spec.add_step_type(gk.build_round(tables)), spec.add_step_type(build_ek_out(tables))
    spec.cycle = [(k_in, [])] + [(k_round, list(gk.RC[r].to_bytes(8, "little"))) for r in range(24)] + [(k_out, [])]
    return spec


def ek_self_check(spec):
    rng = random.Random(3)
    for ok in (1, 0):
        key = [rng.randrange(256) for _ in range(64)]
        st = [rng.randrange(256) for _ in range(200)]
        frees = [key] + [[]] * 24 + [[255 * ok, ok]]
        got = spec.evaluate_cycle(st, frees, 0, 0)
        dig = ec.keccak256(bytes(key))
        want = [0] * 200
        for k in range(12, 32):
            want[k] = dig[k] if ok else 0
        want[EK_STATE_OK] = ok
        assert got == want, "EK netlist != keccak256(key) masked"


# ---- emission ---------------------------------------------------------------------------------------------------------------
def vec_ref(vec):
    """a 16-limb operand as ONE reference (limb i = reference + i): the limbs must be consecutive"""
    e0 = vec[0].enc()
    for i, r in enumerate(vec):
        assert r.kind == vec[0].kind and r.enc() == e0 + i, "operand limbs are not consecutive"
    return e0


def emit_ec(spec, path):
    items, index, cells, homes, outs, rowtab, types = [], [], [], [], [], [], []
    EMPTY = 0xFFFFFFFF
    for st in spec.types:
        item0, index0 = len(items), len(index)
        for it in st.items:
            index.append(len(items) - item0)
            k, row, col = it["k"], it["row"], it["col"]
            assert row < 4096 and col < 256
            w0 = lambda aux: k | (row << 4) | (col << 16) | (aux << 24)  # noqa: E731
            if k == I_LIN:
                c = it["const"] & 0xFFFFFFFFFFFFFFFF
                items += [w0(len(it["known"])), len(it["new"]), c & 0xFFFFFFFF, c >> 32]
                for r, coef in it["known"]:
                    assert -(1 << 31) <= coef < (1 << 31)
                    items += [r.enc(), coef & 0xFFFFFFFF]
                for o, (sh, w) in zip(it["outs"], it["new"]):
                    items += [o.a, sh | (w << 8)]
            elif k == I_SEL:
                items += [w0(0), it["b"].enc(), it["x"].enc(), it["y"].enc(), it["out"].a]
            elif k == I_FMA:
                items += [w0(1 if it["new"] else 0), it["a"].enc(), it["b"].enc(), it["c"].enc(), it["d"].a if it["new"] else it["d"].enc()]
            elif k == I_MUL:
                items += [w0(it["mod"]), vec_ref(it["a"]), vec_ref(it["b"]), vec_ref(it["r"]), it["q"][0].a, it["c"][0].a]
            elif k == I_HINT:
                h = it["hint"]
                out0 = it["outs"][0].a
                if h == H_MULSUB:
                    items += [w0(h), it["mod"], vec_ref(it["a"]), vec_ref(it["b"]), vec_ref(it["c"]) if it["c"] else NONE, vec_ref(it["d"]) if it["d"] else NONE, out0]
                elif h == H_DIV:
                    items += [w0(h), it["mod"], vec_ref(it["a"]), vec_ref(it["b"]), out0]
                elif h == H_SQRT:
                    items += [w0(h), vec_ref(it["a"]), it["b"][0].enc(), out0]
                elif h == H_ISZERO:
                    items += [w0(h), it["a"][0].enc(), out0]
                elif h == H_GE:
                    items += [w0(h), vec_ref(it["a"]), it["big"], out0]
            elif k == I_LOOKUP:
                ins = [r.enc() for r in it["ins"]] + [NONE] * (2 - len(it["ins"]))
                items += [w0(len(it["ins"])), it["table"], ins[0], ins[1], it["outs"][0].a]
        cell0 = len(cells)
        for row in range(st.n_rows):
            for col in range(G_COLS + LW * LR):
                r = st.cells.get((row, col))
                cells.append(EMPTY if r is None else r.enc())
        home0 = len(homes)
        homes += [(st.home[t][0] << 8) | st.home[t][1] for t in range(st.n_tape)]
        out0 = len(outs)
        outs += [r.a for r in st.out]
        assert all(r.kind == K_TAPE for r in st.out) and len(st.out) == 32
        rowtab0 = len(rowtab)
        rowtab += [st.row_table.get(row, 0) for row in range(st.n_rows)]
        types.append((st.n_rows, st.n_tape, item0, len(st.items), index0, cell0, home0, out0, rowtab0))
    pre = spec.types[0]
    in_home = []
    for k in range(128):
        at = [rc for rc, r in pre.cells.items() if r.kind == K_IN and r.a == k]
        in_home.append((min(at)[0] << 8) | min(at)[1])
    o = ["/* GENERATED by tools/gen_ecrecover_circuit.py — do not edit. The EC section of the ECRecover circuit (format: that file's header). */",
         "#ifndef ZKW_ECRECOVER_EC_SPEC_H\n#define ZKW_ECRECOVER_EC_SPEC_H\n#include <stdint.h>"]
    w = o.append
    w(f"#define EC_G {G_COLS}\n#define EC_W {LW}\n#define EC_R {LR}\n#define EC_ROW_CELLS {G_COLS + LW * LR}")
    w(f"#define EC_ROWS_PER_CYCLE {spec.rows_per_cycle}\n#define EC_TAPE_PER_CYCLE {spec.tape_per_cycle}\n#define EC_NUM_TYPES {len(spec.types)}\n#define EC_NUM_RUNS {len(spec.runs)}")
    w(f"#define EC_KMUL {KMUL}\n#define EC_NUM_BIGS {len(BIG_LIST)}\n#define EC_MAX_TAPE {max(st.n_tape for st in spec.types)}\n#define EC_MAX_ITEMS {max(len(st.items) for st in spec.types)}\n#define EC_MAX_ROWS {max(st.n_rows for st in spec.types)}")
    for name, idx in BIGS.items():
        w(f"#define EC_BIG_{name} {idx}")
    w(f"#define EC_GL_RX {GL_RX}\n#define EC_GL_RY {GL_RY}\n#define EC_GL_BITS {GL_BITS}\n#define EC_GL_U1 {GL_U1}\n#define EC_GL_OK {GL_OK}\n#define EC_GL_MASK {GL_MASK}\n#define EC_GL_COUNT {GL_COUNT}")
    w("/* a segment type's items come as MAIN (what the state it leaves — PRE: and the globals — needs), MULS (the MUL rows and what else they need), then the leaves in lists that share no tape value they write: items per part */")
    sq = [i for i, it in enumerate(pre.items) if it["k"] == I_HINT and it["hint"] == H_SQRT]
    assert len(sq) == 1 and sq[0] < pre.parts[0], "the square root is one item of PRE's MAIN part"
    w(f"/* PRE's square-root hint (an item of its MAIN part): k_ec_chain evaluates it with a limb per lane between the items before and after it */\n#define EC_PRE_SQRT_ITEM {sq[0]}")
    mp = max(len(st.parts) for st in spec.types)
    w(f"#define EC_MAX_PARTS {mp}\n#define EC_PART_ITEMS_INIT {{" + ", ".join("{" + ", ".join(str(x) for x in st.parts + [0] * (mp - len(st.parts))) + "}" for st in spec.types) + "}")
    w(f"#define EC_T_XOR8 {T_XOR8}\n#define EC_T_FIXED0 {T_FIXED0}\n#define EC_ROWTAB_PER_INSTANCE {ROWTAB_PER_INSTANCE}")
    w("/* segment types {rows, tape values, item0 (words), items, index0, cell0, home0, out0, rowtab0} */")
    w("#define EC_TYPES_INIT {" + ", ".join("{" + ", ".join(str(x) for x in t) + "}" for t in types) + "}")
    w("/* runs of a cycle {segment type, instances, first row, first tape value} */")
    w("#define EC_RUNS_INIT {" + ", ".join("{" + ", ".join(str(x) for x in r) + "}" for r in spec.runs) + "}")

    def arr(name, vals, per=24):
        w(f"#define {name} {{ \\")
        for i in range(0, len(vals), per):
            o.append("  " + ", ".join(str(v) for v in vals[i:i + per]) + ", \\")
        o.append("}")

    w(f"#define EC_NUM_ITEM_WORDS {len(items)}\n#define EC_NUM_ITEMS {len(index)}\n#define EC_NUM_CELLS {len(cells)}\n#define EC_NUM_HOMES {len(homes)}\n#define EC_NUM_ROWTAB {len(rowtab)}")
    w("/* items (32-bit words; layout: include/zkw_ecrecover.h) */")
    arr("EC_ITEMS_INIT", items)
    w("/* word offset of every item within its segment type's items */")
    arr("EC_ITEM_INDEX_INIT", index)
    w("/* the reference of every cell [cell0 + row * EC_ROW_CELLS + col], 0xFFFFFFFF = empty (zero) */")
    arr("EC_CELLS_INIT", [f"0x{v:X}u" if v >= (1 << 31) else str(v) for v in cells])
    w("/* home cell of every tape value: row << 8 | col */")
    arr("EC_HOME_INIT", homes)
    w("/* the state (accumulator x[16], y[16]) a segment leaves: tape indices */")
    arr("EC_OUT_INIT", outs)
    w("/* lookup table of every row (0: no lookups; | EC_ROWTAB_PER_INSTANCE: + 8 * instance) */")
    arr("EC_ROWTAB_INIT", rowtab)
    w("/* the cycle's globals: tape indices of the PRE segment */")
    arr("EC_GLOB_INIT", pre.globs)
    w("/* 256-bit constants, 16 limbs each */")
    arr("EC_BIG_INIT", [x for b in BIG_LIST for x in b], 16)
    w("/* home cell (row << 8 | col, PRE segment) of input byte k: value byte k % 32 (little end first) of read k / 32 (h, v, r, s) */")
    arr("EC_IN_HOME_INIT", in_home)
    w("/* POST tape index of the key byte the netlist's FREE element k (k < 64) copies: Q.x || Q.y, big end first */")
    kb = spec.key_byte_tape
    arr("EC_KEY_BYTE_INIT", list(reversed(kb[:32])) + list(reversed(kb[32:])))
    o.append("#endif")
    open(path, "w").write("\n".join(o) + "\n")


def main():
    ek = make_ek_spec()
    ek_self_check(ek)
    spec = EcSpec()
    n = self_check(spec)
    inc = os.path.join(nl.root(), "include")
    ek.emit(os.path.join(inc, "zkw_ecrecover_circuit_spec.h"), "tools/gen_ecrecover_circuit.py (ECRecover: Keccak-f over the public key on 80 + 3 x 16 columns)",
            extra_defines=(f"#define EK_FREE_MASK {64 + EK_FREE_MASK}  /* FREE elements of a cycle: 0..63 the key bytes, then the select step's */",
                           f"#define EK_FREE_OK {64 + EK_FREE_OK}", f"#define EK_STATE_OK {EK_STATE_OK}"))
    emit_ec(spec, os.path.join(inc, "zkw_ecrecover_ec_spec.h"))
    print(ek.stats())
    print(spec.stats(), f"; self check: {n} cases")


if __name__ == "__main__":
    main()
