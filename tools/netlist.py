#!/usr/bin/env python3
""""zkw trace v4": one generic netlist format for the bit-gate-heavy base-layer circuits (Sha256RoundFunction, CodeDecommitter,
Keccak256RoundFunction, L1MessagesHasher), on the REFERENCE's geometry and lookup-table sets:

    circuit                columns (copy + lookup w x r)   tables (total rows = `total_tables_len` of setup/base_layer/vk_N.json)
    Sha256RoundFunction    116 + 4 x 9   sha256_round_function.rs:28-39,121-134   TriXor4, Ch4, Maj4, Split4BitChunk<1>, <2>   12 320
    CodeDecommitter        108 + 4 x 11  code_decommitter.rs:28-39,121-134        the same                                     12 320
    Keccak256RoundFunction  86 + 3 x 14  keccak256_round_function.rs:28-39,120-140  Xor8, And8, ByteSplit<1..4>               132 096
    L1MessagesHasher        66 + 3 x 26  linear_hasher.rs:28-39,125-138           the same                                    132 096
plus ONE multiplicity column over the stacked tables. The circuit bodies (gate placement) live in the absent crate
era-zkevm_circuits, so the placement below is this library's own; geometry, table set and capacity are the reference's.

A trace is a sequence of CYCLES (one hash-function call of the precompile: a SHA-256 compression, a Keccak-f[1600]); a cycle is
a fixed sequence of STEPS (SHA: one; Keccak: absorb, 24 x round, select), each step an instance of a STEP TYPE = a netlist:
  * LOOKUP  {table, in[<= 3]} -> out[<= 3]   one width-w slot of the row's lookup columns: cells (in.., out..) zero-padded to w.
            Tables (contents = boojum's create_*_table; keys are the inputs, little end first):
              XOR8 (a, b, a ^ b)   AND8 (a, b, a & b)   BYTESPLIT<k> (x, x mod 2^k, x >> k)
              TRIXOR4 (a, b, c, a ^ b ^ c)   CH4 (e, f, g, (e & f) ^ (~e & g))   MAJ4 (a, b, c, maj)
              SPLIT4<k> (x, lo = x mod 2^k, hi = x >> k, lo << (4 - k) | hi)      the last column is the nibble with its halves swapped
  * GATE    sum_i coef_i * cell_i + const == 0  over consecutive general-purpose cells; coef = +-2^s. Cells are references
            (copies) or NEW cells; the NEW cells of a gate are the digits of the known part S = sum(known) + const at their
            shifts (cell i = bits [s_i, s_{i+1}) of S, the last one takes the rest): one rule covers 32-bit addition (nibbles
            or bytes out + carry), re-chunking a word at another bit phase (rotations), and recomposition (a single NEW cell).
  * HINT    a witness with no constraint of its own: bits [lo, lo + n) of one value, above them bits of another; it must be the
            key of a lookup (its range check, and its home cell).
References are 16-bit: V (a value index: lookup outputs, NEW gate cells and hints, numbered by the generator), HDR + f (header
field of the step's first row), PREV + k (element k of the state the previous step left), CYC + k (of the state before this
cycle), FREE + i (witness element i of the step: message nibbles / block bytes, each used once), RC + k (per-step constant k,
e.g. a byte of Keccak's round constant), CONST + v. Every NEW cell whose range matters is consumed by a lookup.

The generators evaluate every netlist in Python against hashlib / a plain Keccak-f before they write a header."""
import os
import sys

R_HDR, R_PREV, R_CYC, R_FREE, R_RC, R_CONST = 0xC000, 0xC100, 0xC200, 0xC300, 0xC400, 0xC500
MAX_VALUES = 0xC000
FN_XOR8, FN_AND8, FN_BYTESPLIT, FN_TRIXOR4, FN_CH4, FN_MAJ4, FN_SPLIT4 = 1, 2, 3, 4, 5, 6, 7


class Table:
    def __init__(self, name, fn, param, n_in, in_bits, n_out):
        self.name, self.fn, self.param, self.n_in, self.in_bits, self.n_out = name, fn, param, n_in, in_bits, n_out
        self.rows = 1 << (n_in * in_bits)
        self.id = self.offset = None  # id 1-based, offset in the stacked multiplicity column: set by the spec

    def eval(self, ins):
        a = list(ins) + [0, 0, 0]
        f, k = self.fn, self.param
        if f == FN_XOR8:
            return [a[0] ^ a[1]]
        if f == FN_AND8:
            return [a[0] & a[1]]
        if f == FN_BYTESPLIT:
            return [a[0] & ((1 << k) - 1), a[0] >> k]
        if f == FN_TRIXOR4:
            return [a[0] ^ a[1] ^ a[2]]
        if f == FN_CH4:
            return [(a[0] & a[1]) ^ (~a[0] & a[2] & 15)]
        if f == FN_MAJ4:
            return [(a[0] & a[1]) ^ (a[0] & a[2]) ^ (a[1] & a[2])]
        if f == FN_SPLIT4:
            lo, hi = a[0] & ((1 << k) - 1), a[0] >> k
            return [lo, hi, (lo << (4 - k)) | hi]
        raise ValueError(f)  # (FIXEDBASE tables of the ECRecover circuit have field-element outputs: never used by a byte netlist)


def sha_tables():
    return [Table("TRIXOR4", FN_TRIXOR4, 0, 3, 4, 1), Table("CH4", FN_CH4, 0, 3, 4, 1), Table("MAJ4", FN_MAJ4, 0, 3, 4, 1),
            Table("SPLIT4_1", FN_SPLIT4, 1, 1, 4, 3), Table("SPLIT4_2", FN_SPLIT4, 2, 1, 4, 3)]


def keccak_tables():
    return [Table("XOR8", FN_XOR8, 0, 2, 8, 1), Table("AND8", FN_AND8, 0, 2, 8, 1)] + \
           [Table(f"BYTESPLIT_{k}", FN_BYTESPLIT, k, 1, 8, 2) for k in (1, 2, 3, 4)]


def storage_tables():
    """storage_apply.rs:124-140: the Keccak set + ByteSplit<7> = 132 352 rows (`total_tables_len` of vk_10.json)"""
    return keccak_tables() + [Table("BYTESPLIT_7", FN_BYTESPLIT, 7, 1, 8, 2)]


class Val:
    """a value produced inside a step: output `slot` of lookup `item`, NEW cell `slot` of gate `item`, or hint `item`"""
    __slots__ = ("kind", "item", "slot", "index")

    def __init__(self, kind, item, slot):
        self.kind, self.item, self.slot, self.index = kind, item, slot, None


def hdr(f):
    return ("hdr", f)


def prev(k):
    return ("prev", k)


def cyc(k):
    return ("cyc", k)


def free(i):
    return ("free", i)


def rc(k):
    return ("rc", k)


def const(v):
    assert 0 <= v < 256
    return ("const", v)


class StepType:
    def __init__(self, name, tables):
        self.name = name
        self.tables = {t.name: t for t in tables}
        self.ops = []    # [table, [in refs], [Val outs], slot position]
        self.gates = []  # [known [(ref, shift, sign)], new shifts, [Val news], constant]
        self.hints = []  # [Val, (refA, loA, nA), (refB, loB, nB)]
        self.out = None  # list of refs: the state this step leaves
        self.n_free = 0

    # ---- building
    def lookup(self, table, *ins):
        t = self.tables[table]
        assert len(ins) == t.n_in, (table, len(ins))
        outs = [Val("op", len(self.ops), k) for k in range(t.n_out)]
        self.ops.append([t, list(ins), outs, None])
        return outs[0] if t.n_out == 1 else outs

    def hint(self, a, b):
        v = Val("hint", len(self.hints), 0)
        self.hints.append([v, a, b])
        return v

    def gate(self, known, new_shifts, constant=0):
        """known: [(ref, shift, sign)] or [(ref, shift, sign, late)]; NEW cells at new_shifts (ascending): digits of
        S = sum(sign * ref << shift) + constant. Constraint: S - sum(new_i << shift_i) == 0. No NEW cell: a pure assertion.
        A `late` known cell is part of the constraint but not of the evaluation: the generator asserts that the digits at the
        NEW cells' positions are the same without it (a piece that only cancels bits outside them) — the fill then computes
        the gate before the late cell's producer, which shortens the dependency chain; in such a gate the last NEW cell is a digit
        of the common width like the others (without late cells it takes everything that is left). Returns the NEW cells."""
        assert list(new_shifts) == sorted(new_shifts)
        news = [Val("gate", len(self.gates), k) for k in range(len(new_shifts))]
        self.gates.append([[(t[0], t[1], t[2], bool(t[3]) if len(t) > 3 else False) for t in known], list(new_shifts), news, constant])
        return news

    # ---- finishing: group lookups by table, pad to rows, number the values, place the gates, levels
    def finalize(self, lookups_per_row, general_cols, width):
        self.R, self.G, self.W = lookups_per_row, general_cols, width
        order = sorted(range(len(self.ops)), key=lambda j: self.ops[j][0].id)
        self.slots = []  # (lookup index or None for padding, table): tables ascending, every table starts a row
        for j in order:
            t = self.ops[j][0]
            assert t.n_in + t.n_out <= width
            if self.slots and self.slots[-1][1].id != t.id:
                while len(self.slots) % lookups_per_row:
                    self.slots.append((None, self.slots[-1][1]))
            self.ops[j][3] = len(self.slots)
            self.slots.append((j, t))
        while len(self.slots) % lookups_per_row:
            self.slots.append((None, self.slots[-1][1]))
        # values: lookup outputs in slot order, NEW gate cells, hints; home = where the value's own cell is
        self.values = []
        for pos, (j, t) in enumerate(self.slots):
            if j is not None:
                for v in self.ops[j][2]:
                    v.index = len(self.values)
                    self.values.append((0, pos, t.n_in + v.slot))
        self.gate_pos = []
        row, col = 1, 0
        for g, (known, shifts, news, k) in enumerate(self.gates):
            n = len(known) + len(news)
            assert n <= general_cols, f"gate of {n} cells"
            if col + n > general_cols:
                row, col = row + 1, 0
            self.gate_pos.append((row, col))
            for v in news:
                v.index = len(self.values)
                self.values.append((1, g, len(known) + v.slot))
            col += n
        self.gate_rows = row if self.gates else 0
        users = {}
        for j, (t, ins, outs, pos) in enumerate(self.ops):
            for i, r in enumerate(ins):
                if isinstance(r, Val) and r.kind == "hint":
                    users.setdefault(r.item, []).append((pos, i))
        for h, (v, a, b) in enumerate(self.hints):
            assert h in users, "a hint that no lookup consumes"
            v.index = len(self.values)
            self.values.append((2, users[h][0][0], users[h][0][1]))  # home: the first lookup cell that takes it
        assert len(self.values) < MAX_VALUES
        frees = [r[1] for t, ins, *_ in self.ops for r in ins if not isinstance(r, Val) and r[0] == "free"] + \
                [r[1] for g in self.gates for r, *_ in g[0] if not isinstance(r, Val) and r[0] == "free"]
        assert len(frees) == len(set(frees)), "a FREE element is used twice (it has no home cell to copy from)"
        self.n_free = max(frees) + 1 if frees else 0
        self.lookup_rows = len(self.slots) // lookups_per_row
        self.rows = 1 + max(self.lookup_rows, self.gate_rows)
        self._levels()
        used = {r.index for t, ins, *_ in self.ops for r in ins if isinstance(r, Val)}
        self.unchecked = [v for g in self.gates for v in g[2] if v.index not in used]

    def enc(self, r):
        if isinstance(r, Val):
            return r.index
        kind, v = r
        return {"hdr": R_HDR, "prev": R_PREV, "cyc": R_CYC, "free": R_FREE, "rc": R_RC, "const": R_CONST}[kind] + v

    def _levels(self):
        sys.setrecursionlimit(1000000)
        lvl = {}
        # a hint that is the key (only input) of a lookup is evaluated WITH that lookup, as one item: a dependency level less
        # wherever a lookup is keyed by packed bit fields. fused_slot[h] = slot position of that lookup
        self.fused_slot = {}
        for p, (j, _t) in enumerate(self.slots):
            if j is not None and len(self.ops[j][1]) == 1 and isinstance(self.ops[j][1][0], Val) and self.ops[j][1][0].kind == "hint":
                h = self.ops[j][1][0].item
                if h not in self.fused_slot:
                    self.fused_slot[h] = p
        fused_of_slot = {p: h for h, p in self.fused_slot.items()}

        def ref_level(r):
            if not isinstance(r, Val):
                return 0
            return item_level((0, self.ops[r.item][3]) if r.kind == "op" else (1, r.item) if r.kind == "gate" else (2, r.item))

        def item_level(it):
            if it not in lvl:
                kind, i = it
                if kind == 0 and i in fused_of_slot:
                    return item_level((2, fused_of_slot[i]))
                if kind == 0:
                    j = self.slots[i][0]
                    refs = self.ops[j][1] if j is not None else []
                elif kind == 1:
                    refs = [r for r, _, _, late in self.gates[i][0] if not late]
                else:
                    refs = [self.hints[i][1][0], self.hints[i][2][0]]
                lvl[it] = 1 + max([ref_level(r) for r in refs] + [0])
            return lvl[it]

        items = [(item_level((0, p)), 0, p) for p in range(len(self.slots)) if p not in fused_of_slot] + \
                [(item_level((1, g)), 1, g) for g in range(len(self.gates))] + \
                [(item_level((2, h)), 3 if h in self.fused_slot else 2, h) for h in range(len(self.hints))]
        items.sort()
        self.order = [(k, i) for _, k, i in items]
        self.n_levels = items[-1][0] if items else 0
        self.level_start, cur = [], 0
        for idx, (lv, _, _) in enumerate(items):
            while cur < lv:
                self.level_start.append(idx)
                cur += 1
        self.level_start.append(len(items))

    # ---- evaluation (the Python reference semantics of a step)
    def evaluate(self, hdrv, prevv, cycv, freev, rcv):
        vals = [None] * len(self.values)
        self.late_checks = []

        def get(r):
            if isinstance(r, Val):
                return vals[r.index]
            kind, v = r
            return v if kind == "const" else {"hdr": hdrv, "prev": prevv, "cyc": cycv, "free": freev, "rc": rcv}[kind][v]

        for kind, i in self.order:
            if kind == 0:
                j, t = self.slots[i]
                if j is None:
                    continue
                t, ins, outs, _ = self.ops[j]
                iv = [get(r) for r in ins]
                assert all(0 <= x < (1 << t.in_bits) for x in iv), (self.name, t.name, iv)
                for v, x in zip(outs, t.eval(iv)):
                    vals[v.index] = x
            elif kind == 1:
                known, shifts, news, k = self.gates[i]
                S = k + sum(sg * (get(r) << s) for r, s, sg, late in known if not late)
                assert S >= 0, (self.name, "gate", i, S)
                if not news:
                    assert S == 0, (self.name, "assertion gate", i, S)
                    continue
                if any(late for *_, late in known):
                    self.late_checks.append((i, S))  # the full constraint is checked once every value exists
                else:
                    assert S & ((1 << shifts[0]) - 1) == 0
                has_late = any(late for *_, late in known)
                for n, v in enumerate(news):
                    x = S >> shifts[n]
                    if n + 1 < len(shifts):
                        x &= (1 << (shifts[n + 1] - shifts[n])) - 1
                    elif has_late:  # nothing above the last digit cancels in the evaluation: it has the width of the others
                        assert len(shifts) > 1
                        x &= (1 << (shifts[n] - shifts[n - 1])) - 1
                    assert x < 256, (self.name, "gate", i, "cell", n, x)
                    vals[v.index] = x
            else:
                v, (ra, la, na), (rb, lb, nb) = self.hints[i]
                vals[v.index] = ((get(ra) >> la) & ((1 << na) - 1)) | (((get(rb) >> lb) & ((1 << nb) - 1)) << na)
                if kind == 3:  # the lookup this hint keys, in the same item
                    t, ins, outs, _ = self.ops[self.slots[self.fused_slot[i]][0]]
                    assert 0 <= vals[v.index] < (1 << t.in_bits)
                    for o, x in zip(outs, t.eval([vals[v.index]])):
                        vals[o.index] = x
        for i, _ in self.late_checks:  # gates with late cells: the constraint over ALL cells holds
            known, shifts, news, k = self.gates[i]
            full = k + sum(sg * (get(r) << s) for r, s, sg, _ in known)
            assert full == sum(vals[v.index] << sh for v, sh in zip(news, shifts)), (self.name, "gate with late cells", i)
        return vals, [get(r) for r in self.out]


class Spec:
    """one circuit: tables, step types, the cycle's step sequence, geometry, header convention
    header row of a step: [reset, idle, m0, m1] with m0 = m0_a + m0_b * reset, m1 = m1_a + m1_b * idle (the select masks)"""

    def __init__(self, prefix, general_cols, width, lookups_per_row, tables, state_len, masks):
        self.prefix, self.G, self.W, self.R, self.tables = prefix, general_cols, width, lookups_per_row, tables
        self.state_len, self.masks = state_len, masks
        off = 0
        for i, t in enumerate(tables):
            t.id, t.offset = i + 1, off
            off += t.rows
        self.total_table_rows = off
        self.step_types, self.cycle = [], []  # cycle: [(step type index, constant bytes)]

    def add_step_type(self, st):
        st.finalize(self.R, self.G, self.W)
        assert len(st.out) == self.state_len
        self.step_types.append(st)
        return len(self.step_types) - 1

    def rows_per_cycle(self):
        return sum(self.step_types[k].rows for k, _ in self.cycle)

    def header_values(self, reset, idle):
        a0, b0, a1, b1 = self.masks
        return [reset, idle, a0 + b0 * reset, a1 + b1 * idle]

    def evaluate_cycle(self, state, frees, reset, idle):
        """frees: one list per step. Returns the state after the cycle."""
        h = self.header_values(reset, idle)
        cur = list(state)
        for (k, rcb), fr in zip(self.cycle, frees):
            _, cur = self.step_types[k].evaluate(h, cur, state, fr, list(rcb) + [0] * 8)
        return cur

    def emit(self, path, title, extra_defines=()):
        P = self.prefix
        o = [f"/* GENERATED by {title} — do not edit. \"zkw trace v4\" netlist spec (format: tools/netlist.py, include/zkw_netlist.h). */",
             f"#ifndef ZKW_{P}_CIRCUIT_SPEC_H\n#define ZKW_{P}_CIRCUIT_SPEC_H\n#include <stdint.h>\n#include \"zkw_netlist.h\""]
        w = o.append
        w(f"#define {P}_G {self.G}\n#define {P}_W {self.W}\n#define {P}_R {self.R}\n#define {P}_LOOKUP_COL0 {self.G}")
        w(f"#define {P}_MULT_COL {self.G + self.W * self.R}\n#define {P}_COLS {self.G + self.W * self.R + 1}")
        w(f"#define {P}_NUM_TABLES {len(self.tables)}\n#define {P}_TOTAL_TABLE_ROWS {self.total_table_rows}")
        for t in self.tables:
            w(f"#define {P}_T_{t.name} {t.id}")
        w("/* tables {function, parameter, inputs, bits per input, outputs, rows, offset in the multiplicity column} */")
        w(f"#define {P}_TABLES_INIT {{" + ", ".join(f"{{{t.fn}, {t.param}, {t.n_in}, {t.in_bits}, {t.n_out}, {t.rows}, {t.offset}}}" for t in self.tables) + "}")
        w(f"#define {P}_STATE {self.state_len}")
        w(f"#define {P}_MASKS_INIT {{{', '.join(str(x) for x in self.masks)}}}  /* m0 = a0 + b0 * reset, m1 = a1 + b1 * idle */")
        w(f"#define {P}_NUM_STEP_TYPES {len(self.step_types)}\n#define {P}_STEPS_PER_CYCLE {len(self.cycle)}\n#define {P}_ROWS_PER_CYCLE {self.rows_per_cycle()}")
        for d in extra_defines:
            w(d)
        ops, gates, terms, hints, outs, order, levels, homes, types, rowend = [], [], [], [], [], [], [], [], [], []
        for st in self.step_types:
            b = dict(op0=len(ops), gate0=len(gates), term0=len(terms), hint0=len(hints), order0=len(order), level0=len(levels), home0=len(homes))
            for j, t in st.slots:
                if j is None:
                    ops.append((t.id, R_CONST, R_CONST, R_CONST, 0xFFFF))
                else:
                    ins = [st.enc(r) for r in st.ops[j][1]] + [R_CONST] * (3 - len(st.ops[j][1]))
                    ops.append((t.id, ins[0], ins[1], ins[2], st.ops[j][2][0].index))
            for g, (known, shifts, news, k) in enumerate(st.gates):
                row, col = st.gate_pos[g]
                gates.append((len(terms) - b["term0"], len(known), len(news), k, row, col))
                for r, s, sg, late in known:
                    terms.append((st.enc(r), s | (0x80 if sg < 0 else 0) | (0x100 if late else 0)))
                for v, s in zip(news, shifts):
                    terms.append((v.index, s | 0x80))  # NEW cells enter the constraint with coefficient -2^s
            for v, (ra, la, na), (rb, lb, nb) in st.hints:
                hints.append((v.index, st.enc(ra), la, na, st.enc(rb), lb, nb, st.fused_slot.get(len(hints) - b["hint0"], 0xFFFF)))
            for kind, i in st.order:
                order.append(i if kind == 0 else 0x8000 + i if kind == 1 else 0xC000 + i if kind == 2 else 0x4000 + i)
            levels.extend(st.level_start)
            homes.extend(st.values)
            outs.extend(st.enc(r) for r in st.out)
            ends = [0] * st.rows  # general-purpose columns [0, end) of row r hold gate cells (row 0: the header fields)
            ends[0] = 4
            for g, (known, shifts, news, k) in enumerate(st.gates):
                row, col = st.gate_pos[g]
                ends[row] = max(ends[row], col + len(known) + len(news))
            types.append((b["op0"], len(st.slots), b["gate0"], len(st.gates), b["term0"], len(terms) - b["term0"], b["hint0"], len(st.hints),
                          b["order0"], b["level0"], st.n_levels, b["home0"], len(st.values), st.rows, st.lookup_rows, st.gate_rows, st.n_free,
                          len(rowend)))
            rowend.extend(ends)
        w(f"#define {P}_NUM_OPS {len(ops)}\n#define {P}_NUM_GATES {len(gates)}\n#define {P}_NUM_TERMS {len(terms)}\n#define {P}_NUM_HINTS {len(hints)}")
        w(f"#define {P}_NUM_VALUES {len(homes)}\n#define {P}_NUM_ORDER {len(order)}\n#define {P}_NUM_LEVEL_STARTS {len(levels)}")
        w(f"#define {P}_MAX_VALUES {max(len(st.values) for st in self.step_types)}\n#define {P}_MAX_FREE {max(st.n_free for st in self.step_types)}")
        w(f"#define {P}_FREE_PER_CYCLE {sum(self.step_types[k].n_free for k, _ in self.cycle)}\n#define {P}_NUM_ROWEND {len(rowend)}")
        w("/* step types {op0, n_ops, gate0, n_gates, term0, n_terms, hint0, n_hints, order0, level0, n_levels, home0, n_values, rows, lookup_rows, gate_rows, n_free, rowend0} */")
        w(f"#define {P}_STEP_TYPES_INIT {{" + ", ".join("{" + ", ".join(str(x) for x in t) + "}" for t in types) + "}")
        w("/* lookups {table, in0, in1, in2, first output value} in slot order (slot j: row 1 + j / R of the step, lookup j % R of the row) */")
        w(f"#define {P}_OPS_INIT {{ \\")
        for t_, a_, b_, c_, out_ in ops:
            o.append(f"  {{{t_}, {{{a_}, {b_}, {c_}}}, {out_}}}, \\")
        o.append("}")
        w("/* gates {first term (within the step type), known cells, NEW cells, constant, row, first column} */")
        w(f"#define {P}_GATES_INIT {{ \\")
        for g in gates:
            o.append("  {" + ", ".join(str(x) for x in g) + "}, \\")
        o.append("}")
        w("/* gate cells {reference, shift | 0x80 if the coefficient is negative | 0x100 if the cell is LATE (in the constraint, not in the evaluation of the NEW cells)}: a gate's known cells, then its NEW cells */")
        w(f"#define {P}_TERMS_INIT {{" + ", ".join(f"{{{r}, {c}}}" for r, c in terms) + "}")
        w("/* hints {value, ref A, lo A, bits A, ref B, lo B, bits B, slot of the lookup evaluated with the hint or 0xFFFF} */")
        w(f"#define {P}_HINTS_INIT {{" + (", ".join("{" + ", ".join(str(x) for x in h) + "}" for h in hints) or "{0, 0, 0, 0, 0, 0, 0, 65535}") + "}")
        w(f"#define {P}_OUT_INIT {{" + ", ".join(str(x) for x in outs) + "}")
        w("/* evaluation order per step type: lookup slot j, 0x4000 + hint evaluated together with the lookup it keys (hint field 8: that slot), 0x8000 + gate, 0xC000 + hint; level l of a type = [level_start[level0 + l], level_start[level0 + l + 1]) */")
        w(f"#define {P}_ORDER_INIT {{" + ", ".join(str(x) for x in order) + "}")
        w(f"#define {P}_LEVEL_START_INIT {{" + ", ".join(str(x) for x in levels) + "}")
        w("/* where value v has its own cell: {0: lookup slot, cell of the slot | 1: gate, cell of the gate | 2: lookup slot, input cell (a hint)} */")
        w(f"#define {P}_VAL_HOME_INIT {{" + ", ".join(f"{{{a}, {b}, {c}}}" for a, b, c in homes) + "}")
        w("/* per step type and row: general-purpose columns [0, end) of the row hold gate cells (the header fields on row 0) */")
        w(f"#define {P}_GATE_ROW_END_INIT {{" + ", ".join(str(x) for x in rowend) + "}")
        w("/* the steps of a cycle: {step type, first row within the cycle, 8 constants (NL_REF_RC)} */")
        r0, cyc = 0, []
        for k, rcb in self.cycle:
            rcb = list(rcb) + [0] * (8 - len(rcb))
            cyc.append(f"{{{k}, {r0}, {{{', '.join(str(x) for x in rcb)}}}}}")
            r0 += self.step_types[k].rows
        w(f"#define {P}_CYCLE_INIT {{" + ", ".join(cyc) + "}")
        o.append("#endif")
        open(path, "w").write("\n".join(o) + "\n")

    def stats(self):
        per = []
        for st in self.step_types:
            real = sum(1 for j, _ in st.slots if j is not None)
            per.append(f"{st.name}: {real} lookups ({st.lookup_rows} rows of {self.R}), {len(st.gates)} gates / "
                       f"{sum(len(g[0]) + len(g[2]) for g in st.gates)} cells ({st.gate_rows} rows of {self.G}), {len(st.hints)} hints, {st.n_levels} levels, "
                       f"{len(st.values)} values, {st.rows} rows")
        rpc = self.rows_per_cycle()
        return f"{self.prefix}: {'; '.join(per)}; {rpc} rows per cycle -> capacity up to {((1 << 20) - 8) // rpc} in 2^20 rows; tables {self.total_table_rows} rows"


def root():
    return os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
