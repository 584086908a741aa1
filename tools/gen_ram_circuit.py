#!/usr/bin/env python3
"""Generates include/zkw_ram_circuit_spec.h — the declarative layout of the RAMPermutation trace that
libzkw emits ("zkw trace v2"): row types, named slots, lookup cells, polynomial constraints per row
type and the copy links between slots. The fill kernels (csrc), the oracle's fill (oracle/) and both
checkers are written against this one table, so the table IS the contract of `zkw_ram_synthesize`.

What the circuit states follows the reference's circuit for this instance type as far as the in-repo
code shows it (wrapper circuit_definitions/.../base_layer/ram_permutation.rs:26-135: 133 copy columns,
1x15 width-1 range-check lookups, Poseidon2 flattened gate, 2^20 rows; witness semantics
src/witness/individual_circuits/ram_permutation.rs:259-453). The body of `ram_permutation_entry_point`
lives in the absent crate era-zkevm_circuits, so gate placement is OUR design ("parity unpinned" at the
trace-layout level): a fixed template of 6 rows per cycle laid out region-major so that one
lane per cycle writes consecutive rows of every column.
"""
import os
import sys

P = 0xFFFFFFFF00000001
G, L = 133, 15  # general-purpose (copy-permutation) columns, width-1 lookup columns
HEAP_PAGE = 10


class Row:
    def __init__(self, name, per_cycle=True):
        self.name, self.per_cycle = name, per_cycle
        self.slots, self.lookups, self.constraints = [], [], []

    def slot(self, var):
        if var in self.slots:
            return self.slots.index(var)
        if var in self.lookups:
            return G + self.lookups.index(var)
        self.slots.append(var)
        assert len(self.slots) <= G, f"row {self.name}: more than {G} general slots"
        return len(self.slots) - 1

    def lookup(self, var):
        assert var not in self.slots
        if var not in self.lookups:
            self.lookups.append(var)
            assert len(self.lookups) <= L, f"row {self.name}: more than {L} lookups"
        return G + self.lookups.index(var)

    def c(self, terms, why=""):
        """constraint: sum coef * prod(vars) == 0 ; terms = [(coef, [vars...]), ...]"""
        out = []
        for coef, vs in terms:
            assert len(vs) <= 6
            out.append((coef % P, [self.slot(v) for v in vs]))
        self.constraints.append((out, why))

    # ---- gadgets
    def boolean(self, b):
        self.c([(1, [b, b]), (-1, [b])], f"{b} boolean")

    def is_zero(self, terms, w, z, tag):
        """z = [sum(terms) == 0], w = inverse witness. terms: [(coef, var)]"""
        # x*w = 1 - z ; x*z = 0
        self.c([(cf, [v, w]) for cf, v in terms] + [(1, [z]), (-1, [])], f"{tag}: x*w = 1-z")
        self.c([(cf, [v, z]) for cf, v in terms], f"{tag}: x*z = 0")

    def bytes_of(self, x, tag):
        bs = [f"{tag}_b{k}" for k in range(4)]
        for b in bs:
            self.lookup(b)
        self.c([(1, [x])] + [(-(1 << (8 * k)), [bs[k]]) for k in range(4)], f"{x} = sum bytes")
        return bs

    def select(self, flag, a, b, out):
        """out = flag ? a : b  ==> flag*a - flag*b + b - out = 0"""
        self.c([(1, [flag, a]), (-1, [flag, b]), (1, [b]), (-1, [out])], f"{out} = {flag} ? {a} : {b}")


# ---------------- the closed-form section: what the reference's circuits derive INSIDE the trace and this layout used to place -------
# (round 4). Boundary rows, shared by the queue-circuit generators:
#   * SPONGE rows: one flattened Poseidon2 gate per permutation of an overwrite-mode sponge from the zero state with the length in the
#     last capacity word (commit_variable_length_encodable_item / produce_fs_challenges, src/witness/utils.rs:498-550, 269-306): row j
#     absorbs words 8j .. 8j+7 (zero padded) over the capacity the row before left; the first four outputs of the last row are the
#     commitment. A word is a copy of a boundary cell, or FREE (the row's own cell is its home: a struct the trace has no registers for).
#   * links between boundary rows are emitted as kind 5 (cell == cell of another boundary row).
P2_OUT0 = 118  # slots of a flattened Poseidon2 row: 12 inputs, 4 x 12 full-round states, 22 partial S-box outputs, 3 x 12, 12 outputs


def p2_boundary_row(name):
    r = Row(name, False)
    for k in range(12):
        r.slot(f"{name}_i{k}")
    for rr in range(4):
        for k in range(12):
            r.slot(f"{name}_f{rr}_{k}")
    for rr in range(22):
        r.slot(f"{name}_p{rr}")
    for rr in range(3):
        for k in range(12):
            r.slot(f"{name}_f{4 + rr}_{k}")
    for k in range(12):
        r.slot(f"{name}_o{k}")
    assert len(r.slots) == 130 and r.slot(f"{name}_o0") == P2_OUT0
    return r


class ClosedForm:
    """collects the section's rows and its (row, var) == (row, var) copy constraints; resolve() turns them into kind-5 links"""

    SRC_OBS_IN, SRC_FSM_IN, SRC_FSM_OUT, SRC_FLAG, SRC_OBS_OUT = 0, 1, 2, 3, 4  # where a fill takes a FREE cell from: word idx of that encoding

    def __init__(self):
        self.rows, self.copies, self.p2_names, self.free, self.consts, self.products = [], [], [], [], [], []
        self.byte_cells, self.linears = [], []

    def free_cell(self, row, var, src, idx):
        self.free.append((row, var, src, idx))

    def copy(self, row_a, var_a, row_b, var_b):
        self.copies.append((row_a, var_a, row_b, var_b))

    def bytes_of(self, row, limb, tag):
        """the four bytes of a 32-bit limb cell as lookup cells of the same row (limb = sum bytes * 2^8k), computed by a fill; returns their names"""
        bs = row.bytes_of(limb, tag)
        self.byte_cells.append((row, limb, bs[0]))
        return bs

    def linear(self, row, var, terms, const=0, why=""):
        """var = const + sum coef * cell over cells of the same row, computed by a fill (terms: [(coef, var)])"""
        row.c([(1, [var])] + [(-c, [v]) for c, v in terms] + ([(-const, [])] if const else []), why or f"{var} = linear combination")
        self.linears.append((row, var, [(c % P, v) for c, v in terms], const % P))

    def sponge(self, prefix, words, squeeze=0, free_src=None):
        """words: list of None (FREE) | (row, var) (copy) | ("const", v). Returns the rows; word w lives at (rows[w // 8], f"{name}_i{w % 8}").
        squeeze: extra permutations after the absorption (each takes the whole state of the row before)."""
        n = len(words)
        rows = []
        for j in range((n + 7) // 8 + squeeze):
            name = f"{prefix}{j}"
            r = p2_boundary_row(name)
            self.p2_names.append(name)
            absorbing = j < (n + 7) // 8
            for k in range(12):
                cell = f"{name}_i{k}"
                if absorbing and k < 8:
                    w = words[8 * j + k] if 8 * j + k < n else ("const", 0)
                    if w is None:
                        self.free_cell(r, cell, free_src, 8 * j + k)
                        continue
                    if w[0] == "const":
                        r.c([(1, [cell]), (-w[1], [])], f"{cell} = {w[1]}")
                        self.consts.append((r, cell, w[1]))
                    else:
                        self.copy(r, cell, w[0], w[1])
                elif j == 0:
                    v = n if k == 11 else 0  # the zero state with the length in the last element (specialize_for_len)
                    r.c([(1, [cell]), (-v, [])], f"{cell} = {v}")
                    self.consts.append((r, cell, v))
                else:
                    self.copy(r, cell, rows[-1], f"{rows[-1].name}_o{k}")
            rows.append(r)
        self.rows += rows
        return rows

    def word_cell(self, rows, w):
        return rows[w // 8], f"{rows[w // 8].name}_i{w % 8}"

    def resolve(self, all_rows):
        out = []
        for ra, va, rb, vb in self.copies:
            out.append((5, all_rows.index(ra), ra.slot(va), all_rows.index(rb), rb.slot(vb), 0))
        return out

    def tables(self, all_rows, section_rows):
        """what a fill needs next to the links: the section's row types in order, the cells that are constants, the FREE cells, the product
        cells — and a SCHEDULE for a parallel fill: rows grouped into steps by dependency level (a row's copies come from rows of earlier
        steps or from the register rows), every table sorted by step with the steps' index ranges. The PI row is the last step."""
        idx = {id(r): k for k, r in enumerate(all_rows)}
        pi = next(r for r in all_rows if r.name == "PI")
        sect = {id(r) for r in section_rows}
        level = {}
        for r in section_rows:  # section_rows is a valid sequential order: sources come first
            lv = 0
            for ra, _, rb, _ in self.copies:
                if ra is r and id(rb) in sect:
                    assert id(rb) in level, f"{r.name} copies from {rb.name}, which is filled later"
                    lv = max(lv, level[id(rb)] + 1)
            level[id(r)] = lv
        n_levels = max(level.values()) + 1
        level[id(pi)] = n_levels
        steps = [[r for r in section_rows if level[id(r)] == lv] for lv in range(n_levels)] + [[pi]]
        step_of = {id(r): k for k, rows_ in enumerate(steps) for r in rows_}
        copies = sorted(((step_of[id(ra)], idx[id(ra)], ra.slot(va), idx[id(rb)], rb.slot(vb)) for ra, va, rb, vb in self.copies if id(ra) in step_of))
        consts = sorted((step_of[id(r)], idx[id(r)], r.slot(v), val) for r, v, val in self.consts)
        free = sorted((step_of[id(r)], idx[id(r)], r.slot(v), src, i) for r, v, src, i in self.free)
        prods = sorted((step_of[id(r)], idx[id(r)], r.slot(t), r.slot(a), r.slot(b)) for r, t, a, b in self.products)
        byts = sorted((step_of[id(r)], idx[id(r)], r.slot(limb), r.slot(b0)) for r, limb, b0 in self.byte_cells)
        lins = sorted(((step_of[id(r)], idx[id(r)], r.slot(v), [(c, r.slot(x)) for c, x in terms], const) for r, v, terms, const in self.linears), key=lambda e: e[:3])
        lin_terms, lin_entries = [], []
        for st, ri, col, terms, const in lins:
            lin_entries.append((st, ri, col, len(lin_terms), len(terms), const))
            lin_terms += [(c_, coef) for coef, c_ in terms]

        def ranges(tab):
            return [(sum(1 for e in tab if e[0] < k), sum(1 for e in tab if e[0] == k)) for k in range(len(steps))]

        step_rows = [idx[id(r)] for rows_ in steps for r in rows_]
        row0 = [sum(len(x) for x in steps[:k]) for k in range(len(steps))]
        sched = [(row0[k], len(steps[k])) + ranges(copies)[k] + ranges(consts)[k] + ranges(free)[k] + ranges(prods)[k] + ranges(byts)[k] + ranges(lin_entries)[k]
                 for k in range(len(steps))]
        return {"first": all_rows.index(section_rows[0]), "n": len(section_rows),
                "consts": [e[1:] for e in consts], "free": [e[1:] for e in free], "products": [e[1:] for e in prods],
                "copies": [e[1:] for e in copies], "steps": sched, "step_rows": step_rows,
                "bytes": [e[1:] for e in byts], "linears": [e[1:] for e in lin_entries], "lin_terms": lin_terms}


class Selections:
    """Rows of flag-conditional selections, opened as they fill up (each holds its own copy of the flag). The flag is the instance's
    start flag (FREE boolean in the first row: the closed form's word, fill source SRC_FLAG idx 0) or a boolean cell that exists already
    (completion in BND_OUT). Every selected value is a fresh cell tied to its target by a copy."""

    def __init__(self, cf, prefix, flag_cell=None):
        self.cf, self.prefix, self.flag_cell, self.rows, self.n = cf, prefix, flag_cell, [], 0

    def _row(self, need):
        if self.rows and len(self.rows[-1].slots) + need <= G:
            return self.rows[-1]
        r = Row(f"{self.prefix}{len(self.rows)}", False)
        if self.flag_cell is None and not self.rows:
            r.boolean("flag")
            self.cf.free_cell(r, "flag", ClosedForm.SRC_FLAG, 0)
        else:
            r.slot("flag")
            self.cf.copy(r, "flag", *(self.flag_cell or (self.rows[0], "flag")))
        self.rows.append(r)
        return r

    def flag(self):
        """(row, var) of the flag's first cell"""
        return (self._row(0), "flag")

    def _names(self, letters):
        self.n += 1
        return [f"s{self.n - 1}_{x}" for x in letters]

    def sel3(self, a_cell, b_cell, target, why=""):
        """target = flag ? a : b"""
        row = self._row(3)
        a, b, t = self._names("abt")
        row.c([(1, [t]), (-1, ["flag", a]), (-1, [b]), (1, ["flag", b])], why or f"{target[1]} = flag ? {a_cell[1]} : {b_cell[1]}")
        self.cf.copy(row, a, *a_cell)
        self.cf.copy(row, b, *b_cell)
        self.cf.copy(row, t, *target)

    def sel2(self, init, b_cell, target, why=""):
        """target = flag ? init (a constant) : b"""
        row = self._row(2)
        b, t = self._names("bt")
        row.c([(1, [t]), (-init, ["flag"]), (-1, [b]), (1, ["flag", b])], why or f"{target[1]} = flag ? {init} : {b_cell[1]}")
        self.cf.copy(row, b, *b_cell)
        self.cf.copy(row, t, *target)

    def gate(self, b_cell, why=""):
        """a fresh cell = flag ? b : 0; returns it (a sponge word copies it)"""
        row = self._row(2)
        b, t = self._names("bt")
        row.c([(1, [t]), (-1, ["flag", b])], why or f"flag ? {b_cell[1]} : 0")
        self.cf.copy(row, b, *b_cell)
        self.cf.products.append((row, t, "flag", b))  # nothing to copy it from: a fill computes it
        return (row, t)

    def zero_if_flag(self, target, why=""):
        """flag => target = 0 (nothing is said when the flag is clear)"""
        row = self._row(1)
        (t,) = self._names("t")
        row.c([(1, ["flag", t])], why or f"flag => {target[1]} = 0")
        self.cf.copy(row, t, *target)

    def free_unless_flag(self, b_cell, src, idx, why=""):
        """a fresh FREE cell w (word idx of fill source src) with flag = 0 => w = b; returns it (a sponge word copies it). With the
        completion flag: a word of the hidden FSM output is the register unless the instance completes (nobody consumes that output, and the
        reference's builders put placeholders there)"""
        row = self._row(2)
        b, w = self._names("bw")
        row.c([(1, [w]), (-1, [b]), (-1, ["flag", w]), (1, ["flag", b])], why or f"(1 - flag) * (word - {b_cell[1]}) = 0")
        self.cf.copy(row, b, *b_cell)
        self.cf.free_cell(row, w, src, idx)
        return (row, w)

    def eq_unless_flag(self, a_cell, b_cell, why=""):
        """flag = 0 => a = b (copies of both)"""
        row = self._row(2)
        a, b = self._names("ab")
        row.c([(1, [a]), (-1, [b]), (-1, ["flag", a]), (1, ["flag", b])], why or f"(1 - flag) * ({a_cell[1]} - {b_cell[1]}) = 0")
        self.cf.copy(row, a, *a_cell)
        self.cf.copy(row, b, *b_cell)

    def not_flag(self, target, why=""):
        """target = 1 - flag"""
        row = self._row(1)
        (t,) = self._names("t")
        row.c([(1, [t]), (1, ["flag"]), (-1, [])], why or f"{target[1]} = 1 - flag")
        self.cf.copy(row, t, *target)


def challenge_links(cf, BIN, CH, n_absorb, per_rep):
    """BND_IN's g.c{rep}_{k} (k = 1..per_rep; challenge 0 is the constant ONE) are the squeezed words in order: eight per permutation,
    the first eight from the state the last absorption left (produce_fs_challenges, utils.rs:520-548)"""
    for rep in range(2):
        for k in range(1, per_rep + 1):
            j = per_rep * rep + k - 1
            row = CH[n_absorb - 1 + j // 8]
            cf.copy(BIN, f"g.c{rep}_{k}", row, f"{row.name}_o{j % 8}")


def build():
    PU, PS = Row("PU"), Row("PS")
    A, B, Cc, D = Row("A"), Row("B"), Row("C"), Row("D")
    BIN, BOUT, PI = Row("BND_IN", False), Row("BND_OUT", False), Row("PI", False)

    # ---------------- Poseidon2 rows: [in 12][4 x state after full round][22 x sbox out of element 0][4 x state]
    def poseidon(row, enc, cap_prev, out):
        for k in range(8):
            row.slot(enc[k])
        for k in range(4):
            row.slot(cap_prev[k])
        for r in range(4):
            for k in range(12):
                row.slot(f"{row.name}_f{r}_{k}")
        for r in range(22):
            row.slot(f"{row.name}_p{r}")
        for r in range(3):
            for k in range(12):
                row.slot(f"{row.name}_f{4 + r}_{k}")
        for k in range(12):
            row.slot(out[k])
        assert len(row.slots) == 130

    eu = [f"eu{k}" for k in range(8)]
    es = ["ts", "page", "es2", "es3", "es4", "es5", "es6", "v4"]  # es0 = ts, es1 = page, es7 = v4
    poseidon(PU, eu, [f"p.uh{8 + k}" for k in range(4)], [f"uo{k}" for k in range(12)])
    poseidon(PS, es, [f"p.sh{8 + k}" for k in range(4)], [f"so{k}" for k in range(12)])
    # range checks riding in the Poseidon rows' spare slots / lookup cells
    for x in ("idx", "v0", "v1"):
        PU.bytes_of(x, x)
    for x in ("ts", "page", "v4"):
        PS.bytes_of(x, x)

    # ---------------- row A: encoding (es2, es3), grand-product contributions and accumulators
    # (array-like groups are pre-allocated contiguously so that fill code can index them)
    for v in eu + es:
        A.slot(v)
    for r in range(2):
        for k in range(1, 9):
            A.slot(f"g.c{r}_{k}")
    for r in range(2):
        for v in (f"lc{r}", f"p.lhs{r}", f"nl{r}", f"lhs{r}", f"rc{r}", f"p.rhs{r}", f"nr{r}", f"rhs{r}"):
            A.slot(v)
    A.boolean("rw")
    A.boolean("ptr")
    A.c([(1, ["es2"]), (-1, ["idx"]), (-(1 << 32), ["rw"]), (-(1 << 33), ["ptr"])], "es2 = idx + rw<<32 + ptr<<33")
    for x in ("v2", "v3"):
        A.bytes_of(x, x)
    b5 = [f"v5_b{k}" for k in range(4)]
    for b in b5:
        A.lookup(b)
    A.c([(1, ["es3"]), (-1, ["v0"]), (-(1 << 32), [b5[0]]), (-(1 << 40), [b5[1]]), (-(1 << 48), [b5[2]])], "es3")
    A.slot("v5_b3c")  # copy of byte 3 of limb 5 for row B (general slot so that a copy link can carry it)
    A.c([(1, ["v5_b3c"]), (-1, [b5[3]])], "v5_b3c = v5_b3")
    for r in range(2):
        ch = [None] + [f"g.c{r}_{k}" for k in range(1, 9)]  # challenge 0 is the constant ONE (utils.rs:533)
        for side, enc, acc in (("l", eu, "lhs"), ("r", es, "rhs")):
            lc = f"{side}c{r}"
            A.c([(-1, [lc]), (1, [ch[8]]), (1, [enc[0]])] + [(1, [enc[k], ch[k]]) for k in range(1, 8)],
                f"{lc} = c8 + sum enc_k c_k")
            A.c([(1, [f"p.{acc}{r}", lc]), (-1, [f"n{side}{r}"])], f"n{side}{r} = acc*contribution")
            A.select("can_pop", f"n{side}{r}", f"p.{acc}{r}", f"{acc}{r}")

    # ---------------- row B: es4..es6, sort limb 0
    b6 = [f"v6_b{k}" for k in range(4)]
    b7 = [f"v7_b{k}" for k in range(4)]
    for b in b6 + b7:
        B.lookup(b)
    B.c([(1, ["es4"]), (-1, ["v1"]), (-(1 << 32), ["v5_b3c"]), (-(1 << 40), [b6[0]]), (-(1 << 48), [b6[1]])], "es4")
    B.c([(1, ["es5"]), (-1, ["v2"]), (-(1 << 32), [b6[2]]), (-(1 << 40), [b6[3]]), (-(1 << 48), [b7[0]])], "es5")
    B.c([(1, ["es6"]), (-1, ["v3"]), (-(1 << 32), [b7[1]]), (-(1 << 40), [b7[2]]), (-(1 << 48), [b7[3]])], "es6")
    # long subtraction cur - prev of the sorting key (ts least significant, then idx, then page)
    B.bytes_of("d0", "d0")
    B.boolean("bw0")
    B.c([(1, ["d0"]), (-1, ["ts"]), (1, ["p.ts"]), (-(1 << 32), ["bw0"])], "d0 = ts - p.ts + 2^32 bw0")

    # ---------------- row C: sort limbs 1,2; same-cell / value logic; nondeterministic writes
    Cc.bytes_of("d1", "d1")
    Cc.bytes_of("d2", "d2")
    Cc.boolean("bw1")
    Cc.boolean("bw2")
    Cc.c([(1, ["d1"]), (-1, ["idx"]), (1, ["p.idx"]), (1, ["bw0"]), (-(1 << 32), ["bw1"])], "d1")
    Cc.c([(1, ["d2"]), (-1, ["page"]), (1, ["p.page"]), (1, ["bw1"]), (-(1 << 32), ["bw2"])], "d2")
    Cc.c([(1, ["can_pop", "bw2"])], "previous key must not be greater (when popping)")
    Cc.is_zero([(1, "idx"), (-1, "p.idx")], "w_idx", "z_idx", "idx == p.idx")
    Cc.is_zero([(1, "page"), (-1, "p.page")], "w_page", "z_page", "page == p.page")
    Cc.c([(1, ["z_idx", "z_page"]), (-1, ["same"])], "same cell")
    val = ["es3", "es4", "es5", "es6", "v4"]
    for k, v in enumerate(val):
        Cc.is_zero([(1, v), (-1, f"p.{v}")], f"w_eq{k}", f"z_eq{k}", f"{v} == p.{v}")
        Cc.is_zero([(1, v)], f"w_z{k}", f"z_z{k}", f"{v} == 0")
    Cc.c([(1, ["peq"]), (-1, []), (1, ["ptr", "ptr"]), (-2, ["ptr", "p.ptr"]), (1, ["p.ptr", "p.ptr"])], "peq = 1-(ptr-p.ptr)^2")
    Cc.c([(1, ["z_eq0", "z_eq1", "z_eq2"]), (-1, ["eq_a"])], "eq_a")
    Cc.c([(1, ["eq_a", "z_eq3", "z_eq4", "peq"]), (-1, ["value_equal"])], "value_equal")
    Cc.c([(1, ["z_z0", "z_z1", "z_z2"]), (-1, ["zz_a"])], "zz_a")
    Cc.c([(1, ["zz_a", "z_z3", "z_z4"]), (-1, ["zz_a", "z_z3", "z_z4", "ptr"]), (-1, ["all_zero"])], "all_zero")
    # read of an already-touched cell returns the previous value; read of a fresh cell returns zero
    Cc.c([(1, ["can_pop", "same"]), (-1, ["can_pop", "same", "rw"]), (-1, ["can_pop", "same", "value_equal"]),
          (1, ["can_pop", "same", "rw", "value_equal"])], "can_pop*(1-rw)*same*(1-value_equal) = 0")
    Cc.c([(1, ["can_pop"]), (-1, ["can_pop", "same"]), (-1, ["can_pop", "rw"]), (1, ["can_pop", "same", "rw"]),
          (-1, ["can_pop", "all_zero"]), (1, ["can_pop", "same", "all_zero"]), (1, ["can_pop", "rw", "all_zero"]),
          (-1, ["can_pop", "same", "rw", "all_zero"])], "can_pop*(1-rw)*(1-same)*(1-all_zero) = 0")
    Cc.is_zero([(1, "ts")], "w_ts", "z_ts", "ts == 0")
    Cc.c([(1, ["page", "w_heap"]), (-HEAP_PAGE, ["w_heap"]), (1, ["z_heap"]), (-1, [])], "heap: x*w = 1-z")
    Cc.c([(1, ["page", "z_heap"]), (-HEAP_PAGE, ["z_heap"])], "heap: x*z = 0")
    Cc.c([(1, ["can_pop", "z_ts", "z_heap", "rw"]), (-1, ["can_pop", "z_ts", "z_heap", "rw", "ptr"]), (-1, ["nd"])],
         "nd = can_pop & ts==0 & heap page & write & !ptr")
    Cc.c([(1, ["cnt"]), (-1, ["p.cnt"]), (-1, ["nd"])], "cnt = p.cnt + nd")

    # ---------------- row D: queue bookkeeping and head selection
    D.is_zero([(1, "p.len_u")], "w_lu", "z_lu", "len_u == 0")
    D.is_zero([(1, "p.len_s")], "w_ls", "z_ls", "len_s == 0")
    D.c([(1, ["z_lu"]), (-1, ["z_ls"])], "both queues empty together")
    D.c([(1, ["can_pop"]), (1, ["z_lu"]), (-1, [])], "can_pop = 1 - empty")
    D.c([(1, ["len_u"]), (-1, ["p.len_u"]), (1, ["can_pop"])], "len_u = p.len_u - can_pop")
    D.c([(1, ["len_s"]), (-1, ["p.len_s"]), (1, ["can_pop"])], "len_s = p.len_s - can_pop")
    for q, o in (("uh", "uo"), ("sh", "so")):
        for k in range(12):
            D.select("can_pop", f"{o}{k}", f"p.{q}{k}", f"{q}{k}")

    # ---------------- boundary rows
    regs = ([f"uh{k}" for k in range(12)] + [f"sh{k}" for k in range(12)] +
            ["len_u", "len_s", "lhs0", "lhs1", "rhs0", "rhs1", "ts", "idx", "page", "es3", "es4", "es5", "es6", "v4",
             "ptr", "cnt"])
    for v in regs:
        BIN.slot(v)  # the "cycle -1" value of every register = hidden_fsm_input
    for r in range(2):
        for k in range(1, 9):
            BIN.slot(f"g.c{r}_{k}")
    for v in regs:
        BOUT.slot(v)  # copy of the last cycle = hidden_fsm_output
    for q in ("u", "s"):
        for k in range(12):
            BOUT.slot(f"tail_{q}{k}")
    BOUT.slot("completion")
    BOUT.boolean("completion")
    BOUT.is_zero([(1, "len_u")], "w_end", "z_end", "queue exhausted")
    for q, h in (("u", "uh"), ("s", "sh")):
        for k in range(12):
            BOUT.c([(1, ["z_end", f"{h}{k}"]), (-1, ["z_end", f"tail_{q}{k}"])], f"empty queue: head == tail ({q}{k})")
    BOUT.c([(1, ["completion"]), (-1, ["completion", "z_end"])], "completion => queues exhausted")
    for r in range(2):
        BOUT.c([(1, ["completion", f"lhs{r}"]), (-1, ["completion", f"rhs{r}"])], f"completion => lhs{r} == rhs{r}")
    for k in range(4):
        PI.slot(f"pi{k}")

    # ---------------- closed-form section (see ClosedForm): challenges, input / output commitments, start-flag selection
    cf = ClosedForm()
    q25 = lambda q: [f"{q}h{k}" for k in range(12)] + [f"tail_{q[0]}{k}" for k in range(12)] + [f"len_{q[0]}"]  # noqa: E731  head, tail, length
    # observable input (RamPermutationInputData, the block's FIRST instance's: postprocessing/mod.rs:358-364): FREE words
    OI = cf.sponge("OI", [None] * 51, free_src=ClosedForm.SRC_OBS_IN)
    oi = lambda w: cf.word_cell(OI, w)  # noqa: E731
    # hidden FSM input (RamPermutationFSMInputOutput, W/ram_permutation.rs:385-406): FREE words, tied to the registers by the selection rows
    FI = cf.sponge("FI", [None] * 69, free_src=ClosedForm.SRC_FSM_IN)
    fi = lambda w: cf.word_cell(FI, w)  # noqa: E731
    # value limbs of the previous value: the registers hold its ENCODING elements (memory_query.rs:60-110), the closed form its eight limbs
    def value_row(name):
        r = Row(name, False)
        for k in range(8):
            r.slot(f"{name}_v{k}")
        b = {l: [f"{name}_v{l}_b{k}" for k in range(4)] for l in (5, 6, 7)}
        for l in (5, 6, 7):
            for x in b[l]:
                r.lookup(x)
            r.c([(1, [f"{name}_v{l}"])] + [(-(1 << (8 * k)), [b[l][k]]) for k in range(4)], f"v{l} = sum bytes")
        pk = ((3, 0, b[5][0], b[5][1], b[5][2]), (4, 1, b[5][3], b[6][0], b[6][1]), (5, 2, b[6][2], b[6][3], b[7][0]), (6, 3, b[7][1], b[7][2], b[7][3]))
        for e, l, x0, x1, x2 in pk:
            r.c([(1, [f"{name}_e{e}"]), (-1, [f"{name}_v{l}"]), (-(1 << 32), [x0]), (-(1 << 40), [x1]), (-(1 << 48), [x2])], f"es{e} of the limbs")
        return r
    VIN, VOUT = value_row("VIN"), value_row("VOUT")
    for k in range(8):
        cf.copy(VIN, f"VIN_v{k}", *fi(4 + 25 + 25 + 3 + 2 + k))
    for e in (3, 4, 5, 6):
        cf.copy(VOUT, f"VOUT_e{e}", BOUT, f"es{e}")
    for k in range(8):
        if k != 4:
            cf.free_cell(VOUT, f"VOUT_v{k}", ClosedForm.SRC_FSM_OUT, 59 + k)
    # hidden FSM output: the registers after the last cycle
    fo_words = ([(BOUT, f"lhs{r}") for r in range(2)] + [(BOUT, f"rhs{r}") for r in range(2)] + [(BOUT, v) for v in q25("u")] + [(BOUT, v) for v in q25("s")] +
                [(BOUT, "ts"), (BOUT, "idx"), (BOUT, "page"), (BOUT, "idx"), (BOUT, "page")] +
                [(VOUT, f"VOUT_v{k}") if k != 4 else (BOUT, "v4") for k in range(8)] + [(BOUT, "ptr"), (BOUT, "cnt")])
    assert len(fo_words) == 69
    FO = cf.sponge("FO", fo_words)
    cf.copy(VOUT, "VOUT_v4", BOUT, "v4")
    # start-flag selection (utils.rs:269-306 `start_flag`-conditional state; W/ram_permutation.rs:355-365): register at cycle -1 =
    # start ? the observable input's queue state (accumulators ONE, previous key / value / counter ZERO) : the hidden FSM input
    S0, S1 = Row("SEL0", False), Row("SEL1", False)
    S0.boolean("start")
    cf.free_cell(S0, "start", ClosedForm.SRC_FLAG, 0)
    S1.slot("start1")
    cf.copy(S1, "start1", S0, "start")
    n_sel = [0]

    def select3(obs_cell, fi_cell, target):
        row, st = (S0, "start") if len(S0.slots) + 3 <= G else (S1, "start1")
        a, b, t = (f"s{n_sel[0]}_{x}" for x in "abt")
        n_sel[0] += 1
        row.c([(1, [t]), (-1, [st, a]), (-1, [b]), (1, [st, b])], f"{target[1]} = start ? observable input : hidden FSM input")
        cf.copy(row, a, *obs_cell)
        cf.copy(row, b, *fi_cell)
        cf.copy(row, t, *target)

    def select2(init, fi_cell, target):
        b, t = (f"s{n_sel[0]}_{x}" for x in "bt")
        n_sel[0] += 1
        S1.c([(1, [t]), (-init, ["start1"]), (-1, [b]), (1, ["start1", b])], f"{target[1]} = start ? {init} : hidden FSM input")
        cf.copy(S1, b, *fi_cell)
        cf.copy(S1, t, *target)

    for qi, q in enumerate(("u", "s")):
        regs_q = [(BIN, f"{q}h{k}") for k in range(12)] + [(BOUT, f"tail_{q}{k}") for k in range(12)] + [(BIN, f"len_{q}")]
        for k in range(25):
            select3(oi(25 * qi + k), fi(4 + 25 * qi + k), regs_q[k])
    for r in range(2):
        select2(1, fi(r), (BIN, f"lhs{r}"))
        select2(1, fi(2 + r), (BIN, f"rhs{r}"))
    for k, v in enumerate(("ts", "idx", "page")):
        select2(0, fi(54 + k), (BIN, v))
    for k, v in enumerate(("idx", "page")):  # previous_full_key repeats (index, page)
        select2(0, fi(57 + k), (BIN, v))
    for e in (3, 4, 5, 6):
        select2(0, (VIN, f"VIN_e{e}"), (BIN, f"es{e}"))
    select2(0, fi(59 + 4), (BIN, "v4"))
    select2(0, fi(67), (BIN, "ptr"))
    select2(0, fi(68), (BIN, "cnt"))
    # Fiat-Shamir challenges of the permutation argument (produce_fs_challenges, utils.rs:498-550, called with the observable input's
    # queue tails and lengths, W/ram_permutation.rs:80-90): 26 words -> 4 absorbing permutations, 8 challenges, one more permutation, 8 more
    fs_words = [oi(12 + k) for k in range(12)] + [oi(24)] + [oi(25 + 12 + k) for k in range(12)] + [oi(49)]
    CH = cf.sponge("CH", fs_words, squeeze=1)
    for r in range(2):
        for k in range(1, 9):
            cf.copy(BIN, f"g.c{r}_{k}", CH[3 + r], f"{CH[3 + r].name}_o{k - 1}")
    # compact form and the public input (ClosedFormInputCompactForm::from_full_form + commit: utils.rs:294-303)
    last = lambda rows_: rows_[-1]  # noqa: E731
    cp_words = ([(S0, "start"), (BOUT, "completion")] + [(last(OI), f"{last(OI).name}_o{k}") for k in range(4)] + [("const", 0)] * 4 +
                [(last(FI), f"{last(FI).name}_o{k}") for k in range(4)] + [(last(FO), f"{last(FO).name}_o{k}") for k in range(4)])
    CP = cf.sponge("CP", cp_words)
    for k in range(4):
        cf.copy(PI, f"pi{k}", last(CP), f"{last(CP).name}_o{k}")
    cf_rows = OI + FI + [VIN, VOUT] + FO + [S0, S1] + CH + CP

    rows = [PU, PS, A, B, Cc, D, BIN, BOUT, PI] + cf_rows

    # ---------------- copy links
    # every occurrence of a variable is linked to its "home": the first row (in `rows` order) that holds
    # the un-prefixed name. p.x at cycle i links to x's home at cycle i-1 (BND_IN for i = 0); g.x links to
    # BND_IN; registers in BND_OUT link to their home at the last cycle.
    cyc_rows = [r for r in rows if r.per_cycle]
    home = {}
    for ri, r in enumerate(rows):
        if not r.per_cycle:
            continue
        for v in r.slots + r.lookups:
            if not v.startswith(("p.", "g.")) and v not in home:
                home[v] = (ri, r.slot(v))
    links = []  # (kind, row_a, col_a, row_b, col_b)   kind 0 same cycle, 1 prev cycle (col in BIN via row_c), 2 global, 3 last cycle
    for ri, r in enumerate(rows):
        for v in r.slots + r.lookups:
            col = r.slot(v)
            if r.per_cycle:
                if v.startswith("p."):
                    hv = v[2:]
                    assert hv in home, v
                    assert hv in BIN.slots, f"{v}: register missing from BND_IN"
                    links.append((1, ri, col, home[hv][0], home[hv][1], BIN.slot(hv)))
                elif v.startswith("g."):
                    links.append((2, ri, col, rows.index(BIN), BIN.slot(v), 0))
                elif home[v] != (ri, col):
                    links.append((0, ri, col, home[v][0], home[v][1], 0))
            elif r is BOUT and v in regs:
                links.append((3, ri, col, home[v][0], home[v][1], 0))
    links += cf.resolve(rows)
    build.poseidon_rows = ("PU", "PS") + tuple(cf.p2_names)
    build.cf_tables = cf.tables(rows, cf_rows)
    return rows, links, regs


def emit(rows, links, path, prefix="RC", guard="ZKW_RAM_CIRCUIT_SPEC_H",
         title=("/* GENERATED by tools/gen_ram_circuit.py — do not edit. Layout contract of the RAMPermutation trace",
                " * emitted by zkw_ram_synthesize (\"zkw trace v2\"). See the generator's docstring and DESIGN.md. */"),
         poseidon_rows=("PU", "PS"), shared_types=False, cf_tables=None):
    """Writes the spec header. `prefix` replaces RC in every macro so that several circuits can coexist; the
    rc_term / rc_constraint / rc_link types are declared by the RAM header only (shared_types=True skips them)."""
    out = []

    def w(line):
        out.append(line if prefix == "RC" else line.replace("RC_", prefix + "_"))

    for t in title:
        out.append(t)
    out.append(f"#ifndef {guard}\n#define {guard}\n#include <stdint.h>")
    w(f"#define RC_G {G}            /* general-purpose (copy-permutation) columns 0..{G - 1} */")
    w(f"#define RC_L {L}             /* lookup columns {G}..{G + L - 1}: every cell is in [0, 256) */")
    w(f"#define RC_MULT_COL {G + L}     /* multiplicity column of the 8-bit range-check table */")
    w(f"#define RC_COLS {G + L + 1}")
    if prefix == "RC":
        w(f"#define RC_HEAP_PAGE {HEAP_PAGE}")
    n_cyc = sum(1 for r in rows if r.per_cycle)
    w(f"#define RC_ROWS_PER_CYCLE {n_cyc}  /* region-major: row of (region r, cycle i) = r*RC_REGION_STRIDE(capacity) + i */")
    w("/* every region starts on a 64-row (512-byte) boundary so that a wave's 64 x 8-byte store of one column is one")
    w("   aligned 512-byte burst (misaligned regions cap the fills at 3.5 TB/s instead of 5.6, tools/ubench_fill.hip);")
    w("   rows [capacity, stride) of a region are zero */")
    w("#define RC_REGION_ALIGN 64")
    w("#define RC_REGION_STRIDE(capacity) ((((uint64_t)(capacity)) + RC_REGION_ALIGN - 1) & ~(uint64_t)(RC_REGION_ALIGN - 1))")
    w("#define RC_BOUNDARY_ROW(capacity) ((uint64_t)RC_ROWS_PER_CYCLE * RC_REGION_STRIDE(capacity))")
    w(f"#define RC_NUM_ROW_TYPES {len(rows)}")
    w("/* boundary rows sit right after the per-cycle regions */")
    for i, r in enumerate(rows):
        w(f"#define RC_ROW_{r.name} {i}")
    for i, r in enumerate(rows):
        if not r.per_cycle:
            w(f"#define RC_ROWOFF_{r.name} {i - n_cyc}  /* row = RC_BOUNDARY_ROW(capacity) + this */")
    w(f"#define RC_MIN_ROWS(capacity) (RC_BOUNDARY_ROW(capacity) + {len(rows) - n_cyc})")
    w("/* named slots: RC_<row>_<var> = column of that variable in rows of that type */")
    for r in rows:
        for v in r.slots + r.lookups:
            name = v.replace("p.", "P_").replace("g.", "G_").replace("x.", "X_").replace("y.", "Y_")
            w(f"#define RC_{r.name}_{name} {r.slot(v)}")
    if not shared_types:
        out.append("typedef struct { uint64_t coef; uint8_t nf; uint8_t f[6]; } rc_term;")
        out.append("typedef struct { uint16_t first_term; uint16_t n_terms; } rc_constraint;")
    terms, cons, row_first = [], [], []
    for r in rows:
        row_first.append(len(cons))
        for tl, why in r.constraints:
            cons.append((len(terms), len(tl), why, r.name))
            terms.extend(tl)
    row_first.append(len(cons))
    w(f"#define RC_NUM_TERMS {len(terms)}\n#define RC_NUM_CONSTRAINTS {len(cons)}")
    w("#define RC_TERMS_INIT { \\")
    for coef, fs in terms:
        f6 = fs + [0] * (6 - len(fs))
        w(f"  {{0x{coef:016x}ULL, {len(fs)}, {{{', '.join(map(str, f6))}}}}}, \\")
    w("}")
    w("#define RC_CONSTRAINTS_INIT { \\")
    for ft, nt, why, rn in cons:
        w(f"  {{{ft}, {nt}}}, /* {rn}: {why} */ \\")
    w("}")
    w("/* constraints of row type t are [RC_ROW_FIRST_CONSTRAINT[t], RC_ROW_FIRST_CONSTRAINT[t+1]) */")
    w(f"#define RC_ROW_FIRST_CONSTRAINT_INIT {{{', '.join(map(str, row_first))}}}")
    w(f"#define RC_ROW_NUM_SLOTS_INIT {{{', '.join(str(len(r.slots)) for r in rows)}}}")
    w(f"#define RC_ROW_NUM_LOOKUPS_INIT {{{', '.join(str(len(r.lookups)) for r in rows)}}}")
    w("/* rows whose first 130 slots are one flattened Poseidon2 permutation (checked by recomputation) */")
    w(f"#define RC_ROW_IS_POSEIDON_INIT {{{', '.join('1' if r.name in poseidon_rows else '0' for r in rows)}}}")
    w("/* copy links: cell (row_a, col_a) must equal cell (row_b, col_b):")
    w("   kind 0: both at the same cycle; kind 1: b at the previous cycle (for cycle 0: BND_IN column bin_col);")
    w("   kind 2: b = BND_IN (one row per instance); kind 3: a = BND_OUT, b at the LAST cycle */")
    if not shared_types:
        out.append("typedef struct { uint8_t kind, row_a, col_a, row_b, col_b, bin_col; } rc_link;")
    if cf_tables is not None:
        w("/* the closed-form section (boundary rows below PI; tools/gen_ram_circuit.py ClosedForm): row types [first, first + n) in fill order;")
        w("   a fill walks them: copies by the kind-5 links whose row_a is the row, the constant cells, the FREE cells (src 0 observable input /")
        w("   1 hidden FSM input / 2 hidden FSM output / 3 flag / 4 observable output, word idx), then the permutation of a Poseidon2 row */")
        w(f"#define RC_CF_FIRST_ROW_TYPE {cf_tables['first']}\n#define RC_CF_NUM_ROWS {cf_tables['n']}")
        if not shared_types:
            out.append("typedef struct { uint8_t row, col; uint64_t value; } rc_cf_const;")
            out.append("typedef struct { uint8_t row, col, src; uint16_t idx; } rc_cf_free;")
            out.append("typedef struct { uint8_t row, col, col_a, col_b; } rc_cf_product; /* cell = cell a * cell b of the same row, computed by a fill after the row's copies */")
            out.append("typedef struct { uint8_t row_a, col_a, row_b, col_b; } rc_cf_copy;   /* the kind-5 links whose row_a is a section row or the PI row */")
            out.append("/* a step of the parallel fill schedule: rows whose copies come from earlier steps (or the register rows); index ranges into the")
            out.append("   step-sorted tables: STEP_ROWS, COPIES, CONSTS, FREE, PRODUCTS, and the computed cells of the")
            out.append("   step-sorted tables BYTES (lookup cells b0..b0+3 = the bytes of a limb cell of the row) and LINEARS (cell = constant + sum coef * cell of the row: terms in LIN_TERMS) */")
            out.append("typedef struct { uint16_t row0, n_rows, copy0, n_copies, const0, n_consts, free0, n_free, prod0, n_prods, byte0, n_bytes, lin0, n_lins; } rc_cf_step;")
            out.append("typedef struct { uint8_t row, col_limb, col_b0, _pad; } rc_cf_bytes;")
            out.append("typedef struct { uint8_t row, col; uint16_t term0, n_terms; uint64_t constant; } rc_cf_linear;")
            out.append("typedef struct { uint64_t coef; uint8_t col; } rc_cf_lin_term;")
        w(f"#define RC_CF_NUM_CONSTS {len(cf_tables['consts'])}\n#define RC_CF_NUM_FREE {len(cf_tables['free'])}")
        w("#define RC_CF_CONSTS_INIT {" + ", ".join(f"{{{a}, {b}, {c}ULL}}" for a, b, c in cf_tables["consts"]) + "}")
        w("#define RC_CF_FREE_INIT {" + ", ".join(f"{{{a}, {b}, {c}, {d}}}" for a, b, c, d in cf_tables["free"]) + "}")
        prods = cf_tables.get("products", [])
        w(f"#define RC_CF_NUM_PRODUCTS {len(prods)}")
        w("#define RC_CF_PRODUCTS_INIT {" + ", ".join(f"{{{a}, {b}, {c}, {d}}}" for a, b, c, d in prods) + ("}" if prods else "{0, 0, 0, 0}}") + "  /* (one zero entry when there are none) */")
        for name, key, fmt in (("BYTES", "bytes", lambda e: f"{{{e[0]}, {e[1]}, {e[2]}, 0}}"), ("LINEARS", "linears", lambda e: f"{{{e[0]}, {e[1]}, {e[2]}, {e[3]}, {e[4]}ULL}}"),
                               ("LIN_TERMS", "lin_terms", lambda e: f"{{{e[1]}ULL, {e[0]}}}")):
            tab = cf_tables.get(key, [])
            w(f"#define RC_CF_NUM_{name} {len(tab)}")
            w(f"#define RC_CF_{name}_INIT {{" + (", ".join(fmt(e) for e in tab) if tab else "{0}") + "}  /* (one zero entry when there are none) */")
        w(f"#define RC_CF_NUM_COPIES {len(cf_tables['copies'])}")
        w("#define RC_CF_COPIES_INIT {" + ", ".join(f"{{{a}, {b}, {c}, {d}}}" for a, b, c, d in cf_tables["copies"]) + "}")
        w(f"#define RC_CF_NUM_STEPS {len(cf_tables['steps'])}")
        w("#define RC_CF_STEPS_INIT {" + ", ".join("{" + ", ".join(map(str, e)) + "}" for e in cf_tables["steps"]) + "}")
        w(f"#define RC_CF_NUM_STEP_ROWS {len(cf_tables['step_rows'])}")
        w("#define RC_CF_STEP_ROWS_INIT {" + ", ".join(map(str, cf_tables["step_rows"])) + "}")
    w(f"#define RC_NUM_LINKS {len(links)}")
    w("#define RC_LINKS_INIT { \\")
    for k in links:
        w(f"  {{{', '.join(map(str, k))}}}, \\")
    w("}")
    w("#endif")
    with open(path, "w") as f:
        f.write("\n".join(out) + "\n")
    return len(terms), len(cons)


if __name__ == "__main__":
    rows, links, regs = build()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "include", "zkw_ram_circuit_spec.h")
    nt, nc = emit(rows, links, path, poseidon_rows=build.poseidon_rows, cf_tables=build.cf_tables)
    for r in rows:
        print(f"{r.name:8s} slots {len(r.slots):3d} lookups {len(r.lookups):2d} constraints {len(r.constraints)}")
    print(f"{nt} terms, {nc} constraints, {len(links)} links -> {path}")
