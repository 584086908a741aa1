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

    rows = [PU, PS, A, B, Cc, D, BIN, BOUT, PI]

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
    return rows, links, regs


def emit(rows, links, path, prefix="RC", guard="ZKW_RAM_CIRCUIT_SPEC_H",
         title=("/* GENERATED by tools/gen_ram_circuit.py — do not edit. Layout contract of the RAMPermutation trace",
                " * emitted by zkw_ram_synthesize (\"zkw trace v2\"). See the generator's docstring and DESIGN.md. */"),
         poseidon_rows=("PU", "PS"), shared_types=False):
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
    nt, nc = emit(rows, links, path)
    for r in rows:
        print(f"{r.name:8s} slots {len(r.slots):3d} lookups {len(r.lookups):2d} constraints {len(r.constraints)}")
    print(f"{nt} terms, {nc} constraints, {len(links)} links -> {path}")
