#!/usr/bin/env python3
"""Generates include/zkw_decommit_sorter_circuit_spec.h — the declarative layout of the CodeDecommittmentsSorter
trace that libzkw emits ("zkw trace v2", circuit type 2), in the DSL of tools/gen_ram_circuit.py.

Geometry of the reference wrapper (circuit_definitions/.../base_layer/sort_code_decommits.rs:28-39): 130 copy
columns, 1x18 width-1 range-check lookups, Poseidon2 flattened gate, 2^20 rows, capacity 117 500; witness semantics
src/witness/individual_circuits/sort_decommit_requests.rs:20-420. The circuit body (`sort_and_deduplicate_code_
decommittments_entry_point`) lives in the absent crate era-zkevm_circuits, so gate placement is OUR design ("parity
unpinned" at the trace-layout level).

Statement, per cycle (7 rows, region-major): pop the unsorted and the sorted queue in lock step (PU, PS), multiply
both grand-product accumulators by the popped encodings (A), decompose the sorted request and check key >= previous
key on (hash, timestamp) with a 9-limb long subtraction (B, C), detect the start of a new hash group and, when the
previous group is complete, push its first (= fresh) request into the deduplicated queue (PR); D does the queue
bookkeeping. The open group is flushed by one more permutation outside the cycles (PF) when the instance completes —
that is why a non-final instance hands over the deduplicated queue WITHOUT its latest fresh request
(sort_decommit_requests.rs:150-158).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_ram_circuit as dsl  # noqa: E402

dsl.G, dsl.L = 130, 18
Row = dsl.Row


def build():
    PU, PS, PR = Row("PU"), Row("PS"), Row("PR")
    A, B, Cc, D = Row("A"), Row("B"), Row("C"), Row("D")
    BIN, BOUT, PF, PI = Row("BND_IN", False), Row("BND_OUT", False), Row("PF", False), Row("PI", False)

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
    es = [f"es{k}" for k in range(8)]
    ge = [f"ge{k}" for k in range(8)]  # encoding of the first request of the current hash group
    poseidon(PU, eu, [f"p.uh{8 + k}" for k in range(4)], [f"uo{k}" for k in range(12)])
    poseidon(PS, es, [f"p.sh{8 + k}" for k in range(4)], [f"so{k}" for k in range(12)])
    poseidon(PR, [f"p.ge{k}" for k in range(8)], [f"p.rh{8 + k}" for k in range(4)], [f"ro{k}" for k in range(12)])
    # es3..es6 are the hash limbs 3..6 themselves: their range checks ride in the sorted Poseidon row's lookup cells
    for k in (3, 4, 5, 6):
        PS.bytes_of(f"es{k}", f"h{k}")

    # ---------------- row A: grand-product contributions and accumulators (as in the RAM circuit)
    for v in eu + es:
        A.slot(v)
    for r in range(2):
        for k in range(1, 9):
            A.slot(f"g.c{r}_{k}")
    for r in range(2):
        for v in (f"lc{r}", f"p.lhs{r}", f"nl{r}", f"lhs{r}", f"rc{r}", f"p.rhs{r}", f"nr{r}", f"rhs{r}"):
            A.slot(v)
    for r in range(2):
        ch = [None] + [f"g.c{r}_{k}" for k in range(1, 9)]  # challenge 0 is the constant ONE (utils.rs:533)
        for side, enc, acc in (("l", eu, "lhs"), ("r", es, "rhs")):
            lc = f"{side}c{r}"
            A.c([(-1, [lc]), (1, [ch[8]]), (1, [enc[0]])] + [(1, [enc[k], ch[k]]) for k in range(1, 8)],
                f"{lc} = c8 + sum enc_k c_k")
            A.c([(1, [f"p.{acc}{r}", lc]), (-1, [f"n{side}{r}"])], f"n{side}{r} = acc*contribution")
            A.select("can_pop", f"n{side}{r}", f"p.{acc}{r}", f"{acc}{r}")
    # es7 = hash limb 7, and the first four limbs of the key subtraction ride in A's lookup cells
    A.bytes_of("es7", "h7")
    A.bytes_of("d0", "d0")
    A.bytes_of("d1", "d1")
    A.bytes_of("d2", "d2")
    A.boolean("bw0")
    A.boolean("bw1")
    A.boolean("bw2")
    # key = hash * 2^32 + timestamp as nine u32 limbs, least significant first: ts, h0, ..., h7
    A.c([(1, ["d0"]), (-1, ["ts"]), (1, ["p.ts"]), (-(1 << 32), ["bw0"])], "d0 = ts - p.ts + 2^32 bw0")
    A.c([(1, ["d1"]), (-1, ["h0"]), (1, ["p.h0"]), (1, ["bw0"]), (-(1 << 32), ["bw1"])], "d1")
    A.c([(1, ["d2"]), (-1, ["h1"]), (1, ["p.h1"]), (1, ["bw1"]), (-(1 << 32), ["bw2"])], "d2")

    # ---------------- row B: decomposition of the sorted request (encoding elements 0..2)
    for x in ("h0", "h1"):
        B.bytes_of(x, x)  # h2's range check rides in row D
    pb = [f"page_b{k}" for k in range(4)]
    tb = [f"ts_b{k}" for k in range(4)]
    for b in pb + tb:
        B.lookup(b)
    B.c([(1, ["page"])] + [(-(1 << (8 * k)), [pb[k]]) for k in range(4)], "page = sum bytes")
    B.c([(1, ["ts"])] + [(-(1 << (8 * k)), [tb[k]]) for k in range(4)], "ts = sum bytes")
    B.boolean("fresh")
    B.c([(1, ["es0"]), (-1, ["h0"]), (-(1 << 32), [pb[0]]), (-(1 << 40), [pb[1]]), (-(1 << 48), [pb[2]])], "es0")
    B.c([(1, ["es1"]), (-1, ["h1"]), (-(1 << 32), [pb[3]]), (-(1 << 40), [tb[0]]), (-(1 << 48), [tb[1]])], "es1")
    B.c([(1, ["es2"]), (-1, ["h2"]), (-(1 << 32), [tb[2]]), (-(1 << 40), [tb[3]]), (-(1 << 48), ["fresh"])], "es2")

    # ---------------- row C: key limbs 3..8, hash equality with the previous request, group logic
    for k in range(3, 7):
        Cc.bytes_of(f"d{k}", f"d{k}")
    for k in range(3, 9):
        Cc.boolean(f"bw{k}")
    hl = ["h0", "h1", "h2", "es3", "es4", "es5", "es6", "es7"]  # hash limb k as a cell name
    for k in range(3, 9):
        Cc.c([(1, [f"d{k}"]), (-1, [hl[k - 1]]), (1, [f"p.{hl[k - 1]}"]), (1, [f"bw{k - 1}"]), (-(1 << 32), [f"bw{k}"])], f"d{k}")
    Cc.c([(1, ["can_pop", "p.gvalid", "bw8"])], "sorted: key >= previous key (when there is a previous request)")
    for k in range(8):
        Cc.is_zero([(1, hl[k]), (-1, f"p.{hl[k]}")], f"w_e{k}", f"z_e{k}", f"hash limb {k} == previous")
    Cc.c([(1, ["z_e0", "z_e1", "z_e2", "z_e3"]), (-1, ["same_a"])], "same_a")
    Cc.c([(1, ["same_a", "z_e4", "z_e5", "z_e6", "z_e7"]), (-1, ["same_hash"])], "same_hash")
    Cc.c([(1, ["can_pop"]), (-1, ["can_pop", "same_hash", "p.gvalid"]), (-1, ["new_group"])],
         "new_group = can_pop & !(same hash as a valid previous request)")
    Cc.c([(1, ["new_group", "p.gvalid"]), (-1, ["push"])], "push = new_group & previous group exists")
    Cc.c([(1, ["can_pop", "fresh"]), (-1, ["new_group"])], "a request is fresh iff it opens its hash group")
    Cc.c([(1, ["can_pop", "same_hash", "p.gvalid", "page"]), (-1, ["can_pop", "same_hash", "p.gvalid", "p.page"])],
         "same hash => same page (sort_decommit_requests.rs:103-110)")
    Cc.c([(1, ["gvalid"]), (-1, ["p.gvalid"]), (-1, ["can_pop"]), (1, ["can_pop", "p.gvalid"])], "gvalid = p.gvalid | can_pop")

    # ---------------- row D: the last two key limbs' range checks, queue bookkeeping, group registers
    D.bytes_of("d7", "d7")
    D.bytes_of("d8", "d8")
    D.bytes_of("h2", "h2")
    D.is_zero([(1, "p.len_u")], "w_lu", "z_lu", "len_u == 0")
    D.is_zero([(1, "p.len_s")], "w_ls", "z_ls", "len_s == 0")
    D.c([(1, ["z_lu"]), (-1, ["z_ls"])], "both queues empty together")
    D.c([(1, ["can_pop"]), (1, ["z_lu"]), (-1, [])], "can_pop = 1 - empty")
    D.c([(1, ["len_u"]), (-1, ["p.len_u"]), (1, ["can_pop"])], "len_u = p.len_u - can_pop")
    D.c([(1, ["len_s"]), (-1, ["p.len_s"]), (1, ["can_pop"])], "len_s = p.len_s - can_pop")
    D.c([(1, ["len_r"]), (-1, ["p.len_r"]), (-1, ["push"])], "len_r = p.len_r + push")
    for k in range(8):
        D.select("new_group", es[k], f"p.ge{k}", ge[k])

    # queue heads: three rows' worth of selects do not fit one row -> the unsorted / sorted heads live in D,
    # the deduplicated queue's in C's spare slots
    for q, o in (("uh", "uo"), ("sh", "so")):
        for k in range(12):
            D.select("can_pop", f"{o}{k}", f"p.{q}{k}", f"{q}{k}")
    for k in range(12):
        Cc.select("push", f"ro{k}", f"p.rh{k}", f"rh{k}")

    # ---------------- boundary rows
    regs = ([f"uh{k}" for k in range(12)] + [f"sh{k}" for k in range(12)] + [f"rh{k}" for k in range(12)] +
            ["len_u", "len_s", "len_r", "lhs0", "lhs1", "rhs0", "rhs1", "ts", "page", "h0", "h1", "h2", "es3", "es4", "es5", "es6",
             "es7", "gvalid"] + ge)
    for v in regs:
        BIN.slot(v)
    for r in range(2):
        for k in range(1, 9):
            BIN.slot(f"g.c{r}_{k}")
    for v in regs:
        BOUT.slot(v)
    for q in ("u", "s"):
        for k in range(12):
            BOUT.slot(f"tail_{q}{k}")
    BOUT.boolean("completion")
    BOUT.is_zero([(1, "len_u")], "w_end", "z_end", "queue exhausted")
    for q, h in (("u", "uh"), ("s", "sh")):
        for k in range(12):
            BOUT.c([(1, ["z_end", f"{h}{k}"]), (-1, ["z_end", f"tail_{q}{k}"])], f"empty queue: head == tail ({q}{k})")
    BOUT.c([(1, ["completion"]), (-1, ["completion", "z_end"])], "completion => queues exhausted")
    for r in range(2):
        BOUT.c([(1, ["completion", f"lhs{r}"]), (-1, ["completion", f"rhs{r}"])], f"completion => lhs{r} == rhs{r}")
    # the flush: on completion the open group's request goes into the deduplicated queue
    BOUT.c([(1, ["completion", "gvalid"]), (-1, ["flush"])], "flush = completion & a group is open")
    for k in range(12):
        BOUT.slot(f"fo{k}")
    for k in range(12):
        BOUT.select("flush", f"fo{k}", f"rh{k}", f"final_rh{k}")
    BOUT.c([(1, ["final_len_r"]), (-1, ["len_r"]), (-1, ["flush"])], "final_len_r = len_r + flush")
    poseidon(PF, [f"x.ge{k}" for k in range(8)], [f"x.rh{8 + k}" for k in range(4)], [f"x.fo{k}" for k in range(12)])
    for k in range(4):
        PI.slot(f"pi{k}")

    # ---------------- closed-form section (gen_ram_circuit.ClosedForm): what the reference's circuit derives in-trace
    cf = dsl.ClosedForm()
    SRC = dsl.ClosedForm
    # observable input (CodeDecommittmentsDeduplicatorInputData: initial_queue_state, sorted_queue_initial_state), the block's first instance's
    OI = cf.sponge("OI", [None] * 50, free_src=SRC.SRC_OBS_IN)
    oi = lambda w: cf.word_cell(OI, w)  # noqa: E731
    # hidden FSM input (CodeDecommittmentsDeduplicatorFSMInputOutput, sort_decommit_requests.rs:396-411): queue states 0 / 25 / 50,
    # lhs 75, rhs 77, previous_packed_key 79 (timestamp, hash words), previous_record 88 (hash 8, page, is_fresh, timestamp), first_encountered_timestamp 99
    FI = cf.sponge("FI", [None] * 100, free_src=SRC.SRC_FSM_IN)
    fi = lambda w: cf.word_cell(FI, w)  # noqa: E731
    # the open group's first request as the registers hold it — the ENCODING of (previous_record.hash, page, first_encountered_timestamp,
    # fresh) (decommit query encoding: hash words 0..2 carry the page / timestamp bytes and the fresh flag) — from the FSM's words
    GIN = Row("GIN", False)
    for k in range(8):
        GIN.slot(f"gh{k}")
        cf.copy(GIN, f"gh{k}", *fi(88 + k))
    GIN.slot("gpage"), GIN.slot("gfts")
    cf.copy(GIN, "gpage", *fi(96))
    cf.copy(GIN, "gfts", *fi(99))
    pb = GIN.bytes_of("gpage", "gpage")
    tb = GIN.bytes_of("gfts", "gfts")
    GIN.c([(1, ["gge0"]), (-1, ["gh0"]), (-(1 << 32), [pb[0]]), (-(1 << 40), [pb[1]]), (-(1 << 48), [pb[2]])], "ge0 of the open group")
    GIN.c([(1, ["gge1"]), (-1, ["gh1"]), (-(1 << 32), [pb[3]]), (-(1 << 40), [tb[0]]), (-(1 << 48), [tb[1]])], "ge1")
    GIN.c([(1, ["gge2"]), (-1, ["gh2"]), (-(1 << 32), [tb[2]]), (-(1 << 40), [tb[3]]), (-(1 << 48), [])], "ge2 (fresh)")
    cf.rows.append(GIN)
    # start-flag selection: registers at cycle -1 = start ? (observable input's queues, the empty result queue, accumulators ONE, no
    # previous key, no open group) : hidden FSM input
    SEL = dsl.Selections(cf, "SEL")
    for qi, q in enumerate(("u", "s")):
        for k in range(12):
            SEL.sel3(oi(25 * qi + k), fi(25 * qi + k), (BIN, f"{q}h{k}"))
            SEL.sel3(oi(25 * qi + 12 + k), fi(25 * qi + 12 + k), (BOUT, f"tail_{q}{k}"))
        SEL.sel3(oi(25 * qi + 24), fi(25 * qi + 24), (BIN, f"len_{q}"))
    for k in range(12):
        SEL.sel2(0, fi(50 + 12 + k), (BIN, f"rh{k}"))  # the deduplicated queue starts empty (the reference's callers hand in a fresh simulator)
    SEL.sel2(0, fi(74), (BIN, "len_r"))
    for r in range(2):
        SEL.sel2(1, fi(75 + r), (BIN, f"lhs{r}"))
        SEL.sel2(1, fi(77 + r), (BIN, f"rhs{r}"))
    key_regs = ["ts", "h0", "h1", "h2", "es3", "es4", "es5", "es6", "es7"]  # previous_packed_key: timestamp, then the hash words
    for k, v in enumerate(key_regs):
        SEL.sel2(0, fi(79 + k), (BIN, v))
    SEL.sel2(0, fi(96), (BIN, "page"))
    SEL.not_flag((BIN, "gvalid"))
    for k in range(8):
        SEL.sel2(0, (GIN, f"gge{k}" if k < 3 else f"gh{k}"), (BIN, f"ge{k}"))
    # hidden FSM output: the registers after the last cycle (the deduplicated queue WITHOUT the open group's request unless the instance
    # completes: final_rh / final_len_r, the builder's snapshot rule sort_decommit_requests.rs:150-158)
    q25 = lambda q: [(BOUT, f"{q}h{k}") for k in range(12)] + [(BOUT, f"tail_{q}{k}") for k in range(12)] + [(BOUT, f"len_{q}")]  # noqa: E731
    fo_words = (q25("u") + q25("s") + [("const", 0)] * 12 + [(BOUT, f"final_rh{k}") for k in range(12)] + [(BOUT, "final_len_r")] +
                [(BOUT, f"lhs{r}") for r in range(2)] + [(BOUT, f"rhs{r}") for r in range(2)] + [(BOUT, v) for v in key_regs] +
                [(BOUT, v) for v in key_regs[1:]] + [(BOUT, "page"), None, (BOUT, "ts"), None])  # is_fresh / first_encountered_timestamp: only committed
    assert len(fo_words) == 100
    # the handed-over open group: its encoding re-derived from the FSM output's words (hash = the key registers, page, first_encountered_timestamp,
    # fresh) is the register ge unless the instance completes (the builders hand over placeholders then, :174-181)
    GOUT = Row("GOUT", False)
    for v in ("gop", "gof", "goh0", "goh1", "goh2"):
        GOUT.slot(v)
    cf.copy(GOUT, "gop", BOUT, "page")
    cf.free_cell(GOUT, "gof", SRC.SRC_FSM_OUT, 99)
    for k in range(3):
        cf.copy(GOUT, f"goh{k}", BOUT, f"h{k}")
    opb = cf.bytes_of(GOUT, "gop", "gop")
    otb = cf.bytes_of(GOUT, "gof", "gof")
    cf.linear(GOUT, "goge0", [(1, "goh0"), (1 << 32, opb[0]), (1 << 40, opb[1]), (1 << 48, opb[2])], why="ge0 of the handed-over group")
    cf.linear(GOUT, "goge1", [(1, "goh1"), (1 << 32, opb[3]), (1 << 40, otb[0]), (1 << 48, otb[1])], why="ge1")
    cf.linear(GOUT, "goge2", [(1, "goh2"), (1 << 32, otb[2]), (1 << 40, otb[3])], const=1 << 48, why="ge2 (fresh)")
    cf.rows.append(GOUT)
    fo_words[99] = (GOUT, "gof")
    OSEL = dsl.Selections(cf, "OSEL", flag_cell=(BOUT, "completion"))
    for k in range(3):
        OSEL.eq_unless_flag((GOUT, f"goge{k}"), (BOUT, f"ge{k}"))
    for k in range(3, 8):
        OSEL.eq_unless_flag((BOUT, f"es{k}"), (BOUT, f"ge{k}"))
    FO = cf.sponge("FO", fo_words, free_src=SRC.SRC_FSM_OUT)
    # observable output (final_queue_state): completion ? the deduplicated queue after the flush : the placeholder (zeros), :340-372
    oo_words = [("const", 0)] * 12 + [OSEL.gate((BOUT, f"final_rh{k}")) for k in range(12)] + [OSEL.gate((BOUT, "final_len_r"))]
    cf.rows += OSEL.rows
    OO = cf.sponge("OO", oo_words)
    # Fiat-Shamir challenges over the observable input's queue tails and lengths (sort_decommit_requests.rs:205-215)
    fs_words = [oi(12 + k) for k in range(12)] + [oi(24)] + [oi(25 + 12 + k) for k in range(12)] + [oi(49)]
    CH = cf.sponge("CH", fs_words, squeeze=1)
    for r in range(2):
        for k in range(1, 9):
            cf.copy(BIN, f"g.c{r}_{k}", CH[3 + r], f"{CH[3 + r].name}_o{k - 1}")
    last = lambda rows_: rows_[-1]  # noqa: E731
    cp_words = [SEL.flag(), (BOUT, "completion")]
    for sp in (OI, OO, FI, FO):
        cp_words += [(last(sp), f"{last(sp).name}_o{k}") for k in range(4)]
    CP = cf.sponge("CP", cp_words)
    for k in range(4):
        cf.copy(PI, f"pi{k}", last(CP), f"{last(CP).name}_o{k}")
    # fill order: a row's copies come from rows before it (or from the register rows)
    pos = cf.rows.index(GIN) + 1
    cf.rows[pos:pos] = SEL.rows
    build.cf = cf

    rows = [PU, PS, PR, A, B, Cc, D, BIN, BOUT, PF, PI] + cf.rows
    # the previous request's key is simply the previous cycle's cells (a padding cycle holds zeros, which is also what
    # the builder hands over after a partial last chunk, sort_decommit_requests.rs:174-181)
    return rows, regs


def links_of(rows, regs):
    """Copy links (see gen_ram_circuit.build): p.x -> x at the previous cycle, g.x -> BND_IN, x.y in PF -> y in BND_OUT,
    BND_OUT registers -> their home at the last cycle, everything else -> its home in the same cycle."""
    BIN = next(r for r in rows if r.name == "BND_IN")
    BOUT = next(r for r in rows if r.name == "BND_OUT")
    home = {}
    for ri, r in enumerate(rows):
        if not r.per_cycle:
            continue
        for v in r.slots + r.lookups:
            if not v.startswith(("p.", "g.")) and v not in home:
                home[v] = (ri, r.slot(v))
    links = []
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
            elif v.startswith("x."):
                links.append((4, ri, col, rows.index(BOUT), BOUT.slot(v[2:]), 0))
    return links + build.cf.resolve(rows)


if __name__ == "__main__":
    rows, regs = build()
    links = links_of(rows, regs)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "include", "zkw_decommit_sorter_circuit_spec.h")
    nt, nc = dsl.emit(rows, links, path, prefix="DS", guard="ZKW_DECOMMIT_SORTER_CIRCUIT_SPEC_H",
                      title=("/* GENERATED by tools/gen_decommit_sorter_circuit.py — do not edit. Layout contract of the",
                             " * CodeDecommittmentsSorter trace emitted by zkw_decommit_sorter_synthesize (\"zkw trace v2\"). */",
                             "#include \"zkw_ram_circuit_spec.h\" /* rc_term, rc_constraint, rc_link */"),
                      poseidon_rows=("PU", "PS", "PR", "PF") + tuple(build.cf.p2_names), shared_types=True,
                      cf_tables=build.cf.tables(rows, build.cf.rows))
    # scatter lists for the oracle's fill: one struct field per distinct variable, one X-entry per cell
    names = []
    for r in rows:
        for v in r.slots + r.lookups:
            base = v.split(".", 1)[1] if v[:2] in ("p.", "g.", "x.") else v
            if base not in names:
                names.append(base)
    lines = ["", "/* ---- scatter lists (used by the oracle's fill): DS_VARS(X) lists every distinct variable once;",
             "   DS_FILL_<row>(XC, XP, XG, XX) lists the cells of a row: XC(col, v) current value, XP(col, v) value of the",
             "   previous cycle, XG(col, v) per-instance global, XX(col, v) value in BND_OUT (the flush permutation). */",
             "#define DS_VARS(X) " + " ".join(f"X({n})" for n in names)]
    for r in rows:
        lines.append(f"#define DS_NSLOTS_{r.name} {len(r.slots)}")
        lines.append(f"#define DS_NLOOK_{r.name} {len(r.lookups)}")
    lines.append(f"#define DS_LOOKUPS_PER_CYCLE {sum(len(r.lookups) for r in rows if r.per_cycle)}")
    for r in rows:
        ent = []
        for v in r.slots + r.lookups:
            kind = {"p.": "XP", "g.": "XG", "x.": "XX"}.get(v[:2], "XC")
            base = v.split(".", 1)[1] if kind != "XC" else v
            ent.append(f"{kind}({r.slot(v)}, {base})")
        lines.append(f"#define DS_FILL_{r.name}(XC, XP, XG, XX) " + " ".join(ent))
    txt = open(path).read()
    txt = txt.replace("\n#endif\n", "\n" + "\n".join(lines) + "\n#endif\n")
    open(path, "w").write(txt)
    for r in rows:
        print(f"{r.name:8s} slots {len(r.slots):3d} lookups {len(r.lookups):2d} constraints {len(r.constraints)}")
    print(f"{nt} terms, {nc} constraints, {len(links)} links -> {path}")
