#!/usr/bin/env python3
"""Generates include/zkw_storage_sorter_circuit_spec.h — the declarative layout of the StorageSorter trace that libzkw
emits ("zkw trace v2", circuit type 9), in the DSL of tools/gen_ram_circuit.py.

Geometry of the reference wrapper (circuit_definitions/.../base_layer/storage_sort_dedup.rs:29-40): 132 copy columns,
1x16 width-1 range-check lookups, Poseidon2 flattened gate, 2^20 rows, capacity 46 921; witness semantics
src/witness/individual_circuits/storage_sort_dedup.rs:12-703. The circuit body lives in the absent crate
era-zkevm_circuits, so gate placement is OUR design ("parity unpinned" at the trace-layout level).

Statement, per cycle (22 rows, region-major — all the 2^20 rows allow at this capacity): pop the unsorted and the sorted
log queue in lock step (U1-U3, S1-S3; a 4-wide queue hashes enc(20) || tail(4) in three permutations,
circuit_encodings/src/lib.rs:179-221), multiply both grand-product accumulators (A, W = 20; the unsorted side enters
with the extended timestamp = its queue position added to word 19, storage_sort_dedup.rs:128-143), split the sorted
record's words 0..17 into a low u32 and a 24-bit rider (X0-X7, K, C1: 126 + 12 lookups). The riders ARE the sorting
key: key bytes then address bytes, little-endian, three per word (log_query.rs:118-196), so "sorted by (address, key,
extended timestamp)" is a lexicographic comparison of the 18 riders from the top and then of the timestamps (row K:
18 equality tests, prefix products, ONE 32-bit range check of the first difference minus one). Row C1 decides whether
the previous cell's net record is pushed into the result queue (depth > 0 or an explicit read at depth 0,
storage_sort_dedup.rs:394-457) and assembles its encoding from the cell registers (R1-R3 hash it); row C2 is the cell
state machine (base / current value, rollback depth, read-at-depth-zero flag, :339-534) incl. the value-consistency
checks of sort_storage_access.rs:91-203; row Q does the queue bookkeeping and keeps the key registers. The last
cell is flushed by three more permutations outside the cycles (F1-F3) when the instance completes.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_ram_circuit as dsl  # noqa: E402
import gen_events_sorter_circuit as esg  # noqa: E402  (poseidon, queue_rows, links_of, emit_scatter)

Row = dsl.Row
S32 = 1 << 32


def eq8(row, a, b, tag):
    """eqv = [a_k == b_k for all 8 limbs] (two product slots keep every term at <= 6 variables)"""
    for k in range(8):
        row.is_zero([(1, a[k]), (-1, b[k])], f"wq{k}", f"eq{k}", f"{tag} limb {k} equal")
    row.c([(1, ["eq0", "eq1", "eq2", "eq3"]), (-1, ["q1"])], "q1 = limbs 0..3 equal")
    row.c([(1, ["q1", "eq4", "eq5", "eq6", "eq7"]), (-1, ["eqv"])], "eqv = all limbs equal")


def pushed_words(row, out, base, wsel, kc, ksh):
    """encoding of L::create_partially_filled_from_fields(cell, read = base, written = wsel, rw) (log_query.rs:49-72):
    out[k] for k = 0..17 and 19; word 18 (the rw flag) is constrained by the caller"""
    for k in range(8):
        row.c([(1, [out[k]]), (-1, [base[k]]), (-S32, [kc[k]])], f"{out[k]} = base limb + key rider")
        row.c([(1, [out[8 + k]]), (-1, [wsel[k]]), (-S32, [kc[8 + k]])], f"{out[8 + k]} = written limb + rider")
    row.c([(1, [out[16]]), (-S32, [kc[16]])], f"{out[16]} = rider (timestamp 0)")
    row.c([(1, [out[17]]), (-S32, [kc[17]]), (-(1 << 48), [ksh])], f"{out[17]} = address byte 19, shard (tx 0, aux 0)")
    row.c([(1, [out[19]])], f"{out[19]} = 0")


def build():
    dsl.G, dsl.L = 132, 16
    U = [Row("U1"), Row("U2"), Row("U3")]
    S = [Row("S1"), Row("S2"), Row("S3")]
    R = [Row("R1"), Row("R2"), Row("R3")]
    A = Row("A")
    X = [Row(f"X{k}") for k in range(8)]
    K, C1, C2, Q = Row("K"), Row("C1"), Row("C2"), Row("Q")
    BIN, BOUT, PI = Row("BND_IN", False), Row("BND_OUT", False), Row("PI", False)
    F = [Row("F1", False), Row("F2", False), Row("F3", False)]

    eu = [f"eu{k}" for k in range(20)]
    es = [f"es{k}" for k in range(20)]
    pw = [f"pw{k}" for k in range(20)]
    lo = [f"lo{k}" for k in range(17)]
    c = [f"c{k}" for k in range(18)]
    kc = [f"kc{k}" for k in range(18)]
    esg.queue_rows(U, eu, [f"p.uh{k}" for k in range(4)], "u", "uz")
    esg.queue_rows(S, es, [f"p.sh{k}" for k in range(4)], "s", "sz")
    esg.queue_rows(R, pw, [f"p.rh{k}" for k in range(4)], "r", "rz")

    # ---------------- row A: grand products, W = 20 (utils.rs:554-697; challenge 0 is the constant ONE, 20 is additive)
    for v in eu + es:
        A.slot(v)
    for r in range(2):
        for k in range(1, 21):
            A.slot(f"g.c{r}_{k}")
    for r in range(2):
        for v in (f"lc{r}", f"p.lhs{r}", f"nl{r}", f"lhs{r}", f"rc{r}", f"p.rhs{r}", f"nr{r}", f"rhs{r}"):
            A.slot(v)
    for r in range(2):
        ch = [None] + [f"g.c{r}_{k}" for k in range(1, 21)]
        for side, enc, acc in (("l", eu, "lhs"), ("r", es, "rhs")):
            lc = f"{side}c{r}"
            ext = [(256, ["can_pop", "p.cidx", ch[19]])] if side == "l" else []  # extended timestamp = queue position
            A.c([(-1, [lc]), (1, [ch[20]]), (1, [enc[0]])] + [(1, [enc[k], ch[k]]) for k in range(1, 20)] + ext,
                f"{lc} = c20 + sum enc_k c_k")
            A.c([(1, [f"p.{acc}{r}", lc]), (-1, [f"n{side}{r}"])], f"n{side}{r} = acc*contribution")
            A.select("can_pop", f"n{side}{r}", f"p.{acc}{r}", f"{acc}{r}")

    # ---------------- the split of the sorted record: es_k = lo_k + 2^32 c_k, lo_k four bytes, c_k three bytes
    def split(row, k):
        lb = [f"lo{k}_b{j}" for j in range(4)]
        cb = [f"c{k}_b{j}" for j in range(3)]
        for b in lb + cb:
            row.lookup(b)
        row.c([(1, [lo[k]])] + [(-(1 << (8 * j)), [lb[j]]) for j in range(4)], f"lo{k} = sum bytes")
        row.c([(1, [c[k]])] + [(-(1 << (8 * j)), [cb[j]]) for j in range(3)], f"c{k} = sum bytes")
        row.c([(1, [es[k]]), (-1, [lo[k]]), (-S32, [c[k]])], f"es{k} = lo{k} + rider")
    for k in range(16):
        split(X[k // 2], k)

    # ---------------- row K: word 16, the extended timestamp, the order of (riders from the top, timestamp)
    split(K, 16)
    tb = [f"ts_b{j}" for j in range(4)]
    for b in tb:
        K.lookup(b)
    K.c([(1, ["ts"])] + [(-(1 << (8 * j)), [tb[j]]) for j in range(4)], "ts = sum bytes")
    K.boolean("rb")
    K.c([(1, [es[19]]), (-1, ["rb"]), (-256, ["ts"])], "es19 = rollback + 2^8 extended timestamp")
    for k in range(18):
        K.is_zero([(1, c[k]), (-1, f"p.kc{k}")], f"wk{k}", f"ek{k}", f"rider {k} == previous")
    # pe_k = riders above k all equal; pe17 = 1, pe16 = ek17
    K.c([(1, ["ek17", "ek16"]), (-1, ["pe15"])], "pe15")
    for k in range(14, -1, -1):
        K.c([(1, [f"pe{k + 1}", f"ek{k + 1}"]), (-1, [f"pe{k}"])], f"pe{k}")
    K.c([(1, ["pe0", "ek0"]), (-1, ["keq"])], "keq = same cell as the previous record")
    K.c([(-1, ["diff"]), (1, [c[17]]), (-1, ["p.kc17"]), (1, ["ek17", c[16]]), (-1, ["ek17", "p.kc16"])] +
        [t for k in range(16) for t in ((1, [f"pe{k}", c[k]]), (-1, [f"pe{k}", f"p.kc{k}"]))] +
        [(1, ["keq", "ts"]), (-1, ["keq", "p.kts"])], "diff = first difference from the top (riders, then timestamp)")
    db = [f"d_b{j}" for j in range(4)]
    for b in db:
        K.lookup(b)
    K.c([(1, ["can_pop", "p.valid", "diff"]), (-1, ["can_pop", "p.valid"])] + [(-(1 << (8 * j)), ["can_pop", "p.valid", db[j]]) for j in range(4)],
        "sorted: 1 <= diff < 2^32 + 1")

    # ---------------- row C1: word 17/18, does the previous cell emit a record, and which
    xb = [f"tx_b{j}" for j in range(4)]
    for b in xb + [c[17], "aux", "shard"]:
        C1.lookup(b)
    C1.c([(1, [es[17]])] + [(-(1 << (8 * j)), [xb[j]]) for j in range(4)] + [(-S32, [c[17]]), (-(1 << 40), ["aux"]), (-(1 << 48), ["shard"])], "es17")
    C1.boolean("rw")
    C1.boolean("sv")
    C1.c([(1, [es[18]]), (-1, ["rw"]), (-2, ["sv"])], "es18 = rw + 2 is_service")
    pbase = [f"p.base{k}" for k in range(8)]
    pcur = [f"p.cur{k}" for k in range(8)]
    eq8(C1, pcur, pbase, "current vs base")
    C1.is_zero([(1, "p.depth")], "w_d", "z_d", "depth == 0")
    C1.c([(1, ["em"]), (-1, []), (1, ["z_d"]), (-1, ["z_d", "p.has"])], "em = depth > 0 | read at depth 0")
    C1.c([(1, ["can_pop", "p.valid"]), (-1, ["can_pop", "p.valid", "keq"]), (-1, ["nkey"])], "nkey = a new cell follows a cell")
    C1.c([(1, ["nkey", "em"]), (-1, ["push"])], "push")
    wsel = [f"wsel{k}" for k in range(8)]
    for k in range(8):
        C1.select("z_d", pbase[k], pcur[k], wsel[k])
    pushed_words(C1, pw, pbase, wsel, [f"p.kc{k}" for k in range(18)], "p.ksh")
    C1.c([(1, [pw[18]]), (-1, []), (1, ["z_d"]), (1, ["eqv"]), (-1, ["z_d", "eqv"])], "pw18 = rw flag = depth > 0 & value changed")

    # ---------------- row C2: the cell state machine
    rv, wv = lo[0:8], lo[8:16]
    C2.c([(1, ["p.valid", "keq"]), (-1, ["same"])], "same")
    C2.c([(1, ["can_pop", "same"]), (-1, ["sm"])], "sm = a record of the same cell")
    C2.c([(1, ["can_pop"]), (-1, ["sm"]), (-1, ["nc"])], "nc = a record that opens a cell")
    C2.c([(1, ["rw"]), (-1, ["rw", "rb"]), (-1, ["wr"])], "wr = forward write")
    C2.c([(1, ["rw", "rb"]), (-1, ["rbk"])], "rbk = rollback of a write")
    C2.c([(1, ["nc", "rbk"])], "a cell does not open with a rollback")
    C2.c([(1, ["sm", "rbk", "z_d"])], "no rollback at depth 0")
    for k in range(8):
        C2.select("wr", wv[k], rv[k], f"t{k}")
        C2.select("can_pop", f"t{k}", pcur[k], f"cur{k}")
        C2.select("nc", rv[k], pbase[k], f"base{k}")
        C2.c([(1, ["sm", rv[k]]), (-1, ["sm", pcur[k]]), (-1, ["sm", "rbk", rv[k]]), (1, ["sm", "rbk", pcur[k]])],
             f"read / forward write: read value == current value ({k})")
        C2.c([(1, ["sm", "rbk", wv[k]]), (-1, ["sm", "rbk", pcur[k]])], f"rollback: written value == current value ({k})")
    C2.c([(-1, ["depth"]), (1, ["nc", "rw"]), (1, ["p.depth"]), (-1, ["nc", "p.depth"]), (1, ["sm", "rw"]), (-2, ["sm", "rbk"])],
         "depth")
    C2.c([(1, ["sm", "z_d"]), (-1, ["sm", "z_d", "rw"]), (-1, ["u"])], "u = read at depth 0 inside the cell")
    C2.c([(-1, ["has"]), (1, ["nc"]), (-1, ["nc", "rw"]), (1, ["p.has"]), (-1, ["nc", "p.has"]), (1, ["u"]), (-1, ["u", "p.has"])], "has")
    C2.c([(1, ["valid"]), (-1, ["p.valid"]), (-1, ["can_pop"]), (1, ["can_pop", "p.valid"])], "valid = p.valid | can_pop")

    # ---------------- row Q: queue bookkeeping, key registers
    Q.is_zero([(1, "p.len_u")], "w_lu", "z_lu", "len_u == 0")
    Q.is_zero([(1, "p.len_s")], "w_ls", "z_ls", "len_s == 0")
    Q.c([(1, ["z_lu"]), (-1, ["z_ls"])], "both queues empty together")
    Q.c([(1, ["can_pop"]), (1, ["z_lu"]), (-1, [])], "can_pop = 1 - empty")
    Q.c([(1, ["len_u"]), (-1, ["p.len_u"]), (1, ["can_pop"])], "len_u = p.len_u - can_pop")
    Q.c([(1, ["len_s"]), (-1, ["p.len_s"]), (1, ["can_pop"])], "len_s = p.len_s - can_pop")
    for q, o in (("uh", "u3o"), ("sh", "s3o")):
        for k in range(4):
            Q.select("can_pop", f"{o}{k}", f"p.{q}{k}", f"{q}{k}")
    for k in range(4):
        Q.select("push", f"r3o{k}", f"p.rh{k}", f"rh{k}")
    Q.c([(1, ["len_r"]), (-1, ["p.len_r"]), (-1, ["push"])], "len_r = p.len_r + push")
    for k in range(18):
        Q.select("can_pop", c[k], f"p.kc{k}", kc[k])
    Q.select("can_pop", "shard", "p.ksh", "ksh")
    Q.select("can_pop", "ts", "p.kts", "kts")
    Q.c([(1, ["cidx"]), (-1, ["p.cidx"]), (-1, [])], "cidx = p.cidx + 1 (storage_sort_dedup.rs:597)")

    # ---------------- boundary rows
    regs = ([f"uh{k}" for k in range(4)] + [f"sh{k}" for k in range(4)] + [f"rh{k}" for k in range(4)] +
            ["len_u", "len_s", "len_r", "lhs0", "lhs1", "rhs0", "rhs1"] + kc + ["ksh", "kts", "cidx", "valid", "depth", "has"] +
            [f"base{k}" for k in range(8)] + [f"cur{k}" for k in range(8)])
    for v in regs:
        BIN.slot(v)
    for r in range(2):
        for k in range(1, 21):
            BIN.slot(f"g.c{r}_{k}")
    for v in regs:
        BOUT.slot(v)
    for q in ("u", "s"):
        for k in range(4):
            BOUT.slot(f"tail_{q}{k}")
    BOUT.boolean("completion")
    BOUT.is_zero([(1, "len_u")], "w_end", "z_end", "queue exhausted")
    for q, h in (("u", "uh"), ("s", "sh")):
        for k in range(4):
            BOUT.c([(1, ["z_end", f"{h}{k}"]), (-1, ["z_end", f"tail_{q}{k}"])], f"empty queue: head == tail ({q}{k})")
    BOUT.c([(1, ["completion"]), (-1, ["completion", "z_end"])], "completion => queues exhausted")
    for r in range(2):
        BOUT.c([(1, ["completion", f"lhs{r}"]), (-1, ["completion", f"rhs{r}"])], f"completion => lhs{r} == rhs{r}")
    # the flush of the last cell: the same decision and the same record as row C1, over the final registers
    fbase = [f"base{k}" for k in range(8)]
    fcur = [f"cur{k}" for k in range(8)]
    eq8(BOUT, fcur, fbase, "final current vs base")
    BOUT.is_zero([(1, "depth")], "w_d", "z_d", "final depth == 0")
    BOUT.c([(1, ["em"]), (-1, []), (1, ["z_d"]), (-1, ["z_d", "has"])], "em")
    BOUT.c([(1, ["completion", "valid", "em"]), (-1, ["flush"])], "flush = completion & a cell is open & it emits")
    fsel = [f"wsel{k}" for k in range(8)]
    for k in range(8):
        BOUT.select("z_d", fbase[k], fcur[k], fsel[k])
    fw = [f"fw{k}" for k in range(20)]
    pushed_words(BOUT, fw, fbase, fsel, kc, "ksh")
    BOUT.c([(1, [fw[18]]), (-1, []), (1, ["z_d"]), (1, ["eqv"]), (-1, ["z_d", "eqv"])], "fw18 = rw flag")
    for k in range(4):
        BOUT.slot(f"f3o{k}")
    for k in range(4):
        BOUT.select("flush", f"f3o{k}", f"rh{k}", f"final_rh{k}")
    BOUT.c([(1, ["final_len_r"]), (-1, ["len_r"]), (-1, ["flush"])], "final_len_r = len_r + flush")
    zeros = [f"fz{k}" for k in range(4)]
    esg.poseidon(F[0], [f"x.fw{k}" for k in range(8)] + zeros, "f1o")
    for z in zeros:
        F[0].c([(1, [z])], f"{z} = 0")
    esg.poseidon(F[1], [f"x.fw{k}" for k in range(8, 16)] + [f"y.f1o{8 + k}" for k in range(4)], "f2o")
    esg.poseidon(F[2], [f"x.fw{k}" for k in range(16, 20)] + [f"x.rh{k}" for k in range(4)] + [f"y.f2o{8 + k}" for k in range(4)],
                 [f"x.f3o{k}" for k in range(4)] + [f"f3w{k}" for k in range(4, 12)])
    for k in range(4):
        PI.slot(f"pi{k}")

    # ---------------- closed-form section (gen_ram_circuit.ClosedForm): what the reference's circuit derives in-trace
    cf = dsl.ClosedForm()
    SRC = dsl.ClosedForm
    # observable input (StorageDeduplicatorInputData: shard_id_to_process, unsorted_log_queue_state, intermediate_sorted_queue_state)
    OI = cf.sponge("OI", [None] * 19, free_src=SRC.SRC_OBS_IN)
    oi = lambda w: cf.word_cell(OI, w)  # noqa: E731
    # hidden FSM input (StorageDeduplicatorFSMInputOutput, storage_sort_dedup.rs:577-612): lhs 0, rhs 2, unsorted 4, sorted 13, result 22,
    # cycle_idx 31, previous_packed_key 32 (13), previous_key 45 (8), previous_address 53 (5), previous_timestamp 58, has 59, base 60, current 68, depth 76
    FI = cf.sponge("FI", [None] * 77, free_src=SRC.SRC_FSM_IN)
    fi = lambda w: cf.word_cell(FI, w)  # noqa: E731
    SEL = dsl.Selections(cf, "SEL")
    for qi, q in enumerate(("u", "s")):
        for k in range(4):
            SEL.sel3(oi(1 + 9 * qi + k), fi(4 + 9 * qi + k), (BIN, f"{q}h{k}"))
            SEL.sel3(oi(1 + 9 * qi + 4 + k), fi(4 + 9 * qi + 4 + k), (BOUT, f"tail_{q}{k}"))
        SEL.sel3(oi(1 + 9 * qi + 8), fi(4 + 9 * qi + 8), (BIN, f"len_{q}"))
    for k in range(4):
        SEL.sel2(0, fi(22 + 4 + k), (BIN, f"rh{k}"))  # the result queue starts empty
    SEL.sel2(0, fi(30), (BIN, "len_r"))
    for r in range(2):
        SEL.sel2(1, fi(r), (BIN, f"lhs{r}"))
        SEL.sel2(1, fi(2 + r), (BIN, f"rhs{r}"))
    SEL.sel2(0, oi(0), (BIN, "ksh"))  # the open cell's shard is the instance's shard once a cell is open
    SEL.sel2(0, fi(58), (BIN, "kts"))
    SEL.not_flag((BIN, "valid"))
    SEL.sel2(0, fi(76), (BIN, "depth"))
    SEL.sel2(0, fi(59), (BIN, "has"))
    for k in range(8):
        SEL.sel2(0, fi(60 + k), (BIN, f"base{k}"))
        SEL.sel2(0, fi(68 + k), (BIN, f"cur{k}"))
    # kc (the open cell's key as 3-byte chunks of the packed key's 52 bytes) = start ? 0 : re-chunked in-trace from the FSM input's
    # previous_packed_key (five bridge rows); previous_key / previous_address repeat the packed key's limbs
    KI_rows, KIE, _ = packed_key_rows(cf, "KI", lambda r, v, j: cf.copy(r, v, *fi(32 + j)))
    cf.rows += KI_rows
    for k in range(18):
        SEL.sel2(0, (KIE, f"KI_kc{k}"), (BIN, kc[k]))
    for j in range(13):
        cf.copy(*fi(45 + j), *fi(32 + j))
    cf.copy(BIN, "cidx", *fi(31))  # cycle_idx is carried whatever the start flag says (storage_sort_dedup.rs:597)
    # hidden FSM output: the registers after the last cycle; the words the builders replace by placeholders when the instance completes
    # (nobody consumes them) are the registers unless completion
    OSEL = dsl.Selections(cf, "OSEL", flag_cell=(BOUT, "completion"))
    q9 = lambda h, q: [(BOUT, f"{h}{k}") for k in range(4)] + [(BOUT, f"tail_{q}{k}") for k in range(4)] + [(BOUT, f"len_{q}")]  # noqa: E731
    unless = lambda reg, w: OSEL.free_unless_flag((BOUT, reg), SRC.SRC_FSM_OUT, w)  # noqa: E731
    # the handed-over packed key: its chunks, re-derived from the words the FSM-output sponge absorbs, are the registers kc unless the instance completes
    KO_rows, KOE, ko = packed_key_rows(cf, "KO", lambda r, v, j: cf.free_cell(r, v, SRC.SRC_FSM_OUT, 32 + j))
    for k in range(18):
        OSEL.eq_unless_flag((KOE, f"KO_kc{k}"), (BOUT, kc[k]))
    fo_words = ([(BOUT, f"lhs{r}") for r in range(2)] + [(BOUT, f"rhs{r}") for r in range(2)] + q9("uh", "u") + q9("sh", "s") +
                [("const", 0)] * 4 + [(BOUT, f"final_rh{k}") for k in range(4)] + [(BOUT, "final_len_r")] + [unless("cidx", 31)] +
                ko + ko + [unless("kts", 58), unless("has", 59)] + [unless(f"base{k}", 60 + k) for k in range(8)] +
                [unless(f"cur{k}", 68 + k) for k in range(8)] + [unless("depth", 76)])
    assert len(fo_words) == 77
    cf.rows += KO_rows + OSEL.rows
    FO = cf.sponge("FO", fo_words, free_src=SRC.SRC_FSM_OUT)
    # observable output (final_sorted_queue_state): completion ? the result queue after the flush : the placeholder (zeros)
    OS2 = dsl.Selections(cf, "OGATE", flag_cell=(BOUT, "completion"))
    oo_words = [("const", 0)] * 4 + [OS2.gate((BOUT, f"final_rh{k}")) for k in range(4)] + [OS2.gate((BOUT, "final_len_r"))]
    cf.rows += OS2.rows
    OO = cf.sponge("OO", oo_words)
    # Fiat-Shamir challenges over the observable input's queue tails and lengths (storage_sort_dedup.rs:120-130): 10 words, 40 challenges
    fs_words = [oi(1 + 4 + k) for k in range(4)] + [oi(9)] + [oi(10 + 4 + k) for k in range(4)] + [oi(18)]
    CH = cf.sponge("CH", fs_words, squeeze=4)
    dsl.challenge_links(cf, BIN, CH, 2, 20)
    last = lambda rows_: rows_[-1]  # noqa: E731
    cp_words = [SEL.flag(), (BOUT, "completion")]
    for sp in (OI, OO, FI, FO):
        cp_words += [(last(sp), f"{last(sp).name}_o{k}") for k in range(4)]
    CP = cf.sponge("CP", cp_words)
    for k in range(4):
        cf.copy(PI, f"pi{k}", last(CP), f"{last(CP).name}_o{k}")
    pos = cf.rows.index(KIE) + 1
    cf.rows[pos:pos] = SEL.rows  # fill order: a row's copies come from rows before it (or from the register rows)
    esg.build.cf = cf  # links_of appends the section's copies

    rows = U + S + R + [A] + X + [K, C1, C2, Q, BIN, BOUT] + F + [PI] + cf.rows
    return rows, regs


def packed_key_rows(cf, prefix, bind):
    """The open cell's key as the registers hold it — eighteen 3-byte chunks kc0..kc17 of the 52 little-endian bytes of previous_packed_key
    (key limbs, then address limbs: comparison_key, log_query.rs:82-92) — from the FSM's thirteen words: four rows split the limbs into bytes
    (16 lookups per row), one row recomposes the chunks. bind(row, var, j): ties the cell to previous_packed_key[j] (a copy of the FSM-input
    sponge's cell, or a FREE cell the FSM-output sponge copies). Returns (rows, KE row, [(row, var) of limb j])."""
    rows, tb, limbs = [], [], []
    for j0 in range(0, 13, 4):
        r = Row(f"{prefix}B{j0 // 4}", False)
        for j in range(j0, min(j0 + 4, 13)):
            v = f"{prefix}_p{j}"
            r.slot(v)
            bind(r, v, j)
            limbs.append((r, v))
            tb += [(r, b) for b in cf.bytes_of(r, v, v)]
        rows.append(r)
    KE = Row(f"{prefix}E", False)
    for i, (r, b) in enumerate(tb):
        KE.slot(f"kb{i}")
        cf.copy(KE, f"kb{i}", r, b)
    for k in range(17):
        cf.linear(KE, f"{prefix}_kc{k}", [(1, f"kb{3 * k}"), (1 << 8, f"kb{3 * k + 1}"), (1 << 16, f"kb{3 * k + 2}")], why=f"key chunk {k}")
    cf.linear(KE, f"{prefix}_kc17", [(1, "kb51")], why="key chunk 17 = the last byte")
    rows.append(KE)
    return rows, KE, limbs


if __name__ == "__main__":
    rows, regs = build()
    links = esg.links_of(rows, regs)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "include", "zkw_storage_sorter_circuit_spec.h")
    nt, nc = dsl.emit(rows, links, path, prefix="SS", guard="ZKW_STORAGE_SORTER_CIRCUIT_SPEC_H",
                      title=("/* GENERATED by tools/gen_storage_sorter_circuit.py — do not edit. Layout contract of the StorageSorter trace",
                             " * emitted by zkw_storage_sorter_synthesize (\"zkw trace v2\"). */",
                             "#include \"zkw_ram_circuit_spec.h\" /* rc_term, rc_constraint, rc_link */"),
                      poseidon_rows=("U1", "U2", "U3", "S1", "S2", "S3", "R1", "R2", "R3", "F1", "F2", "F3") + tuple(esg.build.cf.p2_names),
                      shared_types=True, cf_tables=esg.build.cf.tables(rows, esg.build.cf.rows))
    esg.emit_scatter(rows, path, "SS")
    for r in rows:
        print(f"{r.name:8s} slots {len(r.slots):3d} lookups {len(r.lookups):2d} constraints {len(r.constraints)}")
    print(f"{nt} terms, {nc} constraints, {len(links)} links -> {path}")
