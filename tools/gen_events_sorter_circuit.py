#!/usr/bin/env python3
"""Generates include/zkw_events_sorter_circuit_spec.h — the declarative layout of the EventsSorter / L1MessagesSorter
trace that libzkw emits ("zkw trace v2", circuit types 11 and 12), in the DSL of tools/gen_ram_circuit.py.

Geometry of the reference wrapper (circuit_definitions/.../base_layer/events_sort_dedup.rs:28-39): 130 copy columns,
1x8 width-1 range-check lookups, Poseidon2 flattened gate, 2^20 rows, capacity 31 287; witness semantics
src/witness/individual_circuits/events_sort_dedup.rs:16-580. The circuit body lives in the absent crate
era-zkevm_circuits, so gate placement is OUR design ("parity unpinned" at the trace-layout level).

Rows per cycle: 13 (round 5; 22 before — the reference's own placement uses 590 817 rows for 31 287 cycles = 18.9 per cycle, ours was
looser). The ten sparse rows N0-N7, T, V (seven range-checked bytes and three linear relations each) are ONE row NTV now: its 123 cells
are the relations' operands, and the 70 bytes are range-checked in the lookup columns of the nine Poseidon2 rows (8 per row), which
the flattened gate leaves unused; NTV holds copies of them.

Statement, per cycle (region-major): pop the unsorted and the sorted log queue in lock step — a 4-wide queue
hashes enc(20) || tail(4) in three permutations (circuit_encodings/src/lib.rs:179-221): rows U1-U3, S1-S3 —, multiply
both grand-product accumulators (A, W = 20), split the sorted record's encoding far enough to (i) read timestamp and
rollback flag and (ii) rebuild the encoding of its *normalised* form — read value, timestamp, aux byte, rw and
rollback flags cleared (events_sort_dedup.rs:541-553) — (N0-N7, T, V), compare timestamps with the previous record
and apply the dedup rule (W): equal timestamps = a forward record and its rollback, both dropped; a forward record is
pushed into the result queue when the NEXT record proves it was not rolled back (R1-R3 hash the previous record's
normalised encoding); Q does the queue bookkeeping. The last record is flushed by three more permutations outside
the cycles (F1-F3) when the instance completes.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_ram_circuit as dsl  # noqa: E402

dsl.G, dsl.L = 130, 8
Row = dsl.Row


def poseidon(row, ins, out):
    assert len(ins) == 12
    for v in ins:
        row.slot(v)
    for r in range(4):
        for k in range(12):
            row.slot(f"{row.name}_f{r}_{k}")
    for r in range(22):
        row.slot(f"{row.name}_p{r}")
    for r in range(3):
        for k in range(12):
            row.slot(f"{row.name}_f{4 + r}_{k}")
    for k in range(12):
        row.slot(out[k] if isinstance(out, list) else f"{out}{k}")
    assert len(row.slots) == 130, (row.name, len(row.slots))


def queue_rows(rows3, enc, old, tag, zero_tag):
    """three permutations of one 4-wide queue operation: enc[0..8] | 0000, enc[8..16] | cap, enc[16..20] old[0..4] | cap"""
    r1, r2, r3 = rows3
    zeros = [f"{zero_tag}{k}" for k in range(4)]
    poseidon(r1, enc[0:8] + zeros, f"{tag}1o")
    for z in zeros:
        r1.c([(1, [z])], f"{z} = 0: the sponge starts from the zero state")
    poseidon(r2, enc[8:16] + [f"{tag}1o{8 + k}" for k in range(4)], f"{tag}2o")
    poseidon(r3, enc[16:20] + old + [f"{tag}2o{8 + k}" for k in range(4)], f"{tag}3o")


def build():
    U = [Row("U1"), Row("U2"), Row("U3")]
    S = [Row("S1"), Row("S2"), Row("S3")]
    R = [Row("R1"), Row("R2"), Row("R3")]
    A = Row("A")
    NTV = Row("NTV")
    N = [NTV] * 8   # (the relations of the former rows N0..N7, T, V all live in NTV)
    T, V, W, Q = NTV, NTV, Row("W"), Row("Q")
    p2_rows = U + S + R
    byte_home = []  # the range checks ride in the Poseidon2 rows' lookup columns, eight per row

    def range_checked(b):
        prow = p2_rows[len(byte_home) // dsl.L]
        prow.lookup(b)
        byte_home.append(b)
    BIN, BOUT, PI = Row("BND_IN", False), Row("BND_OUT", False), Row("PI", False)
    F = [Row("F1", False), Row("F2", False), Row("F3", False)]

    eu = [f"eu{k}" for k in range(20)]
    es = [f"es{k}" for k in range(20)]
    ne = [f"ne{k}" for k in range(20)]   # normalised encoding of the latest popped record (register)
    cn = [f"cn{k}" for k in range(20)]   # normalised encoding of THIS cycle's record
    queue_rows(U, eu, [f"p.uh{k}" for k in range(4)], "u", "uz")
    queue_rows(S, es, [f"p.sh{k}" for k in range(4)], "s", "sz")
    queue_rows(R, [f"p.ne{k}" for k in range(20)], [f"p.rh{k}" for k in range(4)], "r", "rz")

    # ---------------- row A: grand products, W = 20 (utils.rs:554-697, challenge 0 is the constant ONE, 20 is additive)
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
            A.c([(-1, [lc]), (1, [ch[20]]), (1, [enc[0]])] + [(1, [enc[k], ch[k]]) for k in range(1, 20)],
                f"{lc} = c20 + sum enc_k c_k")
            A.c([(1, [f"p.{acc}{r}", lc]), (-1, [f"n{side}{r}"])], f"n{side}{r} = acc*contribution")
            A.select("can_pop", f"n{side}{r}", f"p.{acc}{r}", f"{acc}{r}")

    # ---------------- rows N0..N7: es_k = read_value limb k + three key bytes << 32/40/48 (log_query.rs:118-196)
    for k in range(8):
        row = N[k]
        lo = [f"rv{k}_b{j}" for j in range(4)]
        hi = [f"kb{k}_b{j}" for j in range(3)]
        for b in lo + hi:
            range_checked(b)
        row.c([(1, [f"rv{k}"])] + [(-(1 << (8 * j)), [lo[j]]) for j in range(4)], f"rv{k} = sum bytes")
        row.c([(1, [es[k]]), (-1, [f"rv{k}"])] + [(-(1 << (32 + 8 * j)), [hi[j]]) for j in range(3)], f"es{k} = rv{k} + key bytes")
        row.c([(1, [cn[k]]), (-1, [es[k]]), (1, [f"rv{k}"])], f"cn{k} = es{k} without the read value")

    # ---------------- row T: es16 = timestamp + three address bytes
    tb = [f"ts_b{j}" for j in range(4)]
    ab = [f"a16_b{j}" for j in range(3)]
    for b in tb + ab:
        range_checked(b)
    T.c([(1, ["ts"])] + [(-(1 << (8 * j)), [tb[j]]) for j in range(4)], "ts = sum bytes")
    T.c([(1, [es[16]]), (-1, ["ts"])] + [(-(1 << (32 + 8 * j)), [ab[j]]) for j in range(3)], "es16 = ts + address bytes")
    T.c([(1, [cn[16]]), (-1, [es[16]]), (1, ["ts"])], "cn16 = es16 without the timestamp")
    for k in range(8, 16):
        T.c([(1, [cn[k]]), (-1, [es[k]])], f"cn{k} = es{k}")

    # ---------------- row V: es17 = tx number + address byte 19 << 32 + aux byte << 40 + shard << 48; flags
    xb = [f"tx_b{j}" for j in range(4)]
    for b in xb + ["a19", "aux", "shard"]:
        range_checked(b)
    V.c([(1, ["tx"])] + [(-(1 << (8 * j)), [xb[j]]) for j in range(4)], "tx = sum bytes")
    V.c([(1, [es[17]]), (-1, ["tx"]), (-(1 << 32), ["a19"]), (-(1 << 40), ["aux"]), (-(1 << 48), ["shard"])], "es17")
    V.c([(1, [cn[17]]), (-1, [es[17]]), (1 << 40, ["aux"])], "cn17 = es17 without the aux byte")
    V.boolean("rw")
    V.boolean("sv")
    V.boolean("rb")
    V.c([(1, [es[18]]), (-1, ["rw"]), (-2, ["sv"])], "es18 = rw + 2 is_service")
    V.c([(1, [cn[18]]), (-2, ["sv"])], "cn18 = 2 is_service")
    V.c([(1, [es[19]]), (-1, ["rb"])], "es19 = rollback")
    V.c([(1, [cn[19]])], "cn19 = 0")

    # ---------------- row W: timestamp order and the dedup rule (events_sort_dedup.rs:292-331)
    W.bytes_of("dts", "dts")
    W.boolean("bw")
    W.c([(1, ["dts"]), (-1, ["ts"]), (1, ["p.kts"]), (-(1 << 32), ["bw"])], "dts = ts - previous ts + 2^32 bw")
    W.c([(1, ["can_pop", "p.valid", "bw"])], "sorted by timestamp")
    W.is_zero([(1, "ts"), (-1, "p.kts")], "w_ts", "same_ts", "ts == previous ts")
    W.c([(1, ["can_pop", "p.valid", "same_ts"]), (-1, ["can_pop", "p.valid", "same_ts", "rb"])], "same timestamp => this record is a rollback")
    W.c([(1, ["can_pop", "p.valid", "same_ts", "p.krb"])], "... of a forward record")
    W.c([(1, ["can_pop", "p.valid", "rb"]), (-1, ["can_pop", "p.valid", "same_ts", "rb"])], "a new timestamp starts with a forward record")
    W.c([(1, ["can_pop", "rb"]), (-1, ["can_pop", "p.valid", "rb"])], "the very first record is a forward record")
    W.c([(1, ["can_pop", "p.valid"]), (-1, ["can_pop", "p.valid", "same_ts"]), (-1, ["can_pop", "p.valid", "p.krb"]),
         (1, ["can_pop", "p.valid", "same_ts", "p.krb"]), (-1, ["push"])],
        "push = can_pop & valid & new timestamp & previous record is forward")
    W.c([(1, ["valid"]), (-1, ["p.valid"]), (-1, ["can_pop"]), (1, ["can_pop", "p.valid"])], "valid = p.valid | can_pop")
    W.select("can_pop", "ts", "p.kts", "kts")
    W.select("can_pop", "rb", "p.krb", "krb")
    for k in range(4):
        W.select("push", f"r3o{k}", f"p.rh{k}", f"rh{k}")
    W.c([(1, ["len_r"]), (-1, ["p.len_r"]), (-1, ["push"])], "len_r = p.len_r + push")

    # ---------------- row Q: queue bookkeeping, heads, the normalised-encoding registers
    Q.is_zero([(1, "p.len_u")], "w_lu", "z_lu", "len_u == 0")
    Q.is_zero([(1, "p.len_s")], "w_ls", "z_ls", "len_s == 0")
    Q.c([(1, ["z_lu"]), (-1, ["z_ls"])], "both queues empty together")
    Q.c([(1, ["can_pop"]), (1, ["z_lu"]), (-1, [])], "can_pop = 1 - empty")
    Q.c([(1, ["len_u"]), (-1, ["p.len_u"]), (1, ["can_pop"])], "len_u = p.len_u - can_pop")
    Q.c([(1, ["len_s"]), (-1, ["p.len_s"]), (1, ["can_pop"])], "len_s = p.len_s - can_pop")
    for q, o in (("uh", "u3o"), ("sh", "s3o")):
        for k in range(4):
            Q.select("can_pop", f"{o}{k}", f"p.{q}{k}", f"{q}{k}")
    for k in range(20):
        Q.select("can_pop", cn[k], f"p.ne{k}", ne[k])

    # ---------------- boundary rows
    regs = ([f"uh{k}" for k in range(4)] + [f"sh{k}" for k in range(4)] + [f"rh{k}" for k in range(4)] +
            ["len_u", "len_s", "len_r", "lhs0", "lhs1", "rhs0", "rhs1", "kts", "krb", "valid"] + ne)
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
    BOUT.c([(1, ["completion", "valid"]), (-1, ["completion", "valid", "krb"]), (-1, ["flush"])],
           "flush = completion & the last record is a forward record")
    for k in range(4):
        BOUT.slot(f"f3o{k}")
    for k in range(4):
        BOUT.select("flush", f"f3o{k}", f"rh{k}", f"final_rh{k}")
    BOUT.c([(1, ["final_len_r"]), (-1, ["len_r"]), (-1, ["flush"])], "final_len_r = len_r + flush")
    # the flush: three permutations over the registers in BND_OUT
    zeros = [f"fz{k}" for k in range(4)]
    poseidon(F[0], [f"x.ne{k}" for k in range(8)] + zeros, "f1o")
    for z in zeros:
        F[0].c([(1, [z])], f"{z} = 0")
    poseidon(F[1], [f"x.ne{k}" for k in range(8, 16)] + [f"y.f1o{8 + k}" for k in range(4)], "f2o")
    poseidon(F[2], [f"x.ne{k}" for k in range(16, 20)] + [f"x.rh{k}" for k in range(4)] + [f"y.f2o{8 + k}" for k in range(4)],
             [f"x.f3o{k}" for k in range(4)] + [f"f3w{k}" for k in range(4, 12)])
    for k in range(4):
        PI.slot(f"pi{k}")

    # ---------------- closed-form section (gen_ram_circuit.ClosedForm): what the reference's circuit derives in-trace
    cf = dsl.ClosedForm()
    SRC = dsl.ClosedForm
    # observable input (EventsDeduplicatorInputData: initial_log_queue_state, intermediate_sorted_queue_state; head 4, tail 4, length)
    OI = cf.sponge("OI", [None] * 18, free_src=SRC.SRC_OBS_IN)
    oi = lambda w: cf.word_cell(OI, w)  # noqa: E731
    # hidden FSM input (EventsDeduplicatorFSMInputOutput, events_sort_dedup.rs:426-455): lhs 0, rhs 2, unsorted queue 4, sorted queue 13,
    # result queue 22, previous_key 31, previous_item 32 (LogQuery: address 5, key 8, read 8, written 8, rw, aux, rollback, service, shard, tx, timestamp)
    FI = cf.sponge("FI", [None] * 68, free_src=SRC.SRC_FSM_IN)
    fi = lambda w: cf.word_cell(FI, w)  # noqa: E731
    SEL = dsl.Selections(cf, "SEL")
    for qi, q in enumerate(("u", "s")):
        for k in range(4):
            SEL.sel3(oi(9 * qi + k), fi(4 + 9 * qi + k), (BIN, f"{q}h{k}"))
            SEL.sel3(oi(9 * qi + 4 + k), fi(4 + 9 * qi + 4 + k), (BOUT, f"tail_{q}{k}"))
        SEL.sel3(oi(9 * qi + 8), fi(4 + 9 * qi + 8), (BIN, f"len_{q}"))
    for k in range(4):
        SEL.sel2(0, fi(22 + 4 + k), (BIN, f"rh{k}"))  # the result queue starts empty (the reference's callers hand in a fresh simulator)
    SEL.sel2(0, fi(30), (BIN, "len_r"))
    for r in range(2):
        SEL.sel2(1, fi(r), (BIN, f"lhs{r}"))
        SEL.sel2(1, fi(2 + r), (BIN, f"rhs{r}"))
    SEL.sel2(0, fi(31), (BIN, "kts"))
    SEL.sel2(0, fi(32 + 31), (BIN, "krb"))
    SEL.not_flag((BIN, "valid"))
    # ne (the previous record's normalised encoding, what a later cycle pushes) = start ? 0 : the encoding re-derived in-trace from the
    # FSM input's previous_item (eight bridge rows)
    NI_rows, NEI, _ = normalised_encoding_rows(cf, "NI", lambda r, v, w: cf.copy(r, v, *fi(32 + w)))
    cf.rows += NI_rows
    for k in range(20):
        SEL.sel2(0, (NEI, f"NI_ne{k}"), (BIN, f"ne{k}"))
    # hidden FSM output: the registers after the last cycle
    OSEL = dsl.Selections(cf, "OSEL", flag_cell=(BOUT, "completion"))
    fo_key = OSEL.free_unless_flag((BOUT, "kts"), SRC.SRC_FSM_OUT, 31)  # a completing instance hands over placeholders (:174-181 of the sorter builders)
    fo_rb = OSEL.free_unless_flag((BOUT, "krb"), SRC.SRC_FSM_OUT, 32 + 31)
    # the handed-over previous_item: its normalised encoding, re-derived from the words the FSM-output sponge absorbs, is the register ne
    # unless the instance completes (eight more bridge rows; the words are FREE cells there, the sponge copies them)
    NO_rows, NEO, fo_item = normalised_encoding_rows(cf, "NO", lambda r, v, w: cf.free_cell(r, v, SRC.SRC_FSM_OUT, 32 + 32 + w - 32))
    for k in range(20):
        OSEL.eq_unless_flag((NEO, f"NO_ne{k}"), (BOUT, f"ne{k}"))
    q9 = lambda h, q: [(BOUT, f"{h}{k}") for k in range(4)] + [(BOUT, f"tail_{q}{k}") for k in range(4)] + [(BOUT, f"len_{q}")]  # noqa: E731
    fo_words = ([(BOUT, f"lhs{r}") for r in range(2)] + [(BOUT, f"rhs{r}") for r in range(2)] + q9("uh", "u") + q9("sh", "s") +
                [("const", 0)] * 4 + [(BOUT, f"final_rh{k}") for k in range(4)] + [(BOUT, "final_len_r")] + [fo_key] + [fo_item.get(w) if w != 31 else fo_rb for w in range(36)])
    assert len(fo_words) == 68
    FO = cf.sponge("FO", fo_words, free_src=SRC.SRC_FSM_OUT)
    # observable output (final_queue_state): completion ? the result queue after the flush : the placeholder (zeros)
    oo_words = [("const", 0)] * 4 + [OSEL.gate((BOUT, f"final_rh{k}")) for k in range(4)] + [OSEL.gate((BOUT, "final_len_r"))]
    pos = cf.rows.index(FO[0])
    cf.rows[pos:pos] = NO_rows + OSEL.rows
    OO = cf.sponge("OO", oo_words)
    # Fiat-Shamir challenges over the observable input's queue tails and lengths (events_sort_dedup.rs:104-116): 10 words, 40 challenges
    fs_words = [oi(4 + k) for k in range(4)] + [oi(8)] + [oi(9 + 4 + k) for k in range(4)] + [oi(17)]
    CH = cf.sponge("CH", fs_words, squeeze=4)
    dsl.challenge_links(cf, BIN, CH, 2, 20)
    last = lambda rows_: rows_[-1]  # noqa: E731
    cp_words = [SEL.flag(), (BOUT, "completion")]
    for sp in (OI, OO, FI, FO):
        cp_words += [(last(sp), f"{last(sp).name}_o{k}") for k in range(4)]
    CP = cf.sponge("CP", cp_words)
    for k in range(4):
        cf.copy(PI, f"pi{k}", last(CP), f"{last(CP).name}_o{k}")
    pos = cf.rows.index(NEI) + 1
    cf.rows[pos:pos] = SEL.rows  # fill order: a row's copies come from rows before it (or from the register rows)
    build.cf = cf

    assert len(byte_home) == 70 and len(NTV.slots) <= dsl.G, (len(byte_home), len(NTV.slots))
    rows = U + S + R + [A, NTV, W, Q, BIN, BOUT] + F + [PI] + cf.rows
    return rows, regs


# previous_item (LogQuery) words inside the FSM encoding: address 0..4, key 5..12, read_value 13..20, written_value 21..28, rw 29, aux 30,
# rollback 31, is_service 32, shard 33, tx_number 34, timestamp 35
def normalised_encoding_rows(cf, prefix, bind):
    """The NORMALISED encoding of a log record (read value, timestamp, aux byte, rw and rollback flags cleared: events_sort_dedup.rs:541-553;
    encoding log_query.rs:102-396) from its FSM words: seven rows split the 8 key and 5 address limbs into the 52 bytes that ride as the
    3-byte tails of the encoding's words (two limbs per row: the geometry has 8 lookups), one row recomposes ne0..ne19.
    bind(row, var, w): ties the cell to word w of previous_item (a copy of the FSM-input sponge's cell, or a FREE cell the FSM-output
    sponge copies). Returns (rows, NE row, {w: (row, var)})."""
    cells, rows, tb = {}, [], []
    limb_words = list(range(5, 13)) + list(range(0, 5))  # tail bytes: key bytes, then address bytes
    for j in range(7):
        r = Row(f"{prefix}B{j}", False)
        for w in limb_words[2 * j:2 * j + 2]:
            v = f"{prefix}_w{w}"
            r.slot(v)
            bind(r, v, w)
            cells[w] = (r, v)
            tb += [(r, b) for b in cf.bytes_of(r, v, v)]
        rows.append(r)
    NE = Row(f"{prefix}E", False)
    for i, (r, b) in enumerate(tb):
        NE.slot(f"tb{i}")
        cf.copy(NE, f"tb{i}", r, b)
    for w in list(range(21, 29)) + [32, 33, 34]:
        v = f"{prefix}_w{w}"
        NE.slot(v)
        bind(NE, v, w)
        cells[w] = (NE, v)
    for k in range(17):
        base = [(1, f"{prefix}_w{21 + k - 8}")] if 8 <= k < 16 else []
        cf.linear(NE, f"{prefix}_ne{k}", base + [(1 << 32, f"tb{3 * k}"), (1 << 40, f"tb{3 * k + 1}"), (1 << 48, f"tb{3 * k + 2}")], why=f"normalised encoding word {k}")
    cf.linear(NE, f"{prefix}_ne17", [(1, f"{prefix}_w34"), (1 << 32, "tb51"), (1 << 48, f"{prefix}_w33")], why="ne17 = tx + address byte 19 << 32 + shard << 48 (aux cleared)")
    cf.linear(NE, f"{prefix}_ne18", [(2, f"{prefix}_w32")], why="ne18 = 2 * is_service (rw cleared)")
    cf.linear(NE, f"{prefix}_ne19", [], why="ne19 = 0 (rollback cleared)")
    rows.append(NE)
    return rows, NE, cells


def links_of(rows, regs):
    """As in gen_decommit_sorter_circuit, plus y.v in a boundary row -> v in the boundary row that holds it (kind 5:
    row_b = that row)."""
    BIN = next(r for r in rows if r.name == "BND_IN")
    BOUT = next(r for r in rows if r.name == "BND_OUT")
    home = {}
    for ri, r in enumerate(rows):
        if not r.per_cycle:
            continue
        for v in r.slots + r.lookups:
            if not v.startswith(("p.", "g.")) and v not in home:
                home[v] = (ri, r.slot(v))
    bhome = {}
    for ri, r in enumerate(rows):
        if r.per_cycle or r in (BIN, BOUT):
            continue
        for v in r.slots:
            if not v.startswith(("x.", "y.")) and v not in bhome:
                bhome[v] = (ri, r.slot(v))
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
            elif v.startswith("y."):
                links.append((5, ri, col, bhome[v[2:]][0], bhome[v[2:]][1], 0))
    return links + build.cf.resolve(rows)


def emit_scatter(rows, path, prefix):
    names = []
    for r in rows:
        for v in r.slots + r.lookups:
            base = v.split(".", 1)[1] if v[:2] in ("p.", "g.", "x.", "y.") else v
            if base not in names:
                names.append(base)
    lines = ["", f"/* ---- scatter lists: {prefix}_VARS(X) lists every distinct variable once; {prefix}_FILL_<row>(XC, XP, XG, XX) lists the",
             "   cells of a row: XC(col, v) current value, XP previous cycle, XG per-instance global, XX a boundary row's value */",
             f"#define {prefix}_VARS(X) " + " ".join(f"X({n})" for n in names)]
    for r in rows:
        lines.append(f"#define {prefix}_NSLOTS_{r.name} {len(r.slots)}")
        lines.append(f"#define {prefix}_NLOOK_{r.name} {len(r.lookups)}")
    lines.append(f"#define {prefix}_LOOKUPS_PER_CYCLE {sum(len(r.lookups) for r in rows if r.per_cycle)}")
    for r in rows:
        ent = []
        for v in r.slots + r.lookups:
            kind = {"p.": "XP", "g.": "XG", "x.": "XX", "y.": "XX"}.get(v[:2], "XC")
            base = v.split(".", 1)[1] if kind != "XC" else v
            ent.append(f"{kind}({r.slot(v)}, {base})")
        lines.append(f"#define {prefix}_FILL_{r.name}(XC, XP, XG, XX) " + " ".join(ent))
        if r.lookups:  # the row's lookup cells alone (the Poseidon2 rows: their 130 gate cells are written by the permutation itself)
            lines.append(f"#define {prefix}_LOOK_{r.name}(XC) " + " ".join(f"XC({r.slot(v)}, {v})" for v in r.lookups))
    txt = open(path).read()
    txt = txt.replace("\n#endif\n", "\n" + "\n".join(lines) + "\n#endif\n")
    open(path, "w").write(txt)


if __name__ == "__main__":
    rows, regs = build()
    links = links_of(rows, regs)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "include", "zkw_events_sorter_circuit_spec.h")
    nt, nc = dsl.emit(rows, links, path, prefix="ES", guard="ZKW_EVENTS_SORTER_CIRCUIT_SPEC_H",
                      title=("/* GENERATED by tools/gen_events_sorter_circuit.py — do not edit. Layout contract of the EventsSorter /",
                             " * L1MessagesSorter trace emitted by zkw_events_sorter_synthesize (\"zkw trace v2\"). */",
                             "#include \"zkw_ram_circuit_spec.h\" /* rc_term, rc_constraint, rc_link */"),
                      poseidon_rows=("U1", "U2", "U3", "S1", "S2", "S3", "R1", "R2", "R3", "F1", "F2", "F3") + tuple(build.cf.p2_names),
                      shared_types=True, cf_tables=build.cf.tables(rows, build.cf.rows))
    emit_scatter(rows, path, "ES")
    for r in rows:
        print(f"{r.name:8s} slots {len(r.slots):3d} lookups {len(r.lookups):2d} constraints {len(r.constraints)}")
    print(f"{nt} terms, {nc} constraints, {len(links)} links -> {path}")
