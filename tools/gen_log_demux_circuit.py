#!/usr/bin/env python3
"""Generates include/zkw_log_demux_circuit_spec.h — the declarative layout of the LogDemuxer trace that libzkw emits
("zkw trace v2", circuit type 4), in the DSL of tools/gen_ram_circuit.py.

Geometry of the reference wrapper (circuit_definitions/.../base_layer/log_demux.rs:27-38): 136 copy columns, 1x14
width-1 range-check lookups, Poseidon2 flattened gate, 2^20 rows, capacity 58 750; witness semantics
src/witness/individual_circuits/log_demux.rs:20-388. The circuit body lives in the absent crate era-zkevm_circuits, so
gate placement is OUR design ("parity unpinned" at the trace-layout level).

Statement, per cycle (12 rows, region-major): pop one record of the log queue — a 4-wide queue hashes enc(20) ||
head(4) in three permutations (circuit_encodings/src/lib.rs:179-221): rows I1-I3 —, split the encoding's words 10..17
into bytes (X0-X3) to read the aux byte, the shard id and the 20 address bytes (log_query.rs:118-196), derive the
one-hot route (row R; log_demux.rs:171-251: storage / events / L1 messages by aux byte, the three precompiles by aux
byte and address, anything else of the precompile kind is dropped), and push the SAME encoding into the routed queue:
ONE conditional push (P1-P3) whose old tail is the routed queue's tail, selected by the route flags (row Q), which
also keeps the six tails and lengths and the input queue's head and length.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_ram_circuit as dsl  # noqa: E402
import gen_events_sorter_circuit as esg  # noqa: E402  (poseidon, queue_rows, links_of, emit_scatter)

Row = dsl.Row
QUEUES = ("st", "ev", "l1", "kc", "sh", "ec")  # ZKW_DEMUX_STORAGE .. ZKW_DEMUX_ECRECOVER, in that order
AUX = {"st": 0, "ev": 1, "l1": 2, "pre": 3}     # ZKW_DEMUX_PARAMS_DEFAULT
ADDRESS = {"kc": 0x8010, "sh": 0x02, "ec": 0x01}


def build():
    dsl.G, dsl.L = 136, 14
    I = [Row("I1"), Row("I2"), Row("I3")]
    Pp = [Row("P1"), Row("P2"), Row("P3")]
    X = [Row(f"X{k}") for k in range(4)]
    R, Q = Row("R"), Row("Q")
    BIN, BOUT, PI = Row("BND_IN", False), Row("BND_OUT", False), Row("PI", False)

    es = [f"es{k}" for k in range(20)]
    esg.queue_rows(I, es, [f"p.ih{k}" for k in range(4)], "i", "iz")
    esg.queue_rows(Pp, es, [f"sel{k}" for k in range(4)], "p", "pz")

    # ---------------- rows X0..X3: es10..es17 = low u32 + three riders << 32/40/48
    ab = [f"a{k}" for k in range(20)]  # address bytes, little-endian over the five u32 limbs
    riders = {10: ["kb30", "kb31", ab[0]], 11: ab[1:4], 12: ab[4:7], 13: ab[7:10], 14: ab[10:13], 15: ab[13:16], 16: ab[16:19],
              17: [ab[19], "aux", "shard"]}
    for k in range(10, 18):
        row = X[(k - 10) // 2]
        lo = [f"w{k}_b{j}" for j in range(4)]
        for b in lo + riders[k]:
            row.lookup(b)
        row.c([(1, [es[k]])] + [(-(1 << (8 * j)), [lo[j]]) for j in range(4)] +
              [(-(1 << (32 + 8 * j)), [riders[k][j]]) for j in range(3)], f"es{k} = low word + riders")

    # ---------------- row R: the route (log_demux.rs:171-251)
    for kind, aux in AUX.items():
        R.is_zero([(1, "aux")] + ([(-aux, "one")] if aux else []), f"w_{kind}", f"is_{kind}", f"aux == {aux}")
    R.c([(1, ["one"]), (-1, [])], "one = 1")
    R.is_zero([(1, ab[k]) for k in range(4, 20)], "w_hz", "hz", "address bytes 4..19 are zero (each < 256: no wrap)")
    for q, addr in ADDRESS.items():
        R.is_zero([(1 << (8 * k), ab[k]) for k in range(4)] + [(-addr, "one")], f"w_a{q}", f"eq_{q}", f"address limb 0 == {addr:#x}")
    R.c([(1, ["can_pop"]), (-1, ["can_pop", "is_st"]), (-1, ["can_pop", "is_ev"]), (-1, ["can_pop", "is_l1"]), (-1, ["can_pop", "is_pre"])],
        "the aux byte is one of the four kinds")
    R.c([(1, ["can_pop", "is_st", "shard"])], "storage logs live in shard 0")
    R.c([(1, ["can_pop", "is_pre", es[19]])], "precompile calls are never rolled back")
    for q in ("st", "ev", "l1"):
        R.c([(1, ["can_pop", f"is_{q}"]), (-1, [f"r_{q}"])], f"r_{q} = can_pop & is_{q}")
    R.c([(1, ["can_pop", "is_pre", "hz"]), (-1, ["pre_hz"])], "pre_hz = can_pop & precompile kind & address < 2^32")
    for q in ADDRESS:
        R.c([(1, ["pre_hz", f"eq_{q}"]), (-1, [f"r_{q}"])], f"r_{q} = pre_hz & address match")

    # ---------------- row Q: input queue bookkeeping, tail selection, the six output queues
    Q.is_zero([(1, "p.len_i")], "w_li", "z_li", "len_i == 0")
    Q.c([(1, ["can_pop"]), (1, ["z_li"]), (-1, [])], "can_pop = 1 - empty")
    Q.c([(1, ["len_i"]), (-1, ["p.len_i"]), (1, ["can_pop"])], "len_i = p.len_i - can_pop")
    for k in range(4):
        Q.select("can_pop", f"i3o{k}", f"p.ih{k}", f"ih{k}")
    for k in range(4):
        Q.c([(-1, [f"sel{k}"])] + [(1, [f"r_{q}", f"p.qt_{q}{k}"]) for q in QUEUES], f"sel{k} = tail of the routed queue (0 if dropped)")
    for q in QUEUES:
        for k in range(4):
            Q.select(f"r_{q}", f"p3o{k}", f"p.qt_{q}{k}", f"qt_{q}{k}")
        Q.c([(1, [f"ql_{q}"]), (-1, [f"p.ql_{q}"]), (-1, [f"r_{q}"])], f"ql_{q} = p.ql_{q} + r_{q}")

    # ---------------- boundary rows
    regs = [f"ih{k}" for k in range(4)] + ["len_i"]
    for q in QUEUES:
        regs += [f"qt_{q}{k}" for k in range(4)] + [f"ql_{q}"]
    for v in regs:
        BIN.slot(v)
    for v in regs:
        BOUT.slot(v)
    for k in range(4):
        BOUT.slot(f"tail_i{k}")
    BOUT.boolean("completion")
    BOUT.is_zero([(1, "len_i")], "w_end", "z_end", "queue exhausted")
    for k in range(4):
        BOUT.c([(1, ["z_end", f"ih{k}"]), (-1, ["z_end", f"tail_i{k}"])], f"empty queue: head == tail ({k})")
    BOUT.c([(1, ["completion"]), (-1, ["completion", "z_end"])], "completion => queue exhausted")
    for k in range(4):
        PI.slot(f"pi{k}")

    # ---------------- closed-form section (gen_ram_circuit.ClosedForm): what the reference's circuit derives in-trace
    cf = dsl.ClosedForm()
    SRC = dsl.ClosedForm
    OI = cf.sponge("OI", [None] * 9, free_src=SRC.SRC_OBS_IN)  # LogDemuxerInputData: initial_log_queue_state (head 4, tail 4, length)
    oi = lambda w: cf.word_cell(OI, w)  # noqa: E731
    # hidden FSM input (LogDemuxerFSMInputOutput, log_demux.rs:283-301): the input queue, then the six output queues
    FI = cf.sponge("FI", [None] * 63, free_src=SRC.SRC_FSM_IN)
    fi = lambda w: cf.word_cell(FI, w)  # noqa: E731
    SEL = dsl.Selections(cf, "SEL")
    for k in range(4):
        SEL.sel3(oi(k), fi(k), (BIN, f"ih{k}"))
        SEL.sel3(oi(4 + k), fi(4 + k), (BOUT, f"tail_i{k}"))
    SEL.sel3(oi(8), fi(8), (BIN, "len_i"))
    for c, q in enumerate(QUEUES):  # the output queues start empty (log_demux.rs:110-168)
        for k in range(4):
            SEL.sel2(0, fi(9 + 9 * c + 4 + k), (BIN, f"qt_{q}{k}"))
        SEL.sel2(0, fi(9 + 9 * c + 8), (BIN, f"ql_{q}"))
    # hidden FSM output: the registers after the last cycle
    fo_words = [(BOUT, f"ih{k}") for k in range(4)] + [(BOUT, f"tail_i{k}") for k in range(4)] + [(BOUT, "len_i")]
    for q in QUEUES:
        fo_words += [("const", 0)] * 4 + [(BOUT, f"qt_{q}{k}") for k in range(4)] + [(BOUT, f"ql_{q}")]
    FO = cf.sponge("FO", fo_words)
    # observable output (the six queues): completion ? the registers : the placeholder (zeros)
    OSEL = dsl.Selections(cf, "OSEL", flag_cell=(BOUT, "completion"))
    oo_words = []
    for q in QUEUES:
        oo_words += [("const", 0)] * 4 + [OSEL.gate((BOUT, f"qt_{q}{k}")) for k in range(4)] + [OSEL.gate((BOUT, f"ql_{q}"))]
    cf.rows += OSEL.rows
    OO = cf.sponge("OO", oo_words)
    last = lambda rows_: rows_[-1]  # noqa: E731
    cp_words = [SEL.flag(), (BOUT, "completion")]
    for sp in (OI, OO, FI, FO):
        cp_words += [(last(sp), f"{last(sp).name}_o{k}") for k in range(4)]
    CP = cf.sponge("CP", cp_words)
    for k in range(4):
        cf.copy(PI, f"pi{k}", last(CP), f"{last(CP).name}_o{k}")
    pos = cf.rows.index(last(FI)) + 1
    cf.rows[pos:pos] = SEL.rows  # fill order: a row's copies come from rows before it (or from the register rows)
    esg.build.cf = cf  # links_of appends the section's copies

    rows = I + Pp + X + [R, Q, BIN, BOUT, PI] + cf.rows
    return rows, regs


if __name__ == "__main__":
    rows, regs = build()
    links = esg.links_of(rows, regs)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "include", "zkw_log_demux_circuit_spec.h")
    nt, nc = dsl.emit(rows, links, path, prefix="LD", guard="ZKW_LOG_DEMUX_CIRCUIT_SPEC_H",
                      title=("/* GENERATED by tools/gen_log_demux_circuit.py — do not edit. Layout contract of the LogDemuxer trace",
                             " * emitted by zkw_log_demux_synthesize (\"zkw trace v2\"). */",
                             "#include \"zkw_ram_circuit_spec.h\" /* rc_term, rc_constraint, rc_link */"),
                      poseidon_rows=("I1", "I2", "I3", "P1", "P2", "P3") + tuple(esg.build.cf.p2_names), shared_types=True,
                      cf_tables=esg.build.cf.tables(rows, esg.build.cf.rows))
    esg.emit_scatter(rows, path, "LD")
    for r in rows:
        print(f"{r.name:8s} slots {len(r.slots):3d} lookups {len(r.lookups):2d} constraints {len(r.constraints)}")
    print(f"{nt} terms, {nc} constraints, {len(links)} links -> {path}")
