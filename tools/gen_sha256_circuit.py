#!/usr/bin/env python3
"""Generates include/zkw_sha256_circuit_spec.h (Sha256RoundFunction, type 6) and include/zkw_code_decommitter_circuit_spec.h
(CodeDecommitter, type 3): ONE netlist — a SHA-256 compression on 4-bit chunks — on the reference's two geometries
(sha256_round_function.rs:28-39,121-134: 116 + 4 x 9; code_decommitter.rs:28-39,121-134: 108 + 4 x 11) and its table set
TriXor4 / Ch4 / Maj4 / Split4BitChunk<1> / <2> (12 320 rows = `total_tables_len` of vk_6.json / vk_3.json). Format: tools/netlist.py.

A 32-bit word is 8 nibbles, least significant first (state: 8 words x 8 nibbles = 64 elements).
  * rotr / shr by n = 4q + m, m != 0: the word is re-chunked at phase m — low piece (m bits), seven nibbles at bits m + 4t, high
    piece (4 - m bits) — by ONE gate (the word's nibbles in, the seven middle nibbles NEW, the two boundary pieces taken from a
    lookup) and ONE Split4BitChunk lookup whose key packs the boundary pieces: its outputs are the pieces (range-checked) and the
    nibble with the halves swapped = the rotated word's wrap-around nibble. Nibble i of the rotated word is middle piece (i + q) mod 8
    (or the wrap nibble); shr drops what falls off. Key: low | high << m for m = 1, 2 (Split<m>), high | low << 1 for m = 3 (Split<1>).
  * s0 / s1 / S0 / S1: TriXor4 on the three re-chunked words; ch: Ch4; maj: Maj4.
  * additions: one gate per chain, operands as nibbles, out = 8 NEW nibbles + a carry nibble:
      e' = d + h + S1 + ch + K + W,  a' = h + S1 + ch + K + W + S0 + maj,  W[i] = W[i-16] + s0 + W[i-7] + s1,  H' = H + v.
    Out nibbles are range-checked by the lookups that consume them (Ch4 / Maj4, or a re-chunking that bounds the word), carries
    and otherwise unconsumed nibbles by TriXor4(x, y, z) range lookups (three per lookup).
  * in = reset ? IV : prev and out = idle ? prev : H' are Ch4 selects with the header's masks 15 * reset / 15 * idle.
The 128 message nibbles are FREE elements (block byte j = nibbles 2j (low), 2j + 1 (high); word i = bytes 4i .. 4i + 3 big-endian)."""
import hashlib
import os
import random
import struct
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import netlist as nl  # noqa: E402

K = [0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
     0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
     0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
     0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
     0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
     0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
     0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2]
IV = [0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19]
HDR_RESET, HDR_IDLE, HDR_M0, HDR_M1 = range(4)
ZERO = nl.const(0)


class Sha(nl.StepType):
    def __init__(self):
        super().__init__("compress", nl.sha_tables())
        self.phase_cache = {}
        self.to_range_check = []

    def rechunk(self, word, m):
        """word: 8 nibble refs. Returns (mids[7], high piece, wrap nibble)."""
        key = (id(word[0]), tuple(id(x) for x in word), m)
        if key in self.phase_cache:
            return self.phase_cache[key]
        lo_src, hi_src = word[0], word[7]  # low piece = bits [0, m) of nibble 0, high piece = bits [m, 4) of nibble 7
        if m in (1, 2):
            k = self.hint((lo_src, 0, m), (hi_src, m, 4 - m))          # low | high << m
            low, high, wrap = self.lookup(f"SPLIT4_{m}", k)            # wrap = low << (4 - m) | high
        else:
            k = self.hint((hi_src, 3, 1), (lo_src, 0, 3))              # high | low << 1: the wrap nibble itself
            high, low, _ = self.lookup("SPLIT4_1", k)
            wrap = k
        # the boundary pieces cancel the word's bits below m and from m + 28 up: the middle nibbles do not depend on them (late cells),
        # so the gate is evaluated next to the hint instead of after the lookup — four dependency levels per round instead of five
        known = [(word[i], 4 * i, +1) for i in range(8)] + [(low, 0, -1, True), (high, m + 28, -1, True)]
        mids = self.gate(known, [m + 4 * t for t in range(7)])
        res = (mids, high, wrap)
        self.phase_cache[key] = res
        return res

    def rot(self, word, n, shift=False):
        q, m = divmod(n, 4)
        if m == 0:
            return [word[(i + q) % 8] if not shift or i + q < 8 else ZERO for i in range(8)]
        mids, high, wrap = self.rechunk(word, m)
        out = []
        for i in range(8):
            t = i + q
            if not shift:
                t %= 8
                out.append(mids[t] if t < 7 else wrap)
            else:
                out.append(mids[t] if t < 7 else high if t == 7 else ZERO)
        if shift:  # the middle nibbles that fall off still have to be range-checked (if nothing else consumes them)
            self.to_range_check += [mids[t] for t in range(min(q, 7))]
        return out

    def xor3(self, x, y, z):
        return [self.lookup("TRIXOR4", x[i], y[i], z[i]) for i in range(8)]

    def add(self, words, constant=0):
        known = [(w[i], 4 * i, +1) for w in words for i in range(8)]
        cells = self.gate(known, [4 * i for i in range(9)], constant)
        self.to_range_check.append(cells[8])  # the carry
        return cells[:8]


# A compression is cut into steps of consecutive rounds: a step's values are what a wave of the fill keeps in LDS, so three steps
# put twice as many cycles in flight per CU as one. State between the steps (256 elements, also the engine's
# cycle state — elements 64.. are zero between cycles): a..h (64 nibbles), the 16 most recent message-schedule words (128), the
# chaining value H the compression started from (64, for the final addition).
SEGMENTS = [(0, 24), (24, 44), (44, 64)]
STATE = 256


def build_step(r0, r1):
    s = Sha()
    s.name = f"rounds{r0}_{r1}"
    first, last = r0 == 0, r1 == 64
    W = {}
    if first:
        # chaining state: in = reset ? IV : prev (Ch4(mask, x, y) = mask ? x : y bitwise)
        H = [[s.lookup("CH4", nl.hdr(HDR_M0), nl.const((IV[j] >> (4 * i)) & 15), nl.prev(8 * j + i)) for i in range(8)] for j in range(8)]
        v = list(H)
        for i in range(16):  # message words from the FREE nibbles: a load gate gives every nibble a home cell
            src = []  # nibble t of word i: byte 4i + 3 - t // 2, low nibble for even t
            for t in range(8):
                byte = 4 * i + 3 - t // 2
                src.append(nl.free(2 * byte + (t & 1)))
            W[i] = s.gate([(src[t], 4 * t, +1) for t in range(8)], [4 * t for t in range(8)])
            s.to_range_check += W[i]
    else:
        v = [[nl.prev(8 * j + i) for i in range(8)] for j in range(8)]
        for t in range(16):
            W[r0 - 16 + t] = [nl.prev(64 + 8 * t + i) for i in range(8)]
        H = [[nl.prev(192 + 8 * j + i) for i in range(8)] for j in range(8)]
    for i in range(max(16, r0), r1):
        x, y = W[i - 15], W[i - 2]
        s0 = s.xor3(s.rot(x, 7), s.rot(x, 18), s.rot(x, 3, shift=True))
        s1 = s.xor3(s.rot(y, 17), s.rot(y, 19), s.rot(y, 10, shift=True))
        W[i] = s.add([W[i - 16], s0, W[i - 7], s1])
    a, b, c, d, e, f, g, h = v
    for i in range(r0, r1):
        S1 = s.xor3(s.rot(e, 6), s.rot(e, 11), s.rot(e, 25))
        ch = [s.lookup("CH4", e[t], f[t], g[t]) for t in range(8)]
        S0 = s.xor3(s.rot(a, 2), s.rot(a, 13), s.rot(a, 22))
        maj = [s.lookup("MAJ4", a[t], b[t], c[t]) for t in range(8)]
        new_e = s.add([d, h, S1, ch, W[i]], K[i])
        new_a = s.add([h, S1, ch, W[i], S0, maj], K[i])
        a, b, c, d, e, f, g, h = new_a, a, b, c, new_e, e, f, g
    v = [a, b, c, d, e, f, g, h]
    if last:
        Hn = [s.add([H[j], v[j]]) for j in range(8)]
        # out = idle ? the state before the cycle : H'
        s.out = [s.lookup("CH4", nl.hdr(HDR_M1), nl.cyc(8 * j + i), Hn[j][i]) for j in range(8) for i in range(8)] + [ZERO] * (STATE - 64)
    else:
        s.out = [x for wd in v for x in wd] + [x for t in range(r1 - 16, r1) for x in W[t]] + [x for wd in H for x in wd]
    assert len(s.out) == STATE
    # Range lookups (three nibbles per TriXor4) for what no other lookup consumes and no re-chunking bounds: the carries, the
    # middle nibbles a shift drops, the nibbles of message words that are never re-chunked. (Out nibbles of an addition that only
    # feed further additions need none: with its carry bounded, the word is right modulo 2^32 wherever it is used.)
    consumed = {id(r) for t, ins, *_ in s.ops for r in ins if isinstance(r, nl.Val)}
    rechunked = {k[1] for k in s.phase_cache}
    for wd in (W[i] for i in range(16) if first):
        if tuple(id(x) for x in wd) in rechunked:
            for x in wd:
                consumed.add(id(x))
    pending = []
    for v_ in s.to_range_check:
        if id(v_) not in consumed:
            pending.append(v_)
            consumed.add(id(v_))
    for i in range(0, len(pending), 3):
        grp = pending[i:i + 3] + [ZERO, ZERO]
        s.lookup("TRIXOR4", grp[0], grp[1], grp[2])
    return s


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
        a, b, c, d, e, f, g, h = (t1 + S0 + maj) & 0xFFFFFFFF, a, b, c, (d + t1) & 0xFFFFFFFF, e, f, g
    return [(x + y) & 0xFFFFFFFF for x, y in zip(state, [a, b, c, d, e, f, g, h])]


def nibbles_of_words(ws):
    return [(w >> (4 * i)) & 15 for w in ws for i in range(8)]


def nibbles_of_block(block):
    return [(b >> (4 * k)) & 15 for b in block for k in range(2)]


def make_spec(prefix, general_cols, lookups_per_row):
    tables = nl.sha_tables()
    spec = nl.Spec(prefix, general_cols, 4, lookups_per_row, tables, STATE, (0, 15, 0, 15))
    by_name = {t.name: t for t in tables}
    for r0, r1 in SEGMENTS:
        s = build_step(r0, r1)
        s.tables = by_name
        for op in s.ops:  # the step was built against its own table objects: rebind to the spec's (ids / offsets)
            op[0] = by_name[op[0].name]
        spec.cycle.append((spec.add_step_type(s), []))
    return spec


def cycle_state(words):
    """the engine's cycle state of 8 chaining words: their nibbles, then zeros"""
    return nibbles_of_words(words) + [0] * (STATE - 64)


def self_check(spec):
    rng = random.Random(1)
    st = [rng.getrandbits(32) for _ in range(8)]
    blk = [rng.randrange(256) for _ in range(64)]
    frees = lambda b: [nibbles_of_block(b)] + [[]] * (len(SEGMENTS) - 1)  # noqa: E731
    assert spec.evaluate_cycle(cycle_state(st), frees(blk), 0, 0) == cycle_state(sha_compress(st, blk))
    assert spec.evaluate_cycle(cycle_state(st), frees(blk), 0, 1) == cycle_state(st)  # idle carries the state
    for blk2 in ([0] * 64, [255] * 64):
        for st2 in ([0] * 8, [0xFFFFFFFF] * 8):
            assert spec.evaluate_cycle(cycle_state(st2), frees(blk2), 0, 0) == cycle_state(sha_compress(st2, blk2))
    msg = bytes(rng.randrange(256) for _ in range(100))
    padded = msg + b"\x80" + bytes((55 - len(msg)) % 64) + struct.pack(">Q", 8 * len(msg))
    state = cycle_state([0] * 8)
    for i in range(0, len(padded), 64):
        state = spec.evaluate_cycle(state, frees(padded[i:i + 64]), 1 if i == 0 else 0, 0)
    words = [sum(state[8 * j + t] << (4 * t) for t in range(8)) for j in range(8)]
    assert b"".join(struct.pack(">I", w) for w in words) == hashlib.sha256(msg).digest(), "netlist != SHA-256"


CIRCUITS = {"SC": (116, 9, "zkw_sha256_circuit_spec.h", "Sha256RoundFunction: 116 + 4 x 9 columns"),
            "DC": (108, 11, "zkw_code_decommitter_circuit_spec.h", "CodeDecommitter: the same compression on 108 + 4 x 11 columns")}


def emit(prefix, path=None):
    g, r, header, circuit = CIRCUITS[prefix]
    spec = make_spec(prefix, g, r)
    self_check(spec)
    path = path or os.path.join(nl.root(), "include", header)
    spec.emit(path, f"tools/gen_sha256_circuit.py ({circuit})")
    return spec, path


def main():
    for prefix in CIRCUITS:
        spec, path = emit(prefix)
        print(spec.stats(), path)


if __name__ == "__main__":
    main()
