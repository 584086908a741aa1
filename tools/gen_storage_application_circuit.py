#!/usr/bin/env python3
"""Generates include/zkw_storage_application_circuit_spec.h (StorageApplication, type 10): the Merkle-path walks of the
storage application as a netlist of Blake2s-256 compressions on bytes, on the reference's geometry and table set
(circuit_definitions/src/circuit_definitions/base_layer/storage_apply.rs:28-39: 60 + 3 x 26 columns; :124-140: Xor8, And8,
ByteSplit<1, 2, 3, 4, 7> = 132 352 rows = `total_tables_len` of setup/base_layer/vk_10.json). Format: tools/netlist.py.

What the reference's circuit does per tree query (src/witness/individual_circuits/storage_application.rs:141-153: a read is
ONE tree query, a write TWO — the path of the old leaf, then the path of the new leaf — and `cycles_per_storage_application`
counts tree queries): hash the leaf (Blake2s-256 of index_be(8) || value(32), src/witness/tree/mod.rs:322-329), then walk 256
levels, hashing (left || right) with the sibling on the side the key's bit names (tree/mod.rs:394-402, :187-217).

One CYCLE = one Blake2s compression of such a walk; a walk = 257 cycles (the leaf, then levels 0..255):
  state (65 bytes): cur[32] the running hash, key[33] = the derived key << 1, shifted right by one bit per cycle
  reset = 1 (leaf cycle): X = FREE[0..32), key = FREE[64..97) (the walk's key << 1: its bit 0 is 0, so the leaf cycle does
          not swap), message = X || Y with Y = FREE[32..64) (index_be || value || zeros), t = 40
  reset = 0 (level cycle): X = cur, Y = FREE[32..64) the sibling; bit = key & 1; (left, right) = bit ? (Y, X) : (X, Y); t = 64
  cur' = idle ? cur : Blake2s-256 compression (h = IV ^ 0x01010020, t, final) of left || right;  key' = key >> 1
A word is 4 bytes, least significant first. G's rotations by 16 and 8 renumber bytes; by 12 / 7: ByteSplit<4> / <7> of every
byte and one 2-term gate per result byte. 32-bit additions are one gate each (byte digits + a carry; carries are range-checked
two per Xor8 lookup). Selects are x ^ (mask & (x ^ y)) with the byte masks 255 * reset (a gate over the header's reset
bit), 255 * bit, 255 * idle (header m1); t = 64 - 24 * reset is the header's m0.
Not in the trace (tied by the builder, like the queue gadgets of the other netlist circuits): the equality of a
walk's last hash with the root register, the key derivation (Blake2s of address || key: the same compression, fed as a
reset cycle would be), the Keccak accumulator over the state diffs, the storage-log queue pops.
"""
import hashlib
import os
import random
import struct
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import netlist as nl  # noqa: E402

IV = [0x6A09E667, 0xBB67AE85, 0x3C6EF372, 0xA54FF53A, 0x510E527F, 0x9B05688C, 0x1F83D9AB, 0x5BE0CD19]
SIGMA = [[0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15], [14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3],
         [11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4], [7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8],
         [9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13], [2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9],
         [12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11], [13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10],
         [6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5], [10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0]]
H0 = [IV[0] ^ 0x01010020] + IV[1:]  # digest 32 bytes, no key, fanout = depth = 1
STATE, N_FREE = 65, 97
HDR_RESET, HDR_M0, HDR_M1 = 0, 2, 3


def cbytes(x):
    return [nl.const((x >> (8 * i)) & 255) for i in range(4)]


def is_const(r):
    return not isinstance(r, nl.Val) and r[0] == "const"


class Builder:
    def __init__(self, tables):
        self.st = nl.StepType("walk", tables)
        self.carries = []

    def xor(self, a, b):
        """bytewise XOR of two words (lists of 4 refs); constants fold"""
        out = []
        for x, y in zip(a, b):
            if is_const(x) and is_const(y):
                out.append(nl.const(x[1] ^ y[1]))
            elif is_const(y) and y[1] == 0:
                out.append(x)
            elif is_const(x) and x[1] == 0:
                out.append(y)
            else:
                out.append(self.st.lookup("XOR8", x, y))
        return out

    def add(self, *words):
        """sum of words mod 2^32: one gate, byte digits out + a carry (constant operands go into the gate's constant)"""
        k, known = 0, []
        for w in words:
            for i, r in enumerate(w):
                if is_const(r):
                    k += r[1] << (8 * i)
                else:
                    known.append((r, 8 * i, +1))
        if not known:
            return cbytes(k & 0xFFFFFFFF)
        # (nl_gate.constant is 32 bits; what is dropped only lowers the carry, the byte digits are the sum mod 2^32 either way)
        news = self.st.gate(known, [0, 8, 16, 24, 32], constant=k & 0xFFFFFFFF)
        self.carries.append(news[4])
        return news[:4]

    def rotr(self, w, n):
        q, s = divmod(n, 8)
        w = [w[(i + q) % 4] for i in range(4)]
        if s == 0:
            return w
        if all(is_const(r) for r in w):
            x = sum(r[1] << (8 * i) for i, r in enumerate(w))
            return cbytes(((x >> s) | (x << (32 - s))) & 0xFFFFFFFF)
        parts = [self.st.lookup(f"BYTESPLIT_{s}", r) for r in w]  # (x mod 2^s, x >> s)
        return [self.st.gate([(parts[j][1], 0, +1), (parts[(j + 1) % 4][0], 8 - s, +1)], [0])[0] for j in range(4)]

    def g(self, v, a, b, c, d, x, y):
        v[a] = self.add(v[a], v[b], x)
        v[d] = self.rotr(self.xor(v[d], v[a]), 16)
        v[c] = self.add(v[c], v[d])
        v[b] = self.rotr(self.xor(v[b], v[c]), 12)
        v[a] = self.add(v[a], v[b], y)
        v[d] = self.rotr(self.xor(v[d], v[a]), 8)
        v[c] = self.add(v[c], v[d])
        v[b] = self.rotr(self.xor(v[b], v[c]), 7)

    def select(self, mask, x, y):
        """mask ? y : x on bytes = x ^ (mask & (x ^ y)); y first in the XOR so that a FREE y is used exactly once"""
        d = self.st.lookup("XOR8", x, y)
        e = self.st.lookup("AND8", mask, d)
        return self.st.lookup("XOR8", x, e)


def build(tables):
    b = Builder(tables)
    st = b.st
    m_reset = st.gate([(nl.hdr(HDR_RESET), 8, +1), (nl.hdr(HDR_RESET), 0, -1)], [0])[0]  # 255 * reset
    X = [b.select(m_reset, nl.cyc(k), nl.free(k)) for k in range(32)]
    key = [b.select(m_reset, nl.cyc(32 + k), nl.free(64 + k)) for k in range(33)]
    parts = [st.lookup("BYTESPLIT_1", r) for r in key]  # (bit 0, bits 1..7)
    bit = parts[0][0]
    m_bit = st.gate([(bit, 8, +1), (bit, 0, -1)], [0])[0]  # 255 * bit
    key_out = [st.gate([(parts[k][1], 0, +1), (parts[k + 1][0], 7, +1)], [0])[0] for k in range(32)] + [parts[32][1]]
    left, right = [], []
    for k in range(32):
        t = st.lookup("XOR8", X[k], nl.free(32 + k))
        u = st.lookup("AND8", m_bit, t)
        l_ = st.lookup("XOR8", X[k], u)
        left.append(l_)
        right.append(st.lookup("XOR8", l_, t))
    msg = [(left + right)[4 * i:4 * i + 4] for i in range(16)]
    v = [cbytes(x) for x in H0 + IV]
    v[12] = [st.lookup("XOR8", nl.const(IV[4] & 255), nl.hdr(HDR_M0))] + cbytes(IV[4])[1:]  # t = 64 - 24 * reset < 256
    v[14] = cbytes(IV[6] ^ 0xFFFFFFFF)  # the only (= last) block
    for r in range(10):
        s = SIGMA[r]
        for i, (a_, b_, c_, d_) in enumerate(((0, 4, 8, 12), (1, 5, 9, 13), (2, 6, 10, 14), (3, 7, 11, 15),
                                              (0, 5, 10, 15), (1, 6, 11, 12), (2, 7, 8, 13), (3, 4, 9, 14))):
            b.g(v, a_, b_, c_, d_, msg[s[2 * i]], msg[s[2 * i + 1]])
    new = []
    for i in range(8):
        new += b.xor(b.xor(v[i], v[i + 8]), cbytes(H0[i]))
    out = [b.select(nl.hdr(HDR_M1), new[k], nl.cyc(k)) for k in range(32)]
    for i in range(0, len(b.carries), 2):  # range checks of the additions' carries, two per lookup
        pair = b.carries[i:i + 2] + [nl.const(0)]
        st.lookup("XOR8", pair[0], pair[1])
    st.out = out + key_out
    return st


def blake2s_compress_single(block, t):
    """Blake2s-256 of a message that fits one block (len = t <= 64), unkeyed"""
    m = list(struct.unpack("<16I", bytes(block)))
    v = H0 + IV
    v[12] ^= t
    v[14] ^= 0xFFFFFFFF
    rotr = lambda x, n: ((x >> n) | (x << (32 - n))) & 0xFFFFFFFF  # noqa: E731

    def g(a, b, c, d, x, y):
        v[a] = (v[a] + v[b] + x) & 0xFFFFFFFF
        v[d] = rotr(v[d] ^ v[a], 16)
        v[c] = (v[c] + v[d]) & 0xFFFFFFFF
        v[b] = rotr(v[b] ^ v[c], 12)
        v[a] = (v[a] + v[b] + y) & 0xFFFFFFFF
        v[d] = rotr(v[d] ^ v[a], 8)
        v[c] = (v[c] + v[d]) & 0xFFFFFFFF
        v[b] = rotr(v[b] ^ v[c], 7)

    for r in range(10):
        s = SIGMA[r]
        g(0, 4, 8, 12, m[s[0]], m[s[1]]); g(1, 5, 9, 13, m[s[2]], m[s[3]]); g(2, 6, 10, 14, m[s[4]], m[s[5]]); g(3, 7, 11, 15, m[s[6]], m[s[7]])  # noqa: E702
        g(0, 5, 10, 15, m[s[8]], m[s[9]]); g(1, 6, 11, 12, m[s[10]], m[s[11]]); g(2, 7, 8, 13, m[s[12]], m[s[13]]); g(3, 4, 9, 14, m[s[14]], m[s[15]])  # noqa: E702
    return list(b"".join(struct.pack("<I", H0[i] ^ v[i] ^ v[i + 8]) for i in range(8)))


def make_spec(prefix="SA", general_cols=60, lookups_per_row=26):
    tables = nl.storage_tables()
    spec = nl.Spec(prefix, general_cols, 3, lookups_per_row, tables, STATE, (64, -24, 0, 255))
    spec.cycle = [(spec.add_step_type(build(tables)), [])]
    return spec


def self_check(spec):
    rng = random.Random(7)
    rb = lambda n: [rng.randrange(256) for _ in range(n)]  # noqa: E731
    key = rng.getrandbits(256)
    key33 = list((key << 1).to_bytes(33, "little"))
    # a leaf cycle, then three levels, against hashlib
    index, value = rng.getrandbits(64), bytes(rb(32))
    leaf_msg = index.to_bytes(8, "big") + value + bytes(24)
    state = rb(32) + rb(33)  # whatever the previous walk left
    state = spec.evaluate_cycle(state, [list(leaf_msg[:32]) + list(leaf_msg[32:]) + key33], 1, 0)
    cur = hashlib.blake2s(leaf_msg[:40]).digest()
    assert bytes(state[:32]) == cur, "leaf cycle != Blake2s-256(index || value)"
    assert state[32:] == list(((key << 1) >> 1).to_bytes(33, "little"))
    for level in range(3):
        sib = bytes(rb(32))
        bit = (key >> level) & 1
        cur = hashlib.blake2s((sib + cur) if bit else (cur + sib)).digest()
        state = spec.evaluate_cycle(state, [rb(32) + list(sib) + rb(33)], 0, 0)
        assert bytes(state[:32]) == cur, f"level {level} != Blake2s-256(left || right)"
        assert state[32:] == list((key >> (level + 1)).to_bytes(33, "little"))
    assert spec.evaluate_cycle(state, [rb(97)], 0, 1)[:32] == state[:32]  # idle carries the hash
    for fill in (0, 255):  # extreme bytes through every addition
        st_ = [fill] * 65
        sib = [fill] * 32
        got = spec.evaluate_cycle(st_, [[0] * 32 + sib + [0] * 33], 0, 0)
        want = blake2s_compress_single(sib + st_[:32] if fill & 1 else st_[:32] + sib, 64)
        assert got[:32] == want
    assert blake2s_compress_single(list(leaf_msg), 40) == list(hashlib.blake2s(leaf_msg[:40]).digest())


def emit(path=None):
    spec = make_spec()
    self_check(spec)
    path = path or os.path.join(nl.root(), "include", "zkw_storage_application_circuit_spec.h")
    spec.emit(path, "tools/gen_storage_application_circuit.py (StorageApplication: Blake2s Merkle walks on 60 + 3 x 26 columns)",
              extra_defines=(f"#define SA_CYCLES_PER_WALK 257  /* the leaf hash, then 256 levels */",
                             f"#define SA_FREE_X 0\n#define SA_FREE_Y 32\n#define SA_FREE_KEY 64"))
    return spec, path


if __name__ == "__main__":
    spec, path = emit()
    print(spec.stats(), path)
