"""Seeded synthetic inputs for the RAM-permutation hot path.

The reference gets its memory trace from the out-of-circuit VM (zk_evm, absent); the benches and parity
tests need traces of the same shape without it. `ram_trace` produces a *valid* trace: the first touch
of a cell is a write, later accesses are reads of the current value (70 %) or writes, timestamps are
strictly increasing in queue order, values are uniform 256-bit, 5 % of written values are fat pointers.
RNG: splitmix64 so that every consumer (numpy here, C in the oracle bench) can regenerate the inputs.
"""
import numpy as np

MEM_QUERY = np.dtype(
    [("timestamp", "<u4"), ("page", "<u4"), ("index", "<u4"), ("rw_flag", "u1"), ("value_is_pointer", "u1"),
     ("_pad", "u1", (2,)), ("value", "<u4", (8,))], align=False)
assert MEM_QUERY.itemsize == 48


def splitmix64(seed: int, n: int) -> np.ndarray:
    """n outputs of splitmix64 started at `seed` (vectorised: state_i = seed + (i+1)*gamma)."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + np.arange(1, n + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def ram_trace(n: int, seed: int = 1, pages: int = 64, indices: int = 256, first_page: int = 8,
              read_fraction: float = 0.7, ptr_fraction: float = 0.05, ts_start: int = 1) -> np.ndarray:
    """Valid memory trace of n queries in queue (timestamp) order."""
    r = splitmix64(seed, 12 * n).reshape(12, n)
    page = (r[0] % np.uint64(pages)).astype(np.uint32) + np.uint32(first_page)
    index = (r[1] % np.uint64(indices)).astype(np.uint32)
    want_read = (r[2] >> np.uint64(11)).astype(np.float64) / float(1 << 53) < read_fraction
    is_ptr = (r[3] >> np.uint64(11)).astype(np.float64) / float(1 << 53) < ptr_fraction
    rand_val = np.empty((n, 8), np.uint32)
    for k in range(4):
        rand_val[:, 2 * k] = (r[4 + k] & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        rand_val[:, 2 * k + 1] = (r[4 + k] >> np.uint64(32)).astype(np.uint32)

    cell = page.astype(np.uint64) << np.uint64(32) | index.astype(np.uint64)
    order = np.argsort(cell, kind="stable")  # (cell, time) order
    sc = cell[order]
    first_touch = np.ones(n, bool)
    first_touch[1:] = sc[1:] != sc[:-1]
    is_write_sorted = first_touch | ~want_read[order]
    # index (in sorted order) of the latest write at or before each position, within the cell
    pos = np.where(is_write_sorted, np.arange(n), -1)
    last_write = np.maximum.accumulate(pos)
    src = order[last_write]  # original index of the write whose value is current
    value = np.empty((n, 8), np.uint32)
    value[order] = rand_val[src]
    ptr = np.empty(n, bool)
    ptr[order] = is_ptr[src]
    rw = np.empty(n, bool)
    rw[order] = is_write_sorted

    q = np.zeros(n, MEM_QUERY)
    q["timestamp"] = np.arange(ts_start, ts_start + n, dtype=np.uint32)
    q["page"] = page
    q["index"] = index
    q["rw_flag"] = rw
    q["value_is_pointer"] = ptr
    q["value"] = value
    return q


def random_field_elements(seed: int, shape) -> np.ndarray:
    """Uniform canonical Goldilocks elements (rejection-free: r mod p, bias 2^-32)."""
    n = int(np.prod(shape))
    return (splitmix64(seed, n) % np.uint64(0xFFFFFFFF00000001)).reshape(shape)


LOG_QUERY = np.dtype(
    [("timestamp", "<u4"), ("tx_number_in_block", "<u2"), ("aux_byte", "u1"), ("shard_id", "u1"),
     ("address", "<u4", (5,)), ("key", "<u4", (8,)), ("read_value", "<u4", (8,)), ("written_value", "<u4", (8,)),
     ("rw_flag", "u1"), ("rollback", "u1"), ("is_service", "u1"), ("_pad", "u1")])
DECOMMIT_QUERY = np.dtype([("hash", "<u4", (8,)), ("timestamp", "<u4"), ("memory_page", "<u4"),
                           ("decommitted_length", "<u2"), ("is_fresh", "u1"), ("_pad", "u1", (5,))])


def random_log_queries(n: int, seed: int = 1) -> np.ndarray:
    """n uniformly random log records (every field exercised; no semantic validity)."""
    r = splitmix64(seed, 16 * n).reshape(16, n)
    q = np.zeros(n, LOG_QUERY)
    q["timestamp"] = (r[0] & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    q["tx_number_in_block"] = (r[0] >> np.uint64(32)).astype(np.uint16)
    q["aux_byte"] = (r[0] >> np.uint64(48)).astype(np.uint8)
    q["shard_id"] = (r[0] >> np.uint64(56)).astype(np.uint8)
    w = np.empty((n, 30), np.uint32)
    for k in range(15):
        w[:, 2 * k] = (r[1 + k] & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        w[:, 2 * k + 1] = (r[1 + k] >> np.uint64(32)).astype(np.uint32)
    q["address"], q["key"], q["read_value"], q["written_value"] = w[:, 0:5], w[:, 5:13], w[:, 13:21], w[:, 21:29]
    q["rw_flag"] = (w[:, 29] & 1).astype(np.uint8)
    q["rollback"] = ((w[:, 29] >> 1) & 1).astype(np.uint8)
    q["is_service"] = ((w[:, 29] >> 2) & 1).astype(np.uint8)
    return q


def random_decommit_queries(n: int, seed: int = 1) -> np.ndarray:
    r = splitmix64(seed, 6 * n).reshape(6, n)
    q = np.zeros(n, DECOMMIT_QUERY)
    for k in range(4):
        q["hash"][:, 2 * k] = (r[k] & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        q["hash"][:, 2 * k + 1] = (r[k] >> np.uint64(32)).astype(np.uint32)
    q["timestamp"] = (r[4] & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    q["memory_page"] = (r[4] >> np.uint64(32)).astype(np.uint32)
    q["decommitted_length"] = (r[5] & np.uint64(0xFFFF)).astype(np.uint16)
    q["is_fresh"] = ((r[5] >> np.uint64(16)) & np.uint64(1)).astype(np.uint8)
    return q


def decommit_trace(n: int, n_hashes: int, seed: int = 1) -> np.ndarray:
    """Valid decommit-request queue: n requests over n_hashes distinct bytecode hashes, timestamps strictly
    increasing in queue order, one memory page per hash, is_fresh on the first request of each hash."""
    r = splitmix64(seed, 5 * n_hashes + n).reshape(-1)
    hashes = np.zeros((n_hashes, 8), np.uint32)
    for k in range(4):
        hashes[:, 2 * k] = (r[k * n_hashes:(k + 1) * n_hashes] & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        hashes[:, 2 * k + 1] = (r[k * n_hashes:(k + 1) * n_hashes] >> np.uint64(32)).astype(np.uint32)
    hashes[:, 7] &= 0x0000FFFF  # the top bytes of a versioned hash are small
    hashes[: min(4, n_hashes), 1:] = hashes[0, 1:]  # a few hashes that differ only in the lowest limb
    pages = (8 + 8 * np.arange(n_hashes)).astype(np.uint32)
    pick = (r[5 * n_hashes:] % np.uint64(n_hashes)).astype(np.int64)
    q = np.zeros(n, DECOMMIT_QUERY)
    q["hash"] = hashes[pick]
    q["memory_page"] = pages[pick]
    q["timestamp"] = 1 + 3 * np.arange(n, dtype=np.uint32)
    q["decommitted_length"] = (1 + 2 * (pick % 100)).astype(np.uint16)
    seen = np.zeros(n_hashes, bool)
    fresh = np.zeros(n, np.uint8)
    for i, h in enumerate(pick):
        if not seen[h]:
            seen[h] = True
            fresh[i] = 1
    q["is_fresh"] = fresh
    return q


def events_trace(n_forward: int, rollback_fraction: float = 0.3, seed: int = 1) -> np.ndarray:
    """Valid event queue: n_forward events with strictly increasing timestamps (rw_flag set, shard 0); a
    fraction of them is later rolled back by a twin record (same timestamp and payload, rollback = 1) that
    sits somewhere AFTER its forward in queue order."""
    fw = random_log_queries(n_forward, seed)
    fw["timestamp"] = 10 + 2 * np.arange(n_forward, dtype=np.uint32)
    fw["shard_id"] = 0
    fw["rw_flag"] = 1
    fw["rollback"] = 0
    fw["aux_byte"] = 2
    r = splitmix64(seed + 77, 2 * n_forward)
    rolled = (r[:n_forward] >> np.uint64(11)).astype(np.float64) / float(1 << 53) < rollback_fraction
    items = [(2 * i, fw[i]) for i in range(n_forward)]
    for i in np.flatnonzero(rolled):
        tw = fw[i].copy()
        tw["rollback"] = 1
        pos = 2 * (i + int(r[n_forward + i] % np.uint64(max(1, n_forward - i)))) + 1  # after its forward
        items.append((pos, tw))
    items.sort(key=lambda t: t[0])
    return np.array([t[1] for t in items], dtype=LOG_QUERY)


def mixed_log_queue(n: int, seed: int = 1) -> np.ndarray:
    """A forward-applied log queue as the VM would leave it: storage (aux 0, shard 0), events (aux 1),
    L1 messages (aux 2) and precompile calls (aux 3: keccak 0x8010, sha256 0x02, ecrecover 0x01 and a few other
    addresses that the demuxer drops), increasing timestamps."""
    q = random_log_queries(n, seed)
    r = splitmix64(seed + 5, n)
    kind = (r % np.uint64(10)).astype(np.int64)
    q["timestamp"] = 100 + np.arange(n, dtype=np.uint32)
    q["shard_id"] = 0
    q["aux_byte"] = np.select([kind < 4, kind < 6, kind < 7], [0, 1, 2], 3)
    pre = q["aux_byte"] == 3
    q["rollback"][pre] = 0
    addr_pick = ((r >> np.uint64(8)) % np.uint64(4)).astype(np.int64)
    low = np.array([0x8010, 0x02, 0x01, 0x77], np.uint32)[addr_pick]
    q["address"][pre] = 0
    q["address"][pre, 0] = low[pre]
    return q


def storage_trace(n: int, n_cells: int, seed: int = 1, p_read: float = 0.35, p_rollback: float = 0.2) -> np.ndarray:
    """Valid rollup-storage log of n records over n_cells (address, key) cells: reads return the current
    value, writes record (previous, new), a rollback undoes the most recent pending write of its cell and
    repeats that write's (read_value, written_value). Timestamps increase; aux_byte 0, shard 0."""
    r = splitmix64(seed, 4 * n + 13 * n_cells)
    cells_w = np.zeros((n_cells, 13), np.uint32)
    for k in range(13):
        cells_w[:, k] = (r[4 * n + k * n_cells: 4 * n + (k + 1) * n_cells] & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    if n_cells > 2:
        cells_w[1, 5:] = cells_w[0, 5:]   # same key, different address
        cells_w[2, :5] = cells_w[0, :5]   # same address, different key
    pick = (r[:n] % np.uint64(n_cells)).astype(np.int64)
    u = (r[n:2 * n] >> np.uint64(11)).astype(np.float64) / float(1 << 53)
    newv = np.zeros((n, 8), np.uint32)
    newv[:, 0] = (r[2 * n:3 * n] & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    newv[:, 7] = (r[2 * n:3 * n] >> np.uint64(32)).astype(np.uint32)
    newv[:, 3] = (r[3 * n:4 * n] & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    q = np.zeros(n, LOG_QUERY)
    cur = {}      # cell -> current value
    stack = {}    # cell -> list of (read_value, written_value)
    for i in range(n):
        c = int(pick[i])
        q["address"][i] = cells_w[c, :5]
        q["key"][i] = cells_w[c, 5:]
        q["timestamp"][i] = 1000 + i
        q["tx_number_in_block"][i] = i % 7
        val = cur.get(c, np.zeros(8, np.uint32))
        st = stack.setdefault(c, [])
        if u[i] < p_read:
            q["read_value"][i] = val
            q["written_value"][i] = 0
        elif u[i] < p_read + p_rollback and st:
            rv, wv = st.pop()
            q["rw_flag"][i], q["rollback"][i] = 1, 1
            q["read_value"][i], q["written_value"][i] = rv, wv
            cur[c] = rv
        else:
            q["rw_flag"][i] = 1
            q["read_value"][i], q["written_value"][i] = val, newv[i]
            st.append((val.copy(), newv[i].copy()))
            cur[c] = newv[i].copy()
    return q


def callstack_trace(n_ops, seed=0, max_depth=40, final_unwind=True):
    """A random but valid sequence of call-stack pushes (1) / pops (0) with the pushed ExtendedCallstackEntry records."""
    from .native import CALLSTACK_ENTRY

    rng = np.random.default_rng(seed)
    ops, depth = [], 0
    for _ in range(n_ops):
        push = depth == 0 or (depth < max_depth and rng.random() < 0.55)
        ops.append(1 if push else 0)
        depth += 1 if push else -1
    if final_unwind:
        ops += [0] * depth
    ops = np.array(ops, np.uint8)
    n_push = int(ops.sum())
    e = np.zeros(n_push, CALLSTACK_ENTRY)
    raw = rng.integers(0, 256, (n_push, CALLSTACK_ENTRY.itemsize), dtype=np.uint8)
    e[:] = raw.view(CALLSTACK_ENTRY).reshape(n_push)
    e["rollback_queue_head"] = random_field_elements(seed + 1, (n_push, 4))
    e["rollback_queue_tail"] = random_field_elements(seed + 2, (n_push, 4))
    e["is_static"] &= 1
    e["is_local_frame"] &= 1
    e["_pad"] = 0
    kernel = rng.random(n_push) < 0.5  # half of the frames run in kernel space (address < 2^16)
    e["this_address"][kernel, 1:] = 0
    e["this_address"][kernel, 0] &= 0xFFFF
    return ops, e


_SHA256_K = np.array([
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
    0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
    0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
    0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
    0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
    0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2], dtype=np.uint32)
_SHA256_IV = np.array([0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19], dtype=np.uint32)


def sha256_compress_many(states, blocks):
    """FIPS 180-4 compression of N chaining states [N, 8] with N message blocks [N, 16] (big-endian words), vectorised over N"""
    rotr = lambda x, n: (x >> np.uint32(n)) | (x << np.uint32(32 - n))  # noqa: E731
    w = [blocks[:, t].astype(np.uint32) for t in range(16)]
    for t in range(16, 64):
        s0 = rotr(w[t - 15], 7) ^ rotr(w[t - 15], 18) ^ (w[t - 15] >> np.uint32(3))
        s1 = rotr(w[t - 2], 17) ^ rotr(w[t - 2], 19) ^ (w[t - 2] >> np.uint32(10))
        w.append(w[t - 16] + s0 + w[t - 7] + s1)
    a, b, c, d, e, f, g, h = (states[:, k].astype(np.uint32) for k in range(8))
    for t in range(64):
        t1 = h + (rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25)) + ((e & f) ^ (~e & g)) + _SHA256_K[t] + w[t]
        t2 = (rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22)) + ((a & b) ^ (a & c) ^ (b & c))
        h, g, f, e, d, c, b, a = g, f, e, d + t1, c, b, a, t1 + t2
    return states.astype(np.uint32) + np.stack([a, b, c, d, e, f, g, h], axis=1)


_KECCAK_RC = [0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000, 0x000000000000808B, 0x0000000080000001,
              0x8000000080008081, 0x8000000000008009, 0x000000000000008A, 0x0000000000000088, 0x0000000080008009, 0x000000008000000A,
              0x000000008000808B, 0x800000000000008B, 0x8000000000008089, 0x8000000000008003, 0x8000000000008002, 0x8000000000000080,
              0x000000000000800A, 0x800000008000000A, 0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008]
_KECCAK_ROT = [[0, 36, 3, 41, 18], [1, 44, 10, 45, 2], [62, 6, 43, 15, 61], [28, 55, 25, 21, 56], [27, 20, 39, 8, 14]]  # [x][y]


def keccak_f1600_many(a):
    """Keccak-f[1600] on N states [N, 25] of uint64 lanes (lane x + 5y), vectorised over N"""
    a = [a[:, i].copy() for i in range(25)]
    rol = lambda v, n: v if n == 0 else (v << np.uint64(n)) | (v >> np.uint64(64 - n))  # noqa: E731
    for rc in _KECCAK_RC:
        c = [a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20] for x in range(5)]
        d = [c[(x - 1) % 5] ^ rol(c[(x + 1) % 5], 1) for x in range(5)]
        a = [a[i] ^ d[i % 5] for i in range(25)]
        b = [None] * 25
        for x in range(5):
            for y in range(5):
                b[y + 5 * ((2 * x + 3 * y) % 5)] = rol(a[x + 5 * y], _KECCAK_ROT[x][y])
        a = [b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]) for y in range(5) for x in range(5)]
        a[0] = a[0] ^ np.uint64(rc)
    return np.stack(a, axis=1)


def keccak256_many(messages):
    """Keccak-256 (pad10*1 with the 0x01 domain byte, as the keccak256 precompile pads) of a list of byte strings -> [N, 32] uint8"""
    n = len(messages)
    rounds = np.array([len(m) // 136 + 1 for m in messages])
    buf = np.zeros((n, int(rounds.max()) * 136), np.uint8)
    for i, m in enumerate(messages):
        buf[i, :len(m)] = np.frombuffer(m, np.uint8)
        buf[i, len(m)] ^= 0x01
        buf[i, rounds[i] * 136 - 1] ^= 0x80
    st = np.zeros((n, 25), np.uint64)
    for r in range(int(rounds.max())):
        live = np.nonzero(rounds > r)[0]
        blk = buf[live, 136 * r:136 * (r + 1)].copy().view("<u8")
        cur = st[live]
        cur[:, :17] ^= blk
        st[live] = keccak_f1600_many(cur)
    return np.ascontiguousarray(st[:, :4]).view(np.uint8).reshape(n, 32)


def precompile_trace(kind, n_requests, seed=0, max_rounds=5):
    """Requests of one precompile (0 keccak256, 1 sha256, 2 ecrecover) with the memory queries the VM would have
    made for them, in the order the reference flattens them (reads round by round, then the write(s))."""
    from .native import LOG_QUERY, MEM_QUERY

    rng = np.random.default_rng(seed)
    req = random_log_queries(max(n_requests, 1), seed=seed + 1)[:n_requests]
    req["timestamp"] = np.sort(rng.integers(1, 1 << 30, n_requests).astype(np.uint32) * 2)
    qs = []
    sha_shape = []  # sha256: (index of the request's first query, rounds)
    keccak_shape = []  # keccak256: (index of the request's first query, reads, byte offset in the first word, length)

    def query(ts, page, index, rw):
        m = np.zeros(1, MEM_QUERY)
        m["timestamp"], m["page"], m["index"], m["rw_flag"] = ts, page, index, rw
        m["value"] = rng.integers(0, 1 << 32, 8, dtype=np.uint64).astype(np.uint32)
        return m

    for k in range(n_requests):
        ts = int(req["timestamp"][k])
        page_r, page_w = int(rng.integers(8, 1 << 20)), int(rng.integers(8, 1 << 20))
        out_off = int(rng.integers(0, 1 << 16))
        key = np.zeros(8, np.uint32)
        key[2], key[3], key[4], key[5] = out_off, 1, page_r, page_w
        if kind == 1:
            rounds = int(rng.integers(1, max_rounds + 1))
            in_off = int(rng.integers(0, 1 << 16))
            key[0], key[1], key[6] = in_off, 2 * rounds, rounds
            sha_shape.append((len(qs), rounds))
            for r in range(2 * rounds):
                qs.append(query(ts, page_r, in_off + r, 0))
            qs.append(query(ts + 1, page_w, out_off, 1))
        elif kind == 2:
            in_off = int(rng.integers(0, 1 << 16))
            key[0], key[1], key[3] = in_off, 4, 2
            # hash, v, r, s as the caller laid them out; the precompile writes (1, address) or (0, 0): real signatures, and the failures the
            # VM's precompile has (zk_evm_abstractions ecrecover: r / s out of range, x^3 + 7 not a square)
            from . import secp256k1 as ec

            h = int.from_bytes(rng.bytes(32), "big")
            sk, kk = int.from_bytes(rng.bytes(32), "big") % (ec.N - 1) + 1, int.from_bytes(rng.bytes(32), "big") % (ec.N - 1) + 1
            v, r_, s_ = ec.sign(h, sk, kk)
            mode = int(rng.integers(0, 10))
            if mode == 0:
                r_ = ec.N + int(rng.integers(0, 1000))
            elif mode == 1:
                s_ = 0
            elif mode == 2:
                r_ = int(rng.integers(1, 1 << 62))
                while ec.lift_x(r_, 0) is not None:
                    r_ += 1
            elif mode == 3:
                v = 1 - v  # the other root: another key, still a success
            ok, addr = ec.ecrecover(h, v, r_, s_)
            assert mode in (0, 1, 2) or ok == 1
            for k_, val in enumerate((h, v, r_, s_)):
                m = query(ts, page_r, in_off + k_, 0)
                m["value"] = np.frombuffer(int(val).to_bytes(32, "little"), "<u4")
                qs.append(m)
            for k_, val in enumerate((ok, addr)):
                m = query(ts + 1, page_w, out_off + k_, 1)
                m["value"] = np.frombuffer(int(val).to_bytes(32, "little"), "<u4")
                qs.append(m)
        else:
            # lengths around the interesting edges: empty, one byte short of / exactly / one past whole blocks
            choices = [0, 1, 31, 32, 33, 135, 136, 137, 271, 272, 273, 136 * max_rounds]
            length = int(rng.choice(choices)) if rng.random() < 0.6 else int(rng.integers(0, 136 * max_rounds + 1))
            in_off = int(rng.integers(0, 1 << 12))
            if rng.random() < 0.3:
                in_off = in_off // 32 * 32 + int(rng.choice([0, 31]))
            key[0], key[1] = in_off, length
            keccak_shape.append((len(qs), (in_off + length - 1) // 32 - in_off // 32 + 1 if length else 0, in_off % 32, length))
            if length:
                for wi in range(in_off // 32, (in_off + length - 1) // 32 + 1):
                    qs.append(query(ts, page_r, wi, 0))
            qs.append(query(ts + 1, page_w, out_off, 1))
        req["key"][k] = key
    mq = np.concatenate(qs) if qs else np.zeros(0, MEM_QUERY)
    if kind == 1 and sha_shape:  # the word a sha256 call writes is the chaining state after its rounds (no padding: the caller pads)
        first = np.array([a for a, _ in sha_shape])
        rounds = np.array([b for _, b in sha_shape])
        st = np.tile(_SHA256_IV, (first.size, 1))
        with np.errstate(over="ignore"):
            for r in range(int(rounds.max())):
                live = np.nonzero(rounds > r)[0]
                blocks = np.concatenate([mq["value"][first[live] + 2 * r][:, ::-1], mq["value"][first[live] + 2 * r + 1][:, ::-1]], axis=1)
                st[live] = sha256_compress_many(st[live], blocks)
        mq["value"][first + 2 * rounds] = st[:, ::-1]
    if kind == 0 and keccak_shape:  # the word a keccak256 call writes is the Keccak-256 digest of the bytes it read (U256::from_big_endian)
        msgs = []
        for first, reads, skip, length in keccak_shape:
            words = mq["value"][first:first + reads][:, ::-1].astype(">u4").tobytes()  # U256::to_big_endian of every word read
            msgs.append(words[skip:skip + length])
        dig = keccak256_many(msgs)
        at = np.array([first + reads for first, reads, _, _ in keccak_shape])
        mq["value"][at] = np.ascontiguousarray(dig[:, ::-1]).view("<u4")
    return req, mq


def storage_application_trace(n, seed=0, existing_fraction=0.5, write_fraction=0.6):
    """Deduplicated rollup storage queries (distinct slots, shard 0) for the StorageApplication circuit.
    Returns (queries, existing): `existing[i]` says slot i must already hold read_value in the tree before the block;
    the other slots are empty (read_value = 0)."""
    rng = np.random.default_rng(seed)
    q = random_log_queries(max(n, 1), seed=seed + 1)[:n]
    q["shard_id"] = 0
    q["aux_byte"] = 0
    q["rollback"] = 0
    q["is_service"] = 0
    for i in range(n):  # distinct (address, key)
        q["key"][i][0] = i
    q["rw_flag"] = (rng.random(n) < write_fraction).astype(np.uint8)
    existing = rng.random(n) < existing_fraction
    q["read_value"][~existing] = 0
    ro = q["rw_flag"] == 0
    q["written_value"][ro] = q["read_value"][ro]
    return q, existing


def bytecode_hash(words: np.ndarray) -> np.ndarray:
    """versioned bytecode hash of an odd number of 32-byte words (decommit_code.rs:47-78, 136-401): SHA-256 over the
    big-endian words, the four most significant bytes replaced by version 1 and the length in words."""
    import hashlib

    w = np.ascontiguousarray(words, dtype=np.uint32).reshape(-1, 8)
    assert w.shape[0] % 2 == 1
    code = b"".join(int(x).to_bytes(4, "big") for row in w for x in row[::-1])
    dig = hashlib.sha256(code).digest()
    out = np.zeros(8, np.uint32)
    for j in range(1, 8):
        out[7 - j] = int.from_bytes(dig[4 * j:4 * j + 4], "big")
    out[7] = 0x01000000 | w.shape[0]
    return out


def block_after_vm(seed=1, n_vm_memory=3000, n_bytecodes=5, n_decommits=12, n_storage=150, n_storage_cells=25,
                   n_events=60, n_l1_messages=25, n_precompile_calls=(5, 4, 3), total_memory=None, max_code_words=23,
                   precompile_max_rounds=5):
    """What the VM run of one block hands to the witness builders (src/witness/oracle.rs:185-927), synthesised
    consistently: the VM's memory queue incl. the writes every precompile input needs, the decommit-request queue with
    the bytecodes behind its hashes, the forward-applied log queue (storage, events, L1 messages, precompile calls,
    interleaved) and the memory queries of the three precompiles."""
    from .native import DECOMMIT_QUERY, MEM_QUERY

    rng = np.random.default_rng(seed)
    # bytecodes and the decommit queue over them
    lens = [1 + 2 * int(rng.integers(0, (max_code_words + 1) // 2)) for _ in range(n_bytecodes)]
    codes = [rng.integers(0, 1 << 32, (n, 8), dtype=np.uint64).astype(np.uint32) for n in lens]
    hashes = np.stack([bytecode_hash(c) for c in codes])
    pick = np.concatenate([np.arange(n_bytecodes), rng.integers(0, n_bytecodes, max(0, n_decommits - n_bytecodes))])
    rng.shuffle(pick)
    dq = np.zeros(pick.size, DECOMMIT_QUERY)
    dq["hash"] = hashes[pick]
    dq["memory_page"] = (100000 + 8 * pick).astype(np.uint32)
    dq["timestamp"] = 3 + 4 * np.arange(pick.size, dtype=np.uint32)
    dq["decommitted_length"] = np.array(lens, np.uint16)[pick]
    seen = set()
    for i, h in enumerate(pick):
        dq["is_fresh"][i] = 0 if int(h) in seen else 1
        seen.add(int(h))
    # precompile calls: (requests, memory queries) per kind; every input word is written by the VM one tick earlier
    pre_addr = (0x8010, 0x02, 0x01)
    pre_req, pre_mem, vm_extra = [], [], []
    used_pages = set(range(8, 8 + 64)) | {int(p) for p in dq["memory_page"]}
    for kind, n_calls in enumerate(n_precompile_calls):
        for attempt in range(50):
            req, mq = precompile_trace(kind, n_calls, seed=seed * 100 + 10 * kind + attempt, max_rounds=precompile_max_rounds)
            pages = {int(p) for p in mq["page"]}
            if not (pages & used_pages):
                break
        used_pages |= pages
        req["aux_byte"], req["shard_id"], req["rollback"] = 3, 0, 0
        req["address"] = 0
        req["address"][:, 0] = pre_addr[kind]
        pre_req.append(req)
        pre_mem.append(mq)
        rd = mq[mq["rw_flag"] == 0].copy()
        rd["rw_flag"] = 1
        rd["timestamp"] -= 1
        vm_extra.append(rd)
    if total_memory is not None:  # size the VM's part so that the whole memory queue has exactly `total_memory` items
        n_vm_memory = total_memory - sum(lens) - sum(m.size for m in pre_mem) - sum(m.size for m in vm_extra)
        assert n_vm_memory > 0
    vm_mem = np.concatenate([ram_trace(n_vm_memory, seed=seed + 3)] + vm_extra).astype(MEM_QUERY)
    # the log queue: sub-queues merged at random, each keeping its own order
    from .native import LOG_QUERY

    ev = events_trace(n_events, 0.3, seed=seed + 5) if n_events else np.zeros(0, LOG_QUERY)
    ev["aux_byte"] = 1
    l1 = events_trace(n_l1_messages, 0.2, seed=seed + 6) if n_l1_messages else np.zeros(0, LOG_QUERY)
    l1["aux_byte"] = 2
    sto = storage_trace(n_storage, n_storage_cells, seed=seed + 4) if n_storage else np.zeros(0, LOG_QUERY)
    subs = [sto, ev, l1] + pre_req
    tags = np.concatenate([np.full(s.size, k) for k, s in enumerate(subs)])
    rng.shuffle(tags)
    logs = np.zeros(tags.size, subs[0].dtype)
    for k, sub in enumerate(subs):  # sub-queue k fills, in its own order, the positions tagged k
        logs[np.nonzero(tags == k)[0]] = sub
    return {"vm_memory_queries": vm_mem, "decommit_queries": dq, "bytecodes": {h.tobytes(): c for h, c in zip(hashes, codes)},
            "log_queries": logs, "precompile_memory_queries": pre_mem}


def block_production(seed=1):
    """One block with the instance multiset of the reference's `basic_test` (file list of test_proofs/base_layer/: one
    instance of every type, two of ECRecover and StorageApplication; the three MainVM instances need the VM) with every
    builder at its PRODUCTION capacity (circuit_sequencer_api/src/geometry_config.rs:5-20): a memory queue of exactly
    136 714 queries, 117 500 decommit requests over ~2 800 SHA-256 rounds of bytecode, a log queue of ~58 000 records
    (storage over 30 slots: two StorageApplication instances of 33 tree queries, as in basic_test, events, L1 messages, keccak256 / sha256 / ecrecover calls)."""
    return block_after_vm(seed=seed, total_memory=136714, n_bytecodes=400, n_decommits=117500, n_storage=35000,
                          n_storage_cells=30, n_events=11000, n_l1_messages=700, n_precompile_calls=(60, 700, 14))


class StorageTree:
    """Host-side storage tree, the input provider of the StorageApplication builder: the counterpart of the reference's
    `ZKSyncTestingTree` / `InMemoryStorageTree<256, 32, 8, Blake2s256, ZkSyncStorageLeaf>` (src/witness/tree/mod.rs:
    113-384, handed to `run()` as `tree: impl BinarySparseStorageTree`, src/external_calls.rs:81). Depth 256, Blake2s-256;
    leaf hash = H(index as 8 big-endian bytes || value), node hash = H(left || right); bit `level` of the little-endian
    key picks the side; enumeration indices start at 1. Answers the pre-block questions zkw_block_run's storage_tree
    callback asks (get_leaf) — sequential CPU code by nature, a few hundred slots per block."""
    DEPTH = 256

    def __init__(self):
        import hashlib

        self._h = lambda b: hashlib.blake2s(b, digest_size=32).digest()
        self.next_enumeration_index = 1
        self.leaves = {}   # key int -> (index, value bytes)
        self.nodes = {}    # (level, masked key int) -> hash
        cur = self._leaf_hash(0, bytes(32))
        self.empty = []
        for _level in range(self.DEPTH):
            self.empty.append(cur)
            cur = self._h(cur + cur)
        self.root = cur

    def _leaf_hash(self, index, value):
        return self._h(int(index).to_bytes(8, "big") + value)

    def _sibling(self, k, level):
        masked = ((k ^ (1 << level)) >> level) << level
        return self.nodes.get((level, masked), self.empty[level])

    def get_leaf(self, key: bytes):
        """(enumeration index or 0, value, merkle path [256][32] with level 0 = the leaf's sibling)"""
        k = int.from_bytes(key, "little")
        index, value = self.leaves.get(k, (0, bytes(32)))
        path = np.frombuffer(b"".join(self._sibling(k, level) for level in range(self.DEPTH)), np.uint8).reshape(self.DEPTH, 32)
        return index, value, path

    def insert_leaf(self, key: bytes, value: bytes) -> int:
        k = int.from_bytes(key, "little")
        if k in self.leaves:
            index = self.leaves[k][0]
        else:
            index = self.next_enumeration_index
            self.next_enumeration_index += 1
        self.leaves[k] = (index, value)
        cur = self._leaf_hash(index, value)
        for level in range(self.DEPTH):
            self.nodes[(level, (k >> level) << level)] = cur
            sib = self._sibling(k, level)
            cur = self._h(sib + cur) if (k >> level) & 1 else self._h(cur + sib)
        self.root = cur
        return index


def derive_final_address(q) -> bytes:
    """LogQuery::derive_final_address (zk_evm v1.4.1): Blake2s-256 of the 32-byte left-padded address followed by the
    32-byte big-endian key — the storage tree's leaf key of a log query."""
    import hashlib

    addr = b"".join(int(x).to_bytes(4, "big") for x in q["address"][::-1])
    key = b"".join(int(x).to_bytes(4, "big") for x in q["key"][::-1])
    return hashlib.blake2s(bytes(12) + addr + key, digest_size=32).digest()


def storage_tree_for(dedup_queries, seed=0, extra_leaves=10):
    """A tree that holds, before the block, what the block's first access of every slot expects (read_value) plus a few
    unrelated leaves; returns (tree, answers) with answers(q) -> (leaf_indexes, merkle_paths) for the callback."""
    tree = StorageTree()
    rng = np.random.default_rng(seed)
    for _ in range(extra_leaves):
        tree.insert_leaf(rng.bytes(32), rng.bytes(32))
    for q in dedup_queries:
        if q["read_value"].any():
            tree.insert_leaf(derive_final_address(q), b"".join(int(x).to_bytes(4, "big") for x in q["read_value"][::-1]))

    def answers(q):
        idx = np.zeros(q.size, np.uint64)
        paths = np.zeros((q.size, 256, 32), np.uint8)
        for i in range(q.size):
            idx[i], _, paths[i] = tree.get_leaf(derive_final_address(q[i]))
        return idx, paths

    return tree, answers


def vm_tracer_streams(n_cycles=3000, cycles_per_snapshot=400, seed=0, n_memory=2500, sparse=200, first_snapshot_cycle=0):
    """Cycle-stamped streams of the kind WitnessTracer leaves for the MainVM slicing (src/witness/tracer.rs:221-407 records,
    oracle.rs:1229-1469 consumes): the memory stream (several queries per cycle, reads and writes), seven sparser FIFOs,
    the entry-state histories (decommit queue states, callstack sponge states, storage-log states with strictly ascending
    cycles) and the snapshot cycles. Hash-like payloads are random field elements: the slicing only moves them."""
    rng = np.random.default_rng(seed)
    p = 0xFFFFFFFF00000001

    def cycles(n, strict=False):
        c = np.sort(rng.integers(0, n_cycles, n)).astype(np.uint32)
        return np.unique(c) if strict else c

    def felts(shape):
        return (rng.integers(0, 2**63, shape, dtype=np.uint64) * 2 + rng.integers(0, 2, shape, dtype=np.uint64)) % np.uint64(p)

    mem_cycles = cycles(n_memory)
    mem = ram_trace(max(n_memory, 1), seed=seed + 1)[:n_memory]
    streams = [mem_cycles] + [cycles(int(rng.integers(0, sparse))) for _ in range(7)]
    dec_c, cs_c, sl_c = cycles(int(rng.integers(0, sparse))), cycles(int(rng.integers(0, sparse))), cycles(int(rng.integers(1, sparse)), strict=True)
    from .native import STORAGE_LOG_DETAILED_STATE

    sl = np.zeros(sl_c.size, STORAGE_LOG_DETAILED_STATE)
    for f in ("forward_tail", "rollback_head", "rollback_tail"):
        sl[f] = felts((sl_c.size, 4))
    sl["forward_length"] = rng.integers(0, 1000, sl_c.size)
    sl["rollback_length"] = rng.integers(0, 1000, sl_c.size)
    snaps = np.arange(first_snapshot_cycle, n_cycles + cycles_per_snapshot, cycles_per_snapshot, dtype=np.uint32)
    return {"snapshot_cycles": snaps, "stream_cycles": streams, "vm_memory_queries": mem, "memory_queue_tails": felts((n_memory, 12)),
            "decommit_state_cycles": dec_c, "decommit_queue_tails": felts((dec_c.size, 12)), "callstack_sponge_cycles": cs_c,
            "callstack_sponge_states": felts((cs_c.size, 12)), "storage_log_state_cycles": sl_c, "storage_log_states": sl,
            "global_end_of_storage_log": felts(4)}


VM_EVENT = np.dtype([("kind", "<u4"), ("cycle", "<u4"), ("panicked", "<u4"), ("index", "<u4")])


def vm_events(n_events=400, seed=0, max_depth=12, p_panic=0.3, p_log=0.6, first_cycle=1024):
    """What WitnessTracer hands CallstackWithAuxData over one block (src/witness/tracer.rs:221-407 ->
    callstack_handler.rs:174-460), as plain arrays: events in time order (kind 0 = log query, 1 = far/near call pushing a
    frame, 2 = ret / panic popping one), one event per VM cycle; log_queries[index] for kind 0 (storage reads / writes on
    shard 0, events, L2->L1 messages, precompile calls; unique increasing timestamps, rollback = 0); entries[2 * index] =
    the caller's frame as saved, entries[2 * index + 1] = the new frame, for kind 1. The trace starts with the bootloader
    frame's push (from_initial_callstack) and unwinds every frame at the end, some of them by panics, so that nested
    reverts of reverted frames occur."""
    from .native import CALLSTACK_ENTRY, LOG_QUERY

    rng = np.random.default_rng(seed)
    ev, depth, cycle = [], 0, first_cycle
    n_log = n_push = 0

    def emit(kind, panicked=0, index=0):
        nonlocal cycle
        ev.append((kind, cycle, panicked, index))
        cycle += int(rng.integers(1, 4))

    emit(1, 0, 0)
    n_push, depth = 1, 1
    for _ in range(n_events):
        r = rng.random()
        if r < p_log:
            emit(0, 0, n_log)
            n_log += 1
        elif depth < max_depth and (depth == 1 or rng.random() < 0.55):
            emit(1, 0, n_push)
            n_push += 1
            depth += 1
        elif depth > 1:
            emit(2, int(rng.random() < p_panic), 0)
            depth -= 1
    while depth:
        emit(2, int(depth > 1 and rng.random() < p_panic), 0)
        depth -= 1
    events = np.array(ev, VM_EVENT)
    q = random_log_queries(max(n_log, 1), seed=seed + 7)[:n_log]
    kind = rng.integers(0, 10, n_log)
    q["rollback"] = 0
    q["timestamp"] = 1024 + 4 * np.arange(n_log, dtype=np.uint32)
    q["shard_id"] = 0
    q["aux_byte"] = np.select([kind < 5, kind < 7, kind < 9], [0, 1, 2], 3).astype(np.uint8)  # storage, event, L1 message, precompile
    q["rw_flag"] = np.where(q["aux_byte"] == 0, rng.integers(0, 2, n_log), np.where(q["aux_byte"] == 3, 0, 1)).astype(np.uint8)
    pre = q["aux_byte"] == 3
    q["address"][pre] = 0
    q["address"][pre, 0] = np.array([0x8010, 0x02, 0x01, 0x7777], np.uint32)[rng.integers(0, 4, int(pre.sum()))]  # keccak, sha256, ecrecover, other
    e = np.zeros(2 * n_push, CALLSTACK_ENTRY)
    raw = rng.integers(0, 256, (2 * n_push, CALLSTACK_ENTRY.itemsize), dtype=np.uint8)
    e[:] = raw.view(CALLSTACK_ENTRY).reshape(2 * n_push)
    e["rollback_queue_head"] = 0  # filled by the replay (ExtendedCallstackEntry, oracle.rs:648-653)
    e["rollback_queue_tail"] = 0
    e["rollback_queue_segment_length"] = 0
    e["is_static"] &= 1
    e["is_local_frame"] &= 1
    e["_pad"] = 0
    return events, q, e
