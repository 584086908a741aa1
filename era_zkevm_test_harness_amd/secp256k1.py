"""secp256k1 / ecrecover in plain Python integers: test-data generation (synthetic.precompile_trace signs real messages so that
the ECRecover circuit's write queries hold real results) and the reference semantics tools/gen_ecrecover_circuit.py checks its
netlist against. Semantics of the VM precompile the reference replays (zk_evm_abstractions ecrecover, era-zk_evm v1.4.1, absent
from /root/reference; src/witness/individual_circuits/ecrecover.rs:143-178 only moves its 4 reads + 2 writes): r, s in [1, n),
v = 0 / 1 the parity of R.y, Q = r^-1 (s R - h G); success -> (1, keccak256(Q.x || Q.y)[12..]) else (0, 0)."""

P = 2**256 - 2**32 - 977
N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
GX = 0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798
GY = 0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8
G = (GX, GY)
assert (GY * GY - GX**3 - 7) % P == 0


def add(a, b):
    """affine addition; None = the point at infinity"""
    if a is None:
        return b
    if b is None:
        return a
    if a[0] == b[0]:
        if (a[1] + b[1]) % P == 0:
            return None
        lam = 3 * a[0] * a[0] * pow(2 * a[1], -1, P) % P
    else:
        lam = (b[1] - a[1]) * pow(b[0] - a[0], -1, P) % P
    x = (lam * lam - a[0] - b[0]) % P
    return x, (lam * (a[0] - x) - a[1]) % P


def neg(a):
    return None if a is None else (a[0], (-a[1]) % P)


def mul(k, a):
    r = None
    while k:
        if k & 1:
            r = add(r, a)
        a = add(a, a)
        k >>= 1
    return r


def lift_x(x, odd):
    """the curve point with this x and the given parity of y, or None when x^3 + 7 is not a square"""
    t = (x * x * x + 7) % P
    y = pow(t, (P + 1) // 4, P)
    if y * y % P != t:
        return None
    return (x, y if (y & 1) == odd else P - y)


_RC = [0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000, 0x000000000000808B, 0x0000000080000001,
       0x8000000080008081, 0x8000000000008009, 0x000000000000008A, 0x0000000000000088, 0x0000000080008009, 0x000000008000000A,
       0x000000008000808B, 0x800000000000008B, 0x8000000000008089, 0x8000000000008003, 0x8000000000008002, 0x8000000000000080,
       0x000000000000800A, 0x800000008000000A, 0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008]
_ROT = [0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14]


def keccak256(msg: bytes) -> bytes:
    m = bytearray(msg) + b"\x01" + b"\x00" * ((-len(msg) - 2) % 136) + b"\x80" if (len(msg) + 1) % 136 else bytearray(msg) + b"\x81"
    a = [0] * 25
    mask = 2**64 - 1
    rol = lambda v, n: ((v << n) | (v >> (64 - n))) & mask if n else v  # noqa: E731
    for off in range(0, len(m), 136):
        for i in range(17):
            a[i] ^= int.from_bytes(m[off + 8 * i:off + 8 * i + 8], "little")
        for rnd in range(24):
            c = [a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20] for x in range(5)]
            d = [c[(x - 1) % 5] ^ rol(c[(x + 1) % 5], 1) for x in range(5)]
            a = [a[i] ^ d[i % 5] for i in range(25)]
            b = [0] * 25
            for x in range(5):
                for y in range(5):
                    b[y + 5 * ((2 * x + 3 * y) % 5)] = rol(a[x + 5 * y], _ROT[x + 5 * y])
            a = [b[i] ^ (~b[(i % 5 + 1) % 5 + 5 * (i // 5)] & b[(i % 5 + 2) % 5 + 5 * (i // 5)] & mask) for i in range(25)]
            a[0] ^= _RC[rnd]
    return b"".join(v.to_bytes(8, "little") for v in a[:4])


def ecrecover(h: int, v: int, r: int, s: int):
    """(ok, address as int) of the precompile: h, r, s as 256-bit integers, v = 0 / 1"""
    if not (0 < r < N and 0 < s < N) or v not in (0, 1):
        return 0, 0
    R = lift_x(r, v)
    if R is None:
        return 0, 0
    ri = pow(r, -1, N)
    q = add(mul(s * ri % N, R), neg(mul(h * ri % N, G)))
    if q is None:
        return 0, 0
    return 1, int.from_bytes(keccak256(q[0].to_bytes(32, "big") + q[1].to_bytes(32, "big"))[12:], "big")


def sign(h: int, key: int, k: int):
    """(v, r, s) with the nonce k (no low-s normalisation: both forms recover)"""
    R = mul(k, G)
    r = R[0] % N
    s = pow(k, -1, N) * (h + r * key) % N
    assert r and s and R[0] < N
    return R[1] & 1, r, s


def address_of_key(key: int) -> int:
    q = mul(key, G)
    return int.from_bytes(keccak256(q[0].to_bytes(32, "big") + q[1].to_bytes(32, "big"))[12:], "big")
