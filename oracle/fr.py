"""BLS12-381 scalar field Fr (= Jubjub base field Fq) helpers for the oracle.

Modulus literal: reference sponge/test.rs:6.  ark-ff keeps Fp as 4 x u64
little-endian limbs in Montgomery form (x * 2^256 mod p), fully reduced; that
is the wire format of the C ABI (include/akp.h), so the helpers here convert
python ints <-> that limb layout with numpy.
"""
import numpy as np

P = 52435875175126190479447740508185965837690552500527637822603658699938581184513
R = 1 << 256
R_MOD_P = R % P
R_INV = pow(R, -1, P)
MASK64 = (1 << 64) - 1


def to_mont(x: int) -> int:
    return (x * R_MOD_P) % P


def from_mont(x: int) -> int:
    return (x * R_INV) % P


def int_to_limbs(x: int):
    return [(x >> (64 * i)) & MASK64 for i in range(4)]


def limbs_to_int(l) -> int:
    return int(l[0]) | (int(l[1]) << 64) | (int(l[2]) << 128) | (int(l[3]) << 192)


def ints_to_mont_array(vals) -> np.ndarray:
    """list of canonical ints -> uint64 array [len, 4] of Montgomery limbs."""
    out = np.empty((len(vals), 4), dtype=np.uint64)
    for i, v in enumerate(vals):
        m = to_mont(v % P)
        for j in range(4):
            out[i, j] = (m >> (64 * j)) & MASK64
    return out


def mont_array_to_ints(arr) -> list:
    """uint64 array [..., 4] of Montgomery limbs -> flat list of canonical ints."""
    a = np.asarray(arr, dtype=np.uint64).reshape(-1, 4)
    return [from_mont(limbs_to_int(row)) for row in a]


def ints_to_canon_array(vals) -> np.ndarray:
    out = np.empty((len(vals), 4), dtype=np.uint64)
    for i, v in enumerate(vals):
        for j in range(4):
            out[i, j] = (v >> (64 * j)) & MASK64
    return out


def canon_array_to_ints(arr) -> list:
    a = np.asarray(arr, dtype=np.uint64).reshape(-1, 4)
    return [limbs_to_int(row) for row in a]


class SplitMix64:
    """Synthetic-input PRNG: the reference's test_rng()
    stream is not reproducible without ark-std, so inputs are our own."""

    def __init__(self, seed: int):
        self.s = seed & MASK64

    def next(self) -> int:
        self.s = (self.s + 0x9E3779B97F4A7C15) & MASK64
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
        return z ^ (z >> 31)

    def fr(self) -> int:
        """uniform canonical Fr: 4 limbs, clear top bit, reject if >= p."""
        while True:
            v = 0
            for i in range(4):
                v |= self.next() << (64 * i)
            v &= (1 << 255) - 1
            if v < P:
                return v

    def bytes(self, n: int) -> bytes:
        out = bytearray()
        while len(out) < n:
            out += self.next().to_bytes(8, "little")
        return bytes(out[:n])
