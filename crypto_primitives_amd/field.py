"""Fr wire-format helpers (host side of the product; uses the C ABI, not the oracle).

Wire format = numpy uint64 [..., 4]: little-endian limbs, Montgomery form (include/akp.h).
"""
import numpy as np

from ._lib import lib, check

MODULUS = 52435875175126190479447740508185965837690552500527637822603658699938581184513  # sponge/test.rs:6
_MASK = (1 << 64) - 1


def ints_to_canonical(vals) -> np.ndarray:
    out = np.empty((len(vals), 4), dtype=np.uint64)
    for i, v in enumerate(vals):
        v = int(v) % MODULUS
        for j in range(4):
            out[i, j] = (v >> (64 * j)) & _MASK
    return out


def canonical_to_ints(arr) -> list:
    a = np.ascontiguousarray(arr, dtype=np.uint64).reshape(-1, 4)
    return [int(r[0]) | (int(r[1]) << 64) | (int(r[2]) << 128) | (int(r[3]) << 192) for r in a]


def to_mont(canonical: np.ndarray) -> np.ndarray:
    c = np.ascontiguousarray(canonical, dtype=np.uint64)
    out = np.empty_like(c)
    check(lib.akp_fr_to_mont(c.ctypes.data, out.ctypes.data, c.size // 4))
    return out


def from_mont(mont: np.ndarray) -> np.ndarray:
    m = np.ascontiguousarray(mont, dtype=np.uint64)
    out = np.empty_like(m)
    check(lib.akp_fr_from_mont(m.ctypes.data, out.ctypes.data, m.size // 4))
    return out


def fr(vals) -> np.ndarray:
    """python ints -> wire format [len, 4]."""
    return to_mont(ints_to_canonical(list(vals)))


def to_ints(mont) -> list:
    """wire format -> python ints (canonical)."""
    return canonical_to_ints(from_mont(mont))


def random_fr(n: int, seed: int) -> np.ndarray:
    """n uniform field elements in wire format from a seeded numpy generator (synthetic data for
    benches; rejection-sampled canonical values are re-interpreted as Montgomery residues, which
    is again uniform)."""
    rng = np.random.default_rng(seed)
    out = np.empty((n, 4), dtype=np.uint64)
    filled = 0
    p_limbs = [(MODULUS >> (64 * j)) & _MASK for j in range(4)]
    while filled < n:
        m = n - filled
        cand = rng.integers(0, 1 << 63, size=(m + m // 2 + 16, 4), dtype=np.uint64, endpoint=False)
        cand[:, :3] = rng.integers(0, 1 << 64, size=(cand.shape[0], 3), dtype=np.uint64, endpoint=False)
        # keep cand < p (compare the top limb, then lexicographically)
        top = cand[:, 3]
        ok = top < np.uint64(p_limbs[3])
        eq = top == np.uint64(p_limbs[3])
        if eq.any():
            for idx in np.nonzero(eq)[0]:
                v = canonical_to_ints(cand[idx])[0]
                ok[idx] = v < MODULUS
        good = cand[ok][:m]
        out[filled:filled + len(good)] = good
        filled += len(good)
    return out
