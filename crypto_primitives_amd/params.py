"""Host-side parameter setup for Pedersen / Bowe-Hopwood over Jubjub (product code, no oracle import).

The reference's `setup` draws bases with `C::rand(rng)` (crh/pedersen/mod.rs:48-56,
crh/bowe_hopwood/mod.rs:45-59); that stream depends on ark-std/ark-ec and cannot be reproduced,
and `Parameters.generators` is a public field, so generators are treated as DATA: here they are
G_i = k_i * G for SplitMix64-derived scalars k_i, then the per-scheme multiples.  One-off, host-only
python big-int (the hash evaluation itself never runs here).
"""
import numpy as np

from . import field

Q = field.MODULUS
_D = (-10240 * pow(10241, -1, Q)) % Q  # ark_ed_on_bls12_381 COEFF_D; a = -1
_R = 6554484396890773809930967563523245729705921265872317281365359162392183254199  # prime subgroup order
_G = (8076246640662884909881801758704306714034609987455869804520522091855516602923,
      13262374693698910701929044844600465831413122818447359594527400194675274060458)
_M64 = (1 << 64) - 1


def _splitmix(seed):
    s = seed & _M64
    while True:
        s = (s + 0x9E3779B97F4A7C15) & _M64
        z = s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
        yield z ^ (z >> 31)


def _padd(p, q):  # projective (X:Y:Z) unified addition, a = -1 (add-2008-bbjlp)
    X1, Y1, Z1 = p
    X2, Y2, Z2 = q
    A = Z1 * Z2 % Q
    B = A * A % Q
    C = X1 * X2 % Q
    Dd = Y1 * Y2 % Q
    E = _D * C % Q * Dd % Q
    F = (B - E) % Q
    G = (B + E) % Q
    X3 = A * F % Q * ((X1 + Y1) * (X2 + Y2) - C - Dd) % Q
    Y3 = A * G % Q * (Dd + C) % Q
    return (X3, Y3, F * G % Q)


def _affine(p):
    zi = pow(p[2], -1, Q)
    return (p[0] * zi % Q, p[1] * zi % Q)


def _smul(pt, k):
    acc, base = (0, 1, 1), (pt[0], pt[1], 1)
    while k:
        if k & 1:
            acc = _padd(acc, base)
        base = _padd(base, base)
        k >>= 1
    return acc


def _bases(seed, n):
    g = _splitmix(seed)
    out = []
    for _ in range(n):
        k = 0
        while k == 0:
            while True:
                v = 0
                for i in range(4):
                    v |= next(g) << (64 * i)
                v &= (1 << 255) - 1
                if v < Q:
                    break
            k = v % _R
        out.append(_smul(_G, k))
    return out


def _generators(seed, window_size, num_windows, doublings_per_step):
    pts = []
    for base in _bases(seed, num_windows):
        cur = base
        for _ in range(window_size):
            pts.append(_affine(cur))
            for _ in range(doublings_per_step):
                cur = _padd(cur, cur)
    flat = [c for pt in pts for c in pt]
    return field.fr(flat).reshape(num_windows, window_size, 2, 4)


def pedersen_generators(seed, window_size, num_windows) -> np.ndarray:
    """generators[i][j] = 2^j * G_i (shape of crh/pedersen/mod.rs:40-56) -> [N, W, 2, 4] wire format."""
    return _generators(seed, window_size, num_windows, 1)


def bowe_hopwood_generators(seed, window_size, num_windows) -> np.ndarray:
    """generators[i][j] = 16^j * G_i (crh/bowe_hopwood/mod.rs:45-59)."""
    return _generators(seed, window_size, num_windows, 4)
