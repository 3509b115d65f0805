"""Host-side parameter setup for Pedersen / Bowe-Hopwood over Jubjub (product code, no oracle import).

The reference's `setup` draws bases with `C::rand(rng)` (crh/pedersen/mod.rs:48-56,
crh/bowe_hopwood/mod.rs:45-59); that stream depends on ark-std/ark-ec and cannot be reproduced,
and `Parameters.generators` is a public field, so generators are treated as DATA.  Two seeded procedures:

  * `setup_*_generators` (what `CRH::setup` / `Commitment::setup` use): bases sampled the way ark-ec samples a random
    twisted-Edwards point -- random y, sign bit, solve for x, clear the cofactor -- so NO discrete-log relation between
    the bases is known and the hashes are collision resistant / the commitment binding;
  * `pedersen_generators` / `bowe_hopwood_generators`: G_i = k_i * G for SplitMix64-derived scalars k_i.  Cheap and
    handy as synthetic benchmark / test data, but the k_i are public, so collisions can be computed: TEST DATA ONLY,
    never parameters for real use.

One-off, host-only python big-int (the hash evaluation itself never runs here).
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


SUBGROUP_ORDER = 6554484396890773809930967563523245729705921265872317281365359162392183254199  # r: prime order of the Jubjub subgroup


def _te_mul(pt, k):
    """k * pt for an affine point (x, y) -> affine (x, y); the identity is (0, 1)"""
    return _affine(_smul(pt, k))


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


def _sqrt(a):
    """Tonelli-Shanks in Fq (q - 1 = 2^32 * odd); None for a non-residue"""
    a %= Q
    if a == 0:
        return 0
    if pow(a, (Q - 1) // 2, Q) != 1:
        return None
    s, t = 32, (Q - 1) >> 32
    z = 2
    while pow(z, (Q - 1) // 2, Q) == 1:
        z += 1
    m, c, tt, r = s, pow(z, t, Q), pow(a, t, Q), pow(a, (t + 1) // 2, Q)
    while tt != 1:
        i, u = 0, tt
        while u != 1:
            u, i = u * u % Q, i + 1
        b = pow(c, 1 << (m - i - 1), Q)
        m, c = i, b * b % Q
        tt, r = tt * c % Q, r * b % Q
    return r


def _bases_unknown_dlog(seed, n):
    """ark-ec's `rand` for a twisted-Edwards group, driven by SplitMix64(seed): y uniform, one sign bit,
    x^2 = (y^2 - 1) / (1 + d y^2) (retry without a root), times the cofactor 8.  Projective points."""
    g = _splitmix(seed)
    out = []
    while len(out) < n:
        while True:
            v = 0
            for i in range(4):
                v |= next(g) << (64 * i)
            v &= (1 << 255) - 1
            if v < Q:
                break
        y, greatest = v, next(g) & 1
        y2 = y * y % Q
        x = _sqrt((y2 - 1) * pow(1 + _D * y2, -1, Q))
        if x is None:
            continue
        if (x > Q - x) != bool(greatest):
            x = (Q - x) % Q
        pt = (x, y, 1)
        for _ in range(3):
            pt = _padd(pt, pt)
        if pt[0] % Q != 0:  # not the identity (0 : 1 : 1)
            out.append(pt)
    return out


def _generators(seed, window_size, num_windows, doublings_per_step, bases=None):
    pts = []
    for base in (bases or _bases)(seed, num_windows):
        cur = base
        for _ in range(window_size):
            pts.append(_affine(cur))
            for _ in range(doublings_per_step):
                cur = _padd(cur, cur)
    flat = [c for pt in pts for c in pt]
    return field.fr(flat).reshape(num_windows, window_size, 2, 4)


def pedersen_generators(seed, window_size, num_windows) -> np.ndarray:
    """TEST / BENCH DATA (known discrete logs): generators[i][j] = 2^j * (k_i G) -> [N, W, 2, 4] wire format."""
    return _generators(seed, window_size, num_windows, 1)


def bowe_hopwood_generators(seed, window_size, num_windows) -> np.ndarray:
    """TEST / BENCH DATA (known discrete logs): generators[i][j] = 16^j * (k_i G)."""
    return _generators(seed, window_size, num_windows, 4)


def setup_pedersen_generators(seed, window_size, num_windows) -> np.ndarray:
    """pedersen::CRH::setup (crh/pedersen/mod.rs:40-56): generators[i][j] = 2^j * G_i, G_i random points with unknown
    discrete logs (see the module docstring)."""
    return _generators(seed, window_size, num_windows, 1, _bases_unknown_dlog)


def setup_bowe_hopwood_generators(seed, window_size, num_windows) -> np.ndarray:
    """bowe_hopwood::CRH::setup (crh/bowe_hopwood/mod.rs:45-59): generators[i][j] = 16^j * G_i, G_i as above."""
    return _generators(seed, window_size, num_windows, 4, _bases_unknown_dlog)
