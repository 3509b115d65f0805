"""Oracle: Jubjub (ark_ed_on_bls12_381) twisted-Edwards group law on python ints.

TEST INFRASTRUCTURE.  The curve arithmetic the reference path rests on lives in
ark-ec / ark-ed-on-bls12-381 (git-patched, un-vendored: /root/reference/Cargo.toml:46-58),
so it is restated here from the published curve definition:
    -x^2 + y^2 = 1 + d x^2 y^2  over Fq = BLS12-381 Fr,  d = -(10240/10241),
prime-order subgroup r, cofactor 8, identity (0, 1).
Call sites in the reference: crh/pedersen/mod.rs:50-53,116-128,
crh/bowe_hopwood/mod.rs:49-54,167-185.
Points are affine tuples (x, y) of canonical ints; the unified complete addition
law is used so results are mathematically determined (canonical affine output).
"""
from .fr import P as Q, SplitMix64

A = Q - 1
D = (-10240 * pow(10241, -1, Q)) % Q
assert D == 19257038036680949359750312669786877991949435402254120286184196891950884077233
SUBGROUP_ORDER = 6554484396890773809930967563523245729705921265872317281365359162392183254199
COFACTOR = 8
IDENTITY = (0, 1)
# ark-ed-on-bls12-381 prime-subgroup generator
GENERATOR = (
    8076246640662884909881801758704306714034609987455869804520522091855516602923,
    13262374693698910701929044844600465831413122818447359594527400194675274060458,
)


def is_on_curve(pt) -> bool:
    x, y = pt
    x2, y2 = x * x % Q, y * y % Q
    return (A * x2 + y2) % Q == (1 + D * x2 % Q * y2) % Q


# ---- extended coordinates (X, Y, Z, T), x = X/Z, y = Y/Z, T = XY/Z -------------
def _to_ext(pt):
    x, y = pt
    return (x, y, 1, x * y % Q)


def _ext_add(p1, p2):
    # unified add-2008-hwcd (complete for a = -1 with non-square d)
    X1, Y1, Z1, T1 = p1
    X2, Y2, Z2, T2 = p2
    a_ = X1 * X2 % Q
    b_ = Y1 * Y2 % Q
    c_ = D * T1 % Q * T2 % Q
    d_ = Z1 * Z2 % Q
    e_ = ((X1 + Y1) * (X2 + Y2) - a_ - b_) % Q
    f_ = (d_ - c_) % Q
    g_ = (d_ + c_) % Q
    h_ = (b_ - A * a_) % Q
    return (e_ * f_ % Q, g_ * h_ % Q, f_ * g_ % Q, e_ * h_ % Q)


def _to_affine(pe):
    X, Y, Z, _ = pe
    zi = pow(Z, -1, Q)
    return (X * zi % Q, Y * zi % Q)


def add(p1, p2):
    return _to_affine(_ext_add(_to_ext(p1), _to_ext(p2)))


def neg(pt):
    return ((-pt[0]) % Q, pt[1])


def double(pt):
    return add(pt, pt)


def mul(pt, k: int):
    """scalar multiplication k*pt (double-and-add, extended coordinates)."""
    if k < 0:
        return mul(neg(pt), -k)
    acc = _to_ext(IDENTITY)
    base = _to_ext(pt)
    while k:
        if k & 1:
            acc = _ext_add(acc, base)
        base = _ext_add(base, base)
        k >>= 1
    return _to_affine(acc)


def sum_points(pts):
    acc = _to_ext(IDENTITY)
    for p in pts:
        acc = _ext_add(acc, _to_ext(p))
    return _to_affine(acc)


def serialize_uncompressed(pt) -> bytes:
    """ark-serialize uncompressed TE affine = x (32 B LE canonical) || y (32 B LE).
    Inferred from Window4x256 sizing (merkle_tree/tests/mod.rs:13-17 +
    crh/pedersen/mod.rs:174); unpinned by any byte-level reference test."""
    return pt[0].to_bytes(32, "little") + pt[1].to_bytes(32, "little")


def fq_serialize(x: int) -> bytes:
    return x.to_bytes(32, "little")


# ---- seeded generators: generators are INPUT DATA -----------
def seeded_bases(seed: int, n: int):
    """G_i = k_i * G with k_i = SplitMix64-derived scalar mod r (k_i != 0).  TEST DATA ONLY: the discrete logs between
    these bases are known, so hashes over them are not collision resistant; `random_bases` is what `setup` uses."""
    rng = SplitMix64(seed)
    out = []
    for _ in range(n):
        k = 0
        while k == 0:
            k = rng.fr() % SUBGROUP_ORDER
        out.append(mul(GENERATOR, k))
    return out


def fq_sqrt(a: int):
    """square root in Fq by Tonelli-Shanks (q - 1 = 2^32 * odd), or None for a non-residue"""
    a %= Q
    if a == 0:
        return 0
    if pow(a, (Q - 1) // 2, Q) != 1:
        return None
    s, t = 0, Q - 1
    while t % 2 == 0:
        s, t = s + 1, t // 2
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


def random_bases(seed: int, n: int):
    """Bases with NO known discrete-log relation, the way ark-ec samples `C::rand(rng)` for a twisted-Edwards group
    (crh/pedersen/mod.rs:50, crh/bowe_hopwood/mod.rs:49): draw y and a sign bit, solve the curve equation for x
    (x^2 = (y^2 - 1) / (1 + d y^2)), retry when there is no root, clear the cofactor.  The draws come from SplitMix64(seed)
    (the reference's rng stream is not reproducible), so the result is a public, checkable function of the seed."""
    rng = SplitMix64(seed)
    out = []
    while len(out) < n:
        y = rng.fr()
        greatest = rng.next() & 1
        y2 = y * y % Q
        x = fq_sqrt((y2 - 1) * pow(1 + D * y2, -1, Q))
        if x is None:
            continue
        if (x > Q - x) != bool(greatest):
            x = (Q - x) % Q
        pt = mul((x, y), COFACTOR)
        if pt != IDENTITY:
            out.append(pt)
    return out


def pedersen_generators(seed: int, window_size: int, num_windows: int, bases=seeded_bases):
    """shape of crh/pedersen/mod.rs:40-56: generators[i][j] = 2^j * G_i."""
    gens = []
    for base in bases(seed, num_windows):
        row = []
        cur = base
        for _ in range(window_size):
            row.append(cur)
            cur = double(cur)
        gens.append(row)
    return gens


def bowe_hopwood_generators(seed: int, window_size: int, num_windows: int, bases=seeded_bases):
    """shape of crh/bowe_hopwood/mod.rs:45-59: generators[i][j] = 16^j * G_i."""
    gens = []
    for base in bases(seed, num_windows):
        row = []
        cur = base
        for _ in range(window_size):
            row.append(cur)
            for _ in range(4):
                cur = double(cur)
        gens.append(row)
    return gens
