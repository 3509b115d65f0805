"""Oracle restatement of the reference Bowe-Hopwood Pedersen CRH (test infrastructure).

Follows crh/bowe_hopwood/mod.rs:114-186 (evaluate), :202-239 (TwoToOneCRH),
:81-101 (window bound).  Output = x-coordinate (canonical int).
PARITY UNPINNED at value level as far as the reference goes; the group law underneath is pinned by the upstream curve
crate's scalar-multiplication vector (tests/golden/jubjub_upstream_kat.json); see oracle/__init__.py.
"""
from . import jubjub as jj
from .pedersen import bytes_to_bits, InputLengthPanic

CHUNK_SIZE = 3  # bowe_hopwood/mod.rs:31


def max_chunks_per_segment(scalar_modulus=jj.SUBGROUP_ORDER) -> int:
    """bowe_hopwood/mod.rs:82-93 calculate_num_chunks_in_segment."""
    upper = (scalar_modulus - 1) // 2
    c, rng = 0, 2
    while rng < upper:
        rng <<= 4
        c += 1
    return c


def evaluate_point(generators, window_size, num_windows, data: bytes):
    if len(data) * 8 > window_size * num_windows * CHUNK_SIZE:
        raise InputLengthPanic(len(data))
    bits = bytes_to_bits(data)
    if len(bits) % CHUNK_SIZE:  # pad to a multiple of 3 only (:131-138)
        bits += [False] * (CHUNK_SIZE - len(bits) % CHUNK_SIZE)
    assert len(generators) == num_windows and all(len(g) == window_size for g in generators)
    seg_bits = window_size * CHUNK_SIZE
    pts = []
    n_seg = (len(bits) + seg_bits - 1) // seg_bits
    for s in range(min(n_seg, num_windows)):
        sb = bits[s * seg_bits:(s + 1) * seg_bits]
        for c in range(len(sb) // CHUNK_SIZE):
            g = generators[s][c]
            b0, b1, b2 = sb[3 * c:3 * c + 3]
            enc = g                                   # :167  zero chunk contributes +g
            if b0:
                enc = jj.add(enc, g)                  # :168-170
            if b1:
                enc = jj.add(enc, jj.double(g))       # :171-173
            if b2:
                enc = jj.neg(enc)                     # :174-176
            pts.append(enc)
    return jj.sum_points(pts)


def evaluate(generators, window_size, num_windows, data: bytes) -> int:
    """-> x coordinate (:185)."""
    return evaluate_point(generators, window_size, num_windows, data)[0]


def two_to_one_evaluate(generators, window_size, num_windows, left: bytes, right: bytes) -> int:
    """:202-227: buffer of (W*N)/8 bytes (pedersen INPUT_SIZE_BITS, no x3),
    left||right copy zip-truncated."""
    assert len(left) == len(right)
    buf = bytearray((window_size * num_windows) // 8)
    src = bytes(left) + bytes(right)
    n = min(len(buf), len(src))
    buf[:n] = src[:n]
    return evaluate(generators, window_size, num_windows, bytes(buf))


def two_to_one_compress(generators, window_size, num_windows, left_x: int, right_x: int) -> int:
    """:229-239: 32-byte LE canonical serialisation of each Fq digest."""
    return two_to_one_evaluate(generators, window_size, num_windows,
                               jj.fq_serialize(left_x), jj.fq_serialize(right_x))
