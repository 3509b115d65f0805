"""Oracle restatement of the reference Pedersen CRH (test infrastructure).

Follows crh/pedersen/mod.rs:76-129 (evaluate), :158-197 (TwoToOneCRH),
:200-209 (bytes_to_bits).  Generators are `generators[i][j]` affine tuples.
PARITY UNPINNED at value level as far as the reference goes (it holds no absolute vectors); the group law
underneath is pinned by the upstream curve crate's scalar-multiplication vector
(tests/golden/jubjub_upstream_kat.json, tests/test_oracle_curves.py); see oracle/__init__.py.
"""
from . import jubjub as jj


class InputLengthPanic(Exception):
    """stands for the reference's panic! on oversized input (pedersen/mod.rs:82-89)."""


def bytes_to_bits(data: bytes):
    """pedersen/mod.rs:200-209: bit 8k+j = bit j (LSB first) of byte k."""
    return [bool((b >> i) & 1) for b in data for i in range(8)]


def evaluate(generators, window_size, num_windows, data: bytes):
    """pedersen/mod.rs:76-129 -> affine (x, y)."""
    if len(data) * 8 > window_size * num_windows:
        raise InputLengthPanic(len(data))
    if len(data) * 8 < window_size * num_windows:  # zero-pad :91-99
        data = bytes(data) + bytes((window_size * num_windows) // 8 - len(data))
    assert len(generators) == num_windows
    bits = bytes_to_bits(data)
    pts = []
    # chunks(W).zip(generators): truncates to the shorter side (:113-124)
    n_chunks = (len(bits) + window_size - 1) // window_size
    for i in range(min(n_chunks, num_windows)):
        chunk = bits[i * window_size:(i + 1) * window_size]
        for bit, base in zip(chunk, generators[i]):
            if bit:
                pts.append(base)
    return jj.sum_points(pts)


def two_to_one_evaluate(generators, window_size, num_windows, left: bytes, right: bytes):
    """pedersen/mod.rs:158-182: zero buffer of W*N/8 bytes, copy left||right
    zip-truncated, CRH::evaluate."""
    assert len(left) == len(right)
    half_bits = (window_size * num_windows) // 2
    buf = bytearray((half_bits + half_bits) // 8)
    src = bytes(left) + bytes(right)
    n = min(len(buf), len(src))
    buf[:n] = src[:n]
    return evaluate(generators, window_size, num_windows, bytes(buf))


def two_to_one_compress(generators, window_size, num_windows, left_pt, right_pt):
    """pedersen/mod.rs:187-197: evaluate(uncompressed(left), uncompressed(right))."""
    return two_to_one_evaluate(generators, window_size, num_windows,
                               jj.serialize_uncompressed(left_pt), jj.serialize_uncompressed(right_pt))


# ---- crh/injective_map/mod.rs:16-108: Pedersen followed by TECompressor (affine point -> x) --------------------------
def compressor_evaluate(generators, window_size, num_windows, message: bytes) -> int:
    """PedersenCRHCompressor::evaluate (:54-62): injective_map(pedersen::CRH::evaluate) = its x coordinate (:24-31)"""
    return evaluate(generators, window_size, num_windows, message)[0]


def compressor_two_to_one_evaluate(generators, window_size, num_windows, left: bytes, right: bytes) -> int:
    """PedersenTwoToOneCRHCompressor::evaluate (:81-94)"""
    return two_to_one_evaluate(generators, window_size, num_windows, left, right)[0]


def compressor_two_to_one_compress(generators, window_size, num_windows, left_x: int, right_x: int) -> int:
    """:96-107: evaluate on to_uncompressed_bytes!(x) of both Fq digests (32 bytes little-endian canonical each)"""
    return compressor_two_to_one_evaluate(generators, window_size, num_windows, jj.fq_serialize(left_x), jj.fq_serialize(right_x))
