"""ark-serialize-compatible byte encodings of the objects that cross the boundary.

`#[derive(CanonicalSerialize)]` writes the fields in declaration order; `usize` as u64 LE; `Vec<T>` as a u64 LE
length followed by the elements; `Fp` as 32 bytes little-endian canonical (non-Montgomery); a twisted-Edwards
affine point uncompressed as x || y.  Structs covered: PoseidonConfig (sponge/poseidon/mod.rs:26-45),
pedersen / bowe_hopwood Parameters (crh/pedersen/mod.rs:28-31, crh/bowe_hopwood/mod.rs:33-37), Path
(merkle_tree/mod.rs:139-156), MultiPath (:239-257).  The byte layouts are inferred from ark-serialize's published
conventions -- the reference holds no byte-level vectors for them (unpinned, like the digest encoding).
Host-side glue; field conversions go through the C ABI (field.py).

Both modes of `CanonicalSerialize` are provided (`compress=False / True`): field elements and lengths are identical in
the two; a twisted-Edwards affine point compresses to y (32 bytes LE) with the sign of x in the top bit of the last
byte -- set iff x is the lexicographically larger of (x, -x), i.e. x > (p - 1) / 2 (ark-ec's `TEFlags`), possible
because y < 2^255.  Decompression solves x^2 = (y^2 - 1) / (1 + d y^2) for Jubjub (a = -1) and picks the root the flag
names.  This, too, is restated from ark-ec's published behaviour and pinned by nothing the reference holds.
"""
import struct

import numpy as np

from . import field


def _u64(v):
    return struct.pack("<Q", int(v))


def fr_bytes(wire) -> bytes:
    """wire-format element(s) -> canonical 32-byte LE each"""
    return field.from_mont(np.ascontiguousarray(wire, dtype=np.uint64)).astype("<u8").tobytes()


def fr_from_bytes(b: bytes, n: int) -> np.ndarray:
    if len(b) < 32 * n:
        raise ValueError("truncated input: %d field elements need %d bytes, got %d" % (n, 32 * n, len(b)))
    c = np.frombuffer(b[: 32 * n], dtype="<u8").reshape(n, 4).astype(np.uint64)
    return field.to_mont(c)  # raises on a non-canonical element (>= p), as ark-serialize's Fp deserialisation does


class _Reader:
    """bounds-checked cursor: a truncated or over-long length prefix raises ValueError (ark-serialize returns
    SerializationError::InvalidData / an io error there), never an IndexError or a silent short read"""

    def __init__(self, b):
        self.b, self.o = memoryview(b), 0

    def take(self, nbytes):
        if nbytes < 0 or self.o + nbytes > len(self.b):
            raise ValueError("truncated input: %d bytes wanted at offset %d, %d left" % (nbytes, self.o, len(self.b) - self.o))
        out = bytes(self.b[self.o: self.o + nbytes])
        self.o += nbytes
        return out

    def u64(self):
        return struct.unpack("<Q", self.take(8))[0]

    def count(self, item_bytes):
        """a Vec length prefix, checked against what is left so that a corrupt length cannot ask for gigabytes"""
        n = self.u64()
        if n * item_bytes > len(self.b) - self.o:
            raise ValueError("length prefix %d does not fit the %d bytes left" % (n, len(self.b) - self.o))
        return n

    def fr(self, n):
        return fr_from_bytes(self.take(32 * n), n)


def serialize_poseidon_config(cfg) -> bytes:
    t = cfg.rate + cfg.capacity
    out = [_u64(cfg.full_rounds), _u64(cfg.partial_rounds), _u64(cfg.alpha), _u64(cfg.ark.shape[0])]
    for row in cfg.ark:
        out += [_u64(t), fr_bytes(row)]
    out.append(_u64(t))
    for row in cfg.mds:
        out += [_u64(t), fr_bytes(row)]
    out += [_u64(cfg.rate), _u64(cfg.capacity)]
    return b"".join(out)


def deserialize_poseidon_config(b: bytes):
    from .sponge.poseidon import PoseidonConfig
    r = _Reader(b)
    rf, rp, alpha = r.u64(), r.u64(), r.u64()
    ark = [r.fr(r.count(32)) for _ in range(r.count(8))]
    mds = [r.fr(r.count(32)) for _ in range(r.count(8))]
    rate, cap = r.u64(), r.u64()
    return PoseidonConfig(rf, rp, alpha, np.stack(ark), np.stack(mds), rate, cap)


def te_points_bytes(points_wire, compress=False) -> bytes:
    """affine points [n, 2, 4] (wire format) -> x || y each (uncompressed) or y with the x-sign flag (compressed)"""
    pts = np.ascontiguousarray(points_wire, dtype=np.uint64).reshape(-1, 2, 4)
    if not compress:
        return fr_bytes(pts.reshape(-1, 4))
    out = bytearray()
    for x, y in zip(field.to_ints(pts[:, 0]), field.to_ints(pts[:, 1])):
        b = bytearray(int(y).to_bytes(32, "little"))
        if x > field.MODULUS - x:  # TEFlags::XIsNegative
            b[31] |= 0x80
        out += b
    return bytes(out)


def _validate_points(flat_xy):
    """ark-serialize's default `Validate::Yes` for twisted-Edwards affine points (flat [x0, y0, x1, y1, ...] canonical ints): on
    the curve -x^2 + y^2 = 1 + d x^2 y^2 and in the prime-order subgroup (r * P = O).  Off-curve or small-order generators would
    otherwise go straight into the GPU tables."""
    from .params import _D, SUBGROUP_ORDER, _te_mul
    q = field.MODULUS
    for i in range(0, len(flat_xy), 2):
        x, y = int(flat_xy[i]), int(flat_xy[i + 1])
        x2, y2 = x * x % q, y * y % q
        if (y2 - x2 - 1 - _D * x2 % q * y2) % q:
            raise ValueError("point %d is not on the curve" % (i // 2))
        if _te_mul((x, y), SUBGROUP_ORDER) != (0, 1):
            raise ValueError("point %d is not in the prime-order subgroup" % (i // 2))


def te_points_from_bytes(b: bytes, n: int, compress=False, validate=True) -> np.ndarray:
    """inverse of te_points_bytes -> [n, 2, 4] wire format.  Raises ValueError on truncated input, on a compressed y with no
    point on the curve and -- with validate (the default, ark-serialize's `Validate::Yes`) -- on a point that is not on the
    curve or not in the prime-order subgroup.  validate=False is `deserialize_*_unchecked`."""
    per = 32 if compress else 64
    if len(b) < per * n:
        raise ValueError("truncated input: %d points need %d bytes, got %d" % (n, per * n, len(b)))
    if not compress:
        pts = fr_from_bytes(b, 2 * n).reshape(n, 2, 4)
        if validate:
            _validate_points(field.to_ints(pts.reshape(-1, 4)))
        return pts
    from .params import _sqrt, _D
    q = field.MODULUS
    vals = []
    for i in range(n):
        raw = bytearray(b[32 * i: 32 * i + 32])
        neg = bool(raw[31] & 0x80)
        raw[31] &= 0x7F
        y = int.from_bytes(raw, "little")
        if y >= q:
            raise ValueError("non-canonical y coordinate")
        y2 = y * y % q
        x = _sqrt((y2 - 1) * pow(1 + _D * y2, -1, q))
        if x is None:
            raise ValueError("y is not the coordinate of a point of the curve")
        if (x > q - x) != neg:
            x = (q - x) % q
        vals += [x, y]
    if validate:
        _validate_points(vals)
    return field.fr(vals).reshape(n, 2, 4)


def serialize_te_parameters(params, compress=False) -> bytes:
    """Parameters { generators: Vec<Vec<C>> }: projective points serialise as their affine form"""
    out = [_u64(params.num_windows)]
    for row in params.generators:
        out += [_u64(params.window_size), te_points_bytes(row, compress)]
    return b"".join(out)


def deserialize_te_parameters(b: bytes, cls, compress=False, validate=True):
    r = _Reader(b)
    rows = []
    per = 32 if compress else 64
    for _ in range(r.count(8)):
        w = r.count(per)
        rows.append(te_points_from_bytes(r.take(per * w), w, compress, validate))
    if not rows or any(len(x) != len(rows[0]) for x in rows):
        raise ValueError("generators must be a non-empty rectangular Vec<Vec<_>>")
    return cls(np.stack(rows))


def _digest_bytes(d, compress=False) -> bytes:
    d = np.asarray(d, dtype=np.uint64)
    if compress and d.shape[-2:] == (2, 4):  # an affine point (Pedersen digest)
        return te_points_bytes(d, True)
    return fr_bytes(d.reshape(-1, 4))


def _read_digest(r, config, compress):
    if config.digest_shape == (2, 4):
        return te_points_from_bytes(r.take(32 if compress else 64), 1, compress)[0]
    return r.fr(1).reshape(config.digest_shape)


def serialize_path(path, compress=False) -> bytes:
    """Path { leaf_sibling_hash, auth_path: Vec<InnerDigest>, leaf_index: usize }"""
    out = [_digest_bytes(path.leaf_sibling_hash, compress), _u64(len(path.auth_path))]
    out += [_digest_bytes(a, compress) for a in path.auth_path]
    out.append(_u64(path.leaf_index))
    return b"".join(out)


def deserialize_path(b: bytes, config, compress=False):
    from .merkle_tree import Path
    r = _Reader(b)
    sib = _read_digest(r, config, compress)
    auth = [_read_digest(r, config, compress) for _ in range(r.count(32))]
    return Path(config, sib, auth, r.u64())


def serialize_multi_path(mp, compress=False) -> bytes:
    """MultiPath { leaf_siblings_hashes, auth_paths_prefix_lenghts, auth_paths_suffixes, leaf_indexes }"""
    out = [_u64(len(mp.leaf_siblings_hashes))] + [_digest_bytes(d, compress) for d in mp.leaf_siblings_hashes]
    out += [_u64(len(mp.auth_paths_prefix_lenghts))] + [_u64(v) for v in mp.auth_paths_prefix_lenghts]
    out.append(_u64(len(mp.auth_paths_suffixes)))
    for suf in mp.auth_paths_suffixes:
        out += [_u64(len(suf))] + [_digest_bytes(d, compress) for d in suf]
    out += [_u64(len(mp.leaf_indexes))] + [_u64(v) for v in mp.leaf_indexes]
    return b"".join(out)


def deserialize_multi_path(b: bytes, config, compress=False):
    from .merkle_tree import MultiPath
    r = _Reader(b)
    sibs = [_read_digest(r, config, compress) for _ in range(r.count(32))]
    pre = [r.u64() for _ in range(r.count(8))]
    suf = [[_read_digest(r, config, compress) for _ in range(r.count(32))] for _ in range(r.count(8))]
    idx = [r.u64() for _ in range(r.count(8))]
    return MultiPath(config, sibs, pre, suf, idx)
