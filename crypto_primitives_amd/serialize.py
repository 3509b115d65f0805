"""ark-serialize-compatible byte encodings of the objects that cross the boundary.

`#[derive(CanonicalSerialize)]` writes the fields in declaration order; `usize` as u64 LE; `Vec<T>` as a u64 LE
length followed by the elements; `Fp` as 32 bytes little-endian canonical (non-Montgomery); a twisted-Edwards
affine point uncompressed as x || y.  Structs covered: PoseidonConfig (sponge/poseidon/mod.rs:26-45),
pedersen / bowe_hopwood Parameters (crh/pedersen/mod.rs:28-31, crh/bowe_hopwood/mod.rs:33-37), Path
(merkle_tree/mod.rs:139-156), MultiPath (:239-257).  The byte layouts are inferred from ark-serialize's published
conventions -- the reference holds no byte-level vectors for them (unpinned, like the digest encoding).
Host-side glue; field conversions go through the C ABI (field.py).
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
    c = np.frombuffer(b[: 32 * n], dtype="<u8").reshape(n, 4).astype(np.uint64)
    return field.to_mont(c)


class _Reader:
    def __init__(self, b):
        self.b, self.o = memoryview(b), 0

    def u64(self):
        v = struct.unpack_from("<Q", self.b, self.o)[0]
        self.o += 8
        return v

    def fr(self, n):
        out = fr_from_bytes(bytes(self.b[self.o: self.o + 32 * n]), n)
        self.o += 32 * n
        return out


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
    ark = [r.fr(r.u64()) for _ in range(r.u64())]
    mds = [r.fr(r.u64()) for _ in range(r.u64())]
    rate, cap = r.u64(), r.u64()
    return PoseidonConfig(rf, rp, alpha, np.stack(ark), np.stack(mds), rate, cap)


def serialize_te_parameters(params) -> bytes:
    """Parameters { generators: Vec<Vec<C>> }, points uncompressed (x || y)."""
    out = [_u64(params.num_windows)]
    for row in params.generators:
        out += [_u64(params.window_size), fr_bytes(row.reshape(-1, 4))]
    return b"".join(out)


def deserialize_te_parameters(b: bytes, cls):
    r = _Reader(b)
    rows = []
    for _ in range(r.u64()):
        w = r.u64()
        rows.append(r.fr(2 * w).reshape(w, 2, 4))
    return cls(np.stack(rows))


def _digest_bytes(d) -> bytes:
    return fr_bytes(np.asarray(d, dtype=np.uint64).reshape(-1, 4))


def serialize_path(path) -> bytes:
    """Path { leaf_sibling_hash, auth_path: Vec<InnerDigest>, leaf_index: usize }"""
    out = [_digest_bytes(path.leaf_sibling_hash), _u64(len(path.auth_path))]
    out += [_digest_bytes(a) for a in path.auth_path]
    out.append(_u64(path.leaf_index))
    return b"".join(out)


def deserialize_path(b: bytes, config):
    from .merkle_tree import Path
    fe = 2 if config.digest_shape == (2, 4) else 1
    r = _Reader(b)
    sib = r.fr(fe).reshape(config.digest_shape)
    auth = [r.fr(fe).reshape(config.digest_shape) for _ in range(r.u64())]
    return Path(config, sib, auth, r.u64())


def serialize_multi_path(mp) -> bytes:
    """MultiPath { leaf_siblings_hashes, auth_paths_prefix_lenghts, auth_paths_suffixes, leaf_indexes }"""
    out = [_u64(len(mp.leaf_siblings_hashes))] + [_digest_bytes(d) for d in mp.leaf_siblings_hashes]
    out += [_u64(len(mp.auth_paths_prefix_lenghts))] + [_u64(v) for v in mp.auth_paths_prefix_lenghts]
    out.append(_u64(len(mp.auth_paths_suffixes)))
    for suf in mp.auth_paths_suffixes:
        out += [_u64(len(suf))] + [_digest_bytes(d) for d in suf]
    out += [_u64(len(mp.leaf_indexes))] + [_u64(v) for v in mp.leaf_indexes]
    return b"".join(out)


def deserialize_multi_path(b: bytes, config):
    from .merkle_tree import MultiPath
    fe = 2 if config.digest_shape == (2, 4) else 1
    r = _Reader(b)
    sibs = [r.fr(fe).reshape(config.digest_shape) for _ in range(r.u64())]
    pre = [r.u64() for _ in range(r.u64())]
    suf = [[r.fr(fe).reshape(config.digest_shape) for _ in range(r.u64())] for _ in range(r.u64())]
    idx = [r.u64() for _ in range(r.u64())]
    return MultiPath(config, sibs, pre, suf, idx)
