"""ark-serialize-compatible byte encodings of the objects that cross the boundary.

Python mirror of the C ABI's `akp_serialize_*` / `akp_deserialize_*` (include/akp.h, csrc/capi_serialize.hip): every byte is
written and parsed by the library, so a Rust / C++ host and this module cannot disagree.  Structs covered: PoseidonConfig
(sponge/poseidon/mod.rs:26-45), pedersen / bowe_hopwood Parameters (crh/pedersen/mod.rs:28-31,
crh/bowe_hopwood/mod.rs:33-37), Path (merkle_tree/mod.rs:139-152), MultiPath (:239-254), both `Compress` modes.

`#[derive(CanonicalSerialize)]` writes the fields in declaration order; `usize` as u64 LE; `Vec<T>` as a u64 LE length
followed by the elements; `Fp` as 32 bytes little-endian canonical (non-Montgomery); a twisted-Edwards affine point as
x || y (uncompressed) or as y with the sign of x in the top bit of the last byte (compressed: set iff x > (p - 1) / 2, ark-ec's
`TEFlags`).  The formats are restated from ark-serialize's published conventions; the independent restatement they are
checked against is `oracle/serialize.py` (tests/test_serialize_cpu.py, tests/test_abi.py); byte-level reference vectors
exist only once `shim/examples/emit_vectors.rs` has been run (unpinned until then).

Readers validate like `Validate::Yes` by default (canonical field elements always; points on the curve and in the prime-order
subgroup, in C++: about 0.2 ms per point); `validate=False` is `deserialize_*_unchecked`.  Errors are ValueError.
"""
import ctypes as C

import numpy as np

from ._lib import lib, check, AkpError


def _ser(call):
    """size query, then the real call"""
    n = C.c_size_t()
    check(call(None, 0, C.byref(n)))
    buf = (C.c_uint8 * max(n.value, 1))()
    check(call(buf, n.value, C.byref(n)))
    return bytes(buf[: n.value])


def _de(rc):
    try:
        check(rc)
    except AkpError as e:  # IncorrectInputLength (truncated / trailing bytes) is an AkpError too
        raise ValueError(str(e)) from None


def _in(b):
    b = bytes(b)
    return (C.c_uint8 * max(len(b), 1)).from_buffer_copy(b if b else b"\0"), len(b)


def _wire(a, fe):
    return np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, fe, 4)


def _fe_of(config):
    return 2 if tuple(config.digest_shape) == (2, 4) else 1


def digests_bytes(digests, fe, compress=False) -> bytes:
    """digests [n, fe, 4] wire format -> their encodings back to back (no length prefix)"""
    d = _wire(digests, fe)
    return _ser(lambda out, cap, n: lib.akp_serialize_digests(d.ctypes.data, d.shape[0], fe, int(compress), out, cap, n))


def digests_from_bytes(b, n, fe, compress=False, validate=True) -> np.ndarray:
    buf, ln = _in(b)
    out = np.empty((n, fe, 4), dtype=np.uint64)
    _de(lib.akp_deserialize_digests(buf, ln, n, fe, int(compress), int(validate), out.ctypes.data))
    return out


def fr_bytes(wire) -> bytes:
    """wire-format element(s) -> canonical 32-byte LE each"""
    return digests_bytes(wire, 1)


def fr_from_bytes(b: bytes, n: int) -> np.ndarray:
    if len(b) < 32 * n:
        raise ValueError("truncated input: %d field elements need %d bytes, got %d" % (n, 32 * n, len(b)))
    return digests_from_bytes(b[: 32 * n], n, 1).reshape(n, 4)


def te_points_bytes(points_wire, compress=False) -> bytes:
    """affine points [n, 2, 4] (wire format) -> x || y each (uncompressed) or y with the x-sign flag (compressed)"""
    return digests_bytes(points_wire, 2, compress)


def te_points_from_bytes(b: bytes, n: int, compress=False, validate=True) -> np.ndarray:
    """inverse of te_points_bytes -> [n, 2, 4] wire format"""
    per = 32 if compress else 64
    if len(b) < per * n:
        raise ValueError("truncated input: %d points need %d bytes, got %d" % (n, per * n, len(b)))
    return digests_from_bytes(b[: per * n], n, 2, compress, validate)


def serialize_poseidon_config(cfg) -> bytes:
    h = _host_handle(cfg)  # host-only handle (no device): the library owns the byte format
    try:
        return _ser(lambda out, cap, n: lib.akp_serialize_poseidon_config(h, out, cap, n))
    finally:
        lib.akp_poseidon_params_destroy(h)


def _host_handle(cfg):
    h = C.c_void_p()
    check(lib.akp_poseidon_params_create(None, cfg.full_rounds, cfg.partial_rounds, cfg.alpha, cfg.rate, cfg.capacity, cfg.ark.ctypes.data,
                                         cfg.mds.ctypes.data, C.byref(h)))
    return h


def deserialize_poseidon_config(b: bytes):
    from .sponge.poseidon import PoseidonConfig
    buf, ln = _in(b)
    h = C.c_void_p()
    _de(lib.akp_deserialize_poseidon_config(None, buf, ln, C.byref(h)))
    try:
        rf, rp, r, c = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        alpha = C.c_uint64()
        check(lib.akp_poseidon_params_dims(h, C.byref(rf), C.byref(rp), C.byref(alpha), C.byref(r), C.byref(c)))
        t = r.value + c.value
        ark = np.empty((rf.value + rp.value, t, 4), dtype=np.uint64)
        mds = np.empty((t, t, 4), dtype=np.uint64)
        check(lib.akp_poseidon_params_export(h, ark.ctypes.data, mds.ctypes.data))
        return PoseidonConfig(rf.value, rp.value, alpha.value, ark, mds, r.value, c.value)
    finally:
        lib.akp_poseidon_params_destroy(h)


def serialize_te_parameters(params, compress=False) -> bytes:
    """Parameters { generators: Vec<Vec<C>> }: projective points serialise as their affine form"""
    g = np.ascontiguousarray(params.generators, dtype=np.uint64)
    nw, ws = (g.shape[0], g.shape[1]) if g.size else (0, 0)
    return _ser(lambda out, cap, n: lib.akp_serialize_te_parameters(g.ctypes.data if g.size else None, ws, nw, int(compress), out, cap, n))


def deserialize_te_parameters(b: bytes, cls, compress=False, validate=True):
    buf, ln = _in(b)
    ws, nw = C.c_uint32(), C.c_uint32()
    _de(lib.akp_deserialize_te_parameters(buf, ln, int(compress), 0, None, 0, C.byref(ws), C.byref(nw)))
    g = np.empty((nw.value, ws.value, 2, 4), dtype=np.uint64)
    if g.size:
        _de(lib.akp_deserialize_te_parameters(buf, ln, int(compress), int(validate), g.ctypes.data, nw.value * ws.value, C.byref(ws), C.byref(nw)))
    return cls(g)


def _digest_bytes(d, compress=False) -> bytes:
    d = np.asarray(d, dtype=np.uint64)
    return digests_bytes(d, 2 if d.shape[-2:] == (2, 4) else 1, compress)


def serialize_path(path, compress=False) -> bytes:
    """Path { leaf_sibling_hash, auth_path: Vec<InnerDigest>, leaf_index: usize }"""
    fe = _fe_of(path.config)
    sib = _wire(path.leaf_sibling_hash, fe)
    depth = len(path.auth_path)
    auth = _wire(np.stack([np.asarray(a) for a in path.auth_path]), fe) if depth else np.zeros((0, fe, 4), np.uint64)
    return _ser(lambda out, cap, n: lib.akp_serialize_path(sib.ctypes.data, auth.ctypes.data if depth else None, depth, int(path.leaf_index), fe,
                                                           int(compress), out, cap, n))


def deserialize_path(b: bytes, config, compress=False, validate=True):
    from .merkle_tree import Path
    fe = _fe_of(config)
    buf, ln = _in(b)
    depth, idx = C.c_size_t(), C.c_uint64()
    _de(lib.akp_deserialize_path(buf, ln, fe, int(compress), 0, None, None, 0, C.byref(depth), C.byref(idx)))
    sib = np.empty((fe, 4), dtype=np.uint64)
    auth = np.empty((max(depth.value, 1), fe, 4), dtype=np.uint64)
    _de(lib.akp_deserialize_path(buf, ln, fe, int(compress), int(validate), sib.ctypes.data, auth.ctypes.data, depth.value, C.byref(depth), C.byref(idx)))
    shape = tuple(config.digest_shape)
    return Path(config, sib.reshape(shape), [auth[j].reshape(shape) for j in range(depth.value)], idx.value)


def serialize_multi_path(mp, compress=False) -> bytes:
    """MultiPath { leaf_siblings_hashes, auth_paths_prefix_lenghts, auth_paths_suffixes, leaf_indexes }"""
    fe = _fe_of(mp.config)
    m = len(mp.leaf_indexes)
    if not (len(mp.leaf_siblings_hashes) == len(mp.auth_paths_prefix_lenghts) == len(mp.auth_paths_suffixes) == m):
        raise ValueError("the four vectors of a MultiPath must have the same length")
    sibs = _wire(np.stack([np.asarray(d) for d in mp.leaf_siblings_hashes]), fe) if m else np.zeros((0, fe, 4), np.uint64)
    pre = np.asarray(list(mp.auth_paths_prefix_lenghts), dtype=np.uint64)
    sl = np.asarray([len(s) for s in mp.auth_paths_suffixes], dtype=np.uint64)
    flat = [np.asarray(d) for s in mp.auth_paths_suffixes for d in s]
    suf = _wire(np.stack(flat), fe) if flat else np.zeros((0, fe, 4), np.uint64)
    idx = np.asarray(list(mp.leaf_indexes), dtype=np.uint64)
    return _ser(lambda out, cap, n: lib.akp_serialize_multipath(sibs.ctypes.data if m else None, pre.ctypes.data if m else None, sl.ctypes.data if m else None,
                                                                suf.ctypes.data if flat else None, idx.ctypes.data if m else None, m, 0, fe, int(compress),
                                                                out, cap, n))


def deserialize_multi_path(b: bytes, config, compress=False, validate=True):
    from .merkle_tree import MultiPath
    fe = _fe_of(config)
    buf, ln = _in(b)
    m, ns = C.c_size_t(), C.c_size_t()
    _de(lib.akp_deserialize_multipath(buf, ln, fe, int(compress), 0, C.byref(m), C.byref(ns), None, None, None, None, None, 0, 0))
    M, NS = m.value, ns.value
    sibs = np.empty((max(M, 1), fe, 4), dtype=np.uint64)
    pre, sl, idx = (np.empty(max(M, 1), dtype=np.uint64) for _ in range(3))
    suf = np.empty((max(NS, 1), fe, 4), dtype=np.uint64)
    _de(lib.akp_deserialize_multipath(buf, ln, fe, int(compress), int(validate), C.byref(m), C.byref(ns), sibs.ctypes.data, pre.ctypes.data, sl.ctypes.data,
                                      suf.ctypes.data, idx.ctypes.data, M, NS))
    shape = tuple(config.digest_shape)
    suffixes, at = [], 0
    for i in range(M):
        k = int(sl[i])
        suffixes.append([suf[at + j].reshape(shape) for j in range(k)])
        at += k
    return MultiPath(config, [sibs[i].reshape(shape) for i in range(M)], [int(v) for v in pre[:M]], suffixes, [int(v) for v in idx[:M]])
