"""ark-serialize style encodings (crypto_primitives_amd/serialize.py), both modes, no GPU: round trips and an independent
restatement of the compressed twisted-Edwards encoding (y, 32 bytes LE, top bit of the last byte = x is the larger of
(x, -x)) from the oracle's big-int curve code.  Unpinned: the reference holds no byte-level vector for any of these."""
import numpy as np
import pytest

from oracle import jubjub as jj
from helpers import gens_array


def test_te_point_encodings_both_modes():
    from crypto_primitives_amd import serialize as ser
    pts = [jj.mul(jj.GENERATOR, k) for k in (1, 2, 3, 5, 1234567, jj.SUBGROUP_ORDER - 1)] + [jj.IDENTITY]
    wire = gens_array([pts])[0]  # [n, 2, 4]
    unc = ser.te_points_bytes(wire, False)
    assert unc == b"".join(jj.serialize_uncompressed(p) for p in pts) and len(unc) == 64 * len(pts)
    comp = ser.te_points_bytes(wire, True)
    exp = bytearray()
    for x, y in pts:
        b = bytearray(y.to_bytes(32, "little"))
        if x > (jj.Q - x) % jj.Q:
            b[31] |= 0x80
        exp += b
    assert comp == bytes(exp) and len(comp) == 32 * len(pts)
    assert any(b & 0x80 for b in comp[31::32]) and not all(b & 0x80 for b in comp[31::32])  # both signs occur
    assert np.array_equal(ser.te_points_from_bytes(comp, len(pts), True), wire)
    assert np.array_equal(ser.te_points_from_bytes(unc, len(pts), False), wire)
    k, k_neg = pts[4], jj.neg(pts[4])  # P and -P share y and differ in the flag only
    a, b = ser.te_points_bytes(gens_array([[k]])[0], True), ser.te_points_bytes(gens_array([[k_neg]])[0], True)
    assert a[:31] == b[:31] and (a[31] ^ b[31]) == 0x80
    with pytest.raises(ValueError):  # y = 2 is not on the curve ((4 - 1) / (1 + 4d) is a non-residue) or p itself is non-canonical
        ser.te_points_from_bytes(jj.Q.to_bytes(32, "little"), 1, True)


def test_parameters_paths_round_trip_both_modes():
    import crypto_primitives_amd as cpa
    from crypto_primitives_amd import serialize as ser
    from crypto_primitives_amd.crh import pedersen
    g = gens_array(jj.pedersen_generators(3, 3, 2))
    P = pedersen.Parameters(g)
    for compress in (False, True):
        b = ser.serialize_te_parameters(P, compress)
        assert len(b) == 8 + 2 * (8 + 3 * (32 if compress else 64))
        assert np.array_equal(ser.deserialize_te_parameters(b, pedersen.Parameters, compress).generators, P.generators)
        dig = [g[0, i] for i in range(3)]  # affine points as Pedersen digests
        path = cpa.Path(cpa.PedersenByteConfig, dig[0], [dig[1], dig[2]], 5)
        pb = ser.serialize_path(path, compress)
        assert len(pb) == (32 if compress else 64) * 3 + 16
        back = ser.deserialize_path(pb, cpa.PedersenByteConfig, compress)
        assert back.leaf_index == 5 and np.array_equal(back.leaf_sibling_hash, dig[0]) and np.array_equal(back.auth_path[1], dig[2])
        mp = cpa.MultiPath(cpa.PedersenByteConfig, [dig[0], dig[1]], [0, 1], [[dig[1], dig[2]], [dig[0]]], [2, 3])
        mb = ser.serialize_multi_path(mp, compress)
        back = ser.deserialize_multi_path(mb, cpa.PedersenByteConfig, compress)
        assert back.leaf_indexes == [2, 3] and back.auth_paths_prefix_lenghts == [0, 1] and np.array_equal(back.auth_paths_suffixes[1][0], dig[0])
    # field digests are identical in the two modes
    fpath = cpa.Path(cpa.PoseidonFieldConfig, g[0, 0, 0], [g[0, 1, 0]], 1)
    assert ser.serialize_path(fpath, True) == ser.serialize_path(fpath, False)


def test_deserialisation_rejects_truncated_and_invalid_input():
    """length prefixes and payloads are bounds-checked (ValueError, never IndexError / a silent short read), and
    twisted-Edwards points are validated like ark-serialize's default `Validate::Yes`: on the curve and in the prime-order
    subgroup -- off-curve generators must not reach the GPU tables.  validate=False is the `_unchecked` form."""
    import pytest
    from crypto_primitives_amd import serialize as S, params, field
    from crypto_primitives_amd.crh import pedersen
    P = pedersen.Parameters(params.pedersen_generators(7, 3, 2))
    for compress in (False, True):
        b = S.serialize_te_parameters(P, compress)
        assert np.array_equal(S.deserialize_te_parameters(b, pedersen.Parameters, compress).generators, P.generators)
        for cut in (0, 1, 7, 9, 20, len(b) - 1):
            with pytest.raises(ValueError):
                S.deserialize_te_parameters(b[:cut], pedersen.Parameters, compress)
        huge = (1 << 40).to_bytes(8, "little") + b[8:]  # a length prefix far beyond the payload
        with pytest.raises(ValueError):
            S.deserialize_te_parameters(huge, pedersen.Parameters, compress)
    bad = bytearray(S.serialize_te_parameters(P, False))
    bad[16] ^= 1  # x of the first generator: no longer on the curve
    with pytest.raises(ValueError, match="not on the curve"):
        S.deserialize_te_parameters(bytes(bad), pedersen.Parameters, False)
    assert S.deserialize_te_parameters(bytes(bad), pedersen.Parameters, False, validate=False).generators.shape == P.generators.shape
    order2 = S.te_points_bytes(field.fr([0, field.MODULUS - 1]).reshape(1, 2, 4))  # (0, -1): on the curve, order 2
    with pytest.raises(ValueError, match="prime-order subgroup"):
        S.te_points_from_bytes(order2, 1)
    cfg_b = S.serialize_poseidon_config(__import__("crypto_primitives_amd").get_default_poseidon_parameters(2, False))
    for cut in (5, 30, len(cfg_b) - 3):
        with pytest.raises(ValueError):
            S.deserialize_poseidon_config(cfg_b[:cut])


# ---- round 4: the product's bytes (C ABI: akp_serialize_* / akp_deserialize_*) against the oracle's independent restatement ----
def _ints(a):
    from crypto_primitives_amd import field
    return [int(v) for v in field.to_ints(np.asarray(a, dtype=np.uint64).reshape(-1, 4))]


def _wire(vals, shape):
    from crypto_primitives_amd import field
    return field.fr([int(v) for v in vals]).reshape(shape)


@pytest.mark.parametrize("compress", [False, True])
def test_every_struct_byte_for_byte_against_oracle_serialize(compress):
    """writers: product bytes == oracle bytes; readers: each side parses the other's bytes to the same values"""
    import crypto_primitives_amd as cpa
    from crypto_primitives_amd import serialize as S
    from crypto_primitives_amd.crh import pedersen
    from oracle import serialize as O
    pts = [jj.mul(jj.GENERATOR, k) for k in (1, 2, 3, 5, 7, 11, 13, 17, 19, jj.SUBGROUP_ORDER - 2)]
    assert any(x > (jj.Q - 1) // 2 for x, _ in pts) and any(x <= (jj.Q - 1) // 2 for x, _ in pts)
    fes = [(3 ** (40 + i) % jj.Q,) for i in range(10)] + [(0,), (jj.Q - 1,)]
    # Parameters
    g = [pts[0:3], pts[3:6]]
    P = pedersen.Parameters(gens_array(g))
    assert S.serialize_te_parameters(P, compress) == O.te_parameters(g, compress)
    assert O.read_te_parameters(S.serialize_te_parameters(P, compress), compress) == [[tuple(p) for p in row] for row in g]
    back = S.deserialize_te_parameters(O.te_parameters(g, compress), pedersen.Parameters, compress)
    assert np.array_equal(back.generators, P.generators)
    empty = S.deserialize_te_parameters(O.te_parameters([], compress), pedersen.Parameters, compress)  # ark-serialize accepts an empty Vec
    assert empty.generators.size == 0
    # Path and MultiPath, point digests (Pedersen) and field digests (Poseidon / Bowe-Hopwood)
    for cfg, ds, shape in ((cpa.PedersenByteConfig, pts, (2, 4)), (cpa.PoseidonFieldConfig, fes, cpa.PoseidonFieldConfig.digest_shape)):
        fe = len(ds[0])
        w = [_wire(d, shape) for d in ds]
        pb = S.serialize_path(cpa.Path(cfg, w[0], w[1:6], 29), compress)
        assert pb == O.path(ds[0], ds[1:6], 29, compress)
        sib, auth, idx = O.read_path(pb, fe, compress)
        assert (tuple(sib), [tuple(a) for a in auth], idx) == (tuple(ds[0]), [tuple(d) for d in ds[1:6]], 29)
        p2 = S.deserialize_path(O.path(ds[0], ds[1:6], 29, compress), cfg, compress)
        assert p2.leaf_index == 29 and _ints(p2.leaf_sibling_hash) == list(ds[0]) and [_ints(a) for a in p2.auth_path] == [list(d) for d in ds[1:6]]
        assert S.serialize_path(cpa.Path(cfg, w[0], [], 0), compress) == O.path(ds[0], [], 0, compress)  # a two-leaf tree: empty auth path
        suf = [ds[3:6], [], ds[6:8]]
        mb = S.serialize_multi_path(cpa.MultiPath(cfg, w[0:3], [0, 3, 1], [[_wire(d, shape) for d in s] for s in suf], [4, 5, 9]), compress)
        assert mb == O.multi_path(ds[0:3], [0, 3, 1], suf, [4, 5, 9], compress)
        got = O.read_multi_path(mb, fe, compress)
        assert got["auth_paths_prefix_lenghts"] == [0, 3, 1] and got["leaf_indexes"] == [4, 5, 9] and [len(s) for s in got["auth_paths_suffixes"]] == [3, 0, 2]
        m2 = S.deserialize_multi_path(O.multi_path(ds[0:3], [0, 3, 1], suf, [4, 5, 9], compress), cfg, compress)
        assert m2.leaf_indexes == [4, 5, 9] and m2.auth_paths_prefix_lenghts == [0, 3, 1]
        assert [[_ints(d) for d in s] for s in m2.auth_paths_suffixes] == [[list(d) for d in s] for s in suf]
        assert [_ints(d) for d in m2.leaf_siblings_hashes] == [list(d) for d in ds[0:3]]
    # PoseidonConfig (mode-independent)
    cfgp = cpa.get_default_poseidon_parameters(2, False)
    ob = O.poseidon_config(cfgp.full_rounds, cfgp.partial_rounds, cfgp.alpha, [_ints(r) for r in cfgp.ark], [_ints(r) for r in cfgp.mds], cfgp.rate,
                           cfgp.capacity, compress)
    assert S.serialize_poseidon_config(cfgp) == ob
    rd = O.read_poseidon_config(ob)
    assert (rd["full_rounds"], rd["partial_rounds"], rd["alpha"], rd["rate"], rd["capacity"]) == (8, 31, 17, 2, 1) and len(rd["ark"]) == 39
    c2 = S.deserialize_poseidon_config(ob)
    assert np.array_equal(c2.ark, cfgp.ark) and np.array_equal(c2.mds, cfgp.mds) and (c2.rate, c2.capacity, c2.alpha) == (2, 1, 17)


def test_readers_agree_on_what_is_invalid():
    """the oracle's reader and the product's reader reject the same malformed inputs (and the unchecked forms accept the same)"""
    import crypto_primitives_amd as cpa
    from crypto_primitives_amd import serialize as S
    from oracle import serialize as O
    pts = [jj.mul(jj.GENERATOR, k) for k in (2, 9, 33)]
    good = O.path(pts[0], pts[1:], 3, False)

    def both_reject(b, compress=False, fe=2, cfg=None):
        cfg = cfg or cpa.PedersenByteConfig
        with pytest.raises(O.FormatError):
            O.read_path(b, fe, compress)
        with pytest.raises(ValueError):
            S.deserialize_path(b, cfg, compress)
    both_reject(good[:-1])                       # truncated
    both_reject(good + b"\0")                    # trailing byte
    both_reject(good[:64] + (1 << 50).to_bytes(8, "little") + good[72:])  # absurd Vec length
    off = bytearray(good); off[3] ^= 4
    both_reject(bytes(off))                      # x changed: not on the curve
    assert O.read_path(bytes(off), 2, False, validate=False)[2] == 3
    assert S.deserialize_path(bytes(off), cpa.PedersenByteConfig, False, validate=False).leaf_index == 3
    small = O.path((0, jj.Q - 1), [], 1, False)  # (0, -1): order 2
    both_reject(small)
    noncanon = jj.Q.to_bytes(32, "little") + good[32:]
    both_reject(noncanon)
    # compressed: a y with no point on the curve; a set flag on the point with x = 0 decodes to x = 0 again (the flag cannot be honoured)
    ys = next(y for y in range(2, 50) if jj.fq_sqrt((y * y - 1) * pow(1 + jj.D * y * y, -1, jj.Q)) is None)
    both_reject(ys.to_bytes(32, "little") + (0).to_bytes(8, "little") + (0).to_bytes(8, "little"), compress=True)
    fgood = O.path((5,), [(6,), (7,)], 2, False)
    both_reject(fgood[:32] + (3).to_bytes(8, "little") + fgood[40:], fe=1, cfg=cpa.PoseidonFieldConfig)  # a length one larger than the payload


def test_c_abi_serializer_size_queries_and_small_buffers():
    import ctypes as C
    import crypto_primitives_amd as cpa
    from crypto_primitives_amd._lib import lib, AKP_ERR_BAD_LENGTH, AKP_ERR_BAD_PARAMS
    from crypto_primitives_amd import field
    d = field.fr([1, 2, 3, 4]).reshape(4, 1, 4)
    n = C.c_size_t()
    assert lib.akp_serialize_digests(d.ctypes.data, 4, 1, 0, None, 0, C.byref(n)) == 0 and n.value == 128
    buf = (C.c_uint8 * 127)()
    assert lib.akp_serialize_digests(d.ctypes.data, 4, 1, 0, buf, 127, C.byref(n)) == AKP_ERR_BAD_LENGTH and n.value == 128
    assert b"128 bytes needed" in lib.akp_last_error()
    assert lib.akp_serialize_digests(d.ctypes.data, 4, 3, 0, None, 0, C.byref(n)) == AKP_ERR_BAD_PARAMS
    ok = (C.c_uint8 * 128)()
    assert lib.akp_serialize_digests(d.ctypes.data, 4, 1, 0, ok, 128, C.byref(n)) == 0
    assert bytes(ok) == b"".join(int(v).to_bytes(32, "little") for v in (1, 2, 3, 4))
    back = np.zeros((4, 1, 4), np.uint64)
    assert lib.akp_deserialize_digests(ok, 128, 4, 1, 0, 1, back.ctypes.data) == 0 and np.array_equal(back, d)
    assert lib.akp_deserialize_digests(ok, 127, 4, 1, 0, 1, back.ctypes.data) == AKP_ERR_BAD_LENGTH
    # a MultiPath whose four vectors differ in length is valid ark-serialize input the flat form cannot hold
    from oracle import serialize as O
    ragged = O.multi_path([(1,), (2,)], [0], [[(3,)]], [0, 1], False)
    m, ns = C.c_size_t(), C.c_size_t()
    raw = (C.c_uint8 * len(ragged)).from_buffer_copy(ragged)
    assert lib.akp_deserialize_multipath(raw, len(ragged), 1, 0, 1, C.byref(m), C.byref(ns), None, None, None, None, None, 0, 0) == AKP_ERR_BAD_PARAMS
    assert cpa.lib.akp_abi_version() == 5


@pytest.mark.parametrize("compress", [False, True])
def test_path_with_different_leaf_and_inner_digest_types(compress):
    """a Config whose LeafDigest is a curve point and whose InnerDigest is a field element (or the reverse): Path / MultiPath carry both
    types (merkle_tree/mod.rs:139-152, 239-254); the C ABI takes AKP_FE_PAIR(leaf_fe, inner_fe) and the bytes are the oracle's"""
    import ctypes as C
    import crypto_primitives_amd as cpa
    from oracle import serialize as O
    lib = cpa.lib
    pts = [jj.mul(jj.GENERATOR, k) for k in (1, 2, 3, 5, 7, 11)]
    fes = [(3 ** (50 + i) % jj.Q,) for i in range(6)]
    for leaf, inner, lfe, ife in ((pts, fes, 2, 1), (fes, pts, 1, 2)):
        pair = (lfe << 8) | ife
        want = O.path(leaf[0], inner[1:5], 6, compress)
        sib = np.ascontiguousarray(_wire(leaf[0], (lfe, 4) if lfe == 2 else (4,)), dtype=np.uint64)
        auth = np.ascontiguousarray(np.stack([_wire(d, (ife, 4) if ife == 2 else (4,)) for d in inner[1:5]]), dtype=np.uint64)
        n = C.c_size_t()
        assert lib.akp_serialize_path(sib.ctypes.data, auth.ctypes.data, 4, 6, pair, int(compress), None, 0, C.byref(n)) == 0 and n.value == len(want)
        buf = (C.c_uint8 * n.value)()
        assert lib.akp_serialize_path(sib.ctypes.data, auth.ctypes.data, 4, 6, pair, int(compress), buf, n.value, C.byref(n)) == 0
        assert bytes(buf) == want
        s2, a2 = np.zeros_like(sib), np.zeros_like(auth)
        depth, idx = C.c_size_t(), C.c_uint64()
        raw = (C.c_uint8 * len(want)).from_buffer_copy(want)
        assert lib.akp_deserialize_path(raw, len(want), pair, int(compress), 1, s2.ctypes.data, a2.ctypes.data, 4, C.byref(depth), C.byref(idx)) == 0
        assert (depth.value, idx.value) == (4, 6) and np.array_equal(s2, sib) and np.array_equal(a2, auth)
        if not compress:  # (a compressed point is 32 bytes like a field element: only the uncompressed framing tells the widths apart)
            big_s, big_a = np.zeros(8, np.uint64), np.zeros(64, np.uint64)  # room for the widest reading
            assert lib.akp_deserialize_path(raw, len(want), ife, 0, 1, big_s.ctypes.data, big_a.ctypes.data, 4, C.byref(depth), C.byref(idx)) != 0
        # MultiPath: two paths, siblings of the leaf type, suffixes of the inner type
        suf = [inner[1:4], inner[4:5]]
        wantm = O.multi_path(leaf[0:2], [0, 2], suf, [2, 3], compress)
        sibs = np.ascontiguousarray(np.stack([_wire(d, (lfe, 4) if lfe == 2 else (4,)) for d in leaf[0:2]]), dtype=np.uint64)
        flat = np.ascontiguousarray(np.stack([_wire(d, (ife, 4) if ife == 2 else (4,)) for s in suf for d in s]), dtype=np.uint64)
        pre, sl, li = np.array([0, 2], np.uint64), np.array([3, 1], np.uint64), np.array([2, 3], np.uint64)
        assert lib.akp_serialize_multipath(sibs.ctypes.data, pre.ctypes.data, sl.ctypes.data, flat.ctypes.data, li.ctypes.data, 2, 3, pair, int(compress), None, 0, C.byref(n)) == 0
        bufm = (C.c_uint8 * n.value)()
        assert lib.akp_serialize_multipath(sibs.ctypes.data, pre.ctypes.data, sl.ctypes.data, flat.ctypes.data, li.ctypes.data, 2, 3, pair, int(compress), bufm, n.value, C.byref(n)) == 0
        assert bytes(bufm) == wantm
        m, ns = C.c_size_t(), C.c_size_t()
        rawm = (C.c_uint8 * len(wantm)).from_buffer_copy(wantm)
        s3, f3, p3, l3, i3 = np.zeros_like(sibs), np.zeros_like(flat), np.zeros(2, np.uint64), np.zeros(2, np.uint64), np.zeros(2, np.uint64)
        assert lib.akp_deserialize_multipath(rawm, len(wantm), pair, int(compress), 1, C.byref(m), C.byref(ns), s3.ctypes.data, p3.ctypes.data, l3.ctypes.data,
                                             f3.ctypes.data, i3.ctypes.data, 2, 4) == 0
        assert (m.value, ns.value) == (2, 4) and np.array_equal(s3, sibs) and np.array_equal(f3, flat) and list(l3) == [3, 1] and list(i3) == [2, 3]
    assert lib.akp_serialize_path(sib.ctypes.data, auth.ctypes.data, 4, 6, (3 << 8) | 1, 0, None, 0, C.byref(n)) == cpa._lib.AKP_ERR_BAD_PARAMS
