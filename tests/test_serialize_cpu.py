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
