"""Pedersen / Bowe-Hopwood / Merkle oracle checks.  The reference holds no absolute vectors for these.  The
group law underneath them is pinned at value level by one known answer of the upstream curve crate
(tests/golden/jubjub_upstream_kat.json); bit order, padding and the digest encodings stay pinned structurally only."""
import numpy as np
import pytest

from oracle import jubjub as jj, pedersen as pd, bowe_hopwood as bh, merkle, fr as ofr, cref
from helpers import mont, ints, gens_array


def test_curve_constants():
    assert jj.is_on_curve(jj.GENERATOR) and jj.is_on_curve(jj.IDENTITY)
    assert jj.mul(jj.GENERATOR, jj.SUBGROUP_ORDER) == jj.IDENTITY
    assert jj.mul(jj.GENERATOR, jj.SUBGROUP_ORDER * jj.COFACTOR) == jj.IDENTITY
    assert jj.add(jj.GENERATOR, jj.neg(jj.GENERATOR)) == jj.IDENTITY
    p3 = jj.mul(jj.GENERATOR, 3)
    assert jj.add(jj.double(jj.GENERATOR), jj.GENERATOR) == p3 and jj.is_on_curve(p3)
    assert bh.max_chunks_per_segment() == 63  # bowe_hopwood/mod.rs:82-101 for Jubjub's scalar field


def _kat_pedersen_case(k):
    """Pedersen with ONE window of 256 generators 2^j * g evaluates the scalar multiplication (sum_j m_j 2^j) * g of the message
    read as a little-endian integer (bytes_to_bits is LSB-first, crh/pedersen/mod.rs:200-209)"""
    scalar = (k["f1"] * k["f2"]) % jj.SUBGROUP_ORDER
    gens = [[k["g"]]]
    for _ in range(255):
        gens[0].append(jj.double(gens[0][-1]))
    return scalar.to_bytes(32, "little"), gens


def test_upstream_jubjub_kat_pins_the_group_law(jubjub_kat):
    """VALUE-LEVEL PIN (the only one the curve side has): ark-ed-on-bls12-381's test_scalar_multiplication vector.  The
    python oracle's group law and scalar multiplication, the oracle's Pedersen evaluation, the C oracle's Pedersen
    evaluation and the product's host-side curve code (params.py) must all reproduce f1 * f2 * g."""
    k = jubjub_kat
    assert jj.is_on_curve(k["g"]) and jj.is_on_curve(k["f1f2g"])
    assert jj.mul(jj.mul(k["g"], k["f1"]), k["f2"]) == k["f1f2g"]
    assert jj.mul(k["g"], (k["f1"] * k["f2"]) % jj.SUBGROUP_ORDER) == k["f1f2g"]  # g has prime order
    assert jj.mul(k["g"], jj.SUBGROUP_ORDER) == jj.IDENTITY
    msg, gens = _kat_pedersen_case(k)
    assert pd.evaluate(gens, 256, 1, msg) == k["f1f2g"]
    C = cref.CurveParams(256, 1, gens_array(gens))
    out = C.pedersen_crh_batch(np.frombuffer(msg, dtype=np.uint8), 1, 32)
    assert tuple(ints(out[0])) == k["f1f2g"]
    from crypto_primitives_amd import params as cparams
    scalar = (k["f1"] * k["f2"]) % jj.SUBGROUP_ORDER
    assert cparams._affine(cparams._smul(k["g"], scalar)) == k["f1f2g"]


def test_pedersen_known_answers_and_scalar_form():
    g = jj.pedersen_generators(1, 4, 256)
    assert pd.evaluate(g, 4, 256, b"") == jj.IDENTITY            # nothing is added (pedersen/mod.rs:116-122)
    assert pd.evaluate(g, 4, 256, bytes(128)) == jj.IDENTITY
    msg = bytes(range(128))
    h = pd.evaluate(g, 4, 256, msg)
    nibs = []
    for b in msg:
        nibs += [b & 15, b >> 4]
    assert h == jj.sum_points([jj.mul(g[i][0], nibs[i]) for i in range(256)])  # H(m) = sum nibble_i * G_i
    assert jj.is_on_curve(h) and jj.mul(h, jj.SUBGROUP_ORDER) == jj.IDENTITY
    assert pd.evaluate(g, 4, 256, msg[:40]) == pd.evaluate(g, 4, 256, msg[:40] + bytes(88))  # zero padding :91-99
    with pytest.raises(pd.InputLengthPanic):
        pd.evaluate(g, 4, 256, bytes(129))
    # two-to-one: concatenation semantics :158-182
    assert pd.two_to_one_evaluate(g, 4, 256, msg[:64], msg[64:]) == h


def test_bowe_hopwood_known_answers():
    g = jj.bowe_hopwood_generators(2, 63, 9)
    assert bh.evaluate(g, 63, 9, b"") == 0                         # identity -> x = 0
    flat = [p for row in g for p in row]
    for L in (1, 3, 32):
        n_chunks = (8 * L + 2) // 3
        assert bh.evaluate(g, 63, 9, bytes(L)) == jj.sum_points(flat[:n_chunks])[0]  # zero chunk = +g (:167)
    # signed-digit scalar form: digit = (1 + b0 + 2 b1) * (-1)^b2 times generators[s][0] * 16^j
    msg = ofr.SplitMix64(5).bytes(32)
    bits = pd.bytes_to_bits(msg)
    bits += [False] * (-len(bits) % 3)
    pts = []
    for c in range(len(bits) // 3):
        b0, b1, b2 = bits[3 * c:3 * c + 3]
        d = (1 + b0 + 2 * b1) * (-1 if b2 else 1)
        pts.append(jj.mul(g[c // 63][0], (d * 16 ** (c % 63)) % jj.SUBGROUP_ORDER))
    assert bh.evaluate(g, 63, 9, msg) == jj.sum_points(pts)[0]
    with pytest.raises(pd.InputLengthPanic):
        bh.evaluate(g, 63, 9, bytes(213))


def test_c_oracle_curves_equal_python(derived):
    g = jj.pedersen_generators(0xA5A50004, 4, 256)
    d = derived["pedersen_4x256"]
    assert [str(g[0][0][0]), str(g[0][0][1])] == d["g00"]
    CP = cref.CurveParams(4, 256, gens_array(g))
    msg = bytes.fromhex(d["msg"])
    o = CP.pedersen_crh_batch(np.frombuffer(msg, np.uint8), 1, 128)
    assert [str(v) for v in ints(o[0])] == d["digest"]
    o = CP.pedersen_crh_batch(np.frombuffer(msg[:32], np.uint8), 1, 32)
    assert [str(v) for v in ints(o[0])] == d["digest_first32"]
    gb = jj.bowe_hopwood_generators(0xA5A50005, 63, 9)
    db = derived["bowe_hopwood_63x9"]
    BP = cref.CurveParams(63, 9, gens_array(gb))
    assert str(ints(BP.bh_crh_batch(np.frombuffer(bytes.fromhex(db["msg32"]), np.uint8), 1, 32))[0]) == db["digest32"]
    assert str(ints(BP.bh_crh_batch(np.frombuffer(bytes.fromhex(db["msg70"]), np.uint8), 1, 70))[0]) == db["digest70"]
    assert str(ints(BP.bh_crh_batch(np.zeros(3, np.uint8), 1, 3))[0]) == db["zero_3bytes"]
    leaves = b"".join(bytes.fromhex(x) for x in derived["bowe_hopwood_merkle_4"]["leaves"])
    ln, nl = BP.merkle_build(1, BP, np.frombuffer(leaves, np.uint8), 4, 32, threads=2)
    assert str(ints(nl[0])[0]) == derived["bowe_hopwood_merkle_4"]["root"]
    ln, nl = CP.merkle_build(0, CP, np.frombuffer(leaves, np.uint8), 4, 32, threads=2)
    assert [str(v) for v in ints(nl[0])] == derived["pedersen_merkle_4"]["root"]


def test_merkle_structure_python_oracle():
    """proof round trips + multi-proof prefix lengths [0,2,1,2,0,2,1,2] (merkle_tree/tests/mod.rs:95-182)."""
    from oracle import poseidon as po
    c = po.get_default_poseidon_parameters(2, False)
    leaves = [[i, i + 1, i + 2] for i in range(8)]
    t = merkle.MerkleTree(lambda l: po.crh_evaluate(c, l), lambda a, b: po.two_to_one_compress(c, a, b),
                          lambda a, b: po.two_to_one_compress(c, a, b), lambda x: x, leaves=leaves)
    assert t.height == 4
    for i in range(8):
        assert t.verify(t.generate_proof(i), t.root(), leaves[i])
        assert not t.verify(t.generate_proof(i), (t.root() + 1) % ofr.P, leaves[i])
    mp = t.generate_multi_proof(range(8))
    assert mp["auth_paths_prefix_lenghts"] == [0, 2, 1, 2, 0, 2, 1, 2]
    assert t.verify_multi(mp, t.root(), leaves)
    t.update(3, [9, 9, 9]); leaves[3] = [9, 9, 9]
    for i in range(8):
        assert t.verify(t.generate_proof(i), t.root(), leaves[i])
    with pytest.raises(AssertionError):
        merkle.MerkleTree(t.leaf_hash, t.t_eval, t.t_comp, t.convert, leaves=leaves[:3])


def test_injective_map_compressor_is_x_of_pedersen():
    """crh/injective_map/mod.rs:24-31,54-62,81-107: TECompressor keeps x; compress serialises the two Fq digests (32 bytes
    LE canonical each) and evaluates"""
    g = jj.pedersen_generators(11, 4, 256)
    m = bytes(range(40))
    opd = pd
    pt = opd.evaluate(g, 4, 256, m)
    assert opd.compressor_evaluate(g, 4, 256, m) == pt[0]
    l, r = bytes(range(20)), bytes(range(20, 40))
    assert opd.compressor_two_to_one_evaluate(g, 4, 256, l, r) == opd.two_to_one_evaluate(g, 4, 256, l, r)[0]
    a, b = 12345, jj.Q - 2
    want = opd.evaluate(g, 4, 256, a.to_bytes(32, "little") + b.to_bytes(32, "little") + bytes(64))[0]
    assert opd.compressor_two_to_one_compress(g, 4, 256, a, b) == want
