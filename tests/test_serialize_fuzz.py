"""Mutation fuzz of the byte-format READERS of the C ABI (akp_deserialize_*: the only part of the product library that parses bytes
it did not produce).  Valid oracle-made encodings of every struct are bit-flipped, truncated, extended and spliced; for each mutant
the product reader and the independent oracle reader (oracle/serialize.py) must agree: both reject, or both accept and then
re-serialise to the same bytes.  Host only (no GPU).  tests/test_sanitizers.py runs this file once more against an
AddressSanitizer build of the library (`make -C crypto_primitives_amd/csrc asan`)."""
import os

import numpy as np
import pytest

from oracle import jubjub as jj, serialize as O
from helpers import gens_array

N_MUTANTS = int(os.environ.get("AKP_FUZZ_MUTANTS", "400"))


def _mutants(b, rng, n):
    out = [b[:k] for k in (0, 1, 7, 8, 9, len(b) // 2, len(b) - 1)] + [b + b"\x00", b + b"\xff" * 9]
    for _ in range(n):
        m = bytearray(b)
        kind = rng.integers(0, 5)
        if kind == 0 and m:      # flip one bit
            i = int(rng.integers(0, len(m)))
            m[i] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1 and m:    # overwrite a byte with an extreme
            m[int(rng.integers(0, len(m)))] = int(rng.choice([0, 1, 0x7f, 0x80, 0xff]))
        elif kind == 2 and len(m) > 8:   # corrupt a length prefix-sized field somewhere on an 8-byte boundary
            i = 8 * int(rng.integers(0, len(m) // 8))
            m[i:i + 8] = (0, 1, 2, 2 ** 16, 2 ** 32, 2 ** 63, 2 ** 64 - 1)[int(rng.integers(0, 7))].to_bytes(8, "little")
        elif kind == 3 and len(m) > 2:   # cut a slice out
            i, j = sorted(int(x) for x in rng.integers(0, len(m), size=2))
            del m[i:j]
        else:                    # duplicate a slice
            i, j = sorted(int(x) for x in rng.integers(0, len(m) + 1, size=2))
            m[i:i] = m[i:j]
        out.append(bytes(m))
    return out


def _agree(product_read, oracle_read, mutants, what, usable=None):
    """product_read(b) -> re-serialised bytes or raises; oracle_read(b) -> re-serialised bytes or raises FormatError.
    usable(b): the structure parses but the library cannot build a handle from it (PoseidonConfig only) -> the product must reject"""
    accepted = 0
    for b in mutants:
        try:
            ob = oracle_read(b)
            if ob is not None and usable is not None and not usable(b):
                ob = None
        except O.FormatError:
            ob = None
        try:
            pb = product_read(b)
        except Exception as exc:  # a status code of the C ABI (serialize.py raises it as ValueError("akp error ...")), never a crash
            assert type(exc).__name__ in ("AkpError", "IncorrectInputLength") or (isinstance(exc, ValueError) and "akp error" in str(exc)), (what, type(exc), exc)
            pb = None
        assert (pb is None) == (ob is None), (what, "product %s, oracle %s" % ("rejects" if pb is None else "accepts", "rejects" if ob is None else "accepts"), b.hex()[:200])
        if pb is not None:
            assert pb == ob, (what, b.hex()[:200])
            accepted += 1
    return accepted


@pytest.mark.parametrize("compress", [False, True])
def test_mutated_paths_and_parameters(compress):
    import crypto_primitives_amd as cpa
    from crypto_primitives_amd import serialize as S
    from crypto_primitives_amd.crh import pedersen
    rng = np.random.default_rng(20260927 + int(compress))
    pts = [jj.mul(jj.GENERATOR, k) for k in (1, 2, 3, 5, 7, 11, 13, 17)]
    fes = [(3 ** (40 + i) % jj.Q,) for i in range(8)]
    total = 0
    for cfg, ds, fe in ((cpa.PedersenByteConfig, pts, 2), (cpa.PoseidonFieldConfig, fes, 1)):
        good = O.path(ds[0], ds[1:5], 9, compress)

        def p_read(b, cfg=cfg):
            return S.serialize_path(S.deserialize_path(b, cfg, compress), compress)

        def o_read(b, fe=fe):
            sib, auth, idx = O.read_path(b, fe, compress)
            return O.path(sib, auth, idx, compress)
        assert p_read(good) == o_read(good) == good
        total += _agree(p_read, o_read, _mutants(good, rng, N_MUTANTS), "Path")
        goodm = O.multi_path(ds[0:3], [0, 2, 1], [ds[3:5], [], ds[5:6]], [1, 2, 7], compress)

        def pm_read(b, cfg=cfg):
            return S.serialize_multi_path(S.deserialize_multi_path(b, cfg, compress), compress)

        def om_read(b, fe=fe):
            g = O.read_multi_path(b, fe, compress)
            return O.multi_path(g["leaf_siblings_hashes"], g["auth_paths_prefix_lenghts"], g["auth_paths_suffixes"], g["leaf_indexes"], compress)
        assert pm_read(goodm) == om_read(goodm) == goodm

        def m_usable(b, fe=fe):
            """the C ABI hands a MultiPath over as flat arrays of m paths: the four vectors must have one length (the derive of the
            reference reads any lengths; its verify then fails on the first missing element)"""
            g = O.read_multi_path(b, fe, compress)
            return len({len(g["leaf_siblings_hashes"]), len(g["auth_paths_prefix_lenghts"]), len(g["auth_paths_suffixes"]), len(g["leaf_indexes"])}) == 1
        total += _agree(pm_read, om_read, _mutants(goodm, rng, N_MUTANTS), "MultiPath", m_usable)
    gens = [pts[0:3], pts[3:6]]
    goodp = O.te_parameters(gens, compress)

    def pp_read(b):
        return S.serialize_te_parameters(S.deserialize_te_parameters(b, pedersen.Parameters, compress), compress)

    def op_read(b):
        return O.te_parameters(O.read_te_parameters(b, compress), compress)
    assert pp_read(goodp) == op_read(goodp) == goodp
    def p_usable(b):
        """generators are [num_windows][window_size]: every window holds the same number of points"""
        rows = O.read_te_parameters(b, compress)
        return len({len(r) for r in rows}) <= 1
    total += _agree(pp_read, op_read, _mutants(goodp, rng, N_MUTANTS), "Parameters", p_usable)
    assert total > 0  # some mutants stay valid (an index, a canonical field element) and must round-trip identically


def test_mutated_poseidon_config():
    import crypto_primitives_amd as cpa
    from crypto_primitives_amd import serialize as S, field
    rng = np.random.default_rng(77)
    c = cpa.get_default_poseidon_parameters(2, False)
    ints = lambda a: [int(v) for v in field.to_ints(np.asarray(a, dtype=np.uint64).reshape(-1, 4))]  # noqa: E731
    good = O.poseidon_config(c.full_rounds, c.partial_rounds, c.alpha, [ints(r) for r in c.ark], [ints(r) for r in c.mds], c.rate, c.capacity)

    def p_read(b):
        return S.serialize_poseidon_config(S.deserialize_poseidon_config(b))

    def o_read(b):
        r = O.read_poseidon_config(b)
        return O.poseidon_config(r["full_rounds"], r["partial_rounds"], r["alpha"], r["ark"], r["mds"], r["rate"], r["capacity"])
    assert p_read(good) == o_read(good) == good
    # the header words and the vector lengths are where the structure lives; mutate mostly there and at the very end
    head = [good[:k] + m[k:] for k in (0,) for m in _mutants(good[:200], rng, N_MUTANTS // 2)]
    head = [h + good[len(h):] if len(h) >= 200 else h for h in head]
    def usable(b):
        """what akp_poseidon_params_create needs beyond the byte format: the derive of the reference reads any field values (and the
        first permutation of an inconsistent config panics there); the library builds a handle while reading and refuses instead --
        ark has full + partial rows, every row and the square mds have t = rate + capacity <= 16 columns, rate >= 1, alpha >= 1,
        full_rounds even, round counts in 32 bits"""
        r = O.read_poseidon_config(b)
        t = r["rate"] + r["capacity"]
        return (r["rate"] >= 1 and 1 <= t <= 16 and r["alpha"] >= 1 and r["full_rounds"] % 2 == 0 and r["full_rounds"] + r["partial_rounds"] < 2 ** 32
                and len(r["ark"]) == r["full_rounds"] + r["partial_rounds"] and all(len(x) == t for x in r["ark"])
                and len(r["mds"]) == t and all(len(x) == t for x in r["mds"]))
    _agree(p_read, o_read, head + _mutants(good, rng, N_MUTANTS // 4), "PoseidonConfig", usable)
