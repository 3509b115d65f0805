"""Consumer of tests/golden/reference_vectors.json -- the file shim/examples/emit_vectors.rs writes by running the REFERENCE
crates (pedersen::CRH, bowe_hopwood::CRH, both TwoToOneCRH::{evaluate, compress}, poseidon CRH / sponge, MerkleTree::{new,
generate_proof, generate_multi_proof, update}, CanonicalSerialize) on tests/golden/emitter_inputs.json.

The build image has no Rust toolchain, so the file cannot be produced here: while it is absent the pinning tests SKIP with the
reason "parity unpinned (emitter not run)" -- one `cargo run --example emit_vectors` in shim/ turns them on.  What does run
everywhere:
  * the committed inputs are current (regenerating them reproduces the file);
  * the consumer itself works: the oracle's vectors in the emitter's layout compare equal to themselves through the same
    `diff`, and a flipped digest / byte is reported (so a present-but-different reference file cannot pass silently);
  * (-m gpu) the GPU product reproduces the oracle's vectors on exactly the emitter's inputs, serialised forms included.
"""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, GOLD)
import reference_schema as rs  # noqa: E402

from oracle import poseidon as po, jubjub as jj, pedersen as opd, bowe_hopwood as obh, merkle as omk, serialize as oser  # noqa: E402

UNPINNED = ("parity unpinned (emitter not run): tests/golden/reference_vectors.json is absent -- run "
            "`cargo run --release --example emit_vectors` in shim/ on a machine with a Rust toolchain")


@pytest.fixture(scope="module")
def inputs():
    with open(os.path.join(GOLD, "emitter_inputs.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def reference():
    p = os.path.join(GOLD, "reference_vectors.json")
    if not os.path.exists(p):
        pytest.skip(UNPINNED)
    with open(p) as f:
        return json.load(f)


# ---- serialisation through the product's host code (crypto_primitives_amd/serialize.py; no GPU involved) ---------------
class Ser:
    def __init__(self):
        import crypto_primitives_amd as cpa
        from crypto_primitives_amd import serialize, field
        self.cpa, self.s, self.f = cpa, serialize, field

    def _cfg(self, kind):
        return {"pedersen": self.cpa.PedersenByteConfig, "bowe_hopwood": self.cpa.BoweHopwoodByteConfig, "poseidon": self.cpa.PoseidonFieldConfig}[kind]

    def _w(self, kind, d):  # digest tuple -> wire array of the config's digest shape
        return self.f.fr([int(v) for v in d]).reshape(self._cfg(kind).digest_shape)

    def digest(self, kind, d, compress):
        return self.s._digest_bytes(self._w(kind, d), compress)

    def path(self, kind, sib, auth, idx, compress):
        P = self.cpa.merkle_tree.Path(self._cfg(kind), self._w(kind, sib), [self._w(kind, a) for a in auth], idx)
        return self.s.serialize_path(P, compress)

    def multi_path(self, kind, mp, compress):
        M = self.cpa.merkle_tree.MultiPath(self._cfg(kind), [self._w(kind, d) for d in mp["leaf_siblings_hashes"]], list(mp["auth_paths_prefix_lenghts"]),
                                           [[self._w(kind, d) for d in s] for s in mp["auth_paths_suffixes"]], list(mp["leaf_indexes"]))
        return self.s.serialize_multi_path(M, compress)

    def te_parameters(self, gens, compress):
        from crypto_primitives_amd.crh import pedersen
        g = self.f.fr([int(v) for row in gens for p in row for v in p]).reshape(len(gens), len(gens[0]), 2, 4)
        return self.s.serialize_te_parameters(pedersen.Parameters(g), compress)

    def poseidon_config(self, sec, compress):
        from crypto_primitives_amd.sponge.poseidon import PoseidonConfig
        t = sec["rate"] + sec["capacity"]
        ark = self.f.fr([int(x) for r in sec["ark"] for x in r]).reshape(-1, t, 4)
        mds = self.f.fr([int(x) for r in sec["mds"] for x in r]).reshape(t, t, 4)
        return self.s.serialize_poseidon_config(PoseidonConfig(sec["full_rounds"], sec["partial_rounds"], sec["alpha"], ark, mds, sec["rate"], sec["capacity"]))


# ---- serialisation through the ORACLE's independent restatement (oracle/serialize.py: python ints, nothing of the product) ----
class OracleSer:
    def digest(self, kind, d, compress):
        return oser.digest(d, compress)

    def path(self, kind, sib, auth, idx, compress):
        return oser.path(sib, auth, idx, compress)

    def multi_path(self, kind, mp, compress):
        return oser.multi_path(mp["leaf_siblings_hashes"], mp["auth_paths_prefix_lenghts"], mp["auth_paths_suffixes"], mp["leaf_indexes"], compress)

    def te_parameters(self, gens, compress):
        return oser.te_parameters([[(int(p[0]), int(p[1])) for p in row] for row in gens], compress)

    def poseidon_config(self, sec, compress):
        return oser.poseidon_config(sec["full_rounds"], sec["partial_rounds"], sec["alpha"], [[int(x) for x in r] for r in sec["ark"]],
                                    [[int(x) for x in r] for r in sec["mds"]], sec["rate"], sec["capacity"], compress)


# ---- implementation 1: the python oracle --------------------------------------------------------------------------------
class _OracleTree:
    def __init__(self, t, wrap):
        self.t, self.w = t, wrap

    def root(self):
        return self.w(self.t.root())

    def height(self):
        return self.t.height

    def proof(self, i):
        p = self.t.generate_proof(i)
        return self.w(p.leaf_sibling_hash), [self.w(a) for a in p.auth_path], p.leaf_index

    def multi_proof(self, idxs):
        m = self.t.generate_multi_proof(idxs)
        return {"leaf_indexes": m["leaf_indexes"], "auth_paths_prefix_lenghts": m["auth_paths_prefix_lenghts"],
                "auth_paths_suffixes": [[self.w(d) for d in s] for s in m["auth_paths_suffixes"]],
                "leaf_siblings_hashes": [self.w(d) for d in m["leaf_siblings_hashes"]]}

    def update(self, i, leaf):
        self.t.update(i, leaf)


class OracleImpl:
    def __init__(self):
        self.ser = OracleSer()  # round 4: NOT the product's serializer -- every "uncompressed" / "compressed" field below is oracle bytes

    def curve(self, kind, sec):
        W, N = sec["window_size"], sec["num_windows"]
        g = [[(int(p[0]), int(p[1])) for p in row] for row in sec["generators"]]
        assert all(jj.is_on_curve(p) for row in g for p in row)
        if kind == "pedersen":
            class H:
                crh = staticmethod(lambda m: tuple(opd.evaluate(g, W, N, m)))
                two_to_one_evaluate = staticmethod(lambda l, r: tuple(opd.two_to_one_evaluate(g, W, N, l, r)))
                two_to_one_compress = staticmethod(lambda a, b: tuple(opd.two_to_one_compress(g, W, N, tuple(a), tuple(b))))
                tree = staticmethod(lambda leaves: _OracleTree(omk.MerkleTree(
                    lambda leaf: opd.evaluate(g, W, N, leaf), lambda a, b: opd.two_to_one_evaluate(g, W, N, a, b),
                    lambda a, b: opd.two_to_one_compress(g, W, N, a, b), jj.serialize_uncompressed, leaves=leaves), tuple))
            return H

        class H:
            crh = staticmethod(lambda m: (obh.evaluate(g, W, N, m),))
            two_to_one_evaluate = staticmethod(lambda l, r: (obh.two_to_one_evaluate(g, W, N, l, r),))
            two_to_one_compress = staticmethod(lambda a, b: (obh.two_to_one_compress(g, W, N, a[0], b[0]),))
            tree = staticmethod(lambda leaves: _OracleTree(omk.MerkleTree(
                lambda leaf: obh.evaluate(g, W, N, leaf), lambda a, b: obh.two_to_one_evaluate(g, W, N, a, b),
                lambda a, b: obh.two_to_one_compress(g, W, N, a, b), jj.fq_serialize, leaves=leaves), lambda x: (x,)))
        return H

    def poseidon(self, sec):
        c = po.PoseidonConfig(sec["full_rounds"], sec["partial_rounds"], sec["alpha"], [[int(x) for x in r] for r in sec["ark"]],
                              [[int(x) for x in r] for r in sec["mds"]], sec["rate"], sec["capacity"])

        def script(a1, n1, a2, n2, nbytes, nbits):
            sp = po.PoseidonSponge(c)
            sp.absorb(a1)
            s1 = sp.squeeze_native_field_elements(n1)
            sp.absorb(a2)
            s2 = sp.squeeze_native_field_elements(n2)
            fork = po.PoseidonSponge(c)
            fork.state, fork.mode = list(sp.state), sp.mode
            return s1, s2, fork.squeeze_bytes(nbytes), sp.squeeze_bits(nbits)

        class H:
            crh = staticmethod(lambda x: (po.crh_evaluate(c, x),))
            two_to_one = staticmethod(lambda l, r: (po.two_to_one_compress(c, l, r),))
            sponge_script = staticmethod(script)
            tree = staticmethod(lambda leaves: _OracleTree(omk.MerkleTree(
                lambda leaf: po.crh_evaluate(c, leaf), lambda a, b: po.two_to_one_compress(c, a, b),
                lambda a, b: po.two_to_one_compress(c, a, b), lambda x: x, leaves=leaves), lambda x: (x,)))
        return H


# ---- implementation 2: the GPU product through its host mirror -------------------------------------------------------------
class _GpuTree:
    def __init__(self, cpa, config, lp, tp, leaves, wrap_leaf):
        self.cpa, self.wl = cpa, wrap_leaf
        self.t = cpa.GpuMerkleTree.new(config, lp, tp, leaves)  # the HBM-resident handle: proofs and updates served from the device

    def _i(self, d):
        from crypto_primitives_amd import field
        return tuple(field.to_ints(np.asarray(d).reshape(-1, 4)))

    def root(self):
        return self._i(self.t.root())

    def height(self):
        return self.t.height()

    def proof(self, i):
        p = self.t.generate_proof(i)
        return self._i(p.leaf_sibling_hash), [self._i(a) for a in p.auth_path], p.leaf_index

    def multi_proof(self, idxs):
        m = self.t.generate_multi_proof(idxs)
        return {"leaf_indexes": list(m.leaf_indexes), "auth_paths_prefix_lenghts": list(m.auth_paths_prefix_lenghts),
                "auth_paths_suffixes": [[self._i(d) for d in s] for s in m.auth_paths_suffixes],
                "leaf_siblings_hashes": [self._i(d) for d in m.leaf_siblings_hashes]}

    def update(self, i, leaf):
        self.t.update(i, self.wl(leaf))


class GpuImpl:
    def __init__(self):
        import crypto_primitives_amd as cpa
        self.cpa, self.ser = cpa, Ser()

    def curve(self, kind, sec):
        from crypto_primitives_amd import field
        from crypto_primitives_amd.crh import pedersen, bowe_hopwood
        cpa = self.cpa
        W, N = sec["window_size"], sec["num_windows"]
        g = field.fr([int(v) for row in sec["generators"] for p in row for v in p]).reshape(N, W, 2, 4)
        ped = kind == "pedersen"
        mod = pedersen if ped else bowe_hopwood
        P = mod.Parameters(g)
        cfg = cpa.PedersenByteConfig if ped else cpa.BoweHopwoodByteConfig
        shp = cfg.digest_shape

        def ti(d):
            return tuple(field.to_ints(np.asarray(d).reshape(-1, 4)))

        class H:
            crh = staticmethod(lambda m: ti(mod.CRH.evaluate(P, m)))
            two_to_one_evaluate = staticmethod(lambda l, r: ti(mod.TwoToOneCRH.evaluate(P, l, r)))
            two_to_one_compress = staticmethod(lambda a, b: ti(mod.TwoToOneCRH.compress(P, field.fr(list(a)).reshape(shp), field.fr(list(b)).reshape(shp))))
            tree = staticmethod(lambda leaves: _GpuTree(cpa, cfg, P, P, leaves, bytes))
        return H

    def poseidon(self, sec):
        from crypto_primitives_amd import field
        from crypto_primitives_amd.sponge.poseidon import PoseidonConfig, PoseidonSponge
        from crypto_primitives_amd.crh import poseidon as pcrh
        cpa = self.cpa
        t = sec["rate"] + sec["capacity"]
        c = PoseidonConfig(sec["full_rounds"], sec["partial_rounds"], sec["alpha"], field.fr([int(x) for r in sec["ark"] for x in r]).reshape(-1, t, 4),
                           field.fr([int(x) for r in sec["mds"] for x in r]).reshape(t, t, 4), sec["rate"], sec["capacity"])

        def script(a1, n1, a2, n2, nbytes, nbits):
            sp = PoseidonSponge(c)
            sp.absorb(field.fr(a1))
            s1 = field.to_ints(sp.squeeze_native_field_elements(n1).reshape(-1, 4))
            sp.absorb(field.fr(a2))
            s2 = field.to_ints(sp.squeeze_native_field_elements(n2).reshape(-1, 4))
            fork = sp.clone()
            return s1, s2, fork.squeeze_bytes(nbytes), sp.squeeze_bits(nbits)

        class H:
            crh = staticmethod(lambda x: tuple(field.to_ints(np.asarray(pcrh.CRH.evaluate(c, field.fr(x) if x else np.zeros((0, 4), np.uint64))).reshape(-1, 4))))
            two_to_one = staticmethod(lambda l, r: tuple(field.to_ints(np.asarray(pcrh.TwoToOneCRH.compress(c, field.fr([l]), field.fr([r]))).reshape(-1, 4))))
            sponge_script = staticmethod(script)
            tree = staticmethod(lambda leaves: _GpuTree(cpa, cpa.PoseidonFieldConfig, c, c, np.stack([field.fr(l) for l in leaves]), lambda leaf: field.fr(leaf)))
        return H


# ---- tests -----------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def oracle_vectors(inputs):
    return rs.build_vectors(inputs, OracleImpl())


def test_emitter_inputs_are_current(inputs, tmp_path, monkeypatch):
    """the committed inputs are exactly what tests/golden/make_emitter_inputs.py writes"""
    import make_emitter_inputs as mk
    monkeypatch.setattr(mk, "OUT", str(tmp_path / "inputs.json"))
    mk.main()
    assert json.load(open(tmp_path / "inputs.json")) == inputs


def test_consumer_detects_differences(oracle_vectors):
    """self-check of the comparison (NOT a pin): equal to itself, and one changed digest / serialised byte / list length shows"""
    bad, only_ref, only_ours = rs.diff(oracle_vectors, json.loads(json.dumps(oracle_vectors)))
    assert (bad, only_ref, only_ours) == ([], [], [])
    mut = json.loads(json.dumps(oracle_vectors))
    mut["pedersen"]["crh"][3]["digest"][0] = str(int(mut["pedersen"]["crh"][3]["digest"][0]) ^ 1)
    h = mut["bowe_hopwood"]["tree"]["proofs"][2]["uncompressed"]
    mut["bowe_hopwood"]["tree"]["proofs"][2]["uncompressed"] = h[:-1] + ("0" if h[-1] != "0" else "1")
    mut["poseidon"]["tree"]["multi_proof"]["auth_paths_prefix_lenghts"].append(0)
    del mut["pedersen"]["parameters_head"]
    bad, only_ref, only_ours = rs.diff(mut, oracle_vectors)
    assert "/pedersen/crh[3]/digest[0]" in bad and "/bowe_hopwood/tree/proofs[2]/uncompressed" in bad
    assert any(b.startswith("/poseidon/tree/multi_proof/auth_paths_prefix_lenghts") for b in bad)
    assert only_ours == ["/pedersen/parameters_head"] and only_ref == []
    # structure sanity: the generator-independent known answers the reference's code implies
    assert oracle_vectors["pedersen"]["crh"][0]["digest"] == ["0", "1"]          # empty message -> identity (crh/pedersen/mod.rs:116-122)
    assert oracle_vectors["bowe_hopwood"]["crh"][0]["digest"] == ["0"]           # empty message -> x of the identity
    assert oracle_vectors["poseidon"]["tree"]["multi_proof"]["leaf_indexes"] == [0, 1, 5, 6]
    assert all(len(e["uncompressed"]) == 128 and len(e["compressed"]) == 64 for e in oracle_vectors["pedersen"]["crh"])


class _ProductSerOnOracleValues(OracleImpl):
    """the ORACLE's hashes and trees, serialised by the PRODUCT (serialize.py -> C ABI): differs from `oracle_vectors` only in who
    wrote the bytes, so the diff below is a byte-for-byte comparison of the two serialisers on every emitted struct.  No GPU."""

    def __init__(self):
        self.ser = Ser()


def test_product_serializer_equals_the_oracle_serializer(inputs, oracle_vectors):
    ours = rs.build_vectors(inputs, _ProductSerOnOracleValues())
    bad, only_ref, only_ours = rs.diff(oracle_vectors, ours)
    assert (bad, only_ref, only_ours) == ([], [], [])
    n_hex = sum(1 for k, v in _walk(oracle_vectors) if k.endswith(("uncompressed", "compressed")) and isinstance(v, str))
    assert n_hex >= 100  # digests, paths, multi-paths, parameters and configs, both modes


def _walk(o, at=""):
    if isinstance(o, dict):
        for k, v in o.items():
            yield from _walk(v, at + "/" + k)
    elif isinstance(o, list):
        for i, v in enumerate(o):
            yield from _walk(v, "%s[%d]" % (at, i))
    else:
        yield at, o


@pytest.mark.parametrize("mutation", ["swap_path_fields", "drop_vec_prefix", "flag_polarity", "config_field_order"])
def test_a_wrong_product_serializer_is_caught(inputs, oracle_vectors, monkeypatch, mutation):
    """the comparison above has teeth: a field-order or length-prefix mistake in the product serialiser (simulated by
    monkey-patching serialize.py) is reported at the struct it breaks"""
    from crypto_primitives_amd import serialize as S
    if mutation == "swap_path_fields":  # leaf_index written before auth_path
        good = S.serialize_path

        def bad_path(path, compress=False):
            b = good(path, compress)
            per = 32 if (compress or S._fe_of(path.config) == 1) else 64
            return b[:per] + b[-8:] + b[per:-8]
        monkeypatch.setattr(S, "serialize_path", bad_path)
        where = "/tree/proofs[0]/uncompressed"
    elif mutation == "drop_vec_prefix":  # the Vec<usize> of leaf indexes without its length
        good = S.serialize_multi_path

        def bad_mp(mp, compress=False):
            b = good(mp, compress)
            m = len(mp.leaf_indexes)
            return b[:len(b) - 8 * m - 8] + b[len(b) - 8 * m:]
        monkeypatch.setattr(S, "serialize_multi_path", bad_mp)
        where = "/tree/multi_proof/uncompressed"
    elif mutation == "flag_polarity":  # sign flag set for the SMALLER of (x, -x)
        good = S.digests_bytes

        def bad_dig(d, fe, compress=False):
            b = bytearray(good(d, fe, compress))
            if fe == 2 and compress:
                for i in range(31, len(b), 32):
                    b[i] ^= 0x80
            return bytes(b)
        monkeypatch.setattr(S, "digests_bytes", bad_dig)
        where = "/pedersen/crh[1]/compressed"
    else:  # rate / capacity ahead of the matrices
        good = S.serialize_poseidon_config

        def bad_cfg(cfg):
            b = good(cfg)
            return b[:24] + b[-16:] + b[24:-16]
        monkeypatch.setattr(S, "serialize_poseidon_config", bad_cfg)
        where = "/poseidon/config_uncompressed"
    section = ("poseidon",) if mutation == "config_field_order" else ("pedersen",)
    ours = rs.build_vectors(inputs, _ProductSerOnOracleValues(), sections=section)
    bad, _, _ = rs.diff({k: oracle_vectors[k] for k in section}, ours)
    assert any(b.endswith(where) or where in b for b in bad), (mutation, bad[:5])


def test_reference_vectors_pin_the_oracle(reference, inputs, oracle_vectors):
    """WITH the emitter's file: every value the reference produced is reproduced by the python oracle and by oracle/serialize.py"""
    assert reference["poseidon"]["reference_generator_matches_inputs"] is True
    bad, only_ref, _ = rs.diff({k: reference[k] for k in ("pedersen", "bowe_hopwood", "poseidon")}, oracle_vectors)
    assert not bad, "the oracle / serialize.py differ from the REFERENCE at: %s" % bad[:20]
    assert not only_ref, "the emitter wrote fields this consumer does not know: %s" % only_ref[:20]


def test_reference_vectors_pin_the_c_oracle(reference, inputs):
    """the bulk C oracle (what the GPU parity tests and bench.py use as checker) on the emitter's messages"""
    from oracle import cref
    from helpers import gens_array, ints
    for kind, W, N in (("pedersen", 4, 256), ("bowe_hopwood", 63, 9)):
        sec = inputs[kind]
        g = [[(int(p[0]), int(p[1])) for p in row] for row in sec["generators"]]
        C = cref.CurveParams(W, N, gens_array(g))
        for e in reference[kind]["crh"]:
            m = np.frombuffer(bytes.fromhex(e["msg"]), np.uint8)
            fn = C.pedersen_crh_batch if kind == "pedersen" else C.bh_crh_batch
            got = fn(m if len(m) else np.zeros(1, np.uint8), 1, len(m), threads=1)
            assert [str(v) for v in ints(np.asarray(got).reshape(-1, 4))] == e["digest"], (kind, len(m))


@pytest.mark.gpu
def test_gpu_reproduces_the_oracle_on_the_emitter_inputs(inputs, oracle_vectors):
    """always on the GPU box: the product (HBM-resident trees, batched proofs, update, sponge, serialisation) against the oracle
    on exactly the inputs the reference emitter consumes -- so the day the reference file exists, oracle == reference implies
    GPU == reference for every emitted value"""
    ours = rs.build_vectors(inputs, GpuImpl())
    bad, only_ref, only_ours = rs.diff(oracle_vectors, ours)
    assert (bad, only_ref, only_ours) == ([], [], [])


@pytest.mark.gpu
def test_reference_vectors_pin_the_gpu_path(reference, inputs):
    """WITH the emitter's file: the GPU path against the reference's own outputs, directly"""
    ours = rs.build_vectors(inputs, GpuImpl())
    bad, only_ref, _ = rs.diff({k: reference[k] for k in ("pedersen", "bowe_hopwood", "poseidon")}, ours)
    assert not bad, "the GPU path differs from the REFERENCE at: %s" % bad[:20]
    assert not only_ref
