"""C-ABI checks that need no GPU: the library loads, exports every symbol include/akp.h declares,
host-side parameter generation reproduces the reference KATs, and compute calls fail loudly
(no CPU fallback) when no device context exists."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "akp.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(akp_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    import crypto_primitives_amd as cpa
    L = C.CDLL(cpa.LIB_PATH)
    syms = _header_symbols()
    assert len(syms) >= 35
    for s in syms:
        assert hasattr(L, s), f"{s} declared in include/akp.h but not exported"
    # and the python binding declares prototypes for exactly that set
    assert sorted(cpa._lib.DECLARED_SYMBOLS) == syms
    assert cpa.lib.akp_abi_version() == cpa._lib.AKP_ABI_VERSION == 5


def test_library_exports_nothing_but_the_header():
    """the library is four translation units since round 3; what crosses between them (fail, ctx_scratch, launch_crh,
    te_crh_dev ...) has hidden visibility: the dynamic symbol table holds the header's entry points and nothing else"""
    import subprocess
    import crypto_primitives_amd as cpa
    out = subprocess.run(["nm", "-D", "--defined-only", cpa.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted(l.split()[-1] for l in out.splitlines() if len(l.split()) == 3 and l.split()[1] == "T")  # strong functions (weak = inline C++ library code)
    assert exported == _header_symbols(), sorted(set(exported) ^ set(_header_symbols()))


def test_product_never_imports_oracle():
    """the product package must not reference oracle/ (a CPU fallback would void parity claims)"""
    pkg = os.path.join(ROOT, "crypto_primitives_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h", ".inc")):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, re.M), f
                assert not re.search(r"#\s*include\s*[\"<][^\">]*oracle", txt), f
                assert "akp_oracle" not in txt and "libakp_oracle" not in txt, f


def test_default_parameters_match_reference_kats(kats):
    import crypto_primitives_amd as cpa
    from crypto_primitives_amd import field
    for e in kats["default_params"]:
        c = cpa.get_default_poseidon_parameters(e["rate"], e["optimized_for_weights"])
        assert field.to_ints(c.ark[0][0])[0] == int(e["ark00"])
        assert field.to_ints(c.mds[0][0])[0] == int(e["mds00"])
    assert cpa.get_default_poseidon_parameters(9) is None  # reference returns None
    assert cpa.get_default_poseidon_parameters(1) is None


def test_default_parameters_equal_oracle_everywhere():
    import crypto_primitives_amd as cpa
    from crypto_primitives_amd import field
    from oracle import poseidon as po
    for rate, w in ((2, False), (4, False), (8, True)):
        c = cpa.get_default_poseidon_parameters(rate, w)
        o = po.get_default_poseidon_parameters(rate, w)
        assert (c.full_rounds, c.partial_rounds, c.alpha, c.rate, c.capacity) == (o.full_rounds, o.partial_rounds, o.alpha, o.rate, o.capacity)
        assert field.to_ints(c.ark.reshape(-1, 4)) == [x for r in o.ark for x in r]
        assert field.to_ints(c.mds.reshape(-1, 4)) == [x for r in o.mds for x in r]


def test_field_conversion_roundtrip():
    from crypto_primitives_amd import field
    from oracle import fr as ofr
    vals = [0, 1, 2, field.MODULUS - 1, 12345678901234567890123456789]
    m = field.fr(vals)
    assert np.array_equal(m, ofr.ints_to_mont_array(vals))
    assert field.to_ints(m) == vals
    import crypto_primitives_amd as cpa
    bad = field.ints_to_canonical([0])
    bad[0] = [0xFFFFFFFFFFFFFFFF] * 4  # >= p
    out = np.empty_like(bad)
    assert cpa.lib.akp_fr_to_mont(bad.ctypes.data, out.ctypes.data, 1) == 2  # AKP_ERR_BAD_PARAMS


def test_params_validation_without_device():
    import crypto_primitives_amd as cpa
    from crypto_primitives_amd import field
    h = C.c_void_p()
    ark = field.fr(range(3 * 4)); mds = field.fr(range(9))
    # odd full_rounds, t too large, alpha 0
    assert cpa.lib.akp_poseidon_params_create(None, 3, 1, 5, 2, 1, ark.ctypes.data, mds.ctypes.data, C.byref(h)) == 2
    assert cpa.lib.akp_poseidon_params_create(None, 2, 2, 0, 2, 1, ark.ctypes.data, mds.ctypes.data, C.byref(h)) == 2
    assert cpa.lib.akp_poseidon_params_create(None, 2, 2, 5, 30, 1, ark.ctypes.data, mds.ctypes.data, C.byref(h)) == 2
    assert cpa.lib.akp_poseidon_params_create(None, 2, 2, 5, 2, 1, ark.ctypes.data, mds.ctypes.data, C.byref(h)) == 0
    # host-only handle: compute must fail loudly, never fall back to the CPU
    st = field.fr([0, 1, 2])
    assert cpa.lib.akp_poseidon_permute_batch(h, st.ctypes.data, 1) == 3  # AKP_ERR_HIP
    assert b"no CPU fallback" in cpa.lib.akp_last_error()
    cpa.lib.akp_poseidon_params_destroy(h)


def test_table_budget_and_shaped_create_argument_checks_without_device():
    """the table-shape entry points (akp.h: akp_ctx_set_table_budget, akp_te_params_create_shaped) reject NULL contexts and report no
    budget for them; curve tables are built on the GPU, so a create without a context is AKP_ERR_HIP, never a host-side table"""
    import crypto_primitives_amd as cpa
    from crypto_primitives_amd import field
    assert cpa.lib.akp_ctx_set_table_budget(None, 1 << 30) == 2
    assert cpa.lib.akp_ctx_table_budget(None) == 0
    gens = field.fr([0, 1] * 8).reshape(2, 4, 2, 4)
    h = C.c_void_p()
    for shape in (0, 5, 24):
        assert cpa.lib.akp_te_params_create_shaped(None, 0, 4, 2, gens.ctypes.data, shape, C.byref(h)) == 3
        assert b"device context is required" in cpa.lib.akp_last_error()


def test_no_device_means_loud_failure():
    import crypto_primitives_amd as cpa
    if cpa.lib.akp_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(cpa.AkpError):
        cpa.Context(0)
    cfg = cpa.get_default_poseidon_parameters(2)
    from crypto_primitives_amd.crh import poseidon as pcrh
    from crypto_primitives_amd import field
    with pytest.raises(cpa.AkpError):
        pcrh.CRH.evaluate(cfg, field.fr([1, 2]))


def test_seeded_generators_equal_oracle(derived):
    from crypto_primitives_amd import params, field
    from oracle import jubjub as jj
    g = params.pedersen_generators(0xA5A50004, 4, 2)
    go = jj.pedersen_generators(0xA5A50004, 4, 2)
    assert field.to_ints(g) == [v for row in go for pt in row for v in pt]
    assert [str(v) for v in field.to_ints(g[0][0])] == derived["pedersen_4x256"]["g00"]
    gb = params.bowe_hopwood_generators(0xA5A50005, 3, 2)
    gbo = jj.bowe_hopwood_generators(0xA5A50005, 3, 2)
    assert field.to_ints(gb) == [v for row in gbo for pt in row for v in pt]


def test_setup_generators_have_no_known_dlog_and_equal_oracle():
    """`setup()` must not produce bases k_i * G with public k_i.  The product samples points like ark-ec's
    `rand` (random y, sign, solve for x, clear the cofactor); the oracle restates the procedure independently."""
    from crypto_primitives_amd import params, field
    from oracle import jubjub as jj
    g = params.setup_pedersen_generators(5, 4, 3)
    go = jj.pedersen_generators(5, 4, 3, bases=jj.random_bases)
    assert field.to_ints(g) == [v for row in go for pt in row for v in pt]
    gb = params.setup_bowe_hopwood_generators(6, 3, 2)
    gbo = jj.bowe_hopwood_generators(6, 3, 2, bases=jj.random_bases)
    assert field.to_ints(gb) == [v for row in gbo for pt in row for v in pt]
    for row in go:
        assert jj.is_on_curve(row[0]) and jj.mul(row[0], jj.SUBGROUP_ORDER) == jj.IDENTITY and row[0] != jj.IDENTITY  # prime-order subgroup
        assert row[1] == jj.double(row[0])
    # not the test-data bases, and not a small multiple of the standard generator
    assert go[0][0] != jj.pedersen_generators(5, 4, 3)[0][0]
    assert all(jj.mul(jj.GENERATOR, k) != go[0][0] for k in range(1, 64))
    r = jj.fq_sqrt(1234567 ** 2 % jj.Q)
    assert r in (1234567, jj.Q - 1234567) and jj.fq_sqrt(jj.D) is None  # d is a non-square (completeness of the addition law)


def test_multipath_encode_decode_host(derived):
    """prefix_encode_path / prefix_decode_path (merkle_tree/mod.rs:795-817) through the ABI, no GPU"""
    import ctypes as C
    import crypto_primitives_amd as cpa
    rng = np.random.default_rng(3)
    m, depth = 6, 4
    auth = rng.integers(0, 1 << 62, size=(m, depth, 4), dtype=np.uint64)
    auth[1, :2] = auth[0, :2]
    auth[2] = auth[1]
    auth[4, :3] = auth[3, :3]
    pre = np.zeros(m, np.uint64)
    suf = np.zeros((m * depth, 4), np.uint64)
    cnt = C.c_size_t()
    assert cpa.lib.akp_merkle_multipath_encode(auth.ctypes.data, m, depth, 1, pre.ctypes.data, suf.ctypes.data, C.byref(cnt)) == 0
    assert pre.tolist() == [0, 2, 4, 0, 3, 0] and cnt.value == 4 + 2 + 0 + 4 + 1 + 4
    back = np.zeros_like(auth)
    assert cpa.lib.akp_merkle_multipath_decode(pre.ctypes.data, suf.ctypes.data, cnt.value, m, depth, 1, back.ctypes.data) == 0
    assert np.array_equal(back, auth)
    assert cpa.lib.akp_merkle_multipath_decode(pre.ctypes.data, suf.ctypes.data, cnt.value - 1, m, depth, 1, back.ctypes.data) == 2
    pre[0] = 1  # the first path has nothing to share a prefix with
    assert cpa.lib.akp_merkle_multipath_decode(pre.ctypes.data, suf.ctypes.data, cnt.value, m, depth, 1, back.ctypes.data) == 2


def test_multipath_encode_decode_random_and_hostile_inputs():
    """akp_merkle_multipath_encode / _decode (host only) on random path sets against a direct restatement of prefix_encode_path /
    prefix_decode_path (merkle_tree/mod.rs:795-817), then the decoder on hostile inputs -- prefix lengths beyond the depth or near
    2^64, a first path with a prefix, too few / too many suffix digests: a status code, nothing written past the buffers
    (tests/test_sanitizers.py runs this against the ASan / UBSan builds of the library)"""
    import ctypes as C
    import crypto_primitives_amd as cpa
    rng = np.random.default_rng(11)
    for trial in range(60):
        m, depth, fe = int(rng.integers(1, 9)), int(rng.integers(0, 7)), int(rng.integers(1, 3))
        auth = rng.integers(0, 1 << 62, size=(m, max(depth, 1), fe * 4), dtype=np.uint64)[:, :depth]
        for i in range(1, m):  # share a random prefix with the previous path
            k = int(rng.integers(0, depth + 1))
            auth[i, :k] = auth[i - 1, :k]
        auth = np.ascontiguousarray(auth)
        want_pre, want_suf = [], []
        for i in range(m):
            k = 0
            if i:
                while k < depth and np.array_equal(auth[i, k], auth[i - 1, k]):
                    k += 1
            want_pre.append(k)
            want_suf.extend(auth[i, k:].reshape(-1, fe * 4))
        pre = np.zeros(m, np.uint64)
        suf = np.zeros((max(m * depth, 1), fe * 4), np.uint64)
        cnt = C.c_size_t()
        assert cpa.lib.akp_merkle_multipath_encode(auth.ctypes.data if auth.size else None, m, depth, fe, pre.ctypes.data, suf.ctypes.data, C.byref(cnt)) == 0
        assert pre.tolist() == want_pre and cnt.value == len(want_suf)
        assert cnt.value == 0 or np.array_equal(suf[:cnt.value], np.asarray(want_suf))
        back = np.full_like(auth, 7) if auth.size else auth
        assert cpa.lib.akp_merkle_multipath_decode(pre.ctypes.data, suf.ctypes.data, cnt.value, m, depth, fe, back.ctypes.data if back.size else None) == 0
        assert np.array_equal(back, auth)
        # hostile decodes: never a crash, never success
        guard = np.full((m * max(depth, 1) + 2, fe * 4), 0xA5A5A5A5A5A5A5A5, np.uint64)  # the output buffer with two digests of canary behind it
        out = guard[: m * max(depth, 1)]
        for bad_pre, bad_cnt in ((pre.copy(), cnt.value + 1), (pre.copy(), max(cnt.value, 1) - 1 if cnt.value else 1),
                                 (np.where(np.arange(m) == m - 1, depth + 1, pre).astype(np.uint64), cnt.value),
                                 (np.where(np.arange(m) == 0, 1, pre).astype(np.uint64), cnt.value),
                                 (np.full(m, (1 << 64) - 1, np.uint64), cnt.value), (np.full(m, 1 << 63, np.uint64), 0)):
            if depth == 0 and bad_cnt == cnt.value and np.array_equal(bad_pre, pre):
                continue
            rc = cpa.lib.akp_merkle_multipath_decode(bad_pre.ctypes.data, suf.ctypes.data, bad_cnt, m, depth, fe, out.ctypes.data)
            assert rc in (1, 2), (trial, m, depth, bad_pre.tolist(), bad_cnt, rc)
            assert np.all(guard[m * max(depth, 1):] == 0xA5A5A5A5A5A5A5A5)


def test_multi_device_and_tree_entry_points_fail_loudly_without_gpu():
    import ctypes as C
    import crypto_primitives_amd as cpa
    if cpa.lib.akp_device_count() > 0:
        pytest.skip("a device is present")
    ids = (C.c_int32 * 1)(0)
    h = C.c_void_p()
    assert cpa.lib.akp_multi_create(ids, 1, C.byref(h)) == cpa._lib.AKP_ERR_HIP
    cfg = cpa.get_default_poseidon_parameters(2, False)
    hp = C.c_void_p()
    cpa._lib.check(cpa.lib.akp_poseidon_default_params(None, 2, 0, C.byref(hp)))
    leaves = np.zeros((4, 4), np.uint64)
    t = C.c_void_p()
    assert cpa.lib.akp_merkle_tree_build_poseidon(hp, hp, leaves.ctypes.data, 4, 1, C.byref(t)) == cpa._lib.AKP_ERR_HIP
    assert cpa.lib.akp_poseidon_kernel_for(hp, 1 << 20, 0) == b"none"
    cpa.lib.akp_poseidon_params_destroy(hp)


def test_gather_paths_host_matches_oracle_indexing():
    """akp_merkle_gather_paths is pure index arithmetic (no GPU): compare with the oracle tree's compute_auth_path"""
    import crypto_primitives_amd as cpa
    from oracle import merkle as omk
    n = 32
    leaf_nodes = np.arange(n * 4, dtype=np.uint64).reshape(n, 4) + 1000
    non_leaf = np.arange((n - 1) * 4, dtype=np.uint64).reshape(n - 1, 4)
    t = omk.MerkleTree.__new__(omk.MerkleTree)
    t.leaf_nodes = [tuple(x) for x in leaf_nodes]
    t.non_leaf_nodes = [tuple(x) for x in non_leaf]
    t.height = 6
    idx = np.array([0, 1, 7, 16, 31], dtype=np.uint64)
    sib = np.zeros((5, 4), np.uint64); auth = np.zeros((5, 4, 4), np.uint64)
    assert cpa.lib.akp_merkle_gather_paths(leaf_nodes.ctypes.data, non_leaf.ctypes.data, n, 1, idx.ctypes.data, 5, sib.ctypes.data, auth.ctypes.data) == 0
    for k, i in enumerate(idx):
        assert tuple(sib[k]) == t.get_leaf_sibling_hash(int(i))
        assert [tuple(a) for a in auth[k]] == t.compute_auth_path(int(i))
    assert cpa.lib.akp_merkle_gather_paths(leaf_nodes.ctypes.data, non_leaf.ctypes.data, 24, 1, idx.ctypes.data, 5, sib.ctypes.data, auth.ctypes.data) == 5
    bad = np.array([32], dtype=np.uint64)
    assert cpa.lib.akp_merkle_gather_paths(leaf_nodes.ctypes.data, non_leaf.ctypes.data, n, 1, bad.ctypes.data, 1, sib.ctypes.data, auth.ctypes.data) == 2
