"""Tests that need the TEST BUILD of the library (libakp_testhooks.so, -DAKP_TEST_HOOKS):
  * the G > 1 logic of the multi-device entry points on a ONE-GPU box: sharded build (akp_merkle_build_sharded_*) and the sharded
    RESIDENT tree (akp_multi_tree_*) with G = 2 / 4 / 8 device slots that all name device 0;
  * the settled A/B arms that the product library no longer reads from its environment: the plain Pedersen table
    (AKP_PEDERSEN_PLAIN) and the chunk-by-chunk walk of a zero-padded Bowe-Hopwood tail (AKP_BH_ZERO_TAIL=0).

That needs the shared-device test hook of akp_multi_create, which exists only in the test build of the library
(`make -C crypto_primitives_amd/csrc testhooks` -> lib/libakp_testhooks.so, -DAKP_TEST_HOOKS): the product libakp.so has no such
switch.  `test_run_with_the_testhooks_library` (always collected) re-runs this file in a child process whose AKP_LIB
points at the test build; the slot tests themselves skip when the loaded library is the product one.  RCCL itself runs at
n_dev = 1 in tests/test_gpu_tree_handle.py and over real devices in the driver's scaling run.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import jubjub as jj, poseidon as po, merkle as omk, fr as ofr  # noqa: E402
from helpers import rand_fr_array, gens_array, cref_poseidon  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOOKS_LIB = os.path.join(ROOT, "crypto_primitives_amd", "lib", "libakp_testhooks.so")
IN_CHILD = os.environ.get("AKP_LIB", "").endswith("libakp_testhooks.so")
needs_hooks = pytest.mark.skipif(not IN_CHILD, reason="runs in the child process of test_run_with_the_testhooks_library (AKP_LIB = libakp_testhooks.so)")


@pytest.fixture(scope="module")
def cpa():
    import crypto_primitives_amd as m
    assert m.lib.akp_device_count() >= 1
    return m


def _t(a):
    return tuple(int(x) for x in np.asarray(a).reshape(-1))


def test_run_with_the_testhooks_library():
    if IN_CHILD:
        pytest.skip("this is the child")
    assert os.path.exists(HOOKS_LIB), "build it: make -C crypto_primitives_amd/csrc testhooks (python -c 'import __graft_entry__ as g; g.build()')"
    env = dict(os.environ, AKP_LIB=HOOKS_LIB)
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider"], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=1500)
    tail = (p.stdout + p.stderr)[-3000:]
    assert p.returncode == 0, tail
    assert " passed" in tail and "failed" not in tail, tail


def test_the_product_library_has_no_shared_device_switch(cpa, monkeypatch):
    if IN_CHILD:
        pytest.skip("product library only")
    monkeypatch.setenv("AKP_MULTI_TEST_SHARED_DEVICE", "1")
    h = C.c_void_p()
    assert cpa.lib.akp_multi_create((C.c_int32 * 2)(0, 0), 2, C.byref(h)) == cpa._lib.AKP_ERR_BAD_PARAMS
    assert b"listed twice" in cpa.lib.akp_last_error()


@needs_hooks
@pytest.mark.parametrize("G", [2, 4, 8])
def test_sharded_build_logic_for_several_slots_on_one_device(cpa, G, monkeypatch):
    """The G > 1 logic of akp_merkle_build_sharded_* (leaf ranges per slot, one host thread per slot, the exchange layout of the
    sub-roots, the redundant top levels, the per-slot slices of the global heap array) on a one-GPU box: with the test hook
    AKP_MULTI_TEST_SHARED_DEVICE=1 the G slots all name device 0 and device-to-device copies stand in for the ncclAllGather
    (RCCL itself runs at n_dev = 1 in test_sharded_build_one_process_rccl and over real devices in the driver's scaling run).
    Result == the single-device build, node by node, for Poseidon and for byte-digest trees; phases are reported."""
    monkeypatch.setenv("AKP_MULTI_TEST_SHARED_DEVICE", "1")
    mg = cpa.MultiGpu([0] * G)
    monkeypatch.delenv("AKP_MULTI_TEST_SHARED_DEVICE")
    assert mg.size == G
    c = cpa.get_default_poseidon_parameters(2, False)
    for n, k in ((1 << 12, 1), (2 * G, 2), (1 << 16, 1)):
        leaves = rand_fr_array(n * k, 0xA5A50003 + n).reshape(n, k, 4)
        ln, nl, root = mg.build_sharded(cpa.PoseidonFieldConfig, c, c, leaves)
        ref = cpa.MerkleTree.new(cpa.PoseidonFieldConfig, c, c, leaves)
        assert np.array_equal(ln, ref.leaf_nodes) and np.array_equal(nl, ref.non_leaf_nodes) and np.array_equal(root, ref.root()), (G, n)
        _, _, root_only = mg.build_sharded(cpa.PoseidonFieldConfig, c, c, leaves, want_nodes=False)
        assert np.array_equal(root_only, root)
    ph = mg.last_phases()
    assert ph["copy_in_and_subtree_ms"] > 0 and ph["top_levels_ms"] > 0 and ph["whole_call_ms"] > ph["copy_in_and_subtree_ms"]
    from crypto_primitives_amd.crh import bowe_hopwood, pedersen
    B = bowe_hopwood.Parameters(gens_array(jj.bowe_hopwood_generators(0xA5A50005, 63, 9)))
    P = pedersen.Parameters(gens_array(jj.pedersen_generators(0xA5A50004, 4, 256)))
    for cfg, prm in ((cpa.BoweHopwoodByteConfig, B), (cpa.PedersenByteConfig, P)):
        lv = np.frombuffer(ofr.SplitMix64(31 + G).bytes(512 * 32), dtype=np.uint8).reshape(512, 32).copy()
        ln, nl, root = mg.build_sharded(cfg, prm, prm, lv)
        ref = cpa.MerkleTree.new(cfg, prm, prm, lv)
        assert np.array_equal(nl, ref.non_leaf_nodes) and np.array_equal(ln, ref.leaf_nodes) and np.array_equal(root, ref.root()), (G, cfg.__name__)
    with pytest.raises(cpa.AkpError):  # every slot needs at least two leaves
        mg.build_sharded(cpa.PoseidonFieldConfig, c, c, rand_fr_array(G, 1).reshape(G, 1, 4))
    mg.close()
    h = C.c_void_p()  # without the hook a repeated device id is rejected
    assert cpa.lib.akp_multi_create((C.c_int32 * 2)(0, 0), 2, C.byref(h)) == cpa._lib.AKP_ERR_BAD_PARAMS




@needs_hooks
@pytest.mark.parametrize("G", [1, 2, 4, 8])
def test_sharded_resident_tree_poseidon(cpa, G, monkeypatch):
    """akp_multi_tree_*: build (host leaves and device-resident leaves), root, proofs routed to the owning shard with the top
    siblings in front, batched updates followed by the exchange -- node by node against the single-device handle and the
    oracle's tree (merkle_tree/mod.rs:383-396, 536-579, 629-702)"""
    import torch
    monkeypatch.setenv("AKP_MULTI_TEST_SHARED_DEVICE", "1")
    mg = cpa.MultiGpu([0] * G)
    monkeypatch.delenv("AKP_MULTI_TEST_SHARED_DEVICE")
    c = cpa.get_default_poseidon_parameters(2, False)
    ora = cref_poseidon(po.get_default_poseidon_parameters(2, False))
    for n, k in ((1 << 12, 1), (2 * G if G > 1 else 2, 2), (1 << 16, 2)):
        leaves = rand_fr_array(n * k, 0xA5A50100 + n + G).reshape(n, k, 4)
        st = mg.build_tree(cpa.PoseidonFieldConfig, c, c, leaves)
        assert (st.n_leaves, st.n_dev, st.height()) == (n, G, n.bit_length())
        ref = cpa.GpuMerkleTree.new(cpa.PoseidonFieldConfig, c, c, leaves)
        host, rhost = st.to_host(), ref.to_host()
        assert np.array_equal(host.leaf_nodes, rhost.leaf_nodes) and np.array_equal(host.non_leaf_nodes, rhost.non_leaf_nodes), (G, n)
        assert np.array_equal(st.root(), ref.root())
        oln, onl = ora.merkle_build(ora, leaves, k, threads=8)
        assert np.array_equal(host.non_leaf_nodes.reshape(-1, 4), np.asarray(onl).reshape(-1, 4))
        # proofs: every shard boundary, repeated and unsorted indexes
        per = n // G
        idx = sorted({0, 1, n - 1, n // 2, per - 1, per % n, (per + 1) % n, max(n - per, 0), min(n - 1, 777)}) + [1, 0]
        got, want = st.generate_proofs(idx), ref.generate_proofs(idx)
        for a, b in zip(got, want):
            assert a.leaf_index == b.leaf_index and np.array_equal(a.leaf_sibling_hash, b.leaf_sibling_hash)
            assert len(a.auth_path) == len(b.auth_path) == n.bit_length() - 2
            assert all(np.array_equal(x, y) for x, y in zip(a.auth_path, b.auth_path)), (G, n, a.leaf_index)
        assert all(cpa.merkle_tree.verify_paths(cpa.PoseidonFieldConfig, c, c, st.root(), got, [leaves[i] for i in idx]))
        # the same tree from leaves that already sit in device memory (one pointer per slot)
        d = torch.from_numpy(leaves.view(np.int64)).to("cuda:0")
        ptrs = [d.data_ptr() + r * per * k * 32 for r in range(G)]
        st2 = mg.build_tree(cpa.PoseidonFieldConfig, c, c, device_leaf_ptrs=ptrs, n_leaves=n, leaf_len=k)
        assert np.array_equal(st2.root(), ref.root()) and np.array_equal(st2.to_host().non_leaf_nodes, rhost.non_leaf_nodes)
        st2.close()
        # batched update: shards with several, one and no touched leaves; a repeated index keeps its last leaf
        upd = [0, n - 1, n - 1, per - 1] + ([per, per + 1] if G > 1 else [1])
        new = rand_fr_array(len(upd) * k, 99 + n).reshape(len(upd), k, 4)
        st.update_batch(upd, new)
        ref.update_batch(upd, new)
        assert np.array_equal(st.root(), ref.root())
        h2, r2 = st.to_host(), ref.to_host()
        assert np.array_equal(h2.non_leaf_nodes, r2.non_leaf_nodes) and np.array_equal(h2.leaf_nodes, r2.leaf_nodes)
        pr = st.generate_proof(n - 1)  # a proof after the update carries the refreshed top siblings
        last = {int(i): j for j, i in enumerate(upd)}  # a repeated index keeps its LAST leaf
        assert cpa.merkle_tree.verify_paths(cpa.PoseidonFieldConfig, c, c, st.root(), [pr], [new[last[n - 1]]])[0]
        with pytest.raises(cpa.AkpError):
            st.update_batch([n], new[:1])
        assert np.array_equal(st.root(), ref.root())  # untouched by the rejected call
        # check_update (:707-725): a wrong asserted root leaves the sharded tree as it was, the right one commits the update
        before = st.to_host().non_leaf_nodes.copy()
        cand = rand_fr_array(k, 5 + n).reshape(k, 4)
        assert not st.check_update(per - 1, cand, st.root())
        assert np.array_equal(st.to_host().non_leaf_nodes, before) and np.array_equal(st.root(), ref.root())
        ref.update_batch([per - 1], cand[None])
        assert st.check_update(per - 1, cand, ref.root()) and np.array_equal(st.root(), ref.root())
        assert np.array_equal(st.to_host().non_leaf_nodes, ref.to_host().non_leaf_nodes)
        # generate_multi_proof over the shards == the single-device handle's, and it verifies
        want_mp, got_mp = ref.generate_multi_proof(idx), st.generate_multi_proof(idx)
        assert got_mp.leaf_indexes == want_mp.leaf_indexes and got_mp.auth_paths_prefix_lenghts == want_mp.auth_paths_prefix_lenghts
        assert all(np.array_equal(a, b) for sa, sb in zip(got_mp.auth_paths_suffixes, want_mp.auth_paths_suffixes) for a, b in zip(sa, sb))
        # new_with_leaf_digest over the shards: the same inner nodes from the leaf digests alone
        st3 = mg.build_tree_with_leaf_digest(cpa.PoseidonFieldConfig, c, c, ref.to_host().leaf_nodes)
        assert np.array_equal(st3.root(), ref.root()) and np.array_equal(st3.to_host().non_leaf_nodes, ref.to_host().non_leaf_nodes)
        st3.close()
        st.close()
        ref.close()
    ph = mg.last_phases()
    assert ph["copy_in_and_subtree_ms"] > 0 and ph["whole_call_ms"] >= ph["copy_in_and_subtree_ms"]
    mg.close()


@needs_hooks
@pytest.mark.parametrize("G", [2, 8])
def test_sharded_resident_tree_byte_digests(cpa, G, monkeypatch):
    """the same for Bowe-Hopwood (1 Fr per digest) and Pedersen (2 Fr per digest) trees over byte leaves"""
    from crypto_primitives_amd.crh import bowe_hopwood, pedersen
    monkeypatch.setenv("AKP_MULTI_TEST_SHARED_DEVICE", "1")
    mg = cpa.MultiGpu([0] * G)
    monkeypatch.delenv("AKP_MULTI_TEST_SHARED_DEVICE")
    B = bowe_hopwood.Parameters(gens_array(jj.bowe_hopwood_generators(0xA5A50005, 63, 9)))
    P = pedersen.Parameters(gens_array(jj.pedersen_generators(0xA5A50004, 4, 256)))
    n = 256
    for cfg, prm in ((cpa.BoweHopwoodByteConfig, B), (cpa.PedersenByteConfig, P)):
        lv = np.frombuffer(ofr.SplitMix64(41 + G).bytes(n * 32), dtype=np.uint8).reshape(n, 32).copy()
        st = mg.build_tree(cfg, prm, prm, lv)
        ref = cpa.GpuMerkleTree.new(cfg, prm, prm, lv)
        assert np.array_equal(st.to_host().non_leaf_nodes, ref.to_host().non_leaf_nodes) and np.array_equal(st.root(), ref.root()), (G, cfg.__name__)
        idx = [0, n // G - 1, n // G, n - 1, 5]
        for a, b in zip(st.generate_proofs(idx), ref.generate_proofs(idx)):
            assert np.array_equal(a.leaf_sibling_hash, b.leaf_sibling_hash) and all(np.array_equal(x, y) for x, y in zip(a.auth_path, b.auth_path))
        new = np.frombuffer(ofr.SplitMix64(7).bytes(3 * 32), dtype=np.uint8).reshape(3, 32).copy()
        st.update_batch([n - 1, 3, n // G], new)
        ref.update_batch([n - 1, 3, n // G], new)
        assert np.array_equal(st.root(), ref.root()) and np.array_equal(st.to_host().non_leaf_nodes, ref.to_host().non_leaf_nodes)
        # round 5: the G slots (contexts of their own, all on device 0) and the single-device tree hash with ONE table
        infos = [h.table_info() for h in prm._handles.values()]
        assert len(infos) >= G + 1 and len({i["table_id"] for i in infos}) == 1 and infos[0]["handles_attached"] == len(infos), infos
        # leaves of DIFFERENT lengths over the slots (akp_multi_tree_build_te_ragged: every slot gets its slice of the offsets)
        lens = np.random.default_rng(G).integers(0, 65, size=n)
        lens[:4] = (0, 1, 64, 3)
        rag = [bytes(lv.reshape(-1)[int(a):int(a) + int(L)]) for a, L in zip(np.arange(n) * 7, lens)]
        st2 = mg.build_tree(cfg, prm, prm, rag)
        ref2 = cpa.GpuMerkleTree.new(cfg, prm, prm, rag)
        assert np.array_equal(st2.root(), ref2.root()) and np.array_equal(st2.to_host().leaf_nodes, ref2.to_host().leaf_nodes)
        assert not np.array_equal(st2.root(), st.root())
        pr = st2.generate_proofs([1, n - 1])
        assert all(cpa.merkle_tree.verify_paths(cfg, prm, prm, st2.root(), pr, [rag[1], rag[n - 1]]))
        st.close()
        ref.close()
        st2.close()
        ref2.close()
    mg.close()


@needs_hooks
def test_background_built_wide_tables_equal_the_per_entry_definition(cpa, monkeypatch):
    """round 6: a budget-chosen wide table is built by the library's thread with FEW workgroups that walk the tiles (grid-stride form of
    te_build_combine_kernel), and a longer message later extends it to the complete table: AKP_TE_TABLE_CHECK (test build) recomputes
    a sample of the entries of each build by the per-entry definition -- a build that differs fails, the upgrade is marked failed and
    the handle would stay on the cache-sized table (state 3)"""
    from crypto_primitives_amd._lib import TABLE_BUDGET_DEVICE
    from crypto_primitives_amd.crh import pedersen, bowe_hopwood
    ctx = cpa.default_context(0)
    ctx.set_table_budget(TABLE_BUDGET_DEVICE)
    try:
        if ctx.table_budget() < 71 << 30:
            pytest.skip("needs an idle 288 GB device")
        monkeypatch.setenv("AKP_TE_TABLE_CHECK", "4099")
        P = pedersen.Parameters(gens_array(jj.pedersen_generators(0x5252, 4, 256)))
        B = bowe_hopwood.Parameters(gens_array(jj.bowe_hopwood_generators(0x5353, 63, 9)))
        hp, hb = P.handle(ctx), B.handle(ctx)
    finally:
        ctx.set_table_budget(0)
    m = np.frombuffer(ofr.SplitMix64(11).bytes(300 * 128), dtype=np.uint8).reshape(300, 128).copy()
    d0 = pedersen.CRH.evaluate_batch(P, m)              # on the cache-sized table; asks for the wide one
    assert hp.wait_for_wide_table(128) is not None
    ti = hp.table_info()
    assert ti["last_build"]["upgrade_state"] == 2 and ti["last_build"]["in_background"] == 1 and ti["last_build"]["note"] == "", ti
    assert np.array_equal(pedersen.CRH.evaluate_batch(P, m), d0) and hp.info(128)["digit_bits_or_group"] == 24
    b0 = bowe_hopwood.CRH.evaluate_batch(B, m[:, :32])  # a prefix of the group table in the background ...
    assert hb.wait_for_wide_table(32) is not None
    assert hb.table_info()["last_build"]["upgrade_state"] == 2 and np.array_equal(bowe_hopwood.CRH.evaluate_batch(B, m[:, :32]), b0)
    b1 = bowe_hopwood.CRH.evaluate_batch(B, m[:, :100])  # ... then a longer message: on the cache-sized table while the wide one is extended
    assert hb.wait_for_wide_table(100) is not None
    ti = hb.table_info()
    assert ti["wide_builds"] == 2 and ti["last_build"]["units_to"] == ti["last_build"]["units_total"] == 70 and ti["last_build"]["upgrade_state"] == 2, ti
    assert np.array_equal(bowe_hopwood.CRH.evaluate_batch(B, m[:, :100]), b1) and hb.info(100)["digit_bits_or_group"] == 8


@needs_hooks
def test_build_arms_plain_pedersen_table_and_walked_zero_tail(cpa, monkeypatch):
    """the two arms kept for A/B in the test build agree bit for bit with the shipped paths (and the plain table reproduces the
    upstream Jubjub known answer, as the signed-subset table does in tests/test_gpu_curves.py)"""
    import json
    from crypto_primitives_amd import params as cparams, field
    from crypto_primitives_amd.crh import pedersen, bowe_hopwood
    from oracle import pedersen as opd, bowe_hopwood as obh
    from helpers import ints
    g = jj.pedersen_generators(0x51, 5, 13)
    m = np.frombuffer(ofr.SplitMix64(3).bytes(40 * 8), dtype=np.uint8).reshape(40, 8).copy()
    ref = pedersen.CRH.evaluate_batch(pedersen.Parameters(gens_array(g)), m)  # signed-subset table
    assert tuple(ints(ref[7])) == opd.evaluate(g, 5, 13, bytes(m[7]))
    monkeypatch.setenv("AKP_PEDERSEN_PLAIN", "1")
    for D in (13, 5, 1):
        monkeypatch.setenv("AKP_PEDERSEN_DIGIT_BITS", str(D))
        P = pedersen.Parameters(gens_array(g))
        assert not P.handle().info(8)["signed_subset"]
        assert np.array_equal(pedersen.CRH.evaluate_batch(P, m), ref), ("plain", D)
    monkeypatch.delenv("AKP_PEDERSEN_DIGIT_BITS")
    k = json.load(open(os.path.join(ROOT, "tests", "golden", "jubjub_upstream_kat.json")))
    k = {kk: (tuple(int(x) for x in v) if isinstance(v, list) else int(v)) for kk, v in k.items() if kk in ("f1", "f2", "g", "f1f2g")}
    scalar = (k["f1"] * k["f2"]) % jj.SUBGROUP_ORDER
    pts, cur = [], (k["g"][0], k["g"][1], 1)
    for _ in range(256):
        pts.append(cparams._affine(cur))
        cur = cparams._padd(cur, cur)
    gens = field.fr([c for pt in pts for c in pt]).reshape(1, 256, 2, 4)
    msg = np.frombuffer(scalar.to_bytes(32, "little"), dtype=np.uint8)
    P = pedersen.Parameters(gens)
    assert tuple(ints(pedersen.CRH.evaluate(P, bytes(msg)))) == k["f1f2g"]
    got = pedersen.CRH.evaluate_batch(P, np.tile(msg, (20000, 1)))
    assert tuple(ints(got[0])) == k["f1f2g"] and tuple(ints(got[19999])) == k["f1f2g"]
    monkeypatch.delenv("AKP_PEDERSEN_PLAIN")
    # Bowe-Hopwood compress: the constant of the zero-padded tail against walking the padding
    W, N = 63, 9
    gb = jj.bowe_hopwood_generators(0x52, W, N)
    B = bowe_hopwood.Parameters(gens_array(gb))
    for n in (5, 20000):
        l = field.random_fr(n, seed=W + n).reshape(n, 1, 4)
        r = field.random_fr(n, seed=N + n).reshape(n, 1, 4)
        got = bowe_hopwood.TwoToOneCRH.compress_batch(B, l, r)
        li, ri = field.to_ints(l[3])[0], field.to_ints(r[3])[0]
        assert ints(got[3])[0] == obh.two_to_one_compress(gb, W, N, li, ri)
        monkeypatch.setenv("AKP_BH_ZERO_TAIL", "0")
        assert np.array_equal(bowe_hopwood.TwoToOneCRH.compress_batch(B, l, r), got)
        monkeypatch.delenv("AKP_BH_ZERO_TAIL")


@needs_hooks
def test_config5_shape_sharded_resident_tree_2pow26_over_8_slots(cpa, monkeypatch):
    """BASELINE configs[4] in the shape the north star names -- a Bowe-Hopwood 63x9 tree of 2^26 x 32-byte leaves over EIGHT device
    slots -- as a resident object (akp_multi_tree_build_te_dev: every slot's 2^23 leaves already in device memory): built, asked for
    its root and for proofs, updated, WITHOUT moving its 4 GiB of nodes off the device.  Size-independent checks: the input repeats
    one 2^20-leaf block, so every slot's sub-root equals the root of the 2^23-leaf tree of eight such blocks (built on its own
    through the single-device handle), the proofs of leaves in different slots verify against the sharded root through the
    product's Path::verify AND the oracle's leaf hash, and an update changes exactly the path of its leaf."""
    import torch
    from crypto_primitives_amd import params, field
    from crypto_primitives_amd.crh import bowe_hopwood
    from crypto_primitives_amd._lib import lib, check
    from oracle import cref
    dev = torch.device("cuda", 0)
    if torch.cuda.get_device_properties(dev).total_memory < (96 << 30):
        pytest.skip("needs ~30 GiB of device memory (eight slots' scratch on one device)")
    G, blk = 8, 1 << 20
    per, n = 1 << 23, 1 << 26
    monkeypatch.setenv("AKP_MULTI_TEST_SHARED_DEVICE", "1")
    mg = cpa.MultiGpu([0] * G)
    monkeypatch.delenv("AKP_MULTI_TEST_SHARED_DEVICE")
    gens = params.bowe_hopwood_generators(0xA5A50005, 63, 9)
    B = bowe_hopwood.Parameters(gens)
    host = np.random.default_rng(0xA5A50026).integers(0, 256, size=(blk, 32), dtype=np.uint8)
    shard_leaves = torch.from_numpy(host).to(dev).repeat(per // blk, 1)  # one slot's 2^23 leaves; every slot reads the same buffer
    free0 = torch.cuda.mem_get_info(0)[0]
    st = mg.build_tree(cpa.BoweHopwoodByteConfig, B, B, device_leaf_ptrs=[shard_leaves.data_ptr()] * G, n_leaves=n, leaf_len=32)
    ph = mg.last_phases()
    # round 5: the eight slots attach to ONE table (the reference's shared &Parameters); the tree's 4 GiB of nodes and the slots'
    # scratch are what the build takes -- not eight copies of the tables
    infos = [h.table_info() for h in B._handles.values()]
    assert len(infos) == G and len({i["table_id"] for i in infos}) == 1 and infos[0]["handles_attached"] == G and infos[0]["wide_builds"] <= 2, infos
    assert free0 - torch.cuda.mem_get_info(0)[0] < 30 << 30
    assert (st.n_leaves, st.n_dev, st.height()) == (n, G, 27) and ph["whole_call_ms"] > 0
    ref = cpa.GpuMerkleTree.new(cpa.BoweHopwoodByteConfig, B, B, shard_leaves.cpu().numpy())  # the 2^23-leaf tree of one slot
    sub_root = ref.root()
    # top of the sharded tree from eight equal sub-roots, by the oracle
    cur = cref.CurveParams(63, 9, gens)
    lvl = np.repeat(np.asarray(sub_root).reshape(1, 4), G, axis=0)
    for width in (4, 2, 1):
        b2 = np.zeros((width, 70), np.uint8)
        b2[:, :32] = cref.from_mont(np.ascontiguousarray(lvl[0::2])).view(np.uint8).reshape(width, 32)
        b2[:, 32:64] = cref.from_mont(np.ascontiguousarray(lvl[1::2])).view(np.uint8).reshape(width, 32)
        lvl = np.asarray(cur.bh_crh_batch(b2, width, 70, threads=4)).reshape(width, 4)
    assert np.array_equal(st.root(), lvl[0])
    # proofs from four different slots: local part == the single-slot tree's proof, top part = sub-roots / oracle levels; they verify
    idx = [5, per + 5, 3 * per + (blk - 1), n - 1]
    proofs = st.generate_proofs(idx)
    local = ref.generate_proofs([i % per for i in idx])
    for p, q in zip(proofs, local):
        assert len(p.auth_path) == 25 and np.array_equal(p.leaf_sibling_hash, q.leaf_sibling_hash)
        assert all(np.array_equal(a, b) for a, b in zip(p.auth_path[3:], q.auth_path))
    assert all(cpa.merkle_tree.verify_paths(cpa.BoweHopwoodByteConfig, B, B, st.root(), proofs, [bytes(host[i % blk]) for i in idx]))
    leaf_digest = np.asarray(cur.bh_crh_batch(np.ascontiguousarray(host[[5]]), 1, 32, threads=1)).reshape(4)
    assert np.array_equal(ref.to_host().leaf_nodes[5], leaf_digest)
    # an update in slot 6 changes the root, the same update back restores it (nothing else moved)
    root0 = st.root().copy()
    newleaf = bytes(range(32))
    st.update_batch([6 * per + 77], [newleaf])
    assert not np.array_equal(st.root(), root0)
    pr = st.generate_proof(6 * per + 77)
    assert cpa.merkle_tree.verify_paths(cpa.BoweHopwoodByteConfig, B, B, st.root(), [pr], [newleaf])[0]
    st.update_batch([6 * per + 77], [bytes(host[77])])
    assert np.array_equal(st.root(), root0)
    st.close()
    ref.close()
    mg.close()


@needs_hooks
def test_two_part_table_construction_equals_the_per_entry_definition(cpa, monkeypatch):
    """the wide tables are built from two part tables with one addition per entry and a shared inversion
    (te_build_combine_kernel); AKP_TE_TABLE_CHECK=k (test build) recomputes every k-th entry by the per-entry definition
    (te_pedersen_slut_entry / te_bh_lutg_entry: D additions + an inversion of its own) and fails the creation on any
    difference of the canonical values.  Every entry for the narrow shapes, a sample for the wide ones."""
    from crypto_primitives_amd.crh import pedersen, bowe_hopwood
    g = gens_array(jj.pedersen_generators(0x52, 5, 13))        # 65 generators: the last digit is clipped at every width
    gb = gens_array(jj.bowe_hopwood_generators(0x53, 7, 5))    # 35 chunks
    mp, mbh = np.frombuffer(ofr.SplitMix64(9).bytes(8), dtype=np.uint8).reshape(1, 8).copy(), np.frombuffer(ofr.SplitMix64(10).bytes(13), dtype=np.uint8).reshape(1, 13).copy()

    def built(P, m, cls):  # the table is built by the first hash: a message of the maximum length builds (and checks) all of it
        cls.evaluate_batch(P, m)
        return P.handle().info()

    monkeypatch.setenv("AKP_TE_TABLE_CHECK", "1")
    for D in (2, 3, 4, 5, 8, 11, 13, 16):
        i = built(pedersen.Parameters(g, table_shape=D), mp, pedersen.CRH)
        assert i["digit_bits_or_group"] == D and i["table_bytes"] >= (-(-64 // D) << (D - 1)) * 128  # the digits 64 message bits reach
    for G in (2, 3, 4, 5):
        i = built(bowe_hopwood.Parameters(gb, table_shape=G), mbh, bowe_hopwood.CRH)
        assert i["digit_bits_or_group"] == G and i["table_bytes"] >= ((35 // G) << (3 * G - 1)) * 128
    monkeypatch.setenv("AKP_TE_TABLE_CHECK", "4099")
    for D in (20, 24):
        assert built(pedersen.Parameters(g, table_shape=D), mp, pedersen.CRH)["digit_bits_or_group"] == D
    for G in (6, 7, 8):
        assert built(bowe_hopwood.Parameters(gb, table_shape=G), mbh, bowe_hopwood.CRH)["digit_bits_or_group"] == G
