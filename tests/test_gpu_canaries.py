"""Guard-band ("canary") suite for the C ABI on the GPU -- SURVEY.md section 5 "race detection / sanitizers".

The reference is `#![forbid(unsafe_code)]` (crypto-primitives/src/lib.rs:9): an out-of-bounds write cannot exist there.  The
kernels here do pulled-back unaligned 32-bit message loads (te_kernels.hpp msg_load), strided shared-inversion stores
(te_finalize_lane) and 16-byte vector stores of digests; a parity test does not notice a stray write next to an output.
Every caller-provided buffer of the `_dev` entry points (inputs AND outputs) is therefore placed between two 4 KiB bands
of a known pattern -- byte-message buffers at an ODD address -- the call runs at ragged sizes that cross every kernel
routing threshold (1, 63, 65, 2^14 + 1, 2^15 + 1), and afterwards the bands must be untouched, the inputs unchanged, and the
outputs bit-exact against the oracle.  The host-pointer forms get the same treatment with numpy buffers."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import jubjub as jj, poseidon as po, cref  # noqa: E402
from helpers import rand_fr_array, gens_array, cref_poseidon  # noqa: E402

BAND = 4096
PAT = 0xA7


@pytest.fixture(scope="module")
def cpa():
    import crypto_primitives_amd as m
    assert m.lib.akp_device_count() >= 1
    return m


@pytest.fixture(scope="module")
def torch_dev():
    import torch
    return torch, torch.device("cuda", 0)


class DevGuard:
    """`nbytes` of device memory between two bands; `odd` shifts the payload to an odd address (byte messages only)"""

    def __init__(self, torch_dev, nbytes, init=None, odd=False):
        torch, dev = torch_dev
        self.torch, self.nbytes, self.lead = torch, int(nbytes), BAND + (1 if odd else 0)
        self.t = torch.full((self.lead + self.nbytes + BAND,), PAT, dtype=torch.uint8, device=dev)
        if init is not None:
            raw = np.ascontiguousarray(init).view(np.uint8).reshape(-1)
            assert raw.size == self.nbytes
            if raw.size:
                self.t[self.lead:self.lead + self.nbytes] = torch.from_numpy(raw.copy()).to(dev)
        self.init = None if init is None else np.ascontiguousarray(init).view(np.uint8).reshape(-1).copy()

    @property
    def ptr(self):
        return self.t.data_ptr() + self.lead

    def host(self, dtype=np.uint64):
        return self.t[self.lead:self.lead + self.nbytes].cpu().numpy().view(dtype)

    def check(self, what):
        self.torch.cuda.synchronize()
        lo, hi = self.t[:self.lead].cpu().numpy(), self.t[self.lead + self.nbytes:].cpu().numpy()
        assert (lo == PAT).all(), "%s: bytes BEFORE the buffer were written" % what
        assert (hi == PAT).all(), "%s: bytes AFTER the buffer were written" % what
        if self.init is not None:
            assert np.array_equal(self.host(np.uint8), self.init), "%s: an INPUT buffer was modified" % what


class HostGuard:
    def __init__(self, nbytes, init=None, odd=False):
        self.nbytes, self.lead = int(nbytes), BAND + (1 if odd else 0)
        self.a = np.full(self.lead + self.nbytes + BAND, PAT, dtype=np.uint8)
        self.init = None
        if init is not None:
            raw = np.ascontiguousarray(init).view(np.uint8).reshape(-1)
            assert raw.size == self.nbytes
            self.a[self.lead:self.lead + self.nbytes] = raw
            self.init = raw.copy()

    @property
    def ptr(self):
        return self.a.ctypes.data + self.lead

    def host(self, dtype=np.uint64):
        return self.a[self.lead:self.lead + self.nbytes].copy().view(dtype)

    def check(self, what):
        assert (self.a[:self.lead] == PAT).all(), "%s: bytes BEFORE the host buffer were written" % what
        assert (self.a[self.lead + self.nbytes:] == PAT).all(), "%s: bytes AFTER the host buffer were written" % what
        if self.init is not None:
            assert np.array_equal(self.a[self.lead:self.lead + self.nbytes], self.init), "%s: a host INPUT buffer was modified" % what


RAGGED = (1, 63, 65, (1 << 14) + 1, (1 << 15) + 1)


def _stream(torch_dev):
    torch, dev = torch_dev
    return torch.cuda.current_stream(dev).cuda_stream


@pytest.mark.parametrize("rate", [2, 4])
def test_poseidon_dev_entry_points_stay_inside_their_buffers(cpa, torch_dev, rate):
    """akp_poseidon_{permute,crh,two_to_one}_batch_dev: register kernels (> 2^15), latency kernels (<= 2^15), t = 3 and t = 5"""
    lib, check = cpa.lib, cpa._lib.check
    c = cpa.get_default_poseidon_parameters(rate, False)
    ora = cref_poseidon(po.get_default_poseidon_parameters(rate, False))
    h = c.handle(cpa.default_context(0)).h
    t = rate + 1
    s = _stream(torch_dev)
    for n in RAGGED:
        st = rand_fr_array(n * t, 7000 + n).reshape(n, t, 4)
        g = DevGuard(torch_dev, st.nbytes, st)
        g.init = None  # in place: the payload is the output
        check(lib.akp_poseidon_permute_batch_dev(h, g.ptr, n, s))
        g.check("permute n=%d" % n)
        assert np.array_equal(g.host().reshape(n, t, 4), ora.permute_batch(st, threads=8).reshape(n, t, 4)), n
        for k in (0, 1, 2, 2 * rate + 1):
            x = rand_fr_array(max(n * k, 1), 7100 + n + k)[: n * k].reshape(n, k, 4)
            gi, go = DevGuard(torch_dev, x.nbytes, x), DevGuard(torch_dev, n * 32)
            check(lib.akp_poseidon_crh_batch_dev(h, gi.ptr, n, k, go.ptr, s))
            gi.check("crh in n=%d k=%d" % (n, k))
            go.check("crh out n=%d k=%d" % (n, k))
            exp = ora.crh_batch(x, k, threads=8) if k else np.tile(np.asarray(ora.crh_empty()).reshape(1, 4), (n, 1))
            assert np.array_equal(go.host().reshape(n, 4), np.asarray(exp).reshape(n, 4)), (n, k)
        if rate == 2:
            l, r = rand_fr_array(n, 7200 + n), rand_fr_array(n, 7300 + n)
            gl, gr, go = DevGuard(torch_dev, l.nbytes, l), DevGuard(torch_dev, r.nbytes, r), DevGuard(torch_dev, n * 32)
            check(lib.akp_poseidon_two_to_one_batch_dev(h, gl.ptr, gr.ptr, n, go.ptr, s))
            for gg, nm in ((gl, "left"), (gr, "right"), (go, "out")):
                gg.check("two_to_one %s n=%d" % (nm, n))
            assert np.array_equal(go.host().reshape(n, 4), ora.two_to_one_batch(l, r, threads=8)), n


def _te_sets(cpa):
    from crypto_primitives_amd.crh import pedersen, bowe_hopwood, injective_map
    gp = gens_array(jj.pedersen_generators(0xA5A50004, 4, 256))
    gb = gens_array(jj.bowe_hopwood_generators(0xA5A50005, 63, 9))
    P, B = pedersen.Parameters(gp), bowe_hopwood.Parameters(gb)
    ctx = cpa.default_context(0)
    return [("pedersen", pedersen.te_handle(P, pedersen.CRH, ctx), 2, cref.CurveParams(4, 256, gp), 128),
            ("pedersen_x", pedersen.te_handle(P, injective_map.PedersenCRHCompressor, ctx), 1, cref.CurveParams(4, 256, gp), 128),
            ("bowe_hopwood", pedersen.te_handle(B, bowe_hopwood.CRH, ctx), 1, cref.CurveParams(63, 9, gb), 212)]


def _te_expect(name, cur, msgs, n, L):
    if name == "bowe_hopwood":
        return np.asarray(cur.bh_crh_batch(msgs, n, L, threads=8)).reshape(n, 1, 4)
    e = np.asarray(cur.pedersen_crh_batch(msgs, n, L, threads=8)).reshape(n, 2, 4)
    return e if name == "pedersen" else e[:, :1]


def test_curve_hash_dev_entry_points_stay_inside_their_buffers(cpa, torch_dev):
    """akp_te_crh_batch_dev for Pedersen (x || y), Pedersen + TECompressor (x) and Bowe-Hopwood: message lengths 1, 2, 3 (the
    padded path), 77 (unaligned rows, the pulled-back 32-bit load at the end of every message), the maximum length, at an ODD
    device address; batches on both sides of the split-kernel threshold (2^14)"""
    lib, check = cpa.lib, cpa._lib.check
    s = _stream(torch_dev)
    rng = np.random.default_rng(0xCA7A)
    for name, h, fe, cur, lmax in _te_sets(cpa):
        for n in (1, 63, 65, (1 << 14) + 1):
            for L in (1, 2, 3, 77, lmax):
                if n > (1 << 14) and L not in (3, 77):
                    continue  # the big batch: one padded and one unaligned length are enough
                msgs = rng.integers(0, 256, size=(n, L), dtype=np.uint8)
                gi, go = DevGuard(torch_dev, n * L, msgs, odd=True), DevGuard(torch_dev, n * fe * 32)
                check(lib.akp_te_crh_batch_dev(h.h, gi.ptr, n, L, go.ptr, s))
                gi.check("%s msgs n=%d L=%d" % (name, n, L))
                go.check("%s digests n=%d L=%d" % (name, n, L))
                assert np.array_equal(go.host().reshape(n, fe, 4), _te_expect(name, cur, msgs, n, L)), (name, n, L)


def test_tree_build_and_gather_dev_stay_inside_their_buffers(cpa, torch_dev):
    """akp_merkle_build_{poseidon,te}_dev, akp_merkle_inner_*_dev and akp_merkle_gather_paths_dev with caller buffers: the node
    vectors are exactly n and n - 1 digests, nothing may spill past either end (merkle_tree/mod.rs:441-515 heap order)"""
    torch, dev = torch_dev
    lib, check = cpa.lib, cpa._lib.check
    s = _stream(torch_dev)
    c = cpa.get_default_poseidon_parameters(2, False)
    ora = cref_poseidon(po.get_default_poseidon_parameters(2, False))
    ph = c.handle(cpa.default_context(0)).h
    for n in (2, 64, 1 << 15, 1 << 17):
        leaves = rand_fr_array(n, 7500 + n).reshape(n, 1, 4)
        gl, gln, gnl = DevGuard(torch_dev, leaves.nbytes, leaves), DevGuard(torch_dev, n * 32), DevGuard(torch_dev, (n - 1) * 32)
        check(lib.akp_merkle_build_poseidon_dev(ph, ph, gl.ptr, n, 1, gln.ptr, gnl.ptr, s))
        for gg, nm in ((gl, "leaves"), (gln, "leaf_nodes"), (gnl, "non_leaf_nodes")):
            gg.check("poseidon tree %s n=%d" % (nm, n))
        eln, enl = ora.merkle_build(ora, leaves, 1, threads=8)
        assert np.array_equal(gln.host().reshape(n, 4), eln) and np.array_equal(gnl.host().reshape(n - 1, 4), enl), n
        gnl2 = DevGuard(torch_dev, (n - 1) * 32)
        gln.init = gln.host(np.uint8).copy()  # now an input
        check(lib.akp_merkle_inner_poseidon_dev(ph, gln.ptr, n, gnl2.ptr, s))
        gln.check("inner: leaf_nodes n=%d" % n)
        gnl2.check("inner: non_leaf_nodes n=%d" % n)
        assert np.array_equal(gnl2.host().reshape(n - 1, 4), enl)
        # proofs for ragged index sets, straight from the guarded vectors
        depth = n.bit_length() - 2
        for m in (1, 63, 257):
            idx = np.random.default_rng(n + m).integers(0, n, size=m, dtype=np.uint64)
            gi = DevGuard(torch_dev, idx.nbytes, idx)
            gs, ga = DevGuard(torch_dev, m * 32), DevGuard(torch_dev, m * depth * 32)
            check(lib.akp_merkle_gather_paths_dev(cpa.default_context(0).h, gln.ptr, gnl.ptr, n, 1, gi.ptr, m, gs.ptr, ga.ptr if depth else None, s))
            for gg, nm in ((gi, "indices"), (gs, "siblings"), (ga, "auth paths"), (gln, "leaf_nodes"), (gnl, "non_leaf_nodes")):
                gg.check("gather %s n=%d m=%d" % (nm, n, m))
            sib = gs.host().reshape(m, 4)
            assert np.array_equal(sib, eln[idx.astype(np.int64) ^ 1])
            if depth:
                auth = ga.host().reshape(m, depth, 4)
                node = ((n - 1 + idx.astype(np.int64)) - 1) >> 1  # parent of the leaf in the heap of inner nodes
                for lvl in range(depth - 1, -1, -1):  # auth path is root side first: the last entry is the sibling of the leaf's parent
                    sibn = np.where(node % 2 == 1, node + 1, node - 1)
                    assert np.array_equal(auth[:, lvl], enl[sibn]), (n, m, lvl)
                    node = (node - 1) >> 1
    # byte-digest trees: Bowe-Hopwood (1 Fr per node) and Pedersen (2 Fr per node), leaves at an odd address
    for name, h, fe, cur, _ in _te_sets(cpa):
        kind = 1 if name == "bowe_hopwood" else 0
        for n, L in ((2, 3), (64, 32), (1 << 15, 30)):
            leaves = np.random.default_rng(n + L).integers(0, 256, size=(n, L), dtype=np.uint8)
            gl, gln, gnl = DevGuard(torch_dev, n * L, leaves, odd=True), DevGuard(torch_dev, n * fe * 32), DevGuard(torch_dev, (n - 1) * fe * 32)
            check(lib.akp_merkle_build_te_dev(h.h, h.h, gl.ptr, n, L, gln.ptr, gnl.ptr, s))
            for gg, nm in ((gl, "leaves"), (gln, "leaf_nodes"), (gnl, "non_leaf_nodes")):
                gg.check("%s tree %s n=%d" % (name, nm, n))
            assert np.array_equal(gln.host().reshape(n, fe, 4), _te_expect(name, cur, leaves, n, L)), (name, n)
            if name != "pedersen_x":  # the C oracle builds Pedersen and Bowe-Hopwood trees; the x-only flavour is covered in test_gpu_merkle
                eln, enl = cur.merkle_build(kind, cur, leaves, n, L, threads=8)
                assert np.array_equal(gnl.host().reshape(n - 1, fe * 4), np.asarray(enl).reshape(n - 1, fe * 4)), (name, n)
            gnl2 = DevGuard(torch_dev, (n - 1) * fe * 32)
            check(lib.akp_merkle_inner_te_dev(h.h, gln.ptr, n, gnl2.ptr, s))
            gnl2.check("%s inner n=%d" % (name, n))
            assert np.array_equal(gnl2.host(), gnl.host())


def test_host_pointer_entry_points_stay_inside_their_buffers(cpa):
    """the host-pointer forms (chunked copy pipeline, zero-copy alias for registered memory is not used here): numpy buffers
    between bands, ragged sizes incl. one that spans several pipeline chunks"""
    lib, check = cpa.lib, cpa._lib.check
    c = cpa.get_default_poseidon_parameters(2, False)
    ora = cref_poseidon(po.get_default_poseidon_parameters(2, False))
    ph = c.handle(cpa.default_context(0)).h
    for n in (1, 65, (1 << 18) + 3, (1 << 19) + 1):
        st = rand_fr_array(n * 3, 7600 + n).reshape(n, 3, 4)
        g = HostGuard(st.nbytes, st)
        g.init = None
        check(lib.akp_poseidon_permute_batch(ph, g.ptr, n))
        g.check("host permute n=%d" % n)
        si = np.unique(np.concatenate([np.arange(min(n, 64)), np.arange(max(0, n - 64), n)]))
        assert np.array_equal(g.host().reshape(n, 3, 4)[si], ora.permute_batch(np.ascontiguousarray(st[si]), threads=8).reshape(len(si), 3, 4)), n
        x = rand_fr_array(n * 2, 7700 + n).reshape(n, 2, 4)
        gi, go = HostGuard(x.nbytes, x), HostGuard(n * 32)
        check(lib.akp_poseidon_crh_batch(ph, gi.ptr, n, 2, go.ptr))
        gi.check("host crh in n=%d" % n)
        go.check("host crh out n=%d" % n)
        assert np.array_equal(go.host().reshape(n, 4)[si], ora.crh_batch(np.ascontiguousarray(x[si]), 2, threads=8)), n
    rng = np.random.default_rng(0xCA7B)
    for name, h, fe, cur, lmax in _te_sets(cpa):
        for n, L in ((1, 1), (65, 3), (257, 77), ((1 << 17) + 5, 32)):
            msgs = rng.integers(0, 256, size=(n, L), dtype=np.uint8)
            gi, go = HostGuard(n * L, msgs, odd=True), HostGuard(n * fe * 32)
            check(lib.akp_te_crh_batch(h.h, gi.ptr, n, L, go.ptr))
            gi.check("host %s msgs n=%d L=%d" % (name, n, L))
            go.check("host %s digests n=%d L=%d" % (name, n, L))
            si = np.unique(np.concatenate([np.arange(min(n, 48)), np.arange(max(0, n - 48), n)]))
            assert np.array_equal(go.host().reshape(n, fe, 4)[si], _te_expect(name, cur, np.ascontiguousarray(msgs[si]), len(si), L)), (name, n, L)
    # tree build with all three outputs, and proofs / verification from host vectors
    n = 1 << 12
    leaves = rand_fr_array(n, 7800).reshape(n, 1, 4)
    gl, gln, gnl, gr = HostGuard(leaves.nbytes, leaves), HostGuard(n * 32), HostGuard((n - 1) * 32), HostGuard(32)
    check(lib.akp_merkle_build_poseidon(ph, ph, gl.ptr, n, 1, gln.ptr, gnl.ptr, gr.ptr))
    for gg, nm in ((gl, "leaves"), (gln, "leaf_nodes"), (gnl, "non_leaf_nodes"), (gr, "root")):
        gg.check("host tree %s" % nm)
    eln, enl = ora.merkle_build(ora, leaves, 1, threads=8)
    assert np.array_equal(gnl.host().reshape(n - 1, 4), enl) and np.array_equal(gr.host(), enl[0])
    m, depth = 77, 11  # log2(n) - 1 digests per path
    idx = rng.integers(0, n, size=m, dtype=np.uint64)
    gs, ga = HostGuard(m * 32), HostGuard(m * depth * 32)
    check(lib.akp_merkle_gather_paths(gln.ptr, gnl.ptr, n, 1, idx.ctypes.data, m, gs.ptr, ga.ptr))
    gs.check("host gather siblings")
    ga.check("host gather auth")
    gok = HostGuard(m)
    lv = np.ascontiguousarray(leaves[idx.astype(np.int64)])
    check(lib.akp_merkle_verify_paths_poseidon(ph, ph, gr.ptr, lv.ctypes.data, m, 1, idx.ctypes.data, gs.ptr, ga.ptr, depth, gok.ptr))
    gok.check("verify_paths ok flags")
    assert (gok.host(np.uint8) == 1).all()


def test_resident_tree_update_and_proofs_touch_nothing_else(cpa):
    """akp_merkle_tree_update_batch / gather_paths on the HBM-resident handle: after a batched update the two node vectors equal
    a fresh build of the updated leaves EVERYWHERE (an out-of-range scatter inside the vectors would show), and the host
    outputs of gather_paths stay inside their bands"""
    lib, check = cpa.lib, cpa._lib.check
    c = cpa.get_default_poseidon_parameters(2, False)
    n = 1 << 13
    leaves = rand_fr_array(n, 7900).reshape(n, 1, 4)
    gt = cpa.GpuMerkleTree.new(cpa.PoseidonFieldConfig, c, c, leaves)
    rng = np.random.default_rng(0xCA7C)
    for m in (1, 63, 65, 1000):
        idx = rng.integers(0, n, size=m, dtype=np.uint64)
        new = rand_fr_array(m, 7950 + m).reshape(m, 1, 4)
        gidx, gnew = HostGuard(idx.nbytes, idx), HostGuard(new.nbytes, new)
        check(lib.akp_merkle_tree_update_batch(gt._h, gidx.ptr, gnew.ptr, m, 1))
        gidx.check("update indices m=%d" % m)
        gnew.check("update leaves m=%d" % m)
        for k in range(m):
            leaves[int(idx[k])] = new[k]
        fresh = cpa.MerkleTree.new(cpa.PoseidonFieldConfig, c, c, leaves)
        h = gt.to_host()
        assert np.array_equal(h.leaf_nodes, fresh.leaf_nodes) and np.array_equal(h.non_leaf_nodes, fresh.non_leaf_nodes), m
        gs, ga = HostGuard(m * 32), HostGuard(m * 12 * 32)  # log2(n) - 1 = 12 digests per path
        check(lib.akp_merkle_tree_gather_paths(gt._h, gidx.ptr, m, gs.ptr, ga.ptr))
        gs.check("tree gather siblings m=%d" % m)
        ga.check("tree gather auth m=%d" % m)
    gt.close()


def test_crh_class_decides_the_digest_width(cpa):
    """pedersen::Parameters serve both pedersen::CRH and the TECompressor types in the reference
    (crh/injective_map/mod.rs:45,77); handing Pedersen parameters to the x-only classes must give x of the Pedersen digest in a
    buffer of the x-only size -- never 64 bytes written into 32 -- and mixing Pedersen with Bowe-Hopwood types is a TypeError"""
    from crypto_primitives_amd.crh import pedersen, bowe_hopwood, injective_map
    gp = gens_array(jj.pedersen_generators(0xA5A50004, 4, 256))
    P = pedersen.Parameters(gp)       # plain pedersen::Parameters ...
    X = injective_map.Parameters(gp)  # ... and the x-only flavoured object: either works with either class
    msgs = np.random.default_rng(5).integers(0, 256, size=(300, 100), dtype=np.uint8)
    full = pedersen.CRH.evaluate_batch(P, msgs)
    assert full.shape == (300, 2, 4)
    for params in (P, X):
        x = injective_map.PedersenCRHCompressor.evaluate_batch(params, msgs)
        assert x.shape == (300, 4) and np.array_equal(x, full[:, 0])
        assert np.array_equal(pedersen.CRH.evaluate_batch(params, msgs), full)
    t = cpa.MerkleTree.new(cpa.PedersenXByteConfig, P, P, msgs[:64])
    assert np.array_equal(t.non_leaf_nodes, cpa.MerkleTree.new(cpa.PedersenXByteConfig, X, X, msgs[:64]).non_leaf_nodes)
    gt = cpa.GpuMerkleTree.new(cpa.PedersenXByteConfig, P, P, msgs[:64])
    assert np.array_equal(np.asarray(gt.root()).reshape(-1), np.asarray(t.root()).reshape(-1))
    B = bowe_hopwood.Parameters(gens_array(jj.bowe_hopwood_generators(0xA5A50005, 63, 9)))
    with pytest.raises(TypeError):
        pedersen.CRH.evaluate_batch(B, msgs[:2])
    with pytest.raises(TypeError):
        bowe_hopwood.CRH.evaluate_batch(P, msgs[:2])


def test_sponge_dev_entry_points(cpa, torch_dev):
    """akp_sponge_absorb_dev / akp_sponge_squeeze_dev (sponge/poseidon/mod.rs:236-257, 324-344 on device buffers): a ragged batch
    of sponges, absorb 3 / squeeze 2 / absorb 2 / squeeze 5 (crosses rate blocks in both modes), guard bands around every
    buffer, digests equal to the host-pointer form and to the oracle's sponge"""
    lib, check = cpa.lib, cpa._lib.check
    s = _stream(torch_dev)
    for rate in (2, 3):
        c = cpa.get_default_poseidon_parameters(rate, False)
        ora = cref_poseidon(po.get_default_poseidon_parameters(rate, False))
        ph = c.handle(cpa.default_context(0)).h
        for n in (1, 65, 1000):
            a1, a2 = rand_fr_array(n * 3, 8100 + n).reshape(n, 3, 4), rand_fr_array(n * 2, 8200 + n).reshape(n, 2, 4)
            g1, g2 = DevGuard(torch_dev, a1.nbytes, a1), DevGuard(torch_dev, a2.nbytes, a2)
            o1, o2 = DevGuard(torch_dev, n * 2 * 32), DevGuard(torch_dev, n * 5 * 32)
            sp = C.c_void_p()
            check(lib.akp_sponge_create(ph, n, C.byref(sp)))
            check(lib.akp_sponge_absorb_dev(sp, g1.ptr, 3, s))
            check(lib.akp_sponge_squeeze_dev(sp, o1.ptr, 2, s))
            check(lib.akp_sponge_absorb_dev(sp, g2.ptr, 2, s))
            check(lib.akp_sponge_squeeze_dev(sp, o2.ptr, 5, s))
            for gg, nm in ((g1, "absorb 1"), (g2, "absorb 2"), (o1, "squeeze 1"), (o2, "squeeze 2")):
                gg.check("sponge %s n=%d rate=%d" % (nm, n, rate))
            lib.akp_sponge_destroy(sp)
            got1, got2 = o1.host().reshape(n, 2, 4), o2.host().reshape(n, 5, 4)
            # the host-pointer form on a second sponge batch
            sp2 = C.c_void_p()
            check(lib.akp_sponge_create(ph, n, C.byref(sp2)))
            h1, h2 = np.empty((n, 2, 4), np.uint64), np.empty((n, 5, 4), np.uint64)
            check(lib.akp_sponge_absorb(sp2, a1.ctypes.data, 3))
            check(lib.akp_sponge_squeeze(sp2, h1.ctypes.data, 2))
            check(lib.akp_sponge_absorb(sp2, a2.ctypes.data, 2))
            check(lib.akp_sponge_squeeze(sp2, h2.ctypes.data, 5))
            lib.akp_sponge_destroy(sp2)
            assert np.array_equal(got1, h1) and np.array_equal(got2, h2), (rate, n)
            for i in sorted({0, n // 2, n - 1}):
                exp = ora.sponge_script([3, -2, 2, -5], np.concatenate([a1[i], a2[i]]), 7)
                assert np.array_equal(np.concatenate([got1[i], got2[i]]), exp), (rate, n, i)
