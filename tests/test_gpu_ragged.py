"""Per-item lengths (round 5): the reference hashes every input with ITS length -- MerkleTree::new maps LeafHash::evaluate over any
iterator of leaves (merkle_tree/mod.rs:411-422), Bowe-Hopwood pads each input to a multiple of 3 bits only (crh/bowe_hopwood/
mod.rs:131-138), Pedersen pads each input with zero bits to the window (crh/pedersen/mod.rs:82-99), poseidon::CRH takes any &[F]
(crh/poseidon/mod.rs:30-40).  The `_ragged` entry points of the C ABI against the oracle, bit for bit: lengths 0 .. max including 0,
1..3 bytes, lengths that end in the middle of a chunk / a group / a digit, and the maximum."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import jubjub as jj, fr as ofr, cref, bowe_hopwood as obh, pedersen as opd, merkle as omk, poseidon as opo  # noqa: E402
from helpers import gens_array, ints, mont, oracle_cfg, cref_poseidon  # noqa: E402


@pytest.fixture(scope="module")
def cpa():
    import crypto_primitives_amd as m
    assert m.lib.akp_device_count() >= 1
    return m


def _ragged_bytes(lengths, seed):
    offs = np.zeros(len(lengths) + 1, np.uint64)
    offs[1:] = np.cumsum(np.asarray(lengths, dtype=np.uint64))
    flat = np.frombuffer(ofr.SplitMix64(seed).bytes(max(int(offs[-1]), 1)), dtype=np.uint8)[: int(offs[-1])].copy()
    return flat, offs


def _by_length(flat, offs, width, fn):
    """oracle digests of ragged items through the C oracle's uniform batches: items grouped by length"""
    lens = np.diff(offs).astype(np.int64)
    out = np.zeros((len(lens),) + width, np.uint64)
    for L in np.unique(lens):
        sel = np.nonzero(lens == L)[0]
        if L:
            idx = offs[sel].astype(np.int64)[:, None] + np.arange(L)[None, :]
            arr = np.ascontiguousarray(flat[idx])
        else:
            arr = np.zeros((len(sel), 0), flat.dtype)
        out[sel] = fn(arr, len(sel), int(L)).reshape((len(sel),) + width)
    return out


def _lengths(n, max_len, seed, edge):
    rng = np.random.default_rng(seed)
    lens = rng.integers(0, max_len + 1, size=n)
    lens[: len(edge)] = edge
    return [int(x) for x in lens]


@pytest.mark.parametrize("shape", [0, 8, 1, 3])
@pytest.mark.parametrize("n", [1500, 30000])
def test_bowe_hopwood_ragged_batch(cpa, shape, n):
    """63x9 window, lengths 0 .. 212: n = 1500 launches in item order, n = 30000 through the device counting sort (longest first);
    table shapes: budget default (groups of 5), 8, 3 and the one-chunk table"""
    from crypto_primitives_amd.crh import bowe_hopwood
    g = gens_array(jj.bowe_hopwood_generators(0xD5D50001, 63, 9))
    ora = cref.CurveParams(63, 9, g)
    B = bowe_hopwood.Parameters(g, table_shape=shape)
    lens = _lengths(n, 212, 17 + n + shape, [0, 1, 2, 3, 4, 212, 211, 32, 64, 70, 0, 212, 5, 7, 8, 15])
    flat, offs = _ragged_bytes(lens, 3 + shape)
    out = np.empty((n, 4), np.uint64)
    cpa._lib.check(cpa.lib.akp_te_crh_batch_ragged(B.handle().h, flat.ctypes.data, offs.ctypes.data, n, out.ctypes.data))
    want = _by_length(flat, offs, (4,), lambda a, k, L: ora.bh_crh_batch(a, k, L, threads=8))
    assert np.array_equal(out, want), np.nonzero((out != want).any(axis=1))[0][:10]
    # the python oracle on the edge lengths (the C oracle is its own restatement: cross-check both at the edges)
    gl = jj.bowe_hopwood_generators(0xD5D50001, 63, 9)
    for i in range(10):
        assert ints(out[i])[0] == obh.evaluate(gl, 63, 9, bytes(flat[int(offs[i]):int(offs[i + 1])])), lens[i]
    # the Python mirror routes a list of unequal byte strings to the same entry point
    items = [bytes(flat[int(offs[i]):int(offs[i + 1])]) for i in range(40)]
    assert np.array_equal(bowe_hopwood.CRH.evaluate_batch(B, items), want[:40])
    # ... and a ragged batch whose items happen to share one length equals the uniform entry point
    uni = _ragged_bytes([32] * 300, 9)
    o2 = np.empty((300, 4), np.uint64)
    cpa._lib.check(cpa.lib.akp_te_crh_batch_ragged(B.handle().h, uni[0].ctypes.data, uni[1].ctypes.data, 300, o2.ctypes.data))
    assert np.array_equal(o2, bowe_hopwood.CRH.evaluate_batch(B, uni[0].reshape(300, 32)))


@pytest.mark.parametrize("n", [700, 25000])
def test_pedersen_ragged_batch_equals_the_zero_padded_batch(cpa, n):
    """4x256 window, lengths 0 .. 128: oracle parity, and the property the reference's padding implies -- the digest of an item is
    the digest of the item zero-padded to the window (which is why a host CAN pad Pedersen inputs and cannot pad Bowe-Hopwood's)"""
    from crypto_primitives_amd.crh import pedersen
    g = gens_array(jj.pedersen_generators(0xD5D50002, 4, 256))
    ora = cref.CurveParams(4, 256, g)
    P = pedersen.Parameters(g)
    lens = _lengths(n, 128, 5 + n, [0, 1, 2, 3, 4, 128, 127, 64, 33, 0, 128])
    flat, offs = _ragged_bytes(lens, 11)
    out = np.empty((n, 2, 4), np.uint64)
    cpa._lib.check(cpa.lib.akp_te_crh_batch_ragged(P.handle().h, flat.ctypes.data, offs.ctypes.data, n, out.ctypes.data))
    want = _by_length(flat, offs, (2, 4), lambda a, k, L: ora.pedersen_crh_batch(a, k, L, threads=8))
    assert np.array_equal(out, want)
    padded = np.zeros((n, 128), np.uint8)
    for i in range(n):
        padded[i, : lens[i]] = flat[int(offs[i]):int(offs[i + 1])]
    assert np.array_equal(out, pedersen.CRH.evaluate_batch(P, padded))
    gl = jj.pedersen_generators(0xD5D50002, 4, 256)
    for i in (0, 1, 2, 3, 5):
        assert tuple(ints(out[i])) == opd.evaluate(gl, 4, 256, bytes(flat[int(offs[i]):int(offs[i + 1])]))
    # the TECompressor flavour (x only) through the same kernel
    TE_PEDERSEN_X = 2
    ox = np.empty((n, 4), np.uint64)
    cpa._lib.check(cpa.lib.akp_te_crh_batch_ragged(P.handle(kind=TE_PEDERSEN_X).h, flat.ctypes.data, offs.ctypes.data, n, ox.ctypes.data))
    assert np.array_equal(ox, want[:, 0, :])


def test_ragged_argument_checks(cpa):
    from crypto_primitives_amd.crh import pedersen, bowe_hopwood
    from crypto_primitives_amd._lib import AKP_ERR_BAD_PARAMS, AKP_ERR_BAD_LENGTH
    P = pedersen.Parameters(gens_array(jj.pedersen_generators(0xD5D50003, 4, 16)))   # 8-byte window
    B = bowe_hopwood.Parameters(gens_array(jj.bowe_hopwood_generators(0xD5D50004, 8, 4)))  # 96 bits = 12 bytes
    out = np.zeros((4, 2, 4), np.uint64)
    flat = np.arange(40, dtype=np.uint8)
    ok = np.array([0, 3, 3, 8, 16], np.uint64)
    assert cpa.lib.akp_te_crh_batch_ragged(P.handle().h, flat.ctypes.data, ok.ctypes.data, 4, out.ctypes.data) == 0
    too_long = np.array([0, 3, 12, 12, 16], np.uint64)  # a 9-byte item: the reference panics on it (pedersen/mod.rs:82-89)
    assert cpa.lib.akp_te_crh_batch_ragged(P.handle().h, flat.ctypes.data, too_long.ctypes.data, 4, out.ctypes.data) == AKP_ERR_BAD_LENGTH
    assert cpa.lib.akp_te_crh_batch_ragged(B.handle().h, flat.ctypes.data, np.array([0, 13], np.uint64).ctypes.data, 1, out.ctypes.data) == AKP_ERR_BAD_LENGTH
    assert cpa.lib.akp_te_crh_batch_ragged(B.handle().h, flat.ctypes.data, np.array([0, 12], np.uint64).ctypes.data, 1, out.ctypes.data) == 0
    back = np.array([0, 5, 4, 8, 9], np.uint64)
    assert cpa.lib.akp_te_crh_batch_ragged(P.handle().h, flat.ctypes.data, back.ctypes.data, 4, out.ctypes.data) == AKP_ERR_BAD_PARAMS
    assert "must not decrease" in cpa.lib.akp_last_error().decode()
    assert cpa.lib.akp_te_crh_batch_ragged(P.handle().h, flat.ctypes.data, None, 4, out.ctypes.data) == AKP_ERR_BAD_PARAMS
    assert cpa.lib.akp_te_crh_batch_ragged(P.handle().h, None, ok.ctypes.data, 0, None) == 0  # an empty batch
    # offsets that do not start at 0 (a slice of a larger array: what the sharded builders pass)
    o2 = np.zeros((2, 2, 4), np.uint64)
    assert cpa.lib.akp_te_crh_batch_ragged(P.handle().h, flat.ctypes.data, ok[2:].ctypes.data, 2, o2.ctypes.data) == 0
    assert np.array_equal(o2, out[2:])
    with pytest.raises(cpa.IncorrectInputLength):
        pedersen.CRH.evaluate_batch(P, [b"12345678", b"123456789"])


@pytest.mark.parametrize("rate", [2, 3, 5])
@pytest.mark.parametrize("n", [900, 20000])
def test_poseidon_ragged_batch(cpa, rate, n):
    """poseidon::CRH over inputs of 0 .. 9 elements: t = 3 runs the per-lane kernel (sorted by permutation count from 4096 items),
    wider sponges are grouped by length on the host; against the C oracle's uniform batches"""
    from crypto_primitives_amd import field
    from crypto_primitives_amd.crh import poseidon as pcrh
    cfg = cpa.get_default_poseidon_parameters(rate, False)
    ora = cref.Poseidon(cfg.full_rounds, cfg.partial_rounds, cfg.alpha, cfg.rate, cfg.capacity, cfg.ark, cfg.mds)
    lens = _lengths(n, 9, 31 + rate + n, [0, 1, 2, 3, 4, 5, 9, 0, 8])
    offs = np.zeros(n + 1, np.uint64)
    offs[1:] = np.cumsum(np.asarray(lens, dtype=np.uint64))
    flat = field.random_fr(int(offs[-1]), seed=77 + rate).reshape(-1, 4)
    out = np.empty((n, 4), np.uint64)
    cpa._lib.check(cpa.lib.akp_poseidon_crh_batch_ragged(cfg.handle().h, flat.ctypes.data, offs.ctypes.data, n, out.ctypes.data))

    def uni(arr, k, L):
        return ora.crh_batch(np.ascontiguousarray(arr).reshape(k, L, 4), L, threads=8) if L else np.tile(ora.crh_empty(), (k, 1))
    lens_a = np.asarray(lens)
    want = np.zeros((n, 4), np.uint64)
    for L in np.unique(lens_a):
        sel = np.nonzero(lens_a == L)[0]
        idx = offs[sel].astype(np.int64)[:, None] + np.arange(L)[None, :]
        want[sel] = uni(flat[idx] if L else None, len(sel), int(L)).reshape(len(sel), 4)
    assert np.array_equal(out, want)
    items = [flat[int(offs[i]):int(offs[i + 1])] for i in range(30)]
    assert np.array_equal(pcrh.CRH.evaluate_batch(cfg, items), want[:30])
    if rate == 2:  # the known answers of SURVEY.md 8(c): CRH([]), CRH([1]), CRH([1,2]), CRH([1,2,3]) in ONE ragged call
        kat = field.fr([1, 1, 2, 1, 2, 3]).reshape(-1, 4)
        ko = np.array([0, 0, 1, 3, 6], np.uint64)
        o4 = np.empty((4, 4), np.uint64)
        cpa._lib.check(cpa.lib.akp_poseidon_crh_batch_ragged(cfg.handle().h, kat.ctypes.data, ko.ctypes.data, 4, o4.ctypes.data))
        assert field.to_ints(o4) == [22095061030825764236545407651195963259093673160236850179062763631622849581041,
                                     26511395754353438153956014645716883342078847001336301618301653090674576206984,
                                     15097507876956563474224915811700428590665428051852057322491674301805944149214,
                                     21824348928045436617315507271384468960714716528043069047525533016219835096313]


def test_ragged_dev_entry_points_on_a_side_stream(cpa):
    """the `_dev` forms: messages, offsets and digests in device memory, enqueued on the caller's stream"""
    import torch
    from crypto_primitives_amd import field
    from crypto_primitives_amd.crh import bowe_hopwood
    from crypto_primitives_amd._lib import AKP_ERR_BAD_LENGTH, AKP_ERR_BAD_PARAMS
    dev = torch.device("cuda", 0)
    g = gens_array(jj.bowe_hopwood_generators(0xD5D50005, 63, 9))
    ora = cref.CurveParams(63, 9, g)
    B = bowe_hopwood.Parameters(g)
    n = 12000
    lens = _lengths(n, 100, 71, [0, 1, 100, 2, 3])
    flat, offs = _ragged_bytes(lens, 13)
    side = torch.cuda.Stream(device=dev)
    d_flat, d_offs = torch.from_numpy(flat).to(dev), torch.from_numpy(offs.view(np.int64)).to(dev)
    d_out = torch.zeros((n, 4), dtype=torch.int64, device=dev)
    torch.cuda.synchronize(dev)
    with torch.cuda.stream(side):
        cpa._lib.check(cpa.lib.akp_te_crh_batch_ragged_dev(B.handle().h, d_flat.data_ptr(), d_offs.data_ptr(), n, 100, d_out.data_ptr(), side.cuda_stream))
    side.synchronize()
    want = _by_length(flat, offs, (4,), lambda a, k, L: ora.bh_crh_batch(a, k, L, threads=8))
    assert np.array_equal(d_out.cpu().numpy().view(np.uint64), want)
    assert cpa.lib.akp_te_crh_batch_ragged_dev(B.handle().h, d_flat.data_ptr(), d_offs.data_ptr(), n, 213, d_out.data_ptr(), side.cuda_stream) == AKP_ERR_BAD_LENGTH
    # Poseidon, t = 3
    cfg = cpa.get_default_poseidon_parameters(2, False)
    po = cref.Poseidon(cfg.full_rounds, cfg.partial_rounds, cfg.alpha, cfg.rate, cfg.capacity, cfg.ark, cfg.mds)
    klens = _lengths(n, 6, 5, [0, 1, 2, 6])
    ko = np.zeros(n + 1, np.uint64)
    ko[1:] = np.cumsum(np.asarray(klens, dtype=np.uint64))
    elems = field.random_fr(int(ko[-1]), seed=3).reshape(-1, 4)
    d_e, d_ko = torch.from_numpy(elems.view(np.int64)).to(dev), torch.from_numpy(ko.view(np.int64)).to(dev)
    with torch.cuda.stream(side):
        cpa._lib.check(cpa.lib.akp_poseidon_crh_batch_ragged_dev(cfg.handle().h, d_e.data_ptr(), d_ko.data_ptr(), n, d_out.data_ptr(), side.cuda_stream))
    side.synchronize()
    got = d_out.cpu().numpy().view(np.uint64)
    ka = np.asarray(klens)
    for L in np.unique(ka):
        sel = np.nonzero(ka == L)[0][:400]
        idx = ko[sel].astype(np.int64)[:, None] + np.arange(L)[None, :]
        exp = po.crh_batch(np.ascontiguousarray(elems[idx]).reshape(len(sel), L, 4), int(L), threads=8) if L else np.tile(po.crh_empty(), (len(sel), 1))
        assert np.array_equal(got[sel], exp.reshape(len(sel), 4)), L
    wide = cpa.get_default_poseidon_parameters(4, False)
    assert cpa.lib.akp_poseidon_crh_batch_ragged_dev(wide.handle().h, d_e.data_ptr(), d_ko.data_ptr(), n, d_out.data_ptr(), side.cuda_stream) == AKP_ERR_BAD_PARAMS


@pytest.mark.parametrize("kind", ["bh", "pedersen"])
def test_merkle_tree_over_byte_leaves_of_different_lengths(cpa, kind):
    """MerkleTree::new where every leaf has its own length (the reference's map over the leaf iterator): node by node against the
    python oracle's tree, proofs of every leaf verify with THEIR leaf, a proof does not verify with the leaf truncated by a byte"""
    from crypto_primitives_amd.crh import bowe_hopwood, pedersen
    from oracle import serialize as oser
    n = 64
    if kind == "bh":
        W, N = 63, 9
        gl = jj.bowe_hopwood_generators(0xD5D50006, W, N)
        params, cfg, max_len = bowe_hopwood.Parameters(gens_array(gl)), cpa.BoweHopwoodByteConfig, 90
        leaf_hash = lambda m: obh.evaluate(gl, W, N, m)  # noqa: E731
        two = lambda l, r: obh.two_to_one_compress(gl, W, N, l, r)  # noqa: E731
    else:
        W, N = 4, 256
        gl = jj.pedersen_generators(0xD5D50007, W, N)
        params, cfg, max_len = pedersen.Parameters(gens_array(gl)), cpa.PedersenByteConfig, 128
        leaf_hash = lambda m: opd.evaluate(gl, W, N, m)  # noqa: E731
        two = lambda l, r: opd.two_to_one_compress(gl, W, N, l, r)  # noqa: E731
    lens = _lengths(n, max_len, 3, [0, 1, 2, 3, max_len, 31, 32, 33])
    flat, offs = _ragged_bytes(lens, 19)
    leaves = [bytes(flat[int(offs[i]):int(offs[i + 1])]) for i in range(n)]
    tree = cpa.GpuMerkleTree.new(cfg, params, params, leaves)
    ot = omk.MerkleTree(leaf_hash, two, two, lambda d: d, leaves=leaves)
    host = tree.to_host()
    flat_d = lambda d: list(d) if isinstance(d, tuple) else [d]  # noqa: E731
    assert [v for d in ot.leaf_nodes for v in flat_d(d)] == ints(host.leaf_nodes)
    assert [v for d in ot.non_leaf_nodes for v in flat_d(d)] == ints(host.non_leaf_nodes)
    assert ints(np.asarray(cpa.MerkleTree.new(cfg, params, params, leaves).root())) == flat_d(ot.root())  # the host-vector form takes the same path
    proofs = tree.generate_proofs(range(n))
    ok = cpa.merkle_tree.verify_paths(cfg, params, params, tree.root(), proofs, leaves)
    assert all(ok)
    assert not proofs[4].verify(params, params, tree.root(), leaves[4][:-1])  # one byte shorter is another leaf (Bowe-Hopwood: another length)
    # an update with a leaf of yet another length, against a tree built over the updated leaves
    leaves[9] = bytes(range(17))
    tree.update(9, leaves[9])
    assert np.array_equal(np.asarray(tree.root()), np.asarray(cpa.GpuMerkleTree.new(cfg, params, params, leaves).root()))
    with pytest.raises(cpa.IncorrectInputLength):
        cpa.GpuMerkleTree.new(cfg, params, params, leaves[:-1] + [bytes(max_len + (1 if kind == "pedersen" else 130))])


def test_poseidon_merkle_tree_over_leaves_of_different_lengths(cpa):
    from crypto_primitives_amd import field
    cfg = cpa.get_default_poseidon_parameters(2, False)
    oc = oracle_cfg(2, False)
    n = 32
    lens = _lengths(n, 5, 23, [0, 1, 2, 3, 5])
    vals = [[(7 * i + j + 1) for j in range(lens[i])] for i in range(n)]
    leaves = [field.fr(v).reshape(-1, 4) if v else np.zeros((0, 4), np.uint64) for v in vals]
    tree = cpa.GpuMerkleTree.new(cpa.PoseidonFieldConfig, cfg, cfg, leaves)
    ot = omk.MerkleTree(lambda v: opo.crh_evaluate(oc, v), lambda l, r: opo.two_to_one_compress(oc, l, r), lambda l, r: opo.two_to_one_compress(oc, l, r),
                        lambda d: d, leaves=vals)
    host = tree.to_host()
    assert ints(host.leaf_nodes) == list(ot.leaf_nodes) and ints(host.non_leaf_nodes) == list(ot.non_leaf_nodes)
    proofs = tree.generate_proofs(range(n))
    assert all(cpa.merkle_tree.verify_paths(cpa.PoseidonFieldConfig, cfg, cfg, tree.root(), proofs, leaves))
    # a wide leaf sponge (rate 4: t = 5) with the t = 3 two-to-one hash: the leaf level goes through the host grouping
    wide = cpa.get_default_poseidon_parameters(4, False)
    ow = oracle_cfg(4, False)
    t2 = cpa.GpuMerkleTree.new(cpa.PoseidonFieldConfig, wide, cfg, leaves)
    ot2 = omk.MerkleTree(lambda v: opo.crh_evaluate(ow, v), lambda l, r: opo.two_to_one_compress(oc, l, r), lambda l, r: opo.two_to_one_compress(oc, l, r),
                         lambda d: d, leaves=vals)
    assert ints(np.asarray(t2.root())) == [ot2.root()]


def test_ragged_batch_at_full_size_sampled(cpa):
    """2^20 Bowe-Hopwood items of 0 .. 64 bytes (the sizes of tree leaves) in one call: a strided sample against the oracle, and the
    digests of all items of one length equal the uniform entry point's on the same bytes"""
    from crypto_primitives_amd.crh import bowe_hopwood
    g = gens_array(jj.bowe_hopwood_generators(0xD5D50008, 63, 9))
    ora = cref.CurveParams(63, 9, g)
    B = bowe_hopwood.Parameters(g)
    n = 1 << 20
    rng = np.random.default_rng(8)
    lens = rng.integers(0, 65, size=n).astype(np.uint64)
    offs = np.zeros(n + 1, np.uint64)
    offs[1:] = np.cumsum(lens)
    flat = rng.integers(0, 256, size=int(offs[-1]), dtype=np.uint8)
    out = np.empty((n, 4), np.uint64)
    cpa._lib.check(cpa.lib.akp_te_crh_batch_ragged(B.handle().h, flat.ctypes.data, offs.ctypes.data, n, out.ctypes.data))
    si = np.unique(np.concatenate([np.arange(64), np.linspace(0, n - 1, 1500).astype(np.int64), np.arange(n - 64, n)]))
    sub_offs = np.zeros(len(si) + 1, np.uint64)
    sub_offs[1:] = np.cumsum(lens[si])
    sub_flat = np.concatenate([flat[int(offs[i]):int(offs[i + 1])] for i in si]) if sub_offs[-1] else np.zeros(0, np.uint8)
    assert np.array_equal(out[si], _by_length(sub_flat, sub_offs, (4,), lambda a, k, L: ora.bh_crh_batch(a, k, L, threads=16)))
    sel = np.nonzero(lens == 32)[0]
    idx = offs[sel].astype(np.int64)[:, None] + np.arange(32)[None, :]
    assert np.array_equal(out[sel], bowe_hopwood.CRH.evaluate_batch(B, np.ascontiguousarray(flat[idx])))


def test_ragged_tree_at_2pow18_leaves_sampled(cpa):
    """a Bowe-Hopwood tree over 2^18 leaves of 0 .. 64 bytes: sampled leaf digests against the oracle, the inner nodes equal the tree
    built from those leaf digests (MerkleTree::new == new_with_leaf_digest of the leaf hashes, merkle_tree/mod.rs:411-422), proofs of
    leaves of every length verify with their own leaf"""
    from crypto_primitives_amd.crh import bowe_hopwood
    g = gens_array(jj.bowe_hopwood_generators(0xD5D50009, 63, 9))
    ora = cref.CurveParams(63, 9, g)
    B = bowe_hopwood.Parameters(g)
    n = 1 << 18
    rng = np.random.default_rng(18)
    lens = rng.integers(0, 65, size=n).astype(np.uint64)
    offs = np.zeros(n + 1, np.uint64)
    offs[1:] = np.cumsum(lens)
    flat = rng.integers(0, 256, size=int(offs[-1]), dtype=np.uint8)
    h = C.c_void_p()
    lh = B.handle()
    cpa._lib.check(cpa.lib.akp_merkle_tree_build_te_ragged(lh.h, lh.h, flat.ctypes.data, offs.ctypes.data, n, C.byref(h)))
    tree = cpa.GpuMerkleTree(cpa.BoweHopwoodByteConfig, B, B, h)
    host = tree.to_host()
    si = np.unique(np.concatenate([np.arange(48), np.linspace(0, n - 1, 700).astype(np.int64), np.arange(n - 48, n)]))
    sub_offs = np.zeros(len(si) + 1, np.uint64)
    sub_offs[1:] = np.cumsum(lens[si])
    sub_flat = np.concatenate([flat[int(offs[i]):int(offs[i + 1])] for i in si])
    assert np.array_equal(host.leaf_nodes[si], _by_length(sub_flat, sub_offs, (4,), lambda a, k, L: ora.bh_crh_batch(a, k, L, threads=16)))
    again = cpa.GpuMerkleTree.new_with_leaf_digest(cpa.BoweHopwoodByteConfig, B, B, host.leaf_nodes)
    assert np.array_equal(again.to_host().non_leaf_nodes, host.non_leaf_nodes) and np.array_equal(np.asarray(again.root()), np.asarray(tree.root()))
    pick = [int(np.nonzero(lens == L)[0][0]) for L in (0, 1, 2, 3, 31, 32, 33, 63, 64)]
    proofs = tree.generate_proofs(pick)
    leaves = [bytes(flat[int(offs[i]):int(offs[i + 1])]) for i in pick]
    assert all(cpa.merkle_tree.verify_paths(cpa.BoweHopwoodByteConfig, B, B, tree.root(), proofs, leaves))


def test_update_batch_with_new_leaves_of_different_lengths(cpa):
    """GpuMerkleTree.update_batch with unequal new leaves (and a repeated index): equal to the reference's sequential update() calls --
    i.e. to a tree built over the final leaves"""
    from crypto_primitives_amd.crh import bowe_hopwood
    g = gens_array(jj.bowe_hopwood_generators(0xD5D5000A, 63, 9))
    B = bowe_hopwood.Parameters(g)
    n = 128
    rng = np.random.default_rng(4)
    leaves = [bytes(rng.integers(0, 256, size=int(L), dtype=np.uint8)) for L in rng.integers(0, 40, size=n)]
    tree = cpa.GpuMerkleTree.new(cpa.BoweHopwoodByteConfig, B, B, leaves)
    idx = [5, 77, 5, 0, 127, 64, 77]
    new = [b"a", b"bb", b"the last write to leaf five wins", b"", b"z" * 33, b"four", b"seventy-seven, second write"]
    tree.update_batch(idx, new)
    for i, leaf in zip(idx, new):
        leaves[i] = leaf
    fresh = cpa.GpuMerkleTree.new(cpa.BoweHopwoodByteConfig, B, B, leaves)
    assert np.array_equal(np.asarray(tree.root()), np.asarray(fresh.root()))
    assert np.array_equal(tree.to_host().non_leaf_nodes, fresh.to_host().non_leaf_nodes) and np.array_equal(tree.to_host().leaf_nodes, fresh.to_host().leaf_nodes)
