"""GPU parity: MerkleTree::new / proofs / update for the three tree configurations vs the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import jubjub as jj, pedersen as opd, bowe_hopwood as obh, poseidon as po, merkle as omk, fr as ofr, cref  # noqa: E402
from helpers import ints, mont, rand_fr_array, gens_array, cref_poseidon  # noqa: E402


@pytest.fixture(scope="module")
def cpa():
    import crypto_primitives_amd as m
    assert m.lib.akp_device_count() >= 1
    return m


def test_poseidon_tree_golden(cpa, derived):
    from crypto_primitives_amd import field
    c = cpa.get_default_poseidon_parameters(2, False)
    d = derived["poseidon_merkle_8"]
    leaves = field.fr(range(1, 9)).reshape(8, 1, 4)
    t = cpa.MerkleTree.new(cpa.PoseidonFieldConfig, c, c, leaves)
    assert [str(x) for x in field.to_ints(t.non_leaf_nodes)] == d["non_leaf"]
    assert [str(x) for x in field.to_ints(t.leaf_nodes)] == d["leaf_nodes"]
    assert str(field.to_ints(t.root())[0]) == d["root"] and t.height() == 4


@pytest.mark.parametrize("log2n,leaf_len", [(1, 1), (2, 3), (7, 3), (12, 1), (16, 1)])
def test_poseidon_tree_vs_oracle(cpa, log2n, leaf_len):
    c = cpa.get_default_poseidon_parameters(2, False)
    ora = cref_poseidon(po.get_default_poseidon_parameters(2, False))
    n = 1 << log2n
    leaves = rand_fr_array(n * leaf_len, 0xA5A50003 + log2n).reshape(n, leaf_len, 4)
    t = cpa.MerkleTree.new(cpa.PoseidonFieldConfig, c, c, leaves)
    ln, nl = ora.merkle_build(ora, leaves, leaf_len, threads=8)
    assert np.array_equal(t.leaf_nodes, ln) and np.array_equal(t.non_leaf_nodes, nl)
    assert t.height() == log2n + 1


def test_poseidon_tree_distinct_leaf_and_inner_params(cpa):
    """leaf hash with rate-3 parameters, inner hash with rate-2 (Config allows different parameter sets)"""
    cl, ci = cpa.get_default_poseidon_parameters(3, False), cpa.get_default_poseidon_parameters(2, False)
    ol, oi = cref_poseidon(po.get_default_poseidon_parameters(3, False)), cref_poseidon(po.get_default_poseidon_parameters(2, False))
    leaves = rand_fr_array(64 * 4, 5).reshape(64, 4, 4)
    t = cpa.MerkleTree.new(cpa.PoseidonFieldConfig, cl, ci, leaves)
    ln, nl = ol.merkle_build(oi, leaves, 4, threads=4)
    assert np.array_equal(t.non_leaf_nodes, nl)


def test_not_power_of_two_rejected(cpa):
    c = cpa.get_default_poseidon_parameters(2, False)
    for n in (0, 1, 3, 12):
        with pytest.raises(cpa.NotPowerOfTwo):  # merkle_tree/mod.rs:430-433 asserts
            cpa.MerkleTree.new(cpa.PoseidonFieldConfig, c, c, rand_fr_array(max(n, 1), 1).reshape(-1, 1, 4)[:n])
    import ctypes as C
    buf = rand_fr_array(3, 1)
    out = np.empty((3, 4), np.uint64)
    assert cpa.lib.akp_merkle_build_poseidon(c.handle().h, c.handle().h, buf.ctypes.data, 3, 1, out.ctypes.data, out.ctypes.data, None) == 5


def test_field_tree_proofs_and_update(cpa):
    """merkle_tree/tests/mod.rs:208-309 (field_mt_tests::good_root_test shape: 128 leaves of 3 elements)."""
    c = cpa.get_default_poseidon_parameters(2, False)
    n = 128
    leaves = rand_fr_array(n * 3, 42).reshape(n, 3, 4)
    tree = cpa.MerkleTree.new(cpa.PoseidonFieldConfig, c, c, leaves)
    root = tree.root()
    paths = [tree.generate_proof(i) for i in range(n)]
    assert all(cpa.merkle_tree.verify_paths(cpa.PoseidonFieldConfig, c, c, root, paths, leaves))
    assert paths[5].verify(c, c, root, leaves[5])
    wrong = root.copy(); wrong[0] ^= np.uint64(1)
    assert not paths[0].verify(c, c, wrong, leaves[0])
    assert not paths[0].verify(c, c, root, leaves[1])
    mp = tree.generate_multi_proof(range(n))
    assert mp.verify(c, c, root, leaves) and not mp.verify(c, c, wrong, leaves)
    mp8 = cpa.MerkleTree.new(cpa.PoseidonFieldConfig, c, c, leaves[:8]).generate_multi_proof(range(8))
    assert mp8.auth_paths_prefix_lenghts == [0, 2, 1, 2, 0, 2, 1, 2]  # merkle_tree/tests/mod.rs:166
    new = rand_fr_array(5 * 3, 43).reshape(5, 3, 4)
    for (i, v) in zip((2, 3, 5, 111, 127), new):
        tree.update(i, v); leaves[i] = v
    rebuilt = cpa.MerkleTree.new(cpa.PoseidonFieldConfig, c, c, leaves)
    assert np.array_equal(tree.non_leaf_nodes, rebuilt.non_leaf_nodes) and np.array_equal(tree.leaf_nodes, rebuilt.leaf_nodes)
    root = tree.root()
    assert all(tree.generate_proof(i).verify(c, c, root, leaves[i]) for i in (0, 2, 3, 111, 127))
    assert not tree.check_update(0, leaves[1], wrong)
    assert tree.check_update(0, leaves[1], cpa.MerkleTree.new(cpa.PoseidonFieldConfig, c, c, np.concatenate([leaves[1:2], leaves[1:]])).root())


def _byte_leaves(n, L, seed):
    return np.frombuffer(ofr.SplitMix64(seed).bytes(n * L), dtype=np.uint8).reshape(n, L).copy()


def test_bowe_hopwood_tree(cpa, derived):
    from crypto_primitives_amd.crh import bowe_hopwood
    from crypto_primitives_amd import field
    g = jj.bowe_hopwood_generators(0xA5A50005, 63, 9)
    P = bowe_hopwood.Parameters(gens_array(g))
    C = cref.CurveParams(63, 9, gens_array(g))
    d = derived["bowe_hopwood_merkle_4"]
    leaves = [bytes.fromhex(x) for x in d["leaves"]]
    t = cpa.MerkleTree.new(cpa.BoweHopwoodByteConfig, P, P, leaves)
    assert str(field.to_ints(t.root())[0]) == d["root"]
    n = 512
    lv = _byte_leaves(n, 32, 0xA5A50005)
    t = cpa.MerkleTree.new(cpa.BoweHopwoodByteConfig, P, P, lv)
    ln, nl = C.merkle_build(1, C, lv, n, 32, threads=8)
    assert np.array_equal(t.leaf_nodes, ln.reshape(n, 4)) and np.array_equal(t.non_leaf_nodes, nl.reshape(n - 1, 4))
    root = t.root()
    for i in (0, 1, 255, 511):
        assert t.generate_proof(i).verify(P, P, root, bytes(lv[i]))
    assert not t.generate_proof(3).verify(P, P, root, bytes(lv[4]))
    t.update(9, bytes(lv[10])); lv[9] = lv[10]
    assert np.array_equal(t.non_leaf_nodes, cpa.MerkleTree.new(cpa.BoweHopwoodByteConfig, P, P, lv).non_leaf_nodes)


def test_pedersen_tree(cpa, derived):
    """bytes_mt_tests (merkle_tree/tests/mod.rs:5-131): Window4x256 on Jubjub, ByteDigestConverter."""
    from crypto_primitives_amd.crh import pedersen
    from crypto_primitives_amd import field
    g = jj.pedersen_generators(0xA5A50004, 4, 256)
    P = pedersen.Parameters(gens_array(g))
    C = cref.CurveParams(4, 256, gens_array(g))
    d = derived["pedersen_merkle_4"]
    t = cpa.MerkleTree.new(cpa.PedersenByteConfig, P, P, [bytes.fromhex(x) for x in d["leaves"]])
    assert [str(v) for v in field.to_ints(t.root())] == d["root"]
    for n in (2, 4, 128):
        lv = _byte_leaves(n, 32, 1000 + n)
        t = cpa.MerkleTree.new(cpa.PedersenByteConfig, P, P, lv)
        ln, nl = C.merkle_build(0, C, lv, n, 32, threads=8)
        assert np.array_equal(t.leaf_nodes, ln) and np.array_equal(t.non_leaf_nodes, nl)
        root = t.root()
        assert all(t.generate_proof(i).verify(P, P, root, bytes(lv[i])) for i in range(0, n, max(1, n // 8)))
        if n >= 4:
            mp = t.generate_multi_proof(range(n))
            assert mp.verify(P, P, root, [bytes(x) for x in lv])


def test_poseidon_tree_2pow20_sampled(cpa):
    """large tree: root + sampled nodes vs the oracle path recomputation; heap-layout property:
    every sampled inner node equals compress(children)."""
    from crypto_primitives_amd.crh import poseidon as pcrh
    c = cpa.get_default_poseidon_parameters(2, False)
    ora = cref_poseidon(po.get_default_poseidon_parameters(2, False))
    n = 1 << 20
    leaves = rand_fr_array(n, 0xA5A50003).reshape(n, 1, 4)
    t = cpa.MerkleTree.new(cpa.PoseidonFieldConfig, c, c, leaves)
    idx = np.unique(np.concatenate([np.arange(0, 4096), np.random.default_rng(1).integers(0, n // 2 - 1, 4096)]))
    l = np.where(2 * idx + 1 < n - 1, 0, 0)
    left_is_inner = (2 * idx + 1) < (n - 1)
    li = idx[left_is_inner]
    exp = ora.two_to_one_batch(t.non_leaf_nodes[2 * li + 1], t.non_leaf_nodes[2 * li + 2], threads=8)
    assert np.array_equal(t.non_leaf_nodes[li], exp)
    bottom = np.arange(n // 2 - 1, n // 2 - 1 + 2048)
    exp = ora.two_to_one_batch(t.leaf_nodes[2 * (bottom - (n // 2 - 1))], t.leaf_nodes[2 * (bottom - (n // 2 - 1)) + 1], threads=8)
    assert np.array_equal(t.non_leaf_nodes[bottom], exp)
    assert np.array_equal(t.leaf_nodes[:2048], ora.crh_batch(leaves[:2048], 1, threads=8))
    assert t.generate_proof(123457).verify(c, c, t.root(), leaves[123457])


def test_batched_proofs_all_configs(cpa):
    """generate_proofs (one gather call) == generate_proof per leaf; batched verify accepts exactly the valid ones"""
    from crypto_primitives_amd.crh import pedersen, bowe_hopwood
    c = cpa.get_default_poseidon_parameters(2, False)
    n = 64
    leaves = rand_fr_array(n * 2, 9).reshape(n, 2, 4)
    tree = cpa.MerkleTree.new(cpa.PoseidonFieldConfig, c, c, leaves)
    proofs = tree.generate_proofs(range(n))
    for i in (0, 1, 17, 63):
        p = tree.generate_proof(i)
        assert np.array_equal(proofs[i].leaf_sibling_hash, p.leaf_sibling_hash) and proofs[i].leaf_index == i
        assert all(np.array_equal(a, b) for a, b in zip(proofs[i].auth_path, p.auth_path)) and len(p.auth_path) == 5
    shuffled = leaves.copy(); shuffled[[3, 40]] = shuffled[[40, 3]]
    ok = cpa.merkle_tree.verify_paths(cpa.PoseidonFieldConfig, c, c, tree.root(), proofs, shuffled)
    assert ok == [i not in (3, 40) for i in range(n)]
    # two-leaf tree: empty auth path
    t2 = cpa.MerkleTree.new(cpa.PoseidonFieldConfig, c, c, leaves[:2])
    assert all(cpa.merkle_tree.verify_paths(cpa.PoseidonFieldConfig, c, c, t2.root(), t2.generate_proofs([0, 1]), leaves[:2]))
    gb = jj.bowe_hopwood_generators(0xA5A50005, 63, 9)
    B = bowe_hopwood.Parameters(gens_array(gb))
    lv = _byte_leaves(16, 32, 3)
    tb = cpa.MerkleTree.new(cpa.BoweHopwoodByteConfig, B, B, lv)
    pb = tb.generate_proofs(range(16))
    bad = lv.copy(); bad[5, 0] ^= 1
    assert cpa.merkle_tree.verify_paths(cpa.BoweHopwoodByteConfig, B, B, tb.root(), pb, [bytes(x) for x in bad]) == [i != 5 for i in range(16)]
    g = jj.pedersen_generators(0xA5A50004, 4, 256)
    P = pedersen.Parameters(gens_array(g))
    tp = cpa.MerkleTree.new(cpa.PedersenByteConfig, P, P, lv[:8])
    assert all(cpa.merkle_tree.verify_paths(cpa.PedersenByteConfig, P, P, tp.root(), tp.generate_proofs(range(8)), [bytes(x) for x in lv[:8]]))


def test_new_with_leaf_digest_and_blank(cpa):
    """MerkleTree::new_with_leaf_digest (:424-523) == the inner levels of MerkleTree::new; blank (:400-408)"""
    from crypto_primitives_amd.crh import pedersen, bowe_hopwood
    c = cpa.get_default_poseidon_parameters(2, False)
    leaves = rand_fr_array(32 * 2, 4).reshape(32, 2, 4)
    t = cpa.MerkleTree.new(cpa.PoseidonFieldConfig, c, c, leaves)
    t2 = cpa.MerkleTree.new_with_leaf_digest(cpa.PoseidonFieldConfig, c, c, t.leaf_nodes)
    assert np.array_equal(t2.non_leaf_nodes, t.non_leaf_nodes) and t2.height() == 6
    b = cpa.MerkleTree.blank(cpa.PoseidonFieldConfig, c, c, 4)
    ora = cref_poseidon(po.get_default_poseidon_parameters(2, False))
    z = np.zeros((8, 4), np.uint64)
    lvl = ora.two_to_one_batch(z[0::2], z[1::2]); lvl = ora.two_to_one_batch(lvl[0::2], lvl[1::2]); lvl = ora.two_to_one_batch(lvl[0::2], lvl[1::2])
    assert np.array_equal(b.root(), lvl[0]) and len(b.leaf_nodes) == 8
    with pytest.raises(cpa.NotPowerOfTwo):
        cpa.MerkleTree.new_with_leaf_digest(cpa.PoseidonFieldConfig, c, c, t.leaf_nodes[:3])
    gb = jj.bowe_hopwood_generators(0xA5A50005, 63, 9)
    B = bowe_hopwood.Parameters(gens_array(gb))
    lv = _byte_leaves(16, 32, 8)
    tb = cpa.MerkleTree.new(cpa.BoweHopwoodByteConfig, B, B, lv)
    assert np.array_equal(cpa.MerkleTree.new_with_leaf_digest(cpa.BoweHopwoodByteConfig, B, B, tb.leaf_nodes).non_leaf_nodes, tb.non_leaf_nodes)
    g = jj.pedersen_generators(0xA5A50004, 4, 256)
    Pp = pedersen.Parameters(gens_array(g))
    tp = cpa.MerkleTree.new(cpa.PedersenByteConfig, Pp, Pp, lv[:4])
    assert np.array_equal(cpa.MerkleTree.new_with_leaf_digest(cpa.PedersenByteConfig, Pp, Pp, tp.leaf_nodes).non_leaf_nodes, tp.non_leaf_nodes)
    bp = cpa.MerkleTree.blank(cpa.PedersenByteConfig, Pp, Pp, 3)  # 4 identity leaves
    ident = jj.serialize_uncompressed(jj.IDENTITY)
    lvl1 = opd.two_to_one_evaluate(g, 4, 256, ident, ident)
    assert tuple(ints(bp.root())) == opd.two_to_one_compress(g, 4, 256, lvl1, lvl1)


def test_update_batch_equals_sequential_updates(cpa):
    """update_batch == update() applied in order (repeated index: last leaf wins) == rebuild, for the field tree and
    for a byte-digest tree (Bowe-Hopwood, ByteDigestConverter)."""
    from crypto_primitives_amd import params
    from crypto_primitives_amd.crh import bowe_hopwood
    c = cpa.get_default_poseidon_parameters(2, False)
    n = 256
    leaves = rand_fr_array(n * 2, 77).reshape(n, 2, 4)
    rng = np.random.default_rng(5)
    idx = np.concatenate([rng.integers(0, n, 40), [0, n - 1, 17, 17, 16]])
    new = rand_fr_array(len(idx) * 2, 78).reshape(len(idx), 2, 4)
    a = cpa.MerkleTree.new(cpa.PoseidonFieldConfig, c, c, leaves)
    b = cpa.MerkleTree.new(cpa.PoseidonFieldConfig, c, c, leaves)
    a.update_batch(idx, new)
    final = leaves.copy()
    for i, v in zip(idx, new):
        b.update(int(i), v)
        final[i] = v
    rebuilt = cpa.MerkleTree.new(cpa.PoseidonFieldConfig, c, c, final)
    for t in (a, b):
        assert np.array_equal(t.leaf_nodes, rebuilt.leaf_nodes) and np.array_equal(t.non_leaf_nodes, rebuilt.non_leaf_nodes)
    a.update_batch([], np.zeros((0, 2, 4), np.uint64))  # no-op
    assert np.array_equal(a.root(), rebuilt.root())
    B = bowe_hopwood.Parameters(params.bowe_hopwood_generators(0xA5A50005, 63, 9))
    bl = _byte_leaves(64, 32, 9)
    t = cpa.MerkleTree.new(cpa.BoweHopwoodByteConfig, B, B, bl)
    bi = np.array([3, 3, 63, 0, 31])
    bn = _byte_leaves(5, 32, 10)
    t.update_batch(bi, bn)
    for i, v in zip(bi, bn):
        bl[i] = v
    rb = cpa.MerkleTree.new(cpa.BoweHopwoodByteConfig, B, B, bl)
    assert np.array_equal(t.non_leaf_nodes, rb.non_leaf_nodes) and np.array_equal(t.leaf_nodes, rb.leaf_nodes)
    # two-leaf tree: only the bottom level exists
    t2 = cpa.MerkleTree.new(cpa.PoseidonFieldConfig, c, c, leaves[:2])
    t2.update_batch([1], new[:1])
    assert np.array_equal(t2.root(), cpa.MerkleTree.new(cpa.PoseidonFieldConfig, c, c, np.stack([leaves[0], new[0]])).root())


def test_largest_batches_2pow26(cpa):
    """the upper end of the BASELINE size range (2^26): a batched permutation over 2^26 states and a Poseidon tree over 2^26
    leaves, both resident in HBM (6 GiB of states; 2 + 2 + 2 GiB of leaves and nodes), checked through size-independent
    properties: the input repeats a 2^20 block 64 times, so every block of the permutation output must equal the first one
    (itself sampled against the oracle), and the 64 nodes of tree level 6 must all equal the root of the 2^20-leaf tree."""
    import torch
    from crypto_primitives_amd import field
    from crypto_primitives_amd._lib import lib, check
    c = cpa.get_default_poseidon_parameters(2, False)
    ora = cref_poseidon(po.get_default_poseidon_parameters(2, False))
    dev = torch.device("cuda", 0)
    if torch.cuda.get_device_properties(dev).total_memory < (40 << 30):
        pytest.skip("needs ~16 GiB of device memory")
    h = c.handle(cpa.default_context(0))
    stream = torch.cuda.current_stream(dev).cuda_stream
    blk, reps = 1 << 20, 64
    n = blk * reps
    host = field.random_fr(blk * 3, seed=0xA5A50026).reshape(blk, 3, 4)
    st = torch.from_numpy(host.view(np.int64)).to(dev).repeat(reps, 1, 1)
    check(lib.akp_poseidon_permute_batch_dev(h.h, st.data_ptr(), n, stream))
    torch.cuda.synchronize()
    first = st[:blk]
    for r in (1, 17, 63):
        assert torch.equal(st[r * blk:(r + 1) * blk], first), r
    si = np.unique(np.concatenate([np.arange(64), np.linspace(0, blk - 1, 193).astype(np.int64)]))
    assert np.array_equal(first.cpu().numpy().view(np.uint64)[si], ora.permute_batch(np.ascontiguousarray(host[si]), threads=8).reshape(len(si), 3, 4))
    del st, first
    leaves_blk = torch.from_numpy(host[:, :1].copy().view(np.int64)).to(dev)
    leaves = leaves_blk.repeat(reps, 1, 1)
    ln = torch.empty((n, 4), dtype=torch.int64, device=dev)
    nl = torch.empty((n - 1, 4), dtype=torch.int64, device=dev)
    check(lib.akp_merkle_build_poseidon_dev(h.h, h.h, leaves.data_ptr(), n, 1, ln.data_ptr(), nl.data_ptr(), stream))
    ln_b = torch.empty((blk, 4), dtype=torch.int64, device=dev)
    nl_b = torch.empty((blk - 1, 4), dtype=torch.int64, device=dev)
    check(lib.akp_merkle_build_poseidon_dev(h.h, h.h, leaves_blk.data_ptr(), blk, 1, ln_b.data_ptr(), nl_b.data_ptr(), stream))
    torch.cuda.synchronize()
    level6 = nl[63:127]  # global level 6 = the roots of the 64 sub-trees of 2^20 leaves
    assert torch.equal(level6, nl_b[0:1].expand(64, 4))
    assert torch.equal(nl[127:127 + 128][0::2], nl_b[1:2].expand(64, 4))  # level 7: left children
    top = nl[:63].cpu().numpy().view(np.uint64)
    lvl = level6.cpu().numpy().view(np.uint64)
    for width in (32, 16, 8, 4, 2, 1):  # the top six levels recomputed by the oracle from level 6
        lvl = ora.two_to_one_batch(np.ascontiguousarray(lvl[0::2]), np.ascontiguousarray(lvl[1::2]), threads=4)
        assert np.array_equal(top[width - 1: 2 * width - 1], lvl), width


def test_pedersen_compressor_tree(cpa):
    """JubJubMerkleTreeParams of merkle_tree/tests/constraints.rs: leaf hash PedersenCRHCompressor<JubJub, TECompressor,
    Window4x256>, two-to-one PedersenTwoToOneCRHCompressor, Fq digests, ByteDigestConverter.  Small tree against the generic
    python oracle (oracle/merkle.py over oracle/pedersen.py), a 512-leaf tree level by level against the C oracle; proofs,
    multi-proof and update through the generic machinery."""
    from crypto_primitives_amd.crh import injective_map as inj
    from crypto_primitives_amd import field
    from oracle import merkle as omk, pedersen as opd
    g = jj.pedersen_generators(0xA5A50004, 4, 256)
    X = inj.Parameters(gens_array(g))
    C = cref.CurveParams(4, 256, gens_array(g))
    lv = _byte_leaves(8, 30, 4321)
    t = cpa.MerkleTree.new(cpa.PedersenXByteConfig, X, X, lv)
    ot = omk.MerkleTree(lambda leaf: opd.compressor_evaluate(g, 4, 256, bytes(leaf)),
                        lambda a, b: opd.compressor_two_to_one_evaluate(g, 4, 256, a, b),
                        lambda a, b: opd.compressor_two_to_one_compress(g, 4, 256, a, b),
                        jj.fq_serialize, leaves=[bytes(x) for x in lv])
    assert [field.to_ints(x)[0] for x in t.leaf_nodes] == list(ot.leaf_nodes)
    assert [field.to_ints(x)[0] for x in t.non_leaf_nodes] == list(ot.non_leaf_nodes)
    n = 512
    lv = _byte_leaves(n, 32, 4322)
    t = cpa.MerkleTree.new(cpa.PedersenXByteConfig, X, X, lv)
    ln = np.asarray(C.pedersen_crh_batch(np.ascontiguousarray(lv), n, 32, threads=8)).reshape(n, 2, 4)[:, 0]
    assert np.array_equal(t.leaf_nodes, ln)
    child = ln
    for width in (256, 128, 64, 32, 16, 8, 4, 2, 1):
        buf = np.zeros((width, 128), np.uint8)
        pairs = cref.from_mont(np.ascontiguousarray(child)).view(np.uint8).reshape(width, 64)
        buf[:, :64] = pairs
        exp = np.asarray(C.pedersen_crh_batch(buf, width, 128, threads=8)).reshape(width, 2, 4)[:, 0]
        assert np.array_equal(t.non_leaf_nodes[width - 1: 2 * width - 1], exp), width
        child = exp
    root = t.root()
    assert all(t.generate_proof(i).verify(X, X, root, bytes(lv[i])) for i in (0, 1, 255, 511))
    assert not t.generate_proof(3).verify(X, X, root, bytes(lv[4]))
    mp = t.generate_multi_proof([0, 1, 2, 3, 100, 101])
    assert mp.verify(X, X, root, [bytes(lv[i]) for i in (0, 1, 2, 3, 100, 101)])
    t.update(9, bytes(lv[10])); lv[9] = lv[10]
    assert np.array_equal(t.non_leaf_nodes, cpa.MerkleTree.new(cpa.PedersenXByteConfig, X, X, lv).non_leaf_nodes)
    # HBM-resident handle with the same configuration
    gt = cpa.GpuMerkleTree.new(cpa.PedersenXByteConfig, X, X, lv)
    assert np.array_equal(np.asarray(gt.root()).reshape(-1), np.asarray(t.root()).reshape(-1))
