"""GPU tests of the SURVEY.md section 8(f) rows: Pedersen commitment, ark-serialize encodings."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import jubjub as jj, commitment as ocm, fr as ofr, poseidon as po  # noqa: E402
from helpers import ints, gens_array, rand_fr_array  # noqa: E402


@pytest.fixture(scope="module")
def cpa():
    import crypto_primitives_amd as m
    assert m.lib.akp_device_count() >= 1
    return m


def test_pedersen_commitment_vs_oracle(cpa):
    from crypto_primitives_amd.commitment import pedersen as cped
    from crypto_primitives_amd.crh import pedersen

    class W(pedersen.Window):
        WINDOW_SIZE, NUM_WINDOWS = 4, 16  # 64-bit inputs
    P = cped.Commitment.setup(W, seed=9)
    g = jj.pedersen_generators(9, 4, 16, bases=jj.random_bases)
    rg = jj.pedersen_generators(9 ^ 0x5EED, 252, 1, bases=jj.random_bases)[0]  # 252 doubling powers of one base
    assert ints(P.randomness_generator[3]) == list(rg[3])
    rng = ofr.SplitMix64(4)
    msgs = [rng.bytes(8), rng.bytes(8), bytes(8), rng.bytes(5)]
    rs = [rng.fr() % cped.SCALAR_MODULUS, 0, cped.SCALAR_MODULUS - 1, 12345]
    got = cped.Commitment.commit_batch(P, msgs[:3], rs[:3])
    for i in range(3):
        assert tuple(ints(got[i])) == ocm.commit(g, rg, 4, 16, msgs[i], rs[i])
    assert tuple(ints(cped.Commitment.commit(P, msgs[3], rs[3]))) == ocm.commit(g, rg, 4, 16, msgs[3], rs[3])
    # hiding term really is r * base: commit(0-message, r) == r * randomness base
    assert tuple(ints(cped.Commitment.commit(P, bytes(8), 77))) == jj.mul(rg[0], 77)
    with pytest.raises(cpa.IncorrectInputLength):
        cped.Commitment.commit(P, bytes(9), 1)
    # commitment/injective_map/mod.rs:12-45: PedersenCommCompressor with TECompressor = x of the commitment
    from crypto_primitives_amd.commitment import injective_map as cinj
    gx = cinj.PedersenCommCompressor.commit_batch(P, msgs[:3], rs[:3])
    for i in range(3):
        assert ints(gx[i])[0] == ocm.commit(g, rg, 4, 16, msgs[i], rs[i])[0]
    assert ints(cinj.PedersenCommCompressor.commit(P, msgs[3], rs[3]))[0] == ocm.commit(g, rg, 4, 16, msgs[3], rs[3])[0]


def test_serialization_roundtrips(cpa):
    from crypto_primitives_amd import serialize as ser, field
    from crypto_primitives_amd.crh import pedersen
    c = cpa.get_default_poseidon_parameters(2, False)
    b = ser.serialize_poseidon_config(c)
    o = po.get_default_poseidon_parameters(2, False)
    # layout: 3 u64, then Vec<Vec<Fr>> ... ; first ark element sits after 3*8 + 8 + 8 bytes, canonical LE
    assert int.from_bytes(b[0:8], "little") == 8 and int.from_bytes(b[8:16], "little") == 31 and int.from_bytes(b[16:24], "little") == 17
    assert int.from_bytes(b[24:32], "little") == 39 and int.from_bytes(b[32:40], "little") == 3
    assert int.from_bytes(b[40:72], "little") == o.ark[0][0]
    assert len(b) == 24 + 8 + 39 * (8 + 96) + 8 + 3 * (8 + 96) + 16
    c2 = ser.deserialize_poseidon_config(b)
    assert np.array_equal(c2.ark, c.ark) and np.array_equal(c2.mds, c.mds) and (c2.rate, c2.capacity, c2.alpha) == (2, 1, 17)
    leaves = rand_fr_array(16, 3).reshape(16, 1, 4)
    tree = cpa.MerkleTree.new(cpa.PoseidonFieldConfig, c2, c2, leaves)
    p = tree.generate_proof(5)
    pb = ser.serialize_path(p)
    assert len(pb) == 32 + 8 + 3 * 32 + 8 and int.from_bytes(pb[-8:], "little") == 5
    assert ser.deserialize_path(pb, cpa.PoseidonFieldConfig).verify(c, c, tree.root(), leaves[5])
    mp = tree.generate_multi_proof(range(16))
    mp2 = ser.deserialize_multi_path(ser.serialize_multi_path(mp), cpa.PoseidonFieldConfig)
    assert mp2.auth_paths_prefix_lenghts == mp.auth_paths_prefix_lenghts and mp2.verify(c, c, tree.root(), leaves)
    g = jj.pedersen_generators(5, 4, 8)
    P = pedersen.Parameters(gens_array(g))
    pbytes = ser.serialize_te_parameters(P)
    assert len(pbytes) == 8 + 8 * (8 + 4 * 64) and pbytes[16:48] == g[0][0][0].to_bytes(32, "little")
    P2 = ser.deserialize_te_parameters(pbytes, pedersen.Parameters)
    assert np.array_equal(P2.generators, P.generators)
    assert np.array_equal(pedersen.CRH.evaluate(P2, b"abcd"), pedersen.CRH.evaluate(P, b"abcd"))


def test_sponge_bytes_absorb_and_fork(cpa):
    """`Absorb for [u8]` and CryptographicSponge::fork (sponge/mod.rs:145-153) on the device sponge vs the oracle"""
    from crypto_primitives_amd import field
    c = cpa.get_default_poseidon_parameters(2, False)
    o = po.get_default_poseidon_parameters(2, False)
    a, b = cpa.PoseidonSponge(c), po.PoseidonSponge(o)
    data = bytes(range(70))
    a.absorb_bytes(data); b.absorb(po.bytes_to_field_elements(data))
    assert field.to_ints(a.squeeze_native_field_elements(2)) == b.squeeze_native_field_elements(2)
    fa, fb = a.fork(b"domain-1"), po.sponge_fork(b, b"domain-1")
    fa2 = a.fork(b"domain-2")
    assert field.to_ints(fa.squeeze_native_field_elements(3)) == fb.squeeze_native_field_elements(3)
    assert field.to_ints(fa2.squeeze_native_field_elements(1)) != field.to_ints(a.clone().squeeze_native_field_elements(1))
    # the parent is unchanged by fork
    assert field.to_ints(a.squeeze_native_field_elements(1)) == b.squeeze_native_field_elements(1)
    # different byte strings encode differently (sponge/poseidon/tests.rs:242-350 spirit): length prefix matters
    assert not np.array_equal(cpa.PoseidonSponge.bytes_to_field_elements(b"\x00"), cpa.PoseidonSponge.bytes_to_field_elements(b"\x00\x00"))


def test_large_tree_equals_combined_subtrees(cpa):
    """size-independent property at a BASELINE-scale size: a 2^22-leaf tree built in one call equals four 2^20-leaf
    sub-trees combined through the sharded-build code path (single process: same arithmetic as the N-GPU run)."""
    import torch
    from crypto_primitives_amd.distributed import GpuPoseidonBackend, combine_top, global_node_slices
    c = cpa.get_default_poseidon_parameters(2, False)
    n, G = 1 << 22, 4
    leaves = rand_fr_array(n, 0xA5A50003).reshape(n, 1, 4)
    dev = torch.device("cuda", 0)
    be = GpuPoseidonBackend(c, c, leaf_len=1, device=dev)
    d_all = torch.from_numpy(leaves.view(np.int64)).to(dev)
    ln, nl, root = be.build_subtree(d_all)
    torch.cuda.synchronize()
    nl_host = nl.cpu().numpy().view(np.uint64)
    subs = []
    for r in range(G):
        lnr, nlr, rr = be.build_subtree(d_all[r * (n // G):(r + 1) * (n // G)])
        torch.cuda.synchronize()
        subs.append(rr.copy())
        nlr_host = nlr.cpu().numpy().view(np.uint64)
        for (lvl, gstart, cnt, lstart) in global_node_slices(n, r, G)[::7]:
            assert np.array_equal(nlr_host[lstart:lstart + cnt], nl_host[gstart:gstart + cnt]), (r, lvl)
    top = combine_top(be.two_to_one_compress, np.stack(subs))
    assert np.array_equal(top[0], root) and np.array_equal(top, nl_host[: G - 1])
    # the tensor-resident form the N-GPU run uses: gathered sub-roots -> top nodes in one inner-level build
    for g in (2, 4, 8):
        lvl = nl[g - 1: 2 * g - 1].contiguous()  # global level log2(g) = what the all-gather would deliver
        assert np.array_equal(be.combine_top_tensor(lvl).cpu().numpy().view(np.uint64), nl_host[: g - 1]), g
    assert be.combine_top_tensor(nl[0:1].contiguous()).shape[0] == 0


def test_te_backend_subtrees_combine(cpa):
    """Bowe-Hopwood byte-leaf tree (BASELINE config 5 shape): one-shot build == sub-trees + combined top (the N-GPU path)"""
    import torch
    from crypto_primitives_amd import params
    from crypto_primitives_amd.crh import bowe_hopwood
    from crypto_primitives_amd.distributed import GpuTeBackend, combine_top
    B = bowe_hopwood.Parameters(params.bowe_hopwood_generators(0xA5A50005, 63, 9))
    dev = torch.device("cuda", 0)
    be = GpuTeBackend(B, B, device=dev)
    n, G = 1 << 12, 4
    leaves = np.random.default_rng(3).integers(0, 256, size=(n, 32), dtype=np.uint8)
    d = torch.from_numpy(leaves).to(dev)
    ln, nl, root = be.build_subtree(d)
    torch.cuda.synchronize()
    ref = cpa.MerkleTree.new(cpa.BoweHopwoodByteConfig, B, B, leaves)
    assert np.array_equal(nl.cpu().numpy().view(np.uint64), ref.non_leaf_nodes)
    subs = [be.build_subtree(d[r * (n // G):(r + 1) * (n // G)])[2].copy() for r in range(G)]
    top = combine_top(be.two_to_one_compress, np.stack(subs))
    assert np.array_equal(top, ref.non_leaf_nodes[: G - 1]) and np.array_equal(top[0], root)
    for g in (2, 4, 8):
        lvl = nl[g - 1: 2 * g - 1].contiguous()
        assert np.array_equal(be.combine_top_tensor(lvl).cpu().numpy().view(np.uint64), ref.non_leaf_nodes[: g - 1]), g


def test_full_size_tree_2pow24(cpa):
    """BASELINE config 3 size: 2^24 one-element leaves, built on the device.  Size-independent checks: the root equals
    the combination of the two 2^23 sub-tree roots; sampled nodes of every level equal compress(children) (oracle)."""
    import torch
    from crypto_primitives_amd.distributed import GpuPoseidonBackend, combine_top
    c = cpa.get_default_poseidon_parameters(2, False)
    ora = cref_poseidon_fixture()
    n = 1 << 24
    leaves = rand_fr_array(n, 0xA5A50003).reshape(n, 1, 4)
    dev = torch.device("cuda", 0)
    be = GpuPoseidonBackend(c, c, leaf_len=1, device=dev)
    d_all = torch.from_numpy(leaves.view(np.int64)).to(dev)
    ln, nl, root = be.build_subtree(d_all)
    torch.cuda.synchronize()
    r0 = be.build_subtree(d_all[: n // 2])[2].copy()
    r1 = be.build_subtree(d_all[n // 2:])[2].copy()
    assert np.array_equal(combine_top(be.two_to_one_compress, np.stack([r0, r1]))[0], root)
    # round 6: 2^16 inner nodes of the upper levels + 2^16 of the bottom level + 2^16 leaf digests against the oracle
    # (4096 + 2048 + 2048 before); the permutation is cheap on the CPU at this count (~0.2 s)
    import os
    thr = max(1, min(64, os.cpu_count() or 1))
    rng = np.random.default_rng(24)
    idx = np.unique(np.concatenate([np.arange(0, 1024), rng.integers(0, n // 2 - 1, 1 << 16), (1 << np.arange(1, 23)) - 1]))
    idx_t = torch.from_numpy(idx).to(dev)
    nodes = nl[idx_t].cpu().numpy().view(np.uint64)
    lch = nl[2 * idx_t + 1].cpu().numpy().view(np.uint64)
    rch = nl[2 * idx_t + 2].cpu().numpy().view(np.uint64)
    assert len(idx) >= 60000 and np.array_equal(nodes, ora.two_to_one_batch(lch, rch, threads=thr))
    bottom = torch.from_numpy(rng.integers(0, n // 2, 1 << 16)).to(dev)
    bn = nl[(n // 2 - 1) + bottom].cpu().numpy().view(np.uint64)
    bl, br = ln[2 * bottom].cpu().numpy().view(np.uint64), ln[2 * bottom + 1].cpu().numpy().view(np.uint64)
    assert np.array_equal(bn, ora.two_to_one_batch(bl, br, threads=thr))
    li = rng.integers(0, n, 1 << 16)
    assert np.array_equal(ln[torch.from_numpy(li).to(dev)].cpu().numpy().view(np.uint64), ora.crh_batch(leaves[li], 1, threads=thr))


def cref_poseidon_fixture():
    from helpers import cref_poseidon
    return cref_poseidon(po.get_default_poseidon_parameters(2, False))


def test_full_size_pedersen_2pow20(cpa):
    """BASELINE config 4 size: 2^20 messages of 128 bytes, Jubjub 4x256.  Size-independent property: the hash is a
    subset sum over message bits, so H(m) = H(m with the high half zeroed) + H(m with the low half zeroed)
    (checked with the oracle's point addition on samples); plus sampled digests against the C oracle."""
    from crypto_primitives_amd import params
    from crypto_primitives_amd.crh import pedersen
    from oracle import cref
    gens = params.pedersen_generators(0xA5A50004, 4, 256)
    P = pedersen.Parameters(gens)
    n = 1 << 20
    msgs = np.random.default_rng(0xA5A50004).integers(0, 256, size=(n, 128), dtype=np.uint8)
    full = pedersen.CRH.evaluate_batch(P, msgs)
    lo, hi = msgs.copy(), msgs.copy()
    lo[:, 64:] = 0
    hi[:, :64] = 0
    hlo, hhi = pedersen.CRH.evaluate_batch(P, lo), pedersen.CRH.evaluate_batch(P, hi)
    assert np.array_equal(hlo, pedersen.CRH.evaluate_batch(P, np.ascontiguousarray(msgs[:, :64])))  # zero padding == short message
    for i in np.random.default_rng(1).integers(0, n, 48):
        assert tuple(ints(full[i])) == jj.add(tuple(ints(hlo[i])), tuple(ints(hhi[i])))
    C = cref.CurveParams(4, 256, gens)
    idx = np.random.default_rng(2).integers(0, n, 512)
    assert np.array_equal(full[idx], C.pedersen_crh_batch(np.ascontiguousarray(msgs[idx]), len(idx), 128, threads=8))
    assert len(np.unique(full.reshape(n, -1)[:, 0])) == n  # no accidental collisions / duplicated lanes


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_build_real_backends_multi_rank(cpa, world):
    """N > 1 control flow with the REAL GPU backends: `world` ranks share GPU 0 (gloo carries the one all-gather, on
    device tensors), each builds its leaf-range sub-tree with GpuPoseidonBackend / GpuTeBackend, and root + top nodes
    must equal a single-process build (tools/gpu_gloo2.py)."""
    import os, socket, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(root, "tools", "gpu_gloo2.py")],
                       capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    for k in range(world):
        assert "rank %d ok" % k in r.stdout


def test_sponge_sized_and_foreign_field_squeezes(cpa):
    """CryptographicSponge::squeeze_field_elements_with_sizes / squeeze_field_elements::<F2> and
    FieldBasedCryptographicSponge::squeeze_native_field_elements_with_sizes (sponge/mod.rs:57-100,164-179,
    sponge/poseidon/mod.rs:293-322) on the device sponge against the oracle, single sponge and a batch"""
    from crypto_primitives_amd import field
    c = cpa.get_default_poseidon_parameters(2, False)
    o = po.get_default_poseidon_parameters(2, False)
    FULL = cpa.PoseidonSponge.FULL
    sizes = [FULL, 100, 7, FULL, 253]
    a, b = cpa.PoseidonSponge(c), po.PoseidonSponge(o)
    a.absorb(field.fr([5, 6, 7])); b.absorb([5, 6, 7])
    assert field.to_ints(a.squeeze_field_elements_with_sizes(sizes)) == po.squeeze_field_elements_with_sizes(b, sizes)
    assert field.to_ints(a.squeeze_native_field_elements_with_sizes([FULL, FULL])) == po.squeeze_native_field_elements_with_sizes(b, [po.FULL, po.FULL])
    q = 0xe7db4ea6533afa906673b0101343b00a6682093ccc81082d0970e5ed6f72cb7  # Jubjub's scalar field: a real "F2"
    assert a.squeeze_field_elements(3, modulus=q) == po.squeeze_field_elements(b, 3, modulus=q)
    assert a.squeeze_field_elements_with_sizes([64, FULL], modulus=q) == po.squeeze_field_elements_with_sizes(b, [64, po.FULL], modulus=q)
    assert field.to_ints(a.squeeze_field_elements(2)) == b.squeeze_native_field_elements(2)
    with pytest.raises(ValueError):
        a.squeeze_field_elements_with_sizes([256])
    # a batch of three sponges with different inputs
    A = cpa.PoseidonSponge(c, batch=3)
    ins = [[1, 2], [3, 4], [5, 6]]
    A.absorb(np.stack([field.fr(x) for x in ins]))
    got = A.squeeze_field_elements_with_sizes([FULL, 33])
    fq = A.squeeze_field_elements(1, modulus=q)
    for k in range(3):
        B = po.PoseidonSponge(o); B.absorb(ins[k])
        assert field.to_ints(got[k]) == po.squeeze_field_elements_with_sizes(B, [po.FULL, 33])
        assert fq[k] == po.squeeze_field_elements(B, 1, modulus=q)


def test_absorb_derive_equals_manual_on_the_device_sponge(cpa):
    """sponge/absorb.rs:428-472 test_absorb_derive on the GPU sponge: absorbing a derived struct == absorbing its fields one
    by one; forgetting fields changes the output.  Plus the oracle on the same element stream, and `absorb!`."""
    from crypto_primitives_amd import field
    from crypto_primitives_amd.sponge import absorb as ab
    c = cpa.get_default_poseidon_parameters(2, False)
    o = po.get_default_poseidon_parameters(2, False)
    fields = [ab.U8(1), ab.U16(2), ab.U32(3), ab.U64(4), ab.U128(5), ab.Fe(6), ab.Struct(ab.U8(7), ab.U16(8)), ab.Struct(ab.U16(9)),
              ab.Struct(ab.Fe(10))]
    s = cpa.PoseidonSponge(c)
    s.absorb(ab.Struct(*fields))
    out_derived = s.squeeze_bytes(32)
    s = cpa.PoseidonSponge(c)
    for f in fields[:5]:
        s.absorb(f)
    assert s.squeeze_bytes(32) != out_derived  # "we forgot to absorb some fields"
    s = cpa.PoseidonSponge(c)
    s.absorb_all(*fields)
    out_manual = s.squeeze_bytes(32)
    # NOTE the reference asserts equality here because each absorb() of the manual sequence continues the same rate block
    assert out_manual == out_derived
    b = po.PoseidonSponge(o)
    b.absorb(ab.Struct(*fields).to_sponge_field_elements())
    assert b.squeeze_bytes(32) == out_derived
    # byte strings, options and points through the same door; an empty encoding is a no-op (:238-240)
    a1, b1 = cpa.PoseidonSponge(c), po.PoseidonSponge(o)
    for item in (ab.Bytes(bytes(range(70))), ab.Opt(ab.I32(-5)), ab.TEAffine(3, 4), ab.Seq([]), ab.Str("domain")):
        a1.absorb(item)
        if item.to_sponge_field_elements():
            b1.absorb(item.to_sponge_field_elements())
    assert field.to_ints(a1.squeeze_native_field_elements(3)) == b1.squeeze_native_field_elements(3)
