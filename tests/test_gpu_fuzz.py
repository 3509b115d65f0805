"""Randomised GPU parity: random batch sizes (ragged, tiny, non-multiples of the block), random message lengths,
random CRH arities and random window shapes, all compared bit for bit with the C oracle."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SCALE = int(os.environ.get("AKP_FUZZ_SCALE", "1"))  # multiply the iteration counts (one-off soak runs)

from oracle import jubjub as jj, poseidon as po, fr as ofr, cref  # noqa: E402
from helpers import rand_fr_array, gens_array, cref_poseidon  # noqa: E402


@pytest.fixture(scope="module")
def cpa():
    import crypto_primitives_amd as m
    assert m.lib.akp_device_count() >= 1
    return m


def test_fuzz_poseidon_shapes(cpa):
    from crypto_primitives_amd.crh import poseidon as pcrh
    rng = np.random.default_rng(2024)
    cfgs = {}
    for it in range(24 * SCALE):
        rate = int(rng.choice([2, 2, 2, 3, 4, 5, 8]))
        w = bool(rng.integers(0, 2)) and rate in (2, 8)
        if (rate, w) not in cfgs:
            cfgs[(rate, w)] = (cpa.get_default_poseidon_parameters(rate, w), cref_poseidon(po.get_default_poseidon_parameters(rate, w)))
        c, ora = cfgs[(rate, w)]
        t = rate + 1
        # sizes on both sides of the latency-kernel switch (2^15) included
        n = int(rng.choice([1, 2, 3, 63, 64, 65, 255, 256, 257, 1000, 4097, 32768, 32769, 40000]))
        if rng.integers(0, 2):
            st = rand_fr_array(n * t, int(rng.integers(1 << 30))).reshape(n, t, 4)
            got = st.copy()
            cpa._lib.check(cpa.lib.akp_poseidon_permute_batch(c.handle().h, got.ctypes.data, n))
            assert np.array_equal(got, ora.permute_batch(st, threads=8).reshape(n, t, 4)), (rate, w, n)
        else:
            k = int(rng.integers(1, 3 * rate + 2))
            x = rand_fr_array(n * k, int(rng.integers(1 << 30))).reshape(n, k, 4)
            assert np.array_equal(pcrh.CRH.evaluate_batch(c, x), ora.crh_batch(x, k, threads=8)), (rate, w, n, k)


def test_fuzz_te_shapes(cpa):
    from crypto_primitives_amd.crh import pedersen, bowe_hopwood
    rng = np.random.default_rng(77)
    for it in range(8 * SCALE):
        W, N = int(rng.integers(1, 9)), int(rng.integers(1, 40))
        g = jj.pedersen_generators(500 + it, W, N)
        P = pedersen.Parameters(gens_array(g))
        C = cref.CurveParams(W, N, gens_array(g))
        for _ in range(3):
            L = int(rng.integers(0, W * N // 8 + 1))
            n = int(rng.choice([1, 7, 64, 300, 16384, 16385]))  # both sides of the split-kernel switch (2^14)
            m = rng.integers(0, 256, size=(n, max(L, 1)), dtype=np.uint8)[:, :L]
            got = pedersen.CRH.evaluate_batch(P, np.ascontiguousarray(m) if L else [b""] * n)
            assert np.array_equal(got, C.pedersen_crh_batch(np.ascontiguousarray(m), n, L, threads=16)), (W, N, L, n)
    for it in range(6 * SCALE):
        W, N = int(rng.integers(1, 64)), int(rng.integers(1, 6))
        g = jj.bowe_hopwood_generators(600 + it, W, N)
        B = bowe_hopwood.Parameters(gens_array(g))
        C = cref.CurveParams(W, N, gens_array(g))
        for _ in range(3):
            L = int(rng.integers(0, W * N * 3 // 8 + 1))
            n = int(rng.choice([1, 5, 65, 257, 16384, 16390]))
            m = rng.integers(0, 256, size=(n, max(L, 1)), dtype=np.uint8)[:, :L]
            got = bowe_hopwood.CRH.evaluate_batch(B, np.ascontiguousarray(m) if L else [b""] * n)
            assert np.array_equal(got, C.bh_crh_batch(np.ascontiguousarray(m), n, L, threads=16)), (W, N, L, n)


def test_fuzz_merkle_sizes(cpa):
    c = cpa.get_default_poseidon_parameters(2, False)
    ora = cref_poseidon(po.get_default_poseidon_parameters(2, False))
    rng = np.random.default_rng(5)
    for log2n in (1, 2, 3, 5, 9, 13):
        for leaf_len in (1, 2, 5):
            n = 1 << log2n
            leaves = rand_fr_array(n * leaf_len, int(rng.integers(1 << 30))).reshape(n, leaf_len, 4)
            t = cpa.MerkleTree.new(cpa.PoseidonFieldConfig, c, c, leaves)
            ln, nl = ora.merkle_build(ora, leaves, leaf_len, threads=8)
            assert np.array_equal(t.leaf_nodes, ln) and np.array_equal(t.non_leaf_nodes, nl), (log2n, leaf_len)
