"""Handle lifetimes through the raw C ABI: trees and sponges PIN their parameter handles -- destroying the
parameter handle first only defers its release -- and count as handles of their context, so a context destroyed under a live
tree / sponge leaves calls that fail cleanly instead of touching freed memory."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ctx(lib):
    c = C.c_void_p()
    assert lib.akp_ctx_create(0, C.byref(c)) == 0
    return c


def test_tree_and_sponge_outlive_their_parameter_handles():
    import crypto_primitives_amd as cpa
    from crypto_primitives_amd import field, params as cparams
    from crypto_primitives_amd._lib import lib, AKP_ERR_BAD_PARAMS
    from oracle import cref
    ctx = _ctx(lib)
    p = C.c_void_p()
    assert lib.akp_poseidon_default_params(ctx, 2, 0, C.byref(p)) == 0
    n = 64
    leaves = field.random_fr(n, seed=11).reshape(n, 1, 4)
    t = C.c_void_p()
    assert lib.akp_merkle_tree_build_poseidon(p, p, leaves.ctypes.data, n, 1, C.byref(t)) == 0
    sp = C.c_void_p()
    assert lib.akp_sponge_create(p, 3, C.byref(sp)) == 0
    lib.akp_poseidon_params_destroy(p)  # the host drops its parameter object first (an LRU handle cache does exactly this)
    # the tree still updates (leaf hash + two-to-one with the pinned handle) and agrees with the oracle's rebuilt tree
    idx = np.array([5, 40], dtype=np.uint64)
    new = field.random_fr(2, seed=12).reshape(2, 1, 4)
    assert lib.akp_merkle_tree_update_batch(t, idx.ctypes.data, new.ctypes.data, 2, 1) == 0
    root = np.zeros(4, np.uint64)
    assert lib.akp_merkle_tree_root(t, root.ctypes.data) == 0
    cfg = cpa.get_default_poseidon_parameters(2, False)
    ora = cref.Poseidon(cfg.full_rounds, cfg.partial_rounds, cfg.alpha, cfg.rate, cfg.capacity, cfg.ark, cfg.mds)
    lv2 = leaves.copy()
    lv2[5], lv2[40] = new[0], new[1]
    _, nl = ora.merkle_build(ora, lv2, 1, threads=2)
    assert np.array_equal(root, np.asarray(nl).reshape(-1, 4)[0])
    # the sponge still absorbs and squeezes
    el = field.random_fr(3 * 2, seed=13).reshape(3, 2, 4)
    assert lib.akp_sponge_absorb(sp, el.ctypes.data, 2) == 0
    out = np.zeros((3, 1, 4), np.uint64)
    assert lib.akp_sponge_squeeze(sp, out.ctypes.data, 1) == 0
    assert out.any()
    # context destroyed under the live tree / sponge: their calls fail cleanly, destroying them stays valid
    lib.akp_ctx_destroy(ctx)
    assert lib.akp_merkle_tree_root(t, root.ctypes.data) == AKP_ERR_BAD_PARAMS
    assert lib.akp_merkle_tree_update_batch(t, idx.ctypes.data, new.ctypes.data, 2, 1) == AKP_ERR_BAD_PARAMS
    assert lib.akp_sponge_absorb(sp, el.ctypes.data, 2) == AKP_ERR_BAD_PARAMS
    lib.akp_sponge_destroy(sp)
    lib.akp_merkle_tree_destroy(t)  # last handle: the context struct and the deferred parameter handle go with it

    # the same for a byte tree over curve-hash parameters
    ctx = _ctx(lib)
    g = cparams.bowe_hopwood_generators(3, 63, 9)
    tp = C.c_void_p()
    assert lib.akp_te_params_create(ctx, 1, 63, 9, np.ascontiguousarray(g).ctypes.data, C.byref(tp)) == 0
    lv = np.random.default_rng(2).integers(0, 256, size=(32, 32), dtype=np.uint8)
    t = C.c_void_p()
    assert lib.akp_merkle_tree_build_te(tp, tp, lv.ctypes.data, 32, 32, C.byref(t)) == 0
    r0 = np.zeros(4, np.uint64)
    assert lib.akp_merkle_tree_root(t, r0.ctypes.data) == 0
    lib.akp_te_params_destroy(tp)
    one = np.array([7], dtype=np.uint64)
    assert lib.akp_merkle_tree_update_batch(t, one.ctypes.data, lv[3].ctypes.data, 1, 32) == 0
    r1 = np.zeros(4, np.uint64)
    assert lib.akp_merkle_tree_root(t, r1.ctypes.data) == 0 and not np.array_equal(r0, r1)
    assert lib.akp_merkle_tree_update_batch(t, one.ctypes.data, lv[7].ctypes.data, 1, 32) == 0
    assert lib.akp_merkle_tree_root(t, r1.ctypes.data) == 0 and np.array_equal(r0, r1)  # the old leaf back: the old root
    lib.akp_merkle_tree_destroy(t)
    lib.akp_ctx_destroy(ctx)
