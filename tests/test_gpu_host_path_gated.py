"""The pinned host path of the curve hashes as ONE gated launch (round 5, capi_te.hip te_crh_gated): the accumulate kernel covers the
whole batch while its messages arrive by DMA chunk after chunk; workgroups wait on arrival flags, the host thread releases each chunk's
finalize pass and copy-out.  Digests against the oracle and against the pageable call, over batch shapes that exercise partial
chunks / partial workgroups, repeated calls (the epoch that tags one call's flags) and both hash kinds."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import jubjub as jj, cref  # noqa: E402
from helpers import gens_array  # noqa: E402


@pytest.fixture(scope="module")
def cpa():
    import crypto_primitives_amd as m
    assert m.lib.akp_device_count() >= 1
    return m


class _Pinned:
    def __init__(self, cpa, nbytes):
        self.cpa, self.p = cpa, C.c_void_p()
        cpa._lib.check(cpa.lib.akp_host_alloc(nbytes, C.byref(self.p)))
        self.nbytes = nbytes

    def array(self, dtype, shape):
        n = int(np.prod(shape))
        ct = {np.uint8: C.c_uint8, np.uint64: C.c_uint64}[dtype]
        return np.ctypeslib.as_array((ct * n).from_address(self.p.value)).reshape(shape)

    def free(self):
        self.cpa._lib.check(self.cpa.lib.akp_host_free(self.p))


@pytest.mark.parametrize("kind,W,N,L", [("pedersen", 4, 256, 128), ("bh", 63, 9, 64), ("bh", 63, 9, 32), ("pedersen", 4, 256, 40)])
def test_pinned_batches_through_the_gated_launch(cpa, kind, W, N, L):
    from crypto_primitives_amd.crh import pedersen, bowe_hopwood
    if kind == "pedersen":
        g = gens_array(jj.pedersen_generators(0xE5E50001, W, N))
        prm, fe = pedersen.Parameters(g), 2
        ora = lambda m, k: cref.CurveParams(W, N, g).pedersen_crh_batch(m, k, L, threads=8)  # noqa: E731
    else:
        g = gens_array(jj.bowe_hopwood_generators(0xE5E50002, W, N))
        prm, fe = bowe_hopwood.Parameters(g), 1
        ora = lambda m, k: cref.CurveParams(W, N, g).bh_crh_batch(m, k, L, threads=8)  # noqa: E731
    h = prm.handle()
    chunk = 1 << 17
    for rep, n in enumerate((2 * chunk, 3 * chunk + 1000, 2 * chunk + 1, 5 * chunk - 255)):
        msgs = np.random.default_rng(100 * rep + L).integers(0, 256, size=(n, L), dtype=np.uint8)
        want = np.empty((n, 4 * fe), np.uint64)
        cpa._lib.check(cpa.lib.akp_te_crh_batch(h.h, msgs.ctypes.data, n, L, want.ctypes.data))  # pageable: the chunked launches
        pm, po = _Pinned(cpa, msgs.nbytes), _Pinned(cpa, want.nbytes)
        try:
            pm.array(np.uint8, msgs.shape)[:] = msgs
            out = po.array(np.uint64, want.shape)
            for again in range(2):  # twice: the second call's flags carry the next epoch
                out[:] = 0
                cpa._lib.check(cpa.lib.akp_te_crh_batch(h.h, pm.p, n, L, po.p))
                assert np.array_equal(out, want), (kind, L, n, again, np.nonzero((out != want).any(axis=1))[0][:8])
        finally:
            pm.free()
            po.free()
        si = np.unique(np.concatenate([np.arange(40), np.arange(chunk - 20, chunk + 20), np.linspace(0, n - 1, 300).astype(np.int64), np.arange(n - 40, n)]))
        assert np.array_equal(want[si].reshape(len(si), fe, 4), ora(np.ascontiguousarray(msgs[si]), len(si)).reshape(len(si), fe, 4)), (kind, L, n)
