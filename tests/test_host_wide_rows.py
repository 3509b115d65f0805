"""CPU check (host harness) of the exact wide-state cases of tests/test_gpu_poseidon.py::GENERIC_SHAPES: t = 16 rows are
sums of up to six reduced terms, so in the full form an output lane can reach |v| ~ 13p -- beyond the +-4p window of the
narrow canonicalisation (caught on the GPU in round 2; data dependent, hence the GPU test's own seeds and 130 states)."""
import numpy as np
import pytest

from oracle import poseidon as po
from helpers import mont, rand_fr, rand_fr_array, cref_poseidon
from test_host_harness import H, P  # noqa: F401  (fixture + pointer helper)


@pytest.mark.parametrize("shape", [(12, 4, 2, 3, 3), (15, 1, 4, 9, 5), (7, 2, 8, 20, 5)])
def test_wide_shapes_same_data_as_gpu_test(H, shape):  # noqa: F811
    rate, cap, rf, rp, alpha = shape
    t = rate + cap
    mds = [rand_fr(t, 200 + i + t) for i in range(t)]
    ark_ints = rand_fr((rf + rp) * t, 10 + t)
    o = po.PoseidonConfig(rf, rp, alpha, [ark_ints[i * t:(i + 1) * t] for i in range(rf + rp)], mds, rate, cap)
    ora = cref_poseidon(o)
    n = 130
    st = rand_fr_array(n * t, 3).reshape(n, t, 4)
    exp = ora.permute_batch(st, threads=4).reshape(n, t, 4)
    A, M = mont(ark_ints), mont([x for r in mds for x in r])
    for generic in (0, 2):
        S = st.copy()
        H.hh_poseidon_permute(rf, rp, alpha, rate, cap, P(A), P(M), P(S), n, generic)
        assert np.array_equal(S, exp), (shape, generic)
