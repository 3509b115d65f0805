"""GPU parity: Poseidon permutation / CRH / two-to-one / duplex sponge through the C ABI vs the oracle.
Bit-exact (integer arithmetic)."""
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import poseidon as po, fr as ofr  # noqa: E402
from helpers import mont, ints, rand_fr, rand_fr_array, cref_poseidon  # noqa: E402


@pytest.fixture(scope="module")
def cpa():
    import crypto_primitives_amd as m
    assert m.lib.akp_device_count() >= 1, "no HIP device: the product has no CPU path"
    return m


def _cfg_pair(cpa, rate, weights=False):
    c = cpa.get_default_poseidon_parameters(rate, weights)
    o = po.get_default_poseidon_parameters(rate, weights)
    return c, o, cref_poseidon(o)


def _permute(cpa, cfg, states):
    st = np.ascontiguousarray(states, dtype=np.uint64).copy()
    n = st.size // (4 * cfg.t)
    cpa._lib.check(cpa.lib.akp_poseidon_permute_batch(cfg.handle().h, st.ctypes.data, n))
    return st


def test_permute_kat_vectors(cpa, derived):
    c, o, _ = _cfg_pair(cpa, 2)
    got = _permute(cpa, c, mont([0, 1, 2]))
    assert [str(x) for x in ints(got)] == derived["poseidon_rate2"]["permute_0_1_2"]


@pytest.mark.parametrize("log2n", [0, 6, 8, 14])
def test_permute_batch_rate2_vs_oracle(cpa, log2n):
    c, o, ora = _cfg_pair(cpa, 2)
    n = 1 << log2n
    n += 3 if log2n == 8 else 0  # ragged: not a multiple of the block size
    st = rand_fr_array(n * 3, 0xA5A50002 + log2n).reshape(n, 3, 4)
    got = _permute(cpa, c, st)
    exp = ora.permute_batch(st, threads=8).reshape(n, 3, 4)
    assert np.array_equal(got, exp)


def test_permute_edge_values(cpa):
    c, o, ora = _cfg_pair(cpa, 2)
    P = ofr.P
    vals = [0, 0, 0, P - 1, P - 1, P - 1, 1, P - 1, 0, 2 ** 254, 2 ** 128 - 1, 2 ** 64]
    st = mont(vals).reshape(-1, 3, 4)
    assert np.array_equal(_permute(cpa, c, st), ora.permute_batch(st).reshape(-1, 3, 4))
    assert ints(_permute(cpa, c, st[:1])) == po.permute(o, vals[:3])
    # empty batch is a no-op
    assert cpa.lib.akp_poseidon_permute_batch(c.handle().h, None, 0) == 0


@pytest.mark.parametrize("rate,weights", [(3, False), (4, False), (5, False), (6, False), (7, False), (8, False),
                                          (2, True), (4, True), (8, True)])
def test_permute_all_default_configs(cpa, rate, weights):
    c, o, ora = _cfg_pair(cpa, rate, weights)
    n, t = 257, rate + 1
    st = rand_fr_array(n * t, 77 + rate).reshape(n, t, 4)
    assert np.array_equal(_permute(cpa, c, st), ora.permute_batch(st, threads=8).reshape(n, t, 4))


def test_full_size_batch_sampled_parity(cpa):
    """BASELINE configs[1] size (2^20 states): compare a strided sample of 4096 states with the oracle and
    check the size-independent property that permuting a batch equals permuting its two halves."""
    c, o, ora = _cfg_pair(cpa, 2)
    n = 1 << 20
    st = rand_fr_array(n * 3, 0xA5A50002).reshape(n, 3, 4)
    got = _permute(cpa, c, st)
    idx = np.arange(0, n, n // 4096)
    assert np.array_equal(got[idx], ora.permute_batch(st[idx], threads=8).reshape(-1, 3, 4))
    halves = np.concatenate([_permute(cpa, c, st[: n // 2]), _permute(cpa, c, st[n // 2:])])
    assert np.array_equal(got, halves)
    assert np.array_equal(got[-1], ora.permute_batch(st[-1:]).reshape(3, 4))


def test_crh_lengths_and_golden(cpa, derived):
    from crypto_primitives_amd.crh import poseidon as pcrh
    from crypto_primitives_amd import field
    c, o, ora = _cfg_pair(cpa, 2)
    d = derived["poseidon_rate2"]
    assert str(field.to_ints(pcrh.CRH.evaluate(c, field.fr([1])))[0]) == d["crh_1"]
    assert str(field.to_ints(pcrh.CRH.evaluate(c, field.fr([1, 2])))[0]) == d["crh_1_2"]
    assert str(field.to_ints(pcrh.CRH.evaluate(c, field.fr([1, 2, 3])))[0]) == d["crh_1_2_3"]
    assert str(field.to_ints(pcrh.CRH.evaluate(c, np.zeros((0, 4), np.uint64)))[0]) == d["crh_empty"]
    assert str(field.to_ints(pcrh.TwoToOneCRH.compress(c, field.fr([1]), field.fr([2])))[0]) == d["compress_1_2"]
    assert np.array_equal(pcrh.TwoToOneCRH.evaluate(c, field.fr([1]), field.fr([2])), pcrh.TwoToOneCRH.compress(c, field.fr([1]), field.fr([2])))
    for k in (1, 2, 3, 4, 7):
        n = 1000 + k
        x = rand_fr_array(n * k, 500 + k).reshape(n, k, 4)
        assert np.array_equal(pcrh.CRH.evaluate_batch(c, x), ora.crh_batch(x, k, threads=8)), k
    with pytest.raises(NotImplementedError):
        pcrh.CRH.setup()


def test_crh_config1_2pow16(cpa):
    """BASELINE configs[0]: Poseidon sponge CRH over 2^16 inputs (2 Fr each)."""
    from crypto_primitives_amd.crh import poseidon as pcrh
    c, o, ora = _cfg_pair(cpa, 2)
    n = 1 << 16
    x = rand_fr_array(n * 2, 0xA5A50001).reshape(n, 2, 4)
    got = pcrh.CRH.evaluate_batch(c, x)
    assert np.array_equal(got, ora.crh_batch(x, 2, threads=8))
    # compress(l, r) == CRH([l, r]) (crh/poseidon/constraints.rs:84-92 relies on it)
    assert np.array_equal(pcrh.TwoToOneCRH.compress_batch(c, x[:, 0], x[:, 1]), got)


@pytest.mark.parametrize("rate", [3, 5, 8])
def test_crh_other_rates(cpa, rate):
    from crypto_primitives_amd.crh import poseidon as pcrh
    c, o, ora = _cfg_pair(cpa, rate)
    for k in (0, 1, rate - 1, rate, rate + 1, 2 * rate + 3):
        n = 130
        x = rand_fr_array(n * max(k, 1), 900 + k).reshape(n, max(k, 1), 4)[:, :k]
        got = pcrh.CRH.evaluate_batch(c, np.ascontiguousarray(x))
        if k == 0:
            assert np.array_equal(got, np.repeat(ora.crh_empty(), n, axis=0))
        else:
            assert np.array_equal(got, ora.crh_batch(np.ascontiguousarray(x), k, threads=4)), (rate, k)
    l, r = rand_fr_array(64, 1), rand_fr_array(64, 2)
    assert np.array_equal(pcrh.TwoToOneCRH.compress_batch(c, l, r), ora.two_to_one_batch(l, r))


def test_sponge_consistency_kat_on_gpu(cpa, kats):  # sponge/poseidon/mod.rs:381-404
    from crypto_primitives_amd import field
    k = kats["sponge_consistency"]
    c = cpa.get_default_poseidon_parameters(2, False)
    sp = cpa.PoseidonSponge(c)
    sp.absorb(field.fr([int(x) for x in k["absorb"]]))
    assert [str(x) for x in field.to_ints(sp.squeeze_native_field_elements(3))] == k["squeeze"]


def test_sponge_cross_fuzz(cpa):
    """model-based fuzz of the duplex state machine (sponge/poseidon/tests.rs:68-240), batch of 3 sponges."""
    from crypto_primitives_amd import field
    r = random.Random(11)
    for rate in (2, 3):
        c = cpa.get_default_poseidon_parameters(rate, False)
        o = po.get_default_poseidon_parameters(rate, False)
        for trial in range(6):
            B = 3
            sp = cpa.PoseidonSponge(c, batch=B)
            models = [po.PoseidonSponge(o) for _ in range(B)]
            for _ in range(r.randint(2, 9)):
                k = r.randint(0, 6)
                if r.random() < 0.5:
                    el = [rand_fr(k, r.randint(0, 1 << 30)) for _ in range(B)]
                    for m, e in zip(models, el):
                        m.absorb(e)
                    sp.absorb(field.fr([x for e in el for x in e]).reshape(B, k, 4))
                else:
                    exp = [m.squeeze_native_field_elements(k) for m in models]
                    got = sp.squeeze_native_field_elements(k).reshape(B, k, 4)
                    assert [field.to_ints(got[b]) for b in range(B)] == exp
            st, mode, idx = sp.into_state()
            for b, m in enumerate(models):
                assert field.to_ints(st[b]) == m.state
            assert (mode, idx) == ((0 if models[0].mode[0] == po.ABSORBING else 1), models[0].mode[1])
            sp2 = cpa.PoseidonSponge.from_state((st, mode, idx), c)
            assert [field.to_ints(x) for x in sp2.squeeze_native_field_elements(2).reshape(B, 2, 4)] == \
                   [m.squeeze_native_field_elements(2) for m in models]


def test_sponge_bytes_bits_and_regression(cpa):
    from crypto_primitives_amd import field
    c = cpa.get_default_poseidon_parameters(2, False)
    o = po.get_default_poseidon_parameters(2, False)
    a, b = cpa.PoseidonSponge(c), po.PoseidonSponge(o)
    a.absorb(field.fr([5, 6, 7])); b.absorb([5, 6, 7])
    assert a.squeeze_bytes(70) == b.squeeze_bytes(70)
    assert a.squeeze_bits(300) == b.squeeze_bits(300)
    # demo_bug (sponge/poseidon/tests.rs:12-65)
    x, y = cpa.PoseidonSponge(c), cpa.PoseidonSponge(c)
    x.absorb(field.fr([1, 2, 3])); y.absorb(field.fr([1, 2, 3]))
    part = np.concatenate([x.squeeze_native_field_elements(1), x.squeeze_native_field_elements(2)])
    assert np.array_equal(part, y.squeeze_native_field_elements(3))


def _custom_cfg(cpa, rate, capacity, rf, rp, alpha, mds_ints, seed):
    """a PoseidonConfig with caller-supplied parameters (PoseidonConfig::new, sponge/poseidon/mod.rs:191-217)"""
    from crypto_primitives_amd import field
    t = rate + capacity
    ark_ints = rand_fr((rf + rp) * t, seed)
    c = cpa.PoseidonConfig(rf, rp, alpha, field.fr(ark_ints).reshape(rf + rp, t, 4), field.fr([x for r in mds_ints for x in r]).reshape(t, t, 4), rate, capacity)
    o = po.PoseidonConfig(rf, rp, alpha, [ark_ints[i * t:(i + 1) * t] for i in range(rf + rp)], mds_ints, rate, capacity)
    return c, o


@pytest.mark.parametrize("case", ["near_mds", "rate1_cap2", "alpha5_random", "rp_even", "no_partial", "alpha3_t3", "many_partial",
                                  "many_partial_full_form", "rf2_full_form", "one_partial_full_form"])
def test_custom_t3_parameters(cpa, case):
    """t = 3 instances other than the default one: exercises the sparse-partial-round derivation, its fallback to
    dense rounds when a block is singular (the near-MDS matrix of merkle_tree/tests/test_utils.rs:643-653), other
    capacities / exponents / round counts."""
    from crypto_primitives_amd.crh import poseidon as pcrh
    near = [[1, 0, 1], [1, 1, 0], [0, 1, 1]]
    rnd = [rand_fr(3, 70 + i) for i in range(3)]
    c, o = {
        "near_mds": lambda: _custom_cfg(cpa, 2, 1, 8, 29, 17, near, 1),
        "rate1_cap2": lambda: _custom_cfg(cpa, 1, 2, 8, 31, 17, rnd, 2),
        "alpha5_random": lambda: _custom_cfg(cpa, 2, 1, 8, 56, 5, rnd, 3),
        "rp_even": lambda: _custom_cfg(cpa, 2, 1, 6, 10, 17, rnd, 4),
        "no_partial": lambda: _custom_cfg(cpa, 2, 1, 8, 0, 5, rnd, 5),
        "alpha3_t3": lambda: _custom_cfg(cpa, 2, 1, 2, 1, 3, rnd, 6),
        "many_partial": lambda: _custom_cfg(cpa, 2, 1, 2, 140, 3, rnd, 7),
        # parameter sets that admit every re-parameterised form (alpha coprime to p - 1), at the corners of the derivation
        "many_partial_full_form": lambda: _custom_cfg(cpa, 2, 1, 4, 140, 5, rnd, 8),
        "rf2_full_form": lambda: _custom_cfg(cpa, 2, 1, 2, 9, 5, rnd, 9),
        "one_partial_full_form": lambda: _custom_cfg(cpa, 2, 1, 4, 1, 17, rnd, 10),
    }[case]()
    ora = cref_poseidon(o)
    n = 300
    st = rand_fr_array(n * 3, 5).reshape(n, 3, 4)
    assert np.array_equal(_permute(cpa, c, st), ora.permute_batch(st, threads=4).reshape(n, 3, 4))
    assert ints(_permute(cpa, c, st[:1])) == po.permute(o, ints(st[0]))
    for k in (0, 1, 2, 3, 5):
        x = rand_fr_array(64 * max(k, 1), 50 + k).reshape(64, max(k, 1), 4)[:, :k]
        got = pcrh.CRH.evaluate_batch(c, np.ascontiguousarray(x))
        exp = np.repeat(ora.crh_empty(), 64, axis=0) if k == 0 else ora.crh_batch(np.ascontiguousarray(x), k)
        assert np.array_equal(got, exp), (case, k)
    l, r = rand_fr_array(32, 8), rand_fr_array(32, 9)
    assert np.array_equal(pcrh.TwoToOneCRH.compress_batch(c, l, r), ora.two_to_one_batch(l, r))
    # above the latency-kernel switch: the one-lane-per-item register kernels
    n = 33000
    st = rand_fr_array(n * 3, 6).reshape(n, 3, 4)
    st[0] = 0
    assert np.array_equal(_permute(cpa, c, st), ora.permute_batch(st, threads=16).reshape(n, 3, 4))
    for k in (0, 1, 2, 3):
        x = rand_fr_array(n * max(k, 1), 60 + k).reshape(n, max(k, 1), 4)[:, :k]
        got = pcrh.CRH.evaluate_batch(c, np.ascontiguousarray(x))
        exp = np.repeat(ora.crh_empty(), n, axis=0) if k == 0 else ora.crh_batch(np.ascontiguousarray(x), k, threads=16)
        assert np.array_equal(got, exp), (case, k, "register kernel")


GENERIC_SHAPES = ((1, 1, 4, 7, 5), (2, 2, 4, 7, 5), (3, 1, 4, 70, 5), (7, 2, 8, 20, 5), (12, 4, 2, 3, 3), (15, 1, 4, 9, 5))


@pytest.mark.parametrize("shape", GENERIC_SHAPES)
def test_generic_kernel_on_custom_t(cpa, shape):
    """t = 2 .. 16 with non-default parameters through the generic kernels (batches up to
    AKP_POSEIDON_GENERIC_COOP_MAX: one wave per state lane; above: state in an LDS file, one lane per item -- see
    test_generic_kernels_both_ways): permutation on a ragged batch, CRH with 0, 1, rate, rate + 1, 2*rate + 3 inputs."""
    from crypto_primitives_amd.crh import poseidon as pcrh
    rate, cap, rf, rp, alpha = shape
    t = rate + cap
    mds = [rand_fr(t, 200 + i + t) for i in range(t)]
    c, o = _custom_cfg(cpa, rate, cap, rf, rp, alpha, mds, 10 + t)
    ora = cref_poseidon(o)
    n = 130
    st = rand_fr_array(n * t, 3).reshape(n, t, 4)
    assert np.array_equal(_permute(cpa, c, st), ora.permute_batch(st, threads=8).reshape(n, t, 4))
    for k in (0, 1, rate, rate + 1, 2 * rate + 3):
        x = rand_fr_array(70 * max(k, 1), 40 + k).reshape(70, max(k, 1), 4)[:, :k]
        got = pcrh.CRH.evaluate_batch(c, np.ascontiguousarray(x))
        exp = np.repeat(ora.crh_empty(), 70, axis=0) if k == 0 else ora.crh_batch(np.ascontiguousarray(x), k, threads=8)
        assert np.array_equal(got, exp), (shape, k)


@pytest.mark.parametrize("coop_max,no_reg", [("0", ""), ("0", "1"), ("1000000000", "")])
def test_generic_kernels_both_ways(cpa, coop_max, no_reg):
    """the same checks with every batch forced through the one-lane-per-item kernels (0: register-resident for t = 4 .. 9,
    LDS-file otherwise; with AKP_POSEIDON_NO_REG_T=1 the LDS-file kernels for every t) / the wave-per-lane kernels (10^9),
    in a fresh process (the switches are read once)"""
    import os, subprocess, sys
    here = os.path.abspath(__file__)
    env = dict(os.environ, AKP_POSEIDON_GENERIC_COOP_MAX=coop_max)
    if no_reg:  # an A/B arm: readable only by the test build of the library (make testhooks)
        env["AKP_POSEIDON_NO_REG_T"] = no_reg
        env["AKP_LIB"] = os.path.join(os.path.dirname(os.path.dirname(here)), "crypto_primitives_amd", "lib", "libakp_testhooks.so")
        assert os.path.exists(env["AKP_LIB"]), "make -C crypto_primitives_amd/csrc testhooks"
    r = subprocess.run([sys.executable, "-m", "pytest", here, "-q", "-x", "-m", "gpu", "-k",
                        "test_generic_kernel_on_custom_t or test_crh_other_rates or test_permute_all_default_configs"],
                       env=env, capture_output=True, text=True, timeout=900,
                       cwd=os.path.dirname(os.path.dirname(here)))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


# ---- the CRH path has two kernels for t = 3: one lane per item (large batches) and the wave-per-lane latency kernel
# (batches <= AKP_POSEIDON_COOP_MAX, default 2^15).  Both must equal the oracle on every shape.
@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 191, 193, 5000, (1 << 15), (1 << 15) + 1, 40000])
def test_crh_t3_small_and_large_batches(cpa, n):
    from crypto_primitives_amd.crh import poseidon as pcrh
    c, o, ora = _cfg_pair(cpa, 2)
    l, r = rand_fr_array(n, 100 + n), rand_fr_array(n, 200 + n)
    assert np.array_equal(pcrh.TwoToOneCRH.compress_batch(c, l, r), ora.two_to_one_batch(l, r, threads=8))
    one = np.ascontiguousarray(l.reshape(n, 1, 4))
    assert np.array_equal(pcrh.CRH.evaluate_batch(c, one), ora.crh_batch(one, 1, threads=8))
    two = np.ascontiguousarray(np.stack([l, r], axis=1))
    assert np.array_equal(pcrh.CRH.evaluate_batch(c, two), ora.crh_batch(two, 2, threads=8))
    if n <= 5000:  # three elements = two permutations: never the latency kernel
        three = np.ascontiguousarray(np.stack([l, r, l], axis=1))
        assert np.array_equal(pcrh.CRH.evaluate_batch(c, three), ora.crh_batch(three, 3, threads=8))


def test_crh_t3_empty_input_small_batch(cpa):
    from crypto_primitives_amd.crh import poseidon as pcrh
    c, o, ora = _cfg_pair(cpa, 2)
    got = pcrh.CRH.evaluate_batch(c, np.zeros((5, 0, 4), dtype=np.uint64))
    assert got.shape == (5, 4) and np.array_equal(got, np.repeat(ora.crh_empty(), 5, axis=0))
    assert ints(got[:1])[0] == po.crh_evaluate(o, [])


def test_crh_t3_latency_kernel_disabled_matches(cpa, tmp_path):
    """AKP_POSEIDON_COOP_MAX=0 routes small batches through the one-lane-per-item kernel: same digests."""
    import subprocess, sys, os
    from crypto_primitives_amd.crh import poseidon as pcrh
    c, o, ora = _cfg_pair(cpa, 2)
    n = 777
    l, r = rand_fr_array(n, 31), rand_fr_array(n, 32)
    np.save(tmp_path / "l.npy", l)
    np.save(tmp_path / "r.npy", r)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, numpy as np; sys.path.insert(0, %r); import crypto_primitives_amd as cpa; "
            "from crypto_primitives_amd.crh import poseidon as pcrh; "
            "c = cpa.get_default_poseidon_parameters(2, False); "
            "np.save(%r, pcrh.TwoToOneCRH.compress_batch(c, np.load(%r), np.load(%r)))"
            % (root, str(tmp_path / "o.npy"), str(tmp_path / "l.npy"), str(tmp_path / "r.npy")))
    subprocess.run([sys.executable, "-c", code], check=True, env=dict(os.environ, AKP_POSEIDON_COOP_MAX="0"), timeout=300)
    assert np.array_equal(np.load(tmp_path / "o.npy"), pcrh.TwoToOneCRH.compress_batch(c, l, r))
    assert np.array_equal(np.load(tmp_path / "o.npy"), ora.two_to_one_batch(l, r, threads=8))


@pytest.mark.parametrize("case", ["near_mds_dense", "alpha3_no_roots", "alpha5_full_form", "different_leaf_and_two_to_one", "mixed_forms_decline", "rate1_declines"])
def test_path_verify_walk_kernel_on_custom_parameter_sets(cpa, case):
    """Path::verify for > 2^15 paths (the one-launch kernel in which each lane walks its path) with parameter sets that take its
    other instantiation or make it decline: a near-MDS matrix whose partial-round block is singular (dense rounds), alpha = 3
    (gcd(3, p - 1) = 3: no alpha-th roots, no re-parameterised forms), alpha = 5 (full form), DIFFERENT leaf and two-to-one
    parameters, a full-form set mixed with a plain one and a rate-1 set (both outside the kernel: level by level).  Flags against
    paths built from the oracle's tree; corruptions must fail exactly their own path."""
    from crypto_primitives_amd._lib import lib, check
    near = [[1, 0, 1], [1, 1, 0], [0, 1, 1]]
    rnd = [rand_fr(3, 170 + i) for i in range(3)]
    mk = {"near": lambda s: _custom_cfg(cpa, 2, 1, 8, 29, 17, near, s), "a3": lambda s: _custom_cfg(cpa, 2, 1, 4, 11, 3, rnd, s),
          "a5": lambda s: _custom_cfg(cpa, 2, 1, 8, 22, 5, rnd, s), "r1": lambda s: _custom_cfg(cpa, 1, 2, 8, 12, 5, rnd, s)}
    (cl, ol), (ct, ot) = {"near_mds_dense": (mk["near"](1), mk["near"](1)), "alpha3_no_roots": (mk["a3"](2), mk["a3"](2)),
                          "alpha5_full_form": (mk["a5"](3), mk["a5"](3)), "different_leaf_and_two_to_one": (mk["a5"](4), mk["a5"](5)),
                          "mixed_forms_decline": (mk["a5"](6), mk["a3"](7)), "rate1_declines": (mk["r1"](8), mk["r1"](8))}[case]
    oral, orat = cref_poseidon(ol), cref_poseidon(ot)
    n, m, leaf_len = 256, 33000, 2 if case != "rate1_declines" else 1
    depth = 8 - 1
    leaves = rand_fr_array(n * leaf_len, 31).reshape(n, leaf_len, 4)
    # the oracle's tree with the two parameter sets: leaf digests, then level by level
    level = np.asarray(oral.crh_batch(np.ascontiguousarray(leaves), leaf_len, threads=4)).reshape(n, 4)
    ln, inner = level, []
    while len(level) > 1:
        level = np.asarray(orat.two_to_one_batch(np.ascontiguousarray(level[0::2]), np.ascontiguousarray(level[1::2]), threads=4)).reshape(-1, 4)
        inner.append(level)
    nl = np.concatenate(inner[::-1])  # heap order: root first
    rng = np.random.default_rng(9)
    idx = rng.integers(0, n, size=m).astype(np.uint64)
    sib, auth = np.empty((m, 4), np.uint64), np.empty((m, depth, 4), np.uint64)
    check(lib.akp_merkle_gather_paths(np.ascontiguousarray(ln).ctypes.data, np.ascontiguousarray(nl).ctypes.data, n, 1, idx.ctypes.data, m, sib.ctypes.data,
                                      auth.ctypes.data))
    lv = np.ascontiguousarray(leaves[idx.astype(np.int64)])
    want = np.ones(m, np.uint8)
    bad = rng.choice(m, size=4, replace=False)
    lv[bad[0], leaf_len - 1, 1] ^= 2
    sib[bad[1], 0] ^= 1
    auth[bad[2], 0, 3] ^= 1
    auth[bad[3], depth - 1, 0] ^= 4
    want[bad] = 0
    ok = np.zeros(m, np.uint8)
    root = np.ascontiguousarray(nl[0])
    check(lib.akp_merkle_verify_paths_poseidon(cl.handle().h, ct.handle().h, root.ctypes.data, lv.ctypes.data, m, leaf_len, idx.ctypes.data, sib.ctypes.data,
                                               auth.ctypes.data, depth, ok.ctypes.data))
    assert np.array_equal(ok, want), (case, np.flatnonzero(ok != want)[:8])
