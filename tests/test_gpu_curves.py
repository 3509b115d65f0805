"""GPU parity: Pedersen / Bowe-Hopwood CRH over Jubjub through the C ABI vs the oracle (bit-exact)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import jubjub as jj, pedersen as opd, bowe_hopwood as obh, fr as ofr, cref  # noqa: E402
from helpers import ints, gens_array  # noqa: E402


@pytest.fixture(scope="module")
def cpa():
    import crypto_primitives_amd as m
    assert m.lib.akp_device_count() >= 1
    return m


@pytest.fixture(scope="module")
def ped(cpa):
    from crypto_primitives_amd.crh import pedersen
    g = jj.pedersen_generators(0xA5A50004, 4, 256)
    return pedersen.Parameters(gens_array(g)), g, cref.CurveParams(4, 256, gens_array(g))


@pytest.fixture(scope="module")
def bhp(cpa):
    from crypto_primitives_amd.crh import bowe_hopwood
    g = jj.bowe_hopwood_generators(0xA5A50005, 63, 9)
    return bowe_hopwood.Parameters(gens_array(g)), g, cref.CurveParams(63, 9, gens_array(g))


def _msgs(n, L, seed):
    return np.frombuffer(ofr.SplitMix64(seed).bytes(max(n * L, 1)), dtype=np.uint8)[: n * L].reshape(n, L).copy()


def test_pedersen_golden_and_known_answers(cpa, ped, derived):
    from crypto_primitives_amd.crh import pedersen
    from crypto_primitives_amd import field
    P, g, _ = ped
    d = derived["pedersen_4x256"]
    msg = bytes.fromhex(d["msg"])
    assert [str(v) for v in field.to_ints(pedersen.CRH.evaluate(P, msg))] == d["digest"]
    assert [str(v) for v in field.to_ints(pedersen.CRH.evaluate(P, msg[:32]))] == d["digest_first32"]
    ident = field.to_ints(pedersen.CRH.evaluate(P, b""))
    assert ident == [0, 1] == field.to_ints(pedersen.CRH.evaluate(P, bytes(128)))  # identity (pedersen/mod.rs:116-122)
    h, h32 = pedersen.CRH.evaluate(P, msg), pedersen.CRH.evaluate(P, msg[:32])
    assert [str(v) for v in field.to_ints(pedersen.TwoToOneCRH.compress(P, h, h32))] == d["compress_h_h32"]
    assert np.array_equal(pedersen.TwoToOneCRH.evaluate(P, msg[:64], msg[64:]), h)
    with pytest.raises(cpa.IncorrectInputLength):  # reference panics (pedersen/mod.rs:82-89)
        pedersen.CRH.evaluate(P, bytes(129))

    class W4x256(pedersen.Window):
        WINDOW_SIZE, NUM_WINDOWS = 4, 256

    class W4x128(pedersen.Window):
        WINDOW_SIZE, NUM_WINDOWS = 4, 128
    pedersen.CRH.evaluate(P, msg, window=W4x256)
    with pytest.raises(AssertionError):
        pedersen.CRH.evaluate(P, msg[:8], window=W4x128)


@pytest.mark.parametrize("L", [128, 32, 1, 77])
def test_pedersen_batch_vs_oracle(cpa, ped, L):
    from crypto_primitives_amd.crh import pedersen
    P, g, C = ped
    n = 2051
    m = _msgs(n, L, 0xA5A50004 + L)
    got = pedersen.CRH.evaluate_batch(P, m)
    exp = C.pedersen_crh_batch(m, n, L, threads=8)
    assert np.array_equal(got, exp)
    # python big-int oracle on a few
    for i in (0, n - 1):
        assert tuple(ints(got[i])) == opd.evaluate(g, 4, 256, bytes(m[i]))


def test_pedersen_other_windows(cpa):
    from crypto_primitives_amd.crh import pedersen
    for W, N in ((8, 20), (6, 10), (3, 7), (1, 16)):
        g = jj.pedersen_generators(31 + W, W, N)
        P = pedersen.Parameters(gens_array(g))
        for L in {W * N // 8, 1, 0}:
            m = _msgs(5, L, W * 100 + L)
            got = pedersen.CRH.evaluate_batch(P, m if L else [b""] * 5)
            for i in range(5):
                assert tuple(ints(got[i])) == opd.evaluate(g, W, N, bytes(m[i]) if L else b""), (W, N, L)


def test_bowe_hopwood_golden_and_known_answers(cpa, bhp, derived):
    from crypto_primitives_amd.crh import bowe_hopwood
    from crypto_primitives_amd import field
    P, g, _ = bhp
    d = derived["bowe_hopwood_63x9"]
    x32 = bowe_hopwood.CRH.evaluate(P, bytes.fromhex(d["msg32"]))
    x70 = bowe_hopwood.CRH.evaluate(P, bytes.fromhex(d["msg70"]))
    assert str(field.to_ints(x32)[0]) == d["digest32"] and str(field.to_ints(x70)[0]) == d["digest70"]
    assert str(field.to_ints(bowe_hopwood.TwoToOneCRH.compress(P, x32, x70))[0]) == d["compress"]
    assert str(field.to_ints(bowe_hopwood.CRH.evaluate(P, bytes(3)))[0]) == d["zero_3bytes"]
    assert field.to_ints(bowe_hopwood.CRH.evaluate(P, b""))[0] == 0            # identity -> x = 0
    with pytest.raises(cpa.IncorrectInputLength):                               # 213 B > 63*9*3 bits
        bowe_hopwood.CRH.evaluate(P, bytes(213))
    # test_simple_bh (crh/bowe_hopwood/mod.rs:258-271): 63x8 window, input [1,2,3]
    g8 = jj.bowe_hopwood_generators(3, 63, 8)
    P8 = bowe_hopwood.Parameters(gens_array(g8))
    assert field.to_ints(bowe_hopwood.CRH.evaluate(P8, bytes([1, 2, 3])))[0] == obh.evaluate(g8, 63, 8, bytes([1, 2, 3]))

    class Big(bowe_hopwood._ped.Window):
        WINDOW_SIZE, NUM_WINDOWS = 64, 2
    with pytest.raises(ValueError):
        bowe_hopwood.CRH.setup(Big)


@pytest.mark.parametrize("L", [32, 70, 3, 212])
def test_bowe_hopwood_batch_vs_oracle(cpa, bhp, L):
    from crypto_primitives_amd.crh import bowe_hopwood
    P, g, C = bhp
    n = 1500
    m = _msgs(n, L, 0xA5A50005 + L)
    got = bowe_hopwood.CRH.evaluate_batch(P, m)
    assert np.array_equal(got, C.bh_crh_batch(m, n, L, threads=8))
    assert ints(got[7])[0] == obh.evaluate(g, 63, 9, bytes(m[7]))


def test_two_to_one_batches(cpa, ped, bhp):
    from crypto_primitives_amd.crh import pedersen, bowe_hopwood
    P, g, _ = ped
    l, r = _msgs(9, 64, 1), _msgs(9, 64, 2)
    got = pedersen.TwoToOneCRH.evaluate_batch(P, l, r)
    for i in range(9):
        assert tuple(ints(got[i])) == opd.two_to_one_evaluate(g, 4, 256, bytes(l[i]), bytes(r[i]))
    l, r = _msgs(9, 20, 3), _msgs(9, 20, 4)  # shorter halves: zero tail
    got = pedersen.TwoToOneCRH.evaluate_batch(P, l, r)
    for i in range(9):
        assert tuple(ints(got[i])) == opd.two_to_one_evaluate(g, 4, 256, bytes(l[i]), bytes(r[i]))
    B, gb, _ = bhp
    l, r = _msgs(9, 32, 5), _msgs(9, 32, 6)
    got = bowe_hopwood.TwoToOneCRH.evaluate_batch(B, l, r)
    for i in range(9):
        assert ints(got[i])[0] == obh.two_to_one_evaluate(gb, 63, 9, bytes(l[i]), bytes(r[i]))
    l, r = _msgs(4, 40, 7), _msgs(4, 40, 8)  # 80 bytes > 70-byte buffer: zip truncation (:219-224)
    got = bowe_hopwood.TwoToOneCRH.evaluate_batch(B, l, r)
    for i in range(4):
        assert ints(got[i])[0] == obh.two_to_one_evaluate(gb, 63, 9, bytes(l[i]), bytes(r[i]))


def test_digests_on_curve_and_in_subgroup(cpa, ped):
    from crypto_primitives_amd.crh import pedersen
    P, g, _ = ped
    got = pedersen.CRH.evaluate_batch(P, _msgs(4, 128, 99))
    for i in range(4):
        pt = tuple(ints(got[i]))
        assert jj.is_on_curve(pt) and jj.mul(pt, jj.SUBGROUP_ORDER) == jj.IDENTITY


def test_setup_generators(cpa):
    from crypto_primitives_amd.crh import pedersen, bowe_hopwood
    from crypto_primitives_amd import field

    class W(pedersen.Window):
        WINDOW_SIZE, NUM_WINDOWS = 4, 8
    P = pedersen.CRH.setup(W, seed=5)
    go = jj.pedersen_generators(5, 4, 8, bases=jj.random_bases)
    assert field.to_ints(P.generators) == [v for row in go for pt in row for v in pt]
    assert tuple(field.to_ints(pedersen.CRH.evaluate(P, bytes([0xff, 1, 2, 3])))) == opd.evaluate(go, 4, 8, bytes([0xff, 1, 2, 3]))
    B = bowe_hopwood.CRH.setup(W, seed=6)
    gb = jj.bowe_hopwood_generators(6, 4, 8, bases=jj.random_bases)
    assert field.to_ints(bowe_hopwood.CRH.evaluate(B, bytes(range(12))))[0] == obh.evaluate(gb, 4, 8, bytes(range(12)))


def test_table_variants_agree(cpa):
    """the table digit width (Pedersen) / chunk grouping (Bowe-Hopwood) are tuning choices (akp_te_params_create_shaped): every
    shape must give the same digests -- narrow ones, the widths the table budget picks on a 288 GB device (24-bit digits, groups
    of 8 chunks: built from two part tables, te_build_combine_kernel) and every message length, so that the remainder step of
    the Bowe-Hopwood tables (the < G chunks after the last full group) is exercised for every remainder size."""
    from crypto_primitives_amd.crh import pedersen, bowe_hopwood
    g = jj.pedersen_generators(77, 5, 13)   # 65 generators: not a multiple of any digit width
    gb = jj.bowe_hopwood_generators(78, 7, 5)  # 35 chunks: not a multiple of 3
    m = _msgs(40, 8, 5)
    mb = _msgs(40, 13, 6)
    ref_p = ref_b = None
    for D, grp in ((12, 4), (8, 3), (4, 1), (7, 2), (2, 1), (3, 3), (13, 4), (15, 5), (20, 6), (24, 8), (23, 7), (0, 0)):
        dp = pedersen.CRH.evaluate_batch(pedersen.Parameters(gens_array(g), table_shape=D), m)
        db = bowe_hopwood.CRH.evaluate_batch(bowe_hopwood.Parameters(gens_array(gb), table_shape=grp), mb)
        if ref_p is None:
            ref_p, ref_b = dp, db
            for i in range(0, 40, 13):
                assert tuple(ints(dp[i])) == opd.evaluate(g, 5, 13, bytes(m[i]))
                assert ints(db[i])[0] == obh.evaluate(gb, 7, 5, bytes(mb[i]))
        assert np.array_equal(dp, ref_p) and np.array_equal(db, ref_b), (D, grp)
    # every message length 0 .. 13 bytes (0 .. 35 chunks) at every group size: all remainders 0 .. G - 1, more lengths than a
    # handle keeps remainder tables for (the later ones walk the chunks one by one), against the oracle
    for grp in (8, 7, 5, 3, 2):
        B = bowe_hopwood.Parameters(gens_array(gb), table_shape=grp)
        for L in range(0, 14):
            ml = _msgs(5, L, 90 + L) if L else np.zeros((5, 0), np.uint8)
            d = bowe_hopwood.CRH.evaluate_batch(B, ml)
            for i in (0, 4):
                assert ints(d[i])[0] == obh.evaluate(gb, 7, 5, bytes(ml[i])), (grp, L, i)
        assert B.handle().info(13)["digit_bits_or_group"] == grp
    # (the plain table of round 1 as an arm of its own: tests/test_gpu_multi_slots.py, test build only; here it runs as the
    # fallback for generators outside the prime subgroup, test_pedersen_generators_outside_the_prime_subgroup)


def test_table_budget_picks_the_shape(cpa):
    """akp_ctx_set_table_budget: the handle gets the widest table that fits (4x256 Pedersen: 268 MB for 16-bit digits, 3.5 GB
    for 20), the default follows the device's memory, and the digests do not change"""
    from crypto_primitives_amd import params as cparams
    from crypto_primitives_amd.crh import pedersen, bowe_hopwood
    ctx = cpa.default_context(0)
    gens = cparams.pedersen_generators(0xA5A50004, 4, 256)
    bgens = cparams.bowe_hopwood_generators(0xA5A50005, 63, 9)
    m = _msgs(300, 128, 11)
    mb = _msgs(300, 64, 12)
    out = {}
    try:
        for budget, want_d, want_g in ((320 << 20, 16, 5), (4 << 30, 20, 6), (16 << 30, 22, 7)):
            ctx.set_table_budget(budget)
            assert ctx.table_budget() == budget
            P, B = pedersen.Parameters(gens), bowe_hopwood.Parameters(bgens)
            if want_d > 16:  # round 6: a budget above the default starts on the cache-sized table; the wide one is built in the background --
                st = P.handle(ctx).table_info()["last_build"]["upgrade_state"]
                assert st in (1, 2) and B.handle(ctx).table_info()["last_build"]["upgrade_state"] in (1, 2), st
                assert P.handle(ctx).info(128)["digit_bits_or_group"] == 16 and B.handle(ctx).info(64)["digit_bits_or_group"] == 5  # what a call would use NOW
            P.handle(ctx).prepare(128)  # -- or, as here, by the call that waits for it
            ip = P.handle(ctx).info(128)
            B.handle(ctx).prepare(64)
            ib = B.handle(ctx).info(64)
            assert ip["digit_bits_or_group"] == want_d and ip["table_bytes"] <= budget + (1 << 20), ip
            assert ib["digit_bits_or_group"] == want_g and ib["table_bytes"] <= budget + (1 << 20), ib
            assert ip["steps"] == -(-1024 // want_d) and ib["steps"] == 171 // want_g + 1
            out[budget] = (pedersen.CRH.evaluate_batch(P, m), bowe_hopwood.CRH.evaluate_batch(B, mb))
            assert B.handle(ctx).info(64) == ib and P.handle(ctx).info(128) == ip  # hashing found everything built
            del P, B
    finally:
        ctx.set_table_budget(0)
    assert ctx.table_budget() == 320 << 20  # the default: cache-sized tables (round 5; round 4 sized them from the device's memory)
    from crypto_primitives_amd._lib import TABLE_BUDGET_DEVICE
    ctx.set_table_budget(TABLE_BUDGET_DEVICE)
    try:
        assert ctx.table_budget() >= 64 << 20 and ctx.table_budget() != TABLE_BUDGET_DEVICE  # a quarter of the device's memory
    finally:
        ctx.set_table_budget(0)
    # shapes outside 2..24 bits / 1..8 chunks are the caller's error
    import ctypes as C
    from crypto_primitives_amd._lib import lib, AKP_ERR_BAD_PARAMS
    h = C.c_void_p()
    g = np.ascontiguousarray(gens)
    for kind, shape in ((0, 25), (0, 1), (2, 99), (1, 9)):
        assert lib.akp_te_params_create_shaped(ctx.h, kind, 4 if kind != 1 else 63, 256 if kind != 1 else 9, (g if kind != 1 else np.ascontiguousarray(bgens)).ctypes.data,
                                               shape, C.byref(h)) == AKP_ERR_BAD_PARAMS, (kind, shape)
    a = out[320 << 20]
    for b in out.values():
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_upstream_jubjub_kat_on_the_gpu(cpa, jubjub_kat):
    """the Pedersen kernels (table build from the caller's generators, mixed additions, shared inversion, wire encoding of the
    coordinates) reproduce ark-ed-on-bls12-381's scalar-multiplication vector: one window of 256 generators 2^j * g -- computed
    by the PRODUCT's host code, not by the oracle -- evaluates (f1 * f2) * g of the upstream test.  Both device paths; the plain
    table repeats it in the test build (tests/test_gpu_multi_slots.py)."""
    import os
    from crypto_primitives_amd import params as cparams, field
    from crypto_primitives_amd.crh import pedersen
    k = jubjub_kat
    scalar = (k["f1"] * k["f2"]) % jj.SUBGROUP_ORDER
    pts, cur = [], (k["g"][0], k["g"][1], 1)
    for _ in range(256):
        pts.append(cparams._affine(cur))
        cur = cparams._padd(cur, cur)
    gens = field.fr([c for pt in pts for c in pt]).reshape(1, 256, 2, 4)
    msg = np.frombuffer(scalar.to_bytes(32, "little"), dtype=np.uint8)
    P = pedersen.Parameters(gens)
    assert tuple(ints(pedersen.CRH.evaluate(P, bytes(msg)))) == k["f1f2g"]
    batch = np.tile(msg, (20000, 1))  # accumulate + finalize kernels
    batch[1:, 0] ^= np.arange(1, 20000, dtype=np.uint64).astype(np.uint8)  # other scalars around it; row 0 stays the KAT
    got = pedersen.CRH.evaluate_batch(P, batch)
    assert tuple(ints(got[0])) == k["f1f2g"]
    assert tuple(ints(got[256])) == k["f1f2g"]  # 256 & 0xff == 0: the same scalar again


def test_pedersen_generators_outside_the_prime_subgroup(cpa):
    """`Parameters.generators` is a public field: the points need not lie in the prime-order subgroup.  The signed-subset
    table halves the generators, which only exists for odd order; such parameter sets must fall back to the plain table and
    still hash like the oracle (whose group law is complete on the whole curve)."""
    from crypto_primitives_amd.crh import pedersen
    g = jj.pedersen_generators(91, 4, 8)
    t2 = (0, jj.Q - 1)  # the point of order 2
    g[3][1] = jj.add(g[3][1], t2)
    g[7][0] = t2
    assert jj.mul(g[3][1], jj.SUBGROUP_ORDER) != jj.IDENTITY
    P = pedersen.Parameters(gens_array(g))
    for n in (5, 20000):  # split kernel and accumulate + finalize
        m = _msgs(n, 4, 17 + n)
        got = pedersen.CRH.evaluate_batch(P, m)
        for i in list(range(0, n, max(1, n // 7))) + [n - 1]:
            assert tuple(ints(got[i])) == opd.evaluate(g, 4, 8, bytes(m[i])), (n, i)


# ---- two device paths: accumulate + shared-inversion finalize (large batches) and the 8-wave split kernel
# (batches <= AKP_TE_SPLIT_MAX, default 2^14).  Both must equal the oracle on every shape.
@pytest.mark.parametrize("n", [1, 63, 64, 65, 1000, 1 << 14, (1 << 14) + 1, 40000])
def test_both_device_paths_pedersen(cpa, ped, n):
    from crypto_primitives_amd.crh import pedersen
    P, g, C = ped
    for L in (128, 32):
        m = _msgs(n, L, 7000 + n + L)
        assert np.array_equal(pedersen.CRH.evaluate_batch(P, m), C.pedersen_crh_batch(m, n, L, threads=16)), (n, L)


@pytest.mark.parametrize("n", [1, 63, 64, 65, 1000, 1 << 14, (1 << 14) + 1, 40000])
def test_both_device_paths_bowe_hopwood(cpa, bhp, n):
    from crypto_primitives_amd.crh import bowe_hopwood
    P, g, C = bhp
    for L in (70, 32, 1):
        m = _msgs(n, L, 9000 + n + L)
        assert np.array_equal(bowe_hopwood.CRH.evaluate_batch(P, m), C.bh_crh_batch(m, n, L, threads=16)), (n, L)


def test_split_kernel_disabled_matches(cpa, ped, bhp, tmp_path):
    """AKP_TE_SPLIT_MAX=0 sends small batches through accumulate + finalize: same digests."""
    import os, subprocess, sys
    from crypto_primitives_amd.crh import pedersen, bowe_hopwood
    P, _, _ = ped
    B, _, _ = bhp
    mp, mb = _msgs(333, 128, 1), _msgs(333, 70, 2)
    np.save(tmp_path / "mp.npy", mp)
    np.save(tmp_path / "mb.npy", mb)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r); "
            "from oracle import jubjub as jj; from helpers import gens_array; "
            "from crypto_primitives_amd.crh import pedersen, bowe_hopwood; "
            "P = pedersen.Parameters(gens_array(jj.pedersen_generators(0xA5A50004, 4, 256))); "
            "B = bowe_hopwood.Parameters(gens_array(jj.bowe_hopwood_generators(0xA5A50005, 63, 9))); "
            "np.save(%r, pedersen.CRH.evaluate_batch(P, np.load(%r))); np.save(%r, bowe_hopwood.CRH.evaluate_batch(B, np.load(%r)))"
            % (root, os.path.join(root, "tests"), str(tmp_path / "op.npy"), str(tmp_path / "mp.npy"), str(tmp_path / "ob.npy"), str(tmp_path / "mb.npy")))
    subprocess.run([sys.executable, "-c", code], check=True, env=dict(os.environ, AKP_TE_SPLIT_MAX="0"), timeout=600)
    assert np.array_equal(np.load(tmp_path / "op.npy"), pedersen.CRH.evaluate_batch(P, mp))
    assert np.array_equal(np.load(tmp_path / "ob.npy"), bowe_hopwood.CRH.evaluate_batch(B, mb))


def test_table_info_of_the_baseline_windows(cpa):
    """akp_te_params_info for the BASELINE windows.  With a 320 MiB table budget (the tables of rounds 1-3, inside the Infinity
    Cache): Pedersen 4x256 with 16-bit signed digits (64 steps, one 128-byte line per entry), Bowe-Hopwood 63x9 with five chunks
    per step.  With the default budget on a 288 GB device: 24-bit digits (43 steps, 46 GB) and groups of eight chunks (75 GB);
    the chunks a message length leaves after its last full group are one more step.  `table_bytes` is what the handle HOLDS: the
    wide table is built for the message lengths that arrive (here: after one message of the maximum length)."""
    from crypto_primitives_amd import params as cparams
    from crypto_primitives_amd.crh import pedersen, bowe_hopwood
    from crypto_primitives_amd._lib import TABLE_BUDGET_DEVICE
    ctx = cpa.default_context(0)
    # generators of this test's own: tables are shared per device among handles of equal parameters (round 5), and "nothing
    # hashed yet" below must not meet a table another test's live handle has already built
    pg, bg = cparams.pedersen_generators(0xB5B50004, 4, 256), cparams.bowe_hopwood_generators(0xB5B50005, 63, 9)
    longest_p, longest_b = _msgs(1, 128, 1), _msgs(1, 212, 2)
    ctx.set_table_budget(320 << 20)
    try:
        P, B = pedersen.Parameters(pg), bowe_hopwood.Parameters(bg)
        hp, hb = P.handle(), B.handle()
        assert hp.info(128) == {"digit_bits_or_group": 16, "signed_subset": True, "table_bytes": 65 * 128, "steps": 64}  # nothing hashed yet
        pedersen.CRH.evaluate_batch(P, longest_p)
        assert hp.info(128) == {"digit_bits_or_group": 16, "signed_subset": True, "table_bytes": ((64 << 15) + 65) * 128, "steps": 64}
        assert hp.info(32)["steps"] == 16 and hp.info(0)["steps"] == 0
        assert hb.info(32) == {"digit_bits_or_group": 5, "signed_subset": False, "table_bytes": 567 * 4 * 128, "steps": 18}
        bowe_hopwood.CRH.evaluate_batch(B, longest_b)
        assert hb.info(32) == {"digit_bits_or_group": 5, "signed_subset": False, "table_bytes": (567 * 4 + (113 << 14) + 8) * 128, "steps": 18}  # + the 1-chunk remainder of 212 bytes
        assert hb.info(70)["steps"] == 38 and hb.info(64)["steps"] == 35  # 187 = 37 * 5 + 2 chunks: the two left over are one step
        del hp, hb, P, B
    finally:
        ctx.set_table_budget(0)
    ctx.set_table_budget(TABLE_BUDGET_DEVICE)
    try:
        wide = ctx.table_budget() >= 71 << 30  # an idle 288 GB device
        P, B = pedersen.Parameters(pg), bowe_hopwood.Parameters(bg)
        hp, hb = P.handle(), B.handle()
    finally:
        ctx.set_table_budget(0)
    if wide:
        pedersen.CRH.evaluate_batch(P, longest_p)  # (starts on the cache-sized table, asks for the wide one)
        assert hp.wait_for_wide_table(128) is not None
        assert hp.info(128) == {"digit_bits_or_group": 24, "signed_subset": True, "table_bytes": ((43 << 23) + 44) * 128, "steps": 43}
        assert hp.info(32)["steps"] == 11
        bowe_hopwood.CRH.evaluate_batch(B, _msgs(1, 64, 3))  # a tree node: 21 of the 70 groups + the 3-chunk remainder
        assert hb.wait_for_wide_table(64) is not None
        assert hb.info(32) == {"digit_bits_or_group": 8, "signed_subset": False, "table_bytes": (567 * 4 + (21 << 23) + 512) * 128, "steps": 11}
        assert hb.info(64)["steps"] == 22  # 171 = 21 * 8 + 3 chunks
        assert hb.info(70)["digit_bits_or_group"] == 5  # 187 chunks = 23 groups: not covered yet -- such a call would run on the cache-sized table


def test_tables_grow_with_the_message_lengths(cpa):
    """the wide table covers the digits / chunk groups the messages so far needed (te_ensure_units): every length against the
    oracle while the table grows underneath, `table_bytes` never shrinks, and a resident tree that pinned the handle keeps
    working after a longer message replaced the table it was built with"""
    from crypto_primitives_amd.crh import pedersen, bowe_hopwood
    from crypto_primitives_amd import merkle_tree as mt
    g = jj.pedersen_generators(0x61, 8, 40)   # 320 generators: 14 digits of 24 bits
    gb = jj.bowe_hopwood_generators(0x62, 20, 12)  # 240 chunks: 30 groups of 8
    P, B = pedersen.Parameters(gens_array(g), table_shape=24), bowe_hopwood.Parameters(gens_array(gb), table_shape=8)
    seen_p = seen_b = 0
    for L in (3, 4, 11, 12, 7, 25, 40):
        m = _msgs(3, L, 300 + L)
        dp = pedersen.CRH.evaluate_batch(P, m)
        assert tuple(ints(dp[1])) == opd.evaluate(g, 8, 40, bytes(m[1])), L
        tb = P.handle().info()["table_bytes"]
        assert tb >= seen_p
        seen_p = tb
    assert seen_p == ((14 << 23) + 15) * 128
    for L in (2, 9, 10, 33, 20, 64, 90):
        m = _msgs(3, L, 400 + L)
        db = bowe_hopwood.CRH.evaluate_batch(B, m)
        assert ints(db[2])[0] == obh.evaluate(gb, 20, 12, bytes(m[2])), L
        tb = B.handle().info()["table_bytes"]
        assert tb >= seen_b
        seen_b = tb
    # a RESIDENT byte tree over 32-byte leaves pins the handle; a 90-byte hash on the same handle replaces the table it was built
    # with; updates and proofs of the old tree then run on the new table and must agree with a tree built afterwards
    full_table = B.handle().table_info()["table_id"]
    B2 = bowe_hopwood.Parameters(gens_array(gb), table_shape=8)
    assert B2.handle().table_info()["table_id"] == full_table  # same generators, same shape: the handle attaches to B's table (round 5)
    del B2, B  # ... which goes with its last handle, so that the next one starts from nothing
    B2 = bowe_hopwood.Parameters(gens_array(gb), table_shape=8)
    assert B2.handle().table_info()["wide_builds"] == 0
    leaves = _msgs(64, 32, 77)
    tree = cpa.GpuMerkleTree.new(cpa.BoweHopwoodByteConfig, B2, B2, leaves)
    before = B2.handle().info()["table_bytes"]
    bowe_hopwood.CRH.evaluate_batch(B2, _msgs(2, 90, 5))
    assert B2.handle().info()["table_bytes"] > before
    new_leaf = _msgs(1, 32, 78)
    tree.update_batch([5], new_leaf)
    leaves[5] = new_leaf[0]
    fresh = cpa.GpuMerkleTree.new(cpa.BoweHopwoodByteConfig, B2, B2, leaves)
    assert np.array_equal(np.asarray(tree.root()), np.asarray(fresh.root()))
    pa, pb = tree.generate_proof(9), fresh.generate_proof(9)
    assert np.array_equal(np.asarray(pa.auth_path), np.asarray(pb.auth_path)) and pa.verify(B2, B2, tree.root(), leaves[9])


@pytest.mark.parametrize("W,N", [(63, 9), (40, 14), (63, 13), (30, 19)])
def test_bowe_hopwood_compress_zero_tail_constant(cpa, W, N):
    """TwoToOneCRH::compress of two 32-byte digests in a (W*N)/8-byte buffer: the chunks that lie wholly in the zero
    padding are replaced by one constant table entry (a zero chunk adds +g, crh/bowe_hopwood/mod.rs:167).  Against the oracle
    (which walks every chunk), with the shortcut on and off, through the latency kernel (n <= 2^14) and the table kernels."""
    import os
    from crypto_primitives_amd.crh import bowe_hopwood
    from crypto_primitives_amd import field
    g = jj.bowe_hopwood_generators(300 + W, W, N)
    B = bowe_hopwood.Parameters(gens_array(g))
    assert (W * N) // 8 > 64  # there is a padded tail
    rng = np.random.default_rng(W * 100 + N)
    for n in (5, 20000):
        l = field.random_fr(n, seed=W + n).reshape(n, 1, 4)
        r = field.random_fr(n, seed=N + n).reshape(n, 1, 4)
        got = bowe_hopwood.TwoToOneCRH.compress_batch(B, l, r)
        pick = rng.integers(0, n, size=4)
        for i in pick:
            li, ri = field.to_ints(l[i])[0], field.to_ints(r[i])[0]
            assert ints(got[i])[0] == obh.two_to_one_compress(g, W, N, li, ri), (W, N, n, int(i))


def test_host_batches_larger_than_one_chunk(cpa, ped, bhp):
    """akp_te_crh_batch cuts large host batches into double-buffered chunks (copy-in / kernels / copy-out overlap): three
    chunks and a ragged tail, sampled against the C oracle, whole batch against the device-pointer path"""
    import torch
    from crypto_primitives_amd.crh import pedersen, bowe_hopwood
    n = (1 << 17) * 3 + 777
    for (P, g, Cc), L, crh, fe in ((ped, 128, pedersen.CRH, 2), (bhp, 45, bowe_hopwood.CRH, 1)):
        m = np.random.default_rng(4242 + L).integers(0, 256, size=(n, L), dtype=np.uint8)
        got = crh.evaluate_batch(P, m)
        samp = np.unique(np.concatenate([np.arange(0, 40), np.arange((1 << 17) - 20, (1 << 17) + 20), np.arange(2 * (1 << 17) - 20, 2 * (1 << 17) + 20),
                                         np.arange(n - 40, n)]))
        ms = np.ascontiguousarray(m[samp])
        exp = Cc.pedersen_crh_batch(ms, len(samp), L, threads=8) if fe == 2 else Cc.bh_crh_batch(ms, len(samp), L, threads=8)
        assert np.array_equal(np.asarray(got)[samp].reshape(len(samp), -1), np.asarray(exp).reshape(len(samp), -1))
        d_m = torch.from_numpy(m).cuda()
        d_o = torch.empty((n, fe * 4), dtype=torch.int64, device="cuda")
        cpa._lib.check(cpa.lib.akp_te_crh_batch_dev(P.handle().h, d_m.data_ptr(), n, L, d_o.data_ptr(), torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        assert np.array_equal(d_o.cpu().numpy().view(np.uint64).reshape(n, -1), np.asarray(got).reshape(n, -1))


@pytest.mark.parametrize("half", [0, 1, 2, 10, 31, 35, 40])
def test_bowe_hopwood_two_to_one_short_halves(cpa, bhp, half):
    """TwoToOneCRH::evaluate with halves shorter than half the buffer: the zero padding past left || right is a constant
    (all-zero chunks add +g each); 1- and 2-byte data goes through the 4-byte padding of the message loads; 40 + 40 bytes
    are zip-truncated to the 70-byte buffer (no padding at all)"""
    from crypto_primitives_amd.crh import bowe_hopwood
    B, gb, _ = bhp
    n = 6
    l, r = _msgs(n, half, 900 + half), _msgs(n, half, 901 + half)
    got = bowe_hopwood.TwoToOneCRH.evaluate_batch(B, l, r)
    for i in range(n):
        assert ints(got[i])[0] == obh.two_to_one_evaluate(gb, 63, 9, bytes(l[i]), bytes(r[i])), (half, i)


def test_pedersen_compressor_injective_map(cpa, ped):
    """crh/injective_map/mod.rs:16-108: PedersenCRHCompressor / PedersenTwoToOneCRHCompressor with TECompressor = x of the
    Pedersen hash; compress = evaluate on the 32-byte LE serialisations of the two Fq digests (zero padding behind them).
    Against the python oracle, the (x, y) kernels, and across the latency / table kernel switch."""
    from crypto_primitives_amd.crh import pedersen, injective_map as inj
    from crypto_primitives_amd import field
    P, g, Cc = ped
    X = inj.Parameters(gens_array(g))
    m = _msgs(6, 100, 77)
    got = inj.PedersenCRHCompressor.evaluate_batch(X, m)
    for i in range(6):
        assert ints(got[i])[0] == opd.compressor_evaluate(g, 4, 256, bytes(m[i]))
    assert np.array_equal(got, inj.TECompressor.injective_map(pedersen.CRH.evaluate_batch(P, m)))
    l, r = _msgs(5, 30, 78), _msgs(5, 30, 79)
    got2 = inj.PedersenTwoToOneCRHCompressor.evaluate_batch(X, l, r)
    for i in range(5):
        assert ints(got2[i])[0] == opd.compressor_two_to_one_evaluate(g, 4, 256, bytes(l[i]), bytes(r[i]))
    dl, dr = got[:3], got[3:6]
    cmp_ = inj.PedersenTwoToOneCRHCompressor.compress_batch(X, dl, dr)
    for i in range(3):
        assert ints(cmp_[i])[0] == opd.compressor_two_to_one_compress(g, 4, 256, ints(dl[i])[0], ints(dr[i])[0])
    # table kernels (n > 2^14) against the C oracle: x of the Pedersen digest
    n = 20000
    mm = np.random.default_rng(5).integers(0, 256, size=(n, 128), dtype=np.uint8)
    big = inj.PedersenCRHCompressor.evaluate_batch(X, mm)
    samp = np.arange(0, n, 997)
    exp = np.asarray(Cc.pedersen_crh_batch(np.ascontiguousarray(mm[samp]), len(samp), 128, threads=8)).reshape(len(samp), 2, 4)[:, 0]
    assert np.array_equal(big[samp], exp)
    # 64 bytes of data in the 128-byte buffer: half the table steps
    d = X.handle().info()["digit_bits_or_group"]
    assert X.handle().info(64)["steps"] == -(-512 // d) and X.handle().info(128)["steps"] == -(-1024 // d)


def test_budget_chosen_shape_narrows_when_memory_is_taken_after_creation(cpa):
    """a handle whose shape came from the table budget is created while the device is empty (24-bit digits / groups of eight), then
    most of the memory is taken by someone else before the first hash: the table that is about to be built no longer fits half of
    what is free, so the handle narrows its shape (te_narrow: constants and remainder tables follow) instead of failing -- same
    digests.  An explicit shape does not narrow: it fails with the table size in the message."""
    import torch
    from crypto_primitives_amd import params as cparams
    from crypto_primitives_amd.crh import pedersen, bowe_hopwood
    from crypto_primitives_amd._lib import TABLE_BUDGET_DEVICE
    ctx = cpa.default_context(0)
    ctx.set_table_budget(TABLE_BUDGET_DEVICE)  # the budget that follows the device's memory (opt-in since round 5)
    try:
        if ctx.table_budget() < 71 << 30:
            pytest.skip("needs an idle 288 GB device")
        pg, bg = cparams.pedersen_generators(0xB5B50014, 4, 256), cparams.bowe_hopwood_generators(0xB5B50015, 63, 9)
        P, B, PX = pedersen.Parameters(pg), bowe_hopwood.Parameters(bg), pedersen.Parameters(pg, table_shape=24)
        hp, hb, hx = P.handle(), B.handle(), PX.handle()
    finally:
        ctx.set_table_budget(0)
    m, mb = _msgs(40, 128, 21), _msgs(40, 64, 22)
    want_p = pedersen.CRH.evaluate_batch(pedersen.Parameters(pg, table_shape=12), m)
    want_b = bowe_hopwood.CRH.evaluate_batch(bowe_hopwood.Parameters(bg, table_shape=3), mb)
    assert hp.table_info()["table_id"] != hx.table_info()["table_id"]  # budget-chosen and explicit shapes are filed apart (only the former may narrow)
    assert hp.table_info()["last_build"]["upgrade_state"] == 1 and hb.table_info()["last_build"]["upgrade_state"] == 1  # wide tables wanted, nothing built
    free, _ = torch.cuda.mem_get_info(0)
    hog = torch.empty(free - (40 << 30), dtype=torch.uint8, device="cuda:0")  # leaves ~40 GB: half of it is below 46 GB and below 22.5 GB
    try:
        hp.prepare(128)  # the wide table on this thread (a hash would run on the cache-sized table and leave the same work to the builder)
        assert np.array_equal(pedersen.CRH.evaluate_batch(P, m), want_p)
        d = hp.info(128)
        assert d["digit_bits_or_group"] < 24 and d["steps"] == -(-1024 // d["digit_bits_or_group"]) and d["table_bytes"] < 24 << 30, d
        hb.prepare(64)
        assert np.array_equal(bowe_hopwood.CRH.evaluate_batch(B, mb), want_b)
        g = hb.info(64)
        assert g["digit_bits_or_group"] < 8 and g["table_bytes"] < 12 << 30, g
        with pytest.raises(cpa.AkpError) as err:
            pedersen.CRH.evaluate_batch(PX, m)
        assert "curve table of" in str(err.value) and "MB" in str(err.value)
    finally:
        del hog
        torch.cuda.empty_cache()
    assert np.array_equal(pedersen.CRH.evaluate_batch(PX, m), want_p)  # with the memory back the explicit shape builds
    assert hx.info(128)["digit_bits_or_group"] == 24
