"""Curve tables are shared per DEVICE (round 5): handles created with the same generators, window and table shape on different
contexts of one device attach to ONE set of precomputed tables -- the analogue of the reference's `Parameters: Sync`
(crh/mod.rs:22; one value borrowed by every rayon worker, merkle_tree/mod.rs:417,458,494).  Every digest against the oracle."""
import ctypes as C
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import jubjub as jj, fr as ofr, cref  # noqa: E402
from helpers import gens_array  # noqa: E402


@pytest.fixture(scope="module")
def cpa():
    import crypto_primitives_amd as m
    assert m.lib.akp_device_count() >= 1
    return m


def _msgs(n, L, seed):
    return np.frombuffer(ofr.SplitMix64(seed).bytes(max(n * L, 1)), dtype=np.uint8)[: n * L].reshape(n, L).copy()


def _free_bytes():
    import torch
    return torch.cuda.mem_get_info(0)[0]


def _run_threads(n, fn):
    errs, out = [], [None] * n

    def body(i):
        try:
            out[i] = fn(i)
        except BaseException as e:  # noqa: BLE001 -- reported below, in the test's thread
            errs.append((i, repr(e)))
    ts = [threading.Thread(target=body, args=(i,)) for i in range(n)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    return out


@pytest.mark.parametrize("budget_name", ["default", "device"])
def test_eight_threads_own_contexts_same_generators_share_one_table(cpa, budget_name):
    """8 threads x own context x the same 4x256 Pedersen generators: one table id, built once, device memory grows by ONE table
    (with the device-sized budget eight tables of 46 GB would not even fit), every thread's digests equal the oracle's"""
    from crypto_primitives_amd._lib import Context, TABLE_BUDGET_DEVICE
    from crypto_primitives_amd.crh import pedersen
    g = gens_array(jj.pedersen_generators(0xC5C50001 + (budget_name == "device"), 4, 256))
    ora = cref.CurveParams(4, 256, g)
    budget = TABLE_BUDGET_DEVICE if budget_name == "device" else 0
    n_thr, n_msg = 8, 3000
    ctxs = [Context(0) for _ in range(n_thr)]
    for c in ctxs:
        c.set_table_budget(budget)
    if budget_name == "device" and ctxs[0].table_budget() < 71 << 30:
        pytest.skip("needs an idle 288 GB device")
    P = pedersen.Parameters(g)
    # one throw-away hash first: the runtime's own first-use allocations (code objects, queues, signal pools: ~300 MB on this stack) must
    # not be counted as the table's
    warm = pedersen.Parameters(gens_array(jj.pedersen_generators(0xC5C500FF, 4, 8)))
    pedersen.CRH.evaluate_batch(warm, _msgs(20000, 4, 1))
    del warm
    free0 = _free_bytes()
    start = threading.Barrier(n_thr)

    def work(i):
        h = P.handle(ctxs[i])
        start.wait()  # all first hashes at the same moment: one thread builds, seven wait on the table's lock and find it built
        k = n_msg + 17000 * (i % 2)  # odd threads: the accumulate + finalize kernels; even threads: the split kernel of small batches
        m = _msgs(k, 128, 100 + i)
        out = np.empty((k, 2, 4), dtype=np.uint64)
        cpa._lib.check(cpa.lib.akp_te_crh_batch(h.h, m.ctypes.data, k, 128, out.ctypes.data))
        assert np.array_equal(out, ora.pedersen_crh_batch(m, k, 128, threads=2)), "thread %d: digests differ from the oracle" % i
        return h.table_info(), h.info(128)
    res = _run_threads(n_thr, work)
    ids = {r[0]["table_id"] for r in res}
    assert len(ids) == 1, ids
    if budget_name == "device":
        # round 6: the eight first hashes ran on the (shared) cache-sized table while ONE background thread built the (shared) wide one
        assert all(r[1]["digit_bits_or_group"] in (16, 24) for r in res)
        assert P.handle(ctxs[0]).wait_for_wide_table(128) is not None
    ti = P.handle(ctxs[0]).table_info()
    assert ti["handles_attached"] == n_thr and ti["wide_builds"] == 1, ti
    if budget_name == "device":
        assert ti["last_build"]["in_background"] == 1 and ti["last_build"]["upgrade_state"] == 2 and ti["last_build"]["combine_ms"] > 0, ti
    infos = [P.handle(c).info(128) for c in ctxs]
    table_bytes = infos[0]["table_bytes"]
    want_d = 24 if budget_name == "device" else 16
    assert infos[0]["digit_bits_or_group"] == want_d and all(i == infos[0] for i in infos)
    used = free0 - _free_bytes()
    # one table (+ the 268 MB cache-sized one the handles started on) + eight contexts' scratch and the runtime's per-queue allocations
    # (a few hundred MB in all) -- not eight tables
    assert table_bytes <= used < table_bytes + (3 << 29) and used < 2 * table_bytes + (1 << 30), (used, table_bytes)
    # the creator goes first -- handle AND context -- and the others keep hashing with the table it built
    del res
    P._handles.pop((id(ctxs[0]), P._KIND, 0))
    ctxs[0].close()
    m = _msgs(500, 128, 7)
    out = np.empty((500, 2, 4), dtype=np.uint64)
    cpa._lib.check(cpa.lib.akp_te_crh_batch(P.handle(ctxs[5]).h, m.ctypes.data, 500, 128, out.ctypes.data))
    assert np.array_equal(out, ora.pedersen_crh_batch(m, 500, 128, threads=2))
    assert P.handle(ctxs[5]).table_info()["handles_attached"] == n_thr - 1
    # ... and the memory comes back with the LAST handle
    P._handles.clear()
    for c in ctxs[1:]:
        c.close()
    # (the runtime keeps first-use allocations of its own per queue -- kernel scratch, signal pools: a few hundred MB here -- so the
    # physical check is meaningful for the 46 GB table only; the logical one holds for both: a new handle starts from nothing)
    if budget_name == "device":
        assert free0 - _free_bytes() < 2 << 30, "the table outlived its last handle"
    c9 = Context(0)
    c9.set_table_budget(budget)
    h9 = pedersen.Parameters(g).handle(c9)
    ti9 = h9.table_info()
    assert ti9["handles_attached"] == 1 and ti9["wide_builds"] == 0 and ti9["last_build"]["upgrade_state"] == (1 if budget_name == "device" else 0), ti9
    assert h9.info(128)["table_bytes"] < 1 << 20


def test_table_extends_under_concurrent_hashing(cpa):
    """threads with contexts of their own hash messages of DIFFERENT lengths through one shared Bowe-Hopwood table while it is
    extended underneath them (short messages build a prefix of the group table, longer ones release it and build more, new
    lengths add remainder tables): the launches and the rebuilds are serialised by the table's lock, every digest is the oracle's"""
    from crypto_primitives_amd._lib import Context
    from crypto_primitives_amd.crh import bowe_hopwood
    g = gens_array(jj.bowe_hopwood_generators(0xC5C50003, 63, 9))
    ora = cref.CurveParams(63, 9, g)
    lens = [8, 32, 64, 100, 150, 212, 20, 70]
    ctxs = [Context(0) for _ in lens]
    B = bowe_hopwood.Parameters(g)
    start = threading.Barrier(len(lens))

    def work(i):
        h = B.handle(ctxs[i])
        L = lens[i]
        start.wait()
        for rep in range(6):
            n = 700 + 50 * rep
            m = _msgs(n, L, 1000 + 10 * i + rep)
            out = np.empty((n, 4), dtype=np.uint64)
            cpa._lib.check(cpa.lib.akp_te_crh_batch(h.h, m.ctypes.data, n, L, out.ctypes.data))
            assert np.array_equal(out, ora.bh_crh_batch(m, n, L, threads=1)), "length %d, repetition %d" % (L, rep)
        return h.table_info()
    res = _run_threads(len(lens), work)
    assert len({r["table_id"] for r in res}) == 1
    info = B.handle(ctxs[0]).info(212)
    assert info["digit_bits_or_group"] == 5 and B.handle(ctxs[0]).table_info()["wide_builds"] >= 1


def test_pedersen_flavours_and_threads_of_a_tree_share(cpa):
    """pedersen::CRH (x || y) and the TECompressor flavour (x only) hash with the SAME table (the digest width is the handle's, the
    table the generators'); different generators, windows or shapes do not share"""
    from crypto_primitives_amd._lib import Context, TE_PEDERSEN
    from crypto_primitives_amd.crh import pedersen
    ga = gens_array(jj.pedersen_generators(0xC5C50005, 4, 64))
    gb = gens_array(jj.pedersen_generators(0xC5C50006, 4, 64))
    c1, c2 = Context(0), Context(0)
    A, A2, Bp, As = pedersen.Parameters(ga), pedersen.Parameters(ga.copy()), pedersen.Parameters(gb), pedersen.Parameters(ga, table_shape=12)
    TE_PEDERSEN_X = 2
    h_xy, h_x = A.handle(c1, kind=TE_PEDERSEN), A.handle(c2, kind=TE_PEDERSEN_X)
    assert h_xy.table_info()["table_id"] == h_x.table_info()["table_id"] == A2.handle(c2).table_info()["table_id"]
    assert h_xy.table_info()["handles_attached"] == 3
    assert Bp.handle(c1).table_info()["table_id"] != h_xy.table_info()["table_id"]
    assert As.handle(c1).table_info()["table_id"] != h_xy.table_info()["table_id"]
    m = _msgs(300, 32, 3)
    oxy, ox = np.empty((300, 2, 4), dtype=np.uint64), np.empty((300, 4), dtype=np.uint64)
    cpa._lib.check(cpa.lib.akp_te_crh_batch(h_xy.h, m.ctypes.data, 300, 32, oxy.ctypes.data))
    cpa._lib.check(cpa.lib.akp_te_crh_batch(h_x.h, m.ctypes.data, 300, 32, ox.ctypes.data))
    want = cref.CurveParams(4, 64, ga).pedersen_crh_batch(m, 300, 32, threads=2)
    assert np.array_equal(oxy, want) and np.array_equal(ox, want[:, 0, :])
    assert h_xy.table_info()["wide_builds"] == 1  # the second flavour found the table built


def test_prepare_builds_at_a_time_of_the_hosts_choosing_and_dev_calls_then_only_enqueue(cpa):
    """akp_te_params_prepare / _prepare_compress: the table work happens in the call the host names; the `_dev` calls that follow
    find everything built (wide_builds does not move) -- also inside a stream capture, where an UNPREPARED handle refuses cleanly
    (AKP_ERR_BAD_PARAMS naming akp_te_params_prepare) instead of allocating and draining the device under the capture"""
    import torch
    from crypto_primitives_amd._lib import Context, AKP_ERR_BAD_PARAMS
    from crypto_primitives_amd.crh import bowe_hopwood
    g = gens_array(jj.bowe_hopwood_generators(0xC5C50007, 63, 9))
    ora = cref.CurveParams(63, 9, g)
    ctx = Context(0)
    B, B2 = bowe_hopwood.Parameters(g), bowe_hopwood.Parameters(g, table_shape=4)
    h, h2 = B.handle(ctx), B2.handle(ctx)
    assert h.table_info()["wide_builds"] == 0 and h.info(32)["table_bytes"] == 567 * 4 * 128
    h.prepare(32, compress=True)  # a tree's leaf length and its inner nodes
    ti = h.table_info()
    assert ti["wide_builds"] == 1 and h.info(32)["table_bytes"] > 567 * 4 * 128
    n = 20000
    dev = torch.device("cuda", 0)
    m = _msgs(n, 32, 9)
    d_m = torch.from_numpy(m).to(dev)
    d_out = torch.zeros((n, 4), dtype=torch.int64, device=dev)
    d_tmp = torch.zeros((n, 4), dtype=torch.int64, device=dev)
    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        cpa._lib.check(cpa.lib.akp_te_crh_batch_dev(h.h, d_m.data_ptr(), n, 32, d_out.data_ptr(), side.cuda_stream))  # sizes the context's scratch
        cpa._lib.check(cpa.lib.akp_te_crh_batch_dev(h2.h, d_m.data_ptr(), 16, 8, d_tmp.data_ptr(), side.cuda_stream))  # (h2: 8-byte messages only)
    side.synchronize()
    assert h.table_info()["wide_builds"] == 1  # the hash found its table
    want = ora.bh_crh_batch(m, n, 32, threads=4)
    assert np.array_equal(d_out.cpu().numpy().view(np.uint64), want)
    # the same launch captured into a graph and replayed on new messages
    graph = torch.cuda.CUDAGraph()
    d_out.zero_()
    with torch.cuda.graph(graph, stream=side):
        rc = cpa.lib.akp_te_crh_batch_dev(h.h, d_m.data_ptr(), n, 32, d_out.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
        # an unprepared shape under capture: a clean refusal, the capture stays valid
        rc2 = cpa.lib.akp_te_crh_batch_dev(h2.h, d_m.data_ptr(), n, 32, d_tmp.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
        msg2 = cpa.lib.akp_last_error().decode()
    assert rc == 0 and rc2 == AKP_ERR_BAD_PARAMS and "akp_te_params_prepare" in msg2, (rc, rc2, msg2)
    m2 = _msgs(n, 32, 10)
    d_m.copy_(torch.from_numpy(m2))
    graph.replay()
    torch.cuda.synchronize(dev)
    assert np.array_equal(d_out.cpu().numpy().view(np.uint64), ora.bh_crh_batch(m2, n, 32, threads=4))
    # ANOTHER handle of the same generators (a context of its own) now hashes a longer message -- the shared table is
    # extended underneath the captured graph.  Round 5 freed the table the graph had baked in; round 6 retires it: the replay still reads
    # valid memory and returns the right digests, and new launches use the extended table.
    ctx_b = Context(0)
    hb2 = bowe_hopwood.Parameters(g).handle(ctx_b)
    assert hb2.table_info()["table_id"] == h.table_info()["table_id"]
    before = h.info(32)["table_bytes"]
    long_m = _msgs(300, 212, 11)
    got_long = np.empty((300, 4), dtype=np.uint64)
    cpa._lib.check(cpa.lib.akp_te_crh_batch(hb2.h, long_m.ctypes.data, 300, 212, got_long.ctypes.data))
    assert np.array_equal(got_long, ora.bh_crh_batch(long_m, 300, 212, threads=4))
    assert h.table_info()["wide_builds"] == 2 and h.info(32)["table_bytes"] > before  # extended: the complete table now
    hog = torch.empty(64 << 20, dtype=torch.uint8, device=dev).fill_(0xA5)  # (fresh allocations must not land on the retired table: it is still owned)
    m3 = _msgs(n, 32, 12)
    d_m.copy_(torch.from_numpy(m3))
    graph.replay()
    torch.cuda.synchronize(dev)
    assert np.array_equal(d_out.cpu().numpy().view(np.uint64), ora.bh_crh_batch(m3, n, 32, threads=4)), "the graph captured before the extension no longer hashes"
    del hog
    # oversized lengths are the reference's panic, from prepare as from evaluate
    with pytest.raises(cpa.IncorrectInputLength):
        h.prepare(213)
    assert cpa.lib.akp_te_params_prepare(None, 8) == AKP_ERR_BAD_PARAMS and cpa.lib.akp_te_params_prepare_compress(None) == AKP_ERR_BAD_PARAMS


def test_budget_chosen_wide_table_is_built_in_the_background(cpa):
    """round 6: a handle created under AKP_TABLE_BUDGET_DEVICE starts hashing on the cache-sized table at once -- the first
    batch costs what it costs with the default budget -- while a thread of the library allocates and builds the 46 GB table; calls switch
    to it when it is complete; every digest along the way is the oracle's.  A second handle attaches to both tables; `prepare` from
    another thread while the build runs simply waits for it; destroying the only handle of a table that is still being built waits too."""
    import time
    from crypto_primitives_amd._lib import Context, TABLE_BUDGET_DEVICE
    from crypto_primitives_amd.crh import pedersen
    g = gens_array(jj.pedersen_generators(0xC5C50021, 4, 256))
    ora = cref.CurveParams(4, 256, g)
    c1, c2 = Context(0), Context(0)
    c2.set_table_budget(TABLE_BUDGET_DEVICE)
    if c2.table_budget() < 71 << 30:
        pytest.skip("needs an idle 288 GB device")
    n = 30000
    m = _msgs(n, 128, 31)
    want = ora.pedersen_crh_batch(m, n, 128, threads=8)
    # (a default-budget handle of the same generators first: the wide handle below then finds the cache-sized table built -- its first call
    # is a plain hash -- and this call has paid the context's scratch allocation)
    base = pedersen.Parameters(g)
    out = np.empty((n, 2, 4), dtype=np.uint64)
    cpa._lib.check(cpa.lib.akp_te_crh_batch(base.handle(c1).h, m.ctypes.data, n, 128, out.ctypes.data))
    t0 = time.perf_counter()
    cpa._lib.check(cpa.lib.akp_te_crh_batch(base.handle(c1).h, m.ctypes.data, n, 128, out.ctypes.data))
    plain_ms = (time.perf_counter() - t0) * 1e3
    P = pedersen.Parameters(g)
    c1_handles_before = base.handle(c1).table_info()["handles_attached"]
    c1.set_table_budget(TABLE_BUDGET_DEVICE)
    h = P.handle(c1)
    assert base.handle(c1).table_info()["handles_attached"] == c1_handles_before + 1  # attached to the cache-sized table as well
    ti = h.table_info()
    assert ti["last_build"]["upgrade_state"] == 1 and ti["wide_builds"] == 0 and h.info(128)["digit_bits_or_group"] == 16
    t0 = time.perf_counter()
    cpa._lib.check(cpa.lib.akp_te_crh_batch(h.h, m.ctypes.data, n, 128, out.ctypes.data))
    first_ms = (time.perf_counter() - t0) * 1e3
    assert np.array_equal(out, want)
    assert first_ms < plain_ms + 25.0, (first_ms, plain_ms)  # NOT the 65 ms (1.3 s on a machine's first use) of building 46 GB on this thread
    # a second context's handle while the build runs: same tables; its `prepare` waits for the builder instead of building a second time
    h2 = P.handle(c2)
    assert h2.table_info()["table_id"] == h.table_info()["table_id"]
    seen = set()
    deadline = time.perf_counter() + 60
    while time.perf_counter() < deadline:
        out[:] = 0
        cpa._lib.check(cpa.lib.akp_te_crh_batch(h.h, m.ctypes.data, n, 128, out.ctypes.data))
        assert np.array_equal(out, want)
        st = h.table_info()["last_build"]["upgrade_state"]
        seen.add((st, h.info(128)["digit_bits_or_group"]))
        if st == 2:
            break
    assert (2, 24) in seen and all(s in ((1, 16), (2, 24)) for s in seen), seen
    h2.prepare(128)
    ti = h2.table_info()
    assert ti["wide_builds"] == 1 and ti["last_build"]["in_background"] == 1 and ti["last_build"]["units_to"] == 43 and ti["last_build"]["combine_ms"] > 0, ti
    cpa._lib.check(cpa.lib.akp_te_crh_batch(h2.h, m.ctypes.data, n, 128, out.ctypes.data))
    assert np.array_equal(out, want) and h2.info(128)["digit_bits_or_group"] == 24
    # a table whose only handle goes while the builder is still at work: the release waits for the thread (no use after free, no leak)
    g2 = gens_array(jj.pedersen_generators(0xC5C50022, 4, 256))
    Q = pedersen.Parameters(g2)
    hq = Q.handle(c1)
    cpa._lib.check(cpa.lib.akp_te_crh_batch(hq.h, m.ctypes.data, 2000, 128, out.ctypes.data))  # starts the build
    assert np.array_equal(out[:2000], cref.CurveParams(4, 256, g2).pedersen_crh_batch(m[:2000], 2000, 128, threads=4))
    free_before = _free_bytes()
    Q._handles.clear()
    del hq, Q
    assert _free_bytes() >= free_before - (1 << 30)  # whatever the builder had allocated came back with the handle
    c1.set_table_budget(0)
    c2.set_table_budget(0)


def test_background_upgrade_under_concurrent_mixed_lengths(cpa):
    """threads with contexts of their own hash messages of DIFFERENT lengths through handles under AKP_TABLE_BUDGET_DEVICE while the
    library's thread builds the shared wide Bowe-Hopwood table underneath them: the first request builds a prefix, a longer message
    asks again and the table is extended (once, to the complete 75 GB table; the prefix is retired, not freed -- launches that were
    given it may still be running), remainder tables are added per length.  Whichever table a call lands on, every digest is the
    oracle's; in the end every length runs on the wide table and the table was built at most twice."""
    import time
    from crypto_primitives_amd._lib import Context, TABLE_BUDGET_DEVICE
    from crypto_primitives_amd.crh import bowe_hopwood
    g = gens_array(jj.bowe_hopwood_generators(0xC5C50031, 63, 9))
    ora = cref.CurveParams(63, 9, g)
    lens = [8, 32, 64, 100, 150, 212, 20, 70]
    ctxs = [Context(0) for _ in lens]
    for c in ctxs:
        c.set_table_budget(TABLE_BUDGET_DEVICE)
    if ctxs[0].table_budget() < 71 << 30:
        pytest.skip("needs an idle 288 GB device")
    B = bowe_hopwood.Parameters(g)
    handles = [B.handle(c) for c in ctxs]
    for c in ctxs:
        c.set_table_budget(0)
    start = threading.Barrier(len(lens))

    def work(i):
        h, L = handles[i], lens[i]
        start.wait()
        shapes = set()
        t_end = time.perf_counter() + 6.0
        rep = 0
        while rep < 6 or (time.perf_counter() < t_end and 8 not in shapes):
            n = 20000 + 50 * (rep % 7) if i % 2 else 700 + 50 * (rep % 7)  # odd threads: the accumulate + finalize kernels; even: the split kernel
            m = _msgs(n, L, 2000 + 10 * i + rep)
            out = np.empty((n, 4), dtype=np.uint64)
            cpa._lib.check(cpa.lib.akp_te_crh_batch(h.h, m.ctypes.data, n, L, out.ctypes.data))
            assert np.array_equal(out, ora.bh_crh_batch(m, n, L, threads=1)), "length %d, repetition %d" % (L, rep)
            shapes.add(h.info(L)["digit_bits_or_group"])
            rep += 1
        return shapes
    res = _run_threads(len(lens), work)
    assert all(s <= {5, 8} for s in res), res
    for h, L in zip(handles, lens):
        h.prepare(L)  # (whatever the builder had not got round to)
        assert h.info(L)["digit_bits_or_group"] == 8
    ti = handles[0].table_info()
    assert len({h.table_info()["table_id"] for h in handles}) == 1 and ti["handles_attached"] == len(lens)
    assert 1 <= ti["wide_builds"] <= 2 and ti["last_build"]["upgrade_state"] == 2, ti
    for i, (h, L) in enumerate(zip(handles, lens)):  # and once more, now all on the wide table
        m = _msgs(900, L, 5000 + i)
        out = np.empty((900, 4), dtype=np.uint64)
        cpa._lib.check(cpa.lib.akp_te_crh_batch(h.h, m.ctypes.data, 900, L, out.ctypes.data))
        assert np.array_equal(out, ora.bh_crh_batch(m, 900, L, threads=2)), L


def test_wide_handles_created_and_dropped_at_every_stage_of_the_upgrade(cpa):
    """handles under AKP_TABLE_BUDGET_DEVICE created, used for a few batches and destroyed at different stages of the background build
    (before the first call, right after it, mid-build, after the switch), fresh generators each time so that every round builds anew:
    the release of a table waits for its builder, the digests are the oracle's throughout, and the device's memory comes back."""
    import time
    from crypto_primitives_amd._lib import Context, TABLE_BUDGET_DEVICE
    from crypto_primitives_amd.crh import bowe_hopwood
    ctx = Context(0)
    ctx.set_table_budget(TABLE_BUDGET_DEVICE)
    if ctx.table_budget() < 71 << 30:
        pytest.skip("needs an idle 288 GB device")
    n = 20000
    m = _msgs(n, 64, 77)
    warm = bowe_hopwood.Parameters(gens_array(jj.bowe_hopwood_generators(0xC5C50040, 63, 9)), table_shape=5)
    out = np.empty((n, 4), dtype=np.uint64)
    cpa._lib.check(cpa.lib.akp_te_crh_batch(warm.handle(ctx).h, m.ctypes.data, n, 64, out.ctypes.data))  # the context's scratch
    free0 = _free_bytes()
    stages = ["no call", "one call", "mid build", "after switch", "one call", "mid build", "no call", "after switch"]
    for k, stage in enumerate(stages):
        g = gens_array(jj.bowe_hopwood_generators(0xC5C50041 + k, 63, 9))
        want = cref.CurveParams(63, 9, g).bh_crh_batch(m, n, 64, threads=8)
        B = bowe_hopwood.Parameters(g)
        h = B.handle(ctx)
        calls = {"no call": 0, "one call": 1, "mid build": 4, "after switch": 10 ** 6}[stage]
        t_end = time.perf_counter() + 30
        done = 0
        while done < calls and time.perf_counter() < t_end:
            out[:] = 0
            cpa._lib.check(cpa.lib.akp_te_crh_batch(h.h, m.ctypes.data, n, 64, out.ctypes.data))
            assert np.array_equal(out, want), (stage, done)
            done += 1
            if stage == "after switch" and h.table_info()["last_build"]["upgrade_state"] == 2:
                cpa._lib.check(cpa.lib.akp_te_crh_batch(h.h, m.ctypes.data, n, 64, out.ctypes.data))
                assert np.array_equal(out, want) and h.info(64)["digit_bits_or_group"] == 8
                break
        B._handles.clear()
        del h, B  # the table goes with its only handle: a build that is still running is waited for
    assert free0 - _free_bytes() < 1 << 30, "tables of dropped handles were not released"
    ctx.set_table_budget(0)
