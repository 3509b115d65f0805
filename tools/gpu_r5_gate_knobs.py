#!/usr/bin/env python3
"""round 5: where the time of the gated pinned Pedersen call goes.  Test build (AKP_LIB=.../libakp_testhooks.so); 2^20 Pedersen 4x256
hashes of 128 bytes, pinned buffers on both sides, both table sizes.  Arms (environment switches of the test build, read per call):
the call as shipped; without the copy-in (flags only: the kernel's own pace); without the copy-out; without both; a longer pause
between two polls of a waiting workgroup.  With AKP_TE_GATE_STAMPS the kernel also records when every workgroup was released and
when it ended (100 MHz clock): the per-chunk release / completion times are printed for the shipped arm and the no-copy arm."""
import ctypes as C
import json
import os
import sys
import time
os.environ.setdefault("AKP_TE_PINNED_FORM", "gated")  # these arms choose the form themselves (round 6: the library otherwise measures and picks)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401  (shares the HIP runtime)
import crypto_primitives_amd as cpa  # noqa: E402
from crypto_primitives_amd import params as cparams  # noqa: E402
from crypto_primitives_amd.crh import pedersen  # noqa: E402

lib, check = cpa.lib, cpa._lib.check
assert "AKP_LIB" in os.environ, "needs a test build (AKP_LIB=.../libakp_testhooks.so, or an A/B arm built with -DAKP_TEST_HOOKS)"
n, L, fe = 1 << 20, 128, 2
ctx = cpa.default_context(0)
KNOBS = ("AKP_TE_GATE_RAMP", "AKP_TE_GATE_LDS_FLOOR", "AKP_TE_GATE_SKIP_COPY_IN", "AKP_TE_GATE_SKIP_COPY_OUT", "AKP_TE_GATE_POLL_SLEEP", "AKP_TE_GATE_STAMPS", "AKP_TE_PIPE_CHUNK")
F3 = {"AKP_TE_GATE_LDS_FLOOR": "40960"}  # + 4 static bytes: THREE workgroups per CU (what rounds r05_s8 .. s15 ran with)
ARMS = (("shipped", {}),
        ("uniform_chunks", {"AKP_TE_GATE_RAMP": "0"}),
        ("three_wg_per_cu", F3),
        ("three_wg_per_cu_no_copies", dict(F3, AKP_TE_GATE_SKIP_COPY_IN="1", AKP_TE_GATE_SKIP_COPY_OUT="1")),
        ("no_copy_in", {"AKP_TE_GATE_SKIP_COPY_IN": "1"}),
        ("no_copy_out", {"AKP_TE_GATE_SKIP_COPY_OUT": "1"}),
        ("no_copies", {"AKP_TE_GATE_SKIP_COPY_IN": "1", "AKP_TE_GATE_SKIP_COPY_OUT": "1"}),
        ("poll_sleep_4", {"AKP_TE_GATE_POLL_SLEEP": "4"}),
        ("uniform_chunks_again", {"AKP_TE_GATE_RAMP": "0"}),
        ("chunk_2p16", {"AKP_TE_PIPE_CHUNK": str(1 << 16)}),
        ("chunk_2p15", {"AKP_TE_PIPE_CHUNK": str(1 << 15)}),
        ("chunk_2p16_no_copies", {"AKP_TE_PIPE_CHUNK": str(1 << 16), "AKP_TE_GATE_SKIP_COPY_IN": "1", "AKP_TE_GATE_SKIP_COPY_OUT": "1"}),
        ("shipped_again", {}))


if os.environ.get("GATE_KNOBS_ARMS"):  # a subset, by name
    ARMS = tuple(a for a in ARMS if a[0] in os.environ["GATE_KNOBS_ARMS"].split(","))
TIMELINES = not os.environ.get("GATE_KNOBS_ARMS")


def set_env(env):
    for k in KNOBS:
        os.environ.pop(k, None)
    os.environ.update(env)


def calls(fn, reps=15):
    for _ in range(2):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    return {"ms_median": round(ts[len(ts) // 2], 3), "ms_min": round(ts[0], 3), "ms_max": round(ts[-1], 3)}


def chunk_times(path, chunk_wg=256):  # per 2^17 messages (a chunk of the uniform schedule; the ramped one splits the first and last)
    st = np.fromfile(path, dtype=np.uint64).reshape(-1, 2).astype(np.float64)
    t0 = st[:, 0].min()
    rel, end = (st[:, 0] - t0) / 100e3, (st[:, 1] - t0) / 100e3  # ms
    rows = []
    for k in range(len(st) // chunk_wg):
        sl = slice(k * chunk_wg, (k + 1) * chunk_wg)
        rows.append({"chunk": k, "released_ms": [round(float(rel[sl].min()), 3), round(float(rel[sl].max()), 3)],
                     "ended_ms": [round(float(end[sl].min()), 3), round(float(end[sl].max()), 3)],
                     "workgroup_ms_median": round(float(np.median(end[sl] - rel[sl])), 3)})
    return rows


out = {"hashes_per_call": n, "statistic": "wall ms of akp_te_crh_batch, pinned buffers: median / min / max of 15 calls after 2 warm-up calls"}
for table in ("cache_sized", "hbm_sized"):
    ctx.set_table_budget(0 if table == "cache_sized" else cpa._lib.TABLE_BUDGET_DEVICE)
    prm = pedersen.Parameters(cparams.pedersen_generators(0xA5A50004, 4, 256))
    h = prm.handle(ctx)
    h.prepare(L)  # round 6: the wide table (built in the background otherwise) before the measured calls
    msgs = np.random.default_rng(7).integers(0, 256, size=(n, L), dtype=np.uint8)
    ref = np.empty((n, 4 * fe), np.uint64)
    set_env({})
    check(lib.akp_te_crh_batch(h.h, msgs.ctypes.data, n, L, ref.ctypes.data))
    pm, po = C.c_void_p(), C.c_void_p()
    check(lib.akp_host_alloc(msgs.nbytes, C.byref(pm)))
    check(lib.akp_host_alloc(ref.nbytes, C.byref(po)))
    np.ctypeslib.as_array((C.c_uint8 * msgs.size).from_address(pm.value))[:] = msgs.reshape(-1)
    pout = np.ctypeslib.as_array((C.c_uint64 * ref.size).from_address(po.value)).reshape(ref.shape)
    rec = {}
    for arm, env in ARMS:
        set_env(env)
        pout[:] = 0
        r = calls(lambda: check(lib.akp_te_crh_batch(h.h, pm, n, L, po)))
        if "SKIP" not in " ".join(env):
            r["digests_equal_the_pageable_call"] = bool(np.array_equal(pout, ref))
        rec[arm] = r
    for arm, env in (("shipped", {}), ("no_copies", {"AKP_TE_GATE_SKIP_COPY_IN": "1", "AKP_TE_GATE_SKIP_COPY_OUT": "1"})) if TIMELINES else ():
        path = "/tmp/gate_stamps_%s_%s.bin" % (table, arm)
        set_env(dict(env, AKP_TE_GATE_STAMPS=path))
        for _ in range(3):
            check(lib.akp_te_crh_batch(h.h, pm, n, L, po))
        rec["timeline_" + arm] = chunk_times(path)
    set_env({})
    out[table] = rec
    check(lib.akp_host_free(pm))
    check(lib.akp_host_free(po))
    del h, prm
print(json.dumps(out, indent=1))
