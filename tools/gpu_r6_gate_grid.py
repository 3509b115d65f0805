#!/usr/bin/env python3
"""round 6: the gated launch of the pinned curve-hash path against the chunked launches, with K extra streams created before the
context's copy streams: with the copy-in stream at normal priority some K put its hardware queue on the hash kernel's queue or on that
queue's pipe of the command processor (6.5 / 14 ms per 2^20 Pedersen hashes instead of 3.9: profiles/r06_s41 ... s45).
argv: K [hbm]"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import crypto_primitives_amd as cpa  # noqa: E402
from crypto_primitives_amd import params as cparams  # noqa: E402
from crypto_primitives_amd.crh import pedersen, bowe_hopwood  # noqa: E402

lib, check = cpa.lib, cpa._lib.check
K = int(sys.argv[1]) if len(sys.argv) > 1 else 0
hbm = len(sys.argv) > 2 and sys.argv[2] == "hbm"
n = 1 << 20
ctx = cpa.default_context(0)
extra = [torch.cuda.Stream(device=0) for _ in range(K)]  # kept alive: K more normal-priority streams before the context's pipes exist
for st in extra:
    with torch.cuda.stream(st):
        torch.zeros(16, device="cuda:0").add_(1)
torch.cuda.synchronize()
ctx.set_table_budget(cpa._lib.TABLE_BUDGET_DEVICE if hbm else 0)
out = {"extra_streams": K, "table": "hbm_sized" if hbm else "cache_sized", "statistic": "median ms of 11 pinned calls of 2^20 hashes after 2 warm-up calls"}


def calls(fn, reps=11):
    for _ in range(2):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    return round(ts[len(ts) // 2], 3)


# (the arms of profiles/r06_s42 ... s45 -- a resident grid with tickets / fixed strides, wave priorities, the copy-in stream's priority --
# were switches of the test build while they were compared: profiles/r06_s45/resident_grid_arms.patch; what is left is the library's gated
# launch against the chunked launches)
ARMS = (("library", {}), ("gated", {"AKP_TE_PINNED_FORM": "gated"}), ("chunked", {"AKP_TE_PINNED_FORM": "chunked"}))
if os.environ.get("GATE_GRID_REVERSE") == "1":  # the same arms last-first: is a slow arm slow, or was it early?
    ARMS = ARMS[::-1]
cases = (("pedersen_4x256_128B", pedersen.Parameters(cparams.pedersen_generators(0xA5A50004, 4, 256)), 128, 2),
         ("bowe_hopwood_63x9_64B", bowe_hopwood.Parameters(cparams.bowe_hopwood_generators(0xA5A50005, 63, 9)), 64, 1))
for name, prm, L, fe in cases:
    h = prm.handle(ctx)
    h.prepare(L)
    msgs = np.random.default_rng(7).integers(0, 256, size=(n, L), dtype=np.uint8)
    ref = np.empty((n, 4 * fe), np.uint64)
    check(lib.akp_te_crh_batch(h.h, msgs.ctypes.data, n, L, ref.ctypes.data))  # pageable call (creates the copy streams): the reference digests
    pm, po = C.c_void_p(), C.c_void_p()
    check(lib.akp_host_alloc(msgs.nbytes, C.byref(pm)))
    check(lib.akp_host_alloc(ref.nbytes, C.byref(po)))
    np.ctypeslib.as_array((C.c_uint8 * msgs.size).from_address(pm.value))[:] = msgs.reshape(-1)
    pout = np.ctypeslib.as_array((C.c_uint64 * ref.size).from_address(po.value)).reshape(ref.shape)
    rec = {}
    for arm, env in ARMS:
        for k in ("AKP_TE_PINNED_FORM",):
            os.environ.pop(k, None)
        os.environ.update(env)
        pout[:] = 0
        rec[arm] = calls(lambda: check(lib.akp_te_crh_batch(h.h, pm, n, L, po)))
        if not np.array_equal(pout, ref):
            rec[arm + "_WRONG"] = True
    out[name] = rec
    check(lib.akp_host_free(pm))
    check(lib.akp_host_free(po))
print(json.dumps(out))
