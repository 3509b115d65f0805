#!/usr/bin/env python3
"""round 5: the pinned host path of the curve hashes as ONE gated launch (te_crh_gated) against round 4's chunked launches.
A/B inside one process through the test build's AKP_TE_GATED switch (AKP_LIB=.../libakp_testhooks.so), 2^20 Pedersen 4x256 hashes of
128 bytes and 2^20 Bowe-Hopwood 63x9 hashes of 64 bytes, both table sizes; digests compared with the pageable call every time."""
import ctypes as C
import json
import os
import sys
import time
os.environ.setdefault("AKP_TE_PINNED_FORM", "gated")  # these arms choose the form themselves (round 6: the library otherwise measures and picks)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402  (shares the HIP runtime)
import crypto_primitives_amd as cpa  # noqa: E402
from crypto_primitives_amd import params as cparams  # noqa: E402
from crypto_primitives_amd.crh import pedersen, bowe_hopwood  # noqa: E402

lib, check = cpa.lib, cpa._lib.check
assert cpa._lib.LIB_PATH.endswith("libakp_testhooks.so"), "needs the test build (AKP_LIB)"
n = 1 << int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
ctx = cpa.default_context(0)
dev = torch.device("cuda", 0)
out = {"hashes_per_call": n, "statistic": "wall time of akp_te_crh_batch with pinned buffers on both sides: median / min / max of 15 calls after 2 warm-up calls"}


def calls(fn, reps=15):
    for _ in range(2):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    return {"ms_median": ts[len(ts) // 2], "ms_min": ts[0], "ms_max": ts[-1]}


for table in ("cache_sized", "hbm_sized"):
    ctx.set_table_budget(0 if table == "cache_sized" else cpa._lib.TABLE_BUDGET_DEVICE)
    cases = (("pedersen_4x256_128B", pedersen.Parameters(cparams.pedersen_generators(0xA5A50004, 4, 256)), 128, 2),
             ("bowe_hopwood_63x9_64B", bowe_hopwood.Parameters(cparams.bowe_hopwood_generators(0xA5A50005, 63, 9)), 64, 1))
    for name, prm, L, fe in cases:
        h = prm.handle(ctx)
        h.prepare(L)  # round 6: the wide table (built in the background otherwise) before the measured calls
        msgs = np.random.default_rng(7).integers(0, 256, size=(n, L), dtype=np.uint8)
        ref = np.empty((n, 4 * fe), np.uint64)
        check(lib.akp_te_crh_batch(h.h, msgs.ctypes.data, n, L, ref.ctypes.data))  # pageable call: the reference digests
        pm, po = C.c_void_p(), C.c_void_p()
        check(lib.akp_host_alloc(msgs.nbytes, C.byref(pm)))
        check(lib.akp_host_alloc(ref.nbytes, C.byref(po)))
        np.ctypeslib.as_array((C.c_uint8 * msgs.size).from_address(pm.value))[:] = msgs.reshape(-1)
        pout = np.ctypeslib.as_array((C.c_uint64 * ref.size).from_address(po.value)).reshape(ref.shape)
        rec = {}
        for arm in ("chunked", "gated"):
            os.environ["AKP_TE_GATED"] = "1" if arm == "gated" else "0"
            pout[:] = 0
            r = calls(lambda: check(lib.akp_te_crh_batch(h.h, pm, n, L, po)))
            r["hashes_per_s"] = n / (r["ms_median"] / 1e3)
            r["digests_equal_the_pageable_call"] = bool(np.array_equal(pout, ref))
            rec[arm] = r
        # the resident launch for scale
        d_m = torch.from_numpy(msgs).to(dev)
        d_o = torch.empty((n, 4 * fe), dtype=torch.int64, device=dev)
        st = torch.cuda.current_stream(dev).cuda_stream
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(7)]
        check(lib.akp_te_crh_batch_dev(h.h, d_m.data_ptr(), n, L, d_o.data_ptr(), st))
        for a, b in evs:
            a.record()
            check(lib.akp_te_crh_batch_dev(h.h, d_m.data_ptr(), n, L, d_o.data_ptr(), st))
            b.record()
        torch.cuda.synchronize(dev)
        rec["resident_ms"] = sorted(a.elapsed_time(b) for a, b in evs)[3]
        rec["pcie_bytes_in_out"] = [n * L, n * 32 * fe]
        out.setdefault(table, {})[name] = rec
        check(lib.akp_host_free(pm))
        check(lib.akp_host_free(po))
        del h, prm
print(json.dumps(out, indent=1))
