#!/usr/bin/env python3
"""per-call wall times of akp_te_crh_batch (Pedersen 4x256, 2^20 x 128 B): argv = in-kind out-kind [calls]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import crypto_primitives_amd as cpa  # noqa: E402
from crypto_primitives_amd import params as cparams  # noqa: E402
from crypto_primitives_amd.crh import pedersen as cped  # noqa: E402
from crypto_primitives_amd._lib import lib, check  # noqa: E402
kin, kout = sys.argv[1], sys.argv[2]
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 12
n = 1 << 20
h = cped.Parameters(cparams.pedersen_generators(0xA5A50004, 4, 256)).handle(cpa.default_context(0))
hm = np.random.default_rng(3).integers(0, 256, size=(n, 128), dtype=np.uint8)
ho = np.zeros((n, 8), dtype=np.uint64)
pin, pout = hm.ctypes.data, ho.ctypes.data
if kin == "pinned":
    pm = C.c_void_p(); check(lib.akp_host_alloc(hm.nbytes, C.byref(pm)))
    np.ctypeslib.as_array((C.c_uint8 * hm.size).from_address(pm.value))[:] = hm.reshape(-1)
    pin = pm
if kout == "pinned":
    po = C.c_void_p(); check(lib.akp_host_alloc(ho.nbytes, C.byref(po)))
    np.ctypeslib.as_array((C.c_uint64 * ho.size).from_address(po.value))[:] = 0
    pout = po
ts = []
for i in range(calls):
    t0 = time.perf_counter()
    check(lib.akp_te_crh_batch(h.h, pin, n, 128, pout))
    ts.append((time.perf_counter() - t0) * 1e3)
print("%s in, %s out: %s  | median %.2f ms" % (kin, kout, " ".join("%.2f" % t for t in ts), sorted(ts[1:])[len(ts[1:]) // 2]))
