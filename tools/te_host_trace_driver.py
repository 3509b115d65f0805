#!/usr/bin/env python3
"""akp_te_crh_batch (Pedersen 4x256, 2^20 x 128 B) through the host-pointer entry point, three calls, for
`rocprofv3 --kernel-trace --memory-copy-trace`: argv[1] = in-kind, argv[2] = out-kind, each `pinned` or `pageable`."""
import ctypes as C
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import crypto_primitives_amd as cpa  # noqa: E402
from crypto_primitives_amd import params as cparams  # noqa: E402
from crypto_primitives_amd.crh import pedersen as cped  # noqa: E402
from crypto_primitives_amd._lib import lib, check  # noqa: E402
kin, kout = sys.argv[1], sys.argv[2]
n = 1 << 20
h = cped.Parameters(cparams.pedersen_generators(0xA5A50004, 4, 256)).handle(cpa.default_context(0))
hm = np.random.default_rng(3).integers(0, 256, size=(n, 128), dtype=np.uint8)
ho = np.empty((n, 8), dtype=np.uint64)
pin, pout = hm.ctypes.data, ho.ctypes.data
if kin == "pinned":
    pm = C.c_void_p(); check(lib.akp_host_alloc(hm.nbytes, C.byref(pm)))
    np.ctypeslib.as_array((C.c_uint8 * hm.size).from_address(pm.value))[:] = hm.reshape(-1)
    pin = pm
if kout == "pinned":
    po = C.c_void_p(); check(lib.akp_host_alloc(ho.nbytes, C.byref(po)))
    pout = po
for i in range(3):
    t0 = time.perf_counter()
    check(lib.akp_te_crh_batch(h.h, pin, n, 128, pout))
    print("%s in, %s out, call %d: %.2f ms" % (kin, kout, i, (time.perf_counter() - t0) * 1e3))
