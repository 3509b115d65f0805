#!/usr/bin/env python3
"""akp_poseidon_permute_batch on a pinned 2^20-state buffer, three calls: for `rocprofv3 --kernel-trace --memory-copy-trace`
(timeline of the chunked host path)."""
import ctypes as C
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import crypto_primitives_amd as cpa  # noqa: E402
from crypto_primitives_amd import field  # noqa: E402
from crypto_primitives_amd._lib import lib, check  # noqa: E402
n = 1 << 20
cfg = cpa.get_default_poseidon_parameters(2, False)
ph = cfg.handle()
st = field.random_fr(n * 3, seed=1).reshape(n, 3, 4)
pp = C.c_void_p()
check(lib.akp_host_alloc(st.nbytes, C.byref(pp)))
np.ctypeslib.as_array((C.c_uint64 * st.size).from_address(pp.value))[:] = st.reshape(-1)
for i in range(3):
    t0 = time.perf_counter()
    check(lib.akp_poseidon_permute_batch(ph.h, pp, n))
    print("call %d: %.2f ms" % (i, (time.perf_counter() - t0) * 1e3))
