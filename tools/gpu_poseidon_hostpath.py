#!/usr/bin/env python3
"""akp_poseidon_permute_batch through the host-pointer entry point: per-call wall ms, pageable and pinned (zero copy), 2^20 and 2^22
states; correctness of the pinned call against the pageable one.  A/B arm: AKP_POSEIDON_STAGED_IO=0/1."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import crypto_primitives_amd as cpa  # noqa: E402
from crypto_primitives_amd import field  # noqa: E402
from crypto_primitives_amd._lib import lib, check  # noqa: E402
cfg = cpa.get_default_poseidon_parameters(2, False)
ph = cfg.handle()
print("# AKP_POSEIDON_STAGED_IO=%s" % os.environ.get("AKP_POSEIDON_STAGED_IO", "1"))
for lg in (20, 22):
    n = 1 << lg
    st = field.random_fr(n * 3, seed=lg).reshape(n, 3, 4)
    ref = st.copy()
    check(lib.akp_poseidon_permute_batch(ph.h, ref.ctypes.data, n))
    pp = C.c_void_p()
    check(lib.akp_host_alloc(st.nbytes, C.byref(pp)))
    arr = np.ctypeslib.as_array((C.c_uint64 * st.size).from_address(pp.value)).reshape(st.shape)
    arr[:] = st
    check(lib.akp_poseidon_permute_batch(ph.h, pp, n))
    same = bool(np.array_equal(arr, ref))
    for label, ptr in (("pageable", st.ctypes.data), ("pinned", pp)):
        ts = []
        for _ in range(10):
            t0 = time.perf_counter()
            check(lib.akp_poseidon_permute_batch(ph.h, ptr, n))
            ts.append((time.perf_counter() - t0) * 1e3)
        ts = sorted(ts[1:])
        print("2^%d states %-8s median %.3f ms (min %.3f)  %.4g perm/s  %.1f GB/s each way%s" % (lg, label, ts[len(ts) // 2], ts[0], n / (ts[len(ts) // 2] / 1e3),
              96.0 * n / (ts[len(ts) // 2] / 1e3) / 1e9, "  pinned == pageable: %s" % same if label == "pinned" else ""))
    check(lib.akp_host_free(pp))
