#!/usr/bin/env python3
"""te_host_calls.py under the conditions of bench.py's host_path leg: argv[1] = comma list of {torch, heat}: `torch` imports torch and
initialises its HIP state first; `heat` runs 3 s of back-to-back 2^20-state permutation launches right before the timed calls."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
conds = set(sys.argv[1].split(",")) if len(sys.argv) > 1 and sys.argv[1] != "none" else set()
if "torch" in conds:
    import torch
    torch.zeros(1, device="cuda:0")
import crypto_primitives_amd as cpa  # noqa: E402
from crypto_primitives_amd import params as cparams, field  # noqa: E402
from crypto_primitives_amd.crh import pedersen as cped  # noqa: E402
from crypto_primitives_amd._lib import lib, check  # noqa: E402
n = 1 << 20
ctx = cpa.default_context(0)
h = cped.Parameters(cparams.pedersen_generators(0xA5A50004, 4, 256)).handle(ctx)
hm = np.random.default_rng(3).integers(0, 256, size=(n, 128), dtype=np.uint8)
ho = np.zeros((n, 8), dtype=np.uint64)
pm, po = C.c_void_p(), C.c_void_p()
check(lib.akp_host_alloc(hm.nbytes, C.byref(pm))); check(lib.akp_host_alloc(ho.nbytes, C.byref(po)))
np.ctypeslib.as_array((C.c_uint8 * hm.size).from_address(pm.value))[:] = hm.reshape(-1)


def calls(pin, pout, k=12):
    ts = []
    for _ in range(k):
        t0 = time.perf_counter()
        check(lib.akp_te_crh_batch(h.h, pin, n, 128, pout))
        ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts[1:])[len(ts[1:]) // 2], min(ts[1:])


def heat():
    cfg = cpa.get_default_poseidon_parameters(2, False)
    ph = cfg.handle(ctx)
    st = field.random_fr(n * 3, seed=1).reshape(n, 3, 4)
    d = C.c_void_p()
    import torch as T
    t = T.from_numpy(st.view(np.int64)).to("cuda:0")
    s = T.cuda.current_stream().cuda_stream
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 3.0:
        for _ in range(50):
            check(lib.akp_poseidon_permute_batch_dev(ph.h, t.data_ptr(), n, s))
        T.cuda.synchronize()


for rnd in range(2):
    if "heat" in conds:
        heat()
    a = calls(hm.ctypes.data, ho.ctypes.data)
    if "heat" in conds:
        heat()
    b = calls(pm, po)
    print("conditions %-12s round %d: pageable median %.2f (min %.2f)   pinned median %.2f (min %.2f)" % (",".join(sorted(conds)) or "none", rnd, a[0], a[1], b[0], b[1]))
