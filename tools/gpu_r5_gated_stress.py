#!/usr/bin/env python3
"""round 5: stress of the gated pinned path -- for `seconds` (argv[1], default 60) random batch sizes between 2^17 + 1 and 2^21 messages,
Pedersen 4x256 (random message length 4..128 bytes) and Bowe-Hopwood 63x9 (4..70 bytes), default and HBM-sized tables in turn, every
batch with FRESH random bytes in the same pinned buffer; digests compared with the resident launch over the same bytes.  Prints the
number of calls, how many of them ran gated (test build, AKP_LIB=.../libakp_testhooks.so: AKP_TE_GATE_REPORT; the context stops gating after a
failed gate), and the mismatches (must be 0)."""
import ctypes as C
import json
import os
import sys
import time
os.environ.setdefault("AKP_TE_PINNED_FORM", "gated")  # these arms choose the form themselves (round 6: the library otherwise measures and picks)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import crypto_primitives_amd as cpa  # noqa: E402
from crypto_primitives_amd import params as cparams  # noqa: E402
from crypto_primitives_amd.crh import pedersen, bowe_hopwood  # noqa: E402

lib, check = cpa.lib, cpa._lib.check
REPORT = "/tmp/akp_gate_report.txt"
if os.path.exists(REPORT):
    os.remove(REPORT)
os.environ["AKP_TE_GATE_REPORT"] = REPORT
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream(dev).cuda_stream
NMAX = 1 << 21
pm, po = C.c_void_p(), C.c_void_p()
check(lib.akp_host_alloc(NMAX * 128, C.byref(pm)))
check(lib.akp_host_alloc(NMAX * 64, C.byref(po)))
hm = np.ctypeslib.as_array((C.c_uint8 * (NMAX * 128)).from_address(pm.value))
ho = np.ctypeslib.as_array((C.c_uint64 * (NMAX * 8)).from_address(po.value))
rng = np.random.default_rng(2025)
pool = rng.integers(0, 256, size=NMAX * 128 + 4096, dtype=np.uint8)  # fresh bytes per call = a random window of this pool
out = {"seconds": seconds, "calls": 0, "mismatching_calls": 0, "by_case": {}}
t_end = time.time() + seconds
turn = 0
while time.time() < t_end:
    table = ("cache_sized", "hbm_sized")[turn & 1]
    turn += 1
    ctx = cpa.default_context(0)
    ctx.set_table_budget(0 if table == "cache_sized" else cpa._lib.TABLE_BUDGET_DEVICE)
    for kind in ("pedersen", "bowe_hopwood"):
        prm = pedersen.Parameters(cparams.pedersen_generators(0xA5A50004, 4, 256)) if kind == "pedersen" else \
            bowe_hopwood.Parameters(cparams.bowe_hopwood_generators(0xA5A50005, 63, 9))
        fe = 2 if kind == "pedersen" else 1
        h = prm.handle(ctx)
        h.prepare(128 if kind == "pedersen" else 189)  # round 6: the wide table (built in the background otherwise) before the measured calls
        for _ in range(12):
            if time.time() >= t_end:
                break
            n = int(rng.integers((1 << 17) + 1, NMAX + 1))
            L = int(rng.integers(4, 129 if kind == "pedersen" else 71))
            off = int(rng.integers(0, 4096))
            hm[:n * L] = pool[off:off + n * L]
            ho[:n * 4 * fe] = 0
            check(lib.akp_te_crh_batch(h.h, pm, n, L, po))
            d_m = torch.from_numpy(pool[off:off + n * L]).to(dev)
            d_o = torch.empty((n, 4 * fe), dtype=torch.int64, device=dev)
            check(lib.akp_te_crh_batch_dev(h.h, d_m.data_ptr(), n, L, d_o.data_ptr(), st))
            torch.cuda.synchronize(dev)
            same = bool(np.array_equal(ho[:n * 4 * fe].reshape(n, 4 * fe), d_o.cpu().numpy().view(np.uint64)))
            out["calls"] += 1
            out["mismatching_calls"] += 0 if same else 1
            c = out["by_case"].setdefault(table + "/" + kind, {"calls": 0, "messages": 0})
            c["calls"] += 1
            c["messages"] += n
        del h, prm
check(lib.akp_host_free(pm))
check(lib.akp_host_free(po))
if os.path.exists(REPORT):  # test build only
    rows = [ln.split() for ln in open(REPORT)]
    out["pinned_calls_reported"] = len(rows)
    out["ran_gated"] = sum(int(r[2]) for r in rows)
print(json.dumps(out, indent=1))
sys.exit(1 if out["mismatching_calls"] else 0)
