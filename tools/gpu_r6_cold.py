"""round 6: the cold start of an HBM-sized curve table, taken apart (run as the FIRST GPU process of a lease, then again).
Usage: python tools/gpu_r6_cold.py [torch]   -- `torch`: import torch and allocate / free through it first, as bench.py does"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
t_start = time.perf_counter()
if "torch" in sys.argv[1:]:
    import torch
    x = torch.empty(1 << 30, dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
import numpy as np
import crypto_primitives_amd as cpa
from crypto_primitives_amd import params as cparams
from crypto_primitives_amd.crh import pedersen, bowe_hopwood

print("imports + context: %.1f ms" % ((time.perf_counter() - t_start) * 1e3), flush=True)
ctx = cpa.default_context(0)
for name, cls, gens, ln in (("pedersen 4x256", pedersen, cparams.pedersen_generators(0xA5A50004, 4, 256), 128),
                            ("bowe-hopwood 63x9", bowe_hopwood, cparams.bowe_hopwood_generators(0xA5A50005, 63, 9), 64)):
    for budget in (0, cpa._lib.TABLE_BUDGET_DEVICE):
        ctx.set_table_budget(budget)
        P = cls.Parameters(gens)
        t0 = time.perf_counter()
        h = P.handle(ctx)
        t1 = time.perf_counter()
        h.prepare(ln)
        t2 = time.perf_counter()
        msgs = np.random.default_rng(1).integers(0, 256, size=(1 << 16, ln), dtype=np.uint8)
        dg = cls.CRH.evaluate_batch(P, msgs)
        t3 = time.perf_counter()
        import hashlib
        chk = hashlib.sha256(np.ascontiguousarray(dg).tobytes()).hexdigest()[:16]
        print("%s, budget %s: create %.2f ms, prepare(%d) %.2f ms, first 2^16 hashes %.2f ms; %s" % (name, "DEVICE" if budget else "default", (t1 - t0) * 1e3, ln, (t2 - t1) * 1e3,
                                                                                                  (t3 - t2) * 1e3, h.info(ln)), "digests sha256", chk, "build", {k: (round(v, 2) if isinstance(v, float) else v) for k, v in h.table_info()["last_build"].items() if k.endswith("_ms")}, flush=True)
        ctx.set_table_budget(0)
