#!/usr/bin/env python3
"""Table width sweep past the Infinity Cache: Pedersen 4x256 digit widths 16 .. 24 (268 MB .. 46 GB tables) and Bowe-Hopwood 63x9
chunk groups 5 .. 8 over 2^20 messages resident in HBM, shapes through akp_te_params_create_shaped.  Every width must give the same
digests (full compare against the first width).  PED_D / BH_G: comma-separated shapes; LOG2_N: messages."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import crypto_primitives_amd as cpa  # noqa: E402
from crypto_primitives_amd import params as cparams  # noqa: E402
from crypto_primitives_amd._lib import lib, check  # noqa: E402
from crypto_primitives_amd.crh import pedersen, bowe_hopwood  # noqa: E402

dev = torch.device("cuda", 0)
ctx = cpa.default_context(0)
st = torch.cuda.current_stream().cuda_stream
n = int(os.environ["N"]) if os.environ.get("N") else 1 << int(os.environ.get("LOG2_N", "20"))
ped_gens = cparams.pedersen_generators(0xA5A50004, 4, 256)
bh_gens = cparams.bowe_hopwood_generators(0xA5A50005, 63, 9)


def info(h):
    i = h.info()
    return i["digit_bits_or_group"], i["table_bytes"]


def measure(h, L, words, reps=9):
    m = torch.from_numpy(np.random.default_rng(4).integers(0, 256, size=(n, L), dtype=np.uint8)).to(dev)
    o = torch.empty((n, words), dtype=torch.int64, device=dev)

    def run():
        check(lib.akp_te_crh_batch_dev(h.h, m.data_ptr(), n, L, o.data_ptr(), st))
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    ms = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(); b.record()
        torch.cuda.synchronize()
        ms.append(a.elapsed_time(b))
    ms.sort()
    return ms[0], ms[len(ms) // 2], o


ref = {}
for D in [int(x) for x in os.environ.get("PED_D", "16,17,18,20,22,24").split(",") if x]:
    t0 = time.perf_counter()
    P = pedersen.Parameters(ped_gens, table_shape=D)
    h = P.handle(ctx)
    torch.cuda.synchronize()
    build = time.perf_counter() - t0
    d, tb = info(h)
    for L in (128, 32):
        best, med, o = measure(h, L, 8)
        key = ("ped", L)
        same = True
        if key in ref:
            same = bool(torch.equal(ref[key], o))
        else:
            ref[key] = o.clone()
        steps = h.info(L)["steps"]
        print("pedersen D=%2d table %8.1f MB build %6.2f s | %3d B: best %.3f ms median %.3f ms  %.1f us/step (%d steps)  %.1f M/s  same digests %s"
              % (d, tb / 1e6, build, L, best, med, med * 1e3 / steps, steps, n / med / 1e3, same), flush=True)
    del h, P
    import gc
    gc.collect()

for G in [int(x) for x in os.environ.get("BH_G", "5,6,7,8").split(",") if x]:
    t0 = time.perf_counter()
    B = bowe_hopwood.Parameters(bh_gens, table_shape=G)
    h = B.handle(ctx)
    torch.cuda.synchronize()
    build = time.perf_counter() - t0
    g, tb = info(h)
    for L in (64, 32):
        best, med, o = measure(h, L, 4)
        key = ("bh", L)
        same = True
        if key in ref:
            same = bool(torch.equal(ref[key], o))
        else:
            ref[key] = o.clone()
        steps = h.info(L)["steps"]
        print("bowe-hopwood G=%d table %8.1f MB build %6.2f s | %3d B: best %.3f ms median %.3f ms  %.1f us/step (%d steps)  %.1f M/s  same digests %s"
              % (g, tb / 1e6, build, L, best, med, med * 1e3 / steps, steps, n / med / 1e3, same), flush=True)
    del h, B
    import gc
    gc.collect()
