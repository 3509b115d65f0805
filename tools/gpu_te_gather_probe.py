#!/usr/bin/env python3
"""How much of the Pedersen / Bowe-Hopwood kernel time is the table gather?  Same launch with random messages (every lane
fetches its own 144-byte entry per step) and with 2^20 copies of one message (every lane fetches the same entry: the
loads are broadcasts that hit L1).  The arithmetic is identical."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import crypto_primitives_amd as cpa  # noqa: E402
from crypto_primitives_amd import params as cparams  # noqa: E402
from crypto_primitives_amd.crh import pedersen as cped, bowe_hopwood as cbh  # noqa: E402
from crypto_primitives_amd._lib import lib, check  # noqa: E402
dev = torch.device("cuda", 0)
ctx = cpa.default_context(0)
st = torch.cuda.current_stream().cuda_stream
n = 1 << 20


def timed(fn, reps=15, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ms = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        torch.cuda.synchronize()
        ms.append(a.elapsed_time(b))
    ms.sort()
    return ms[len(ms) // 2]


rng = np.random.default_rng(1)
for name, P, L, fe in (("pedersen 4x256, 128 B", cped.Parameters(cparams.pedersen_generators(0xA5A50004, 4, 256)), 128, 8),
                       ("bowe-hopwood 63x9, 64 B", cbh.Parameters(cparams.bowe_hopwood_generators(0xA5A50005, 63, 9)), 64, 4)):
    h = P.handle(ctx)
    out = torch.empty((n, fe), dtype=torch.int64, device=dev)
    rnd = torch.from_numpy(rng.integers(0, 256, size=(n, L), dtype=np.uint8)).to(dev)
    same = rnd[:1].repeat(n, 1).contiguous()
    few = rnd[:64].repeat(n // 64, 1).contiguous()  # 64 distinct messages: one per lane, the same in every wave
    oct8 = rnd[:8].repeat_interleave(8, dim=0).repeat(n // 64, 1).contiguous()  # 8 distinct per wave: lanes 8j .. 8j+7 share one
    for label, m in (("random messages", rnd), ("64 distinct messages", few), ("8 distinct per wave", oct8), ("one message", same)):
        ms = timed(lambda: check(lib.akp_te_crh_batch_dev(h.h, m.data_ptr(), n, L, out.data_ptr(), st)))
        print("%-26s %-22s %.3f ms  %.4g hashes/s" % (name, label, ms, n / ms * 1e3))
