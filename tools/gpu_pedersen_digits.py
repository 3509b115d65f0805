#!/usr/bin/env python3
"""Pedersen 4x256 over 2^20 x 128 B at the table digit width given by AKP_PEDERSEN_DIGIT_BITS (read at parameter creation)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import crypto_primitives_amd as cpa  # noqa: E402
from crypto_primitives_amd import params as cparams  # noqa: E402
from crypto_primitives_amd._lib import lib, check  # noqa: E402
from crypto_primitives_amd.crh import pedersen  # noqa: E402
dev = torch.device("cuda", 0)
ctx = cpa.default_context(0)
st = torch.cuda.current_stream().cuda_stream
n = 1 << 20
P = pedersen.Parameters(cparams.pedersen_generators(0xA5A50004, 4, 256))
h = P.handle(ctx)
for L in (128, 32):
    m = torch.from_numpy(np.random.default_rng(4).integers(0, 256, size=(n, L), dtype=np.uint8)).to(dev)
    o = torch.empty((n, 8), dtype=torch.int64, device=dev)

    def run():
        check(lib.akp_te_crh_batch_dev(h.h, m.data_ptr(), n, L, o.data_ptr(), st))
    for _ in range(6):
        run()
    torch.cuda.synchronize()
    ms = []
    for _ in range(7):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(); b.record()
        torch.cuda.synchronize()
        ms.append(a.elapsed_time(b))
    ms.sort()
    print("D=%s  %3d-byte messages: best %.3f ms median %.3f ms  %.1f M hashes/s  checksum %d" % (os.environ.get("AKP_PEDERSEN_DIGIT_BITS", "13 (default)"), L, ms[0], ms[3], n / ms[0] / 1e3, int(o.sum().item()) & 0xffffffff))
