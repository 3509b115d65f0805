#!/usr/bin/env python3
"""Permutation throughput of every default parameter set at 2^20 states (warm clocks: 8 untimed launches, best and median of
7 timed ones).  `AKP_POSEIDON_NO_REG_T=1` in the environment gives the LDS-file arm for t = 4, 5."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import crypto_primitives_amd as cpa  # noqa: E402
from crypto_primitives_amd import field  # noqa: E402
from crypto_primitives_amd._lib import lib, check  # noqa: E402
dev = torch.device("cuda", 0)
ctx = cpa.default_context(0)
st = torch.cuda.current_stream().cuda_stream
n = 1 << 20
print("AKP_POSEIDON_NO_REG_T =", os.environ.get("AKP_POSEIDON_NO_REG_T"))
for rate, w in ((2, False), (3, False), (4, False), (5, False), (8, False), (2, True), (3, True), (4, True), (8, True)):
    c = cpa.get_default_poseidon_parameters(rate, w)
    h = c.handle(ctx)
    t = rate + 1
    x = torch.from_numpy(field.random_fr(n * t, seed=rate).view(np.int64)).to(dev)

    def run():
        check(lib.akp_poseidon_permute_batch_dev(h.h, x.data_ptr(), n, st))
    for _ in range(8):
        run()
    torch.cuda.synchronize()
    ms = []
    for _ in range(7):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); run(); b.record()
        torch.cuda.synchronize()
        ms.append(a.elapsed_time(b))
    ms.sort()
    print("rate %d weights=%-5s alpha=%3d rounds=%d+%-2d %-28s best %.3f ms median %.3f ms  %.1f M perm/s  %.1f M elements absorbed/s"
          % (rate, w, c.alpha, c.full_rounds, c.partial_rounds, lib.akp_poseidon_kernel_for(h.h, n, 0).decode(), ms[0], ms[3], n / ms[0] / 1e3, n * rate / ms[0] / 1e3))
