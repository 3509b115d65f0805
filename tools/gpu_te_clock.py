#!/usr/bin/env python3
"""Effective shader clock and board power under the Pedersen 4x256 kernels at two table widths: is the longer table step of the wide
table (HBM instead of the Infinity Cache) a memory stall or a lower clock?  ClockProbe (one wave of dependent multiply-adds on a side
stream) beside a loop of 2^20-hash launches; power from hwmon while the queue drains."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import crypto_primitives_amd as cpa  # noqa: E402
from crypto_primitives_amd import params as cparams  # noqa: E402
from crypto_primitives_amd._lib import lib, check  # noqa: E402
from crypto_primitives_amd.crh import pedersen  # noqa: E402
from bench_legs.common import ClockProbe, Env, gpu_sensors  # noqa: E402

dev = torch.device("cuda", 0)
env = Env()
env.torch, env.dev, env.lib, env.check, env.ctx = torch, dev, lib, check, cpa.default_context(0)
st = torch.cuda.current_stream().cuda_stream
n = 1 << 20
gens = cparams.pedersen_generators(0xA5A50004, 4, 256)
m = torch.from_numpy(np.random.default_rng(4).integers(0, 256, size=(n, 128), dtype=np.uint8)).to(dev)
o = torch.empty((n, 8), dtype=torch.int64, device=dev)
for rnd in range(2):
    for D in (16, 20, 24):
        h = pedersen.Parameters(gens, table_shape=D).handle(env.ctx)
        steps = h.info(128)["steps"]
        for _ in range(100):
            check(lib.akp_te_crh_batch_dev(h.h, m.data_ptr(), n, 128, o.data_ptr(), st))
        probe = ClockProbe(env)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(300):
            if i % 30 == 15:
                probe.launch()
            check(lib.akp_te_crh_batch_dev(h.h, m.data_ptr(), n, 128, o.data_ptr(), st))
        b.record()
        time.sleep(0.2)
        sens = gpu_sensors()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 300
        mhz = [p["mhz"] for p in probe.read()]
        mean = sum(mhz) / len(mhz)
        print("round %d D=%2d: %.3f ms per 2^20 hashes, %.1f us per table step (%d steps); effective sclk %.0f MHz (min %.0f max %.0f) -> %.0f kcycles per step; power %s W of %s"
              % (rnd, D, ms, ms * 1e3 / steps, steps, mean, min(mhz), max(mhz), ms * 1e3 / steps * mean / 1e3, sens["power_w"], sens["power_cap_w"]), flush=True)
        del h
