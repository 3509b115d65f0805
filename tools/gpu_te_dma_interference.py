#!/usr/bin/env python3
"""Does DMA traffic slow the curve-hash kernel?  Resident Pedersen batches of 2^17 messages (one chunk of the host pipeline)
timed alone, beside a continuous pinned H2D copy stream, beside a D2H copy stream, and beside both (profiles/r04_s5)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import crypto_primitives_amd as cpa  # noqa: E402
from crypto_primitives_amd import params as cparams  # noqa: E402
from crypto_primitives_amd.crh import pedersen as cped  # noqa: E402
from crypto_primitives_amd._lib import lib, check  # noqa: E402
dev = torch.device("cuda", 0)
ctx = cpa.default_context(0)
h = cped.Parameters(cparams.pedersen_generators(0xA5A50004, 4, 256)).handle(ctx)
n = 1 << 17
msgs = torch.from_numpy(np.random.default_rng(1).integers(0, 256, size=(n, 128), dtype=np.uint8)).to(dev)
out = torch.empty((n, 8), dtype=torch.int64, device=dev)
st = torch.cuda.current_stream().cuda_stream
pin_in = torch.empty(16 << 20, dtype=torch.uint8).pin_memory()
pin_out = torch.empty(8 << 20, dtype=torch.uint8).pin_memory()
d_in = torch.empty(16 << 20, dtype=torch.uint8, device=dev)
d_out = torch.empty(8 << 20, dtype=torch.uint8, device=dev)
s_in, s_out = torch.cuda.Stream(), torch.cuda.Stream()


def run(label, h2d, d2h, reps=24):
    for _ in range(4):
        check(lib.akp_te_crh_batch_dev(h.h, msgs.data_ptr(), n, 128, out.data_ptr(), st))
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        if h2d:
            with torch.cuda.stream(s_in):
                d_in.copy_(pin_in, non_blocking=True)
        if d2h:
            with torch.cuda.stream(s_out):
                pin_out.copy_(d_out, non_blocking=True)
        a.record()
        check(lib.akp_te_crh_batch_dev(h.h, msgs.data_ptr(), n, 128, out.data_ptr(), st))
        b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in evs)
    print("%-46s accumulate + finalize of 2^17 messages: median %.3f ms  min %.3f  max %.3f" % (label, ms[len(ms) // 2], ms[0], ms[-1]))


run("alone", False, False)
run("beside a 16 MiB pinned H2D copy per launch", True, False)
run("beside an 8 MiB D2H copy to pinned per launch", False, True)
run("beside both", True, True)
run("alone again", False, False)
