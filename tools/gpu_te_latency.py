"""Latency of the Bowe-Hopwood / Pedersen CRH on small batches (the upper levels of a byte-digest tree):
device-side timing (HIP events) of akp_te_crh_batch_dev over a batch-size sweep."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
import crypto_primitives_amd as cpa
from crypto_primitives_amd import params
from crypto_primitives_amd._lib import lib, check
from crypto_primitives_amd.crh import bowe_hopwood, pedersen

B = bowe_hopwood.Parameters(params.bowe_hopwood_generators(0xA5A50005, 63, 9))
P = pedersen.Parameters(params.pedersen_generators(0xA5A50004, 4, 256))
dev = torch.device("cuda", 0)
ctx = cpa.default_context(0)
stream = torch.cuda.current_stream(dev).cuda_stream
rng = np.random.default_rng(1)
for name, prm, ln, fe in (("bh 70B", B, 70, 1), ("bh 32B", B, 32, 1), ("pedersen 128B", P, 128, 2)):
    h = prm.handle(ctx)
    for log2n in (0, 6, 10, 12, 14, 15, 16, 17, 18, 20):
        n = 1 << log2n
        msgs = torch.from_numpy(rng.integers(0, 256, size=(n, ln), dtype=np.uint8)).to(dev)
        out = torch.empty((n, fe * 4), dtype=torch.int64, device=dev)
        for _ in range(3):
            check(lib.akp_te_crh_batch_dev(h.h, msgs.data_ptr(), n, ln, out.data_ptr(), stream))
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
        for a, b in ev:
            a.record(); check(lib.akp_te_crh_batch_dev(h.h, msgs.data_ptr(), n, ln, out.data_ptr(), stream)); b.record()
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in ev)
        print("%-14s n=2^%-2d median %.4f ms" % (name, log2n, ms[len(ms) // 2]))
