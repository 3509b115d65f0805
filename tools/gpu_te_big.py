"""Bowe-Hopwood / Pedersen CRH at 2^20 messages, a few launches each (target of PMC passes)."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
import crypto_primitives_amd as cpa
from crypto_primitives_amd import params
from crypto_primitives_amd._lib import lib, check
from crypto_primitives_amd.crh import bowe_hopwood, pedersen
B = bowe_hopwood.Parameters(params.bowe_hopwood_generators(0xA5A50005, 63, 9))
P = pedersen.Parameters(params.pedersen_generators(0xA5A50004, 4, 256))
dev = torch.device("cuda", 0); ctx = cpa.default_context(0)
stream = torch.cuda.current_stream(dev).cuda_stream
rng = np.random.default_rng(1)
n = 1 << 20
for prm, ln, fe in ((B, 70, 1), (P, 128, 2)):
    h = prm.handle(ctx)
    msgs = torch.from_numpy(rng.integers(0, 256, size=(n, ln), dtype=np.uint8)).to(dev)
    out = torch.empty((n, fe * 4), dtype=torch.int64, device=dev)
    for _ in range(4):
        check(lib.akp_te_crh_batch_dev(h.h, msgs.data_ptr(), n, ln, out.data_ptr(), stream))
    torch.cuda.synchronize()
