import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import crypto_primitives_amd as cpa
from crypto_primitives_amd import params as cparams
from crypto_primitives_amd.crh import injective_map as inj, pedersen
from crypto_primitives_amd._lib import lib, check
dev = torch.device("cuda", 0); ctx = cpa.default_context(0); st = torch.cuda.current_stream().cuda_stream
g = cparams.pedersen_generators(0xA5A50004, 4, 256)
for name, P, fe in (("pedersen-x (TECompressor)", inj.Parameters(g), 1), ("pedersen (x, y)", pedersen.Parameters(g), 2)):
    h = P.handle(ctx)
    for lg in (20, 22):
        n = 1 << lg
        lv = torch.from_numpy(np.random.default_rng(1).integers(0, 256, size=(n, 32), dtype=np.uint8)).to(dev)
        ln = torch.empty((n, fe * 4), dtype=torch.int64, device=dev); nl = torch.empty((n - 1, fe * 4), dtype=torch.int64, device=dev)
        f = lambda: check(lib.akp_merkle_build_te_dev(h.h, h.h, lv.data_ptr(), n, 32, ln.data_ptr(), nl.data_ptr(), st))
        f(); f(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); f(); f(); b.record(); torch.cuda.synchronize()
        print("%-28s tree 2^%d x 32-byte leaves: %.2f ms (%.3g leaves/s)" % (name, lg, a.elapsed_time(b) / 3, n / (a.elapsed_time(b) / 3e3)))
