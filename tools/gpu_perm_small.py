"""latency of akp_poseidon_permute_batch_dev on small batches (duplex-sponge steps), t = 3"""
import sys, numpy as np, torch
sys.path.insert(0, ".")
import crypto_primitives_amd as cpa
from crypto_primitives_amd import field
from crypto_primitives_amd._lib import lib, check
dev = torch.device("cuda", 0); ctx = cpa.default_context(0)
cfg = cpa.get_default_poseidon_parameters(2, False); ph = cfg.handle(ctx)
stream = torch.cuda.current_stream(dev).cuda_stream
x = torch.from_numpy(field.random_fr(3 << 17, seed=1).view(np.int64)).to(dev)
for _ in range(20): check(lib.akp_poseidon_permute_batch_dev(ph.h, x.data_ptr(), 1 << 17, stream))
for log2n in (0, 6, 10, 14, 15, 16, 17):
    n = 1 << log2n
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
    for a, b in ev:
        a.record(); check(lib.akp_poseidon_permute_batch_dev(ph.h, x.data_ptr(), n, stream)); b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    print("permute t=3 n=2^%-2d median %.4f ms" % (log2n, ms[5]))
