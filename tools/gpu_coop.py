"""wave-per-lane latency kernel vs the one-lane-per-item kernel: two_to_one timings over a batch-size sweep, plus an
equality check of the digests.  Run once per mode (the switch is read once per process):
  AKP_POSEIDON_COOP_MAX=0 python tools/gpu_coop.py ; AKP_POSEIDON_COOP_MAX=1000000000 python tools/gpu_coop.py"""
import os, sys, hashlib, numpy as np, torch
sys.path.insert(0, ".")
import crypto_primitives_amd as cpa
from crypto_primitives_amd import field
from crypto_primitives_amd._lib import lib, check

dev = torch.device("cuda", 0)
ctx = cpa.default_context(0)
cfg = cpa.get_default_poseidon_parameters(2, False)
ph = cfg.handle(ctx)
N = 1 << 19
l = torch.from_numpy(field.random_fr(N, seed=11).view(np.int64)).to(dev)
r = torch.from_numpy(field.random_fr(N, seed=12).view(np.int64)).to(dev)
o = torch.zeros_like(l)
stream = torch.cuda.current_stream(dev).cuda_stream
for _ in range(30):  # clocks
    check(lib.akp_poseidon_two_to_one_batch_dev(ph.h, l.data_ptr(), r.data_ptr(), N, o.data_ptr(), stream))
print("mode AKP_POSEIDON_COOP_MAX=%s" % os.environ.get("AKP_POSEIDON_COOP_MAX", "default"))
for log2n in list(range(0, 20)):
    n = 1 << log2n
    o.zero_()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for a, b in ev:
        a.record(); check(lib.akp_poseidon_two_to_one_batch_dev(ph.h, l.data_ptr(), r.data_ptr(), n, o.data_ptr(), stream)); b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    h = hashlib.sha256(o.cpu().numpy().tobytes()).hexdigest()[:16]
    print("n=2^%-2d  median %.4f ms  min %.4f ms  digest-hash %s" % (log2n, ms[len(ms) // 2], ms[0], h))
