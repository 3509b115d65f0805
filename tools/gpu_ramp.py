"""Per-launch durations of the headline kernel from a cold device (clock ramp / steady state)."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
import crypto_primitives_amd as cpa
from crypto_primitives_amd import field
from crypto_primitives_amd._lib import lib, check

dev = torch.device("cuda", 0)
ctx = cpa.default_context(0)
cfg = cpa.get_default_poseidon_parameters(2, False)
ph = cfg.handle(ctx)
for log2n in (20, 22):
    n = 1 << log2n
    st = torch.from_numpy(field.random_fr(n * 3, seed=1).reshape(n, 3, 4).view(np.int64)).to(dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    K = 120
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
    torch.cuda.synchronize()
    for a, b in ev:
        a.record(); check(lib.akp_poseidon_permute_batch_dev(ph.h, st.data_ptr(), n, stream)); b.record()
    torch.cuda.synchronize()
    ms = [a.elapsed_time(b) for a, b in ev]
    print(log2n, " ".join("%.3f" % x for x in ms[:12]), "... median %.3f min %.3f last10 %.3f" % (np.median(ms), min(ms), np.mean(ms[-10:])))
