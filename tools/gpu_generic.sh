#!/bin/bash
# generic (t != 3) Poseidon kernels: one wave per state lane (default) vs the LDS-file kernel, all default rates
for mode in 1000000000 0; do
echo "== AKP_POSEIDON_GENERIC_COOP_MAX=$mode"
AKP_POSEIDON_GENERIC_COOP_MAX=$mode timeout 600 python - <<PY 2>&1 | grep -v amdgpu.ids
import sys, numpy as np, torch
sys.path.insert(0,'.')
import crypto_primitives_amd as cpa
from crypto_primitives_amd import field
from crypto_primitives_amd._lib import lib, check
dev=torch.device('cuda',0); ctx=cpa.default_context(0); st=torch.cuda.current_stream().cuda_stream
for rate,w in ((3,False),(4,False),(5,False),(8,False),(8,True)):
    c=cpa.get_default_poseidon_parameters(rate,w); h=c.handle(ctx); t=rate+1
    for log2n in (10, 14, 16, 18, 20):
        n=1<<log2n
        x=torch.from_numpy(field.random_fr(n*t,seed=rate).view(np.int64)).to(dev)
        def run(): check(lib.akp_poseidon_permute_batch_dev(h.h,x.data_ptr(),n,st))
        for _ in range(3): run()
        torch.cuda.synchronize(); best=1e9
        for _ in range(5):
            a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); a.record(); run(); b.record(); torch.cuda.synchronize(); best=min(best,a.elapsed_time(b))
        print('rate %d weights=%s alpha=%d rounds=%d+%d n=2^%d: %.3f ms  %.1f M perm/s  %.1f M elements absorbed/s'%(rate,w,c.alpha,c.full_rounds,c.partial_rounds,log2n,best,n/best/1e3,n*rate/best/1e3))
PY
done
