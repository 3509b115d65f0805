#!/bin/bash
for cfg in "12 3" "13 4" "14 4" "12 4"; do set -- $cfg; echo "== D=$1 G=$2"; AKP_PEDERSEN_DIGIT_BITS=$1 AKP_BH_GROUP=$2 timeout 600 python - <<PY
import sys, time, numpy as np, torch
sys.path.insert(0,'.')
import crypto_primitives_amd as cpa
from crypto_primitives_amd import params
from crypto_primitives_amd._lib import lib, check
from crypto_primitives_amd.crh import pedersen, bowe_hopwood
dev=torch.device('cuda',0); ctx=cpa.default_context(0)
n=1<<20
msgs=np.random.default_rng(4).integers(0,256,size=(n,128),dtype=np.uint8)
st=torch.cuda.current_stream().cuda_stream
o=torch.empty(n*8,dtype=torch.int64,device=dev)
def bench(h,L):
    d=torch.from_numpy(np.ascontiguousarray(msgs[:,:L]).reshape(-1)).to(dev)
    def run(): check(lib.akp_te_crh_batch_dev(h.h,d.data_ptr(),n,L,o.data_ptr(),st))
    run(); torch.cuda.synchronize(); best=1e9
    for _ in range(3):
        a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); a.record(); run(); b.record(); torch.cuda.synchronize(); best=min(best,a.elapsed_time(b))
    return best, int(o[:n].sum().item())&0xffffffff
g=params.pedersen_generators(0xA5A50004,4,256)
t0=time.time(); P=pedersen.Parameters(g); h=P.handle(ctx); torch.cuda.synchronize(); tb=time.time()-t0
ms,cs=bench(h,128); print('pedersen 128B: %.3f ms %.1f M/s  (table build %.3f s) cs %x'%(ms,n/ms/1e3,tb,cs))
gb=params.bowe_hopwood_generators(0xA5A50005,63,9)
t0=time.time(); B=bowe_hopwood.Parameters(gb); hb=B.handle(ctx); torch.cuda.synchronize(); tb=time.time()-t0
for L in (32,70):
    ms,cs=bench(hb,L); print('bh %dB: %.3f ms %.1f M/s (table build %.3f s) cs %x'%(L,ms,n/ms/1e3,tb,cs))
PY
done
