# LDS-file Poseidon kernels (t != 3, large batches): throughput per block size (0 = built-in choice)
for B in 0 64 128 256; do
echo "== block $B"
if [ $B != 0 ]; then export AKP_POSEIDON_FILE_BLOCK=$B; fi
timeout 600 python - <<PY 2>&1 | grep -v amdgpu.ids
import sys, numpy as np, torch
sys.path.insert(0,'.')
import crypto_primitives_amd as cpa
from crypto_primitives_amd import field
from crypto_primitives_amd._lib import lib, check
dev=torch.device('cuda',0); ctx=cpa.default_context(0); st=torch.cuda.current_stream().cuda_stream
for rate in (3,4,5,6,7,8):
    c=cpa.get_default_poseidon_parameters(rate,False); h=c.handle(ctx); t=rate+1
    n=1<<20
    x=torch.from_numpy(field.random_fr(n*t,seed=rate).view(np.int64)).to(dev)
    def run(): check(lib.akp_poseidon_permute_batch_dev(h.h,x.data_ptr(),n,st))
    for _ in range(4): run()
    torch.cuda.synchronize(); best=1e9
    for _ in range(5):
        a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); a.record(); run(); b.record(); torch.cuda.synchronize(); best=min(best,a.elapsed_time(b))
    print('rate %d: %.3f ms  %.1f M perm/s'%(rate,best,n/best/1e3))
PY
done
