#!/bin/bash
# round 5: timeline (kernels + copies) of the gated pinned Pedersen call: rocprofv3 --kernel-trace --memory-copy-trace on a driver that
# makes 5 calls; the last call's records printed relative to its first record.  $1 = session dir, $2 = AKP_TE_GATED (1 / 0), $3 = hbm|cache
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05_s8}; mkdir -p $OUT
export TMPDIR=/tmp
cat > /tmp/gated_driver.py <<'PY'
import ctypes as C, os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np, torch
import crypto_primitives_amd as cpa
from crypto_primitives_amd import params as cparams
from crypto_primitives_amd.crh import pedersen
lib, check = cpa.lib, cpa._lib.check
ctx = cpa.default_context(0)
if os.environ.get("TABLE") == "hbm":
    ctx.set_table_budget(cpa._lib.TABLE_BUDGET_DEVICE)
n = 1 << 20
h = pedersen.Parameters(cparams.pedersen_generators(0xA5A50004, 4, 256)).handle(ctx)
msgs = np.random.default_rng(7).integers(0, 256, size=(n, 128), dtype=np.uint8)
pm, po = C.c_void_p(), C.c_void_p()
check(lib.akp_host_alloc(msgs.nbytes, C.byref(pm))); check(lib.akp_host_alloc(n * 64, C.byref(po)))
np.ctypeslib.as_array((C.c_uint8 * msgs.size).from_address(pm.value))[:] = msgs.reshape(-1)
for _ in range(5):
    check(lib.akp_te_crh_batch(h.h, pm, n, 128, po))
PY
for G in ${2:-1}; do
  rm -rf $OUT/trace_g$G
  (cd /tmp && AKP_TE_GATED=$G TABLE=${3:-hbm} AKP_LIB=$GRAFT_REPO_ROOT/crypto_primitives_amd/lib/libakp_testhooks.so timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/trace_g$G -o t -- python /tmp/gated_driver.py > $OUT/trace_g$G.log 2>&1)
  python $GRAFT_REPO_ROOT/tools/trace_timeline.py $OUT/trace_g$G ${4:-30} > $OUT/timeline_gated${G}_${3:-hbm}.txt
  rm -rf $OUT/trace_g$G
  cat $OUT/timeline_gated${G}_${3:-hbm}.txt
done
