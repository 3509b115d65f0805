"""round 6: Bowe-Hopwood 63x9 trees of 2^23 leaves built back to back on a handle under AKP_TABLE_BUDGET_DEVICE, from the moment it exists:
per tree the time (stream-level wait) and the upgrade state -- what does the FIRST tree cost beside the start of the background build?
argv[1] = "null" (torch's default stream) or "own" (a non-blocking stream)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import crypto_primitives_amd as cpa
from crypto_primitives_amd import params as cparams
from crypto_primitives_amd.crh import bowe_hopwood
from crypto_primitives_amd._lib import lib, check
dev = torch.device("cuda", 0)
ctx = cpa.default_context(0)
own = len(sys.argv) > 1 and sys.argv[1] == "own"
alloc = len(sys.argv) > 2 and sys.argv[2]  # "alloc": fresh output tensors inside every timed tree (the previous ones kept alive); "cpu": + root.cpu()
keep = []
st = torch.cuda.Stream(device=dev) if own else torch.cuda.current_stream(dev)
n = 1 << 23
leaves = torch.from_numpy(np.random.default_rng(3).integers(0, 256, size=(n, 32), dtype=np.uint8)).to(dev)
ln = torch.empty((n, 4), dtype=torch.int64, device=dev)
nl = torch.empty((n - 1, 4), dtype=torch.int64, device=dev)
gens = cparams.bowe_hopwood_generators(0xA5A50705, 63, 9)
base = bowe_hopwood.Parameters(gens).handle(ctx)
def tree(h):
    global ln, nl
    t0 = time.perf_counter()
    if alloc:
        keep.append((ln, nl))
        ln = torch.empty((n, 4), dtype=torch.int64, device=dev)
        nl = torch.empty((n - 1, 4), dtype=torch.int64, device=dev)
    check(lib.akp_merkle_build_te_dev(h.h, h.h, leaves.data_ptr(), n, 32, ln.data_ptr(), nl.data_ptr(), st.cuda_stream))
    if alloc == "cpu":
        with torch.cuda.stream(st):
            nl[0].cpu()
    st.synchronize()
    return (time.perf_counter() - t0) * 1e3
warm = [tree(base) for _ in range(4)]
ctx.set_table_budget(cpa._lib.TABLE_BUDGET_DEVICE)
t0 = time.perf_counter()
h = bowe_hopwood.Parameters(gens).handle(ctx)
create_ms = (time.perf_counter() - t0) * 1e3
ctx.set_table_budget(0)
rows = []
for i in range(8):
    ms = tree(h)
    rows.append((round(ms, 2), h.table_info()["last_build"]["upgrade_state"]))
print(json.dumps({"stream": "own non-blocking" if own else "NULL", "mode": alloc or "preallocated outputs", "default_table_trees_ms": [round(x, 2) for x in warm], "create_ms": round(create_ms, 2), "trees_ms_state": rows,
                  "last_build": h.table_info()["last_build"]}))
