"""round 6: soak of the shared-table store with background upgrades.  T threads, each with a context of its own, for S seconds: pick one of
three Bowe-Hopwood generator sets, create a handle under a random budget (default / device), hash a few batches of random length and size
(host-pointer entry point), sometimes `prepare`, drop the handle -- so that tables are attached, upgraded in the background, extended,
released (the release waits for a running builder) and re-created concurrently.  Every digest against digests of a reference handle with a
small explicit shape.  Prints counts; exits non-zero on any mismatch / error."""
import json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import crypto_primitives_amd as cpa
from crypto_primitives_amd._lib import Context, TABLE_BUDGET_DEVICE
from crypto_primitives_amd import params as cparams
from crypto_primitives_amd.crh import bowe_hopwood
lib, check = cpa.lib, cpa._lib.check
T, S = int(os.environ.get("SOAK_THREADS", "6")), float(os.environ.get("SOAK_SECONDS", "60"))
gens = [cparams.bowe_hopwood_generators(0xA5A50A00 + i, 63, 9) for i in range(3)]
LENS = (8, 32, 64, 70, 100, 212)
ref_ctx = Context(0)
ref = {}
msgs = {L: np.random.default_rng(L).integers(0, 256, size=(3000, L), dtype=np.uint8) for L in LENS}
for gi, g in enumerate(gens):
    h = bowe_hopwood.Parameters(g, table_shape=3).handle(ref_ctx)
    for L in LENS:
        out = np.empty((3000, 4), np.uint64)
        check(lib.akp_te_crh_batch(h.h, msgs[L].ctypes.data, 3000, L, out.ctypes.data))
        ref[(gi, L)] = out
    del h
stats = {"handles": 0, "batches": 0, "on_wide": 0, "prepares": 0, "errors": []}
lock = threading.Lock()
t_end = time.perf_counter() + S

def work(i):
    rng = np.random.default_rng(1000 + i)
    ctx = Context(0)
    try:
        while time.perf_counter() < t_end:
            gi = int(rng.integers(0, 3))
            ctx.set_table_budget(TABLE_BUDGET_DEVICE if rng.random() < 0.7 else 0)
            P = bowe_hopwood.Parameters(gens[gi])
            h = P.handle(ctx)
            nb = 0
            for _ in range(int(rng.integers(1, 12))):
                L = LENS[int(rng.integers(0, len(LENS)))]
                n = int(rng.integers(1, 3000))
                if rng.random() < 0.15:
                    h.prepare(L)
                    with lock:
                        stats["prepares"] += 1
                out = np.empty((n, 4), np.uint64)
                check(lib.akp_te_crh_batch(h.h, msgs[L].ctypes.data, n, L, out.ctypes.data))
                if not np.array_equal(out, ref[(gi, L)][:n]):
                    raise AssertionError("thread %d: digests differ (generators %d, length %d, n %d)" % (i, gi, L, n))
                nb += 1
                if h.info(L)["digit_bits_or_group"] == 8:
                    with lock:
                        stats["on_wide"] += 1
            with lock:
                stats["handles"] += 1
                stats["batches"] += nb
            P._handles.clear()
            del h, P
    except BaseException as e:  # noqa: BLE001
        with lock:
            stats["errors"].append(repr(e)[:300])
    finally:
        ctx.close()

ts = [threading.Thread(target=work, args=(i,)) for i in range(T)]
for t in ts:
    t.start()
for t in ts:
    t.join()
print(json.dumps(stats))
sys.exit(1 if stats["errors"] else 0)
