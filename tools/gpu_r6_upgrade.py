"""round 6: the background upgrade to an HBM-sized curve table, watched from the host.  A handle created under AKP_TABLE_BUDGET_DEVICE
hashes 2^20-message batches back to back from the moment it exists; per call: wall time, the table the call used (akp_te_params_info),
the upgrade state -- until a few calls after the switch.  Digests of every call against those of the first (same messages)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import crypto_primitives_amd as cpa
from crypto_primitives_amd import params as cparams
from crypto_primitives_amd.crh import pedersen, bowe_hopwood

dev = torch.device("cuda", 0)
ctx = cpa.default_context(0)
stream = torch.cuda.current_stream(dev).cuda_stream
out = {}
for name, cls, gens, ln, fe in (("pedersen_4x256_128B", pedersen, cparams.pedersen_generators(0xA5A50604, 4, 256), 128, 2),
                                ("bowe_hopwood_63x9_64B", bowe_hopwood, cparams.bowe_hopwood_generators(0xA5A50605, 63, 9), 64, 1)):
    n = 1 << 20
    d_m = torch.from_numpy(np.random.default_rng(6).integers(0, 256, size=(n, ln), dtype=np.uint8)).to(dev)
    d_o = torch.empty((n, 4 * fe), dtype=torch.int64, device=dev)
    # warm the context (scratch) with a default-budget handle of other generators
    W = cls.Parameters(cparams.pedersen_generators(0xA5A50699, 4, 256) if cls is pedersen else cparams.bowe_hopwood_generators(0xA5A50698, 63, 9))
    cpa._lib.check(cpa.lib.akp_te_crh_batch_dev(W.handle(ctx).h, d_m.data_ptr(), n, ln, d_o.data_ptr(), stream))
    torch.cuda.synchronize(dev)
    ctx.set_table_budget(cpa._lib.TABLE_BUDGET_DEVICE)
    t_create = time.perf_counter()
    P = cls.Parameters(gens)
    h = P.handle(ctx)
    ctx.set_table_budget(0)
    create_ms = (time.perf_counter() - t_create) * 1e3
    calls, first, after = [], None, 0
    while after < 5 and len(calls) < 4000:
        t0 = time.perf_counter()
        cpa._lib.check(cpa.lib.akp_te_crh_batch_dev(h.h, d_m.data_ptr(), n, ln, d_o.data_ptr(), stream))
        torch.cuda.current_stream(dev).synchronize()  # THIS stream: a device-wide synchronize would wait for the background build as well
        ms = (time.perf_counter() - t0) * 1e3
        st = h.table_info()["last_build"]["upgrade_state"]
        info = h.info(ln)
        got = d_o.clone()
        if first is None:
            first = got
        same = bool(torch.equal(got, first))
        calls.append({"at_ms": (time.perf_counter() - t_create) * 1e3, "call_ms": ms, "shape": info["digit_bits_or_group"], "state": st, "same_digests": same})
        if st == 2 and info["table_bytes"] > (1 << 30):
            after += 1
    rep = h.table_info()
    out[name] = {"create_ms": create_ms, "first_call_ms": calls[0]["call_ms"], "calls_before_switch": sum(1 for c in calls if c["state"] != 2),
                 "switched_after_ms": next((c["at_ms"] for c in calls if c["state"] == 2), None), "slowest_call_ms": max(c["call_ms"] for c in calls),
                 "all_digests_equal": all(c["same_digests"] for c in calls), "table": rep, "info": h.info(ln),
                 "calls_head": calls[:6], "calls_tail": calls[-6:]}
    del P, h
print(json.dumps(out, indent=1))
