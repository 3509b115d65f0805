#!/usr/bin/env python3
"""Small fixed workload for rocprofv3 (kernel trace and --pmc passes): a few launches of every hot kernel at the
BASELINE sizes, inputs resident in HBM.  `python tools/prof_driver.py [poseidon] [te] [tree]`; PROF_TABLES=hbm: the HBM-sized curve
tables instead of the library's default (cache-sized) ones"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import crypto_primitives_amd as cpa  # noqa: E402
from crypto_primitives_amd import field, params as cparams  # noqa: E402
from crypto_primitives_amd._lib import lib, check  # noqa: E402

what = set(sys.argv[1:]) or {"poseidon", "te", "tree"}
dev = torch.device("cuda", 0)
ctx = cpa.default_context(0)
if os.environ.get("PROF_TABLES") == "hbm":  # the HBM-sized curve tables (opt-in since round 5); default: the cache-sized ones
    ctx.set_table_budget(cpa._lib.TABLE_BUDGET_DEVICE)
stream = torch.cuda.current_stream(dev).cuda_stream
REPS = int(os.environ.get("PROF_REPS", "3"))

if "poseidon" in what or "tree" in what:
    cfg = cpa.get_default_poseidon_parameters(2, False)
    ph = cfg.handle(ctx)
if "poseidon" in what:
    n = 1 << 20
    st = torch.from_numpy(field.random_fr(n * 3, seed=1).reshape(n, 3, 4).view(np.int64)).to(dev)
    out = torch.empty((n, 4), dtype=torch.int64, device=dev)
    for _ in range(REPS):
        check(lib.akp_poseidon_permute_batch_dev(ph.h, st.data_ptr(), n, stream))
    for _ in range(REPS):
        check(lib.akp_poseidon_crh_batch_dev(ph.h, st.data_ptr(), n, 2, out.data_ptr(), stream))
    for rate in (3, 4):
        c = cpa.get_default_poseidon_parameters(rate, False)
        h = c.handle(ctx)
        s2 = torch.from_numpy(field.random_fr(n * (rate + 1), seed=2).reshape(n, rate + 1, 4).view(np.int64)).to(dev)
        for _ in range(REPS):
            check(lib.akp_poseidon_permute_batch_dev(h.h, s2.data_ptr(), n, stream))
    torch.cuda.synchronize()
if "tree" in what:
    n = 1 << 22
    lv = torch.from_numpy(field.random_fr(1 << 20, seed=3).reshape(-1, 1, 4).view(np.int64)).to(dev).repeat(4, 1, 1)
    ln = torch.empty((n, 4), dtype=torch.int64, device=dev)
    nl = torch.empty((n - 1, 4), dtype=torch.int64, device=dev)
    for _ in range(2):
        check(lib.akp_merkle_build_poseidon_dev(ph.h, ph.h, lv.data_ptr(), n, 1, ln.data_ptr(), nl.data_ptr(), stream))
    torch.cuda.synchronize()
if "te" in what:
    from crypto_primitives_amd.crh import pedersen, bowe_hopwood
    n = 1 << 20
    P = pedersen.Parameters(cparams.pedersen_generators(0xA5A50004, 4, 256))
    B = bowe_hopwood.Parameters(cparams.bowe_hopwood_generators(0xA5A50005, 63, 9))
    hp, hb = P.handle(ctx), B.handle(ctx)
    hp.prepare(128)  # (PROF_TABLES=hbm: the wide tables built before the profiled launches, not in the background beside them)
    hb.prepare(70)
    hb.prepare(32)
    rng = np.random.default_rng(4)
    m128 = torch.from_numpy(rng.integers(0, 256, size=(n, 128), dtype=np.uint8)).to(dev)
    m32 = torch.from_numpy(rng.integers(0, 256, size=(n, 32), dtype=np.uint8)).to(dev)
    m70 = torch.from_numpy(rng.integers(0, 256, size=(n, 70), dtype=np.uint8)).to(dev)
    o2 = torch.empty((n, 8), dtype=torch.int64, device=dev)
    o1 = torch.empty((n, 4), dtype=torch.int64, device=dev)
    for _ in range(REPS):
        check(lib.akp_te_crh_batch_dev(hp.h, m128.data_ptr(), n, 128, o2.data_ptr(), stream))
    for _ in range(REPS):
        check(lib.akp_te_crh_batch_dev(hb.h, m32.data_ptr(), n, 32, o1.data_ptr(), stream))
    for _ in range(REPS):
        check(lib.akp_te_crh_batch_dev(hb.h, m70.data_ptr(), n, 70, o1.data_ptr(), stream))
    for _ in range(REPS):  # the small-batch (split) kernel: one level near the top of a tree
        check(lib.akp_te_crh_batch_dev(hb.h, m70.data_ptr(), 1 << 12, 70, o1.data_ptr(), stream))
    torch.cuda.synchronize()
print("prof_driver done:", sorted(what))
