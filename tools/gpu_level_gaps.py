#!/usr/bin/env python3
"""Where does the time of the narrow tree levels go -- kernel or launch gap?  (fused top levels.)

  run:      rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python tools/gpu_level_gaps.py run
  analyse:  python tools/gpu_level_gaps.py analyse DIR

`run` builds a 2^20-leaf Poseidon tree and a 2^20-leaf Bowe-Hopwood tree three times each through the `_dev` entry points (one
stream, nothing but the level launches).  `analyse` lists, for the last build of each, every launch bottom-up with its grid,
its duration and the idle gap since the previous kernel ended, and sums both over the levels of <= 2^15 nodes."""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run():
    import numpy as np
    import torch
    import crypto_primitives_amd as cpa
    from crypto_primitives_amd import field, params as cparams
    from crypto_primitives_amd._lib import lib, check
    from crypto_primitives_amd.crh import bowe_hopwood, pedersen
    dev = torch.device("cuda", 0)
    ctx = cpa.default_context(0)
    st = torch.cuda.current_stream(dev).cuda_stream
    n = 1 << 20
    cfg = cpa.get_default_poseidon_parameters(2, False)
    ph = cfg.handle(ctx)
    lv = torch.from_numpy(field.random_fr(n, seed=3).reshape(n, 1, 4).view(np.int64)).to(dev)
    ln = torch.empty((n, 4), dtype=torch.int64, device=dev)
    nl = torch.empty((n - 1, 4), dtype=torch.int64, device=dev)
    for _ in range(3):
        check(lib.akp_merkle_build_poseidon_dev(ph.h, ph.h, lv.data_ptr(), n, 1, ln.data_ptr(), nl.data_ptr(), st))
        torch.cuda.synchronize()
    B = bowe_hopwood.Parameters(cparams.bowe_hopwood_generators(0xA5A50005, 63, 9))
    hb = pedersen.te_handle(B, bowe_hopwood.CRH, ctx)
    bl = torch.from_numpy(np.random.default_rng(5).integers(0, 256, size=(n, 32), dtype=np.uint8)).to(dev)
    for _ in range(3):
        check(lib.akp_merkle_build_te_dev(hb.h, hb.h, bl.data_ptr(), n, 32, ln.data_ptr(), nl.data_ptr(), st))
        torch.cuda.synchronize()
    print("built")


def analyse(d):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    hot = [r for r in rows if any(k in r["Kernel_Name"] for k in ("poseidon_crh", "te_accumulate", "te_finalize", "te_crh_small", "te_serialize"))]
    for label, pick in (("Poseidon tree", lambda r: "poseidon_crh" in r["Kernel_Name"]), ("Bowe-Hopwood tree", lambda r: "te_" in r["Kernel_Name"])):
        ks = [r for r in hot if pick(r)]
        if not ks:
            continue
        # the last build = the trailing run of launches after the last gap of > 1 ms (the synchronize between builds)
        cut = 0
        for i in range(1, len(ks)):
            if int(ks[i]["Start_Timestamp"]) - int(ks[i - 1]["End_Timestamp"]) > 15_000:  # the synchronize between two builds
                cut = i
        ks = ks[cut:]
        print("== %s: last build, %d launches ==" % (label, len(ks)))
        tot_k = tot_g = nar_k = nar_g = 0.0
        prev_end = None
        for r in ks:
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            gap = 0.0 if prev_end is None else (s - prev_end) / 1e3
            dur = (e - s) / 1e3
            grid = int(r.get("Grid_Size", 0) or 0)
            wg = int(r.get("Workgroup_Size", 256) or 256)
            narrow = "coop" in r["Kernel_Name"] or "small" in r["Kernel_Name"]  # the latency kernels: levels of <= 2^15 / 2^14 nodes
            print("  %-34s grid %9d wg %4d  %9.1f us  gap %7.1f us" % (r["Kernel_Name"].split("(")[0][-34:], grid, wg, dur, gap))
            tot_k += dur
            tot_g += gap
            if narrow:
                nar_k += dur
                nar_g += gap
            prev_end = e
        print("  total: kernels %.1f us, gaps %.1f us; narrow levels (latency kernels): kernels %.1f us, gaps before them %.1f us" % (tot_k, tot_g, nar_k, nar_g))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "analyse":
        analyse(sys.argv[2])
    else:
        run()
