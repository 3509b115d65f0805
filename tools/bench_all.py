#!/usr/bin/env python3
"""bench_all.py -- throughput of every kernel family at the BASELINE.json configurations (1 GPU),
inputs resident in HBM, with a bounded CPU-baseline sample (oracle C restatement) beside each.
Prints one JSON object per line.  Evidence for DESIGN.md section 6."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import crypto_primitives_amd as cpa  # noqa: E402
from crypto_primitives_amd import field, params  # noqa: E402
from crypto_primitives_amd._lib import lib, check  # noqa: E402
from crypto_primitives_amd.crh import pedersen, bowe_hopwood  # noqa: E402
from oracle import cref  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
ctx = cpa.default_context(0)
STREAM = torch.cuda.current_stream(dev).cuda_stream


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / 1e3)
    return best


def dev_bytes(arr):
    return torch.from_numpy(np.ascontiguousarray(arr).view(np.uint8).reshape(-1)).to(dev)


def emit(**kw):
    print(json.dumps(kw), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--max-log2", type=int, default=24)
    ap.add_argument("--cpu-seconds", type=float, default=4.0)
    args = ap.parse_args()
    threads = cref.hardware_threads()
    cfg = cpa.get_default_poseidon_parameters(2, False)
    ph = cfg.handle(ctx)
    ora = cref.Poseidon(cfg.full_rounds, cfg.partial_rounds, cfg.alpha, cfg.rate, cfg.capacity, cfg.ark, cfg.mds)

    # ---- Poseidon permutation sweep (config 2: 2^20; sweep to 2^max) ----
    big = field.random_fr((1 << args.max_log2) * 3, seed=1).reshape(-1, 3, 4)
    d_big = dev_bytes(big)
    for lg in range(10, args.max_log2 + 1, 2):
        n = 1 << lg
        s = timed(lambda: check(lib.akp_poseidon_permute_batch_dev(ph.h, d_big.data_ptr(), n, STREAM)))
        emit(kernel="poseidon_permute_t3", log2n=lg, seconds=s, per_s=n / s, algorithmic_GBps=192 * n / s / 1e9)
    # ---- Poseidon CRH (config 1 shape: 2 Fr per input) and two-to-one ----
    n = 1 << 20
    d_out = torch.empty(n * 4, dtype=torch.int64, device=dev)
    for k in (1, 2, 3):
        s = timed(lambda: check(lib.akp_poseidon_crh_batch_dev(ph.h, d_big.data_ptr(), n, k, d_out.data_ptr(), STREAM)))
        emit(kernel="poseidon_crh", log2n=20, elems_per_input=k, seconds=s, per_s=n / s)
    # ---- Poseidon Merkle (config 3) ----
    for lg in (16, 20, min(24, args.max_log2)):
        n = 1 << lg
        ln = torch.empty(n * 4, dtype=torch.int64, device=dev)
        nl = torch.empty(n * 4, dtype=torch.int64, device=dev)
        s = timed(lambda: check(lib.akp_merkle_build_poseidon_dev(ph.h, ph.h, d_big.data_ptr(), n, 1, ln.data_ptr(), nl.data_ptr(), STREAM)))
        emit(kernel="merkle_poseidon", log2n=lg, seconds=s, leaves_per_s=n / s, algorithmic_GBps=160 * n / s / 1e9)
    cal = big[: 1 << 14, :1].copy()
    c0 = time.perf_counter(); ora.merkle_build(ora, cal, 1, threads=threads); cs = time.perf_counter() - c0
    emit(kernel="merkle_poseidon", cpu_baseline=True, cores=threads, log2n=14, seconds=cs, leaves_per_s=(1 << 14) / cs)
    del d_big

    # ---- Pedersen 4x256 (config 4) ----
    t0 = time.perf_counter()
    gens = params.pedersen_generators(0xA5A50004, 4, 256)
    P = pedersen.Parameters(gens)
    hP = P.handle(ctx)
    torch.cuda.synchronize()
    emit(kernel="pedersen_setup_4x256", seconds=time.perf_counter() - t0, note="host generators + device LUT build")
    n = 1 << 20
    rng = np.random.default_rng(4)
    msgs = rng.integers(0, 256, size=(n, 128), dtype=np.uint8)
    d_msgs = dev_bytes(msgs)
    d_o = torch.empty(n * 8, dtype=torch.int64, device=dev)
    for L in (128, 32):
        d_m = dev_bytes(msgs[:, :L])
        s = timed(lambda: check(lib.akp_te_crh_batch_dev(hP.h, d_m.data_ptr(), n, L, d_o.data_ptr(), STREAM)))
        emit(kernel="pedersen_crh_4x256", log2n=20, msg_len=L, seconds=s, per_s=n / s, algorithmic_GBps=(L + 64) * n / s / 1e9)
    CP = cref.CurveParams(4, 256, gens)
    m = 1 << 12
    c0 = time.perf_counter(); CP.pedersen_crh_batch(msgs[:m], m, 128, threads=threads); cs = time.perf_counter() - c0
    emit(kernel="pedersen_crh_4x256", cpu_baseline=True, cores=threads, n=m, seconds=cs, per_s=m / cs)

    # ---- Bowe-Hopwood 63x9 (config 5 hashes) + byte-leaf trees ----
    gb = params.bowe_hopwood_generators(0xA5A50005, 63, 9)
    B = bowe_hopwood.Parameters(gb)
    hB = B.handle(ctx)
    for L in (32, 70):
        d_m = dev_bytes(msgs[:, :L])
        s = timed(lambda: check(lib.akp_te_crh_batch_dev(hB.h, d_m.data_ptr(), n, L, d_o.data_ptr(), STREAM)))
        emit(kernel="bowe_hopwood_crh_63x9", log2n=20, msg_len=L, seconds=s, per_s=n / s)
    BP = cref.CurveParams(63, 9, gb)
    c0 = time.perf_counter(); BP.bh_crh_batch(msgs[:m, :32], m, 32, threads=threads); cs = time.perf_counter() - c0
    emit(kernel="bowe_hopwood_crh_63x9", cpu_baseline=True, cores=threads, n=m, msg_len=32, seconds=cs, per_s=m / cs)
    for lg in (16, 20, 22):
        nn = 1 << lg
        leaves = rng.integers(0, 256, size=(nn, 32), dtype=np.uint8)
        d_l = dev_bytes(leaves)
        ln = torch.empty(nn * 4, dtype=torch.int64, device=dev)
        nl = torch.empty(nn * 4, dtype=torch.int64, device=dev)
        s = timed(lambda: check(lib.akp_merkle_build_te_dev(hB.h, hB.h, d_l.data_ptr(), nn, 32, ln.data_ptr(), nl.data_ptr(), STREAM)), reps=2)
        emit(kernel="merkle_bowe_hopwood_63x9", log2n=lg, seconds=s, leaves_per_s=nn / s, algorithmic_GBps=160 * nn / s / 1e9)
    nn = 1 << 16
    leaves = rng.integers(0, 256, size=(nn, 32), dtype=np.uint8)
    d_l = dev_bytes(leaves)
    ln = torch.empty(nn * 8, dtype=torch.int64, device=dev)
    nl = torch.empty(nn * 8, dtype=torch.int64, device=dev)
    s = timed(lambda: check(lib.akp_merkle_build_te_dev(hP.h, hP.h, d_l.data_ptr(), nn, 32, ln.data_ptr(), nl.data_ptr(), STREAM)), reps=2)
    emit(kernel="merkle_pedersen_4x256", log2n=16, seconds=s, leaves_per_s=nn / s)
    c0 = time.perf_counter(); BP.merkle_build(1, BP, leaves[:4096], 4096, 32, threads=threads); cs = time.perf_counter() - c0
    emit(kernel="merkle_bowe_hopwood_63x9", cpu_baseline=True, cores=threads, log2n=12, seconds=cs, leaves_per_s=4096 / cs)


if __name__ == "__main__":
    main()
