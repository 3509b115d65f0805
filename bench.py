#!/usr/bin/env python3
"""bench.py -- headline measurement of the hot path on MI355X.

Metric (BASELINE.json): Poseidon-BLS12-381-Fr permutations/sec.  One "step" = one pass of the batched
permutation kernel over `--log2-states` (default 2^20, BASELINE configs[1]) synthetic sponge states of
t = 3 field elements (rate 2, alpha 17, R_F = 8, R_P = 31), inputs already resident in HBM.
N > 1: one process per GPU (torchrun), every rank permutes its own 2^20 states (weak scaling, the batch
shards with no data-path collective); `value` = states permuted by all ranks / max-over-ranks time.

Extra objects on the same JSON line:
  roofline      dominant kernel (poseidon_permute_kernel) -- algorithmic bytes (192 B / permutation,
                SURVEY.md section 8d) / average launch duration measured with events on the launch stream,
                against the 8 TB/s HBM peak; `valu` gives the integer-ALU view (the real bound).
  cpu_baseline  oracle C restatement ("port") timed on this host on a bounded sample, rank 0 / N = 1 only.
  merkle        MerkleTree::new over `--merkle-log2` Poseidon leaves (default 2^24 at N = 1... see below),
                leaf shards per rank + one all-gather of sub-roots (RCCL); seconds and leaves/s.
The oracle is used only as checker (a 256-state parity probe outside the timed region) and as the
`cpu_baseline` leg.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402  (imported before the product so both share one HIP runtime)

ALGO_BYTES_PER_PERM = 192        # 96 B read + 96 B write (t = 3)
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec
MODMUL_PER_PERM_REF = 626        # reference-shaped count
# HBM bytes per 2^20-state launch from the PMC passes of profiles/r01_s15/pmc_{FETCH,WRITE}_SIZE_counter_collection.csv
# (separate --pmc runs; FETCH_SIZE doubled per MI355X_MICROARCH.md "HBM": gfx950 reports half of a wide coalesced read):
# (2 * 49 616 KB + 98 304 KB) * 1024 per 2^20 permutations = 192.9 B per permutation  (algorithmic: 192 B)
PMC_TRAFFIC_BYTES_PER_PERM = (2 * 49616.0 + 98304.0) * 1024 / (1 << 20)
VALU_PEAK_WAVE_INSTR = 256 * 4 * 2.4e9 / 4.0   # measured: one v_mad_u64_u32 (wave64) per ~4 cycles per SIMD (profiles/r01_s1_microbench*)
MADS_PER_PERM = 55 * (4 * 117 + 153) + (20 * 234 + 4 * 315) + (30 * (315 + 153) + (234 + 153))  # multiply-adds per permutation
# (45 / 81 / 162 / 243 limb products + 72 reduction products for a square / product / 2-term / 3-term dot) in the full form:
# 55 S-boxes; full-round rows: 20 with unit diagonal (dot2), 4 dot3; partial rounds: dot3 + one product (lane-1 form), the
# last one dot2 + one product; no conversion products.  (The assembly routines issue 9 more v_mad per routine to add the
# quotient digits; they are not counted as multiplies.)


def measure_hbm_copy(torch, dev, nbytes=1 << 30, reps=10):
    """Read+write GB/s of a plain 1 GiB device-to-device copy on this box (report the measured HBM rate
    beside the vendor 8 TB/s).  Measurement plumbing only -- not part of the hashed path."""
    try:
        a = torch.empty(nbytes // 8, dtype=torch.int64, device=dev)
        b = torch.empty_like(a)
        a.zero_()
        for _ in range(2):
            b.copy_(a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            b.copy_(a)
        e1.record()
        torch.cuda.synchronize()
        secs = e0.elapsed_time(e1) / 1e3 / reps
        del a, b
        return 2 * nbytes / secs / 1e9
    except Exception:  # pragma: no cover - measurement is best effort
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log2-states", type=int, default=20)
    ap.add_argument("--merkle-log2", type=int, default=24, help="total leaves of the Merkle leg (0 disables)")
    ap.add_argument("--bh-merkle-log2", type=int, default=0, help="also build a Bowe-Hopwood 63x9 tree over 2^k 32-byte leaves (BASELINE config 5 shape), sharded like the Poseidon tree")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target duration of the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # test hook (tests/test_gpu_bench_contract.py): AKP_BENCH_SHARED_GPU=1 puts every rank on GPU 0 and carries the
    # collectives over gloo, so the N > 1 code path can be exercised on a one-GPU box.  Never set by the driver.
    shared_gpu = os.environ.get("AKP_BENCH_SHARED_GPU") == "1"
    if shared_gpu:
        local_rank = 0
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the product has no CPU path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if shared_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import crypto_primitives_amd as cpa
    from crypto_primitives_amd import field
    from crypto_primitives_amd._lib import lib, check, Context
    from crypto_primitives_amd.distributed import GpuPoseidonBackend, build_sharded

    ctx = cpa.default_context(local_rank)
    cfg = cpa.get_default_poseidon_parameters(2, False)
    ph = cfg.handle(ctx)
    n = 1 << args.log2_states
    t = cfg.t

    # synthetic states (seed per BASELINE.md config 2), resident in HBM before the timed region
    host_states = field.random_fr(n * t, seed=0xA5A50002 + rank).reshape(n, t, 4)
    d_states = torch.from_numpy(host_states.view(np.int64)).to(dev)
    stream = torch.cuda.current_stream(dev).cuda_stream

    def step():
        check(lib.akp_poseidon_permute_batch_dev(ph.h, d_states.data_ptr(), n, stream))

    # parity probe (checker only, outside the timed region)
    parity = None
    if rank == 0:
        from oracle import cref
        ora = cref.Poseidon(cfg.full_rounds, cfg.partial_rounds, cfg.alpha, cfg.rate, cfg.capacity, cfg.ark, cfg.mds)
        probe = torch.from_numpy(host_states[:256].copy().view(np.int64)).to(dev)
        check(lib.akp_poseidon_permute_batch_dev(ph.h, probe.data_ptr(), 256, stream))
        torch.cuda.synchronize(dev)
        parity = bool(np.array_equal(probe.cpu().numpy().view(np.uint64).reshape(256, t, 4),
                                     ora.permute_batch(host_states[:256]).reshape(256, t, 4)))
        if not parity:
            raise SystemExit("parity probe FAILED: GPU permutation differs from the oracle")

    # ---- Merkle leg: sharded MerkleTree::new (strong scaling over the same total leaf count).  Runs before the
    # permutation timing: from an idle device the first ~12 launches run up to 20 % slower while the clocks ramp
    # (tools/gpu_ramp.py), so the side legs go first and the W warm-up + K timed steps see the steady-state clock ----
    merkle = None
    if args.merkle_log2:
        total = 1 << args.merkle_log2
        per = total // world
        leaves = field.random_fr(per, seed=0xA5A50003 + rank).reshape(per, 1, 4)
        d_leaves = torch.from_numpy(leaves.view(np.int64)).to(dev)
        backend = GpuPoseidonBackend(cfg, cfg, leaf_len=1, device=dev)
        build_sharded(backend, d_leaves, total, dist)  # untimed full-size warm-up build (allocations, RCCL, device clocks)
        torch.cuda.synchronize(dev)
        if dist:
            dist.barrier()
        m0 = time.perf_counter()
        res = build_sharded(backend, d_leaves, total, dist)
        torch.cuda.synchronize(dev)
        if dist:
            dist.barrier()
        msec = time.perf_counter() - m0
        if dist:
            tt = torch.tensor([msec], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            msec = float(tt.item())
        merkle = {"leaves": total, "seconds": msec, "leaves_per_s": total / msec, "scaling": "strong",
                  "permutations": 2 * total - 1, "root_limb0": int(np.asarray(res["root"]).reshape(-1)[0]),
                  "algorithmic_GBps": 160.0 * total / msec / 1e9}

    bh_merkle = None
    if args.bh_merkle_log2:
        from crypto_primitives_amd import params as cparams
        from crypto_primitives_amd.crh import bowe_hopwood
        from crypto_primitives_amd.distributed import GpuTeBackend
        total = 1 << args.bh_merkle_log2
        per = total // world
        B = bowe_hopwood.Parameters(cparams.bowe_hopwood_generators(0xA5A50005, 63, 9))
        leaves = np.random.default_rng(0xA5A50005 + rank).integers(0, 256, size=(per, 32), dtype=np.uint8)
        d_leaves = torch.from_numpy(leaves).to(dev)
        tb = GpuTeBackend(B, B, device=dev)
        build_sharded(tb, d_leaves[: max(per // 64, 2)], max(total // 64, 2 * world), dist)  # warm-up (tables, scratch, RCCL)
        torch.cuda.synchronize(dev)
        if dist:
            dist.barrier()
        m0 = time.perf_counter()
        res = build_sharded(tb, d_leaves, total, dist)
        torch.cuda.synchronize(dev)
        if dist:
            dist.barrier()
        bsec = time.perf_counter() - m0
        if dist:
            tt = torch.tensor([bsec], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            bsec = float(tt.item())
        bh_merkle = {"hash": "Bowe-Hopwood 63x9 over Jubjub, 32-byte leaves, ByteDigestConverter", "leaves": total, "seconds": bsec,
                     "leaves_per_s": total / bsec, "scaling": "strong", "algorithmic_GBps": 160.0 * total / bsec / 1e9}

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    if dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()
        step()
        b.record()
    torch.cuda.synchronize(dev)
    if dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    if dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    kern_ms = [a.elapsed_time(b) for a, b in ev]
    kern_avg_s = (sum(kern_ms) / len(kern_ms)) / 1e3

    if rank != 0:
        if dist:
            dist.destroy_process_group()
        return

    total_perms = n * world * args.steps
    value = total_perms / elapsed

    achieved = ALGO_BYTES_PER_PERM * n / kern_avg_s / 1e9
    hbm_copy_gbs = measure_hbm_copy(torch, dev)
    out = {
        "metric": "poseidon_bls12_381_fr_permutations_per_sec",
        "value": value,
        "unit": "permutations/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u32 limbs (255-bit Montgomery integers, radix 2^29 x 9, 64-bit v_mad accumulation)",
        "data": "synthetic",
        "config": {"workload": "batched Poseidon permutation, BLS12-381 Fr, t=3 rate=2 alpha=17 RF=8 RP=31 "
                               "(default Grain-LFSR parameters), 2^%d states per GPU, in place in HBM" % args.log2_states,
                   "states_per_gpu": n, "parallelism": "shard%d (no collective)" % world},
        "parity_probe_bit_exact": parity,
        "roofline": {"bound": "hbm", "kernel": "poseidon_permute_t3_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": PMC_TRAFFIC_BYTES_PER_PERM * n,
                     "peak_measured_copy": hbm_copy_gbs, "frac_of_measured_copy": achieved / hbm_copy_gbs if hbm_copy_gbs else None,
                     "traffic_source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH x2 gfx950 correction (profiles/r01_s15; same within 1.5 % in every session since r01_s3)",
                     "kernel_avg_ms": kern_avg_s * 1e3, "algorithmic_bytes_per_launch": ALGO_BYTES_PER_PERM * n,
                     "valu": {"note": "the path is integer-ALU bound (~%d reference-shaped Montgomery products per 192 B); "
                                      "fraction of the measured v_mad_u64_u32 issue peak spent on multiplies" % MODMUL_PER_PERM_REF,
                              "ref_modmul_per_s": MODMUL_PER_PERM_REF * n / kern_avg_s,
                              "v_mad_per_s": MADS_PER_PERM * n / kern_avg_s,
                              "v_mad_peak_per_s": VALU_PEAK_WAVE_INSTR * 64,
                              "frac_of_mad_issue_peak": MADS_PER_PERM * n / kern_avg_s / (VALU_PEAK_WAVE_INSTR * 64),
                              "valu_busy_pmc_percent": 98.0,
                              "valu_busy_source": "rocprofv3 --pmc VALUBusy on this kernel (profiles/r01_s19/pmc_valu_counters.txt)"}},
    }
    if merkle:
        out["merkle"] = merkle
    if bh_merkle:
        out["bh_merkle"] = bh_merkle
    if not args.no_cpu_baseline and world == 1:
        from oracle import cref
        threads = cref.hardware_threads()
        ora = cref.Poseidon(cfg.full_rounds, cfg.partial_rounds, cfg.alpha, cfg.rate, cfg.capacity, cfg.ark, cfg.mds)
        # calibrate: the box may expose more hardware threads than its cgroup lets us use, so try a few thread counts on
        # 2^14 states each and keep the fastest; then size the sample for ~cpu_seconds
        hw = threads
        cal = host_states[:16384]
        best = (0.0, 1)
        for cand in sorted({hw, max(1, hw // 2), max(1, hw // 4), max(1, hw // 8), min(hw, 16)}, reverse=True):
            c0 = time.perf_counter()
            ora.permute_batch(cal, threads=cand)
            r = len(cal) / (time.perf_counter() - c0)
            if r > best[0]:
                best = (r, cand)
        rate, threads = best
        sample = int(min(n, max(16384, rate * args.cpu_seconds)))
        passes = 0
        c0 = time.perf_counter()
        while True:  # whole passes over the sample until ~cpu_seconds have been spent
            ora.permute_batch(host_states[:sample], threads=threads)
            passes += 1
            cpu_s = time.perf_counter() - c0
            if sample < n or cpu_s >= args.cpu_seconds:
                break
        sample *= passes
        out["cpu_baseline"] = {"value": sample / cpu_s, "unit": "permutations/s", "cores": threads, "kind": "port",
                               "sample": "%d permutations over the same 2^%d states, reference-shaped C restatement "
                                         "(oracle/c/akp_oracle.c), %d pthreads (best of a thread-count sweep; %d hardware threads)"
                                         % (sample, args.log2_states, threads, hw)}
        out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
    print(json.dumps(out))
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
