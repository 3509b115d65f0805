#!/usr/bin/env python3
"""bench.py -- headline measurement of the hot path on MI355X.

Metric (BASELINE.json): Poseidon-BLS12-381-Fr permutations/sec.  One "step" = one pass of the batched
permutation kernel over `--log2-states` (default 2^20, BASELINE configs[1]) synthetic sponge states of
t = 3 field elements (rate 2, alpha 17, R_F = 8, R_P = 31), inputs already resident in HBM.

Launch: `python bench.py --gpus N --steps K --warmup W`.  With N > 1 and no WORLD_SIZE in the environment the
script re-launches itself under `torch.distributed.run` with N ranks (one process per GPU, nccl = RCCL); under an
existing torchrun launch it uses the ranks it was given and asserts world size == --gpus.  Every rank permutes its
own 2^20 states (weak scaling, the batch shards with no data-path collective); `value` = states permuted by all
ranks / max-over-ranks time between barriers.

Output (rank 0): ONE short JSON line on stdout (tools/bench_legs/line.py: < 6 KB -- the contract keys, `roofline` and `cpu_baseline` with
scalar fields, `parity`, `curve_parity`, one scalar per side leg under `legs`, `"full": "bench_full.json"`) and the FULL record in
bench_full.json beside this file (AKP_BENCH_FULL overrides the path).  Objects of the full record:
  roofline      dominant kernel -- algorithmic bytes (192 B / permutation) / average launch duration measured with events
                on the launch stream, against the 8 TB/s HBM peak; `valu` gives the integer-ALU view (the real bound); `traffic` from the
                PMC passes of this round (`traffic_source`)
  parity        which kernel the probe exercised and how many states of the TIMED buffer were checked
  sustained     the same launch looped for >= `--sustain-seconds` (2^20 and 2^24 states)
  merkle        BASELINE config 3: MerkleTree::new over 2^24 Poseidon leaves (strong scaling over ranks; leaf shards + ONE all-gather)
  pedersen      BASELINE config 4: Pedersen 4x256 CRH over 2^20 x 128 B per GPU; `tables`: every figure from a FRESH handle, cold first
                batch and warm rate, for the library's default (cache-sized) table and the HBM-sized one (opt-in); roofline incl. the
                moved-bytes view (counter traffic, calibrated: profiles/r05_s6)
  bh_merkle     BASELINE config 5: Bowe-Hopwood 63x9 tree, 2^23 x 32 B leaves per GPU (2^26 on 8 GPUs); `tables` as above + ONE 2^26-leaf
                tree on one GPU from nothing
  ragged        batches whose items differ in length (one launch, lanes ordered by step count): Bowe-Hopwood, Pedersen, Poseidon
  proofs        SURVEY.md 8(f): batched generate_proof, Path::verify, generate_multi_proof (encoded on the device), MultiPath::verify,
                update_batch on an HBM-resident 2^20-leaf tree, Poseidon and Bowe-Hopwood (tools/bench_proofs.py)
  host_path     PCIe-inclusive rates of the host-pointer entry points (pageable and pinned buffers; the pinned curve-hash call is one
                gated launch), median wall time per call
  sweep         permutations/s and tree leaves/s at 2^20 .. 2^26 on one GPU (clocks settled per point)
  predicted_scaling  the 2 / 4 / 8-GPU tree build times the design implies from this run's one-GPU numbers (a model, not a measurement)
  curve_parity  pin status of the curve half of the oracle; `curve_parity_emitter`: what the cargo probe found on this box
  cpu_baseline  oracle C restatement ("port") on this host: 1 thread and the best thread count; `cores` is the EFFECTIVE core count;
                Pedersen and Bowe-Hopwood-tree legs beside the permutation (rank 0 at N = 1 only; null in the line otherwise)
The legs live in tools/bench_legs/ (one module per object); this file keeps the launch plumbing, the headline measurement and
the assembly of the record.  The oracle is used only as checker and as the `cpu_baseline` leg.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

from bench_legs.line import compact  # noqa: E402
from bench_legs.common import (ALGO_BYTES_PER_PERM, HBM_PEAK_GBS, MADS_PER_PERM, MODMUL_PER_PERM_REF, NOMINAL_SCLK_MHZ, PMC_TRAFFIC_BYTES_PER_PERM,  # noqa: E402
                               PMC_TRAFFIC_SOURCE, ClockProbe, Env, gpu_clock_mhz, gpu_sensors, measure_hbm_copy, valu_peak_wave_instr)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def curve_parity_status(attempt=False):
    """pin status of the curve half (Pedersen / Bowe-Hopwood digests, byte formats): `tests/golden/reference_vectors.json` is written
    by shim/examples/emit_vectors.rs on a machine with a Rust toolchain; without it the oracle those legs are checked against is a
    faithful but UNPINNED restatement.  tools/bench_legs/pin.py looks for `cargo` and runs the emitter when it can."""
    from bench_legs import pin
    return pin.status(attempt)[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log2-states", type=int, default=20)
    ap.add_argument("--merkle-log2", type=int, default=24, help="total leaves of the Poseidon Merkle leg (0 disables)")
    ap.add_argument("--pedersen-log2", type=int, default=20, help="Pedersen 4x256 messages per GPU (BASELINE config 4; 0 disables)")
    ap.add_argument("--bh-merkle-log2", type=int, default=23,
                    help="Bowe-Hopwood 63x9 tree: leaves PER GPU (BASELINE config 5 is 2^23 per GPU on 8 GPUs; 0 disables)")
    ap.add_argument("--proofs-log2", type=int, default=20, help="leaves of the HBM-resident trees of the proof / verify / update legs (0 disables)")
    ap.add_argument("--proofs-m-log2", type=int, default=16, help="paths per call of the proof / verify legs")
    ap.add_argument("--ragged-log2", type=int, default=20, help="items of the ragged (per-item length) batches (0 disables)")
    ap.add_argument("--sustain-seconds", type=float, default=3.0, help="length of each sustained loop (0 disables)")
    ap.add_argument("--sustain-log2-big", type=int, default=24, help="second sustained size (0 disables)")
    ap.add_argument("--settle-launches", type=int, default=120, help="untimed launches before the W warm-up steps (clock ramp)")
    ap.add_argument("--no-host-path", action="store_true")
    ap.add_argument("--no-sweep", action="store_true", help="skip the 2^20 .. 2^26 size sweep")
    ap.add_argument("--sweep-max-log2", type=int, default=26, help="largest point of the size sweep (2^20, 2^22, ... up to this)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target duration of the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--one-process-leg", type=int, default=0, help=argparse.SUPPRESS)  # internal: child process of the Merkle leg
    ap.add_argument("--one-process-bh", type=int, default=0, help=argparse.SUPPRESS)   # internal: its Bowe-Hopwood share per device
    args = ap.parse_args()
    if args.one_process_leg:
        from bench_legs import merkle as merkle_leg
        return merkle_leg.one_process_leg(args.one_process_leg, args.one_process_bh)

    # test hook (tests/test_gpu_bench_contract.py): AKP_BENCH_SHARED_GPU=1 puts every rank on GPU 0 and carries the
    # collectives over gloo, so the N > 1 code path can be exercised on a one-GPU box.  Never set by the driver.
    shared_gpu = os.environ.get("AKP_BENCH_SHARED_GPU") == "1"

    # ---- N > 1 from a plain `python bench.py --gpus N`: become the launcher of N ranks ------------------------------
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import torch
        ndev = torch.cuda.device_count()
        if ndev < args.gpus and not shared_gpu:
            raise SystemExit("bench.py --gpus %d: only %d HIP device(s) visible" % (args.gpus, ndev))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus, "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        raise SystemExit(subprocess.call(cmd, env=env))

    import numpy as np
    import torch  # imported before the product so both share one HIP runtime

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s)" % (args.gpus, world))
    if shared_gpu:
        local_rank = 0
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the product has no CPU path)")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d wants device %d, only %d visible" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    rank_devices = [local_rank]
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if shared_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        assert dist.get_world_size() == args.gpus, "world size %d != --gpus %d" % (dist.get_world_size(), args.gpus)
        gathered = [None] * world
        dist.all_gather_object(gathered, {"rank": rank, "device": local_rank, "name": torch.cuda.get_device_name(dev)})
        rank_devices = [g["device"] for g in gathered]

    import crypto_primitives_amd as cpa
    from crypto_primitives_amd import field
    from crypto_primitives_amd._lib import lib, check
    from crypto_primitives_amd.distributed import GpuPoseidonBackend, GpuTeBackend, build_sharded
    from bench_legs import bh_merkle as bh_leg, cpu_baseline as cpu_leg, host_path as host_leg, merkle as merkle_leg, pedersen as ped_leg, ragged as ragged_leg, scaling, sustained as sus_leg, sweep as sweep_leg

    ctx = cpa.default_context(local_rank)
    # curve tables: every leg runs with the LIBRARY DEFAULT (the cache-sized table: akp_ctx_set_table_budget 0); the pedersen and
    # bh_merkle legs measure the HBM-sized tables (AKP_TABLE_BUDGET_DEVICE, opt-in) beside it on handles of their own, cold and warm
    cfg = cpa.get_default_poseidon_parameters(2, False)
    ph = cfg.handle(ctx)
    n = 1 << args.log2_states
    t = cfg.t
    stream = torch.cuda.current_stream(dev).cuda_stream

    from bench_legs.runner import LegRunner
    runner = LegRunner(torch, dist, rank, world, "cpu" if shared_gpu else dev, sync_device=lambda: torch.cuda.synchronize(dev))


    def barrier():  # the timed region's barrier: the process group's own
        torch.cuda.synchronize(dev)
        if dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def max_over_ranks(x):
        if not dist:
            return x
        tt = torch.tensor([x], dtype=torch.float64, device=dev if not shared_gpu else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    env = Env()
    env.args, env.np, env.torch, env.cpa, env.field, env.lib, env.check = args, np, torch, cpa, field, lib, check
    env.dev, env.ctx, env.stream, env.cfg, env.ph = dev, ctx, stream, cfg, ph
    env.rank, env.world, env.local_rank, env.dist, env.shared_gpu = rank, world, local_rank, dist, shared_gpu
    env.barrier, env.max_over_ranks = runner.barrier, runner.max_over_ranks  # the legs' collectives carry the abort flag (bench_legs/runner.py)
    env.GpuPoseidonBackend, env.GpuTeBackend, env.build_sharded = GpuPoseidonBackend, GpuTeBackend, build_sharded
    # handles with HBM-sized tables stay alive until the run ends: freeing tens of GB in the middle of it would leave the driver wiping the
    # released memory (~35 GB/s) while the next leg allocates -- and an allocation made then waits for the wipe (profiles/r06_s2, r06_s20)
    env.keepalive = []
    env.ora_threads = max(1, min(32, (os.cpu_count() or 1)))
    env.bench_path = os.path.abspath(__file__)
    env.ora = None
    if rank == 0:
        from oracle import cref
        env.ora = cref.Poseidon(cfg.full_rounds, cfg.partial_rounds, cfg.alpha, cfg.rate, cfg.capacity, cfg.ark, cfg.mds)

    # synthetic states (seed per BASELINE.md config 2), resident in HBM before the timed region
    host_states = field.random_fr(n * t, seed=0xA5A50002 + rank).reshape(n, t, 4)
    d_states = torch.from_numpy(host_states.view(np.int64)).to(dev)

    def step():
        check(lib.akp_poseidon_permute_batch_dev(ph.h, d_states.data_ptr(), n, stream))

    # ================= side legs first: from an idle device the first launches run on ramping clocks =================
    # every leg through runner.run: whatever a leg raises becomes `leg_errors[name]` and the run goes on to the headline (bench_legs/runner.py)
    import gc

    def leg_cleanup():
        ctx.set_table_budget(0)
        gc.collect()
        torch.cuda.empty_cache()

    def proofs_leg():  # SURVEY.md 8(f) ranks 1-2 (+ rank 4: device-resident sponges); one process only
        if not (args.proofs_log2 and world == 1 and rank == 0):
            return None
        import bench_proofs
        proofs = {}
        for name in ("poseidon", "bh"):
            proofs[name] = bench_proofs.run(name, args.proofs_log2, min(args.proofs_m_log2, args.proofs_log2), local_rank)
            if not proofs[name]["all_parity_bit_exact"]:
                raise SystemExit("proofs leg (%s): a parity / control check failed: %s" % (name, json.dumps(proofs[name])))
        proofs["sponge"] = bench_proofs.run_sponge(args.proofs_log2, local_rank)
        if not proofs["sponge"]["sampled_parity_bit_exact"]:
            raise SystemExit("sponge leg: sampled squeezes differ from the oracle")
        proofs["profiles"] = "profiles/r04_s4/proofs_poseidon_kernel_stats_walk*.csv, profiles/r03_s6/proofs_* (rocprofv3 --kernel-trace --stats of `python tools/bench_proofs.py`)"
        return proofs
    merkle = runner.run("merkle", merkle_leg.run, env, cleanup=leg_cleanup)        # BASELINE configs[2]
    pedersen = runner.run("pedersen", ped_leg.run, env, cleanup=leg_cleanup)       # BASELINE configs[3]
    bh_merkle = runner.run("bh_merkle", bh_leg.run, env, cleanup=leg_cleanup)      # BASELINE configs[4]
    proofs = runner.run("proofs", proofs_leg, cleanup=leg_cleanup)
    ragged = runner.run("ragged", ragged_leg.run, env, cleanup=leg_cleanup)        # per-item lengths (the reference's per-leaf evaluate)
    sweep = runner.run("sweep", lambda: sweep_leg.run(env, tuple(lg for lg in (20, 22, 24, 26) if lg <= max(20, args.sweep_max_log2))) if world == 1 else None,
                       cleanup=leg_cleanup)

    # ================= the headline: W warm-up + K timed steps of the 2^20-state permutation ==========================
    parity = {"probe_kernel": lib.akp_poseidon_kernel_for(ph.h, n, 0).decode(), "timed_buffer_states_checked": 0, "bit_exact": None}
    step()  # one pass outside W (and outside K): its output is checked against the oracle -- the WHOLE timed buffer when this host has >= 8
    torch.cuda.synchronize(dev)  # cores (2^20 permutations take the C oracle ~1-3 s there), a strided sample of it otherwise
    if rank == 0:
        full_probe = (os.cpu_count() or 1) >= 8 and os.environ.get("AKP_BENCH_PROBE") != "sample"
        if full_probe:
            got = d_states.cpu().numpy().view(np.uint64).reshape(n, t, 4)
            exp = env.ora.permute_batch(host_states, threads=env.ora_threads).reshape(n, t, 4)
            checked = n
        else:
            si = np.unique(np.concatenate([np.arange(128), np.linspace(0, n - 1, 385).astype(np.int64), np.arange(n - 128, n)]))
            got = d_states[torch.from_numpy(si).to(dev)].cpu().numpy().view(np.uint64).reshape(len(si), t, 4)
            exp = env.ora.permute_batch(np.ascontiguousarray(host_states[si]), threads=env.ora_threads).reshape(len(si), t, 4)
            checked = len(si)
        parity["timed_buffer_states_checked"] = int(checked)
        parity["of_states"] = int(n)
        parity["bit_exact"] = bool(np.array_equal(got, exp))
        del got, exp
        if not parity["bit_exact"]:
            raise SystemExit("parity probe FAILED: the timed kernel's output differs from the oracle")
    # the oracle check above left the GPU idle for ~0.1 s and the clocks drop within milliseconds: settle them with untimed
    # launches (like the side legs, outside W and K) so that the W + K steps measure the steady state, not the ramp
    # (everything the timed region needs is prepared BEFORE the settle launches: reading the clock can cost a `rocm-smi`
    # subprocess -- hundreds of ms of idle device right before t0 would hand the first timed steps a ramping clock)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    probe = ClockProbe(env)  # effective shader clock: one extra wave beside the ~16 000 of each timed launch (its own stream)
    for _ in range(8):
        step()
    clk0 = gpu_clock_mhz(local_rank)  # sampled while launches are in flight
    for _ in range(args.settle_launches):
        step()
    # effective clock UNDER THIS LOAD, immediately before the timed region: three probes (3.6 ms each) beside twelve more
    # untimed launches -- outside W and K, nothing is added to the timed region itself
    for i in range(12):
        if i % 4 == 0:
            probe.launch()
        step()
    for _ in range(args.warmup):
        step()
    barrier()
    pre = probe.read()
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()
        step()
        b.record()
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    for i in range(12):  # ... and immediately after it, under the same load
        if i % 4 == 0:
            probe.launch()
        step()
    sens = gpu_sensors()  # while those launches are in flight
    torch.cuda.synchronize(dev)
    post = probe.read()
    clk1 = gpu_clock_mhz(local_rank)
    kern_ms = [a.elapsed_time(b) for a, b in ev]
    kern_avg_s = (sum(kern_ms) / len(kern_ms)) / 1e3
    around = pre + post
    eff_mhz = sum(p["mhz"] for p in around) / len(around) if around else None

    sustained = runner.run("sustained", sus_leg.run, env, d_states, n, kern_avg_s, cleanup=leg_cleanup)
    host_path = runner.run("host_path", host_leg.run, env, host_states, n, cleanup=leg_cleanup)

    if rank != 0:
        if dist:
            dist.barrier()
            dist.destroy_process_group()
        return

    from bench_legs import pin
    # AKP_RUN_EMITTER=1: run the reference-vector emitter (`cargo run`, writes tests/golden/reference_vectors.json) if this box has cargo + the
    # crates offline; without the switch only the presence of cargo and of the vectors is reported
    pin_status = pin.status(attempt=os.environ.get("AKP_RUN_EMITTER") == "1")
    total_perms = n * world * args.steps
    value = total_perms / elapsed
    achieved = ALGO_BYTES_PER_PERM * n / kern_avg_s / 1e9
    hbm_copy_gbs = measure_hbm_copy(torch, dev)
    mad_rate = MADS_PER_PERM * n / kern_avg_s
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
        "dtype": "i32",
        "dtype_detail": "255-bit Montgomery integers as 9 limbs of 29 bits in 32-bit registers, 64-bit v_mad_i64_i32 accumulation",
        "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: batched Poseidon permutation, BLS12-381 Fr, t=3 rate=2 alpha=17 RF=8 RP=31 "
                               "(default Grain-LFSR parameters), 2^%d states per GPU, in place in HBM" % args.log2_states,
                   "states_per_gpu": n, "parallelism": "shard%d (no data-path collective)" % world},
        "settle_launches_before_warmup": args.settle_launches,
        "launch": {"ranks": world, "backend": None if not dist else ("gloo (shared-GPU test hook)" if shared_gpu else "nccl (RCCL)"),
                   "rank_devices": rank_devices},
        "parity_probe_bit_exact": parity["bit_exact"],
        "parity": parity,
        "curve_parity": pin_status[0],
        "curve_parity_emitter": pin_status[1],
        "curve_tables": "every leg runs with the library default (the cache-sized table, akp_ctx_set_table_budget 0); pedersen.tables / bh_merkle.tables hold "
                        "cold-start and warm figures of the HBM-sized tables (AKP_TABLE_BUDGET_DEVICE, opt-in) beside it",
        "roofline": {"bound": "hbm", "kernel": parity["probe_kernel"], "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": PMC_TRAFFIC_BYTES_PER_PERM * n,
                     "traffic_measured_in_this_run": False, "traffic_source": PMC_TRAFFIC_SOURCE,
                     "traffic_static_from": PMC_TRAFFIC_SOURCE + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH x2 gfx950 "
                                            "correction; NOT measured in this run)",
                     "peak_measured_copy": hbm_copy_gbs, "frac_of_measured_copy": achieved / hbm_copy_gbs if hbm_copy_gbs else None,
                     "kernel_avg_ms": kern_avg_s * 1e3, "kernel_min_ms": min(kern_ms), "kernel_max_ms": max(kern_ms),
                     "algorithmic_bytes_per_launch": ALGO_BYTES_PER_PERM * n,
                     "effective_sclk_mhz": eff_mhz,
                     "effective_sclk": {"before_timed_steps_mhz": [round(p["mhz"], 1) for p in pre], "after_timed_steps_mhz": [round(p["mhz"], 1) for p in post],
                                        "cycles_per_dependent_mad": around[0]["cycles_per_dependent_mad"] if around else None,
                                        "method": "akp_clock_probe_dev: one wave on a side stream runs 2^20 dependent v_mad_u64_u32 beside UNTIMED launches of the "
                                                  "same kernel immediately before and after the K timed steps; MHz = 100 x d(s_memtime) / d(s_memrealtime) "
                                                  "(shader-clock cycles over the constant 100 MHz clock); `effective_sclk_mhz` is their mean"},
                     "sclk_level_mhz_before": clk0, "sclk_level_mhz_after": clk1,
                     "sclk_level_note": "the DPM level sysfs reports: a ceiling of the power state, not the effective clock",
                     "power_w_under_load": sens["power_w"], "power_cap_w": sens["power_cap_w"], "temp_c_max": sens["temp_c_max"], "sensors_source": sens["source"],
                     "valu": {"note": "the path is integer-ALU bound (~%d reference-shaped Montgomery products per 192 B); "
                                      "fraction of the v_mad issue peak spent on multiplies" % MODMUL_PER_PERM_REF,
                              "ref_modmul_per_s": MODMUL_PER_PERM_REF * n / kern_avg_s,
                              "v_mad_per_s": mad_rate,
                              "v_mad_peak_per_s": valu_peak_wave_instr(eff_mhz or NOMINAL_SCLK_MHZ) * 64,
                              "v_mad_peak_basis": "1024 SIMDs x effective_sclk_mhz / 4 cycles per wave-instruction x 64 lanes" if eff_mhz else
                                                  "effective clock unavailable: nominal 2400 MHz",
                              "frac_of_mad_issue_peak": mad_rate / (valu_peak_wave_instr(eff_mhz or NOMINAL_SCLK_MHZ) * 64),
                              "frac_of_mad_issue_peak_at_nominal_2400mhz": mad_rate / (valu_peak_wave_instr() * 64),
                              "valu_instructions_per_permutation": 1224736768 // 16384, "valu_busy_percent": 97.8,
                              "valu_counters_measured_in_this_run": False,
                              "valu_counters_static_from": "profiles/r06_s25/pmc_poseidon.txt (SQ_INSTS_VALU / 16384 waves; VALUBusy 95.8-95.9 on that box, 97.8-97.9 in profiles/r05_s7, 93.6-96.1 in rounds 3-4; NOT measured in this run)"}},
    }
    for key, leg in (("sustained", sustained), ("merkle", merkle), ("pedersen", pedersen), ("bh_merkle", bh_merkle), ("proofs", proofs), ("host_path", host_path),
                     ("ragged", ragged), ("sweep", sweep)):
        if leg:
            out[key] = leg

    def predicted():
        if world == 1:
            return scaling.predict(merkle, bh_merkle)
        if args.merkle_log2 == 24 and args.bh_merkle_log2 == 23:
            # N > 1: the model calibrated on the committed one-GPU line (profiles/r04_s14/bench.json), so that THIS run's measured tree times
            # stand next to what the design implied for them
            ref = scaling.predict({"leaves": 1 << 24, "seconds": 0.0656}, {"leaves_per_gpu": 1 << 23, "seconds": 0.0170})
            key = "%d_gpus" % world
            return {"calibrated_on": "profiles/r04_s14/bench.json (one GPU: Poseidon 2^24 leaves 0.0656 s, Bowe-Hopwood 2^23 leaves 0.0170 s)",
                    "model": ref["model"],
                    "merkle_strong": {"predicted": ref["merkle_strong"].get(key), "measured_seconds": merkle["seconds"] if merkle else None},
                    "bh_merkle_weak": {"predicted": ref["bh_merkle_weak"].get(key), "measured_seconds": bh_merkle["seconds"] if bh_merkle else None}}
        return None
    # rank 0 only from here on (the other ranks wait in the closing barrier): no collectives, so a failure cannot desynchronise anything
    solo = LegRunner(torch, None, 0, 1, "cpu")
    ps = solo.run("predicted_scaling", predicted)
    if ps:
        out["predicted_scaling"] = ps
    if not args.no_cpu_baseline and world == 1:
        solo.run("cpu_baseline", cpu_leg.run, env, out, host_states, n, value, pedersen, bh_merkle)
    errors = dict(runner.errors, **solo.errors)
    out["legs_failed"] = sorted(errors)
    out["leg_errors"] = errors
    # the full record goes to a file; stdout gets ONE short line (tools/bench_legs/line.py: < 6 KB, the driver parses its last line)
    full_path = os.environ.get("AKP_BENCH_FULL", os.path.join(ROOT, "bench_full.json"))
    with open(full_path, "w") as fh:
        json.dump(out, fh, indent=1)
    sys.stdout.flush()
    print(json.dumps(compact(out, os.path.basename(full_path))), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
