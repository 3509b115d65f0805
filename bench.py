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

Extra objects on the same JSON line (rank 0):
  roofline      dominant kernel -- algorithmic bytes (192 B / permutation) / average launch
                duration measured with events on the launch stream, against the 8 TB/s HBM peak; `valu` gives the
                integer-ALU view (the real bound).  Fields copied from earlier profiling sessions say so
                (`static_from`).
  parity        which kernel the probe exercised and how many states of the TIMED buffer were checked
  sustained     the same launch looped for >= `--sustain-seconds` (2^20 and 2^24 states): rate, min / median / max
                launch time, GPU clock read from sysfs before / after
  merkle        BASELINE config 3: MerkleTree::new over 2^24 Poseidon leaves (strong scaling over ranks; leaf
                shards + ONE all-gather of sub-roots over RCCL)
  pedersen      BASELINE config 4: Pedersen 4x256 CRH over 2^20 x 128 B per GPU, sampled oracle parity, roofline
  bh_merkle     BASELINE config 5: Bowe-Hopwood 63x9 tree, 2^23 x 32 B leaves per GPU (2^26 on 8 GPUs)
  proofs        SURVEY.md 8(f) ranks 1-2 as the reference benches them (benches/merkle_tree.rs:60-191): batched generate_proof,
                Path::verify, generate_multi_proof + MultiPath::verify and update_batch on an HBM-resident 2^20-leaf tree, Poseidon
                and Bowe-Hopwood configurations (tools/bench_proofs.py), items/s + device ms + sampled oracle parity
  host_path     PCIe-inclusive rates of the host-pointer entry points (pageable and pinned buffers)
  cpu_baseline  oracle C restatement ("port") on this host: 1 thread and the best thread count; `cores` is the EFFECTIVE core
                count (min of affinity, cgroup quota, hardware threads); Pedersen and Bowe-Hopwood-tree legs beside the permutation
The oracle is used only as checker and as the `cpu_baseline` leg.
"""
import argparse
import glob
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_PERM = 192        # 96 B read + 96 B write (t = 3)
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec
MODMUL_PER_PERM_REF = 626        # reference-shaped count
# HBM bytes per permutation from PMC passes of an earlier session (NOT measured in this run; see `static_from`):
# (2 * 49 583.19 KB + 98 304 KB) * 1024 per 2^20 permutations = 192.8 B  (algorithmic: 192 B); re-collected in round 3 on the
# four-unit library: FETCH_SIZE 49 583 KB, WRITE_SIZE 98 304 KB, SQ_INSTS_VALU 1 224 736 768, VALUBusy 95.6-96.1 %
PMC_TRAFFIC_BYTES_PER_PERM = (2 * 49583.1875 + 98304.0) * 1024 / (1 << 20)
PMC_TRAFFIC_SOURCE = "profiles/r03_s11/pmc_counters_poseidon.txt"
VALU_PEAK_WAVE_INSTR = 256 * 4 * 2.4e9 / 4.0   # one v_mad (wave64) per ~4 cycles per SIMD (profiles/r01_s1_microbench*)
# curve-hash kernels, per 2^20-hash launch, KB / instructions (profiles/r03_s4/pmc_te_line128.txt: rocprofv3 --pmc, one counter per pass;
# NOT measured in this run).  FETCH x 2 as MI355X_MICROARCH.md prescribes for gfx950 (128-byte requests tallied at 64 B): with it the
# accumulate kernels fetch ~1.07 x the table bytes they gather -- every table line comes from the Infinity Cache / HBM, the L2 only
# serves the second half of a line (TCC_HIT = TCC_MISS: two 64-byte requests per 128-byte entry, the first misses, the second hits).
PMC_TE = {"source": "profiles/r03_s4/pmc_te_line128.txt",
          "pedersen_128B": {"fetch_kb": 4520651 + 164359, "write_kb": 147466 + 114688, "valu_instr": 1531920384 + 47370240},  # accumulate<2> + finalize<0>
          "bh_32B": {"fetch_kb": 957788 + 163841, "write_kb": 147473 + 81920, "valu_instr": 421838848 + 38817792},          # accumulate<1> + finalize<1>
          "bh_70B": {"fetch_kb": 2267504 + 163841, "write_kb": 147473 + 81920, "valu_instr": 936509440 + 38817792, "steps": 39}}
MADS_PER_PRODUCT = 153           # multiply-adds of one field product (81 limb products + 72 reduction products)


def te_counters(key, hashes, steps_scale=1.0):
    """bytes / instructions for `hashes` hashes from the per-2^20 PMC figures (gather-proportional parts scaled by steps_scale)"""
    c = PMC_TE[key]
    per = hashes / float(1 << 20)
    return {"traffic": (2.0 * c["fetch_kb"] * steps_scale + c["write_kb"]) * 1024.0 * per, "valu_instr": c["valu_instr"] * steps_scale * per}


MADS_PER_PERM = 55 * (4 * 117 + 153) + (20 * 234 + 4 * 315) + (30 * (315 + 153) + (234 + 153))  # multiply-adds per permutation
# (45 / 81 / 162 / 243 limb products + 72 reduction products for a square / product / 2-term / 3-term dot) in the full form:
# 55 S-boxes; full-round rows: 20 with unit diagonal (dot2), 4 dot3; partial rounds: dot3 + one product (lane-1 form), the
# last one dot2 + one product; no conversion products.


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def gpu_clock_mhz(index=0):
    """current shader clock: the level pp_dpm_sclk marks with '*' (highest over the cards that expose one -- a box may list an
    idle integrated device first), else `rocm-smi --showclocks --json`; None when unreadable"""
    best = None
    try:
        for path in sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk")):
            for line in open(path).read().splitlines():
                if line.strip().endswith("*"):
                    v = float(line.split(":")[1].strip().split("M")[0])
                    best = v if best is None else max(best, v)
    except Exception:
        pass
    if best is not None and best > 200.0:
        return best
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--json"], capture_output=True, text=True, timeout=10).stdout
        vals = []
        for card in json.loads(out).values():
            for k, v in card.items():
                if "sclk" in k.lower() and "(" in str(v):
                    vals.append(float(str(v).split("(")[1].split("M")[0]))
        if vals:
            return max(vals + ([best] if best else []))
    except Exception:
        pass
    return best


def gpu_power_w():
    """average socket power of the busiest card, watts (hwmon power1_average, microwatts); None when unreadable"""
    best = None
    try:
        for path in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average"):
            v = float(open(path).read().strip()) / 1e6
            best = v if best is None else max(best, v)
    except Exception:
        pass
    return best


def cpu_info():
    model = None
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if q == "max" else float(q) / float(per)
    except Exception:
        pass
    try:
        aff = len(os.sched_getaffinity(0))
    except Exception:
        aff = None
    return {"cpu_model": model, "logical_cpus": os.cpu_count(), "affinity_cpus": aff, "cgroup_cpu_quota": quota}


def measure_hbm_copy(torch, dev, nbytes=1 << 30, reps=10):
    """Read+write GB/s of a plain 1 GiB device-to-device copy on this box (the measured HBM rate beside the
    vendor 8 TB/s).  Measurement plumbing only -- not part of the hashed path."""
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


def one_process_leg(log2_leaves, bh_log2_per_gpu=0):
    """Child process of the Merkle leg: the same 2^k-leaf Poseidon tree -- and the Bowe-Hopwood tree of BASELINE configs[4] at
    2^j leaves per device -- through the C ABI's single-process multi-device entry points (what a Rust host calls): all visible
    GPUs (a power of two, at most 8), leaves in pageable host memory, one host thread per device, RCCL all-gather of the
    sub-roots inside libakp.so.  PCIe-inclusive, with the per-phase breakdown the library records (akp_multi_last_phases), so
    that the first run on a multi-GPU node yields copy-in + sub-tree / all-gather / top / copy-out, not one number.  Prints one
    JSON line."""
    import numpy as np
    import torch
    import crypto_primitives_amd as cpa
    from crypto_primitives_amd import field
    g = 1
    while g * 2 <= min(torch.cuda.device_count(), 8):
        g *= 2
    total = 1 << log2_leaves
    cfg = cpa.get_default_poseidon_parameters(2, False)
    leaves = field.random_fr(total, seed=0xA5A50003).reshape(total, 1, 4)
    mg = cpa.MultiGpu(list(range(g)))

    def timed(config, lp, tp, lv):
        mg.build_sharded(config, lp, tp, lv[: 1 << 12], want_nodes=False)  # handles, tables, scratch, RCCL warm-up
        mg.build_sharded(config, lp, tp, lv, want_nodes=False)
        best = None
        for _ in range(3):
            m0 = time.perf_counter()
            _, _, mroot = mg.build_sharded(config, lp, tp, lv, want_nodes=False)
            sec = time.perf_counter() - m0
            if best is None or sec < best[0]:
                best = (sec, mg.last_phases(), mroot)
        return best
    secs, phases, mroot = timed(cpa.PoseidonFieldConfig, cfg, cfg, leaves)
    res = {"entry_point": "akp_merkle_build_sharded_poseidon", "devices": g, "seconds": secs, "phases_ms": phases,
           "includes": "copy-in of the leaves from pageable memory over PCIe (one host thread per device)",
           "collective": "ncclAllGather of %d sub-roots" % g, "root_limb0": int(np.asarray(mroot).reshape(-1)[0])}
    if bh_log2_per_gpu:
        from crypto_primitives_amd import params as cparams
        from crypto_primitives_amd.crh import bowe_hopwood
        B = bowe_hopwood.Parameters(cparams.bowe_hopwood_generators(0xA5A50005, 63, 9))
        nb = g << bh_log2_per_gpu
        per = 1 << bh_log2_per_gpu
        lv = np.concatenate([np.random.default_rng(0xA5A50005 + r).integers(0, 256, size=(per, 32), dtype=np.uint8) for r in range(g)])  # rank r's shard of the torchrun leg
        bsecs, bphases, broot = timed(cpa.BoweHopwoodByteConfig, B, B, lv)
        res["bowe_hopwood"] = {"entry_point": "akp_merkle_build_sharded_te", "leaves": nb, "leaves_per_device": per, "seconds": bsecs, "leaves_per_s": nb / bsecs,
                               "phases_ms": bphases, "root_limb0": int(np.asarray(broot).reshape(-1)[0])}
    mg.close()
    print(json.dumps(res))
    return 0


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
    ap.add_argument("--sustain-seconds", type=float, default=3.0, help="length of each sustained loop (0 disables)")
    ap.add_argument("--sustain-log2-big", type=int, default=24, help="second sustained size (0 disables)")
    ap.add_argument("--settle-launches", type=int, default=120, help="untimed launches before the W warm-up steps (clock ramp)")
    ap.add_argument("--no-host-path", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target duration of the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--one-process-leg", type=int, default=0, help=argparse.SUPPRESS)  # internal: child process of the Merkle leg
    ap.add_argument("--one-process-bh", type=int, default=0, help=argparse.SUPPRESS)   # internal: its Bowe-Hopwood share per device
    args = ap.parse_args()
    if args.one_process_leg:
        return one_process_leg(args.one_process_leg, args.one_process_bh)

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

    ctx = cpa.default_context(local_rank)
    cfg = cpa.get_default_poseidon_parameters(2, False)
    ph = cfg.handle(ctx)
    n = 1 << args.log2_states
    t = cfg.t
    stream = torch.cuda.current_stream(dev).cuda_stream
    ora_threads = max(1, min(32, (os.cpu_count() or 1)))

    def barrier():
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

    # synthetic states (seed per BASELINE.md config 2), resident in HBM before the timed region
    host_states = field.random_fr(n * t, seed=0xA5A50002 + rank).reshape(n, t, 4)
    d_states = torch.from_numpy(host_states.view(np.int64)).to(dev)

    def step():
        check(lib.akp_poseidon_permute_batch_dev(ph.h, d_states.data_ptr(), n, stream))

    ora = None
    if rank == 0:
        from oracle import cref
        ora = cref.Poseidon(cfg.full_rounds, cfg.partial_rounds, cfg.alpha, cfg.rate, cfg.capacity, cfg.ark, cfg.mds)

    # ================= side legs first: from an idle device the first launches run on ramping clocks =================
    # ---- BASELINE config 3: sharded MerkleTree::new, Poseidon, strong scaling over the same total leaf count ---------
    merkle = None
    if args.merkle_log2:
        total = 1 << args.merkle_log2
        per = total // world
        leaves = field.random_fr(per, seed=0xA5A50003 + rank).reshape(per, 1, 4)
        d_leaves = torch.from_numpy(leaves.view(np.int64)).to(dev)
        backend = GpuPoseidonBackend(cfg, cfg, leaf_len=1, device=dev)
        build_sharded(backend, d_leaves, total, dist)  # untimed full-size warm-up build (allocations, RCCL, clocks)
        barrier()
        m0 = time.perf_counter()
        res = build_sharded(backend, d_leaves, total, dist)
        barrier()
        msec = max_over_ranks(time.perf_counter() - m0)
        merkle = {"config": "BASELINE configs[2]: MerkleTree::new, Poseidon leaf + two-to-one, 1-Fr leaves", "leaves": total, "seconds": msec,
                  "leaves_per_s": total / msec, "scaling": "strong", "permutations": 2 * total - 1,
                  "root_limb0": int(np.asarray(res["root"]).reshape(-1)[0]), "algorithmic_GBps": 160.0 * total / msec / 1e9,
                  "hbm_frac": 160.0 * total / msec / 1e9 / HBM_PEAK_GBS}
        if rank == 0:
            # sampled parity on rank 0's sub-tree: leaf digests, and inner nodes recomputed by the oracle from their children
            ln = res["leaf_nodes"].cpu().numpy().view(np.uint64).reshape(per, 4)
            nl = res["non_leaf_nodes"].cpu().numpy().view(np.uint64).reshape(per - 1, 4)
            si = np.unique(np.linspace(0, per - 1, 257).astype(np.int64))
            ok = np.array_equal(ln[si], ora.crh_batch(np.ascontiguousarray(leaves[si]), 1, threads=ora_threads))
            ni = np.unique(np.concatenate([np.arange(0, min(64, per - 1)), np.linspace(0, per - 2, 257).astype(np.int64)]))

            def child(ix):  # heap children: inner nodes below per - 1, then the leaf digests
                return np.where((ix < per - 1)[:, None], nl[np.clip(ix, 0, per - 2)], ln[np.clip(ix - (per - 1), 0, per - 1)])
            ok = ok and np.array_equal(nl[ni], ora.two_to_one_batch(np.ascontiguousarray(child(2 * ni + 1)), np.ascontiguousarray(child(2 * ni + 2)),
                                                                    threads=ora_threads))
            merkle["sampled_parity_bit_exact"] = bool(ok)
            if not ok:
                raise SystemExit("Merkle leg: sampled nodes differ from the oracle")
        # the same tree through the C ABI's single-process multi-device entry point (what a Rust host calls): all visible GPUs
        # (a power of two), leaves in host memory, RCCL all-gather of the sub-roots inside libakp.so.  PCIe-inclusive.  Only when
        # this is the one process of the run; any failure is reported, not fatal (n_dev > 1 cannot be tested on a one-GPU box).
        if world == 1 and not shared_gpu and os.environ.get("AKP_BENCH_NO_MULTI") != "1":
            # in a child process with a time limit: a first-ever n_dev > 1 RCCL bring-up must not be able to take the headline
            # measurement down with it (a crash or a hang there is reported here, nothing else)
            cmd = [sys.executable, os.path.abspath(__file__), "--one-process-leg", str(args.merkle_log2), "--one-process-bh", str(args.bh_merkle_log2)]
            try:
                cp = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
                line = [l for l in cp.stdout.splitlines() if l.startswith("{")]
                if cp.returncode == 0 and line:
                    leg = json.loads(line[-1])
                    leg["root_matches"] = leg.pop("root_limb0", None) == merkle["root_limb0"]
                    merkle["one_process_c_abi"] = leg
                else:
                    merkle["one_process_c_abi"] = {"error": "exit %d: %s" % (cp.returncode, (cp.stderr or cp.stdout)[-300:])}
            except subprocess.TimeoutExpired:  # pragma: no cover
                merkle["one_process_c_abi"] = {"error": "no result within 300 s (child process stopped)"}
            except Exception as exc:  # pragma: no cover
                merkle["one_process_c_abi"] = {"error": repr(exc)[:300]}
        del d_leaves, res, backend

    # ---- BASELINE config 4: Pedersen 4x256 over Jubjub, 2^k x 128 B per GPU ------------------------------------------
    pedersen = None
    if args.pedersen_log2:
        from crypto_primitives_amd import params as cparams
        from crypto_primitives_amd.crh import pedersen as cped
        npd = 1 << args.pedersen_log2
        gens = cparams.pedersen_generators(0xA5A50004, 4, 256)
        PP = cped.Parameters(gens)
        hP = PP.handle(ctx)
        msgs = np.random.default_rng(0xA5A50004 + rank).integers(0, 256, size=(npd, 128), dtype=np.uint8)
        d_msgs = torch.from_numpy(msgs).to(dev)
        d_out = torch.empty((npd, 8), dtype=torch.int64, device=dev)

        def ped_step():
            check(lib.akp_te_crh_batch_dev(hP.h, d_msgs.data_ptr(), npd, 128, d_out.data_ptr(), stream))
        for _ in range(3):
            ped_step()
        barrier()
        reps = 10
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        p0 = time.perf_counter()
        for a, b in evs:
            a.record()
            ped_step()
            b.record()
        barrier()
        psec = max_over_ranks(time.perf_counter() - p0)
        kms = sorted(a.elapsed_time(b) for a, b in evs)
        kavg = sum(kms) / len(kms) / 1e3
        pinfo = hP.info(128)
        psteps = pinfo["steps"]
        pedersen = {"config": "BASELINE configs[3]: pedersen::CRH, Jubjub, window 4x256, 128-byte messages", "messages_per_gpu": npd,
                    "hashes_per_s": npd * world * reps / psec, "ms_per_batch": psec / reps * 1e3,
                    "roofline": {"bound": "hbm", "kernels": "te_accumulate_kernel<2> + te_finalize_kernel<0>", "algorithmic_bytes_per_hash": 192,
                                 "kernel_avg_ms": kavg * 1e3, "achieved": 192.0 * npd / kavg / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": 192.0 * npd / kavg / 1e9 / HBM_PEAK_GBS,
                                 "traffic": te_counters("pedersen_128B", npd)["traffic"],
                                 "traffic_over_algorithmic": te_counters("pedersen_128B", npd)["traffic"] / (192.0 * npd),
                                 "traffic_static_from": PMC_TE["source"] + " (FETCH_SIZE x 2 + WRITE_SIZE of te_accumulate_kernel<2> + te_finalize_kernel<0>; "
                                                        "NOT measured in this run)",
                                 "table_bytes_gathered_per_hash": psteps * int(lib.akp_te_entry_bytes()),
                                 "gather_over_algorithmic": psteps * int(lib.akp_te_entry_bytes()) / 192.0,
                                 "table": pinfo,
                                 "valu": {"table_steps_per_hash": psteps, "field_products_per_step": 7,
                                          "valu_instructions_per_hash": te_counters("pedersen_128B", 1)["valu_instr"],
                                          "v_mad_per_s": (psteps * 7 + 6) * MADS_PER_PRODUCT * npd / kavg,
                                          "frac_of_mad_issue_peak": (psteps * 7 + 6) * MADS_PER_PRODUCT * npd / kavg / (VALU_PEAK_WAVE_INSTR * 64),
                                          "v_mad_note": "7 products per table step + ~6 per hash in the shared-inversion pass, 153 multiply-adds each",
                                          "note": "VALU-issue bound like the permutation: one mixed addition of 7 products per table step "
                                                  "(signed-subset table); one 128-byte line per entry, gathered through L2 / Infinity Cache / HBM "
                                                  "(counters and entry-layout A/B: profiles/r03_s4; gather share: profiles/r03_s4/te_gather_probe_line128.txt, profiles/r03_s10)"}}}
        if args.sustain_seconds > 0:  # clock / power under the gather-heavy kernel (the permutation's figures are in `sustained`)
            count = int(min(2000, max(8, 0.5 * args.sustain_seconds / max(kavg, 1e-4))))
            torch.cuda.synchronize(dev)
            s0 = time.perf_counter()
            for _ in range(count):
                ped_step()
            time.sleep(min(0.25, 0.25 * count * kavg))  # sample while the queue is still draining
            cmid, wmid = gpu_clock_mhz(local_rank), gpu_power_w()
            torch.cuda.synchronize(dev)
            ssec = time.perf_counter() - s0
            pedersen["sustained"] = {"launches": count, "seconds": ssec, "hashes_per_s": npd * count / ssec, "sclk_mhz_during": cmid, "power_w_during": wmid}
        if rank == 0:
            from oracle import cref
            cur = cref.CurveParams(4, 256, gens)
            si = np.unique(np.concatenate([np.arange(64), np.linspace(0, npd - 1, 193).astype(np.int64)]))
            got = d_out.cpu().numpy().view(np.uint64).reshape(npd, 2, 4)[si]
            ok = bool(np.array_equal(got, cur.pedersen_crh_batch(np.ascontiguousarray(msgs[si]), len(si), 128, threads=ora_threads)))
            pedersen["sampled_parity_bit_exact"] = ok
            pedersen["parity_samples"] = int(len(si))
            if not ok:
                raise SystemExit("Pedersen leg: sampled digests differ from the oracle")
        del d_msgs, d_out

    # ---- BASELINE config 5: Bowe-Hopwood 63x9 tree, 2^k x 32 B leaves PER GPU (weak: 2^26 on 8 GPUs) ------------------
    bh_merkle = None
    if args.bh_merkle_log2:
        from crypto_primitives_amd import params as cparams
        from crypto_primitives_amd.crh import bowe_hopwood
        per = 1 << args.bh_merkle_log2
        total = per * world
        gens = cparams.bowe_hopwood_generators(0xA5A50005, 63, 9)
        B = bowe_hopwood.Parameters(gens)
        leaves = np.random.default_rng(0xA5A50005 + rank).integers(0, 256, size=(per, 32), dtype=np.uint8)
        d_leaves = torch.from_numpy(leaves).to(dev)
        tb = GpuTeBackend(B, B, device=dev)
        build_sharded(tb, d_leaves, total, dist)  # untimed full-size warm-up (tables, scratch, RCCL)
        barrier()
        reps = 3
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        m0 = time.perf_counter()
        for a, b in evs:
            a.record()
            res = build_sharded(tb, d_leaves, total, dist)
            b.record()
        barrier()
        bsec = max_over_ranks(time.perf_counter() - m0) / reps
        dev_ms = sum(a.elapsed_time(b) for a, b in evs) / reps
        bh_merkle = {"config": "BASELINE configs[4]: MerkleTree::new, Bowe-Hopwood 63x9 over Jubjub, 32-byte leaves, ByteDigestConverter",
                     "leaves": total, "leaves_per_gpu": per, "seconds": bsec, "leaves_per_s": total / bsec, "scaling": "weak",
                     "roofline": {"bound": "hbm", "kernels": "te_accumulate_kernel<1> + te_finalize_kernel<1> + te_serialize_pairs_kernel per level",
                                  "algorithmic_bytes_per_leaf": 160, "device_ms_per_build": dev_ms, "achieved": 160.0 * per / (dev_ms / 1e3) / 1e9,
                                  "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": 160.0 * per / (dev_ms / 1e3) / 1e9 / HBM_PEAK_GBS,
                                  "table": B.handle(ctx).info(32),
                                  "traffic": te_counters("bh_32B", per)["traffic"] + te_counters("bh_70B", per - 1, (B.handle(ctx).info(64)["steps"] + 1) / 39.0)["traffic"],
                                  "traffic_static_from": PMC_TE["source"] + " (FETCH_SIZE x 2 + WRITE_SIZE of te_accumulate_kernel<1> + te_finalize_kernel<1> at 2^20 x 32 B "
                                                         "for the leaf level and 2^20 x 70 B scaled to the inner nodes' table steps; levels of <= 2^14 nodes run the split kernel; "
                                                         "NOT measured in this run)",
                                  "valu": {"table_steps_per_leaf_hash": B.handle(ctx).info(32)["steps"],
                                           "table_steps_per_inner_node": B.handle(ctx).info(64)["steps"] + 1,
                                           "inner_node_note": "64 bytes of digests in a 70-byte buffer: the table steps of the 64 data bytes + one constant "
                                                              "entry for the zero-padded tail (a zero chunk adds +g); %d steps if the padding is walked" % B.handle(ctx).info(70)["steps"],
                                           "field_products_per_step": 7}}}
        rfb = bh_merkle["roofline"]
        rfb["traffic_over_algorithmic"] = rfb["traffic"] / (160.0 * per)
        bh_mads = (per * (rfb["valu"]["table_steps_per_leaf_hash"] * 7 + 6) + (per - 1) * (rfb["valu"]["table_steps_per_inner_node"] * 7 + 6)) * MADS_PER_PRODUCT
        rfb["valu"]["v_mad_per_s"] = bh_mads / (dev_ms / 1e3)
        rfb["valu"]["frac_of_mad_issue_peak"] = bh_mads / (dev_ms / 1e3) / (VALU_PEAK_WAVE_INSTR * 64)
        if rank == 0:
            from oracle import cref
            cur = cref.CurveParams(63, 9, gens)
            ln = res["leaf_nodes"].cpu().numpy().view(np.uint64).reshape(per, 4)
            nl = res["non_leaf_nodes"].cpu().numpy().view(np.uint64).reshape(per - 1, 4)
            si = np.unique(np.linspace(0, per - 1, 129).astype(np.int64))
            ok = np.array_equal(ln[si], cur.bh_crh_batch(np.ascontiguousarray(leaves[si]), len(si), 32, threads=ora_threads))
            # inner nodes from their children: buffer = LE(left) || LE(right) zero-padded to (63 * 9) / 8 = 70 bytes
            ni = np.unique(np.concatenate([np.arange(0, min(32, per - 1)), np.linspace(0, per - 2, 97).astype(np.int64)]))

            def child(ix):
                return np.where((ix < per - 1)[:, None], nl[np.clip(ix, 0, per - 2)], ln[np.clip(ix - (per - 1), 0, per - 1)])
            buf = np.zeros((len(ni), 70), np.uint8)
            buf[:, :32] = cref.from_mont(np.ascontiguousarray(child(2 * ni + 1))).view(np.uint8).reshape(len(ni), 32)
            buf[:, 32:64] = cref.from_mont(np.ascontiguousarray(child(2 * ni + 2))).view(np.uint8).reshape(len(ni), 32)
            ok = ok and np.array_equal(nl[ni], cur.bh_crh_batch(buf, len(ni), 70, threads=ora_threads))
            bh_merkle["sampled_parity_bit_exact"] = bool(ok)
            if not ok:
                raise SystemExit("Bowe-Hopwood leg: sampled nodes differ from the oracle")
        del d_leaves, res, tb

    # ---- proofs, verification, updates on a resident 2^20-leaf tree (one process only) ----------
    proofs = None
    if args.proofs_log2 and world == 1 and rank == 0:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_proofs
        proofs = {}
        for name in ("poseidon", "bh"):
            proofs[name] = bench_proofs.run(name, args.proofs_log2, min(args.proofs_m_log2, args.proofs_log2), local_rank)
            if not proofs[name]["all_parity_bit_exact"]:
                raise SystemExit("proofs leg (%s): a parity / control check failed: %s" % (name, json.dumps(proofs[name])))
        proofs["sponge"] = bench_proofs.run_sponge(args.proofs_log2, local_rank)  # 8(f) rank 4: device-resident duplex sponges
        if not proofs["sponge"]["sampled_parity_bit_exact"]:
            raise SystemExit("sponge leg: sampled squeezes differ from the oracle")
        proofs["profiles"] = "profiles/r03_s*/proofs_* (rocprofv3 --kernel-trace --stats of `python tools/bench_proofs.py`)"

    # ================= the headline: W warm-up + K timed steps of the 2^20-state permutation ==========================
    parity = {"probe_kernel": lib.akp_poseidon_kernel_for(ph.h, n, 0).decode(), "timed_buffer_states_checked": 0, "bit_exact": None}
    step()  # one pass outside W: its output is checked against the oracle on a strided sample of the TIMED buffer
    torch.cuda.synchronize(dev)
    if rank == 0:
        si = np.unique(np.concatenate([np.arange(128), np.linspace(0, n - 1, 385).astype(np.int64), np.arange(n - 128, n)]))
        got = d_states[torch.from_numpy(si).to(dev)].cpu().numpy().view(np.uint64).reshape(len(si), t, 4)
        exp = ora.permute_batch(np.ascontiguousarray(host_states[si]), threads=ora_threads).reshape(len(si), t, 4)
        parity["timed_buffer_states_checked"] = int(len(si))
        parity["bit_exact"] = bool(np.array_equal(got, exp))
        if not parity["bit_exact"]:
            raise SystemExit("parity probe FAILED: the timed kernel's output differs from the oracle")
    # the oracle check above left the GPU idle for ~0.1 s and the clocks drop within milliseconds: settle them with untimed
    # launches (like the side legs, outside W and K) so that the W + K steps measure the steady state, not the ramp
    # (everything the timed region needs is prepared BEFORE the settle launches: reading the clock can cost a `rocm-smi`
    # subprocess -- hundreds of ms of idle device right before t0 would hand the first timed steps a ramping clock)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    for _ in range(8):
        step()
    clk0 = gpu_clock_mhz(local_rank)  # sampled while launches are in flight
    for _ in range(args.settle_launches):
        step()
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()
        step()
        b.record()
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    clk1 = gpu_clock_mhz(local_rank)
    kern_ms = [a.elapsed_time(b) for a, b in ev]
    kern_avg_s = (sum(kern_ms) / len(kern_ms)) / 1e3

    # ---- sustained: the same launch looped for seconds (rank 0's device; other ranks idle at the barrier) ------------
    sustained = None
    if args.sustain_seconds > 0 and rank == 0:
        sustained = {}
        for lg in sorted({args.log2_states, args.sustain_log2_big} - {0}):
            ns = 1 << lg
            if ns == n:
                buf = d_states
            else:
                try:
                    buf = torch.from_numpy(field.random_fr(min(ns, 1 << 20) * t, seed=0xA5A50012).reshape(-1, t, 4).view(np.int64)).to(dev).repeat(max(1, ns >> 20), 1, 1)
                except Exception:
                    continue
            per_launch = max(kern_avg_s * ns / n, 1e-4)
            count = int(min(4000, max(8, args.sustain_seconds / per_launch)))
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(count)]
            c0 = gpu_clock_mhz(local_rank)
            torch.cuda.synchronize(dev)
            s0 = time.perf_counter()
            for a, b in evs:
                a.record()
                check(lib.akp_poseidon_permute_batch_dev(ph.h, buf.data_ptr(), ns, stream))
                b.record()
            cmid, wmid = gpu_clock_mhz(local_rank), gpu_power_w()
            torch.cuda.synchronize(dev)
            secs = time.perf_counter() - s0
            ms = sorted(a.elapsed_time(b) for a, b in evs)
            sustained["2^%d" % lg] = {"launches": count, "seconds": secs, "permutations_per_s": ns * count / secs,
                                      "launch_ms_min": ms[0], "launch_ms_median": ms[len(ms) // 2], "launch_ms_max": ms[-1],
                                      "sclk_mhz_before": c0, "sclk_mhz_during": cmid, "sclk_mhz_after": gpu_clock_mhz(local_rank), "power_w_during": wmid}
            if buf is not d_states:
                del buf

    # ---- host-pointer entry points (what a Rust host calls): PCIe-inclusive -----------------------------------------
    host_path = None
    if not args.no_host_path and rank == 0:
        import ctypes as C
        host_path = {"states": n, "pcie_bound_note": "96 B in + 96 B out per permutation; PCIe Gen5 x16 is 63 GB/s per direction (spec)"}
        work = host_states.copy()
        for label in ("pageable", "pinned"):
            if label == "pinned":
                pp = C.c_void_p()
                check(lib.akp_host_alloc(work.nbytes, C.byref(pp)))
                arr = np.ctypeslib.as_array((C.c_uint64 * (work.size)).from_address(pp.value))
                arr[:] = host_states.reshape(-1)
                ptr = pp
            else:
                ptr = work.ctypes.data
            check(lib.akp_poseidon_permute_batch(ph.h, ptr, n))  # warm-up (scratch, streams)
            reps = 5
            h0 = time.perf_counter()
            for _ in range(reps):
                check(lib.akp_poseidon_permute_batch(ph.h, ptr, n))
            hs = (time.perf_counter() - h0) / reps
            host_path[label] = {"permutations_per_s": n / hs, "ms_per_batch": hs * 1e3, "GBps_each_direction": 96.0 * n / hs / 1e9,
                                "mode": "zero copy: the kernel addresses the pinned buffer over PCIe" if label == "pinned"
                                        else "chunked copy-in / kernel / copy-out on three streams (runtime-staged copies)"}
            if label == "pinned":
                check(lib.akp_host_free(pp))
        if args.pedersen_log2:  # config 4 through the host-pointer entry point: 128 B in, 64 B out per hash
            from crypto_primitives_amd import params as cparams2
            from crypto_primitives_amd.crh import pedersen as cped2
            nph = 1 << args.pedersen_log2
            hPh = cped2.Parameters(cparams2.pedersen_generators(0xA5A50004, 4, 256)).handle(ctx)
            hm = np.random.default_rng(0xA5A50014).integers(0, 256, size=(nph, 128), dtype=np.uint8)
            ho = np.empty((nph, 8), dtype=np.uint64)
            check(lib.akp_te_crh_batch(hPh.h, hm.ctypes.data, nph, 128, ho.ctypes.data))
            reps = 3
            h0 = time.perf_counter()
            for _ in range(reps):
                check(lib.akp_te_crh_batch(hPh.h, hm.ctypes.data, nph, 128, ho.ctypes.data))
            hs = (time.perf_counter() - h0) / reps
            host_path["pedersen_pageable"] = {"hashes_per_s": nph / hs, "ms_per_batch": hs * 1e3, "GBps_in": 128.0 * nph / hs / 1e9, "GBps_out": 64.0 * nph / hs / 1e9,
                                              "mode": "double-buffered chunks of 2^17 messages: copy-in / kernels / copy-out on three streams"}
            # the same with pinned buffers (akp_host_alloc): asynchronous DMA in, digests written straight into host memory by the finalize pass
            pm, po = C.c_void_p(), C.c_void_p()
            check(lib.akp_host_alloc(hm.nbytes, C.byref(pm)))
            check(lib.akp_host_alloc(ho.nbytes, C.byref(po)))
            np.ctypeslib.as_array((C.c_uint8 * hm.size).from_address(pm.value))[:] = hm.reshape(-1)
            check(lib.akp_te_crh_batch(hPh.h, pm, nph, 128, po))
            pinned_out = np.ctypeslib.as_array((C.c_uint64 * ho.size).from_address(po.value)).reshape(ho.shape)
            same = bool(np.array_equal(pinned_out, ho))
            h0 = time.perf_counter()
            for _ in range(reps):
                check(lib.akp_te_crh_batch(hPh.h, pm, nph, 128, po))
            hs2 = (time.perf_counter() - h0) / reps
            host_path["pedersen_pinned"] = {"hashes_per_s": nph / hs2, "ms_per_batch": hs2 * 1e3, "GBps_in": 128.0 * nph / hs2 / 1e9, "GBps_out": 64.0 * nph / hs2 / 1e9,
                                            "digests_equal_the_pageable_call": same,
                                            "mode": "pinned buffers: DMA copy-in of 2^17-message chunks under the kernels, zero-copy output (the finalize pass stores into host memory)"}
            check(lib.akp_host_free(pm))
            check(lib.akp_host_free(po))
            if not same:
                raise SystemExit("host path: the pinned Pedersen call differs from the pageable one")
        if args.merkle_log2:
            ntree = 1 << min(args.merkle_log2, 22)
            lv = field.random_fr(ntree, seed=0xA5A50013).reshape(ntree, 1, 4)
            root = np.empty(4, np.uint64)
            check(lib.akp_merkle_build_poseidon(ph.h, ph.h, lv.ctypes.data, ntree, 1, None, None, root.ctypes.data))
            h0 = time.perf_counter()
            check(lib.akp_merkle_build_poseidon(ph.h, ph.h, lv.ctypes.data, ntree, 1, None, None, root.ctypes.data))
            host_path["merkle_root_only"] = {"leaves": ntree, "seconds": time.perf_counter() - h0}

    if rank != 0:
        if dist:
            dist.barrier()
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
        "dtype": "i32 limbs (255-bit Montgomery integers, radix 2^29 x 9, 64-bit v_mad_i64_i32 accumulation)",
        "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: batched Poseidon permutation, BLS12-381 Fr, t=3 rate=2 alpha=17 RF=8 RP=31 "
                               "(default Grain-LFSR parameters), 2^%d states per GPU, in place in HBM" % args.log2_states,
                   "states_per_gpu": n, "parallelism": "shard%d (no data-path collective)" % world},
        "settle_launches_before_warmup": args.settle_launches,
        "launch": {"ranks": world, "backend": None if not dist else ("gloo (shared-GPU test hook)" if shared_gpu else "nccl (RCCL)"),
                   "rank_devices": rank_devices},
        "parity_probe_bit_exact": parity["bit_exact"],
        "parity": parity,
        "roofline": {"bound": "hbm", "kernel": parity["probe_kernel"], "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": PMC_TRAFFIC_BYTES_PER_PERM * n,
                     "traffic_static_from": PMC_TRAFFIC_SOURCE + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH x2 gfx950 "
                                            "correction; NOT measured in this run)",
                     "peak_measured_copy": hbm_copy_gbs, "frac_of_measured_copy": achieved / hbm_copy_gbs if hbm_copy_gbs else None,
                     "kernel_avg_ms": kern_avg_s * 1e3, "kernel_min_ms": min(kern_ms), "kernel_max_ms": max(kern_ms),
                     "algorithmic_bytes_per_launch": ALGO_BYTES_PER_PERM * n,
                     "sclk_mhz_before": clk0, "sclk_mhz_after": clk1,
                     "valu": {"note": "the path is integer-ALU bound (~%d reference-shaped Montgomery products per 192 B); "
                                      "fraction of the measured v_mad issue peak spent on multiplies" % MODMUL_PER_PERM_REF,
                              "ref_modmul_per_s": MODMUL_PER_PERM_REF * n / kern_avg_s,
                              "v_mad_per_s": MADS_PER_PERM * n / kern_avg_s,
                              "v_mad_peak_per_s": VALU_PEAK_WAVE_INSTR * 64,
                              "frac_of_mad_issue_peak": MADS_PER_PERM * n / kern_avg_s / (VALU_PEAK_WAVE_INSTR * 64),
                              "valu_instructions_per_permutation": 1224736768 // 16384, "valu_busy_percent": 96.1,
                              "valu_counters_static_from": "profiles/r03_s11/pmc_counters_poseidon.txt (SQ_INSTS_VALU / 16384 waves, VALUBusy; NOT measured in this run)"}},
    }
    for key, leg in (("sustained", sustained), ("merkle", merkle), ("pedersen", pedersen), ("bh_merkle", bh_merkle), ("proofs", proofs), ("host_path", host_path)):
        if leg:
            out[key] = leg
    if not args.no_cpu_baseline and world == 1:
        from oracle import cref
        hw = cref.hardware_threads()
        info = cpu_info()
        # what this process can really use: affinity mask, cgroup CPU quota and hardware threads, whichever is smallest
        bounds = {"hardware threads": hw, "affinity mask": info["affinity_cpus"], "cgroup quota": info["cgroup_cpu_quota"]}
        basis, eff_cores = min(((k, v) for k, v in bounds.items() if v), key=lambda kv: kv[1])
        cands = sorted({hw, max(1, hw // 2), max(1, hw // 4), max(1, hw // 8), min(hw, 16), min(hw, 8), max(1, int(round(eff_cores)))}, reverse=True)

        def cpu_leg(run, total, seconds, cal, quant=lambda k: k):
            """`run(k, threads)` processes the first k items; returns (rate at the best thread count, threads, 1-thread rate, items
            timed).  Whole passes over min(total, rate * seconds) items until `seconds` have been spent."""
            c0 = time.perf_counter()
            run(quant(max(2, cal // 8)), 1)
            rate1 = quant(max(2, cal // 8)) / (time.perf_counter() - c0)
            best = (rate1, 1)
            for cand in cands:
                c0 = time.perf_counter()
                run(cal, cand)
                r = cal / (time.perf_counter() - c0)
                if r > best[0]:
                    best = (r, cand)
            rate, threads = best
            sample = quant(int(min(total, max(cal, rate * seconds))))
            passes = 0
            c0 = time.perf_counter()
            while True:
                run(sample, threads)
                passes += 1
                cpu_s = time.perf_counter() - c0
                if sample < total or cpu_s >= seconds:
                    break
            return sample * passes / cpu_s, threads, rate1, sample * passes
        rate_n, threads, rate1, sample = cpu_leg(lambda k, th: ora.permute_batch(host_states[:k], threads=th), n, args.cpu_seconds, 8192)
        out["cpu_baseline"] = {"value": rate_n, "unit": "permutations/s", "cores": eff_cores, "cores_basis": basis, "kind": "port",
                               "threads_used": threads, "rate_1_thread": rate1, "effective_cores": rate_n / rate1,
                               "hardware_threads": hw, **info,
                               "sample": "%d permutations over the same 2^%d states, reference-shaped C restatement (oracle/c/akp_oracle.c: "
                                         "dense MDS, square-and-multiply, one permutation per call as the reference), %d pthreads (best of a "
                                         "thread-count sweep); `cores` = min(affinity, cgroup quota, hardware threads) = what this container "
                                         "may use; `effective_cores` = that rate / the 1-thread rate: what it really delivered" % (sample, args.log2_states, threads)}
        out["gpu_over_cpu"] = value / rate_n
        curve_seconds = max(2.0, args.cpu_seconds / 3.0)
        if pedersen:  # BASELINE configs[3] on the CPU: bit-by-bit conditional additions as crh/pedersen/mod.rs:112-124
            from crypto_primitives_amd import params as cparams3
            cur = cref.CurveParams(4, 256, cparams3.pedersen_generators(0xA5A50004, 4, 256))
            cm = np.random.default_rng(0xA5A50004).integers(0, 256, size=(1 << 16, 128), dtype=np.uint8)
            r_n, th, r1, smp = cpu_leg(lambda k, t_: cur.pedersen_crh_batch(cm[:k], k, 128, threads=t_), len(cm), curve_seconds, 2048)
            out["cpu_baseline"]["pedersen"] = {"value": r_n, "unit": "hashes/s", "threads_used": th, "rate_1_thread": r1, "effective_cores": r_n / r1,
                                               "sample": "%d Pedersen 4x256 hashes of 128-byte messages (orc_pedersen_crh_batch)" % smp,
                                               "gpu_over_cpu": pedersen["hashes_per_s"] / r_n}
        if bh_merkle:  # BASELINE configs[4] on the CPU: the whole tree (leaf hashes + inner levels, barrier per level), 2^k leaves
            from crypto_primitives_amd import params as cparams4
            curb = cref.CurveParams(63, 9, cparams4.bowe_hopwood_generators(0xA5A50005, 63, 9))
            cl = np.random.default_rng(0xA5A50005).integers(0, 256, size=(1 << 16, 32), dtype=np.uint8)

            def pow2(k):
                return 1 << max(1, int(k).bit_length() - 1)  # the largest power of two <= k (a tree needs one)
            r_n, th, r1, smp = cpu_leg(lambda k, t_: curb.merkle_build(1, curb, cl[:k], k, 32, threads=t_), len(cl), curve_seconds, 2048, pow2)
            out["cpu_baseline"]["bh_merkle"] = {"value": r_n, "unit": "leaves/s", "threads_used": th, "rate_1_thread": r1, "effective_cores": r_n / r1,
                                                "sample": "Bowe-Hopwood 63x9 trees over %d leaves of 32 bytes in total (orc_curve_merkle_build, power-of-two trees)" % smp,
                                                "gpu_over_cpu": bh_merkle["leaves_per_s"] / r_n}
    print(json.dumps(out))
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
