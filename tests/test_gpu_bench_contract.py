"""bench.py contract on a GPU box: one JSON line with the required keys -- single process, under torch.distributed.run
(world size 1 exercises the nccl init / barrier / all-gather code path), with several ranks sharing the one GPU, and
self-launched from a plain `python bench.py --gpus 2`."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline", "parity", "curve_parity", "legs", "full"}
ROOFLINE_KEYS = {"bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "kernel_avg_ms", "algorithmic_bytes_per_launch",
                 "effective_sclk_mhz", "frac_of_mad_issue_peak"}
LINE_LIMIT = 6000  # the driver parses the LAST stdout line out of a bounded tail: round 4's 22 KB line was lost
SMALL = ["--log2-states", "16", "--merkle-log2", "12", "--pedersen-log2", "10", "--bh-merkle-log2", "9", "--sustain-seconds", "0.2",
         "--sustain-log2-big", "0", "--proofs-log2", "12", "--proofs-m-log2", "8", "--sweep-max-log2", "20", "--ragged-log2", "13"]


def _check(out, full_path):
    """the printed line: last line of stdout, short, contract keys + roofline + cpu_baseline (null when not measured); returns the
    FULL record (bench_full.json, which the line names) with the line under "_line" """
    lines = out.strip().splitlines()
    line = lines[-1]
    assert line.startswith("{") and len(line) < LINE_LIMIT, (len(line), line[:200])
    assert len([l for l in lines if l.startswith("{")]) == 1  # rank 0 only, one JSON line
    ln = json.loads(line)
    assert REQUIRED <= set(ln), REQUIRED - set(ln)
    assert ln["parity"]["bit_exact"] is True and ln["value"] > 1e6
    assert ln["parity"]["states_checked"] >= 256 and ln["roofline"]["traffic_measured_in_this_run"] is False  # the probe reads the buffer the timed kernel wrote
    r = ln["roofline"]
    assert ROOFLINE_KEYS <= set(r), ROOFLINE_KEYS - set(r)
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-6
    assert r["kernel"] == ln["parity"]["kernel"] and r["traffic_source"].startswith("profiles/")
    assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["kernel_avg_ms"] * 1e-3) / 1e9) < 1e-3 * r["achieved"]
    assert abs(ln["ms_per_step"] - ln["config"]["states_per_gpu"] * ln["n_gpus"] / ln["value"] * 1e3) < 1e-6 * ln["ms_per_step"]
    assert all(not isinstance(v, (dict, list)) for v in ln["legs"].values())  # one scalar per leg
    assert ln["full"] == os.path.basename(full_path)
    d = json.load(open(full_path))
    assert d["value"] == ln["value"] and d["roofline"]["kernel"] == r["kernel"]
    d["_line"] = ln
    return d


def _run(cmd, tmp_path, timeout=900, env=None):
    full = str(tmp_path / "bench_full.json")
    env = dict(os.environ if env is None else env, AKP_BENCH_FULL=full)
    return subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=timeout, env=env), full


def test_bench_single_process(tmp_path):
    p, full = _run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--cpu-seconds", "0.5"] + SMALL, tmp_path)
    assert p.returncode == 0, p.stderr[-3000:]
    d = _check(p.stdout, full)
    ln = d["_line"]
    lcb = ln["cpu_baseline"]  # in the LINE: what the contract asks for, scalars only
    assert lcb["kind"] == "port" and lcb["cores"] >= 1 and lcb["value"] > 0 and lcb["cpu_model"] and lcb["sample"] and ln["gpu_over_cpu"] > 1
    for k in ("merkle_s", "pedersen_hashes_per_s", "bh_s", "bh_leaves_per_s", "verify_paths_hashes_per_s", "host_pinned_perm_per_s",
              "pedersen_cold_first_call_ms", "bh_cold_first_tree_ms", "predicted_8gpu_merkle_s", "predicted_8gpu_bh_s", "ragged_bh_hashes_per_s",
              "ragged_poseidon_hashes_per_s"):
        assert ln["legs"][k] > 0, k
    assert d["n_gpus"] == 1 and d["roofline"]["kernel"] == "poseidon_permute_t3_kernel<true>"  # 2^16 states: the register kernel
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["rate_1_thread"] > 0 and cb["effective_cores"] > 0.5 and cb["cpu_model"]
    assert d["pedersen"]["sampled_parity_bit_exact"] and d["bh_merkle"]["sampled_parity_bit_exact"] and d["merkle"]["sampled_parity_bit_exact"]
    assert d["bh_merkle"]["leaves"] == 512 and d["pedersen"]["roofline"]["frac"] > 0
    assert d["host_path"]["pinned"]["permutations_per_s"] > 0 and d["sustained"]["2^16"]["launches"] >= 8
    assert cb["cores_basis"] in ("cgroup quota", "affinity mask", "hardware threads") and cb["cores"] <= cb["hardware_threads"] and cb["threads_used"] >= 1
    assert cb["pedersen"]["value"] > 0 and cb["bh_merkle"]["value"] > 0
    for cfg in ("poseidon", "bh"):  # proofs, verification, updates on a resident tree
        pr = d["proofs"][cfg]
        assert pr["all_parity_bit_exact"] and pr["leaves"] == 1 << 12
        for leg in ("generate_proof", "verify_paths", "generate_multi_proof", "verify_multipath"):
            assert pr[leg]["wall_ms"] > 0 and pr[leg]["device_ms"] > 0
        assert pr["create"]["wall_ms"] > 0 and pr["create"]["hashes"] == 2 * pr["leaves"] - 1
        assert set(pr["update_batch"]) == {"2^8", "2^10"}
    assert d["proofs"]["sponge"]["sampled_parity_bit_exact"] and d["proofs"]["sponge"]["permutations_per_sponge"] == 4
    for leg in ("pedersen", "bh_merkle"):  # the curve-hash roofline blocks carry counter traffic and the v_mad view
        rf = d[leg]["roofline"]
        assert rf["traffic"] > 0 and rf["traffic_over_algorithmic"] > 1 and "NOT measured in this run" in rf["traffic_static_from"]
        assert 0 < rf["valu"]["frac_of_mad_issue_peak"] < 1
    leg = d["merkle"]["one_process_c_abi"]  # the C ABI's multi-device entry points, with the phase breakdown
    assert leg["root_matches"] and leg["phases_ms"]["whole_call_ms"] > 0 and leg["bowe_hopwood"]["phases_ms"]["copy_in_and_subtree_ms"] > 0
    rt = leg["resident_tree"]  # the sharded RESIDENT tree (akp_multi_tree_*): built, queried and updated without moving its nodes
    assert rt["root_before_updates_matches_the_sharded_build"] and rt["sampled_proofs_verify"] and rt["proofs_ms"] > 0 and rt["update_ms"] > 0
    # round 4: the line explains itself across boxes
    rf = d["roofline"]
    assert 500 < rf["effective_sclk_mhz"] < 3500 and 7.5 < rf["effective_sclk"]["cycles_per_dependent_mad"] < 20  # 8.25 on an idle SIMD of gfx950
    assert len(rf["effective_sclk"]["before_timed_steps_mhz"]) >= 1 and len(rf["effective_sclk"]["after_timed_steps_mhz"]) >= 1
    assert 0 < rf["valu"]["frac_of_mad_issue_peak"] < 1 and (rf["power_w_under_load"] is None or rf["power_w_under_load"] > 20)
    pt = d["sweep"]["points"]["2^20"]
    assert pt["permutations_per_s"] > 1e6 and pt["tree_leaves_per_s"] > 1e5 and 0 < pt["permute_hbm_frac"] < 1
    ps = d["predicted_scaling"]
    assert 0 < ps["merkle_strong"]["8_gpus"]["efficiency"] <= 1.0 and 0 < ps["bh_merkle_weak"]["8_gpus"]["efficiency"] <= 1.0
    assert d["curve_parity"].startswith(("unpinned (emitter not run)", "pinned by"))
    hp = d["host_path"]
    assert hp["pedersen_pinned"]["digests_equal_the_pageable_call"] and hp["pedersen_pinned"]["ms_min"] <= hp["pedersen_pinned"]["ms_per_batch"] <= hp["pedersen_pinned"]["ms_max"]


def test_bench_under_torchrun_world1(tmp_path):
    p, full = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                    "--master-port", "29531", "bench.py", "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-host-path"]
                   + SMALL, tmp_path)
    assert p.returncode == 0, (p.stdout[-1000:], p.stderr[-3000:])
    d = _check(p.stdout, full)
    assert d["_line"]["cpu_baseline"] is None
    assert d["n_gpus"] == 1 and "merkle" in d and d["bh_merkle"]["leaves"] == 512 and d["launch"]["backend"].startswith("nccl")


@pytest.mark.parametrize("world", [2, 4, 8])
def test_bench_multi_rank_code_path_on_one_gpu(world, tmp_path):
    """The N > 1 branch of bench.py (rank env, barriers, MAX-reduced timing, sharded Merkle legs, rank-0 printing) with
    `world` ranks sharing GPU 0 and gloo carrying the collectives (AKP_BENCH_SHARED_GPU=1, a test hook)."""
    args = ["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-host-path"] + SMALL
    args[args.index("--log2-states") + 1] = "14"
    env = dict(os.environ, AKP_BENCH_SHARED_GPU="1")
    p, full = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
                    "127.0.0.1", "--master-port", str(29540 + world), "bench.py", "--gpus", str(world)] + args, tmp_path, env=env)
    assert p.returncode == 0, (p.stdout[-1000:], p.stderr[-3000:])
    d = _check(p.stdout, full)
    assert d["n_gpus"] == world and d["config"]["states_per_gpu"] == 1 << 14 and "cpu_baseline" not in d
    assert d["_line"]["cpu_baseline"] is None and d["_line"]["legs"]["merkle_s"] > 0 and d["_line"]["legs"]["bh_s"] > 0
    assert d["merkle"]["leaves"] == 1 << 12 and d["bh_merkle"]["leaves"] == world << 9 and d["bh_merkle"]["scaling"] == "weak"
    assert d["launch"]["ranks"] == world and len(d["launch"]["rank_devices"]) == world


def test_bench_on_all_real_devices_nccl(tmp_path):
    """One rank per PHYSICAL GPU over RCCL: world = the largest power of two <= min(device_count, 8).  On a one-GPU box there is
    nothing beyond test_bench_under_torchrun_world1 to run, so it skips; the moment a node has more devices this is a true
    world > 1 run of the sharded legs (leaf-range shards, one all-gather of the sub-roots) with no test hook involved."""
    import torch
    ndev = torch.cuda.device_count()
    world = 1
    while world * 2 <= min(ndev, 8):
        world *= 2
    if world < 2:
        pytest.skip("one visible GPU: world > 1 over RCCL needs a multi-GPU node (world 1 is test_bench_under_torchrun_world1)")
    env = {k: v for k, v in os.environ.items() if k != "AKP_BENCH_SHARED_GPU"}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    p, full = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
                    "--master-port", "29561", "bench.py", "--gpus", str(world), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-host-path"]
                   + SMALL, tmp_path, timeout=1200, env=env)
    assert p.returncode == 0, (p.stdout[-1000:], p.stderr[-3000:])
    d = _check(p.stdout, full)
    assert d["n_gpus"] == world and d["launch"]["backend"].startswith("nccl") and sorted(d["launch"]["rank_devices"]) == list(range(world))
    assert d["merkle"]["sampled_parity_bit_exact"] and d["bh_merkle"]["sampled_parity_bit_exact"] and d["bh_merkle"]["leaves"] == world << 9


def test_bench_self_launches_ranks_from_plain_python(tmp_path):
    """`python bench.py --gpus 2` with no WORLD_SIZE must start two ranks itself (the driver launched N = 1 that way);
    here both share GPU 0 through the test hook."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["AKP_BENCH_SHARED_GPU"] = "1"
    args = ["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-host-path"] + SMALL
    p, full = _run([sys.executable, "bench.py", "--gpus", "2"] + args, tmp_path, env=env)
    assert p.returncode == 0, (p.stdout[-1000:], p.stderr[-3000:])
    d = _check(p.stdout, full)
    assert d["n_gpus"] == 2 and d["launch"]["ranks"] == 2
    # and it refuses when the devices are not there (no hook): 1-GPU box, 2 ranks asked
    env.pop("AKP_BENCH_SHARED_GPU")
    import torch
    if torch.cuda.device_count() < 2:
        p = subprocess.run([sys.executable, "bench.py", "--gpus", "2"] + args, cwd=ROOT, capture_output=True, text=True, timeout=300, env=env)
        assert p.returncode != 0 and "only 1 HIP device" in (p.stderr + p.stdout)


def test_bench_line_survives_a_failing_side_leg_world1(tmp_path):
    """a side leg that raises (an out-of-memory on a wide table, an RCCL error, a leg's own parity SystemExit) must not
    void the headline: the line is printed, rc 0, the leg is named under legs_failed / leg_errors, its scalars are absent, every other leg
    is there.  (AKP_BENCH_FAIL_LEG is the test hook of tools/bench_legs/runner.py.)"""
    env = dict(os.environ, AKP_BENCH_FAIL_LEG="pedersen")
    p, full = _run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--cpu-seconds", "0.5"] + SMALL, tmp_path, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    d = _check(p.stdout, full)
    ln = d["_line"]
    assert ln["legs_failed"] == ["pedersen"] and "injected" in ln["leg_errors"]["pedersen"] and "pedersen" not in d
    assert "pedersen_hashes_per_s" not in ln["legs"] and ln["legs"]["merkle_s"] > 0 and ln["legs"]["bh_s"] > 0 and ln["legs"]["ragged_bh_hashes_per_s"] > 0
    assert ln["cpu_baseline"]["value"] > 0 and "pedersen" not in d["cpu_baseline"]  # the CPU leg skips the part whose GPU leg is missing
    # ... and a leg that fails AFTER it has run (its own parity check, say)
    env = dict(os.environ, AKP_BENCH_FAIL_LEG="sweep:end")
    p, full = _run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-host-path"] + SMALL, tmp_path, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    assert _check(p.stdout, full)["_line"]["legs_failed"] == ["sweep"]


@pytest.mark.parametrize("where", ["bh_merkle@3", "merkle@0:end"])
def test_bench_line_survives_a_leg_failing_on_one_rank_world8(where, tmp_path):
    """the same at world 8 (ranks sharing GPU 0, gloo): ONE rank's leg raises -- before the leg's first collective, or after its last -- and
    the other seven must neither hang in a barrier nor lose step with it: every rank leaves the leg after the same number of
    all-reduces (runner.py), the line is printed by rank 0 with the leg named"""
    world = 8
    args = ["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-host-path"] + SMALL
    args[args.index("--log2-states") + 1] = "14"
    env = dict(os.environ, AKP_BENCH_SHARED_GPU="1", AKP_BENCH_FAIL_LEG=where)
    p, full = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
                    "127.0.0.1", "--master-port", str(29570 + len(where)), "bench.py", "--gpus", str(world)] + args, tmp_path, env=env)
    assert p.returncode == 0, (p.stdout[-1000:], p.stderr[-3000:])
    d = _check(p.stdout, full)
    leg = where.split("@")[0]
    other = "merkle_s" if leg == "bh_merkle" else "bh_s"
    assert d["_line"]["legs_failed"] == [leg] and d["n_gpus"] == world and d["_line"]["legs"][other] > 0
