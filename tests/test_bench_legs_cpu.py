"""The pure-python parts of bench.py's legs (tools/bench_legs): the scaling model, the sensor readers and the line-assembly helpers
run without a GPU -- a ZeroDivisionError in the model once took a whole bench run down."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_predicted_scaling_model():
    from bench_legs import scaling
    merkle = {"leaves": 1 << 24, "seconds": 0.0641, "one_process_c_abi": {"phases_ms": {"allgather_ms": 0.02}}}
    bh = {"leaves_per_gpu": 1 << 23, "seconds": 0.0246}
    p = scaling.predict(merkle, bh)
    ms = p["merkle_strong"]
    assert 0.9 < ms["2_gpus"]["efficiency"] <= 1.0 and ms["8_gpus"]["efficiency"] < ms["4_gpus"]["efficiency"] < ms["2_gpus"]["efficiency"]
    assert abs(ms["8_gpus"]["seconds"] * 8 * ms["8_gpus"]["efficiency"] - 0.0641) < 2e-3  # t(1) of the model reproduces the measured one-GPU time
    assert 0.97 < p["bh_merkle_weak"]["8_gpus"]["efficiency"] <= 1.0
    # trees with no wide level at all (the contract test's 2^12 leaves), missing legs, one-process leg absent
    tiny = scaling.predict({"leaves": 1 << 12, "seconds": 0.002}, {"leaves_per_gpu": 1 << 9, "seconds": 0.001})
    assert tiny["merkle_strong"]["8_gpus"]["seconds"] > 0 and tiny["bh_merkle_weak"]["2_gpus"]["efficiency"] <= 1.0
    assert "merkle_strong" not in scaling.predict(None, None) and json.dumps(scaling.predict(None, bh))


def test_sensors_and_constants_do_not_need_a_gpu():
    from bench_legs import common
    s = common.gpu_sensors()
    assert set(s) == {"power_w", "power_cap_w", "temp_c_max", "source"}
    info = common.cpu_info()
    assert info["logical_cpus"] >= 1
    assert common.valu_peak_wave_instr(2400.0) == common.VALU_PEAK_WAVE_INSTR and common.valu_peak_wave_instr(1200.0) * 2 == common.VALU_PEAK_WAVE_INSTR
    assert common.MADS_PER_PERM == 54522 or common.MADS_PER_PERM > 50000
    t = common.te_counters("pedersen_128B", 1 << 20)
    assert t["traffic"] > 192 * (1 << 20) and t["valu_instr"] > 1e9


def test_bench_curve_parity_string_and_help():
    import subprocess
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    s = bench.curve_parity_status()
    assert s.startswith("unpinned (emitter not run)") or s.startswith("pinned by")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and "--sweep-max-log2" in p.stdout and "--gpus" in p.stdout


def test_compact_line_of_committed_full_records_is_short_and_complete():
    """tools/bench_legs/line.py on every full record committed under profiles/ (round 4's 22 KB line among them): the printed line
    stays under the limit, carries the contract keys with scalar-only legs, and agrees with the record it was cut from"""
    import glob
    from bench_legs import line as line_mod
    recs = [p for p in glob.glob(os.path.join(ROOT, "profiles", "r0[4-9]_*", "bench*.json"))]
    assert recs
    seen = 0
    for p in recs:
        try:
            full = json.load(open(p))
        except ValueError:
            continue
        if not isinstance(full, dict) or "roofline" not in full or "parity" not in full or "timed_buffer_states_checked" not in full["parity"]:
            continue
        ln = line_mod.compact(full)
        s = json.dumps(ln)
        assert len(s) < line_mod.LIMIT == 6000, (p, len(s))
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                  "data", "config", "roofline", "cpu_baseline", "legs", "full"):
            assert k in ln, (p, k)
        assert ln["value"] == full["value"] and set(ln["config"]) == {"workload", "states_per_gpu", "parallelism"}
        assert all(not isinstance(v, (dict, list)) for v in ln["legs"].values())
        assert all(not isinstance(v, (dict, list)) for v in ln["roofline"].values())
        if full.get("cpu_baseline"):
            assert ln["cpu_baseline"]["value"] > 0 and ln["cpu_baseline"]["kind"] in ("port", "reference")
        else:
            assert ln["cpu_baseline"] is None
        seen += 1
    assert seen >= 1
    # N > 1 shape: no cpu_baseline in the record -> an explicit null in the line
    full = json.load(open(os.path.join(ROOT, "profiles", "r04_s14", "bench.json")))
    full.pop("cpu_baseline"), full.pop("gpu_over_cpu")
    full["n_gpus"] = 8
    assert line_mod.compact(full)["cpu_baseline"] is None


# ---- a failing side leg must not void the headline: tools/bench_legs/runner.py ----------------------------------
def test_leg_runner_single_process_catches_everything_a_leg_raises():
    import torch
    from bench_legs.runner import LegRunner
    r = LegRunner(torch, None, 0, 1, "cpu")
    assert r.run("fine", lambda x: {"v": x}, 3) == {"v": 3} and not r.errors

    def exits():
        raise SystemExit("parity check of this leg failed")

    def oom():
        raise MemoryError("hipErrorOutOfMemory")
    cleaned = []
    assert r.run("exits", exits, cleanup=lambda: cleaned.append(1)) is None
    assert r.run("oom", oom) is None
    assert r.run("after", lambda: 7) == 7  # the run goes on
    assert sorted(r.errors) == ["exits", "oom"] and "SystemExit" in r.errors["exits"] and "MemoryError" in r.errors["oom"] and cleaned == [1]
    os.environ["AKP_BENCH_FAIL_LEG"] = "pedersen"
    try:
        r2 = LegRunner(torch, None, 0, 1, "cpu")
        assert r2.run("merkle", lambda: 1) == 1 and r2.run("pedersen", lambda: 2) is None and "injected" in r2.errors["pedersen"]
    finally:
        del os.environ["AKP_BENCH_FAIL_LEG"]


RUNNER_WORKER = r'''
import os, sys, json
sys.path.insert(0, os.path.join(%(root)r, "tools"))
import torch, torch.distributed as dist
from datetime import timedelta
from bench_legs.runner import LegRunner
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world, timeout=timedelta(seconds=60))
r = LegRunner(torch, dist, rank, world, "cpu")
where, bad = os.environ["AKP_T_WHERE"], int(os.environ["AKP_T_RANK"])

def leg(tag):  # the shape of a real leg: barrier, work, barrier, max over ranks, rank-0-only check at the end
    r.barrier()
    if tag == "B" and where == "mid" and rank == bad:
        raise RuntimeError("rank %%d fell over between two barriers" %% rank)
    r.barrier()
    m = r.max_over_ranks(float(rank))
    assert m == world - 1
    if tag == "B" and where == "late" and rank == bad:
        raise SystemExit("rank-local parity check after the last collective")
    return {"leg": tag, "max": m}
a = r.run("A", leg, "A")
b = r.run("B", leg, "B")
c = r.run("C", leg, "C")   # every rank must arrive here in step, whichever rank failed where in B
assert a == {"leg": "A", "max": world - 1} and c == {"leg": "C", "max": world - 1}, (a, c)
assert sorted(r.errors) == ["B"], r.errors
if where == "mid":
    assert b is None
else:  # the failure came after the leg's last collective: the other ranks keep their result, and still name the leg as failed
    assert (b is None) == (rank == bad)
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok", json.dumps(r.errors))
'''


def _run_runner_world(world, where, bad):
    import socket
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for rk in range(world):
        env = dict(os.environ, RANK=str(rk), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), AKP_T_WHERE=where, AKP_T_RANK=str(bad))
        procs.append(subprocess.Popen([sys.executable, "-c", RUNNER_WORKER % {"root": ROOT}], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    for rk, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and ("rank %d ok" % rk) in o, "rank %d:\n%s" % (rk, o)


def test_leg_runner_world2_gloo_one_rank_fails_between_barriers():
    _run_runner_world(2, "mid", 1)


def test_leg_runner_world4_gloo_rank0_fails_between_barriers():
    _run_runner_world(4, "mid", 0)


def test_leg_runner_world4_gloo_one_rank_fails_after_the_last_collective():
    _run_runner_world(4, "late", 2)


def test_compact_line_names_failed_legs_and_static_counters():
    from bench_legs import line as line_mod
    import glob
    full = None
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "r05_s*", "bench_full.json"))) + sorted(glob.glob(os.path.join(ROOT, "profiles", "r04_s14", "bench.json"))):
        try:
            full = json.load(open(p))
        except ValueError:
            continue
        if isinstance(full, dict) and "roofline" in full and "timed_buffer_states_checked" in full.get("parity", {}):
            break
    assert full
    full.pop("pedersen", None)  # what a failed leg leaves behind: no object, an entry in leg_errors
    full["leg_errors"] = {"pedersen": "RuntimeError: hipErrorOutOfMemory " + "x" * 500}
    ln = line_mod.compact(full)
    assert ln["legs_failed"] == ["pedersen"] and len(ln["leg_errors"]["pedersen"]) <= 160 and "pedersen_hashes_per_s" not in ln["legs"]
    assert ln["roofline"]["traffic_measured_in_this_run"] is False
    assert ln["value"] == full["value"] and len(json.dumps(ln)) < line_mod.LIMIT
