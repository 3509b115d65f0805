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
