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
