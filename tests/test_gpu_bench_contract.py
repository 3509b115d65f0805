"""bench.py contract on a GPU box: one JSON line with the required keys, also when launched through
torch.distributed.run (world size 1 exercises the nccl init / barrier / all-gather code path)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline"}


def _check(out):
    line = [l for l in out.strip().splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert REQUIRED <= set(d), REQUIRED - set(d)
    assert d["parity_probe_bit_exact"] is True and d["value"] > 1e6
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert 0.5 < r["traffic"] / r["algorithmic_bytes_per_launch"] < 2.0
    return d


def test_bench_single_process():
    p = subprocess.run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--log2-states", "16", "--merkle-log2", "12",
                        "--cpu-seconds", "0.5"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    d = _check(p.stdout)
    assert d["n_gpus"] == 1 and d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1


def test_bench_under_torchrun_world1():
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", "29531", "bench.py", "--gpus", "1", "--steps", "2", "--warmup", "1", "--log2-states", "14",
                        "--merkle-log2", "10", "--bh-merkle-log2", "8", "--no-cpu-baseline"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout[-1000:], p.stderr[-3000:])
    d = _check(p.stdout)
    assert d["n_gpus"] == 1 and "merkle" in d and d["bh_merkle"]["leaves"] == 256


@pytest.mark.parametrize("world", [2, 4])
def test_bench_multi_rank_code_path_on_one_gpu(world):
    """The N > 1 branch of bench.py (rank env, barriers, MAX-reduced timing, sharded Merkle legs, rank-0 printing) with
    `world` ranks sharing GPU 0 and gloo carrying the collectives (AKP_BENCH_SHARED_GPU=1, a test hook)."""
    args = ["--steps", "2", "--warmup", "1", "--log2-states", "14", "--merkle-log2", "12", "--bh-merkle-log2", "10", "--no-cpu-baseline"]
    env = dict(os.environ, AKP_BENCH_SHARED_GPU="1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
                        "127.0.0.1", "--master-port", str(29540 + world), "bench.py", "--gpus", str(world)] + args,
                       cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, (p.stdout[-1000:], p.stderr[-3000:])
    d = _check(p.stdout)
    assert d["n_gpus"] == world and d["config"]["states_per_gpu"] == 1 << 14 and "cpu_baseline" not in d
    assert d["merkle"]["leaves"] == 1 << 12 and d["bh_merkle"]["leaves"] == 1 << 10
    assert len([l for l in p.stdout.splitlines() if l.startswith("{")]) == 1  # rank 0 only
