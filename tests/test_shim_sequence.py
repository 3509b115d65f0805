"""tests/cpp/test_shim_sequence.cpp = the Rust shim's call sequence through the C ABI (the shim itself cannot be compiled
in this image): compiles here, runs on the GPU box."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "test_shim_sequence")


def _build():
    lib = os.path.join(ROOT, "crypto_primitives_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-pthread", os.path.join(ROOT, "tests", "cpp", "test_shim_sequence.cpp"), "-o", EXE,
                           "-L", lib, "-lakp", f"-Wl,-rpath,{lib}", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])


def test_shim_sequence_compiles_and_links():
    _build()
    assert os.path.exists(EXE)


def _write_te_cases(tmp_path):
    """generators + one message + its oracle digest for the te.rs section (Pedersen 4x256 = the reference's Window4x256, a small
    Pedersen window, Bowe-Hopwood 63x9 and a small Bowe-Hopwood window)"""
    import struct
    import sys
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import jubjub as jj, pedersen as opd, bowe_hopwood as obh
    from helpers import gens_array, mont
    path = os.path.join(str(tmp_path), "shim_te_cases.bin")
    with open(path, "wb") as f:
        for kind, W, N, L in ((0, 4, 256, 128), (0, 5, 9, 3), (1, 63, 9, 32), (1, 7, 3, 6)):
            g = jj.pedersen_generators(190 + W, W, N) if kind == 0 else jj.bowe_hopwood_generators(191 + W, W, N)
            msg = bytes((41 * i + 7) & 0xFF for i in range(L))
            f.write(struct.pack("<4I", kind, W, N, L))
            f.write(np.ascontiguousarray(gens_array(g), dtype=np.uint64).tobytes())
            f.write(msg)
            want = list(opd.evaluate(g, W, N, msg)) if kind == 0 else [obh.evaluate(g, W, N, msg)]
            f.write(mont(want).tobytes())
    return path


@pytest.mark.gpu
def test_shim_sequence_runs(tmp_path):
    _build()
    p = subprocess.run([EXE, _write_te_cases(tmp_path)], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "OK" in p.stdout and "te cases 4" in p.stdout, p.stdout + p.stderr
