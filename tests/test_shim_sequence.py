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


@pytest.mark.gpu
def test_shim_sequence_runs():
    _build()
    p = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "OK" in p.stdout, p.stdout + p.stderr
