"""include/akp.hpp (C++ mirror of the reference's trait surface): compiles with g++ against the library; the
program itself needs a GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "test_akp_hpp")


def _build():
    lib = os.path.join(ROOT, "crypto_primitives_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", os.path.join(ROOT, "tests", "cpp", "test_akp_hpp.cpp"), "-o", EXE,
                           "-L", lib, "-lakp", f"-Wl,-rpath,{lib}", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])


def test_cpp_header_compiles_and_links():
    _build()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_cpp_header_program_runs():
    _build()
    p = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "OK" in p.stdout
    # root of the 8-leaf tree [1]..[8] (tests/golden/derived_vectors.json)
    import json
    d = json.load(open(os.path.join(ROOT, "tests", "golden", "derived_vectors.json")))
    root = int(d["poseidon_merkle_8"]["root"])
    line = [l for l in p.stdout.splitlines() if l.startswith("root limbs")][0]
    got = int("".join(line.split()[2:]), 16)
    assert got == root
