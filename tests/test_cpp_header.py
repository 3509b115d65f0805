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


def _write_te_cases(tmp_path):
    """generators + message + expected digest (oracle) for the Pedersen / Bowe-Hopwood classes of akp.hpp"""
    import struct
    import sys
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import jubjub as jj, pedersen as opd, bowe_hopwood as obh
    from helpers import gens_array, mont
    path = os.path.join(str(tmp_path), "te_cases.bin")
    with open(path, "wb") as f:
        for kind, W, N, L in ((0, 4, 16, 8), (0, 5, 9, 3), (1, 7, 3, 6)):
            g = jj.pedersen_generators(90 + W, W, N) if kind == 0 else jj.bowe_hopwood_generators(91 + W, W, N)
            msg = bytes((37 * i + 11) & 0xFF for i in range(L))
            f.write(struct.pack("<4I", kind, W, N, L))
            f.write(np.ascontiguousarray(gens_array(g), dtype=np.uint64).tobytes())
            f.write(msg)
            want = list(opd.evaluate(g, W, N, msg)) if kind == 0 else [obh.evaluate(g, W, N, msg)]
            f.write(mont(want).tobytes())
    return path


def test_cpp_header_compiles_and_links():
    _build()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_cpp_header_program_runs(tmp_path):
    _build()
    case_file = _write_te_cases(tmp_path)
    p = subprocess.run([EXE, case_file], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "OK" in p.stdout and "te cases 3" in p.stdout
    # root of the 8-leaf tree [1]..[8] (tests/golden/derived_vectors.json)
    import json
    d = json.load(open(os.path.join(ROOT, "tests", "golden", "derived_vectors.json")))
    root = int(d["poseidon_merkle_8"]["root"])
    line = [l for l in p.stdout.splitlines() if l.startswith("root limbs")][0]
    got = int("".join(line.split()[2:]), 16)
    assert got == root


def test_cpp_serialize_wrappers_against_the_oracle_bytes(tmp_path):
    """akp.hpp `serialize` (over akp_serialize_* / akp_deserialize_*; host only, no GPU): parse the ORACLE's bytes, write them
    again, require identity -- Path, MultiPath, Parameters (both modes), PoseidonConfig; a truncated Path throws code 1"""
    import struct
    from oracle import serialize as oser, jubjub as jj, poseidon as po
    lib = os.path.join(ROOT, "crypto_primitives_amd", "lib")
    exe = os.path.join(ROOT, "tests", "cpp", "test_serialize_host")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", os.path.join(ROOT, "tests", "cpp", "test_serialize_host.cpp"), "-o", exe,
                           "-L", lib, "-lakp", f"-Wl,-rpath,{lib}", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"])
    cases = []
    d = [(1234567 * (i + 1) ** 5 % jj.Q,) for i in range(12)]
    pb = oser.path(d[0], d[1:5], 11, False)
    cases.append((0, 0, 0, 0, pb))
    cases.append((0, 1, 0, 0, oser.path(d[5], [], 0, True)))
    cases.append((1, 0, 0, 0, oser.multi_path(d[0:3], [0, 2, 1], [d[3:6], d[6:7], d[7:9]], [1, 2, 7], False)))
    g = jj.pedersen_generators(21, 3, 2)
    for compress in (0, 1):
        cases.append((2, compress, 3, 2, oser.te_parameters(g, bool(compress))))
    if True:
        import json
        k = json.load(open(os.path.join(ROOT, "tests", "golden", "emitter_inputs.json")))["poseidon"]
        cb = oser.poseidon_config(k["full_rounds"], k["partial_rounds"], k["alpha"], [[int(x) for x in r] for r in k["ark"]],
                                  [[int(x) for x in r] for r in k["mds"]], k["rate"], k["capacity"])
        want_cfg = "config rounds %d+%d alpha %d rate %d capacity %d" % (k["full_rounds"], k["partial_rounds"], k["alpha"], k["rate"], k["capacity"])
    cases.append((3, 0, 0, 0, cb))
    cases.append((4, 0, 0, 0, pb[:-5]))
    path = os.path.join(str(tmp_path), "ser_cases.bin")
    with open(path, "wb") as f:
        for kind, compress, a, b, payload in cases:
            f.write(struct.pack("<4IQ", kind, compress, a, b, len(payload)) + payload)
    p = subprocess.run([exe, path], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    out = p.stdout.splitlines()
    assert out[0] == "path depth 4 index 11" and out[1] == "path depth 0 index 0" and out[2] == "multipath m 3 suffix digests 6"
    assert out[3] == out[4] == "parameters 3 x 2" and out[5] == want_cfg and out[-1] == "OK %d cases" % len(cases)


def test_te_shape_arithmetic():
    """csrc/te_shape.hpp (what capi_te.hip decides table shapes, step counts and table growth with) against brute force: host only"""
    exe = os.path.join(ROOT, "tests", "cpp", "test_te_shape")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", os.path.join(ROOT, "tests", "cpp", "test_te_shape.cpp"), "-o", exe])
    p = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and p.stdout.strip() == "OK", p.stdout + p.stderr
