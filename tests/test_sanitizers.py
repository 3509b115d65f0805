"""Sanitizer runs of the host-side code ("race detection / sanitizers"; the reference's counterpart is
`#![forbid(unsafe_code)]`, crypto-primitives/src/lib.rs:9).

  * tests/host_harness `make asan`  : the PRODUCT's per-item device logic (the __host__ __device__ functions the kernels wrap:
    message-bit access with the pulled-back 32-bit loads, table construction and gathers, shared-inversion arrays, digest
    serialisation, the whole round code) compiled for the CPU with AddressSanitizer; the complete harness suite runs on it.
  * tests/host_harness `make ubsan` : the same with signed-integer-overflow / shift / alignment / bounds checks -- the lazy
    radix-2^29 arithmetic rests on accumulator bounds, so a signed overflow on the CPU is a wrong digest on the GPU.
  * oracle `make asan`              : the C oracle (the checker of every parity test) with ASan + UBSan.
Each run is a child pytest process with the sanitizer runtime preloaded; a deliberately wrong call shows the sanitizer is live.
The GPU side of the same row is tests/test_gpu_canaries.py (guard bands around every caller buffer).
"""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = os.path.join(ROOT, "tests", "host_harness")
RT_DIRS = glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux")


def _rt(name):
    for d in RT_DIRS:
        p = os.path.join(d, name)
        if os.path.exists(p):
            return p
    pytest.skip("clang sanitizer runtime %s not found" % name)


def _child(env_extra, args, timeout=1500):
    env = dict(os.environ)
    env.update(env_extra)
    env.pop("PYTEST_CURRENT_TEST", None)
    workers = ["-n", "4"] if _have_xdist() else []  # the sanitized builds run 3-10x slower: spread the cases over a few processes
    return subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + workers + args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def _have_xdist():
    try:
        import xdist  # noqa: F401
        return True
    except Exception:
        return False


@pytest.fixture(scope="module", autouse=True)
def _sanitizer_builds():
    """the three sanitizer builds side by side (2.5 min + 40 s + 2 s when nothing is built yet; no-ops afterwards)"""
    jobs = [subprocess.Popen(["make", "-C", HARNESS, "-s", "asan"]), subprocess.Popen(["make", "-C", HARNESS, "-s", "ubsan"]),
            subprocess.Popen(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"], stderr=subprocess.DEVNULL),
            subprocess.Popen(["make", "-C", os.path.join(ROOT, "crypto_primitives_amd", "csrc"), "-s", "asan"], stdout=subprocess.DEVNULL),
            subprocess.Popen(["make", "-C", os.path.join(ROOT, "crypto_primitives_amd", "csrc"), "-s", "ubsan"], stdout=subprocess.DEVNULL)]
    for j in jobs:
        assert j.wait() == 0, "a sanitizer build failed"


def _assert_clean(cp, what):
    tail = (cp.stdout[-3000:] + "\n" + cp.stderr[-3000:])
    assert cp.returncode == 0, "%s failed under the sanitizer:\n%s" % (what, tail)
    assert "AddressSanitizer" not in tail and "runtime error:" not in tail, tail
    assert " passed" in cp.stdout


def test_host_harness_under_address_sanitizer():
    so = os.path.join(HARNESS, "harness_asan.so")
    env = {"LD_PRELOAD": _rt("libclang_rt.asan-x86_64.so"), "ASAN_OPTIONS": "detect_leaks=0:abort_on_error=0", "AKP_HARNESS_SO": so}
    cp = _child(env, ["tests/test_host_harness.py", "tests/test_host_wide_rows.py"])
    _assert_clean(cp, "host harness (ASan)")
    # the sanitizer is live: the same library, a digest buffer one byte too short -> heap-buffer-overflow report
    code = ("import ctypes as C, numpy as np\n"
            "h = C.CDLL(%r)\n"
            "vp = C.c_void_p\n"
            "h.hh_te_serialize_pairs.argtypes = [vp, vp, C.c_uint32, C.c_size_t, vp, C.c_size_t]\n"
            "l = np.zeros((64, 4), np.uint64); r = np.zeros((64, 4), np.uint64)\n"
            "buf = np.empty(64 * 64 - 1, np.uint8)\n"   # 64 pairs of 32-byte digests need 4096 bytes (malloc'ed: numpy caches only small blocks)
            "h.hh_te_serialize_pairs(l.ctypes.data, r.ctypes.data, 1, 64, buf.ctypes.data, 64)\n"
            "print('not caught')\n") % so
    e = dict(os.environ)
    e.update(env)
    cp = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=300)
    assert cp.returncode != 0 and "AddressSanitizer" in cp.stderr and "not caught" not in cp.stdout, cp.stderr[-1500:]


def test_host_harness_under_ub_sanitizer():
    so = os.path.join(HARNESS, "harness_ubsan.so")
    env = {"LD_PRELOAD": _rt("libclang_rt.ubsan_standalone-x86_64.so"), "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=0", "AKP_HARNESS_SO": so}
    # host pass at -O0: the widest Bowe-Hopwood group tables and rate-8 round loops take 5-15 s each there and add no new
    # arithmetic routine (the same f29 functions run in the smaller cases), so they are left to the ASan run
    cp = _child(env, ["tests/test_host_harness.py", "tests/test_host_wide_rows.py", "-k",
                      "not (bowe_hopwood_table_path and (7-3-5 or 6-2-5)) and not (round_code and 8-False) and not many_partial_full_form "
                      "and not (ragged_items and (63-9 or 4-256 or 7-3-5))"])  # the full-size ragged windows: 6 minutes at -O0; the small windows run the same per-item code
    _assert_clean(cp, "host harness (UBSan)")
    # live check: f29_dot3 on limbs far beyond its documented bound must overflow its signed 64-bit column accumulator
    code = ("import ctypes as C, numpy as np\n"
            "h = C.CDLL(%r)\n"
            "vp = C.c_void_p\n"
            "h.hh_f29_raw_mul.argtypes = [vp, vp, C.c_int, vp]\n"
            "a = np.full(9, 0x7fffffff, np.uint32); o = np.zeros((3, 4), np.uint64)\n"
            "h.hh_f29_raw_mul(a.ctypes.data, a.ctypes.data, 1 | 2, o.ctypes.data)\n"
            "print('not caught')\n") % so
    e = dict(os.environ)
    e.update(env)
    cp = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=300)
    assert cp.returncode != 0 and "signed integer overflow" in cp.stderr and "not caught" not in cp.stdout, cp.stderr[-1500:]


def test_c_oracle_under_sanitizers():
    so = os.path.join(ROOT, "oracle", "_build", "libakp_oracle_asan.so")
    env = {"LD_PRELOAD": _rt("libclang_rt.asan-x86_64.so"), "ASAN_OPTIONS": "detect_leaks=0", "UBSAN_OPTIONS": "halt_on_error=1", "AKP_ORACLE_SO": so}
    cp = _child(env, ["tests/test_oracle_poseidon.py", "tests/test_oracle_curves.py", "tests/test_reference_vectors.py", "-m", "not gpu"])
    _assert_clean(cp, "C oracle (ASan + UBSan)")


def test_byte_format_readers_of_the_library_under_sanitizers():
    """libakp_asan.so (`make -C crypto_primitives_amd/csrc asan`: the product library with AddressSanitizer on its host code) behind
    the byte-format tests: every struct against the oracle's bytes, the rejection cases, and the mutation fuzz of
    tests/test_serialize_fuzz.py -- akp_deserialize_* is the one place where the library parses bytes it did not produce"""
    so = os.path.join(ROOT, "crypto_primitives_amd", "lib", "libakp_asan.so")
    env = {"LD_PRELOAD": _rt("libclang_rt.asan-x86_64.so"), "ASAN_OPTIONS": "detect_leaks=0:abort_on_error=0", "AKP_LIB": so}
    # (+ tests/test_abi.py: the other host-only entry points -- parameter generation, MultiPath encode / decode, path gathering)
    cp = _child(env, ["tests/test_serialize_cpu.py", "tests/test_serialize_fuzz.py", "tests/test_abi.py"], timeout=1500)
    _assert_clean(cp, "byte-format readers (ASan)")
    # the sanitizer is live in THAT library: a digest array one element too short for akp_deserialize_digests -> heap-buffer-overflow
    code = ("import ctypes as C, numpy as np\n"
            "L = C.CDLL(%r)\n"
            "vp = C.c_void_p\n"
            "L.akp_deserialize_digests.argtypes = [vp, C.c_size_t, C.c_size_t, C.c_uint32, C.c_int32, C.c_int32, vp]\n"
            "src = np.zeros(4096 * 32, np.uint8)\n"
            "dst = np.empty((4095, 4), np.uint64)\n"   # 4096 field elements need 4096 x 4 words
            "rc = L.akp_deserialize_digests(src.ctypes.data, src.size, 4096, 1, 0, 1, dst.ctypes.data)\n"
            "print('not caught', rc)\n") % so
    e = dict(os.environ)
    e.update(env)
    cp = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=300)
    assert cp.returncode != 0 and "AddressSanitizer" in cp.stderr and "not caught" not in cp.stdout, cp.stderr[-1500:]
    # the same tests against libakp_ubsan.so (`make ... ubsan`): signed overflow, shifts, alignment and bounds in the host code --
    # the field arithmetic of the point decoders (square roots, curve and subgroup checks) and the cursor arithmetic of the readers
    so = os.path.join(ROOT, "crypto_primitives_amd", "lib", "libakp_ubsan.so")
    env = {"LD_PRELOAD": _rt("libclang_rt.ubsan_standalone-x86_64.so"), "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=0", "AKP_LIB": so}
    cp = _child(env, ["tests/test_serialize_cpu.py", "tests/test_serialize_fuzz.py", "tests/test_abi.py"], timeout=1500)
    _assert_clean(cp, "byte-format readers (UBSan)")
