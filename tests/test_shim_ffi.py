"""The Rust shim's FFI declarations are generated from include/akp.h: the committed file must be the generator's output,
cover every symbol of the header, and name only symbols the built library exports (no GPU needed)."""
import ctypes as C
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ffi_rs_is_generated_from_the_header():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "shim", "gen_ffi.py")], capture_output=True, text=True, check=True).stdout
    assert out == open(os.path.join(ROOT, "shim", "src", "ffi.rs")).read(), "run: python3 shim/gen_ffi.py > shim/src/ffi.rs"


def test_ffi_prototypes_equal_header_symbols_and_library_exports():
    import crypto_primitives_amd as cpa
    rs = open(os.path.join(ROOT, "shim", "src", "ffi.rs")).read()
    protos = sorted(re.findall(r"pub fn (akp_[a-z0-9_]+)\(", rs))
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "akp.h")).read(), flags=re.S)
    syms = sorted(set(re.findall(r"\b(akp_[a-z0-9_]+)\s*\(", hdr)))
    assert protos == syms and len(protos) >= 71
    L = C.CDLL(cpa.LIB_PATH)
    for s in protos:
        assert hasattr(L, s), s
    # pointer mutability follows the header: const inputs are *const, outputs *mut, handle arrays *const *mut
    assert "akp_poseidon_crh_batch(p: *mut AkpPoseidon, inputs: *const u64, n: usize, elems_per_input: usize, out: *mut u64) -> i32" in rs
    assert "leaf_params: *const *mut AkpPoseidon" in rs and "akp_multi_ctx(m: *mut AkpMulti, i: i32) -> *mut AkpCtx" in rs
    # every extern the hand-written sources call exists in ffi.rs
    for f in ("runtime.rs", "poseidon.rs", "te.rs", "merkle.rs", "sharded.rs"):
        src = open(os.path.join(ROOT, "shim", "src", f)).read()
        for name in set(re.findall(r"ffi::(akp_[a-z0-9_]+)", src)):
            assert name in protos, (f, name)
        for const in set(re.findall(r"ffi::(AKP_[A-Z0-9_]+)", src)):
            assert "pub const %s:" % const in rs, (f, const)
