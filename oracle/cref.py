"""ctypes binding of oracle/_build/libakp_oracle.so (TEST INFRASTRUCTURE ONLY).

Used by tests/ (bulk parity), __graft_entry__.smoke() and bench.py's cpu_baseline.
All field arrays are numpy uint64 [..., 4] Montgomery limbs (the C-ABI wire format).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("AKP_ORACLE_SO") or os.path.join(_HERE, "_build", "libakp_oracle.so")  # AKP_ORACLE_SO: the sanitizer build (tests/test_sanitizers.py)
_lib = None

u64p = C.POINTER(C.c_uint64)
u8p = C.POINTER(C.c_uint8)


def build():
    subprocess.check_call(["make", "-C", _HERE, "-s"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        L.orc_poseidon_new.restype = C.c_void_p
        L.orc_poseidon_new.argtypes = [C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32, u64p, u64p]
        L.orc_poseidon_free.argtypes = [C.c_void_p]
        L.orc_poseidon_permute_batch.argtypes = [C.c_void_p, u64p, C.c_size_t, C.c_int]
        L.orc_poseidon_crh_batch.argtypes = [C.c_void_p, u64p, C.c_size_t, C.c_size_t, u64p, C.c_int]
        L.orc_poseidon_two_to_one_batch.argtypes = [C.c_void_p, u64p, u64p, C.c_size_t, u64p, C.c_int]
        L.orc_poseidon_merkle_build.argtypes = [C.c_void_p, C.c_void_p, u64p, C.c_size_t, C.c_size_t, u64p, u64p, C.c_int]
        L.orc_poseidon_merkle_build.restype = C.c_int
        L.orc_poseidon_sponge_script.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_size_t, u64p, u64p]
        L.orc_te_params_new.restype = C.c_void_p
        L.orc_te_params_new.argtypes = [C.c_uint32, C.c_uint32, u64p]
        L.orc_te_params_free.argtypes = [C.c_void_p]
        L.orc_pedersen_crh_batch.argtypes = [C.c_void_p, u8p, C.c_size_t, C.c_size_t, u64p, C.c_int]
        L.orc_pedersen_crh_batch.restype = C.c_int
        L.orc_bh_crh_batch.argtypes = [C.c_void_p, u8p, C.c_size_t, C.c_size_t, u64p, C.c_int]
        L.orc_bh_crh_batch.restype = C.c_int
        L.orc_curve_merkle_build.argtypes = [C.c_int, C.c_void_p, C.c_void_p, u8p, C.c_size_t, C.c_size_t, u64p, u64p, C.c_int]
        L.orc_curve_merkle_build.restype = C.c_int
        L.orc_fr_to_mont.argtypes = [u64p, u64p, C.c_size_t]
        L.orc_fr_from_mont.argtypes = [u64p, u64p, C.c_size_t]
        L.orc_fr_mul.argtypes = [u64p, u64p, u64p]
        L.orc_hardware_threads.restype = C.c_int
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(u64p)


def _b(a):
    return a.ctypes.data_as(u8p)


def _c(a, dtype=np.uint64):
    return np.ascontiguousarray(a, dtype=dtype)


def hardware_threads():
    return lib().orc_hardware_threads()


def to_mont(canon):
    canon = _c(canon)
    out = np.empty_like(canon)
    lib().orc_fr_to_mont(_p(canon), _p(out), canon.size // 4)
    return out


def from_mont(mont):
    mont = _c(mont)
    out = np.empty_like(mont)
    lib().orc_fr_from_mont(_p(mont), _p(out), mont.size // 4)
    return out


class Poseidon:
    """holds an orc_poseidon handle built from Montgomery-limb ark/mds arrays."""

    def __init__(self, full_rounds, partial_rounds, alpha, rate, capacity, ark_mont, mds_mont):
        self.t = rate + capacity
        self.rate, self.capacity = rate, capacity
        ark_mont, mds_mont = _c(ark_mont), _c(mds_mont)
        assert ark_mont.size == (full_rounds + partial_rounds) * self.t * 4
        assert mds_mont.size == self.t * self.t * 4
        self.h = lib().orc_poseidon_new(full_rounds, partial_rounds, alpha, rate, capacity, _p(ark_mont), _p(mds_mont))
        assert self.h

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_poseidon_free(self.h)
            self.h = None

    def permute_batch(self, states, threads=1):
        st = _c(states).copy()
        n = st.size // (4 * self.t)
        lib().orc_poseidon_permute_batch(self.h, _p(st), n, threads)
        return st

    def crh_batch(self, inputs, elems_per_input, threads=1):
        inp = _c(inputs)
        n = inp.size // (4 * elems_per_input) if elems_per_input else inp.shape[0]
        out = np.empty((n, 4), dtype=np.uint64)
        lib().orc_poseidon_crh_batch(self.h, _p(inp), n, elems_per_input, _p(out), threads)
        return out

    def crh_empty(self):
        out = np.empty((1, 4), dtype=np.uint64)
        dummy = np.zeros(4, dtype=np.uint64)
        lib().orc_poseidon_crh_batch(self.h, _p(dummy), 1, 0, _p(out), 1)
        return out

    def two_to_one_batch(self, left, right, threads=1):
        l, r = _c(left), _c(right)
        n = l.size // 4
        out = np.empty((n, 4), dtype=np.uint64)
        lib().orc_poseidon_two_to_one_batch(self.h, _p(l), _p(r), n, _p(out), threads)
        return out

    def sponge_script(self, ops, inputs, n_out):
        ops = np.ascontiguousarray(ops, dtype=np.int32)
        inp = _c(inputs) if len(inputs) else np.zeros(4, dtype=np.uint64)
        out = np.empty((max(n_out, 1), 4), dtype=np.uint64)
        lib().orc_poseidon_sponge_script(self.h, ops.ctypes.data_as(C.POINTER(C.c_int32)), len(ops), _p(inp), _p(out))
        return out[:n_out]

    def merkle_build(self, two_to_one: "Poseidon", leaves, leaf_len, threads=1):
        lv = _c(leaves)
        n = lv.size // (4 * leaf_len)
        leaf_nodes = np.empty((n, 4), dtype=np.uint64)
        non_leaf = np.empty((n - 1, 4), dtype=np.uint64)
        rc = lib().orc_poseidon_merkle_build(self.h, two_to_one.h, _p(lv), n, leaf_len, _p(leaf_nodes), _p(non_leaf), threads)
        if rc:
            raise ValueError("leaves.len() should be power of two and greater than one")
        return leaf_nodes, non_leaf


class CurveParams:
    """Pedersen / Bowe-Hopwood generators: uint64 [N, W, 2, 4] affine Montgomery."""

    def __init__(self, window_size, num_windows, gens_affine_mont):
        g = _c(gens_affine_mont)
        assert g.size == window_size * num_windows * 8
        self.window_size, self.num_windows = window_size, num_windows
        self.h = lib().orc_te_params_new(window_size, num_windows, _p(g))

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_te_params_free(self.h)
            self.h = None

    def pedersen_crh_batch(self, msgs, n, msg_len, threads=1):
        m = _c(msgs, np.uint8)
        out = np.empty((n, 2, 4), dtype=np.uint64)
        if lib().orc_pedersen_crh_batch(self.h, _b(m), n, msg_len, _p(out), threads):
            raise ValueError("incorrect input length")
        return out

    def bh_crh_batch(self, msgs, n, msg_len, threads=1):
        m = _c(msgs, np.uint8)
        out = np.empty((n, 4), dtype=np.uint64)
        if lib().orc_bh_crh_batch(self.h, _b(m), n, msg_len, _p(out), threads):
            raise ValueError("incorrect input length")
        return out

    def merkle_build(self, kind, two_to_one: "CurveParams", leaves, n, leaf_len, threads=1):
        """kind 0 = pedersen (digest x||y), 1 = bowe-hopwood (digest x)."""
        m = _c(leaves, np.uint8)
        nfe = 2 if kind == 0 else 1
        leaf_nodes = np.empty((n, nfe, 4), dtype=np.uint64)
        non_leaf = np.empty((n - 1, nfe, 4), dtype=np.uint64)
        rc = lib().orc_curve_merkle_build(kind, self.h, two_to_one.h, _b(m), n, leaf_len, _p(leaf_nodes), _p(non_leaf), threads)
        if rc:
            raise ValueError("bad merkle input")
        return leaf_nodes, non_leaf
