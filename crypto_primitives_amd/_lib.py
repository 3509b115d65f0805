"""ctypes binding of the product library (include/akp.h -> crypto_primitives_amd/lib/libakp.so).

There is no fallback of any kind: if the shared library is missing this module raises at
import, and every compute call fails loudly (AkpError) when no HIP device is usable.
"""
import ctypes as C
import importlib.util
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AKP_LIB", os.path.join(_HERE, "lib", "libakp.so"))


def _keep_large_buffers_out_of_the_heap():
    """glibc raises its mmap threshold (128 KB at start) up to 32 MB once a large block has been freed: from then on multi-megabyte numpy /
    torch CPU buffers live in the brk heap, back to back with each other and with small objects.  The host-pointer entry points hand
    PAGEABLE buffers to the HIP runtime, which pins their pages for the duration of a copy (sources read-only); with heap-resident
    buffers the full GPU test-suite of round 6 -- whose early tests free 100 MB arrays -- ended twice in twelve runs with "Memory access
    fault ... Write access to a read-only page" at a heap address, in a device-to-host copy of a LATER test (profiles/r06_s38); with every
    large buffer in a mapping of its own (the situation of rounds 1-5) it never has.  Fixing the threshold restores that.  A host that
    wants no part of this sets AKP_KEEP_MALLOC=1 -- or passes pinned / registered buffers, which never go through the runtime's pinning."""
    if os.environ.get("AKP_KEEP_MALLOC") == "1":
        return
    try:
        C.CDLL("libc.so.6").mallopt(-3, 128 * 1024)  # M_MMAP_THRESHOLD: a fixed value also switches the dynamic adjustment off
    except Exception:
        pass


_keep_large_buffers_out_of_the_heap()

AKP_OK, AKP_ERR_BAD_LENGTH, AKP_ERR_BAD_PARAMS, AKP_ERR_HIP, AKP_ERR_RCCL, AKP_ERR_NOT_POW2 = 0, 1, 2, 3, 4, 5
AKP_ABI_VERSION = 5
TE_PEDERSEN, TE_BOWE_HOPWOOD, TE_PEDERSEN_X = 0, 1, 2


class AkpError(RuntimeError):
    """lib.rs:47-52 `Error` analogue carrying the C-ABI status code."""

    def __init__(self, code, msg):
        super().__init__(f"akp error {code}: {msg}")
        self.code = code


class TeBuildReport(C.Structure):
    """akp_te_build_report (include/akp.h): phases of the last build / extension of a handle's wide curve table"""
    _fields_ = [("table_bytes", C.c_uint64), ("shape", C.c_uint32), ("units_from", C.c_uint32), ("units_to", C.c_uint32), ("units_total", C.c_uint32),
                ("in_background", C.c_uint32), ("upgrade_state", C.c_uint32), ("alloc_ms", C.c_double), ("parts_ms", C.c_double), ("combine_ms", C.c_double),
                ("constants_ms", C.c_double), ("total_ms", C.c_double), ("note", C.c_char * 96)]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_}
        d["note"] = self.note.decode(errors="replace")
        d["upgrade"] = {0: "single table", 1: "wide table pending (hashing on the cache-sized one)", 2: "hashing on the wide table",
                        3: "wide table could not be built (staying on the cache-sized one)"}.get(self.upgrade_state, "?")
        return d


class IncorrectInputLength(AkpError):
    """Error::IncorrectInputLength / the reference's input-length panics."""


class NotPowerOfTwo(AkpError):
    """merkle_tree/mod.rs:430-433 assertion."""


def _preload_torch_hip_runtime():
    # torch wheels bundle their own libamdhip64.so (same SONAME as /opt/rocm's).  If torch is used
    # in this process both must resolve to ONE runtime instance, or device pointers cannot be
    # shared; load torch's copy first so libakp.so binds to it.
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C crypto_primitives_amd/csrc` (there is no CPU fallback)")
    _preload_torch_hip_runtime()
    L = C.CDLL(LIB_PATH)
    vp, u64p, u8p, sz = C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t
    i32, u32, u64 = C.c_int32, C.c_uint32, C.c_uint64
    pp = C.POINTER(C.c_void_p)
    sig = {
        "akp_abi_version": (i32, []),
        "akp_last_error": (C.c_char_p, []),
        "akp_device_count": (i32, []),
        "akp_ctx_create": (i32, [i32, pp]),
        "akp_ctx_destroy": (None, [vp]),
        "akp_ctx_synchronize": (i32, [vp]),
        "akp_ctx_set_table_budget": (i32, [vp, sz]),
        "akp_ctx_table_budget": (sz, [vp]),
        "akp_ctx_stream": (vp, [vp]),
        "akp_clock_probe_dev": (i32, [vp, u32, u64p, vp]),
        "akp_fr_to_mont": (i32, [u64p, u64p, sz]),
        "akp_fr_from_mont": (i32, [u64p, u64p, sz]),
        "akp_poseidon_params_create": (i32, [vp, u32, u32, u64, u32, u32, u64p, u64p, pp]),
        "akp_poseidon_default_params": (i32, [vp, u32, i32, pp]),
        "akp_poseidon_params_destroy": (None, [vp]),
        "akp_poseidon_params_dims": (i32, [vp, C.POINTER(u32), C.POINTER(u32), C.POINTER(u64), C.POINTER(u32), C.POINTER(u32)]),
        "akp_poseidon_params_export": (i32, [vp, u64p, u64p]),
        "akp_poseidon_permute_batch": (i32, [vp, u64p, sz]),
        "akp_poseidon_permute_batch_dev": (i32, [vp, u64p, sz, vp]),
        "akp_poseidon_crh_batch": (i32, [vp, u64p, sz, sz, u64p]),
        "akp_poseidon_crh_batch_dev": (i32, [vp, u64p, sz, sz, u64p, vp]),
        "akp_poseidon_two_to_one_batch": (i32, [vp, u64p, u64p, sz, u64p]),
        "akp_poseidon_two_to_one_batch_dev": (i32, [vp, u64p, u64p, sz, u64p, vp]),
        "akp_sponge_create": (i32, [vp, sz, pp]),
        "akp_sponge_destroy": (None, [vp]),
        "akp_sponge_absorb": (i32, [vp, u64p, sz]),
        "akp_sponge_squeeze": (i32, [vp, u64p, sz]),
        "akp_sponge_absorb_dev": (i32, [vp, u64p, sz, vp]),
        "akp_sponge_squeeze_dev": (i32, [vp, u64p, sz, vp]),
        "akp_sponge_get_state": (i32, [vp, u64p, C.POINTER(i32), C.POINTER(u32)]),
        "akp_sponge_set_state": (i32, [vp, u64p, i32, u32]),
        "akp_te_params_create": (i32, [vp, i32, u32, u32, u64p, pp]),
        "akp_te_params_create_shaped": (i32, [vp, i32, u32, u32, u64p, u32, pp]),
        "akp_te_params_destroy": (None, [vp]),
        "akp_te_params_info": (i32, [vp, vp, vp, vp, sz, vp]),
        "akp_te_params_prepare": (i32, [vp, sz]),
        "akp_te_params_prepare_compress": (i32, [vp]),
        "akp_te_params_table_info": (i32, [vp, C.POINTER(u64), C.POINTER(u32), C.POINTER(u64), C.POINTER(TeBuildReport)]),
        "akp_te_entry_bytes": (u32, []),
        "akp_te_crh_batch": (i32, [vp, u8p, sz, sz, u64p]),
        "akp_te_crh_batch_dev": (i32, [vp, u8p, sz, sz, u64p, vp]),
        "akp_te_crh_batch_ragged": (i32, [vp, u8p, u64p, sz, u64p]),
        "akp_te_crh_batch_ragged_dev": (i32, [vp, u8p, u64p, sz, sz, u64p, vp]),
        "akp_poseidon_crh_batch_ragged": (i32, [vp, u64p, u64p, sz, u64p]),
        "akp_poseidon_crh_batch_ragged_dev": (i32, [vp, u64p, u64p, sz, u64p, vp]),
        "akp_merkle_tree_build_poseidon_ragged": (i32, [vp, vp, u64p, u64p, sz, pp]),
        "akp_merkle_tree_build_te_ragged": (i32, [vp, vp, u8p, u64p, sz, pp]),
        "akp_multi_tree_build_poseidon_ragged": (i32, [vp, pp, pp, u64p, u64p, sz, pp]),
        "akp_multi_tree_build_te_ragged": (i32, [vp, pp, pp, u8p, u64p, sz, pp]),
        "akp_te_two_to_one_batch": (i32, [vp, u8p, u8p, sz, sz, u64p]),
        "akp_te_compress_batch": (i32, [vp, u64p, u64p, sz, u64p]),
        "akp_merkle_build_poseidon": (i32, [vp, vp, u64p, sz, sz, u64p, u64p, u64p]),
        "akp_merkle_build_poseidon_dev": (i32, [vp, vp, u64p, sz, sz, u64p, u64p, vp]),
        "akp_merkle_inner_poseidon_dev": (i32, [vp, u64p, sz, u64p, vp]),
        "akp_merkle_inner_poseidon": (i32, [vp, u64p, sz, u64p]),
        "akp_merkle_inner_te": (i32, [vp, u64p, sz, u64p]),
        "akp_merkle_inner_te_dev": (i32, [vp, u64p, sz, u64p, vp]),
        "akp_merkle_build_te": (i32, [vp, vp, u8p, sz, sz, u64p, u64p, u64p]),
        "akp_merkle_build_te_dev": (i32, [vp, vp, u8p, sz, sz, u64p, u64p, vp]),
        "akp_merkle_build_poseidon_ragged_dev": (i32, [vp, vp, u64p, u64p, sz, u64p, u64p, vp]),
        "akp_merkle_build_te_ragged_dev": (i32, [vp, vp, u8p, u64p, sz, sz, u64p, u64p, vp]),
        "akp_merkle_gather_paths": (i32, [u64p, u64p, sz, u32, u64p, sz, u64p, u64p]),
        "akp_merkle_gather_paths_dev": (i32, [vp, u64p, u64p, sz, u32, u64p, sz, u64p, u64p, vp]),
        "akp_merkle_verify_paths_poseidon": (i32, [vp, vp, u64p, u64p, sz, sz, u64p, u64p, u64p, sz, u8p]),
        "akp_merkle_verify_paths_poseidon_dev": (i32, [vp, vp, u64p, u64p, sz, sz, u64p, u64p, u64p, sz, u8p, vp]),
        "akp_merkle_verify_paths_te": (i32, [vp, vp, u64p, u8p, sz, sz, u64p, u64p, u64p, sz, u8p]),
        "akp_poseidon_kernel_for": (C.c_char_p, [vp, sz, i32]),
        "akp_host_alloc": (i32, [sz, pp]),
        "akp_host_free": (i32, [vp]),
        "akp_host_register": (i32, [vp, sz]),
        "akp_host_unregister": (i32, [vp]),
        "akp_merkle_tree_build_poseidon": (i32, [vp, vp, u64p, sz, sz, pp]),
        "akp_merkle_tree_build_te": (i32, [vp, vp, u8p, sz, sz, pp]),
        "akp_merkle_tree_from_digests_poseidon": (i32, [vp, vp, u64p, sz, pp]),
        "akp_merkle_tree_from_digests_te": (i32, [vp, vp, u64p, sz, pp]),
        "akp_merkle_tree_destroy": (None, [vp]),
        "akp_merkle_tree_info": (i32, [vp, C.POINTER(sz), C.POINTER(u32), C.POINTER(sz)]),
        "akp_merkle_tree_root": (i32, [vp, u64p]),
        "akp_merkle_tree_export": (i32, [vp, u64p, u64p]),
        "akp_merkle_tree_device_ptrs": (i32, [vp, pp, pp]),
        "akp_merkle_tree_gather_paths": (i32, [vp, u64p, sz, u64p, u64p]),
        "akp_merkle_tree_multi_proof": (i32, [vp, u64p, sz, u64p, u64p, u64p, sz, C.POINTER(sz)]),
        "akp_merkle_tree_update_batch": (i32, [vp, u64p, vp, sz, sz]),
        "akp_merkle_tree_check_update": (i32, [vp, u64, vp, sz, u64p, C.POINTER(i32)]),
        "akp_merkle_multipath_encode": (i32, [u64p, sz, sz, u32, u64p, u64p, C.POINTER(sz)]),
        "akp_merkle_multipath_decode": (i32, [u64p, u64p, sz, sz, sz, u32, u64p]),
        "akp_merkle_verify_multipath_poseidon": (i32, [vp, vp, u64p, u64p, sz, sz, u64p, u64p, u64p, u64p, sz, sz, C.POINTER(i32)]),
        "akp_merkle_verify_multipath_te": (i32, [vp, vp, u64p, u8p, sz, sz, u64p, u64p, u64p, u64p, sz, sz, C.POINTER(i32)]),
        "akp_merkle_tree_build_poseidon_dev": (i32, [vp, vp, u64p, sz, sz, pp]),
        "akp_merkle_tree_build_te_dev": (i32, [vp, vp, u8p, sz, sz, pp]),
        "akp_multi_tree_build_poseidon": (i32, [vp, pp, pp, u64p, sz, sz, pp]),
        "akp_multi_tree_build_te": (i32, [vp, pp, pp, u8p, sz, sz, pp]),
        "akp_multi_tree_build_poseidon_dev": (i32, [vp, pp, pp, pp, sz, sz, pp]),
        "akp_multi_tree_build_te_dev": (i32, [vp, pp, pp, pp, sz, sz, pp]),
        "akp_multi_tree_destroy": (None, [vp]),
        "akp_multi_tree_info": (i32, [vp, C.POINTER(sz), C.POINTER(u32), C.POINTER(sz), C.POINTER(i32)]),
        "akp_multi_tree_root": (i32, [vp, u64p]),
        "akp_multi_tree_shard": (vp, [vp, i32]),
        "akp_multi_tree_gather_paths": (i32, [vp, u64p, sz, u64p, u64p]),
        "akp_multi_tree_update_batch": (i32, [vp, u64p, vp, sz, sz]),
        "akp_multi_tree_export": (i32, [vp, u64p, u64p]),
        "akp_multi_tree_check_update": (i32, [vp, u64, vp, sz, u64p, C.POINTER(i32)]),
        "akp_multi_tree_from_digests_poseidon": (i32, [vp, pp, pp, u64p, sz, pp]),
        "akp_multi_tree_from_digests_te": (i32, [vp, pp, pp, u64p, sz, pp]),
        "akp_serialize_digests": (i32, [u64p, sz, u32, i32, u8p, sz, C.POINTER(sz)]),
        "akp_deserialize_digests": (i32, [u8p, sz, sz, u32, i32, i32, u64p]),
        "akp_serialize_poseidon_config": (i32, [vp, u8p, sz, C.POINTER(sz)]),
        "akp_deserialize_poseidon_config": (i32, [vp, u8p, sz, pp]),
        "akp_serialize_te_parameters": (i32, [u64p, u32, u32, i32, u8p, sz, C.POINTER(sz)]),
        "akp_deserialize_te_parameters": (i32, [u8p, sz, i32, i32, u64p, sz, C.POINTER(u32), C.POINTER(u32)]),
        "akp_serialize_path": (i32, [u64p, u64p, sz, u64, u32, i32, u8p, sz, C.POINTER(sz)]),
        "akp_deserialize_path": (i32, [u8p, sz, u32, i32, i32, u64p, u64p, sz, C.POINTER(sz), C.POINTER(u64)]),
        "akp_serialize_multipath": (i32, [u64p, u64p, u64p, u64p, u64p, sz, sz, u32, i32, u8p, sz, C.POINTER(sz)]),
        "akp_deserialize_multipath": (i32, [u8p, sz, u32, i32, i32, C.POINTER(sz), C.POINTER(sz), u64p, u64p, u64p, u64p, u64p, sz, sz]),
        "akp_multi_create": (i32, [C.POINTER(i32), i32, pp]),
        "akp_multi_destroy": (None, [vp]),
        "akp_multi_size": (i32, [vp]),
        "akp_multi_ctx": (vp, [vp, i32]),
        "akp_multi_last_phases": (i32, [vp, C.POINTER(C.c_double)]),
        "akp_merkle_build_sharded_poseidon": (i32, [vp, pp, pp, u64p, sz, sz, u64p, u64p, u64p]),
        "akp_merkle_build_sharded_te": (i32, [vp, pp, pp, u8p, sz, sz, u64p, u64p, u64p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)  # AttributeError here == header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    return L, sorted(sig)


lib, DECLARED_SYMBOLS = _load()


def check(rc):
    if rc == AKP_OK:
        return
    msg = lib.akp_last_error().decode("utf-8", "replace")
    if rc == AKP_ERR_BAD_LENGTH:
        raise IncorrectInputLength(rc, msg)
    if rc == AKP_ERR_NOT_POW2:
        raise NotPowerOfTwo(rc, msg)
    raise AkpError(rc, msg)


TABLE_BUDGET_DEVICE = (1 << (8 * C.sizeof(C.c_size_t))) - 1  # AKP_TABLE_BUDGET_DEVICE


class Context:
    """akp_ctx wrapper (one per device)."""

    def __init__(self, device_id=0):
        h = C.c_void_p()
        check(lib.akp_ctx_create(device_id, C.byref(h)))
        self.h = h
        self.device_id = device_id
        self.table_budget_setting = 0  # what set_table_budget was last given (0: the library default)

    def synchronize(self):
        check(lib.akp_ctx_synchronize(self.h))

    def set_table_budget(self, nbytes):
        """HBM one precomputed Pedersen / Bowe-Hopwood table may take on this device (akp_ctx_set_table_budget); 0 = the default
        (320 MiB: cache-sized tables, built in milliseconds); TABLE_BUDGET_DEVICE = a quarter of the device's memory, at most half
        of what is free (HBM-sized tables: ~0.1 s to build, -23 % per hash afterwards)"""
        check(lib.akp_ctx_set_table_budget(self.h, int(nbytes)))
        self.table_budget_setting = int(nbytes)

    def table_budget(self):
        return int(lib.akp_ctx_table_budget(self.h))

    def close(self):
        if getattr(self, "h", None):
            lib.akp_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        # handles that depend on the context hold a reference to it, so it is destroyed last
        try:
            self.close()
        except Exception:
            pass


_default_ctx = {}


def default_context(device_id=None):
    """process-wide context for `device_id` (default: LOCAL_RANK or 0)."""
    if device_id is None:
        device_id = int(os.environ.get("AKP_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        if lib.akp_device_count() == 1:
            device_id = 0
    if device_id not in _default_ctx:
        _default_ctx[device_id] = Context(device_id)
    return _default_ctx[device_id]
