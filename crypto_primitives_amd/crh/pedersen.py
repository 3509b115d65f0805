"""pedersen::{Window, Parameters, CRH, TwoToOneCRH} over Jubjub (crh/pedersen/mod.rs) on the GPU.

Digests are affine points: wire-format arrays [..., 2, 4] (x, y).  Messages are bytes / uint8 arrays.
"""
import ctypes as C

import numpy as np

from .._lib import lib, check, default_context, TE_PEDERSEN, TE_BOWE_HOPWOOD, IncorrectInputLength  # noqa: F401


class Window:
    """pedersen::Window (crh/pedersen/mod.rs:23-26): subclass and set the two constants."""
    WINDOW_SIZE = 0
    NUM_WINDOWS = 0


class Parameters:
    """pedersen::Parameters<C> / bowe_hopwood::Parameters<P> { generators } (pedersen/mod.rs:28-31).
    generators: wire-format affine points [NUM_WINDOWS, WINDOW_SIZE, 2, 4], used verbatim.
    table_shape: digit width (Pedersen, 2..24) / chunks per table step (Bowe-Hopwood, 1..8) of the device tables built for
    these generators; 0 = the widest the context's table budget admits (akp_te_params_create_shaped; by default the cache-sized tables).  A tuning choice: the
    digests do not depend on it."""

    _KIND = TE_PEDERSEN

    def __init__(self, generators, window_size=None, num_windows=None, table_shape=0):
        g = np.ascontiguousarray(generators, dtype=np.uint64)
        if g.ndim == 4:
            num_windows, window_size = g.shape[0], g.shape[1]
        self.generators = g.reshape(num_windows, window_size, 2, 4)
        self.window_size, self.num_windows = window_size, num_windows
        self.table_shape = int(table_shape)
        self._handles = {}

    def handle(self, ctx=None, kind=None, shape=None):
        """device tables of these generators for one kernel kind (default: the kind of this Parameters class).  In the
        reference pedersen::Parameters serve both pedersen::CRH and the TECompressor types (injective_map/mod.rs:45,77), so
        the CRH class -- not the parameter object -- decides the digest width: callers go through `te_handle`."""
        ctx = ctx or default_context()
        kind = self._KIND if kind is None else kind
        shape = self.table_shape if shape is None else shape
        key = (id(ctx), kind, int(shape))
        if key not in self._handles:
            h = C.c_void_p()
            check(lib.akp_te_params_create_shaped(ctx.h, kind, self.window_size, self.num_windows,
                                                  self.generators.ctypes.data, int(shape), C.byref(h)))
            self._handles[key] = _TeHandle(h, ctx, kind)
        return self._handles[key]


def te_handle(parameters, crh_cls, ctx=None):
    """the handle a CRH class must use with `parameters`: the kernel kind (and with it the number of Fr the library writes
    per digest) is the CLASS's, so the output buffers sized from `crh_cls._FE` always match what libakp writes.  Pedersen
    parameters may serve either Pedersen flavour (x||y or x only); mixing Pedersen and Bowe-Hopwood parameter types is the
    type error the reference's generics rule out."""
    kind = crh_cls._KIND
    if (kind == TE_BOWE_HOPWOOD) != (parameters._KIND == TE_BOWE_HOPWOOD):
        raise TypeError("%s cannot be evaluated with %s.%s" % (crh_cls.__name__, type(parameters).__module__, type(parameters).__name__))
    h = parameters.handle(ctx, kind=kind)
    assert h.fe_per_digest == crh_cls._FE, "digest width of the handle and of the CRH class disagree"
    return h


class _TeHandle:
    def __init__(self, h, ctx, kind=TE_PEDERSEN):
        self.h, self.ctx, self.kind = h, ctx, kind
        self.fe_per_digest = 2 if kind == TE_PEDERSEN else 1  # what libakp writes per digest (te_fe_per_digest, capi)

    def info(self, msg_len=0):
        """tuning facts of the device tables (akp_te_params_info): digit width / chunks per step, signed-subset flag,
        table bytes, table steps of a msg_len-byte input"""
        d, sg, tb, st = C.c_uint32(), C.c_int32(), C.c_size_t(), C.c_uint32()
        check(lib.akp_te_params_info(self.h, C.byref(d), C.byref(sg), C.byref(tb), msg_len, C.byref(st)))
        return {"digit_bits_or_group": d.value, "signed_subset": bool(sg.value), "table_bytes": tb.value, "steps": st.value}

    def prepare(self, msg_len=None, compress=False):
        """build the device tables now (akp_te_params_prepare / _prepare_compress) instead of inside the first hash"""
        if compress:  # first: the two-to-one buffer is usually the longer message, and a longer message later would rebuild the table
            check(lib.akp_te_params_prepare_compress(self.h))
        if msg_len is not None:
            check(lib.akp_te_params_prepare(self.h, int(msg_len)))

    def table_info(self):
        """the shared table behind this handle (akp_te_params_table_info): id (equal for handles that share), handles attached,
        wide-table builds so far, and the phases of the last build (`last_build`: allocation / part tables / combine kernel /
        constants in ms, whether it ran in the background, `upgrade_state`)"""
        from .._lib import TeBuildReport
        tid, refs, builds, rep = C.c_uint64(), C.c_uint32(), C.c_uint64(), TeBuildReport()
        check(lib.akp_te_params_table_info(self.h, C.byref(tid), C.byref(refs), C.byref(builds), C.byref(rep)))
        return {"table_id": tid.value, "handles_attached": refs.value, "wide_builds": builds.value, "last_build": rep.as_dict()}

    def wait_for_wide_table(self, msg_len=None, compress=False, timeout_s=30.0):
        """a handle created under a table budget above the default starts on the cache-sized table while the wide one is built in
        the background: poll (without blocking the builder) until calls for this length use the wide table; returns seconds waited,
        or None if the handle has a single table / the wide one could not be built"""
        import time
        t0 = time.perf_counter()
        while True:
            st = self.table_info()["last_build"]["upgrade_state"]
            if st in (0, 3):
                return None
            if st == 2:
                self.prepare(msg_len, compress)  # (complete for another length, perhaps: this makes it so for the one asked for)
                return time.perf_counter() - t0
            if time.perf_counter() - t0 > timeout_s:
                raise TimeoutError("the wide table was not ready after %.0f s" % timeout_s)
            time.sleep(0.002)

    def __del__(self):
        try:
            if self.h:
                lib.akp_te_params_destroy(self.h)
                self.h = None
        except Exception:
            pass


def ragged_bytes(msgs):
    """byte strings of DIFFERENT lengths -> (flat uint8 array, offsets uint64 [n + 1]) for the `_ragged` entry points, or None when
    `msgs` is an array / a list of equal-length strings (the uniform entry points take those)"""
    if not isinstance(msgs, (list, tuple)) or not msgs or isinstance(msgs[0], (int, np.integer)):
        return None
    lens = [len(x) for x in msgs]
    if all(L == lens[0] for L in lens):
        return None
    offs = np.zeros(len(msgs) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(np.asarray(lens, dtype=np.uint64))
    flat = np.frombuffer(b"".join(bytes(np.asarray(x, dtype=np.uint8)) if not isinstance(x, (bytes, bytearray)) else bytes(x) for x in msgs), dtype=np.uint8)
    return np.ascontiguousarray(flat), offs


def _as_msgs(msgs, n=None, msg_len=None):
    if isinstance(msgs, (bytes, bytearray)):
        m = np.frombuffer(bytes(msgs), dtype=np.uint8)
        return m, 1, len(m)
    if isinstance(msgs, (list, tuple)) and msgs and isinstance(msgs[0], (bytes, bytearray)):
        L = len(msgs[0])
        assert all(len(x) == L for x in msgs), "the uniform batch forms take equal-length inputs (evaluate_batch routes others to the ragged entry point)"
        return np.frombuffer(b"".join(bytes(x) for x in msgs), dtype=np.uint8), len(msgs), L
    m = np.ascontiguousarray(msgs, dtype=np.uint8)
    if m.ndim == 1:
        return m, 1, m.shape[0]
    return m, m.shape[0], m.shape[1]


class _TeCRH:
    _FE = 2  # field elements per digest
    _KIND = TE_PEDERSEN  # kernel kind (AKP_TE_*): fixes the digest width on the library side

    @classmethod
    def _check_window(cls, W, parameters):
        if W is not None:
            # crh/pedersen/mod.rs:101-109 assert_eq!(parameters.generators.len(), W::NUM_WINDOWS)
            assert parameters.num_windows == W.NUM_WINDOWS and parameters.window_size == W.WINDOW_SIZE, \
                "Incorrect pp size for window params"

    @classmethod
    def evaluate(cls, parameters, input_, window=None):
        """CRH::evaluate (crh/pedersen/mod.rs:76-129 / crh/bowe_hopwood/mod.rs:114-186) for one byte string."""
        return cls.evaluate_batch(parameters, [bytes(input_)], window)[0]

    @classmethod
    def evaluate_batch(cls, parameters, msgs, window=None):
        """n inputs -> n digests.  Inputs of different lengths (a list of byte strings) are hashed each with ITS length, as the
        reference's per-item evaluate does (akp_te_crh_batch_ragged)."""
        cls._check_window(window, parameters)
        rg = ragged_bytes(msgs)
        if rg is not None:
            flat, offs = rg
            n = len(offs) - 1
            out = np.empty((n, cls._FE, 4), dtype=np.uint64)
            check(lib.akp_te_crh_batch_ragged(te_handle(parameters, cls).h, flat.ctypes.data if flat.size else None, offs.ctypes.data, n, out.ctypes.data))
            return out if cls._FE == 2 else out.reshape(n, 4)
        m, n, L = _as_msgs(msgs)
        out = np.empty((n, cls._FE, 4), dtype=np.uint64)
        check(lib.akp_te_crh_batch(te_handle(parameters, cls).h, m.ctypes.data if m.size else None, n, L, out.ctypes.data))
        return out if cls._FE == 2 else out.reshape(n, 4)


class CRH(_TeCRH):
    """pedersen::CRH<JubJub, W>: Input = [u8], Output = affine point."""

    @staticmethod
    def setup(window, seed=0):
        """CRHScheme::setup (crh/pedersen/mod.rs:64-74).  The reference draws `C::rand(rng)` bases and
        their doublings (:40-56); its rng stream is not reproducible outside ark-std, so the bases here
        come from a seeded procedure of our own (host big-int, see params.py)."""
        from ..params import setup_pedersen_generators
        return Parameters(setup_pedersen_generators(seed, window.WINDOW_SIZE, window.NUM_WINDOWS))


class TwoToOneCRH:
    """pedersen::TwoToOneCRH<JubJub, W> (crh/pedersen/mod.rs:149-198)."""
    _crh = CRH

    @classmethod
    def setup(cls, window, seed=0):
        return cls._crh.setup(window, seed)

    @classmethod
    def evaluate(cls, parameters, left_input, right_input):
        return cls.evaluate_batch(parameters, [bytes(left_input)], [bytes(right_input)])[0]

    @classmethod
    def evaluate_batch(cls, parameters, left, right):
        """:158-182: left/right equal-length byte strings -> buffer of W*N/8 bytes -> CRH::evaluate."""
        l, n, L = _as_msgs(left)
        r, n2, L2 = _as_msgs(right)
        assert (n, L) == (n2, L2), "left and right input should be of equal length"
        fe = cls._crh._FE
        out = np.empty((n, fe, 4), dtype=np.uint64)
        check(lib.akp_te_two_to_one_batch(te_handle(parameters, cls._crh).h, l.ctypes.data if l.size else None,
                                          r.ctypes.data if r.size else None, n, L, out.ctypes.data))
        return out if fe == 2 else out.reshape(n, 4)

    @classmethod
    def compress(cls, parameters, left_input, right_input):
        fe = cls._crh._FE
        l = np.ascontiguousarray(left_input, dtype=np.uint64).reshape(1, fe, 4)
        r = np.ascontiguousarray(right_input, dtype=np.uint64).reshape(1, fe, 4)
        return cls.compress_batch(parameters, l, r)[0]

    @classmethod
    def compress_batch(cls, parameters, left, right):
        """:187-197 / bowe_hopwood :229-239: digests are serialised uncompressed first (on the GPU)."""
        fe = cls._crh._FE
        l = np.ascontiguousarray(left, dtype=np.uint64).reshape(-1, fe, 4)
        r = np.ascontiguousarray(right, dtype=np.uint64).reshape(-1, fe, 4)
        out = np.empty_like(l)
        check(lib.akp_te_compress_batch(te_handle(parameters, cls._crh).h, l.ctypes.data, r.ctypes.data, l.shape[0], out.ctypes.data))
        return out if fe == 2 else out.reshape(-1, 4)
