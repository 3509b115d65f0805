"""Host mirror of the reference Poseidon sponge over BLS12-381 Fr, backed by the GPU library.

reference                              here
sponge/poseidon/mod.rs:27-45           PoseidonConfig (ark / mds in Fr wire format)
sponge/poseidon/traits.rs:148-155      get_default_poseidon_parameters(rate, optimized_for_weights)
sponge/poseidon/mod.rs:54-63,220-345   PoseidonSponge: CryptographicSponge + FieldBasedCryptographicSponge
                                       for a BATCH of independent sponges following one schedule
sponge/mod.rs:184-191                  into_state / from_state (SpongeExt)
Field elements cross this API as numpy uint64 [..., 4] wire-format arrays (field.fr / field.to_ints
convert python ints).  Every permutation runs on the GPU; there is no CPU path.
"""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from .. import field
from .._lib import lib, check, default_context, Context


class DuplexSpongeMode:
    """sponge/mod.rs:195-206."""
    ABSORBING = 0
    SQUEEZING = 1


@dataclass
class PoseidonConfig:
    """PoseidonConfig<Fr> (sponge/poseidon/mod.rs:27-45).  `ark`: [full+partial, t, 4], `mds`: [t, t, 4]."""
    full_rounds: int
    partial_rounds: int
    alpha: int
    ark: np.ndarray
    mds: np.ndarray
    rate: int
    capacity: int

    def __post_init__(self):
        t = self.rate + self.capacity
        self.ark = np.ascontiguousarray(self.ark, dtype=np.uint64).reshape(-1, t, 4)
        self.mds = np.ascontiguousarray(self.mds, dtype=np.uint64).reshape(-1, t, 4)
        # PoseidonConfig::new asserts (:191-217)
        assert self.ark.shape[0] == self.full_rounds + self.partial_rounds
        assert self.mds.shape[0] == t
        self._handles = {}

    @property
    def t(self):
        return self.rate + self.capacity

    def handle(self, ctx: Context = None):
        """device-resident akp_poseidon for this config (created once per context)."""
        ctx = ctx or default_context()
        key = id(ctx)
        if key not in self._handles:
            h = C.c_void_p()
            check(lib.akp_poseidon_params_create(ctx.h, self.full_rounds, self.partial_rounds, self.alpha, self.rate,
                                                 self.capacity, self.ark.ctypes.data, self.mds.ctypes.data, C.byref(h)))
            self._handles[key] = _PoseidonHandle(h, ctx)
        return self._handles[key]


class _PoseidonHandle:
    def __init__(self, h, ctx):
        self.h, self.ctx = h, ctx

    def __del__(self):
        try:
            if self.h:
                lib.akp_poseidon_params_destroy(self.h)
                self.h = None
        except Exception:
            pass


def get_default_poseidon_parameters(rate: int, optimized_for_weights: bool = False):
    """PoseidonDefaultConfigField::get_default_poseidon_parameters for BLS12-381 Fr
    (traits.rs:69-155).  Returns None where the reference returns None.  Host-only (Grain LFSR)."""
    h = C.c_void_p()
    rc = lib.akp_poseidon_default_params(None, rate, 1 if optimized_for_weights else 0, C.byref(h))
    if rc != 0:
        return None
    try:
        rf, rp, r, c = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        alpha = C.c_uint64()
        check(lib.akp_poseidon_params_dims(h, C.byref(rf), C.byref(rp), C.byref(alpha), C.byref(r), C.byref(c)))
        t = r.value + c.value
        ark = np.empty((rf.value + rp.value, t, 4), dtype=np.uint64)
        mds = np.empty((t, t, 4), dtype=np.uint64)
        check(lib.akp_poseidon_params_export(h, ark.ctypes.data, mds.ctypes.data))
    finally:
        lib.akp_poseidon_params_destroy(h)
    return PoseidonConfig(rf.value, rp.value, alpha.value, ark, mds, r.value, c.value)


class PoseidonSponge:
    """`batch` PoseidonSponge<Fr> instances sharing one absorb/squeeze schedule; state on the GPU.

    new(&config) -> PoseidonSponge(config)                       (:223-234)
    absorb(&impl Absorb) for Fr / [Fr] inputs -> absorb(elems)   (:236-257; other Absorb encodings
                                                                  are host glue and out of scope)
    squeeze_native_field_elements / squeeze_field_elements::<Fr> (:309-344)
    squeeze_bytes / squeeze_bits                                  (:259-289)
    """

    def __init__(self, config: PoseidonConfig, batch: int = 1, ctx: Context = None):
        self.config = config
        self.batch = batch
        self._ph = config.handle(ctx)
        h = C.c_void_p()
        check(lib.akp_sponge_create(self._ph.h, batch, C.byref(h)))
        self.h = h

    def __del__(self):
        try:
            if getattr(self, "h", None):
                lib.akp_sponge_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def absorb(self, elems):
        """CryptographicSponge::absorb (sponge/poseidon/mod.rs:236-257).  elems: wire-format array [batch, k, 4] (or [k, 4]
        when batch == 1), or an `absorb.Absorbable` (every sponge of the batch absorbs its encoding; an empty encoding is a
        no-op, :238-240)."""
        from .absorb import Absorbable
        if isinstance(elems, Absorbable):
            vals = elems.to_sponge_field_elements()
            if not vals:
                return
            el = field.fr(vals)
            elems = np.broadcast_to(el, (self.batch,) + el.shape).copy()
        e = np.ascontiguousarray(elems, dtype=np.uint64)
        k = e.size // (4 * self.batch)
        assert e.size == self.batch * k * 4
        check(lib.akp_sponge_absorb(self.h, e.ctypes.data, k))

    @staticmethod
    def bytes_to_field_elements(data: bytes) -> np.ndarray:
        """`Absorb for &[u8]` / `Vec<u8>` (sponge/absorb.rs:124-143): the u64 LE length is prepended, then the byte
        string is cut into 31-byte ((MODULUS_BIT_SIZE - 1) / 8) little-endian chunks, one field element each
        (ark-ff's `ToConstraintField for [u8]`)."""
        b = len(data).to_bytes(8, "little") + bytes(data)
        vals = [int.from_bytes(b[i:i + 31], "little") for i in range(0, len(b), 31)]
        return field.fr(vals)

    def absorb_bytes(self, data: bytes):
        """absorb(&data) for a byte slice; every sponge of the batch absorbs the same bytes."""
        el = self.bytes_to_field_elements(data)
        self.absorb(np.broadcast_to(el, (self.batch,) + el.shape).copy())

    def absorb_all(self, *items):
        """absorb! (sponge/absorb.rs:348-356): each item absorbed on its own, in order"""
        for it in items:
            self.absorb(it)

    def clone(self):
        """`Clone` (the reference sponge is a value type)."""
        return PoseidonSponge.from_state(self.into_state(), self.config, self._ph.ctx)

    def fork(self, domain: bytes):
        """CryptographicSponge::fork (sponge/mod.rs:145-153): clone, then absorb  u64_le(len(domain)) || domain
        as a byte vector (which itself is length-prefixed by the byte-slice encoding)."""
        new = self.clone()
        new.absorb_bytes(len(domain).to_bytes(8, "little") + bytes(domain))
        return new

    def squeeze_native_field_elements(self, n: int) -> np.ndarray:
        out = np.empty((self.batch, n, 4), dtype=np.uint64)
        check(lib.akp_sponge_squeeze(self.h, out.ctypes.data, n))
        return out[0] if self.batch == 1 else out

    # ---- sized / foreign-field squeezes: host logic over squeeze_bits (sponge/mod.rs:28-100,164-179; poseidon :293-322) ----
    FULL = None  # FieldElementSize::Full; an int n stands for FieldElementSize::Truncated(n)

    @staticmethod
    def _size_num_bits(size, modulus):
        """FieldElementSize::num_bits (sponge/mod.rs:38-48)"""
        if size is None:
            return modulus.bit_length() - 1
        if size > modulus.bit_length():
            raise ValueError("num_bits is greater than the capacity of the field.")
        return int(size)

    def _squeeze_with_sizes_default(self, sizes, modulus):
        """squeeze_field_elements_with_sizes_default_impl (sponge/mod.rs:57-100): one squeeze_bits for all elements, each
        element = its bits little-endian, reduced mod the target modulus.  Returns python ints [batch][len(sizes)]."""
        if not sizes:
            return [] if self.batch == 1 else [[] for _ in range(self.batch)]
        nb = [self._size_num_bits(sz, modulus) for sz in sizes]
        bits = self.squeeze_bits(sum(nb))
        rows = [bits] if self.batch == 1 else bits
        out = []
        for row in rows:
            vals, pos = [], 0
            for n in nb:
                w = np.asarray(row[pos:pos + n], dtype=np.uint8)
                pos += n
                vals.append(int.from_bytes(np.packbits(w, bitorder="little").tobytes(), "little") % modulus)
            out.append(vals)
        return out[0] if self.batch == 1 else out

    def squeeze_native_field_elements_with_sizes(self, sizes):
        """FieldBasedCryptographicSponge::squeeze_native_field_elements_with_sizes (sponge/mod.rs:164-179): all `Full`
        -> the plain native squeeze; otherwise bits.  Wire-format array [batch, n, 4] (or [n, 4])."""
        sizes = list(sizes)
        if all(sz is None for sz in sizes):
            return self.squeeze_native_field_elements(len(sizes))
        vals = self._squeeze_with_sizes_default(sizes, field.MODULUS)
        if self.batch == 1:
            return field.fr(vals).reshape(len(sizes), 4)
        return np.stack([field.fr(v).reshape(len(sizes), 4) for v in vals])

    def squeeze_field_elements_with_sizes(self, sizes, modulus=None):
        """CryptographicSponge::squeeze_field_elements_with_sizes::<F2> (sponge/poseidon/mod.rs:293-308).  `modulus` None or
        p: F2 is the native field (wire-format array); any other prime: python ints reduced mod it."""
        if modulus is None or modulus == field.MODULUS:
            return self.squeeze_native_field_elements_with_sizes(sizes)
        return self._squeeze_with_sizes_default(list(sizes), int(modulus))

    def squeeze_field_elements(self, n: int, modulus=None):
        """squeeze_field_elements::<F2> (:310-322): the native squeeze for F2 = Fr (identity field_cast), else n `Full`-sized
        elements of the foreign field through the bit path"""
        if modulus is None or modulus == field.MODULUS:
            return self.squeeze_native_field_elements(n)
        return self.squeeze_field_elements_with_sizes([None] * n, modulus)

    def squeeze_bytes(self, num_bytes: int):
        usable = (field.MODULUS.bit_length() - 1) // 8  # :260
        n = (num_bytes + usable - 1) // usable
        el = self.squeeze_native_field_elements(n).reshape(self.batch, n, 4)
        canon = field.from_mont(el).reshape(self.batch, n, 4)
        raw = canon.view(np.uint8).reshape(self.batch, n, 32)[:, :, :usable].reshape(self.batch, n * usable)
        out = [bytes(raw[b, :num_bytes]) for b in range(self.batch)]
        return out[0] if self.batch == 1 else out

    def squeeze_bits(self, num_bits: int):
        usable = field.MODULUS.bit_length() - 1  # :276
        n = (num_bits + usable - 1) // usable
        el = self.squeeze_native_field_elements(n).reshape(self.batch, n, 4)
        canon = field.from_mont(el).reshape(self.batch, n, 4)
        raw = np.unpackbits(canon.view(np.uint8).reshape(self.batch, n, 32), axis=2, bitorder="little")[:, :, :usable]
        raw = raw.reshape(self.batch, n * usable)[:, :num_bits].astype(bool)
        out = [list(raw[b]) for b in range(self.batch)]
        return out[0] if self.batch == 1 else out

    def into_state(self):
        """SpongeExt::into_state (sponge/mod.rs:184-191) -> (state [batch, t, 4], mode, index)."""
        st = np.empty((self.batch, self.config.t, 4), dtype=np.uint64)
        mode, idx = C.c_int32(), C.c_uint32()
        check(lib.akp_sponge_get_state(self.h, st.ctypes.data, C.byref(mode), C.byref(idx)))
        return st, mode.value, idx.value

    @classmethod
    def from_state(cls, state, config: PoseidonConfig, ctx: Context = None):
        st, mode, idx = state
        st = np.ascontiguousarray(st, dtype=np.uint64)
        sp = cls(config, st.shape[0], ctx)
        check(lib.akp_sponge_set_state(sp.h, st.ctypes.data, mode, idx))
        return sp
