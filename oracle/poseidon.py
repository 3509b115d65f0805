"""Oracle restatement of the reference Poseidon sponge / CRH (test infrastructure).

Follows, in python big-int arithmetic on canonical integers:
  * Grain LFSR:        sponge/poseidon/grain_lfsr.rs:16-181
  * default params:    sponge/poseidon/traits.rs:69-146, table sponge/test.rs:13-31
  * permutation:       sponge/poseidon/mod.rs:66-121
  * duplex sponge:     sponge/poseidon/mod.rs:124-186, 223-257, 259-289, 324-344
  * CRH / TwoToOneCRH: crh/poseidon/mod.rs:30-79
Pinned by the reference KATs (tests/test_oracle_poseidon.py).
"""
from dataclasses import dataclass, field
from typing import List

from .fr import P


# ---------------------------------------------------------------- Grain LFSR
class PoseidonGrainLFSR:
    """grain_lfsr.rs:7-182."""

    def __init__(self, is_sbox_an_inverse, prime_num_bits, state_len, num_full_rounds, num_partial_rounds):
        st = [False] * 80
        st[1] = True  # b0,b1: field (grain_lfsr.rs:26)
        st[5] = bool(is_sbox_an_inverse)  # b2..b5: s-box (:29-33)

        def put(lo, hi, val):  # big-endian binary of val into st[lo..=hi] (:36-68)
            cur = val
            for i in range(hi, lo - 1, -1):
                st[i] = (cur & 1) == 1
                cur >>= 1

        put(6, 17, prime_num_bits)
        put(18, 29, state_len)
        put(30, 39, num_full_rounds)
        put(40, 49, num_partial_rounds)
        for i in range(50, 80):
            st[i] = True
        self.prime_num_bits = prime_num_bits
        self.state = st
        self.head = 0
        for _ in range(160):  # init (:177-181)
            self.update()

    def update(self):  # :163-175
        s, h = self.state, self.head
        nb = s[(h + 62) % 80] ^ s[(h + 51) % 80] ^ s[(h + 38) % 80] ^ s[(h + 23) % 80] ^ s[(h + 13) % 80] ^ s[h]
        s[h] = nb
        self.head = (h + 1) % 80
        return nb

    def get_bits(self, num_bits):  # :87-107 (shrinking: keep 2nd bit of a pair iff the 1st is 1)
        res = []
        for _ in range(num_bits):
            new_bit = self.update()
            while not new_bit:
                self.update()
                new_bit = self.update()
            res.append(self.update())
        return res

    def _int_msb_first(self):
        # :120-124 -- bits arrive most-significant first; reversed then from_bits_le
        bits = self.get_bits(self.prime_num_bits)
        v = 0
        for b in bits:
            v = (v << 1) | int(b)
        return v

    def get_field_elements_rejection_sampling(self, num_elems, p=P):  # :109-134
        res = []
        for _ in range(num_elems):
            while True:
                v = self._int_msb_first()
                if v < p:
                    res.append(v)
                    break
        return res

    def get_field_elements_mod_p(self, num_elems, p=P):  # :136-160
        return [self._int_msb_first() % p for _ in range(num_elems)]


# ------------------------------------------------------------------- params
# (rate, alpha, full_rounds, partial_rounds, skip_matrices) -- sponge/test.rs:13-31
PARAMS_OPT_FOR_CONSTRAINTS = [
    (2, 17, 8, 31, 0), (3, 5, 8, 56, 0), (4, 5, 8, 56, 0), (5, 5, 8, 57, 0),
    (6, 5, 8, 57, 0), (7, 5, 8, 57, 0), (8, 5, 8, 57, 0),
]
PARAMS_OPT_FOR_WEIGHTS = [(r, 257, 8, 13, 0) for r in range(2, 9)]


@dataclass
class PoseidonConfig:
    """sponge/poseidon/mod.rs:27-45 (values are canonical ints here)."""
    full_rounds: int
    partial_rounds: int
    alpha: int
    ark: List[List[int]]
    mds: List[List[int]]
    rate: int
    capacity: int

    def __post_init__(self):  # ctor asserts :191-217
        t = self.rate + self.capacity
        assert len(self.ark) == self.full_rounds + self.partial_rounds
        assert all(len(r) == t for r in self.ark)
        assert len(self.mds) == t and all(len(r) == t for r in self.mds)


def find_poseidon_ark_and_mds(prime_bits, rate, full_rounds, partial_rounds, skip_matrices, p=P):
    """traits.rs:105-146."""
    lfsr = PoseidonGrainLFSR(False, prime_bits, rate + 1, full_rounds, partial_rounds)
    ark = [lfsr.get_field_elements_rejection_sampling(rate + 1, p) for _ in range(full_rounds + partial_rounds)]
    for _ in range(skip_matrices):
        lfsr.get_field_elements_mod_p(2 * (rate + 1), p)
    xs = lfsr.get_field_elements_mod_p(rate + 1, p)
    ys = lfsr.get_field_elements_mod_p(rate + 1, p)
    mds = [[pow((xs[i] + ys[j]) % p, -1, p) for j in range(rate + 1)] for i in range(rate + 1)]
    return ark, mds


def get_default_poseidon_parameters(rate, optimized_for_weights=False):
    """traits.rs:69-102 for BLS12-381 Fr (255 bits)."""
    table = PARAMS_OPT_FOR_WEIGHTS if optimized_for_weights else PARAMS_OPT_FOR_CONSTRAINTS
    for (r, alpha, rf, rp, skip) in table:
        if r == rate:
            ark, mds = find_poseidon_ark_and_mds(255, rate, rf, rp, skip)
            return PoseidonConfig(rf, rp, alpha, ark, mds, rate, 1)
    return None


# -------------------------------------------------------------- permutation
def permute(cfg: PoseidonConfig, state: List[int], p=P) -> List[int]:
    """sponge/poseidon/mod.rs:98-121 (ARK -> S-box -> MDS per round)."""
    t = len(state)
    s = list(state)
    half = cfg.full_rounds // 2
    total = cfg.full_rounds + cfg.partial_rounds
    for r in range(total):
        s = [(s[i] + cfg.ark[r][i]) % p for i in range(t)]  # apply_ark :79-83
        if r < half or r >= half + cfg.partial_rounds:  # apply_s_box :66-77
            s = [pow(x, cfg.alpha, p) for x in s]
        else:
            s[0] = pow(s[0], cfg.alpha, p)
        s = [sum(s[j] * cfg.mds[i][j] for j in range(t)) % p for i in range(t)]  # apply_mds :85-96
    return s


# ------------------------------------------------------------ duplex sponge
ABSORBING, SQUEEZING = "absorbing", "squeezing"


class PoseidonSponge:
    """sponge/poseidon/mod.rs:54-63, 124-186, 223-344 (field-element absorb only)."""

    def __init__(self, cfg: PoseidonConfig, p=P):
        self.cfg = cfg
        self.p = p
        self.state = [0] * (cfg.rate + cfg.capacity)
        self.mode = (ABSORBING, 0)

    def _permute(self):
        self.state = permute(self.cfg, self.state, self.p)

    def _absorb_internal(self, idx, elems):  # :124-153
        rate, cap = self.cfg.rate, self.cfg.capacity
        rem = list(elems)
        while True:
            if idx + len(rem) <= rate:
                for i, e in enumerate(rem):
                    self.state[cap + i + idx] = (self.state[cap + i + idx] + e) % self.p
                self.mode = (ABSORBING, idx + len(rem))
                return
            n = rate - idx
            for i, e in enumerate(rem[:n]):
                self.state[cap + i + idx] = (self.state[cap + i + idx] + e) % self.p
            self._permute()
            rem = rem[n:]
            idx = 0

    def _squeeze_internal(self, idx, n_out):  # :156-186
        rate, cap = self.cfg.rate, self.cfg.capacity
        out = []
        remaining = n_out
        while True:
            if idx + remaining <= rate:
                out += self.state[cap + idx: cap + idx + remaining]
                self.mode = (SQUEEZING, idx + remaining)
                return out
            n = rate - idx
            out += self.state[cap + idx: cap + idx + n]
            remaining -= n
            if remaining != 0:
                self._permute()
            idx = 0

    def absorb(self, elems):  # :236-257
        elems = list(elems)
        if not elems:
            return
        kind, idx = self.mode
        if kind == ABSORBING:
            if idx == self.cfg.rate:
                self._permute()
                idx = 0
            self._absorb_internal(idx, elems)
        else:
            self._absorb_internal(0, elems)

    def squeeze_native_field_elements(self, n):  # :324-344
        kind, idx = self.mode
        if kind == ABSORBING:
            self._permute()
            return self._squeeze_internal(0, n)
        if idx == self.cfg.rate:
            self._permute()
            idx = 0
        return self._squeeze_internal(idx, n)

    def squeeze_bytes(self, num_bytes):  # :259-273
        usable = (self.p.bit_length() - 1) // 8
        n = (num_bytes + usable - 1) // usable
        out = bytearray()
        for e in self.squeeze_native_field_elements(n):
            out += e.to_bytes(32, "little")[:usable]
        return bytes(out[:num_bytes])

    def squeeze_bits(self, num_bits):  # :275-289
        usable = self.p.bit_length() - 1
        n = (num_bits + usable - 1) // usable
        out = []
        for e in self.squeeze_native_field_elements(n):
            out += [bool((e >> i) & 1) for i in range(usable)]
        return out[:num_bits]


# ---- sized squeezes (sponge/mod.rs:28-100, sponge/poseidon/mod.rs:293-322) ----------------------------------------
FULL = None  # FieldElementSize::Full; an int n stands for FieldElementSize::Truncated(n)


def size_num_bits(size, modulus):  # FieldElementSize::num_bits (:38-48)
    if size is FULL:
        return modulus.bit_length() - 1
    if size > modulus.bit_length():
        raise ValueError("num_bits is greater than the capacity of the field.")
    return size


def squeeze_field_elements_with_sizes_default_impl(sp, sizes, modulus):  # sponge/mod.rs:57-100
    if not sizes:
        return []
    nb = [size_num_bits(sz, modulus) for sz in sizes]
    bits = sp.squeeze_bits(sum(nb))
    out, pos = [], 0
    for n in nb:
        window = bits[pos:pos + n]
        pos += n
        by = bytearray((n + 7) // 8)
        for i, b in enumerate(window):
            if b:
                by[i // 8] |= 1 << (i % 8)
        out.append(int.from_bytes(bytes(by), "little") % modulus)  # from_le_bytes_mod_order
    return out


def squeeze_native_field_elements_with_sizes(sp, sizes):  # sponge/mod.rs:164-179
    if all(sz is FULL for sz in sizes):
        return sp.squeeze_native_field_elements(len(sizes))
    return squeeze_field_elements_with_sizes_default_impl(sp, sizes, sp.p)


def squeeze_field_elements_with_sizes(sp, sizes, modulus=None):  # sponge/poseidon/mod.rs:293-308
    if modulus is None or modulus == sp.p:  # same characteristic: the native path + identity field_cast
        return squeeze_native_field_elements_with_sizes(sp, sizes)
    return squeeze_field_elements_with_sizes_default_impl(sp, sizes, modulus)


def squeeze_field_elements(sp, n, modulus=None):  # :310-322
    if modulus is None or modulus == sp.p:
        return sp.squeeze_native_field_elements(n)
    return squeeze_field_elements_with_sizes(sp, [FULL] * n, modulus)


def bytes_to_field_elements(data: bytes, p=P):
    """sponge/absorb.rs:124-143 (`Absorb for [u8]`): u64 LE length || bytes, 31-byte LE chunks (ark-ff
    `ToConstraintField<F> for [u8]`; external crate, restated from its published behaviour -- unpinned)."""
    b = len(data).to_bytes(8, "little") + bytes(data)
    return [int.from_bytes(b[i:i + 31], "little") % p for i in range(0, len(b), 31)]


def sponge_fork(sp: "PoseidonSponge", domain: bytes) -> "PoseidonSponge":
    """sponge/mod.rs:145-153."""
    new = PoseidonSponge(sp.cfg, sp.p)
    new.state, new.mode = list(sp.state), sp.mode
    new.absorb(bytes_to_field_elements(len(domain).to_bytes(8, "little") + bytes(domain), sp.p))
    return new


# -------------------------------------------------------------------- CRHs
def crh_evaluate(cfg: PoseidonConfig, inputs: List[int], p=P) -> int:
    """crh/poseidon/mod.rs:30-40."""
    sp = PoseidonSponge(cfg, p)
    sp.absorb(inputs)
    return sp.squeeze_native_field_elements(1)[0]


def two_to_one_compress(cfg: PoseidonConfig, left: int, right: int, p=P) -> int:
    """crh/poseidon/mod.rs:66-79 (evaluate == compress, :58-64)."""
    sp = PoseidonSponge(cfg, p)
    sp.absorb([left])
    sp.absorb([right])
    return sp.squeeze_native_field_elements(1)[0]
