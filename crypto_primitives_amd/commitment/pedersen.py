"""commitment::pedersen::{Parameters, Randomness, Commitment} (commitment/pedersen/mod.rs) over Jubjub.

commit(params, input, r) = pedersen::CRH(generators, pad(input)) + sum_i bit_i(r) * randomness_generator[i]
(:62-105).  Both terms are subset sums over fixed bases, so the commitment is ONE evaluation of the GPU table
kernel over the flat generator list  generators || randomness_generator  with the message  pad(input) || r (LE):
message bit g selects flat generator g (see csrc/te_kernels.hpp).  Four identity points pad the randomness
generators from 252 (= MODULUS_BIT_SIZE of the Jubjub scalar field, :53) to 256 so that the 32-byte encoding of r
fits; r < 2^252 so those bits are always zero, exactly like the reference's zip() truncation (:92-99).
"""
import numpy as np

from .. import field
from ..crh import pedersen as _ped

SCALAR_MODULUS = 6554484396890773809930967563523245729705921265872317281365359162392183254199  # Jubjub Fr
SCALAR_BITS = 252


class Parameters:
    """commitment::pedersen::Parameters { randomness_generator: Vec<C>, generators: Vec<Vec<C>> } (:17-21);
    points are wire-format affine arrays: randomness_generator [252, 2, 4], generators [N, W, 2, 4]."""

    def __init__(self, randomness_generator, generators):
        self.randomness_generator = np.ascontiguousarray(randomness_generator, dtype=np.uint64).reshape(-1, 2, 4)
        self.generators = np.ascontiguousarray(generators, dtype=np.uint64)
        assert self.generators.ndim == 4
        self.num_windows, self.window_size = self.generators.shape[0], self.generators.shape[1]
        self._flat = None

    def flat(self):
        if self._flat is None:
            bits = self.window_size * self.num_windows
            if bits % 8:
                raise NotImplementedError("WINDOW_SIZE * NUM_WINDOWS must be a multiple of 8 (the reference pads to (W*N)/8 bytes, :71-75)")
            ident = field.fr([0, 1]).reshape(1, 2, 4)
            rg = self.randomness_generator[:256]
            pad = np.repeat(ident, 256 - rg.shape[0], axis=0)
            allg = np.concatenate([self.generators.reshape(-1, 2, 4), rg, pad], axis=0)
            self._flat = _ped.Parameters(allg.reshape(-1, 1, 2, 4))  # WINDOW_SIZE 1: a flat list of bases
        return self._flat


class Commitment:
    """CommitmentScheme for Pedersen (commitment/mod.rs:15-27): Output = affine point [2, 4]."""

    @staticmethod
    def setup(window, seed=0):
        """:44-60 -- 252 doubling powers of one base for the randomness, plus the CRH generators; bases from the
        seeded procedure of params.py (the reference's rng stream is not reproducible)."""
        from ..params import setup_pedersen_generators
        rg = setup_pedersen_generators(seed ^ 0x5EED, SCALAR_BITS, 1).reshape(SCALAR_BITS, 2, 4)
        return Parameters(rg, setup_pedersen_generators(seed, window.WINDOW_SIZE, window.NUM_WINDOWS))

    @staticmethod
    def commit(parameters: Parameters, input_: bytes, randomness: int):
        return Commitment.commit_batch(parameters, [bytes(input_)], [randomness])[0]

    @staticmethod
    def commit_batch(parameters: Parameters, inputs, randomness):
        """inputs: equal-length byte strings; randomness: python ints (Randomness<C>(ScalarField))."""
        return _ped.CRH.evaluate_batch(parameters.flat(), Commitment.encode_batch(parameters, inputs, randomness))

    @staticmethod
    def encode_batch(parameters: Parameters, inputs, randomness):
        """the messages pad(input) || r (little-endian) the table kernel is evaluated on, with the reference's length checks"""
        m, n, L = _ped._as_msgs(inputs)
        bits = parameters.window_size * parameters.num_windows
        if L > bits:  # :70-72 (the reference compares the BYTE length with W*N here)
            raise _ped.IncorrectInputLength(1, f"incorrect input length: {L}")
        if L * 8 > bits:  # the inner CRH::evaluate panics (crh/pedersen/mod.rs:82-89)
            raise _ped.IncorrectInputLength(1, f"incorrect input length {L} for window params {parameters.window_size}x{parameters.num_windows}")
        padded = bits // 8
        buf = np.zeros((n, padded + 32), dtype=np.uint8)
        buf[:, :L] = m.reshape(n, L)
        for i, r in enumerate(randomness):
            r = int(r)
            assert 0 <= r < SCALAR_MODULUS
            buf[i, padded:] = np.frombuffer(r.to_bytes(32, "little"), dtype=np.uint8)
        return buf
