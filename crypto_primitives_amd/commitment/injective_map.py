"""commitment::injective_map::PedersenCommCompressor with TECompressor (commitment/injective_map/mod.rs:12-45) over Jubjub:
the Pedersen commitment followed by the injective map (x, y) -> x.  Output = Fq (wire-format [4]).

Like the commitment itself (commitment/pedersen.py) this is ONE evaluation of the table kernels over the flat base list
generators || randomness_generator -- here on a handle of the x-only kind (AKP_TE_PEDERSEN_X), so the y coordinate is never
computed.
"""
import numpy as np

from ..crh import injective_map as _inj
from . import pedersen as _cp


class PedersenCommCompressor:
    @staticmethod
    def setup(window, seed=0):
        """:25-30 delegates to pedersen::Commitment::setup"""
        return _cp.Commitment.setup(window, seed)

    @staticmethod
    def _flat_x(parameters: _cp.Parameters):
        if getattr(parameters, "_flat_x", None) is None:
            parameters._flat_x = _inj.Parameters(parameters.flat().generators)
        return parameters._flat_x

    @staticmethod
    def commit(parameters: _cp.Parameters, input_: bytes, randomness: int):
        return PedersenCommCompressor.commit_batch(parameters, [bytes(input_)], [randomness])[0]

    @staticmethod
    def commit_batch(parameters: _cp.Parameters, inputs, randomness):
        """:32-44: injective_map(pedersen::Commitment::commit(parameters, input, randomness))"""
        buf = _cp.Commitment.encode_batch(parameters, inputs, randomness)
        return _inj.PedersenCRHCompressor.evaluate_batch(PedersenCommCompressor._flat_x(parameters), buf)
