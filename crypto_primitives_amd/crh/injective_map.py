"""injective_map::{TECompressor, PedersenCRHCompressor, PedersenTwoToOneCRHCompressor} over Jubjub
(crh/injective_map/mod.rs:16-108) on the GPU: the Pedersen hash followed by the injective map (x, y) -> x.

Digest = x coordinate (Fq = BLS12-381 Fr): wire-format arrays [..., 4].  These are the leaf / two-to-one hashes of the
reference's R1CS Merkle-tree tests (merkle_tree/tests/constraints.rs); here they run as the third kind of the curve-hash
kernels (AKP_TE_PEDERSEN_X): the Pedersen tables and accumulate kernels, the x-only finalisation of Bowe-Hopwood.
`compress` serialises the two x coordinates (64 bytes) into the (W*N)/8-byte buffer; the zero padding behind them selects
nothing, so a 4x256 inner node takes half the table steps of a full-width message.
"""
from .._lib import TE_PEDERSEN_X
from . import pedersen as _ped


class Parameters(_ped.Parameters):
    """pedersen::Parameters<C> (the compressor types reuse them, injective_map/mod.rs:45,77), bound to the x-only kernels"""
    _KIND = TE_PEDERSEN_X


class TECompressor:
    """TECompressor::injective_map (:24-31): affine point -> x.  Host helper for digests already on the host."""

    @staticmethod
    def injective_map(point):
        import numpy as np
        return np.ascontiguousarray(point, dtype=np.uint64).reshape(-1, 2, 4)[:, 0].copy()


class PedersenCRHCompressor(_ped._TeCRH):
    """PedersenCRHCompressor<JubJub, TECompressor, W> (:33-62): Input = [u8], Output = Fq."""
    _FE = 1
    _KIND = TE_PEDERSEN_X

    @staticmethod
    def setup(window, seed=0):
        """:47-52 delegates to pedersen::CRH::setup"""
        from ..params import setup_pedersen_generators
        return Parameters(setup_pedersen_generators(seed, window.WINDOW_SIZE, window.NUM_WINDOWS))


class PedersenTwoToOneCRHCompressor(_ped.TwoToOneCRH):
    """PedersenTwoToOneCRHCompressor<JubJub, TECompressor, W> (:64-108): evaluate = x of pedersen::TwoToOneCRH::evaluate,
    compress = evaluate on the uncompressed serialisations of the two Fq digests"""
    _crh = PedersenCRHCompressor
