"""bowe_hopwood::{Parameters, CRH, TwoToOneCRH} over Jubjub (crh/bowe_hopwood/mod.rs) on the GPU.

Digest = x coordinate (Fq = BLS12-381 Fr): wire-format arrays [..., 4].
"""
from .._lib import TE_BOWE_HOPWOOD
from . import pedersen as _ped

CHUNK_SIZE = 3  # crh/bowe_hopwood/mod.rs:31
MAX_CHUNKS_PER_SEGMENT = 63  # calculate_num_chunks_in_segment::<Jubjub Fr>() (:82-93)


class Parameters(_ped.Parameters):
    _KIND = TE_BOWE_HOPWOOD


class CRH(_ped._TeCRH):
    """bowe_hopwood::CRH<EdwardsConfig, W>: Input = [u8], Output = Fq."""
    _FE = 1
    _KIND = TE_BOWE_HOPWOOD

    @staticmethod
    def setup(window, seed=0):
        """CRHScheme::setup (:81-112) incl. the window bound check; bases from our seeded procedure,
        spaced by 2^4 per chunk (:45-59)."""
        if window.WINDOW_SIZE > MAX_CHUNKS_PER_SEGMENT:
            raise ValueError("Bowe-Hopwood-PedersenCRH hash must have a window size resulting in scalars < (p-1)/2, "
                             f"maximum segment size is {MAX_CHUNKS_PER_SEGMENT}")
        from ..params import setup_bowe_hopwood_generators
        return Parameters(setup_bowe_hopwood_generators(seed, window.WINDOW_SIZE, window.NUM_WINDOWS))


class TwoToOneCRH(_ped.TwoToOneCRH):
    """bowe_hopwood::TwoToOneCRH (:189-240)."""
    _crh = CRH
