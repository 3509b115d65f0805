"""poseidon::{CRH, TwoToOneCRH} (crh/poseidon/mod.rs) on the GPU.

CRHScheme (crh/mod.rs:18-28):  setup / evaluate;  TwoToOneCRHScheme (:31-51): setup / evaluate / compress.
The per-item trait methods are kept (batch of one, still on the GPU -- no CPU fallback) and each
gains a `*_batch` form, which is what MerkleTree and the benches use.
Inputs/outputs are Fr wire-format numpy arrays (see field.py).
"""
import numpy as np

from .._lib import lib, check
from ..sponge.poseidon import PoseidonConfig


def ragged_fr(inputs):
    """a list of [k_i, 4] wire-format inputs with DIFFERENT k_i -> (flat [sum k_i, 4] array, offsets uint64 [n + 1] in elements);
    None for an array or a list of equal shapes"""
    if not isinstance(inputs, (list, tuple)) or not inputs:
        return None
    items = [np.ascontiguousarray(x, dtype=np.uint64).reshape(-1, 4) for x in inputs]
    lens = [len(x) for x in items]
    if all(k == lens[0] for k in lens):
        return None
    offs = np.zeros(len(items) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum(np.asarray(lens, dtype=np.uint64))
    return np.ascontiguousarray(np.concatenate(items, axis=0)), offs


class CRH:
    """poseidon::CRH<Fr>: Input = [Fr], Output = Fr, Parameters = PoseidonConfig<Fr>."""

    @staticmethod
    def setup(rng=None):
        # crh/poseidon/mod.rs:24-28 is `unimplemented!()`: parameters must be supplied by the caller
        raise NotImplementedError("automatic generation of parameters is not implemented in the reference either")

    @staticmethod
    def evaluate(parameters: PoseidonConfig, input_) -> np.ndarray:
        """crh/poseidon/mod.rs:30-40: one input slice [k, 4] -> digest [4]."""
        x = np.ascontiguousarray(input_, dtype=np.uint64).reshape(-1, 4)
        return CRH.evaluate_batch(parameters, x.reshape(1, -1, 4))[0]

    @staticmethod
    def evaluate_batch(parameters: PoseidonConfig, inputs) -> np.ndarray:
        """inputs [n, k, 4] -> digests [n, 4].  k may be 0.  A LIST of inputs with different element counts is hashed item by item
        with its own length (akp_poseidon_crh_batch_ragged), as the reference's evaluate(&[F]) does."""
        rg = ragged_fr(inputs)
        if rg is not None:
            flat, offs = rg
            n = len(offs) - 1
            out = np.empty((n, 4), dtype=np.uint64)
            check(lib.akp_poseidon_crh_batch_ragged(parameters.handle().h, flat.ctypes.data if flat.size else None, offs.ctypes.data, n, out.ctypes.data))
            return out
        x = np.ascontiguousarray(inputs, dtype=np.uint64)
        n, k = x.shape[0], x.shape[1]
        out = np.empty((n, 4), dtype=np.uint64)
        h = parameters.handle()
        check(lib.akp_poseidon_crh_batch(h.h, x.ctypes.data if x.size else None, n, k, out.ctypes.data))
        return out


class TwoToOneCRH:
    """poseidon::TwoToOneCRH<Fr>: Input = Output = Fr."""

    setup = CRH.setup

    @staticmethod
    def compress(parameters: PoseidonConfig, left_input, right_input) -> np.ndarray:
        """crh/poseidon/mod.rs:66-79."""
        l = np.ascontiguousarray(left_input, dtype=np.uint64).reshape(1, 4)
        r = np.ascontiguousarray(right_input, dtype=np.uint64).reshape(1, 4)
        return TwoToOneCRH.compress_batch(parameters, l, r)[0]

    evaluate = compress  # :58-64

    @staticmethod
    def compress_batch(parameters: PoseidonConfig, left, right) -> np.ndarray:
        l = np.ascontiguousarray(left, dtype=np.uint64).reshape(-1, 4)
        r = np.ascontiguousarray(right, dtype=np.uint64).reshape(-1, 4)
        assert l.shape == r.shape
        out = np.empty_like(l)
        check(lib.akp_poseidon_two_to_one_batch(parameters.handle().h, l.ctypes.data, r.ctypes.data, l.shape[0], out.ctypes.data))
        return out

    evaluate_batch = compress_batch
