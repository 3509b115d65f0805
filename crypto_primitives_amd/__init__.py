"""crypto_primitives_amd -- MI355X-native (gfx950) implementation of the native CRH / sponge /
Merkle-tree hot path of ark-crypto-primitives, behind the reference's operator surface.

The compute lives in csrc/ (hand-written HIP kernels + the C ABI of include/akp.h); this package
is the host-side mirror of the reference interface for that path.  Importing it requires the
built library (lib/libakp.so); nothing here falls back to the CPU.
"""
from ._lib import lib, AkpError, IncorrectInputLength, NotPowerOfTwo, Context, default_context, LIB_PATH  # noqa: F401
from . import field, params, sponge, crh, merkle_tree, commitment, serialize  # noqa: F401
from .sponge import PoseidonConfig, PoseidonSponge, get_default_poseidon_parameters  # noqa: F401
from .merkle_tree import MerkleTree, GpuMerkleTree, MultiGpu, ShardedMerkleTree, Path, MultiPath, PoseidonFieldConfig, PedersenByteConfig, BoweHopwoodByteConfig, PedersenXByteConfig  # noqa: F401

__all__ = ["field", "params", "sponge", "crh", "merkle_tree", "PoseidonConfig", "PoseidonSponge",
           "get_default_poseidon_parameters", "MerkleTree", "Path", "MultiPath", "Context", "default_context"]
