"""commitment layer (reference: crypto-primitives/src/commitment/): Pedersen commitment on the GPU table kernel, and its composition with TECompressor (commitment/injective_map)."""
from . import pedersen, injective_map  # noqa: F401
