"""commitment layer (reference: crypto-primitives/src/commitment/): Pedersen commitment on the GPU table kernel."""
from . import pedersen  # noqa: F401
