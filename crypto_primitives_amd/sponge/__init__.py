"""sponge layer (reference: crypto-primitives/src/sponge/)."""
from .poseidon import PoseidonConfig, PoseidonSponge, get_default_poseidon_parameters, DuplexSpongeMode  # noqa: F401
