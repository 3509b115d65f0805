"""CRH layer (reference: crypto-primitives/src/crh/): CRHScheme / TwoToOneCRHScheme implementations."""
from . import poseidon, pedersen, bowe_hopwood  # noqa: F401
