"""Oracle restatement of the reference Pedersen commitment (test infrastructure).
Follows commitment/pedersen/mod.rs:62-105: CRH of the zero-padded input, plus randomness_generator[i] for every set
bit i (little-endian) of the scalar r.  PARITY UNPINNED at value level (see oracle/__init__.py)."""
from . import jubjub as jj, pedersen as pd


def commit(generators, randomness_generator, window_size, num_windows, data: bytes, r: int):
    if len(data) > window_size * num_windows:  # :70-72
        raise pd.InputLengthPanic(len(data))
    h = pd.evaluate(generators, window_size, num_windows, data)  # pads and checks the bit length itself
    pts = [h]
    for i, power in enumerate(randomness_generator):  # BitIteratorLE(r).zip(randomness_generator) :92-99
        if (r >> i) & 1:
            pts.append(power)
    return jj.sum_points(pts)
