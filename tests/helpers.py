"""shared test helpers (oracle-side conversions, synthetic inputs)."""
import numpy as np

from oracle import fr as ofr, poseidon as opo, cref


def mont(vals):
    return ofr.ints_to_mont_array(list(vals))


def ints(arr):
    return ofr.mont_array_to_ints(arr)


def rand_fr(n, seed):
    rng = ofr.SplitMix64(seed)
    return [rng.fr() for _ in range(n)]


def rand_fr_array(n, seed):
    """fast uniform wire-format array (canonical values re-read as Montgomery residues)."""
    rng = np.random.default_rng(seed)
    out = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64, endpoint=False)
    out[:, 3] &= np.uint64(0x3FFFFFFFFFFFFFFF)  # < 2^254 < p
    return out


def oracle_cfg(rate=2, weights=False):
    return opo.get_default_poseidon_parameters(rate, weights)


def cref_poseidon(cfg):
    """oracle C handle from a python-oracle PoseidonConfig (canonical ints)."""
    ark = mont([x for r in cfg.ark for x in r])
    mds = mont([x for r in cfg.mds for x in r])
    return cref.Poseidon(cfg.full_rounds, cfg.partial_rounds, cfg.alpha, cfg.rate, cfg.capacity, ark, mds)


def gens_array(gens):
    """oracle generators (list[list[(x, y)]]) -> wire format [N, W, 2, 4]."""
    n, w = len(gens), len(gens[0])
    return mont([v for row in gens for pt in row for v in pt]).reshape(n, w, 2, 4)
