#!/usr/bin/env python3
"""Write tests/golden/emitter_inputs.json: the INPUTS of the reference-vector emitter (shim/examples/emit_vectors.rs).

The curve half of the path has no reference-produced vectors (the reference's tests only assert native == gadget and
proof round trips, and its generators come from `C::rand(test_rng())`).  `Parameters.generators` is a public field
(crh/pedersen/mod.rs:30, crh/bowe_hopwood/mod.rs:36), so the emitter is handed generators and inputs as DATA (this file),
runs the REFERENCE's own `pedersen::CRH`, `bowe_hopwood::CRH`, both `TwoToOneCRH::{evaluate, compress}`,
`MerkleTree::{new, generate_proof, generate_multi_proof, update}` and `CanonicalSerialize` on them, and writes
tests/golden/reference_vectors.json, which tests/test_reference_vectors.py consumes (python oracle, C oracle, serialize.py on
the CPU; the GPU path with -m gpu).  Nothing here is a reference output: generators are our seeded bases (valid points of the
prime-order subgroup), messages are SplitMix64 bytes.

Run:  python tests/golden/make_emitter_inputs.py
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import poseidon as po, jubjub as jj, fr  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "emitter_inputs.json")


def _gens(g):
    return [[[str(p[0]), str(p[1])] for p in row] for row in g]


def _msgs(seed, lengths):
    rng = fr.SplitMix64(seed)
    return [rng.bytes(L).hex() if L else "" for L in lengths]


def curve_section(seed, W, N, g, lengths, half_lengths, leaf_len):
    rng = fr.SplitMix64(seed ^ 0x5EED)
    return {
        "window_size": W, "num_windows": N, "generator_seed": hex(seed), "generators": _gens(g),
        "messages": _msgs(seed, lengths),
        # TwoToOneCRH::evaluate inputs: equal-length halves (crh/pedersen/mod.rs:158-182, crh/bowe_hopwood/mod.rs:202-227)
        "pairs": [[rng.bytes(h).hex(), rng.bytes(h).hex()] for h in half_lengths],
        # TwoToOneCRH::compress is applied to the digests of consecutive `messages` (i, i + 1)
        "tree_leaves": [rng.bytes(leaf_len).hex() for _ in range(8)],
        "multi_proof_indexes": [5, 0, 1, 5, 6],   # unsorted, with a repeat: generate_multi_proof sorts and de-duplicates (:596)
        "update": {"index": 3, "new_leaf": rng.bytes(leaf_len).hex()},
        "parameters_head_windows": 2,             # CanonicalSerialize of Parameters { generators: first 2 windows }
    }


def main():
    v = {"note": "inputs of shim/examples/emit_vectors.rs (data, not reference outputs); field elements are decimal canonical "
                 "integers, bytes are hex; generators are affine (x, y) points of the prime-order subgroup of Jubjub, row-major "
                 "[window][power]"}
    gp = jj.pedersen_generators(0xA5A50004, 4, 256)
    v["pedersen"] = curve_section(0xA5A50004, 4, 256, gp, [0, 1, 2, 3, 32, 64, 77, 127, 128], [1, 32, 64], 30)
    gb = jj.bowe_hopwood_generators(0xA5A50005, 63, 9)
    # 63 x 9: input limit (63 * 9 * 3) / 8 = 212 bytes; TwoToOneCRH buffer (63 * 9) / 8 = 70 bytes -> halves of at most 35 bytes
    v["bowe_hopwood"] = curve_section(0xA5A50005, 63, 9, gb, [0, 1, 2, 3, 4, 32, 64, 70, 100, 211, 212], [1, 32, 35], 32)
    c = po.get_default_poseidon_parameters(2, False)
    rng = fr.SplitMix64(0xA5A50001)
    v["poseidon"] = {
        "full_rounds": c.full_rounds, "partial_rounds": c.partial_rounds, "alpha": c.alpha, "rate": c.rate, "capacity": c.capacity,
        "prime_bits": 255, "skip_matrices": 0,
        "ark": [[str(x) for x in row] for row in c.ark], "mds": [[str(x) for x in row] for row in c.mds],
        "crh_inputs": [[str(rng.fr()) for _ in range(k)] for k in (0, 1, 2, 3, 4, 5)],
        "pairs": [[str(rng.fr()), str(rng.fr())] for _ in range(3)],
        "tree_leaves": [[str(rng.fr())] for _ in range(8)],
        "multi_proof_indexes": [5, 0, 1, 5, 6],
        "update": {"index": 3, "new_leaf": [str(rng.fr())]},
        # sponge script (sponge/poseidon/mod.rs:236-257,324-344): absorb a[0..3], squeeze 2, absorb a[3..5], squeeze 4
        "sponge": {"absorb_1": [str(rng.fr()) for _ in range(3)], "squeeze_1": 2, "absorb_2": [str(rng.fr()) for _ in range(2)], "squeeze_2": 4,
                   "squeeze_bytes": 40, "squeeze_bits": 70},
    }
    with open(OUT, "w") as f:
        json.dump(v, f, indent=0, separators=(",", ":"))
        f.write("\n")
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
