#!/usr/bin/env python3
"""Generate tests/golden/derived_vectors.json from the (KAT-pinned) python oracle.

These are NOT reference outputs (the Rust reference cannot run here); they are outputs of the
oracle restatement, committed so that the C oracle, the CPU harness and the GPU path are all
compared against one frozen set of values, including on the GPU box where only the repo travels.
Run:  python tests/golden/make_derived_vectors.py
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import poseidon as po, jubjub as jj, pedersen as pd, bowe_hopwood as bh, merkle, fr  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "derived_vectors.json")


def main():
    v = {"note": "outputs of the KAT-pinned python oracle; decimal canonical integers; bytes as hex"}
    c = po.get_default_poseidon_parameters(2, False)
    v["poseidon_rate2"] = {
        "ark_last": str(c.ark[38][2]), "mds_last": str(c.mds[2][2]),
        "permute_0_1_2": [str(x) for x in po.permute(c, [0, 1, 2])],
        "crh_1": str(po.crh_evaluate(c, [1])), "crh_1_2": str(po.crh_evaluate(c, [1, 2])),
        "crh_1_2_3": str(po.crh_evaluate(c, [1, 2, 3])), "crh_empty": str(po.crh_evaluate(c, [])),
        "compress_1_2": str(po.two_to_one_compress(c, 1, 2)),
    }
    t = merkle.MerkleTree(lambda l: po.crh_evaluate(c, l), lambda a, b: po.two_to_one_compress(c, a, b),
                          lambda a, b: po.two_to_one_compress(c, a, b), lambda x: x, leaves=[[i] for i in range(1, 9)])
    v["poseidon_merkle_8"] = {"leaves": [[str(i)] for i in range(1, 9)], "root": str(t.root()),
                              "non_leaf": [str(x) for x in t.non_leaf_nodes], "leaf_nodes": [str(x) for x in t.leaf_nodes]}
    for rate, w in ((3, False), (8, False), (2, True)):
        cc = po.get_default_poseidon_parameters(rate, w)
        st = list(range(rate + 1))
        v["poseidon_rate%d_%s" % (rate, "w" if w else "c")] = {
            "permute_iota": [str(x) for x in po.permute(cc, st)],
            "crh_1_to_5": str(po.crh_evaluate(cc, [1, 2, 3, 4, 5]))}
    # Pedersen 4x256 and Bowe-Hopwood 63x9 over Jubjub with seeded generators (seed = config seeds of BASELINE.md)
    msg128 = fr.SplitMix64(0xA5A50004).bytes(128)
    g = jj.pedersen_generators(0xA5A50004, 4, 256)
    h = pd.evaluate(g, 4, 256, msg128)
    h32 = pd.evaluate(g, 4, 256, msg128[:32])
    v["pedersen_4x256"] = {"generator_seed": "0xA5A50004", "g00": [str(g[0][0][0]), str(g[0][0][1])],
                           "g255_3": [str(g[255][3][0]), str(g[255][3][1])],
                           "msg": msg128.hex(), "digest": [str(h[0]), str(h[1])], "digest_first32": [str(h32[0]), str(h32[1])],
                           "compress_h_h32": [str(x) for x in pd.two_to_one_compress(g, 4, 256, h, h32)]}
    gb = jj.bowe_hopwood_generators(0xA5A50005, 63, 9)
    msg32 = fr.SplitMix64(0xA5A50005).bytes(32)
    x32 = bh.evaluate(gb, 63, 9, msg32)
    x70 = bh.evaluate(gb, 63, 9, msg128[:70])
    v["bowe_hopwood_63x9"] = {"generator_seed": "0xA5A50005", "g00": [str(gb[0][0][0]), str(gb[0][0][1])],
                              "msg32": msg32.hex(), "digest32": str(x32), "msg70": msg128[:70].hex(), "digest70": str(x70),
                              "compress": str(bh.two_to_one_compress(gb, 63, 9, x32, x70)),
                              "zero_3bytes": str(bh.evaluate(gb, 63, 9, bytes(3)))}
    leaves = [fr.SplitMix64(100 + i).bytes(32) for i in range(4)]
    tb = merkle.MerkleTree(lambda l: bh.evaluate(gb, 63, 9, l), lambda a, b: bh.two_to_one_evaluate(gb, 63, 9, a, b),
                           lambda a, b: bh.two_to_one_compress(gb, 63, 9, a, b), jj.fq_serialize, leaves=leaves)
    v["bowe_hopwood_merkle_4"] = {"leaves": [l.hex() for l in leaves], "root": str(tb.root())}
    tp = merkle.MerkleTree(lambda l: pd.evaluate(g, 4, 256, l), lambda a, b: pd.two_to_one_evaluate(g, 4, 256, a, b),
                           lambda a, b: pd.two_to_one_compress(g, 4, 256, a, b), jj.serialize_uncompressed, leaves=leaves)
    v["pedersen_merkle_4"] = {"leaves": [l.hex() for l in leaves], "root": [str(x) for x in tp.root()]}
    with open(OUT, "w") as f:
        json.dump(v, f, indent=1)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
