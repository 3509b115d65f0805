"""Schema of tests/golden/reference_vectors.json (what shim/examples/emit_vectors.rs writes) restated for an arbitrary
implementation of the path, so that the SAME structure can be produced by the python oracle or by the GPU product and
compared field by field with the reference's file (tests/test_reference_vectors.py).

`build_vectors(inputs, impl)` -> dict in the emitter's layout.  `impl` supplies the hashes and trees:
    impl.curve(kind, section) -> object with  crh(msg) / two_to_one_evaluate(l, r) / two_to_one_compress(dl, dr) -> digest tuple,
                                              tree(leaves) -> tree object
    impl.poseidon(section)    -> object with  crh(inputs) / two_to_one(l, r) -> (d,), sponge_script(...) and tree(leaves)
    tree object: root(), height(), proof(i) -> (leaf_sibling_hash, auth_path, leaf_index), multi_proof(indexes) -> dict of the
                 four MultiPath fields, update(index, leaf)
Digests are tuples of python ints: (x, y) for Pedersen, (x,) otherwise.  Serialised forms ("uncompressed" / "compressed" hex)
are produced by `impl.ser` (the product's serialize.py -- host code, no GPU needed) from those integers.

`diff(reference, ours)` lists every leaf value of `reference` that `ours` does not reproduce; keys only one side has are
reported separately, so an emitter of a newer schema does not fail the comparison silently.
"""


def _d(t):
    return [str(int(v)) for v in t]


def _tree(tobj, leaves, requested, upd_index, upd_leaf, ser, kind):
    out = {"root": _d(tobj.root()), "height": tobj.height(), "proofs": []}
    for i in range(len(leaves)):
        sib, auth, idx = tobj.proof(i)
        out["proofs"].append({"leaf_index": idx, "leaf_sibling_hash": _d(sib), "auth_path": [_d(a) for a in auth],
                              "uncompressed": ser.path(kind, sib, auth, idx, False).hex(), "compressed": ser.path(kind, sib, auth, idx, True).hex()})
    mp = tobj.multi_proof(requested)
    out["multi_proof"] = {"requested": list(requested), "leaf_indexes": list(mp["leaf_indexes"]),
                          "auth_paths_prefix_lenghts": list(mp["auth_paths_prefix_lenghts"]),
                          "auth_paths_suffixes": [[_d(x) for x in s] for s in mp["auth_paths_suffixes"]],
                          "leaf_siblings_hashes": [_d(x) for x in mp["leaf_siblings_hashes"]],
                          "uncompressed": ser.multi_path(kind, mp, False).hex(), "compressed": ser.multi_path(kind, mp, True).hex()}
    tobj.update(upd_index, upd_leaf)
    out["update"] = {"index": upd_index, "root_after": _d(tobj.root())}
    return out


def _curve(kind, sec, impl):
    h = impl.curve(kind, sec)
    ser = impl.ser
    msgs = [bytes.fromhex(m) for m in sec["messages"]]
    digests = [h.crh(m) for m in msgs]
    out = {"crh": [{"msg": m.hex(), "digest": _d(d), "uncompressed": ser.digest(kind, d, False).hex(), "compressed": ser.digest(kind, d, True).hex()}
                   for m, d in zip(msgs, digests)]}
    out["two_to_one_evaluate"] = []
    for l, r in sec["pairs"]:
        lb, rb = bytes.fromhex(l), bytes.fromhex(r)
        out["two_to_one_evaluate"].append({"left": l, "right": r, "digest": _d(h.two_to_one_evaluate(lb, rb))})
    out["two_to_one_compress"] = [{"left": _d(a), "right": _d(b), "digest": _d(h.two_to_one_compress(a, b))} for a, b in zip(digests, digests[1:])]
    k = sec["parameters_head_windows"]
    out["parameters_head"] = {"windows": k, "uncompressed": ser.te_parameters(sec["generators"][:k], False).hex(),
                              "compressed": ser.te_parameters(sec["generators"][:k], True).hex()}
    leaves = [bytes.fromhex(x) for x in sec["tree_leaves"]]
    out["tree"] = _tree(h.tree(leaves), leaves, sec["multi_proof_indexes"], sec["update"]["index"], bytes.fromhex(sec["update"]["new_leaf"]), ser, kind)
    return out


def _poseidon(sec, impl):
    h = impl.poseidon(sec)
    ser = impl.ser
    out = {"reference_generator_matches_inputs": True,  # the emitter sets this from the reference's own find_poseidon_ark_and_mds
           "config_uncompressed": ser.poseidon_config(sec, False).hex(), "config_compressed": ser.poseidon_config(sec, True).hex()}
    out["crh"] = [{"input": list(inp), "digest": _d(h.crh([int(x) for x in inp]))} for inp in sec["crh_inputs"]]
    out["two_to_one"] = []
    for l, r in sec["pairs"]:
        d = h.two_to_one(int(l), int(r))
        out["two_to_one"].append({"left": l, "right": r, "evaluate": _d(d), "compress": _d(d)})
    sp = sec["sponge"]
    s1, s2, by, bits = h.sponge_script([int(x) for x in sp["absorb_1"]], sp["squeeze_1"], [int(x) for x in sp["absorb_2"]], sp["squeeze_2"],
                                       sp["squeeze_bytes"], sp["squeeze_bits"])
    out["sponge"] = {"squeeze_1": _d(s1), "squeeze_2": _d(s2), "squeeze_bytes_after": bytes(by).hex(), "squeeze_bits_after": [int(bool(b)) for b in bits]}
    leaves = [[int(x) for x in l] for l in sec["tree_leaves"]]
    out["tree"] = _tree(h.tree(leaves), leaves, sec["multi_proof_indexes"], sec["update"]["index"], [int(x) for x in sec["update"]["new_leaf"]], ser, "poseidon")
    return out


def build_vectors(inputs, impl, sections=("pedersen", "bowe_hopwood", "poseidon")):
    out = {}
    for s in sections:
        out[s] = _poseidon(inputs[s], impl) if s == "poseidon" else _curve(s, inputs[s], impl)
    return out


def diff(reference, ours, path=""):
    """(mismatches, only_in_reference, only_in_ours): lists of JSON paths"""
    bad, only_ref, only_ours = [], [], []
    if isinstance(reference, dict) and isinstance(ours, dict):
        for k in reference:
            if k not in ours:
                only_ref.append(path + "/" + k)
            else:
                b, r, o = diff(reference[k], ours[k], path + "/" + k)
                bad += b
                only_ref += r
                only_ours += o
        only_ours += [path + "/" + k for k in ours if k not in reference]
    elif isinstance(reference, list) and isinstance(ours, list):
        if len(reference) != len(ours):
            bad.append("%s (length %d vs %d)" % (path, len(reference), len(ours)))
        else:
            for i, (a, b) in enumerate(zip(reference, ours)):
                bb, r, o = diff(a, b, "%s[%d]" % (path, i))
                bad += bb
                only_ref += r
                only_ours += o
    else:
        a, b = reference, ours
        if isinstance(a, str) and isinstance(b, str):
            a, b = a.lower(), b.lower()
        if a != b:
            bad.append(path)
    return bad, only_ref, only_ours
