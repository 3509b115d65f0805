#!/usr/bin/env python3
"""Extract the known-answer constants the reference's own unit tests hold for the
Poseidon path into tests/golden/poseidon_kats.json (DATA only: decimal constants
and the call arguments they belong to; no reference source text is kept).

Run in the build container, where /root/reference exists:
    python tests/golden/extract_reference_kats.py
Sources:
  sponge/poseidon/traits.rs:163-358    ark[0][0] / mds[0][0] of the 14 default configs
  sponge/poseidon/grain_lfsr.rs:190-218 first LFSR outputs
  sponge/poseidon/mod.rs:381-404       sponge absorb [0,1,2] squeeze 3
"""
import json
import os
import re

REF = "/root/reference/crypto-primitives/src/sponge/poseidon"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "poseidon_kats.json")

num = re.compile(r'"(\d{60,80})"')


def main():
    kats = {"source": "arkworks-rs/crypto-primitives @ 2024-10-24 unit-test constants", "default_params": [], }
    txt = open(os.path.join(REF, "traits.rs")).read()
    test = txt[txt.index("mod test"):]
    # blocks: let <name> = Fr::get_default_poseidon_parameters(rate, weights) ... ark[0][0] .. mds[0][0]
    pat = re.compile(
        r"get_default_poseidon_parameters\((\d+),\s*(true|false)\).*?ark\[0\]\[0\].*?\"(\d+)\".*?mds\[0\]\[0\].*?\"(\d+)\"",
        re.S)
    for m in pat.finditer(test):
        kats["default_params"].append({
            "rate": int(m.group(1)), "optimized_for_weights": m.group(2) == "true",
            "ark00": m.group(3), "mds00": m.group(4)})
    assert len(kats["default_params"]) == 14, len(kats["default_params"])

    txt = open(os.path.join(REF, "grain_lfsr.rs")).read()
    test = txt[txt.index("mod test"):]
    args = re.search(r"PoseidonGrainLFSR::new\(false,\s*(\d+),\s*(\d+),\s*(\d+),\s*(\d+)\)", test)
    vals = num.findall(test)
    assert len(vals) == 4
    kats["grain_lfsr"] = {
        "prime_num_bits": int(args.group(1)), "state_len": int(args.group(2)),
        "full_rounds": int(args.group(3)), "partial_rounds": int(args.group(4)),
        "rejection_sampling": vals[:2], "mod_p": vals[2:]}

    txt = open(os.path.join(REF, "mod.rs")).read()
    test = txt[txt.index("fn test_poseidon_sponge_consistency"):]
    vals = num.findall(test)
    assert len(vals) == 3
    kats["sponge_consistency"] = {"rate": 2, "optimized_for_weights": False,
                                  "absorb": ["0", "1", "2"], "squeeze": vals}
    with open(OUT, "w") as f:
        json.dump(kats, f, indent=1)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
