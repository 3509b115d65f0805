"""CPU oracle for the native CRH/sponge hot path of ark-crypto-primitives.

THIS PACKAGE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it; the product (``crypto_primitives_amd``) never does and fails loudly
when its HIP library is missing.

It restates, in plain Python big-int arithmetic (small cases, parameter
generation) and in plain C (``oracle/c/akp_oracle.c``, bulk cases and the timed
CPU baseline), the algorithms of the reference crate at
``/root/reference/crypto-primitives/src``; every function cites the reference
file:line it follows.

Pinning status (see DESIGN.md "Oracle"):
  * Poseidon (Grain-LFSR parameters, permutation, duplex sponge, CRH):
    PINNED by the reference's own known-answer tests
    (sponge/poseidon/mod.rs:381-404, sponge/poseidon/traits.rs:163-358,
    sponge/poseidon/grain_lfsr.rs:190-218) -> tests/golden/poseidon_kats.json.
  * Pedersen / Bowe-Hopwood / Merkle digests: PARITY UNPINNED at the value
    level -- the reference holds no absolute vectors for them (its tests only
    assert native==gadget and proof round trips) and the Rust reference cannot
    be built here (no rustc/cargo; ark-ff/ark-ec are un-vendored git
    dependencies).  They are pinned structurally: group-law identities,
    scalar-multiplication form of the hash, the generator-independent known
    answers that follow from the reference code, and proof round trips.
    Since round 2 the Jubjub GROUP LAW underneath them is pinned at value
    level by the one absolute vector of the un-vendored curve dependency,
    ark-ed-on-bls12-381's test_scalar_multiplication (f1 * f2 * g), kept in
    tests/golden/jubjub_upstream_kat.json: a Pedersen hash over one window
    of generators 2^j * g IS that scalar multiplication.  Bit order, padding,
    the Bowe-Hopwood chunk encoding and the byte encodings stay unpinned.
"""
