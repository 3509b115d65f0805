//! MI355X backend for the native (non-R1CS) CRH / sponge / Merkle path of `ark-crypto-primitives`.
//!
//! The reference's operator API is three traits of per-item static functions
//! (`crh/mod.rs:18-51`, `sponge/mod.rs:101-191`).  This crate implements them over `libakp.so`
//! (`include/akp.h`) so that existing generic code keeps compiling, and adds what a GPU needs to win:
//! batch entry points and [`merkle::GpuMerkleTree`], whose `new` hashes whole levels per launch and keeps the
//! tree in HBM.
//!
//! * [`poseidon::CRH`], [`poseidon::TwoToOneCRH`] -- `CRHScheme` / `TwoToOneCRHScheme` for Poseidon over BLS12-381 Fr
//! * [`poseidon::GpuPoseidonSponge`] -- `CryptographicSponge + FieldBasedCryptographicSponge<Fr> + SpongeExt`
//! * [`te::PedersenCRH`], [`te::PedersenTwoToOneCRH`], [`te::BoweHopwoodCRH`], [`te::BoweHopwoodTwoToOneCRH`]
//!   over Jubjub (`ark_ed_on_bls12_381`); [`te::PedersenCRHCompressor`], [`te::PedersenTwoToOneCRHCompressor`] = the Pedersen
//!   hashes composed with `TECompressor` (`crh/injective_map/mod.rs`)
//! * [`merkle::GpuMerkleTree`] -- `MerkleTree<P>` resident on the device (new / blank / root / generate_proof /
//!   generate_multi_proof / update / check_update), plus `into_reference_vectors` for code that reads the
//!   reference's `leaf_nodes` / `non_leaf_nodes`.
//! * [`sharded::MultiGpu`], [`sharded::GpuShardedMerkleTree`] -- the same tree over all GPUs of a node from this one process
//!   (leaf-range shards resident per device, ONE all-gather of the sub-roots over RCCL inside the library): new / root /
//!   generate_proof(s) / update_batch.
//!
//! Threading: an `akp_ctx` is not thread-safe, distinct contexts are.  The reference calls `evaluate` /
//! `compress` concurrently from rayon workers (`merkle_tree/mod.rs:417,458,494`); here every OS thread lazily gets
//! its own context and its own parameter-handle cache ([`runtime`]), so the trait methods are re-entrant without a
//! lock.  (Per-item calls cost one kernel launch each -- use the `*_batch` functions or `GpuMerkleTree`.)
#![forbid(unsafe_op_in_unsafe_fn)]

pub mod ffi;
pub mod merkle;
pub mod poseidon;
pub mod runtime;
pub mod sharded;
pub mod te;

pub use ark_crypto_primitives::Error;

/// BLS12-381 scalar field = Jubjub base field: the only field the kernels implement.
pub type Fr = ark_bls12_381::Fr;
