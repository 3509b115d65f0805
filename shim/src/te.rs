//! Pedersen and Bowe-Hopwood CRH over Jubjub (`ark_ed_on_bls12_381`) on the GPU
//! (`crh/pedersen/mod.rs:23-209`, `crh/bowe_hopwood/mod.rs:31-240`) and the Pedersen hashes composed with `TECompressor`
//! (`crh/injective_map/mod.rs:16-108`).
use crate::runtime::{check, fr_from_limbs, with_runtime, words};
use crate::{ffi, Error, Fr};
use ark_crypto_primitives::crh::{bowe_hopwood, pedersen, CRHScheme, TwoToOneCRHScheme};
use ark_ec::CurveGroup;
use ark_ed_on_bls12_381::{EdwardsAffine, EdwardsConfig, EdwardsProjective};
use ark_serialize::CanonicalSerialize;
use ark_std::{borrow::Borrow, marker::PhantomData, rand::Rng, vec::Vec};

/// generators -> affine x || y in wire format, row-major [window][power]
fn affine_words(gens: &[Vec<EdwardsProjective>]) -> Vec<Fr> {
    let flat: Vec<EdwardsProjective> = gens.iter().flat_map(|r| r.iter().copied()).collect();
    EdwardsProjective::normalize_batch(&flat).into_iter().flat_map(|a| [a.x, a.y]).collect()
}
fn te_handle(kind: i32, gens: &[Vec<EdwardsProjective>]) -> Result<*mut ffi::AkpTeParams, Error> {
    let num_windows = gens.len();
    let window_size = gens.first().map_or(0, |r| r.len());
    assert!(gens.iter().all(|r| r.len() == window_size), "ragged generator table");
    let xy = affine_words(gens);
    // kind and window shape in the tag, every generator coordinate in the key (compared in full on a hit)
    let tag = (kind as u64) << 56 ^ (window_size as u64) << 32 ^ num_windows as u64;
    with_runtime(|rt| {
        let ctx = rt.ctx.0;
        rt.te.get_or_create(tag, &xy, || {
            let mut h = core::ptr::null_mut();
            check(unsafe { ffi::akp_te_params_create(ctx, kind, window_size as u32, num_windows as u32, words(&xy), &mut h) }, 0)?;
            Ok(h)
        })
    })
}
/// n equal-length messages -> digests as wire words (2 Fr per Pedersen digest, 1 per Bowe-Hopwood digest)
fn crh_words(h: *mut ffi::AkpTeParams, fe: usize, msgs: &[u8], n: usize, msg_len: usize) -> Result<Vec<u64>, Error> {
    let mut out = vec![0u64; n * fe * 4];
    check(unsafe { ffi::akp_te_crh_batch(h, msgs.as_ptr(), n, msg_len, out.as_mut_ptr()) }, msg_len)?;
    Ok(out)
}
fn two_to_one_words(h: *mut ffi::AkpTeParams, fe: usize, left: &[u8], right: &[u8]) -> Result<Vec<u64>, Error> {
    let mut out = vec![0u64; fe * 4];
    check(unsafe { ffi::akp_te_two_to_one_batch(h, left.as_ptr(), right.as_ptr(), 1, left.len(), out.as_mut_ptr()) }, left.len())?;
    Ok(out)
}
/// n inputs of DIFFERENT lengths in one launch (`akp_te_crh_batch_ragged`): the analogue of mapping `evaluate` over a slice of inputs,
/// each hashed with its own length (`crh/pedersen/mod.rs:82-99`, `crh/bowe_hopwood/mod.rs:131-138`)
fn crh_words_ragged(h: *mut ffi::AkpTeParams, fe: usize, msgs: &[&[u8]]) -> Result<Vec<u64>, Error> {
    let mut offs = Vec::with_capacity(msgs.len() + 1);
    let mut at = 0u64;
    offs.push(at);
    for m in msgs {
        at += m.len() as u64;
        offs.push(at);
    }
    let flat: Vec<u8> = msgs.iter().flat_map(|m| m.iter().copied()).collect();
    let mut out = vec![0u64; msgs.len() * fe * 4];
    check(unsafe { ffi::akp_te_crh_batch_ragged(h, flat.as_ptr(), offs.as_ptr(), msgs.len(), out.as_mut_ptr()) }, msgs.iter().map(|m| m.len()).max().unwrap_or(0))?;
    Ok(out)
}
fn point(w: &[u64]) -> EdwardsAffine {
    // the library returns a point of the curve (sum of the caller's generators): no curve / subgroup re-check
    EdwardsAffine::new_unchecked(fr_from_limbs([w[0], w[1], w[2], w[3]]), fr_from_limbs([w[4], w[5], w[6], w[7]]))
}

/// `pedersen::CRH<EdwardsProjective, W>` (`crh/pedersen/mod.rs:33-130`)
pub struct PedersenCRH<W: pedersen::Window>(PhantomData<W>);
impl<W: pedersen::Window> PedersenCRH<W> {
    /// n messages of `msg_len` bytes each, concatenated
    pub fn evaluate_batch(parameters: &pedersen::Parameters<EdwardsProjective>, msgs: &[u8], msg_len: usize) -> Result<Vec<EdwardsAffine>, Error> {
        let n = if msg_len == 0 { 1 } else { msgs.len() / msg_len };
        let w = crh_words(te_handle(ffi::AKP_TE_PEDERSEN, &parameters.generators)?, 2, msgs, n, msg_len)?;
        Ok(w.chunks_exact(8).map(point).collect())
    }
    /// inputs of different lengths, each hashed as `evaluate` would hash it
    pub fn evaluate_many(parameters: &pedersen::Parameters<EdwardsProjective>, msgs: &[&[u8]]) -> Result<Vec<EdwardsAffine>, Error> {
        let w = crh_words_ragged(te_handle(ffi::AKP_TE_PEDERSEN, &parameters.generators)?, 2, msgs)?;
        Ok(w.chunks_exact(8).map(point).collect())
    }
}
impl<W: pedersen::Window> CRHScheme for PedersenCRH<W> {
    type Input = [u8];
    type Output = EdwardsAffine;
    type Parameters = pedersen::Parameters<EdwardsProjective>;

    fn setup<R: Rng>(rng: &mut R) -> Result<Self::Parameters, Error> {
        pedersen::CRH::<EdwardsProjective, W>::setup(rng) // host-side sampling, unchanged (crh/pedersen/mod.rs:64-74)
    }
    fn evaluate<T: Borrow<Self::Input>>(parameters: &Self::Parameters, input: T) -> Result<Self::Output, Error> {
        let input = input.borrow();
        // the reference panics on both conditions (crh/pedersen/mod.rs:82-109); keep that contract
        assert!(input.len() * 8 <= W::WINDOW_SIZE * W::NUM_WINDOWS, "incorrect input length {:?} for window params {:?}✕{:?}", input.len(), W::WINDOW_SIZE, W::NUM_WINDOWS);
        assert_eq!(parameters.generators.len(), W::NUM_WINDOWS, "Incorrect pp of size {:?}✕{:?} for window params {:?}✕{:?}",
                   parameters.generators[0].len(), parameters.generators.len(), W::WINDOW_SIZE, W::NUM_WINDOWS);
        Ok(Self::evaluate_batch(parameters, input, input.len())?[0])
    }
}
/// `pedersen::TwoToOneCRH<EdwardsProjective, W>` (`crh/pedersen/mod.rs:132-198`)
pub struct PedersenTwoToOneCRH<W: pedersen::Window>(PhantomData<W>);
impl<W: pedersen::Window> TwoToOneCRHScheme for PedersenTwoToOneCRH<W> {
    type Input = [u8];
    type Output = EdwardsAffine;
    type Parameters = pedersen::Parameters<EdwardsProjective>;

    fn setup<R: Rng>(rng: &mut R) -> Result<Self::Parameters, Error> {
        pedersen::CRH::<EdwardsProjective, W>::setup(rng)
    }
    fn evaluate<T: Borrow<Self::Input>>(parameters: &Self::Parameters, left_input: T, right_input: T) -> Result<Self::Output, Error> {
        let (l, r) = (left_input.borrow(), right_input.borrow());
        assert_eq!(l.len(), r.len(), "left and right input should be of equal length"); // :169-173
        debug_assert!(l.len() * 8 <= W::WINDOW_SIZE * W::NUM_WINDOWS / 2); // :166-167 (release builds zip-truncate, as the library does)
        Ok(point(&two_to_one_words(te_handle(ffi::AKP_TE_PEDERSEN, &parameters.generators)?, 2, l, r)?))
    }
    fn compress<T: Borrow<Self::Output>>(parameters: &Self::Parameters, left_input: T, right_input: T) -> Result<Self::Output, Error> {
        // :187-197: serialise uncompressed, then evaluate (the device-side akp_te_compress_batch does the same in one call
        // for whole levels; one pair at a time the host serialisation is as good)
        let (mut l, mut r) = (Vec::new(), Vec::new());
        left_input.borrow().serialize_uncompressed(&mut l).map_err(Error::SerializationError)?;
        right_input.borrow().serialize_uncompressed(&mut r).map_err(Error::SerializationError)?;
        Self::evaluate(parameters, l, r)
    }
}

/// `bowe_hopwood::CRH<EdwardsConfig, W>` (`crh/bowe_hopwood/mod.rs:38-187`): digest = x coordinate
pub struct BoweHopwoodCRH<W: pedersen::Window>(PhantomData<W>);
impl<W: pedersen::Window> BoweHopwoodCRH<W> {
    pub fn evaluate_batch(parameters: &bowe_hopwood::Parameters<EdwardsConfig>, msgs: &[u8], msg_len: usize) -> Result<Vec<Fr>, Error> {
        let n = if msg_len == 0 { 1 } else { msgs.len() / msg_len };
        let w = crh_words(te_handle(ffi::AKP_TE_BOWE_HOPWOOD, &parameters.generators)?, 1, msgs, n, msg_len)?;
        Ok(w.chunks_exact(4).map(|c| fr_from_limbs([c[0], c[1], c[2], c[3]])).collect())
    }
    /// inputs of different lengths, each hashed as `evaluate` would hash it (padded to a multiple of 3 bits only, `:131-138`)
    pub fn evaluate_many(parameters: &bowe_hopwood::Parameters<EdwardsConfig>, msgs: &[&[u8]]) -> Result<Vec<Fr>, Error> {
        let w = crh_words_ragged(te_handle(ffi::AKP_TE_BOWE_HOPWOOD, &parameters.generators)?, 1, msgs)?;
        Ok(w.chunks_exact(4).map(|c| fr_from_limbs([c[0], c[1], c[2], c[3]])).collect())
    }
}
impl<W: pedersen::Window> CRHScheme for BoweHopwoodCRH<W> {
    type Input = [u8];
    type Output = Fr; // <EdwardsConfig as CurveConfig>::BaseField
    type Parameters = bowe_hopwood::Parameters<EdwardsConfig>;

    fn setup<R: Rng>(rng: &mut R) -> Result<Self::Parameters, Error> {
        bowe_hopwood::CRH::<EdwardsConfig, W>::setup(rng) // includes the 63-chunk bound check (:81-101)
    }
    fn evaluate<T: Borrow<Self::Input>>(parameters: &Self::Parameters, input: T) -> Result<Self::Output, Error> {
        let input = input.borrow();
        assert!(input.len() * 8 <= W::WINDOW_SIZE * W::NUM_WINDOWS * bowe_hopwood::CHUNK_SIZE, "incorrect input bitlength {:?} for window params {:?}x{:?}x{}",
                input.len() * 8, W::WINDOW_SIZE, W::NUM_WINDOWS, bowe_hopwood::CHUNK_SIZE); // :121-129
        Ok(Self::evaluate_batch(parameters, input, input.len())?[0])
    }
}
/// `bowe_hopwood::TwoToOneCRH<EdwardsConfig, W>` (`crh/bowe_hopwood/mod.rs:189-240`)
pub struct BoweHopwoodTwoToOneCRH<W: pedersen::Window>(PhantomData<W>);
impl<W: pedersen::Window> TwoToOneCRHScheme for BoweHopwoodTwoToOneCRH<W> {
    type Input = [u8];
    type Output = Fr;
    type Parameters = bowe_hopwood::Parameters<EdwardsConfig>;

    fn setup<R: Rng>(rng: &mut R) -> Result<Self::Parameters, Error> {
        bowe_hopwood::CRH::<EdwardsConfig, W>::setup(rng)
    }
    fn evaluate<T: Borrow<Self::Input>>(parameters: &Self::Parameters, left_input: T, right_input: T) -> Result<Self::Output, Error> {
        let (l, r) = (left_input.borrow(), right_input.borrow());
        assert_eq!(l.len(), r.len(), "left and right input should be of equal length"); // :213-217
        let w = two_to_one_words(te_handle(ffi::AKP_TE_BOWE_HOPWOOD, &parameters.generators)?, 1, l, r)?;
        Ok(fr_from_limbs([w[0], w[1], w[2], w[3]]))
    }
    fn compress<T: Borrow<Self::Output>>(parameters: &Self::Parameters, left_input: T, right_input: T) -> Result<Self::Output, Error> {
        let (mut l, mut r) = (Vec::new(), Vec::new()); // :229-239
        left_input.borrow().serialize_uncompressed(&mut l).map_err(Error::SerializationError)?;
        right_input.borrow().serialize_uncompressed(&mut r).map_err(Error::SerializationError)?;
        Self::evaluate(parameters, l, r)
    }
}

/// `PedersenCRHCompressor<EdwardsProjective, TECompressor, W>` (`crh/injective_map/mod.rs:33-62`): the Pedersen hash followed
/// by the injective map (x, y) -> x.  Runs as the library's third kind (`AKP_TE_PEDERSEN_X`): Pedersen tables, x-only output.
pub struct PedersenCRHCompressor<W: pedersen::Window>(PhantomData<W>);
impl<W: pedersen::Window> PedersenCRHCompressor<W> {
    pub fn evaluate_batch(parameters: &pedersen::Parameters<EdwardsProjective>, msgs: &[u8], msg_len: usize) -> Result<Vec<Fr>, Error> {
        let n = if msg_len == 0 { 1 } else { msgs.len() / msg_len };
        let w = crh_words(te_handle(ffi::AKP_TE_PEDERSEN_X, &parameters.generators)?, 1, msgs, n, msg_len)?;
        Ok(w.chunks_exact(4).map(|c| fr_from_limbs([c[0], c[1], c[2], c[3]])).collect())
    }
}
impl<W: pedersen::Window> CRHScheme for PedersenCRHCompressor<W> {
    type Input = [u8];
    type Output = Fr; // TECompressor::Output = <EdwardsConfig as CurveConfig>::BaseField (:21-22)
    type Parameters = pedersen::Parameters<EdwardsProjective>;

    fn setup<R: Rng>(rng: &mut R) -> Result<Self::Parameters, Error> {
        pedersen::CRH::<EdwardsProjective, W>::setup(rng) // :47-52
    }
    fn evaluate<T: Borrow<Self::Input>>(parameters: &Self::Parameters, input: T) -> Result<Self::Output, Error> {
        let input = input.borrow();
        assert!(input.len() * 8 <= W::WINDOW_SIZE * W::NUM_WINDOWS, "incorrect input length {:?} for window params {:?}✕{:?}", input.len(), W::WINDOW_SIZE, W::NUM_WINDOWS);
        assert_eq!(parameters.generators.len(), W::NUM_WINDOWS);
        Ok(Self::evaluate_batch(parameters, input, input.len())?[0])
    }
}
/// `PedersenTwoToOneCRHCompressor<EdwardsProjective, TECompressor, W>` (`crh/injective_map/mod.rs:64-108`)
pub struct PedersenTwoToOneCRHCompressor<W: pedersen::Window>(PhantomData<W>);
impl<W: pedersen::Window> TwoToOneCRHScheme for PedersenTwoToOneCRHCompressor<W> {
    type Input = [u8];
    type Output = Fr;
    type Parameters = pedersen::Parameters<EdwardsProjective>;

    fn setup<R: Rng>(rng: &mut R) -> Result<Self::Parameters, Error> {
        pedersen::CRH::<EdwardsProjective, W>::setup(rng) // :76-78
    }
    fn evaluate<T: Borrow<Self::Input>>(parameters: &Self::Parameters, left_input: T, right_input: T) -> Result<Self::Output, Error> {
        let (l, r) = (left_input.borrow(), right_input.borrow());
        assert_eq!(l.len(), r.len(), "left and right input should be of equal length");
        let w = two_to_one_words(te_handle(ffi::AKP_TE_PEDERSEN_X, &parameters.generators)?, 1, l, r)?;
        Ok(fr_from_limbs([w[0], w[1], w[2], w[3]]))
    }
    fn compress<T: Borrow<Self::Output>>(parameters: &Self::Parameters, left_input: T, right_input: T) -> Result<Self::Output, Error> {
        let (mut l, mut r) = (Vec::new(), Vec::new()); // :96-107 to_uncompressed_bytes!
        left_input.borrow().serialize_uncompressed(&mut l).map_err(Error::SerializationError)?;
        right_input.borrow().serialize_uncompressed(&mut r).map_err(Error::SerializationError)?;
        Self::evaluate(parameters, l, r)
    }
}

/// (window_size, num_windows, affine x || y words) of a generator table: what `akp_te_params_create` takes (the sharded tree
/// creates its handles on other contexts than this thread's: [`crate::sharded`])
pub(crate) fn window_words(gens: &[Vec<EdwardsProjective>]) -> (usize, usize, Vec<Fr>) {
    let window_size = gens.first().map_or(0, |r| r.len());
    assert!(gens.iter().all(|r| r.len() == window_size), "ragged generator table");
    (window_size, gens.len(), affine_words(gens))
}
pub(crate) fn pedersen_handle(p: &pedersen::Parameters<EdwardsProjective>) -> Result<*mut ffi::AkpTeParams, Error> {
    te_handle(ffi::AKP_TE_PEDERSEN, &p.generators)
}
pub(crate) fn pedersen_x_handle(p: &pedersen::Parameters<EdwardsProjective>) -> Result<*mut ffi::AkpTeParams, Error> {
    te_handle(ffi::AKP_TE_PEDERSEN_X, &p.generators)
}
pub(crate) fn bowe_hopwood_handle(p: &bowe_hopwood::Parameters<EdwardsConfig>) -> Result<*mut ffi::AkpTeParams, Error> {
    te_handle(ffi::AKP_TE_BOWE_HOPWOOD, &p.generators)
}
