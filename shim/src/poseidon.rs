//! Poseidon over BLS12-381 Fr on the GPU: `CRHScheme`, `TwoToOneCRHScheme` (`crh/poseidon/mod.rs:14-79`) and the
//! duplex sponge (`sponge/poseidon/mod.rs:47-370`).
use crate::runtime::{check, flatten, fr_from_limbs, with_runtime, words, words_mut};
use crate::{ffi, Error, Fr};
use ark_crypto_primitives::crh::{CRHScheme, TwoToOneCRHScheme};
use ark_crypto_primitives::sponge::poseidon::PoseidonConfig;
use ark_crypto_primitives::sponge::{Absorb, CryptographicSponge, DuplexSpongeMode, FieldBasedCryptographicSponge, SpongeExt};
use ark_ff::{BigInteger, PrimeField, Zero};
use ark_std::{borrow::Borrow, rand::Rng, vec::Vec};

/// device handle for `cfg` on the calling thread's context (created once per distinct parameter set)
pub(crate) fn handle(cfg: &PoseidonConfig<Fr>) -> Result<*mut ffi::AkpPoseidon, Error> {
    // scalar parameters in the tag, every round key and matrix entry in the key: a cache hit compares all of them
    let tag = (cfg.full_rounds as u64) << 48 ^ (cfg.partial_rounds as u64) << 32 ^ (cfg.rate as u64) << 16 ^ (cfg.capacity as u64) ^ cfg.alpha.rotate_left(24);
    // PoseidonConfig::new's shape asserts (sponge/poseidon/mod.rs:191-217) are repeated by the library
    let (ark, mds) = (flatten(&cfg.ark), flatten(&cfg.mds));
    let mut key = Vec::with_capacity(ark.len() + mds.len() + 1);
    key.push(Fr::from(cfg.alpha)); // the full alpha (the tag only carries a mix of it)
    key.extend_from_slice(&ark);
    key.extend_from_slice(&mds);
    with_runtime(|rt| {
        let ctx = rt.ctx.0;
        rt.poseidon.get_or_create(tag, &key, || {
            let mut h = core::ptr::null_mut();
            check(
                unsafe {
                    ffi::akp_poseidon_params_create(ctx, cfg.full_rounds as u32, cfg.partial_rounds as u32, cfg.alpha, cfg.rate as u32,
                                                    cfg.capacity as u32, words(&ark), words(&mds), &mut h)
                },
                0,
            )?;
            Ok(h)
        })
    })
}

/// n inputs of `k` elements each, row-major -> n digests (`akp_poseidon_crh_batch`)
pub fn crh_batch(cfg: &PoseidonConfig<Fr>, inputs: &[Fr], k: usize) -> Result<Vec<Fr>, Error> {
    let n = if k == 0 { 1 } else { inputs.len() / k };
    assert!(k == 0 || inputs.len() == n * k, "inputs must hold n * k elements");
    let h = handle(cfg)?;
    let mut out = vec![Fr::zero(); n];
    check(unsafe { ffi::akp_poseidon_crh_batch(h, words(inputs), n, k, words_mut(&mut out)) }, k)?;
    Ok(out)
}
/// out[i] = H(left[i], right[i]) (`akp_poseidon_two_to_one_batch`)
pub fn two_to_one_batch(cfg: &PoseidonConfig<Fr>, left: &[Fr], right: &[Fr]) -> Result<Vec<Fr>, Error> {
    assert_eq!(left.len(), right.len());
    let h = handle(cfg)?;
    let mut out = vec![Fr::zero(); left.len()];
    check(unsafe { ffi::akp_poseidon_two_to_one_batch(h, words(left), words(right), left.len(), words_mut(&mut out)) }, 2)?;
    Ok(out)
}
/// `PoseidonSponge::permute` on n states of t elements, in place (`akp_poseidon_permute_batch`)
pub fn permute_batch(cfg: &PoseidonConfig<Fr>, states: &mut [Fr]) -> Result<(), Error> {
    let t = cfg.rate + cfg.capacity;
    assert_eq!(states.len() % t, 0);
    let h = handle(cfg)?;
    check(unsafe { ffi::akp_poseidon_permute_batch(h, words_mut(states), states.len() / t) }, t)
}

/// `CRH::evaluate` over inputs of DIFFERENT lengths in one launch (`akp_poseidon_crh_batch_ragged`; `crh/poseidon/mod.rs:30-40`
/// takes any `&[F]`)
pub fn crh_many(cfg: &PoseidonConfig<Fr>, inputs: &[&[Fr]]) -> Result<Vec<Fr>, Error> {
    let mut offs = Vec::with_capacity(inputs.len() + 1);
    let mut at = 0u64;
    offs.push(at);
    for x in inputs {
        at += x.len() as u64;
        offs.push(at);
    }
    let flat: Vec<Fr> = inputs.iter().flat_map(|x| x.iter().copied()).collect();
    let mut out = vec![Fr::zero(); inputs.len()];
    check(unsafe { ffi::akp_poseidon_crh_batch_ragged(handle(cfg)?, words(&flat), offs.as_ptr(), inputs.len(), words_mut(&mut out)) }, 0)?;
    Ok(out)
}

/// `poseidon::CRH<Fr>` (`crh/poseidon/mod.rs:14-41`)
pub struct CRH;
impl CRHScheme for CRH {
    type Input = [Fr];
    type Output = Fr;
    type Parameters = PoseidonConfig<Fr>;

    fn setup<R: Rng>(_rng: &mut R) -> Result<Self::Parameters, Error> {
        // automatic generation of parameters is not implemented in the reference either (crh/poseidon/mod.rs:24-28)
        unimplemented!()
    }
    fn evaluate<T: Borrow<Self::Input>>(parameters: &Self::Parameters, input: T) -> Result<Self::Output, Error> {
        let input = input.borrow();
        Ok(crh_batch(parameters, input, input.len())?[0])
    }
}

/// `poseidon::TwoToOneCRH<Fr>` (`crh/poseidon/mod.rs:43-80`)
pub struct TwoToOneCRH;
impl TwoToOneCRHScheme for TwoToOneCRH {
    type Input = Fr;
    type Output = Fr;
    type Parameters = PoseidonConfig<Fr>;

    fn setup<R: Rng>(_rng: &mut R) -> Result<Self::Parameters, Error> {
        unimplemented!()
    }
    fn evaluate<T: Borrow<Self::Input>>(parameters: &Self::Parameters, left_input: T, right_input: T) -> Result<Self::Output, Error> {
        Self::compress(parameters, left_input, right_input)
    }
    fn compress<T: Borrow<Self::Output>>(parameters: &Self::Parameters, left_input: T, right_input: T) -> Result<Self::Output, Error> {
        Ok(two_to_one_batch(parameters, &[*left_input.borrow()], &[*right_input.borrow()])?[0])
    }
}

/// `PoseidonSponge<Fr>` with its state on the device (a batch of one `akp_sponge`).  The duplex bookkeeping
/// (`DuplexSpongeMode`, `sponge/mod.rs:195-206`) lives in the library; `into_state` / `from_state` move it across.
pub struct GpuPoseidonSponge {
    pub parameters: PoseidonConfig<Fr>,
    h: *mut ffi::AkpSponge,
}
/// what `SpongeExt::into_state` returns (the reference's `PoseidonSpongeState` has private fields)
#[derive(Clone)]
pub struct GpuPoseidonSpongeState {
    pub state: Vec<Fr>,
    pub mode: DuplexSpongeMode,
}
impl GpuPoseidonSponge {
    fn raw_state(&self) -> GpuPoseidonSpongeState {
        let t = self.parameters.rate + self.parameters.capacity;
        let mut state = vec![Fr::zero(); t];
        let (mut mode, mut index) = (0i32, 0u32);
        let rc = unsafe { ffi::akp_sponge_get_state(self.h, words_mut(&mut state), &mut mode, &mut index) };
        check(rc, t).expect("akp_sponge_get_state");
        let mode = if mode == 0 {
            DuplexSpongeMode::Absorbing { next_absorb_index: index as usize }
        } else {
            DuplexSpongeMode::Squeezing { next_squeeze_index: index as usize }
        };
        GpuPoseidonSpongeState { state, mode }
    }
}
impl Clone for GpuPoseidonSponge {
    fn clone(&self) -> Self {
        Self::from_state(self.raw_state(), &self.parameters)
    }
}
impl Drop for GpuPoseidonSponge {
    fn drop(&mut self) {
        unsafe { ffi::akp_sponge_destroy(self.h) }
    }
}
impl CryptographicSponge for GpuPoseidonSponge {
    type Config = PoseidonConfig<Fr>;

    fn new(parameters: &Self::Config) -> Self {
        let p = handle(parameters).expect("Poseidon parameters rejected by libakp");
        let mut h = core::ptr::null_mut();
        check(unsafe { ffi::akp_sponge_create(p, 1, &mut h) }, 0).expect("akp_sponge_create");
        Self { parameters: parameters.clone(), h }
    }
    fn absorb(&mut self, input: &impl Absorb) {
        let elems = input.to_sponge_field_elements_as_vec::<Fr>();
        if elems.is_empty() {
            return; // sponge/poseidon/mod.rs:238-240
        }
        check(unsafe { ffi::akp_sponge_absorb(self.h, words(&elems), elems.len()) }, elems.len()).expect("akp_sponge_absorb");
    }
    fn squeeze_bytes(&mut self, num_bytes: usize) -> Vec<u8> {
        // sponge/poseidon/mod.rs:259-274
        let usable_bytes = ((Fr::MODULUS_BIT_SIZE - 1) / 8) as usize;
        let num_elements = (num_bytes + usable_bytes - 1) / usable_bytes;
        let mut bytes = Vec::with_capacity(usable_bytes * num_elements);
        for elem in self.squeeze_native_field_elements(num_elements) {
            bytes.extend_from_slice(&elem.into_bigint().to_bytes_le()[..usable_bytes]);
        }
        bytes.truncate(num_bytes);
        bytes
    }
    fn squeeze_bits(&mut self, num_bits: usize) -> Vec<bool> {
        // sponge/poseidon/mod.rs:276-291
        let usable_bits = (Fr::MODULUS_BIT_SIZE - 1) as usize;
        let num_elements = (num_bits + usable_bits - 1) / usable_bits;
        let mut bits = Vec::with_capacity(usable_bits * num_elements);
        for elem in self.squeeze_native_field_elements(num_elements) {
            bits.extend_from_slice(&elem.into_bigint().to_bits_le()[..usable_bits]);
        }
        bits.truncate(num_bits);
        bits
    }
    // squeeze_field_elements_with_sizes / squeeze_field_elements / fork: the trait's default implementations
    // (sponge/mod.rs:116-153) are built on squeeze_bits / absorb and work unchanged.
}
impl FieldBasedCryptographicSponge<Fr> for GpuPoseidonSponge {
    fn squeeze_native_field_elements(&mut self, num_elements: usize) -> Vec<Fr> {
        // a squeeze of zero elements in absorbing mode still permutes (sponge/poseidon/mod.rs:331-334): the library does too
        let mut out = vec![Fr::zero(); num_elements];
        check(unsafe { ffi::akp_sponge_squeeze(self.h, words_mut(&mut out), num_elements) }, num_elements).expect("akp_sponge_squeeze");
        out
    }
    // squeeze_native_field_elements_with_sizes: the trait's default (sponge/mod.rs:162-178) -- full sizes go through
    // squeeze_native_field_elements above, truncated sizes through squeeze_bits.
}
impl SpongeExt for GpuPoseidonSponge {
    type State = GpuPoseidonSpongeState;

    fn from_state(state: Self::State, params: &Self::Config) -> Self {
        let s = Self::new(params);
        let (mode, index) = match state.mode {
            DuplexSpongeMode::Absorbing { next_absorb_index } => (0, next_absorb_index as u32),
            DuplexSpongeMode::Squeezing { next_squeeze_index } => (1, next_squeeze_index as u32),
        };
        check(unsafe { ffi::akp_sponge_set_state(s.h, words(&state.state), mode, index) }, state.state.len()).expect("akp_sponge_set_state");
        s
    }
    fn into_state(self) -> Self::State {
        self.raw_state()
    }
}

#[allow(dead_code)]
pub(crate) fn fr_from_words(w: &[u64]) -> Fr {
    fr_from_limbs([w[0], w[1], w[2], w[3]])
}
