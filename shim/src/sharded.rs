//! `MerkleTree<P>` sharded over the GPUs of one node and RESIDENT there (`akp_multi_*`, `akp_multi_tree_*`).
//!
//! The reference's tree is one object that is built (`merkle_tree/mod.rs:411-422`) and then asked for its root (`:526-528`), for
//! proofs (`:572-579`) and updated (`:692-702`).  [`GpuShardedMerkleTree`] keeps that shape over several devices: device r of G
//! holds the sub-tree of leaves `[r n/G, (r+1) n/G)` in its own HBM, the top `2G - 1` nodes are replicated, the only exchange is
//! one all-gather of the sub-roots inside `libakp.so` (RCCL over xGMI).  Proofs come back as the reference's own `Path<P>`.
//!
//! Parameter handles: a device needs its own tables, so the handles are created on [`MultiGpu`]'s per-device contexts (not
//! through the per-thread caches of [`crate::runtime`]).  The library pins them for the tree (ABI version 3), so this module
//! destroys its references right after the build and the tree releases the tables when it is dropped.
use crate::merkle::{BoweHopwoodByteConfig, GpuConfig, PedersenByteConfig, PoseidonFieldConfig};
use crate::runtime::{check, flatten, layout_check, words};
use crate::{ffi, te, Error, Fr};
use ark_crypto_primitives::crh::pedersen::Window;
use ark_crypto_primitives::merkle_tree::{LeafParam, Path, TwoToOneParam};
use ark_crypto_primitives::sponge::poseidon::PoseidonConfig;
use ark_std::{marker::PhantomData, vec::Vec};

/// `akp_multi`: one context per device + the RCCL communicator over them (`ncclCommInitAll` inside the library).
/// `device_ids.len()` must be a power of two; ids must be distinct.
pub struct MultiGpu {
    h: *mut ffi::AkpMulti,
    n_dev: usize,
}
impl MultiGpu {
    pub fn new(device_ids: &[i32]) -> Result<Self, Error> {
        layout_check();
        let mut h = core::ptr::null_mut();
        check(unsafe { ffi::akp_multi_create(device_ids.as_ptr(), device_ids.len() as i32, &mut h) }, 0)?;
        Ok(Self { h, n_dev: device_ids.len() })
    }
    pub fn size(&self) -> usize {
        self.n_dev
    }
    fn ctx(&self, r: usize) -> *mut ffi::AkpCtx {
        unsafe { ffi::akp_multi_ctx(self.h, r as i32) }
    }
    /// milliseconds of the last build: [sub-trees, all-gather, top levels, copy-out, whole call] (`akp_multi_last_phases`)
    pub fn last_phases(&self) -> [f64; 5] {
        let mut ms = [0f64; 5];
        let _ = check(unsafe { ffi::akp_multi_last_phases(self.h, ms.as_mut_ptr()) }, 0);
        ms
    }
}
impl Drop for MultiGpu {
    fn drop(&mut self) {
        unsafe { ffi::akp_multi_destroy(self.h) }
    }
}

/// per-device parameter handles of one build; dropped right after it (the tree has pinned them)
enum DeviceParams {
    Poseidon(Vec<*mut ffi::AkpPoseidon>),
    Te(Vec<*mut ffi::AkpTeParams>),
}
impl Drop for DeviceParams {
    fn drop(&mut self) {
        match self {
            DeviceParams::Poseidon(v) => v.iter().for_each(|h| unsafe { ffi::akp_poseidon_params_destroy(*h) }),
            DeviceParams::Te(v) => v.iter().for_each(|h| unsafe { ffi::akp_te_params_destroy(*h) }),
        }
    }
}
fn poseidon_on_every_device(m: &MultiGpu, cfg: &PoseidonConfig<Fr>) -> Result<DeviceParams, Error> {
    let (ark, mds) = (flatten(&cfg.ark), flatten(&cfg.mds));
    let mut v = Vec::with_capacity(m.size());
    for r in 0..m.size() {
        let mut h = core::ptr::null_mut();
        let rc = unsafe {
            ffi::akp_poseidon_params_create(m.ctx(r), cfg.full_rounds as u32, cfg.partial_rounds as u32, cfg.alpha, cfg.rate as u32, cfg.capacity as u32,
                                            words(&ark), words(&mds), &mut h)
        };
        if let Err(e) = check(rc, 0) {
            drop(DeviceParams::Poseidon(v));
            return Err(e);
        }
        v.push(h);
    }
    Ok(DeviceParams::Poseidon(v))
}
fn te_on_every_device(m: &MultiGpu, kind: i32, window_size: usize, num_windows: usize, xy: &[Fr]) -> Result<DeviceParams, Error> {
    let mut v = Vec::with_capacity(m.size());
    for r in 0..m.size() {
        let mut h = core::ptr::null_mut();
        let rc = unsafe { ffi::akp_te_params_create(m.ctx(r), kind, window_size as u32, num_windows as u32, words(xy), &mut h) };
        if let Err(e) = check(rc, 0) {
            drop(DeviceParams::Te(v));
            return Err(e);
        }
        v.push(h);
    }
    Ok(DeviceParams::Te(v))
}

/// a Merkle `Config` whose tree can be sharded: how to create its parameter handles on every device and which build entry
/// point takes its leaves
pub trait ShardedGpuConfig: GpuConfig {
    fn build_sharded(m: &MultiGpu, leaf: &LeafParam<Self>, two: &TwoToOneParam<Self>, leaves: &[&Self::Leaf]) -> Result<*mut ffi::AkpMultiTree, Error>;
}
impl ShardedGpuConfig for PoseidonFieldConfig {
    fn build_sharded(m: &MultiGpu, leaf: &PoseidonConfig<Fr>, two: &PoseidonConfig<Fr>, leaves: &[&[Fr]]) -> Result<*mut ffi::AkpMultiTree, Error> {
        let (lp, tp) = (poseidon_on_every_device(m, leaf)?, poseidon_on_every_device(m, two)?);
        let (DeviceParams::Poseidon(l), DeviceParams::Poseidon(t)) = (&lp, &tp) else { unreachable!() };
        let (uniform, offs) = crate::merkle::leaf_offsets(leaves);
        let flat: Vec<Fr> = leaves.iter().flat_map(|x| x.iter().copied()).collect();
        let mut out = core::ptr::null_mut();
        match uniform {
            Some(k) => check(unsafe { ffi::akp_multi_tree_build_poseidon(m.h, l.as_ptr(), t.as_ptr(), words(&flat), leaves.len(), k, &mut out) }, k)?,
            None => check(unsafe { ffi::akp_multi_tree_build_poseidon_ragged(m.h, l.as_ptr(), t.as_ptr(), words(&flat), offs.as_ptr(), leaves.len(), &mut out) }, 0)?,
        }
        Ok(out) // lp / tp dropped here: the tree has pinned the handles
    }
}
fn te_build<P: ShardedGpuConfig<Leaf = [u8]>>(m: &MultiGpu, kind: i32, lw: (usize, usize, Vec<Fr>), tw: (usize, usize, Vec<Fr>), leaves: &[&[u8]])
                                               -> Result<*mut ffi::AkpMultiTree, Error> {
    let (lp, tp) = (te_on_every_device(m, kind, lw.0, lw.1, &lw.2)?, te_on_every_device(m, kind, tw.0, tw.1, &tw.2)?);
    let (DeviceParams::Te(l), DeviceParams::Te(t)) = (&lp, &tp) else { unreachable!() };
    let (uniform, offs) = crate::merkle::leaf_offsets(leaves);
    let buf: Vec<u8> = leaves.iter().flat_map(|x| x.iter().copied()).collect();
    let mut out = core::ptr::null_mut();
    match uniform {
        Some(len) => check(unsafe { ffi::akp_multi_tree_build_te(m.h, l.as_ptr(), t.as_ptr(), buf.as_ptr(), leaves.len(), len, &mut out) }, len)?,
        None => check(unsafe { ffi::akp_multi_tree_build_te_ragged(m.h, l.as_ptr(), t.as_ptr(), buf.as_ptr(), offs.as_ptr(), leaves.len(), &mut out) }, 0)?,
    }
    Ok(out)
}
impl<W: Window> ShardedGpuConfig for PedersenByteConfig<W> {
    fn build_sharded(m: &MultiGpu, leaf: &LeafParam<Self>, two: &TwoToOneParam<Self>, leaves: &[&[u8]]) -> Result<*mut ffi::AkpMultiTree, Error> {
        te_build::<Self>(m, ffi::AKP_TE_PEDERSEN, te::window_words(&leaf.generators), te::window_words(&two.generators), leaves)
    }
}
impl<W: Window> ShardedGpuConfig for BoweHopwoodByteConfig<W> {
    fn build_sharded(m: &MultiGpu, leaf: &LeafParam<Self>, two: &TwoToOneParam<Self>, leaves: &[&[u8]]) -> Result<*mut ffi::AkpMultiTree, Error> {
        te_build::<Self>(m, ffi::AKP_TE_BOWE_HOPWOOD, te::window_words(&leaf.generators), te::window_words(&two.generators), leaves)
    }
}

/// `MerkleTree<P>` over the devices of a [`MultiGpu`].  Borrowing the `MultiGpu` keeps the drop order right (tree first).
pub struct GpuShardedMerkleTree<'m, P: ShardedGpuConfig> {
    h: *mut ffi::AkpMultiTree,
    n_leaves: usize,
    height: usize,
    _m: &'m MultiGpu,
    _p: PhantomData<P>,
}
impl<P: ShardedGpuConfig> Drop for GpuShardedMerkleTree<'_, P> {
    fn drop(&mut self) {
        unsafe { ffi::akp_multi_tree_destroy(self.h) }
    }
}
impl<'m, P: ShardedGpuConfig> GpuShardedMerkleTree<'m, P> {
    /// `MerkleTree::new` (`:411-422`): leaves in global order; panics like the reference when their number is not a power of two > 1
    pub fn new<'a>(m: &'m MultiGpu, leaf_hash_param: &LeafParam<P>, two_to_one_hash_param: &TwoToOneParam<P>, leaves: impl IntoIterator<Item = &'a P::Leaf>) -> Result<Self, Error>
    where
        P::Leaf: 'a,
    {
        let leaves: Vec<&P::Leaf> = leaves.into_iter().collect();
        let h = P::build_sharded(m, leaf_hash_param, two_to_one_hash_param, &leaves)?;
        let (mut n, mut fe, mut height, mut g) = (0usize, 0u32, 0usize, 0i32);
        check(unsafe { ffi::akp_multi_tree_info(h, &mut n, &mut fe, &mut height, &mut g) }, 0)?;
        assert_eq!(fe as usize, P::FE_PER_DIGEST);
        Ok(Self { h, n_leaves: n, height, _m: m, _p: PhantomData })
    }
    /// `:526-528` -- read from the replicated top, no device access
    pub fn root(&self) -> P::InnerDigest {
        let mut w = vec![0u64; 4 * P::FE_PER_DIGEST];
        check(unsafe { ffi::akp_multi_tree_root(self.h, w.as_mut_ptr()) }, 0).expect("akp_multi_tree_root");
        P::inner_digest(&w)
    }
    /// `:531-533`
    pub fn height(&self) -> usize {
        self.height
    }
    /// `generate_proof` (`:572-579`) for many GLOBAL leaf indexes: each is served by the device that owns it
    pub fn generate_proofs(&self, indexes: &[usize]) -> Result<Vec<Path<P>>, Error> {
        let fe = 4 * P::FE_PER_DIGEST;
        let depth = self.height - 2;
        let idx: Vec<u64> = indexes.iter().map(|i| *i as u64).collect();
        let (mut sib, mut auth) = (vec![0u64; idx.len() * fe], vec![0u64; idx.len() * depth * fe]);
        check(unsafe { ffi::akp_multi_tree_gather_paths(self.h, idx.as_ptr(), idx.len(), sib.as_mut_ptr(), auth.as_mut_ptr()) }, 0)?;
        Ok(indexes
            .iter()
            .enumerate()
            .map(|(k, &i)| Path {
                leaf_sibling_hash: P::leaf_digest(&sib[k * fe..(k + 1) * fe]),
                auth_path: (0..depth).map(|j| P::inner_digest(&auth[(k * depth + j) * fe..(k * depth + j + 1) * fe])).collect(),
                leaf_index: i,
            })
            .collect())
    }
    pub fn generate_proof(&self, index: usize) -> Result<Path<P>, Error> {
        Ok(self.generate_proofs(&[index])?.pop().unwrap())
    }
    /// `update` (`:692-702`) for many leaves: equal to the reference's sequential updates in the given order
    pub fn update_batch(&mut self, updates: &[(usize, &P::Leaf)]) -> Result<(), Error> {
        assert!(updates.iter().all(|(i, _)| *i < self.n_leaves), "index out of range");
        let idx: Vec<u64> = updates.iter().map(|(i, _)| *i as u64).collect();
        let leaves: Vec<&P::Leaf> = updates.iter().map(|(_, l)| *l).collect();
        let (buf, leaf_len) = P::encode_leaves(&leaves);
        check(unsafe { ffi::akp_multi_tree_update_batch(self.h, idx.as_ptr(), buf.as_ptr() as *const _, idx.len(), leaf_len) }, leaf_len)
    }
}
