//! `MerkleTree<P>` (`merkle_tree/mod.rs:383-726`) resident in HBM.
//!
//! `GpuMerkleTree::new` is the batch boundary: one leaf-hash launch and one two-to-one launch per level
//! (`akp_merkle_tree_build_*`), nodes kept on the device in the reference's heap order.  Proofs are gathered from
//! there and returned as the reference's own `Path<P>` / `MultiPath<P>` (their fields are public), so verification
//! code is unchanged.  `into_reference_vectors` copies out `leaf_nodes` / `non_leaf_nodes` for code that wants
//! the reference's in-memory tree.
//!
//! Which hashes a `Config` uses on the device is described by [`GpuConfig`]; the three configurations of the
//! reference's own tests are provided: Poseidon field tree (`merkle_tree/tests/mod.rs:198-206`), Pedersen and
//! Bowe-Hopwood byte trees with `ByteDigestConverter` (`merkle_tree/tests/mod.rs:13-33`).
use crate::runtime::{check, fr_from_limbs, words};
use crate::{ffi, poseidon, te, Error, Fr};
use ark_crypto_primitives::crh::pedersen::Window;
use ark_crypto_primitives::merkle_tree::{Config, LeafParam, MultiPath, Path, TwoToOneParam};
use ark_ed_on_bls12_381::EdwardsAffine;
use ark_std::{collections::BTreeSet, marker::PhantomData, vec::Vec};

/// How a Merkle `Config` maps onto the library: leaf encoding, digest decoding, parameter handles.
pub trait GpuConfig: Config {
    /// Fr per digest on the wire: 1 (Poseidon, Bowe-Hopwood) or 2 (Pedersen affine point)
    const FE_PER_DIGEST: usize;
    /// build the device tree over `leaves`, each hashed with its own length (the `_ragged` entry points when the lengths differ)
    fn build(leaf: &LeafParam<Self>, two: &TwoToOneParam<Self>, leaves: &[&Self::Leaf]) -> Result<*mut ffi::AkpMerkleTree, Error>;
    fn from_digests(leaf: &LeafParam<Self>, two: &TwoToOneParam<Self>, digests: &[Self::LeafDigest]) -> Result<*mut ffi::AkpMerkleTree, Error>;
    /// flat encoding of leaves for `update` / `check_update`: (buffer, leaf_len in the library's unit)
    fn encode_leaves(leaves: &[&Self::Leaf]) -> (Vec<u8>, usize);
    fn leaf_digest(w: &[u64]) -> Self::LeafDigest;
    fn inner_digest(w: &[u64]) -> Self::InnerDigest;
    fn inner_words(d: &Self::InnerDigest) -> Vec<u64>;
}

fn fr_of(w: &[u64]) -> Fr {
    fr_from_limbs([w[0], w[1], w[2], w[3]])
}
/// `update_batch` hashes its new leaves in one uniform launch (the reference's `update` takes ONE leaf: any batch of one is uniform)
fn same_len<T: AsRef<[U]> + ?Sized, U>(leaves: &[&T]) -> usize {
    let l = leaves.first().map_or(0, |x| x.as_ref().len());
    assert!(leaves.iter().all(|x| x.as_ref().len() == l), "update_batch takes new leaves of equal length (call it once per length)");
    l
}
/// Leaves as the reference takes them -- each with its own length (`MerkleTree::new` maps `LeafHash::evaluate` over the iterator,
/// `merkle_tree/mod.rs:411-422`): `Some(len)` when they all agree (the uniform entry points), else the `n + 1` offsets of the
/// `_ragged` entry points (in elements of the leaf type).
pub(crate) fn leaf_offsets<T: AsRef<[U]> + ?Sized, U>(leaves: &[&T]) -> (Option<usize>, Vec<u64>) {
    let l = leaves.first().map_or(0, |x| x.as_ref().len());
    if leaves.iter().all(|x| x.as_ref().len() == l) {
        return (Some(l), Vec::new());
    }
    let mut offs = Vec::with_capacity(leaves.len() + 1);
    let mut at = 0u64;
    offs.push(at);
    for x in leaves {
        at += x.as_ref().len() as u64;
        offs.push(at);
    }
    (None, offs)
}

/// Poseidon field tree: `Leaf = [Fr]`, digests `Fr`, `IdentityDigestConverter`
pub struct PoseidonFieldConfig;
impl Config for PoseidonFieldConfig {
    type Leaf = [Fr];
    type LeafDigest = Fr;
    type LeafInnerDigestConverter = ark_crypto_primitives::merkle_tree::IdentityDigestConverter<Fr>;
    type InnerDigest = Fr;
    type LeafHash = poseidon::CRH;
    type TwoToOneHash = poseidon::TwoToOneCRH;
}
impl GpuConfig for PoseidonFieldConfig {
    const FE_PER_DIGEST: usize = 1;
    fn build(leaf: &LeafParam<Self>, two: &TwoToOneParam<Self>, leaves: &[&[Fr]]) -> Result<*mut ffi::AkpMerkleTree, Error> {
        let (uniform, offs) = leaf_offsets(leaves);
        let flat: Vec<Fr> = leaves.iter().flat_map(|l| l.iter().copied()).collect();
        let mut t = core::ptr::null_mut();
        match uniform {
            Some(k) => check(unsafe { ffi::akp_merkle_tree_build_poseidon(poseidon::handle(leaf)?, poseidon::handle(two)?, words(&flat), leaves.len(), k, &mut t) }, k)?,
            None => check(unsafe { ffi::akp_merkle_tree_build_poseidon_ragged(poseidon::handle(leaf)?, poseidon::handle(two)?, words(&flat), offs.as_ptr(), leaves.len(), &mut t) }, 0)?,
        }
        Ok(t)
    }
    fn from_digests(leaf: &LeafParam<Self>, two: &TwoToOneParam<Self>, digests: &[Fr]) -> Result<*mut ffi::AkpMerkleTree, Error> {
        let mut t = core::ptr::null_mut();
        check(unsafe { ffi::akp_merkle_tree_from_digests_poseidon(poseidon::handle(leaf)?, poseidon::handle(two)?, words(digests), digests.len(), &mut t) }, 0)?;
        Ok(t)
    }
    fn encode_leaves(leaves: &[&[Fr]]) -> (Vec<u8>, usize) {
        let k = same_len(leaves);
        let mut out = Vec::with_capacity(leaves.len() * k * 32);
        for l in leaves {
            for e in l.iter() {
                for w in (e.0).0 {
                    out.extend_from_slice(&w.to_le_bytes()); // the wire format: Montgomery limbs, little-endian words
                }
            }
        }
        (out, k)
    }
    fn leaf_digest(w: &[u64]) -> Fr {
        fr_of(w)
    }
    fn inner_digest(w: &[u64]) -> Fr {
        fr_of(w)
    }
    fn inner_words(d: &Fr) -> Vec<u64> {
        (d.0).0.to_vec()
    }
}

/// Pedersen byte tree over Jubjub (`JubJubMerkleTreeParams`, `merkle_tree/tests/mod.rs:19-33`)
pub struct PedersenByteConfig<W: Window>(PhantomData<W>);
impl<W: Window> Config for PedersenByteConfig<W> {
    type Leaf = [u8];
    type LeafDigest = EdwardsAffine;
    type LeafInnerDigestConverter = ark_crypto_primitives::merkle_tree::ByteDigestConverter<EdwardsAffine>;
    type InnerDigest = EdwardsAffine;
    type LeafHash = te::PedersenCRH<W>;
    type TwoToOneHash = te::PedersenTwoToOneCRH<W>;
}
fn point_of(w: &[u64]) -> EdwardsAffine {
    EdwardsAffine::new_unchecked(fr_of(&w[0..4]), fr_of(&w[4..8]))
}
/// byte-leaf tree build: the uniform entry point when the leaves agree in length, `akp_merkle_tree_build_te_ragged` otherwise
fn te_tree_build(lh: *mut ffi::AkpTeParams, th: *mut ffi::AkpTeParams, leaves: &[&[u8]]) -> Result<*mut ffi::AkpMerkleTree, Error> {
    let (uniform, offs) = leaf_offsets(leaves);
    let flat: Vec<u8> = leaves.iter().flat_map(|x| x.iter().copied()).collect();
    let mut t = core::ptr::null_mut();
    match uniform {
        Some(l) => check(unsafe { ffi::akp_merkle_tree_build_te(lh, th, flat.as_ptr(), leaves.len(), l, &mut t) }, l)?,
        None => check(unsafe { ffi::akp_merkle_tree_build_te_ragged(lh, th, flat.as_ptr(), offs.as_ptr(), leaves.len(), &mut t) }, 0)?,
    }
    Ok(t)
}
fn bytes_flat(leaves: &[&[u8]]) -> (Vec<u8>, usize) {
    let l = same_len(leaves);
    (leaves.iter().flat_map(|x| x.iter().copied()).collect(), l)
}
impl<W: Window> GpuConfig for PedersenByteConfig<W> {
    const FE_PER_DIGEST: usize = 2;
    fn build(leaf: &LeafParam<Self>, two: &TwoToOneParam<Self>, leaves: &[&[u8]]) -> Result<*mut ffi::AkpMerkleTree, Error> {
        te_tree_build(te::pedersen_handle(leaf)?, te::pedersen_handle(two)?, leaves)
    }
    fn from_digests(leaf: &LeafParam<Self>, two: &TwoToOneParam<Self>, digests: &[EdwardsAffine]) -> Result<*mut ffi::AkpMerkleTree, Error> {
        let flat: Vec<Fr> = digests.iter().flat_map(|p| [p.x, p.y]).collect();
        let mut t = core::ptr::null_mut();
        check(unsafe { ffi::akp_merkle_tree_from_digests_te(te::pedersen_handle(leaf)?, te::pedersen_handle(two)?, words(&flat), digests.len(), &mut t) }, 0)?;
        Ok(t)
    }
    fn encode_leaves(leaves: &[&[u8]]) -> (Vec<u8>, usize) {
        bytes_flat(leaves)
    }
    fn leaf_digest(w: &[u64]) -> EdwardsAffine {
        point_of(w)
    }
    fn inner_digest(w: &[u64]) -> EdwardsAffine {
        point_of(w)
    }
    fn inner_words(d: &EdwardsAffine) -> Vec<u64> {
        [(d.x.0).0, (d.y.0).0].concat()
    }
}

/// Bowe-Hopwood byte tree over Jubjub (BASELINE config 5): digests are x coordinates
pub struct BoweHopwoodByteConfig<W: Window>(PhantomData<W>);
impl<W: Window> Config for BoweHopwoodByteConfig<W> {
    type Leaf = [u8];
    type LeafDigest = Fr;
    type LeafInnerDigestConverter = ark_crypto_primitives::merkle_tree::ByteDigestConverter<Fr>;
    type InnerDigest = Fr;
    type LeafHash = te::BoweHopwoodCRH<W>;
    type TwoToOneHash = te::BoweHopwoodTwoToOneCRH<W>;
}
impl<W: Window> GpuConfig for BoweHopwoodByteConfig<W> {
    const FE_PER_DIGEST: usize = 1;
    fn build(leaf: &LeafParam<Self>, two: &TwoToOneParam<Self>, leaves: &[&[u8]]) -> Result<*mut ffi::AkpMerkleTree, Error> {
        te_tree_build(te::bowe_hopwood_handle(leaf)?, te::bowe_hopwood_handle(two)?, leaves)
    }
    fn from_digests(leaf: &LeafParam<Self>, two: &TwoToOneParam<Self>, digests: &[Fr]) -> Result<*mut ffi::AkpMerkleTree, Error> {
        let mut t = core::ptr::null_mut();
        check(unsafe { ffi::akp_merkle_tree_from_digests_te(te::bowe_hopwood_handle(leaf)?, te::bowe_hopwood_handle(two)?, words(digests), digests.len(), &mut t) }, 0)?;
        Ok(t)
    }
    fn encode_leaves(leaves: &[&[u8]]) -> (Vec<u8>, usize) {
        bytes_flat(leaves)
    }
    fn leaf_digest(w: &[u64]) -> Fr {
        fr_of(w)
    }
    fn inner_digest(w: &[u64]) -> Fr {
        fr_of(w)
    }
    fn inner_words(d: &Fr) -> Vec<u64> {
        (d.0).0.to_vec()
    }
}

/// Pedersen byte tree with `TECompressor` digests (`JubJubMerkleTreeParams` of `merkle_tree/tests/constraints.rs`):
/// leaf hash `PedersenCRHCompressor`, two-to-one `PedersenTwoToOneCRHCompressor`, Fq digests, `ByteDigestConverter`
pub struct PedersenXByteConfig<W: Window>(PhantomData<W>);
impl<W: Window> Config for PedersenXByteConfig<W> {
    type Leaf = [u8];
    type LeafDigest = Fr;
    type LeafInnerDigestConverter = ark_crypto_primitives::merkle_tree::ByteDigestConverter<Fr>;
    type InnerDigest = Fr;
    type LeafHash = te::PedersenCRHCompressor<W>;
    type TwoToOneHash = te::PedersenTwoToOneCRHCompressor<W>;
}
impl<W: Window> GpuConfig for PedersenXByteConfig<W> {
    const FE_PER_DIGEST: usize = 1;
    fn build(leaf: &LeafParam<Self>, two: &TwoToOneParam<Self>, leaves: &[&[u8]]) -> Result<*mut ffi::AkpMerkleTree, Error> {
        te_tree_build(te::pedersen_x_handle(leaf)?, te::pedersen_x_handle(two)?, leaves)
    }
    fn from_digests(leaf: &LeafParam<Self>, two: &TwoToOneParam<Self>, digests: &[Fr]) -> Result<*mut ffi::AkpMerkleTree, Error> {
        let mut t = core::ptr::null_mut();
        check(unsafe { ffi::akp_merkle_tree_from_digests_te(te::pedersen_x_handle(leaf)?, te::pedersen_x_handle(two)?, words(digests), digests.len(), &mut t) }, 0)?;
        Ok(t)
    }
    fn encode_leaves(leaves: &[&[u8]]) -> (Vec<u8>, usize) {
        bytes_flat(leaves)
    }
    fn leaf_digest(w: &[u64]) -> Fr {
        fr_of(w)
    }
    fn inner_digest(w: &[u64]) -> Fr {
        fr_of(w)
    }
    fn inner_words(d: &Fr) -> Vec<u64> {
        (d.0).0.to_vec()
    }
}

/// `MerkleTree<P>` with `leaf_nodes` / `non_leaf_nodes` in device memory.
///
/// Not `Sync`: the handle belongs to the context of the thread that built it (see [`crate::runtime`]).
pub struct GpuMerkleTree<P: GpuConfig> {
    h: *mut ffi::AkpMerkleTree,
    n_leaves: usize,
    height: usize,
    _p: PhantomData<P>,
}
impl<P: GpuConfig> Drop for GpuMerkleTree<P> {
    fn drop(&mut self) {
        unsafe { ffi::akp_merkle_tree_destroy(self.h) }
    }
}
impl<P: GpuConfig> GpuMerkleTree<P> {
    fn wrap(h: *mut ffi::AkpMerkleTree) -> Result<Self, Error> {
        let (mut n, mut fe, mut height) = (0usize, 0u32, 0usize);
        check(unsafe { ffi::akp_merkle_tree_info(h, &mut n, &mut fe, &mut height) }, 0)?;
        assert_eq!(fe as usize, P::FE_PER_DIGEST);
        Ok(Self { h, n_leaves: n, height, _p: PhantomData })
    }
    /// `MerkleTree::new` (`:411-422`).  Panics like the reference when `leaves.len()` is not a power of two > 1.
    pub fn new<'a>(leaf_hash_param: &LeafParam<P>, two_to_one_hash_param: &TwoToOneParam<P>, leaves: impl IntoIterator<Item = &'a P::Leaf>) -> Result<Self, Error>
    where
        P::Leaf: 'a,
    {
        let leaves: Vec<&P::Leaf> = leaves.into_iter().collect();
        Self::wrap(P::build(leaf_hash_param, two_to_one_hash_param, &leaves)?)
    }
    /// `MerkleTree::new_with_leaf_digest` (`:424-523`)
    pub fn new_with_leaf_digest(leaf_hash_param: &LeafParam<P>, two_to_one_hash_param: &TwoToOneParam<P>, leaf_digests: Vec<P::LeafDigest>) -> Result<Self, Error> {
        Self::wrap(P::from_digests(leaf_hash_param, two_to_one_hash_param, &leaf_digests)?)
    }
    /// `MerkleTree::blank` (`:400-408`)
    pub fn blank(leaf_hash_param: &LeafParam<P>, two_to_one_hash_param: &TwoToOneParam<P>, height: usize) -> Result<Self, Error> {
        Self::new_with_leaf_digest(leaf_hash_param, two_to_one_hash_param, vec![P::LeafDigest::default(); 1 << (height - 1)])
    }
    /// `:526-528`
    pub fn root(&self) -> P::InnerDigest {
        let mut w = vec![0u64; 4 * P::FE_PER_DIGEST];
        check(unsafe { ffi::akp_merkle_tree_root(self.h, w.as_mut_ptr()) }, 0).expect("akp_merkle_tree_root");
        P::inner_digest(&w)
    }
    /// `:531-533`
    pub fn height(&self) -> usize {
        self.height
    }
    /// the reference's two vectors, `(leaf_nodes, non_leaf_nodes)` (`:384-388`)
    pub fn into_reference_vectors(&self) -> (Vec<P::LeafDigest>, Vec<P::InnerDigest>) {
        let fe = 4 * P::FE_PER_DIGEST;
        let (mut ln, mut nl) = (vec![0u64; self.n_leaves * fe], vec![0u64; (self.n_leaves - 1) * fe]);
        check(unsafe { ffi::akp_merkle_tree_export(self.h, ln.as_mut_ptr(), nl.as_mut_ptr()) }, 0).expect("akp_merkle_tree_export");
        (ln.chunks_exact(fe).map(P::leaf_digest).collect(), nl.chunks_exact(fe).map(P::inner_digest).collect())
    }
    /// `generate_proof` for many leaves in one device gather (`:536-579`)
    pub fn generate_proofs(&self, indexes: &[usize]) -> Result<Vec<Path<P>>, Error> {
        let fe = 4 * P::FE_PER_DIGEST;
        let depth = self.height - 2;
        let idx: Vec<u64> = indexes.iter().map(|i| *i as u64).collect();
        let (mut sib, mut auth) = (vec![0u64; idx.len() * fe], vec![0u64; idx.len() * depth * fe]);
        check(unsafe { ffi::akp_merkle_tree_gather_paths(self.h, idx.as_ptr(), idx.len(), sib.as_mut_ptr(), auth.as_mut_ptr()) }, 0)?;
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
    /// `:572-579`
    pub fn generate_proof(&self, index: usize) -> Result<Path<P>, Error> {
        Ok(self.generate_proofs(&[index])?.pop().unwrap())
    }
    /// `:592-625`: sorted, de-duplicated indexes, front-coded paths (`prefix_encode_path`, `:795-805`, done by the library)
    pub fn generate_multi_proof(&self, indexes: impl IntoIterator<Item = usize>) -> Result<MultiPath<P>, Error> {
        let sorted: BTreeSet<usize> = indexes.into_iter().collect();
        let leaf_indexes: Vec<usize> = sorted.into_iter().collect();
        let fe = 4 * P::FE_PER_DIGEST;
        let (m, depth) = (leaf_indexes.len(), self.height - 2);
        let idx: Vec<u64> = leaf_indexes.iter().map(|i| *i as u64).collect();
        // gather + prefix_encode_path (`:795-805`) + suffix compaction on the device: only the suffixes cross PCIe (round 5)
        let mut sib = vec![0u64; m * fe];
        let (mut pre, mut suf, mut cnt) = (vec![0u64; m], vec![0u64; m * depth * fe], 0usize);
        check(unsafe { ffi::akp_merkle_tree_multi_proof(self.h, idx.as_ptr(), m, sib.as_mut_ptr(), pre.as_mut_ptr(), suf.as_mut_ptr(), m * depth, &mut cnt) }, 0)?;
        let mut off = 0;
        let mut auth_paths_suffixes = Vec::with_capacity(m);
        for k in 0..m {
            let len = depth - pre[k] as usize;
            auth_paths_suffixes.push((0..len).map(|j| P::inner_digest(&suf[(off + j) * fe..(off + j + 1) * fe])).collect());
            off += len;
        }
        Ok(MultiPath {
            leaf_siblings_hashes: (0..m).map(|k| P::leaf_digest(&sib[k * fe..(k + 1) * fe])).collect(),
            auth_paths_prefix_lenghts: pre.iter().map(|p| *p as usize).collect(),
            auth_paths_suffixes,
            leaf_indexes,
        })
    }
    /// `update` (`:692-702`) for many leaves at once: same result as updating one by one in order (a repeated index keeps
    /// its last leaf), one hash launch per level.  Panics like the reference on an index out of range.
    pub fn update_batch(&mut self, updates: &[(usize, &P::Leaf)]) -> Result<(), Error> {
        assert!(updates.iter().all(|(i, _)| *i < self.n_leaves), "index out of range");
        let idx: Vec<u64> = updates.iter().map(|(i, _)| *i as u64).collect();
        let leaves: Vec<&P::Leaf> = updates.iter().map(|(_, l)| *l).collect();
        let (buf, leaf_len) = P::encode_leaves(&leaves);
        check(unsafe { ffi::akp_merkle_tree_update_batch(self.h, idx.as_ptr(), buf.as_ptr() as *const _, idx.len(), leaf_len) }, leaf_len)
    }
    /// `:692-702`
    pub fn update(&mut self, index: usize, new_leaf: &P::Leaf) -> Result<(), Error> {
        self.update_batch(&[(index, new_leaf)])
    }
    /// `:707-725`: the tree is modified only when the new root equals `asserted_new_root`
    pub fn check_update(&mut self, index: usize, new_leaf: &P::Leaf, asserted_new_root: &P::InnerDigest) -> Result<bool, Error> {
        assert!(index < self.n_leaves, "index out of range");
        let (buf, leaf_len) = P::encode_leaves(&[new_leaf]);
        let root = P::inner_words(asserted_new_root);
        let mut ok = 0i32;
        check(unsafe { ffi::akp_merkle_tree_check_update(self.h, index as u64, buf.as_ptr() as *const _, leaf_len, root.as_ptr(), &mut ok) }, leaf_len)?;
        Ok(ok == 1)
    }
}

/// Batched `Path::verify` (`:172-212`) for the Poseidon field tree: all paths advance one level per launch.
pub fn verify_paths_poseidon(leaf_hash_param: &LeafParam<PoseidonFieldConfig>, two_to_one_hash_param: &TwoToOneParam<PoseidonFieldConfig>, root: &Fr,
                             paths: &[Path<PoseidonFieldConfig>], leaves: &[&[Fr]]) -> Result<Vec<bool>, Error> {
    assert_eq!(paths.len(), leaves.len());
    let m = paths.len();
    if m == 0 {
        return Ok(Vec::new());
    }
    let depth = paths[0].auth_path.len();
    assert!(paths.iter().all(|p| p.auth_path.len() == depth), "paths of one tree have one depth");
    let k = same_len(leaves);
    let flat: Vec<Fr> = leaves.iter().flat_map(|l| l.iter().copied()).collect();
    let idx: Vec<u64> = paths.iter().map(|p| p.leaf_index as u64).collect();
    let sib: Vec<Fr> = paths.iter().map(|p| p.leaf_sibling_hash).collect();
    let auth: Vec<Fr> = paths.iter().flat_map(|p| p.auth_path.iter().copied()).collect();
    let mut ok = vec![0u8; m];
    check(
        unsafe {
            ffi::akp_merkle_verify_paths_poseidon(poseidon::handle(leaf_hash_param)?, poseidon::handle(two_to_one_hash_param)?, words(core::slice::from_ref(root)),
                                                  words(&flat), m, k, idx.as_ptr(), words(&sib), words(&auth), depth, ok.as_mut_ptr())
        },
        k,
    )?;
    Ok(ok.into_iter().map(|b| b == 1).collect())
}
