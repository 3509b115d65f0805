//! Reference-vector emitter: turns "parity unpinned" into "pinned" for the curve half of the hot path.
//!
//!     cd shim && cargo run --release --example emit_vectors
//!
//! reads  ../tests/golden/emitter_inputs.json   (generators + inputs, written by tests/golden/make_emitter_inputs.py)
//! writes ../tests/golden/reference_vectors.json (outputs of the REFERENCE crates on those inputs)
//!
//! Everything below calls `ark-crypto-primitives` itself -- never this crate's GPU backend and never the repository's
//! oracle: `pedersen::CRH` / `pedersen::TwoToOneCRH` (crh/pedersen/mod.rs:76-129,158-197, bit order :200-209),
//! `bowe_hopwood::CRH` / `TwoToOneCRH` (crh/bowe_hopwood/mod.rs:114-186,202-239), `poseidon::CRH` / `TwoToOneCRH`
//! (crh/poseidon/mod.rs:30-79), `PoseidonSponge` (sponge/poseidon/mod.rs:223-344), `find_poseidon_ark_and_mds`
//! (sponge/poseidon/traits.rs:105-146), `MerkleTree::{new, root, generate_proof, generate_multi_proof, update}`
//! (merkle_tree/mod.rs:411-523,572-625,692-702) with `ByteDigestConverter` / `IdentityDigestConverter` (:53-78), and
//! `CanonicalSerialize` (compressed and uncompressed) of digests, `Parameters`, `PoseidonConfig`, `Path`, `MultiPath`
//! (macros.rs:3-13).  tests/test_reference_vectors.py then compares the python oracle, the C oracle, serialize.py and
//! (with -m gpu) the GPU path against the file; while the file is absent that test reports "unpinned (emitter not run)".
//!
//! Not compiled in the build image (no Rust toolchain there); written against the same arkworks versions as the crate.
use ark_crypto_primitives::crh::{bowe_hopwood, pedersen, poseidon, CRHScheme, TwoToOneCRHScheme};
use ark_crypto_primitives::merkle_tree::{
    ByteDigestConverter, Config, IdentityDigestConverter, LeafParam, MerkleTree, MultiPath, Path, TwoToOneParam,
};
use ark_crypto_primitives::sponge::poseidon::{find_poseidon_ark_and_mds, PoseidonConfig, PoseidonSponge};
use ark_crypto_primitives::sponge::{CryptographicSponge, FieldBasedCryptographicSponge};
use ark_ed_on_bls12_381::{EdwardsAffine, EdwardsConfig, EdwardsProjective, Fq};
use ark_ff::PrimeField;
use ark_serialize::CanonicalSerialize;
use serde_json::Value;
use std::fmt::Write as _;
use std::str::FromStr;

type Fr = ark_bls12_381::Fr; // == Fq of Jubjub: the field every digest lives in

// ---- windows (merkle_tree/tests/mod.rs:13-17; SURVEY.md 8a note on 63 x 9) -------------------------------------------
#[derive(Clone)]
struct Window4x256;
impl pedersen::Window for Window4x256 {
    const WINDOW_SIZE: usize = 4;
    const NUM_WINDOWS: usize = 256;
}
#[derive(Clone)]
struct Window63x9;
impl pedersen::Window for Window63x9 {
    const WINDOW_SIZE: usize = 63;
    const NUM_WINDOWS: usize = 9;
}

type PedH = pedersen::CRH<EdwardsProjective, Window4x256>;
type PedT = pedersen::TwoToOneCRH<EdwardsProjective, Window4x256>;
type BhH = bowe_hopwood::CRH<EdwardsConfig, Window63x9>;
type BhT = bowe_hopwood::TwoToOneCRH<EdwardsConfig, Window63x9>;
type PosH = poseidon::CRH<Fr>;
type PosT = poseidon::TwoToOneCRH<Fr>;

struct PedersenTree; // JubJubMerkleTreeParams of merkle_tree/tests/mod.rs:24-33
impl Config for PedersenTree {
    type Leaf = [u8];
    type LeafDigest = EdwardsAffine;
    type LeafInnerDigestConverter = ByteDigestConverter<EdwardsAffine>;
    type InnerDigest = EdwardsAffine;
    type LeafHash = PedH;
    type TwoToOneHash = PedT;
}
struct BoweHopwoodTree; // BASELINE configs[4]: Bowe-Hopwood leaf + two-to-one hashes, Fq digests, ByteDigestConverter
impl Config for BoweHopwoodTree {
    type Leaf = [u8];
    type LeafDigest = Fq;
    type LeafInnerDigestConverter = ByteDigestConverter<Fq>;
    type InnerDigest = Fq;
    type LeafHash = BhH;
    type TwoToOneHash = BhT;
}
struct PoseidonTree; // FieldMTConfig of merkle_tree/tests/mod.rs:198-206
impl Config for PoseidonTree {
    type Leaf = [Fr];
    type LeafDigest = Fr;
    type LeafInnerDigestConverter = IdentityDigestConverter<Fr>;
    type InnerDigest = Fr;
    type LeafHash = PosH;
    type TwoToOneHash = PosT;
}

// ---- small helpers: hex, decimal field elements, hand-written JSON -----------------------------------------------------
fn hex(b: &[u8]) -> String {
    let mut s = String::with_capacity(2 * b.len());
    for x in b {
        write!(s, "{:02x}", x).unwrap();
    }
    s
}
fn unhex(s: &str) -> Vec<u8> {
    (0..s.len() / 2).map(|i| u8::from_str_radix(&s[2 * i..2 * i + 2], 16).expect("hex")).collect()
}
fn fe(s: &Value) -> Fr {
    Fr::from_str(s.as_str().expect("decimal string")).expect("canonical decimal field element")
}
fn dec(x: &Fr) -> String {
    format!("\"{}\"", x.into_bigint())
}
fn ser<T: CanonicalSerialize>(v: &T, compressed: bool) -> String {
    let mut b = Vec::new();
    if compressed {
        v.serialize_compressed(&mut b).unwrap();
    } else {
        v.serialize_uncompressed(&mut b).unwrap();
    }
    hex(&b)
}
fn list(items: impl IntoIterator<Item = String>) -> String {
    format!("[{}]", items.into_iter().collect::<Vec<_>>().join(","))
}
fn usizes(v: &[usize]) -> String {
    list(v.iter().map(|x| x.to_string()))
}
fn show_point(p: &EdwardsAffine) -> String {
    list([dec(&p.x), dec(&p.y)])
}
fn show_fq(x: &Fq) -> String {
    list([dec(x)])
}
fn strs<'a>(v: &'a Value) -> impl Iterator<Item = &'a str> + 'a {
    v.as_array().expect("array").iter().map(|x| x.as_str().expect("string"))
}

fn generators(sec: &Value) -> Vec<Vec<EdwardsProjective>> {
    sec["generators"]
        .as_array()
        .unwrap()
        .iter()
        .map(|row| {
            row.as_array()
                .unwrap()
                .iter()
                // EdwardsAffine::new asserts on-curve and prime-order-subgroup membership
                .map(|p| EdwardsProjective::from(EdwardsAffine::new(fe(&p[0]), fe(&p[1]))))
                .collect()
        })
        .collect()
}

// ---- a tree over any of the three configurations ------------------------------------------------------------------------
fn emit_path<P: Config, D>(p: &Path<P>, show: &dyn Fn(&D) -> String) -> String
where
    P: Config<LeafDigest = D, InnerDigest = D>,
{
    format!(
        "{{\"leaf_index\":{},\"leaf_sibling_hash\":{},\"auth_path\":{},\"uncompressed\":\"{}\",\"compressed\":\"{}\"}}",
        p.leaf_index,
        show(&p.leaf_sibling_hash),
        list(p.auth_path.iter().map(|d| show(d))),
        ser(p, false),
        ser(p, true)
    )
}
fn emit_multi<P: Config, D>(requested: &[usize], m: &MultiPath<P>, show: &dyn Fn(&D) -> String) -> String
where
    P: Config<LeafDigest = D, InnerDigest = D>,
{
    format!(
        "{{\"requested\":{},\"leaf_indexes\":{},\"auth_paths_prefix_lenghts\":{},\"auth_paths_suffixes\":{},\"leaf_siblings_hashes\":{},\"uncompressed\":\"{}\",\"compressed\":\"{}\"}}",
        usizes(requested),
        usizes(&m.leaf_indexes),
        usizes(&m.auth_paths_prefix_lenghts),
        list(m.auth_paths_suffixes.iter().map(|s| list(s.iter().map(|d| show(d))))),
        list(m.leaf_siblings_hashes.iter().map(|d| show(d))),
        ser(m, false),
        ser(m, true)
    )
}
/// MerkleTree::new over `leaves`, every single proof (verified here with the reference's Path::verify), one multi-proof
/// (verified with MultiPath::verify), then update(index, new_leaf) and the new root.
fn emit_tree<P, D, L>(
    leaf_p: &LeafParam<P>,
    two_p: &TwoToOneParam<P>,
    leaves: &[Vec<L>],
    requested: &[usize],
    upd_index: usize,
    upd_leaf: &[L],
    show: &dyn Fn(&D) -> String,
) -> String
where
    P: Config<Leaf = [L], LeafDigest = D, InnerDigest = D>,
    D: Clone,
    L: Clone + Send + Sync,
{
    let mut tree = MerkleTree::<P>::new(leaf_p, two_p, leaves.iter().map(|l| l.as_slice()).collect::<Vec<_>>()).expect("MerkleTree::new");
    let root = tree.root();
    let mut proofs = Vec::new();
    for (i, leaf) in leaves.iter().enumerate() {
        let p = tree.generate_proof(i).expect("generate_proof");
        assert!(p.verify(leaf_p, two_p, &root, leaf.as_slice()).expect("Path::verify"), "reference proof {} does not verify", i);
        proofs.push(emit_path::<P, D>(&p, show));
    }
    let mp = tree.generate_multi_proof(requested.iter().copied()).expect("generate_multi_proof");
    let ordered: Vec<&[L]> = mp.leaf_indexes.iter().map(|i| leaves[*i].as_slice()).collect();
    assert!(mp.verify(leaf_p, two_p, &root, ordered).expect("MultiPath::verify"), "reference multi-proof does not verify");
    let multi = emit_multi::<P, D>(requested, &mp, show);
    tree.update(upd_index, upd_leaf).expect("update");
    format!(
        "{{\"root\":{},\"height\":{},\"proofs\":{},\"multi_proof\":{},\"update\":{{\"index\":{},\"root_after\":{}}}}}",
        show(&root),
        tree.height(),
        list(proofs),
        multi,
        upd_index,
        show(&tree.root())
    )
}

// ---- Pedersen / Bowe-Hopwood sections -------------------------------------------------------------------------------------
fn emit_pedersen(sec: &Value) -> String {
    let params = pedersen::Parameters::<EdwardsProjective> { generators: generators(sec) };
    assert_eq!(params.generators.len(), 256);
    let msgs: Vec<Vec<u8>> = strs(&sec["messages"]).map(unhex).collect();
    let digests: Vec<EdwardsAffine> = msgs.iter().map(|m| PedH::evaluate(&params, m.as_slice()).expect("pedersen::CRH::evaluate")).collect();
    let crh = list(msgs.iter().zip(&digests).map(|(m, d)| {
        format!("{{\"msg\":\"{}\",\"digest\":{},\"uncompressed\":\"{}\",\"compressed\":\"{}\"}}", hex(m), show_point(d), ser(d, false), ser(d, true))
    }));
    let pairs = list(sec["pairs"].as_array().unwrap().iter().map(|p| {
        let (l, r) = (unhex(p[0].as_str().unwrap()), unhex(p[1].as_str().unwrap()));
        let d = <PedT as TwoToOneCRHScheme>::evaluate(&params, l.as_slice(), r.as_slice()).expect("pedersen::TwoToOneCRH::evaluate");
        format!("{{\"left\":\"{}\",\"right\":\"{}\",\"digest\":{}}}", hex(&l), hex(&r), show_point(&d))
    }));
    let compress = list(digests.windows(2).map(|w| {
        let d = <PedT as TwoToOneCRHScheme>::compress(&params, &w[0], &w[1]).expect("pedersen::TwoToOneCRH::compress");
        format!("{{\"left\":{},\"right\":{},\"digest\":{}}}", show_point(&w[0]), show_point(&w[1]), show_point(&d))
    }));
    let head_n = sec["parameters_head_windows"].as_u64().unwrap() as usize;
    let head = pedersen::Parameters::<EdwardsProjective> { generators: params.generators[..head_n].to_vec() };
    let leaves: Vec<Vec<u8>> = strs(&sec["tree_leaves"]).map(unhex).collect();
    let req: Vec<usize> = sec["multi_proof_indexes"].as_array().unwrap().iter().map(|x| x.as_u64().unwrap() as usize).collect();
    let upd = unhex(sec["update"]["new_leaf"].as_str().unwrap());
    let tree = emit_tree::<PedersenTree, EdwardsAffine, u8>(&params, &params, &leaves, &req, sec["update"]["index"].as_u64().unwrap() as usize, &upd, &show_point);
    format!(
        "{{\"crh\":{},\"two_to_one_evaluate\":{},\"two_to_one_compress\":{},\"parameters_head\":{{\"windows\":{},\"uncompressed\":\"{}\",\"compressed\":\"{}\"}},\"tree\":{}}}",
        crh, pairs, compress, head_n, ser(&head, false), ser(&head, true), tree
    )
}
fn emit_bowe_hopwood(sec: &Value) -> String {
    let params = bowe_hopwood::Parameters::<EdwardsConfig> { generators: generators(sec) };
    assert_eq!(params.generators.len(), 9);
    let msgs: Vec<Vec<u8>> = strs(&sec["messages"]).map(unhex).collect();
    let digests: Vec<Fq> = msgs.iter().map(|m| BhH::evaluate(&params, m.as_slice()).expect("bowe_hopwood::CRH::evaluate")).collect();
    let crh = list(msgs.iter().zip(&digests).map(|(m, d)| {
        format!("{{\"msg\":\"{}\",\"digest\":{},\"uncompressed\":\"{}\",\"compressed\":\"{}\"}}", hex(m), show_fq(d), ser(d, false), ser(d, true))
    }));
    let pairs = list(sec["pairs"].as_array().unwrap().iter().map(|p| {
        let (l, r) = (unhex(p[0].as_str().unwrap()), unhex(p[1].as_str().unwrap()));
        let d = <BhT as TwoToOneCRHScheme>::evaluate(&params, l.as_slice(), r.as_slice()).expect("bowe_hopwood::TwoToOneCRH::evaluate");
        format!("{{\"left\":\"{}\",\"right\":\"{}\",\"digest\":{}}}", hex(&l), hex(&r), show_fq(&d))
    }));
    let compress = list(digests.windows(2).map(|w| {
        let d = <BhT as TwoToOneCRHScheme>::compress(&params, &w[0], &w[1]).expect("bowe_hopwood::TwoToOneCRH::compress");
        format!("{{\"left\":{},\"right\":{},\"digest\":{}}}", show_fq(&w[0]), show_fq(&w[1]), show_fq(&d))
    }));
    let head_n = sec["parameters_head_windows"].as_u64().unwrap() as usize;
    let head = bowe_hopwood::Parameters::<EdwardsConfig> { generators: params.generators[..head_n].to_vec() };
    let leaves: Vec<Vec<u8>> = strs(&sec["tree_leaves"]).map(unhex).collect();
    let req: Vec<usize> = sec["multi_proof_indexes"].as_array().unwrap().iter().map(|x| x.as_u64().unwrap() as usize).collect();
    let upd = unhex(sec["update"]["new_leaf"].as_str().unwrap());
    let tree = emit_tree::<BoweHopwoodTree, Fq, u8>(&params, &params, &leaves, &req, sec["update"]["index"].as_u64().unwrap() as usize, &upd, &show_fq);
    format!(
        "{{\"crh\":{},\"two_to_one_evaluate\":{},\"two_to_one_compress\":{},\"parameters_head\":{{\"windows\":{},\"uncompressed\":\"{}\",\"compressed\":\"{}\"}},\"tree\":{}}}",
        crh, pairs, compress, head_n, ser(&head, false), ser(&head, true), tree
    )
}

// ---- Poseidon section (already pinned by the reference's own KATs; here as an end-to-end cross-check) ------------------------
fn emit_poseidon(sec: &Value) -> String {
    let u = |k: &str| sec[k].as_u64().unwrap();
    let rows = |k: &str| -> Vec<Vec<Fr>> { sec[k].as_array().unwrap().iter().map(|r| r.as_array().unwrap().iter().map(fe).collect()).collect() };
    let (ark, mds) = (rows("ark"), rows("mds"));
    // the reference's own generator on the same dimensions: must reproduce the constants the inputs file carries
    let (ref_ark, ref_mds) = find_poseidon_ark_and_mds::<Fr>(u("prime_bits"), u("rate") as usize, u("full_rounds"), u("partial_rounds"), u("skip_matrices"));
    let generator_matches = ref_ark == ark && ref_mds == mds;
    let cfg = PoseidonConfig::<Fr>::new(u("full_rounds") as usize, u("partial_rounds") as usize, u("alpha"), mds, ark, u("rate") as usize, u("capacity") as usize);
    let crh = list(sec["crh_inputs"].as_array().unwrap().iter().map(|inp| {
        let x: Vec<Fr> = inp.as_array().unwrap().iter().map(fe).collect();
        let d = PosH::evaluate(&cfg, x.as_slice()).expect("poseidon::CRH::evaluate");
        format!("{{\"input\":{},\"digest\":{}}}", list(x.iter().map(dec)), show_fq(&d))
    }));
    let pairs = list(sec["pairs"].as_array().unwrap().iter().map(|p| {
        let (l, r) = (fe(&p[0]), fe(&p[1]));
        let e = <PosT as TwoToOneCRHScheme>::evaluate(&cfg, &l, &r).expect("poseidon::TwoToOneCRH::evaluate");
        let c = <PosT as TwoToOneCRHScheme>::compress(&cfg, &l, &r).expect("poseidon::TwoToOneCRH::compress");
        format!("{{\"left\":{},\"right\":{},\"evaluate\":{},\"compress\":{}}}", dec(&l), dec(&r), show_fq(&e), show_fq(&c))
    }));
    let sp = &sec["sponge"];
    let fes = |k: &str| -> Vec<Fr> { sp[k].as_array().unwrap().iter().map(fe).collect() };
    let mut sponge = PoseidonSponge::<Fr>::new(&cfg);
    sponge.absorb(&fes("absorb_1"));
    let s1 = sponge.squeeze_native_field_elements(sp["squeeze_1"].as_u64().unwrap() as usize);
    sponge.absorb(&fes("absorb_2"));
    let s2 = sponge.squeeze_native_field_elements(sp["squeeze_2"].as_u64().unwrap() as usize);
    let mut fork = sponge.clone();
    let bytes = fork.squeeze_bytes(sp["squeeze_bytes"].as_u64().unwrap() as usize);
    let bits = sponge.squeeze_bits(sp["squeeze_bits"].as_u64().unwrap() as usize);
    let leaves: Vec<Vec<Fr>> = sec["tree_leaves"].as_array().unwrap().iter().map(|r| r.as_array().unwrap().iter().map(fe).collect()).collect();
    let req: Vec<usize> = sec["multi_proof_indexes"].as_array().unwrap().iter().map(|x| x.as_u64().unwrap() as usize).collect();
    let upd: Vec<Fr> = sec["update"]["new_leaf"].as_array().unwrap().iter().map(fe).collect();
    let tree = emit_tree::<PoseidonTree, Fr, Fr>(&cfg, &cfg, &leaves, &req, sec["update"]["index"].as_u64().unwrap() as usize, &upd, &show_fq);
    format!(
        "{{\"reference_generator_matches_inputs\":{},\"config_uncompressed\":\"{}\",\"config_compressed\":\"{}\",\"crh\":{},\"two_to_one\":{},\"sponge\":{{\"squeeze_1\":{},\"squeeze_2\":{},\"squeeze_bytes_after\":\"{}\",\"squeeze_bits_after\":{}}},\"tree\":{}}}",
        generator_matches,
        ser(&cfg, false),
        ser(&cfg, true),
        crh,
        pairs,
        list(s1.iter().map(dec)),
        list(s2.iter().map(dec)),
        hex(&bytes),
        list(bits.iter().map(|b| (if *b { "1" } else { "0" }).to_string())),
        tree
    )
}

fn fnv1a64_hex(bytes: &[u8]) -> String {
    let mut h = 0xcbf2_9ce4_8422_2325u64;
    for b in bytes {
        h = (h ^ *b as u64).wrapping_mul(0x0000_0100_0000_01b3);
    }
    format!("{:016x}", h)
}

fn main() {
    let dir = std::path::Path::new(env!("CARGO_MANIFEST_DIR")).join("..").join("tests").join("golden");
    let inputs: Value = serde_json::from_reader(std::fs::File::open(dir.join("emitter_inputs.json")).expect("tests/golden/emitter_inputs.json")).expect("JSON");
    // provenance: which sources produced the file -- the emitter's git commit and a digest of Cargo.lock (the reference's arkworks
    // dependencies are git patches with no pinned revision: Cargo.lock is what names the `algebra` HEAD that was compiled)
    let manifest = std::path::Path::new(env!("CARGO_MANIFEST_DIR"));
    let git_sha = std::process::Command::new("git").arg("-C").arg(manifest).args(["rev-parse", "HEAD"]).output().ok()
        .and_then(|o| String::from_utf8(o.stdout).ok()).map(|s| s.trim().to_string()).filter(|s| !s.is_empty()).unwrap_or_else(|| "unknown".into());
    let lock_digest = std::fs::read(manifest.join("Cargo.lock")).map(|b| fnv1a64_hex(&b)).unwrap_or_else(|_| "no Cargo.lock".into());
    let lock_sha256 = std::process::Command::new("sha256sum").arg(manifest.join("Cargo.lock")).output().ok()
        .and_then(|o| String::from_utf8(o.stdout).ok()).and_then(|s| s.split_whitespace().next().map(|x| x.to_string())).unwrap_or_else(|| "sha256sum unavailable".into());
    let out = format!(
        "{{\"source\":\"outputs of the reference crates (ark-crypto-primitives + arkworks algebra) on tests/golden/emitter_inputs.json, written by shim/examples/emit_vectors.rs\",\n\"emitter\":{{\"git_sha\":\"{}\",\"cargo_lock_sha256\":\"{}\",\"cargo_lock_fnv1a64\":\"{}\"}},\n\"pedersen\":{},\n\"bowe_hopwood\":{},\n\"poseidon\":{}}}\n",
        git_sha, lock_sha256, lock_digest,
        emit_pedersen(&inputs["pedersen"]),
        emit_bowe_hopwood(&inputs["bowe_hopwood"]),
        emit_poseidon(&inputs["poseidon"])
    );
    let path = dir.join("reference_vectors.json");
    std::fs::write(&path, out).expect("write reference_vectors.json");
    println!("wrote {}", path.display());
}
