// akp.hpp -- header-only C++17 mirror of the reference's operator surface over the C ABI (akp.h).
//
// The reference is Rust; its host-side API for this path is three traits with static functions.  This header
// gives a C++ caller the same shapes (same names, argument meaning and error behaviour):
//   CRHScheme            (crh/mod.rs:18-28)   -> struct with static setup / evaluate           (+ evaluate_batch)
//   TwoToOneCRHScheme    (crh/mod.rs:31-51)   -> static evaluate / compress                      (+ *_batch)
//   CryptographicSponge  (sponge/mod.rs:101-179) -> PoseidonSponge: absorb / squeeze_native_field_elements
//   MerkleTree<P>        (merkle_tree/mod.rs:383-726) -> MerkleTree<Config>: new_ / root / height / generate_proof,
//                                                  Path<Config>::verify
// Errors: the reference returns Err(Error::...) or panics; here every failure throws akp::Error carrying the
// ABI status (1 = IncorrectInputLength / length panic, 5 = leaf count not a power of two, 3 = no device...).
// There is no CPU fallback behind any of these calls.
#pragma once
#include <array>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "akp.h"

namespace akp {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error("akp error " + std::to_string(c) + ": " + m), code(c) {}
};
inline void check(int32_t rc) {
    if (rc != AKP_OK) throw Error(rc, akp_last_error());
}

// ark-ff Fp256 memory image: 4 x u64 LE limbs, Montgomery form
using FrWire = std::array<uint64_t, 4>;
inline std::vector<FrWire> fr_from_canonical(const std::vector<FrWire>& c) {
    std::vector<FrWire> m(c.size());
    check(akp_fr_to_mont(c.empty() ? nullptr : c[0].data(), m.empty() ? nullptr : m[0].data(), c.size()));
    return m;
}
inline FrWire fr_from_u64(uint64_t v) { return fr_from_canonical({FrWire{v, 0, 0, 0}})[0]; }
inline std::vector<FrWire> fr_to_canonical(const std::vector<FrWire>& m) {
    std::vector<FrWire> c(m.size());
    check(akp_fr_from_mont(m.empty() ? nullptr : m[0].data(), c.empty() ? nullptr : c[0].data(), m.size()));
    return c;
}

class Context {
  public:
    explicit Context(int device_id = 0) { check(akp_ctx_create(device_id, &h_)); }
    ~Context() { akp_ctx_destroy(h_); }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    akp_ctx* get() const { return h_; }
    // HBM one precomputed Pedersen / Bowe-Hopwood table may take on this device (akp_ctx_set_table_budget; 0 = the default: 320 MiB,
    // cache-sized tables; AKP_TABLE_BUDGET_DEVICE = a quarter of the device's memory, at most half of what is free)
    void set_table_budget(size_t bytes) { check(akp_ctx_set_table_budget(h_, bytes)); }
    size_t table_budget() const { return akp_ctx_table_budget(h_); }

  private:
    akp_ctx* h_ = nullptr;
};

// ---- PoseidonConfig<Fr> (sponge/poseidon/mod.rs:27-45) -----------------------------------------------------
class PoseidonConfig {
  public:
    // PoseidonConfig::new (:191-217); ark is [full+partial][t], mds is [t][t], flattened
    PoseidonConfig(const Context& ctx, uint32_t full_rounds, uint32_t partial_rounds, uint64_t alpha, const std::vector<FrWire>& mds,
                   const std::vector<FrWire>& ark, uint32_t rate, uint32_t capacity) {
        check(akp_poseidon_params_create(ctx.get(), full_rounds, partial_rounds, alpha, rate, capacity, ark[0].data(), mds[0].data(), &h_));
        load_dims();
    }
    // Fr::get_default_poseidon_parameters(rate, optimized_for_weights) (traits.rs:148-155); throws where the
    // reference returns None
    static PoseidonConfig get_default_poseidon_parameters(const Context& ctx, uint32_t rate, bool optimized_for_weights) {
        akp_poseidon* h = nullptr;
        check(akp_poseidon_default_params(ctx.get(), rate, optimized_for_weights ? 1 : 0, &h));
        return PoseidonConfig(h);
    }
    // CanonicalDeserialize (ark-serialize bytes of sponge/poseidon/mod.rs:26-45); ctx == nullptr: a host-only handle (inspection,
    // re-serialisation; compute calls on it fail with AKP_ERR_HIP)
    static PoseidonConfig deserialize(const Context* ctx, const std::vector<uint8_t>& bytes) {
        akp_poseidon* h = nullptr;
        check(akp_deserialize_poseidon_config(ctx ? ctx->get() : nullptr, bytes.data(), bytes.size(), &h));
        return PoseidonConfig(h);
    }
    ~PoseidonConfig() { akp_poseidon_params_destroy(h_); }
    PoseidonConfig(PoseidonConfig&& o) noexcept { *this = std::move(o); }
    PoseidonConfig& operator=(PoseidonConfig&& o) noexcept {
        std::swap(h_, o.h_);
        full_rounds = o.full_rounds; partial_rounds = o.partial_rounds; alpha = o.alpha; rate = o.rate; capacity = o.capacity;
        return *this;
    }
    PoseidonConfig(const PoseidonConfig&) = delete;
    akp_poseidon* get() const { return h_; }
    uint32_t full_rounds = 0, partial_rounds = 0, rate = 0, capacity = 0;
    uint64_t alpha = 0;
    std::vector<FrWire> ark() const { std::vector<FrWire> a((size_t)(full_rounds + partial_rounds) * (rate + capacity)); check(akp_poseidon_params_export(h_, a[0].data(), nullptr)); return a; }
    std::vector<FrWire> mds() const { std::vector<FrWire> m((size_t)(rate + capacity) * (rate + capacity)); check(akp_poseidon_params_export(h_, nullptr, m[0].data())); return m; }

  private:
    explicit PoseidonConfig(akp_poseidon* h) : h_(h) { load_dims(); }
    void load_dims() { check(akp_poseidon_params_dims(h_, &full_rounds, &partial_rounds, &alpha, &rate, &capacity)); }
    akp_poseidon* h_ = nullptr;
};

// ---- PoseidonSponge<Fr> (batch of sponges sharing one schedule; sponge/poseidon/mod.rs:54-63,220-345) ------
class PoseidonSponge {
  public:
    // CryptographicSponge::new(&config)
    explicit PoseidonSponge(const PoseidonConfig& cfg, size_t batch = 1) : batch_(batch) { check(akp_sponge_create(cfg.get(), batch, &h_)); }
    ~PoseidonSponge() { akp_sponge_destroy(h_); }
    PoseidonSponge(const PoseidonSponge&) = delete;
    // absorb(&[Fr]) for each sponge of the batch: input is [batch][k]
    void absorb(const std::vector<FrWire>& input) { check(akp_sponge_absorb(h_, input.empty() ? nullptr : input[0].data(), input.size() / batch_)); }
    // FieldBasedCryptographicSponge::squeeze_native_field_elements(n): [batch][n]
    std::vector<FrWire> squeeze_native_field_elements(size_t n) {
        std::vector<FrWire> out(batch_ * n);
        check(akp_sponge_squeeze(h_, out.empty() ? nullptr : out[0].data(), n));
        return out;
    }
    // the same on device buffers ([batch][k] Fr in, [batch][n] Fr out), enqueued on `stream` (akp_sponge_{absorb,squeeze}_dev)
    void absorb_dev(const uint64_t* d_elems, size_t elems_per_instance, void* stream) { check(akp_sponge_absorb_dev(h_, d_elems, elems_per_instance, stream)); }
    void squeeze_native_field_elements_dev(uint64_t* d_out, size_t n, void* stream) { check(akp_sponge_squeeze_dev(h_, d_out, n, stream)); }

  private:
    akp_sponge* h_ = nullptr;
    size_t batch_;
};

namespace poseidon {
// poseidon::CRH<Fr> (crh/poseidon/mod.rs:15-41): Input = [Fr], Output = Fr, Parameters = PoseidonConfig<Fr>
struct CRH {
    using Parameters = PoseidonConfig;
    using Output = FrWire;
    [[noreturn]] static Parameters setup() { throw std::logic_error("not implemented in the reference either (crh/poseidon/mod.rs:24-28)"); }
    static std::vector<Output> evaluate_batch(const Parameters& p, const std::vector<FrWire>& inputs, size_t elems_per_input) {
        const size_t n = elems_per_input ? inputs.size() / elems_per_input : 1;
        std::vector<Output> out(n);
        check(akp_poseidon_crh_batch(p.get(), inputs.empty() ? nullptr : inputs[0].data(), n, elems_per_input, out[0].data()));
        return out;
    }
    static Output evaluate(const Parameters& p, const std::vector<FrWire>& input) { return evaluate_batch(p, input, input.size())[0]; }
    // inputs of DIFFERENT lengths in one launch, each hashed as evaluate() would hash it (akp_poseidon_crh_batch_ragged)
    static std::vector<Output> evaluate_many(const Parameters& p, const std::vector<std::vector<FrWire>>& inputs) {
        std::vector<uint64_t> offs(inputs.size() + 1, 0);
        std::vector<FrWire> flat;
        for (size_t i = 0; i < inputs.size(); ++i) {
            flat.insert(flat.end(), inputs[i].begin(), inputs[i].end());
            offs[i + 1] = flat.size();
        }
        std::vector<Output> out(inputs.size());
        if (!inputs.empty()) check(akp_poseidon_crh_batch_ragged(p.get(), flat.empty() ? nullptr : flat[0].data(), offs.data(), inputs.size(), out[0].data()));
        return out;
    }
};
// poseidon::TwoToOneCRH<Fr> (crh/poseidon/mod.rs:43-80)
struct TwoToOneCRH {
    using Parameters = PoseidonConfig;
    using Output = FrWire;
    static std::vector<Output> compress_batch(const Parameters& p, const std::vector<FrWire>& left, const std::vector<FrWire>& right) {
        if (left.size() != right.size()) throw Error(AKP_ERR_BAD_LENGTH, "left and right batches differ in length");
        std::vector<Output> out(left.size());
        if (!left.empty()) check(akp_poseidon_two_to_one_batch(p.get(), left[0].data(), right[0].data(), left.size(), out[0].data()));
        return out;
    }
    static Output compress(const Parameters& p, const FrWire& l, const FrWire& r) { return compress_batch(p, {l}, {r})[0]; }
    static Output evaluate(const Parameters& p, const FrWire& l, const FrWire& r) { return compress(p, l, r); }
};
}  // namespace poseidon

// ---- Pedersen / Bowe-Hopwood over Jubjub -------------------------------------------------------------------
struct AffineWire { FrWire x, y; };  // ark_ed_on_bls12_381::EdwardsAffine coordinates
template <int KIND>
class TeParameters {  // pedersen::Parameters / bowe_hopwood::Parameters { generators } as affine points [N][W]
  public:
    // table_shape: digit width (Pedersen, 2..24) / chunks per table step (Bowe-Hopwood, 1..8) of the device table; 0 = the widest
    // the context's table budget admits (akp_te_params_create_shaped).  The digests do not depend on it.
    TeParameters(const Context& ctx, uint32_t window_size, uint32_t num_windows, const std::vector<AffineWire>& generators, uint32_t table_shape = 0)
        : window_size(window_size), num_windows(num_windows) {
        if (generators.size() != (size_t)window_size * num_windows) throw Error(AKP_ERR_BAD_PARAMS, "Incorrect pp size for window params");
        check(akp_te_params_create_shaped(ctx.get(), KIND, window_size, num_windows, generators[0].x.data(), table_shape, &h_));
    }
    ~TeParameters() { akp_te_params_destroy(h_); }
    TeParameters(const TeParameters&) = delete;
    akp_te_params* get() const { return h_; }
    // tuning facts of the device tables (akp_te_params_info)
    struct Info {
        uint32_t digit_bits_or_group = 0;
        bool signed_subset = false;
        size_t table_bytes = 0;
        uint32_t steps = 0;
    };
    Info info(size_t msg_len = 0) const {
        Info i;
        int32_t sg = 0;
        check(akp_te_params_info(h_, &i.digit_bits_or_group, &sg, &i.table_bytes, msg_len, &i.steps));
        i.signed_subset = sg != 0;
        return i;
    }
    // build the device tables now instead of inside the first hash (akp_te_params_prepare / _prepare_compress)
    void prepare(size_t msg_len) const { check(akp_te_params_prepare(h_, msg_len)); }
    void prepare_compress() const { check(akp_te_params_prepare_compress(h_)); }
    // the shared per-device table behind this handle (akp_te_params_table_info): handles with equal id share it
    struct TableInfo {
        uint64_t table_id = 0;
        uint32_t handles_attached = 0;
        uint64_t wide_builds = 0;
        akp_te_build_report last_build{};  // phases of the last build of the wide table; upgrade_state (akp.h)
    };
    TableInfo table_info() const {
        TableInfo t;
        check(akp_te_params_table_info(h_, &t.table_id, &t.handles_attached, &t.wide_builds, &t.last_build));
        return t;
    }
    uint32_t window_size, num_windows;

  private:
    akp_te_params* h_ = nullptr;
};
// byte strings of different lengths -> flat buffer + offsets (the `_ragged` entry points)
inline void flatten_ragged(const std::vector<std::vector<uint8_t>>& msgs, std::vector<uint8_t>& flat, std::vector<uint64_t>& offs) {
    offs.assign(msgs.size() + 1, 0);
    flat.clear();
    for (size_t i = 0; i < msgs.size(); ++i) {
        flat.insert(flat.end(), msgs[i].begin(), msgs[i].end());
        offs[i + 1] = flat.size();
    }
}
namespace pedersen {
using Parameters = TeParameters<AKP_TE_PEDERSEN>;
struct CRH {  // crh/pedersen/mod.rs:58-130: Input = [u8], Output = affine point
    using Output = AffineWire;
    static std::vector<Output> evaluate_batch(const Parameters& p, const std::vector<uint8_t>& msgs, size_t msg_len) {
        const size_t n = msg_len ? msgs.size() / msg_len : 1;
        std::vector<Output> out(n);
        check(akp_te_crh_batch(p.get(), msgs.data(), n, msg_len, out[0].x.data()));
        return out;
    }
    static Output evaluate(const Parameters& p, const std::vector<uint8_t>& input) { return evaluate_batch(p, input, input.size())[0]; }
    // inputs of different lengths, each padded with zero bits to the window as evaluate() does (:82-99); one launch
    static std::vector<Output> evaluate_many(const Parameters& p, const std::vector<std::vector<uint8_t>>& msgs) {
        std::vector<uint8_t> flat;
        std::vector<uint64_t> offs;
        flatten_ragged(msgs, flat, offs);
        std::vector<Output> out(msgs.size());
        if (!msgs.empty()) check(akp_te_crh_batch_ragged(p.get(), flat.data(), offs.data(), msgs.size(), out[0].x.data()));
        return out;
    }
};
struct TwoToOneCRH {  // :149-198
    using Output = AffineWire;
    static Output evaluate(const Parameters& p, const std::vector<uint8_t>& l, const std::vector<uint8_t>& r) {
        if (l.size() != r.size()) throw Error(AKP_ERR_BAD_LENGTH, "left and right input should be of equal length");
        Output out;
        check(akp_te_two_to_one_batch(p.get(), l.data(), r.data(), 1, l.size(), out.x.data()));
        return out;
    }
    static Output compress(const Parameters& p, const Output& l, const Output& r) {
        Output out;
        check(akp_te_compress_batch(p.get(), l.x.data(), r.x.data(), 1, out.x.data()));
        return out;
    }
};
}  // namespace pedersen
namespace bowe_hopwood {
using Parameters = TeParameters<AKP_TE_BOWE_HOPWOOD>;
struct CRH {  // crh/bowe_hopwood/mod.rs:75-187: Output = Fq (x coordinate)
    using Output = FrWire;
    static std::vector<Output> evaluate_batch(const Parameters& p, const std::vector<uint8_t>& msgs, size_t msg_len) {
        const size_t n = msg_len ? msgs.size() / msg_len : 1;
        std::vector<Output> out(n);
        check(akp_te_crh_batch(p.get(), msgs.data(), n, msg_len, out[0].data()));
        return out;
    }
    static Output evaluate(const Parameters& p, const std::vector<uint8_t>& input) { return evaluate_batch(p, input, input.size())[0]; }
    // inputs of different lengths, each padded to a multiple of 3 bits only (:131-138): the digest depends on the length; one launch
    static std::vector<Output> evaluate_many(const Parameters& p, const std::vector<std::vector<uint8_t>>& msgs) {
        std::vector<uint8_t> flat;
        std::vector<uint64_t> offs;
        flatten_ragged(msgs, flat, offs);
        std::vector<Output> out(msgs.size());
        if (!msgs.empty()) check(akp_te_crh_batch_ragged(p.get(), flat.data(), offs.data(), msgs.size(), out[0].data()));
        return out;
    }
};
struct TwoToOneCRH {  // :189-240
    using Output = FrWire;
    static Output evaluate(const Parameters& p, const std::vector<uint8_t>& l, const std::vector<uint8_t>& r) {
        if (l.size() != r.size()) throw Error(AKP_ERR_BAD_LENGTH, "left and right input should be of equal length");
        Output out;
        check(akp_te_two_to_one_batch(p.get(), l.data(), r.data(), 1, l.size(), out.data()));
        return out;
    }
    static Output compress(const Parameters& p, const Output& l, const Output& r) {
        Output out;
        check(akp_te_compress_batch(p.get(), l.data(), r.data(), 1, out.data()));
        return out;
    }
};
}  // namespace bowe_hopwood
namespace injective_map {  // crh/injective_map/mod.rs:16-108 with TECompressor: digest = x coordinate of the Pedersen hash
using Parameters = TeParameters<AKP_TE_PEDERSEN_X>;
struct PedersenCRHCompressor {  // :33-62
    using Output = FrWire;
    static std::vector<Output> evaluate_batch(const Parameters& p, const std::vector<uint8_t>& msgs, size_t msg_len) {
        const size_t n = msg_len ? msgs.size() / msg_len : 1;
        std::vector<Output> out(n);
        check(akp_te_crh_batch(p.get(), msgs.data(), n, msg_len, out[0].data()));
        return out;
    }
    static Output evaluate(const Parameters& p, const std::vector<uint8_t>& input) { return evaluate_batch(p, input, input.size())[0]; }
};
struct PedersenTwoToOneCRHCompressor {  // :64-108
    using Output = FrWire;
    static Output evaluate(const Parameters& p, const std::vector<uint8_t>& l, const std::vector<uint8_t>& r) {
        if (l.size() != r.size()) throw Error(AKP_ERR_BAD_LENGTH, "left and right input should be of equal length");
        Output out;
        check(akp_te_two_to_one_batch(p.get(), l.data(), r.data(), 1, l.size(), out.data()));
        return out;
    }
    static Output compress(const Parameters& p, const Output& l, const Output& r) {
        Output out;
        check(akp_te_compress_batch(p.get(), l.data(), r.data(), 1, out.data()));
        return out;
    }
};
}  // namespace injective_map

// ---- MerkleTree (merkle_tree/mod.rs) -----------------------------------------------------------------------
// Config for Leaf = [Fr], poseidon CRH + TwoToOneCRH, IdentityDigestConverter (merkle_tree/tests/mod.rs:198-206)
struct PoseidonFieldConfig {
    using LeafParam = PoseidonConfig;
    using TwoToOneParam = PoseidonConfig;
    using Digest = FrWire;
};
template <class P>
struct Path {  // merkle_tree::Path (:146-213)
    typename P::Digest leaf_sibling_hash;
    std::vector<typename P::Digest> auth_path;  // root side first, root excluded
    size_t leaf_index = 0;
    // Path::verify (:172-212)
    bool verify(const typename P::LeafParam& leaf_params, const typename P::TwoToOneParam& two_params, const typename P::Digest& root,
                const std::vector<FrWire>& leaf) const {
        const FrWire claimed = poseidon::CRH::evaluate(leaf_params, leaf);
        FrWire cur = (leaf_index & 1) == 0 ? poseidon::TwoToOneCRH::evaluate(two_params, claimed, leaf_sibling_hash)
                                           : poseidon::TwoToOneCRH::evaluate(two_params, leaf_sibling_hash, claimed);
        size_t index = leaf_index >> 1;
        for (size_t level = auth_path.size(); level-- > 0;) {
            cur = (index & 1) == 0 ? poseidon::TwoToOneCRH::compress(two_params, cur, auth_path[level])
                                   : poseidon::TwoToOneCRH::compress(two_params, auth_path[level], cur);
            index >>= 1;
        }
        return cur == root;
    }
};
template <class P>
class MerkleTree {  // merkle_tree::MerkleTree<P> (:383-726)
  public:
    // MerkleTree::new (:411-422): leaves is [n][leaf_len]; n must be a power of two > 1 (else Error code 5)
    static MerkleTree new_(const typename P::LeafParam& leaf_params, const typename P::TwoToOneParam& two_params,
                           const std::vector<FrWire>& leaves, size_t leaf_len) {
        MerkleTree t;
        const size_t n = leaf_len ? leaves.size() / leaf_len : 0;
        t.leaf_nodes_.resize(n);
        t.non_leaf_nodes_.resize(n ? n - 1 : 0);
        check(akp_merkle_build_poseidon(leaf_params.get(), two_params.get(), leaves.empty() ? nullptr : leaves[0].data(), n, leaf_len,
                                        n ? t.leaf_nodes_[0].data() : nullptr, n > 1 ? t.non_leaf_nodes_[0].data() : nullptr, nullptr));
        t.height_ = 1;
        while (((size_t)1 << (t.height_ - 1)) < n) ++t.height_;
        return t;
    }
    typename P::Digest root() const { return non_leaf_nodes_[0]; }  // :526-528
    size_t height() const { return height_; }                       // :531-533
    const std::vector<typename P::Digest>& leaf_nodes() const { return leaf_nodes_; }
    const std::vector<typename P::Digest>& non_leaf_nodes() const { return non_leaf_nodes_; }  // heap order, root first
    // generate_proof (:572-579) / compute_auth_path (:547-569)
    Path<P> generate_proof(size_t index) const {
        Path<P> p;
        p.leaf_index = index;
        p.leaf_sibling_hash = leaf_nodes_[(index & 1) == 0 ? index + 1 : index - 1];
        size_t cur = (index + ((size_t)1 << (height_ - 1)) - 1 - 1) >> 1;  // parent of the leaf's tree index
        while (cur != 0) {
            const size_t sib = (cur % 2 == 1) ? cur + 1 : cur - 1;
            p.auth_path.push_back(non_leaf_nodes_[sib]);
            cur = (cur - 1) >> 1;
        }
        std::vector<typename P::Digest> rev(p.auth_path.rbegin(), p.auth_path.rend());
        p.auth_path.swap(rev);
        return p;
    }

  private:
    std::vector<typename P::Digest> leaf_nodes_, non_leaf_nodes_;
    size_t height_ = 0;
};

// MerkleTree<P> kept in HBM (akp_merkle_tree_*): what the Rust shim's GpuMerkleTree<P> wraps.  Poseidon field config.
class GpuMerkleTree {
  public:
    using Digest = FrWire;
    // MerkleTree::new (:411-422)
    GpuMerkleTree(const PoseidonConfig& leaf_params, const PoseidonConfig& two_params, const std::vector<FrWire>& leaves, size_t leaf_len)
        : leaf_(&leaf_params), two_(&two_params), leaf_len_(leaf_len) {
        check(akp_merkle_tree_build_poseidon(leaf_params.get(), two_params.get(), leaves.empty() ? nullptr : leaves[0].data(),
                                             leaf_len ? leaves.size() / leaf_len : 0, leaf_len, &h_));
        uint32_t fe = 0;
        check(akp_merkle_tree_info(h_, &n_, &fe, &height_));
    }
    ~GpuMerkleTree() { akp_merkle_tree_destroy(h_); }
    GpuMerkleTree(const GpuMerkleTree&) = delete;
    GpuMerkleTree& operator=(const GpuMerkleTree&) = delete;
    Digest root() const {  // :526-528
        Digest r;
        check(akp_merkle_tree_root(h_, r.data()));
        return r;
    }
    size_t height() const { return height_; }  // :531-533
    // generate_proof (:572-579), many leaves per call
    std::vector<Path<PoseidonFieldConfig>> generate_proofs(const std::vector<uint64_t>& indexes) const {
        const size_t m = indexes.size(), depth = height_ - 2;
        std::vector<FrWire> sib(m), auth(m * depth);
        check(akp_merkle_tree_gather_paths(h_, indexes.data(), m, m ? sib[0].data() : nullptr, (m && depth) ? auth[0].data() : nullptr));
        std::vector<Path<PoseidonFieldConfig>> out(m);
        for (size_t i = 0; i < m; ++i) {
            out[i].leaf_index = (size_t)indexes[i];
            out[i].leaf_sibling_hash = sib[i];
            out[i].auth_path.assign(auth.begin() + i * depth, auth.begin() + (i + 1) * depth);
        }
        return out;
    }
    // update (:692-702), batched: (index, leaf) pairs applied in order, one launch per level
    void update_batch(const std::vector<uint64_t>& indexes, const std::vector<FrWire>& new_leaves) {
        check(akp_merkle_tree_update_batch(h_, indexes.data(), new_leaves.empty() ? nullptr : new_leaves[0].data(), indexes.size(), leaf_len_));
    }
    // check_update (:707-725)
    bool check_update(uint64_t index, const std::vector<FrWire>& new_leaf, const Digest& asserted_new_root) {
        int32_t ok = 0;
        check(akp_merkle_tree_check_update(h_, index, new_leaf.empty() ? nullptr : new_leaf[0].data(), leaf_len_, asserted_new_root.data(), &ok));
        return ok == 1;
    }
    // the reference's two vectors (heap order)
    void export_nodes(std::vector<Digest>& leaf_nodes, std::vector<Digest>& non_leaf_nodes) const {
        leaf_nodes.resize(n_);
        non_leaf_nodes.resize(n_ - 1);
        check(akp_merkle_tree_export(h_, leaf_nodes[0].data(), non_leaf_nodes[0].data()));
    }

  private:
    const PoseidonConfig* leaf_;
    const PoseidonConfig* two_;
    size_t leaf_len_ = 0, n_ = 0, height_ = 0;
    akp_merkle_tree* h_ = nullptr;
};


// ---- several GPUs from one process (akp_multi_*): MerkleTree<P> sharded by leaf range, resident in every device's HBM ------------
// Poseidon field config.  One context + one parameter handle set per device; the RCCL communicator lives inside the library.
class MultiGpu {
  public:
    explicit MultiGpu(const std::vector<int32_t>& device_ids) { check(akp_multi_create(device_ids.data(), (int32_t)device_ids.size(), &h_)); }
    ~MultiGpu() {
        params_.clear();  // parameter handles first: they were created on this object's contexts
        akp_multi_destroy(h_);
    }
    MultiGpu(const MultiGpu&) = delete;
    MultiGpu& operator=(const MultiGpu&) = delete;
    int size() const { return akp_multi_size(h_); }
    akp_multi* get() const { return h_; }
    // Fr::get_default_poseidon_parameters on every device (kept until this object goes)
    std::vector<akp_poseidon*> default_poseidon_parameters(uint32_t rate, bool optimized_for_weights) {
        std::vector<akp_poseidon*> out;
        for (int r = 0; r < size(); ++r) {
            akp_poseidon* p = nullptr;
            check(akp_poseidon_default_params(akp_multi_ctx(h_, r), rate, optimized_for_weights ? 1 : 0, &p));
            params_.emplace_back(p, &akp_poseidon_params_destroy);
            out.push_back(p);
        }
        return out;
    }
    // phases of the last sharded build, milliseconds: sub-trees / all-gather / top levels / copy-out / whole call
    std::array<double, 5> last_phases() const {
        std::array<double, 5> ms{};
        check(akp_multi_last_phases(h_, ms.data()));
        return ms;
    }

  private:
    akp_multi* h_ = nullptr;
    std::vector<std::unique_ptr<akp_poseidon, void (*)(akp_poseidon*)>> params_;
};
// MerkleTree<P> over the devices of a MultiGpu: akp_multi_tree_* (root :526-528, generate_proof :572-579, update :692-702)
class ShardedMerkleTree {
  public:
    using Digest = FrWire;
    // MerkleTree::new (:411-422): host leaves in global order; leaf_params[r] / two_params[r] live on device slot r
    ShardedMerkleTree(MultiGpu& m, const std::vector<akp_poseidon*>& leaf_params, const std::vector<akp_poseidon*>& two_params,
                      const std::vector<FrWire>& leaves, size_t leaf_len)
        : leaf_len_(leaf_len) {
        check(akp_multi_tree_build_poseidon(m.get(), leaf_params.data(), two_params.data(), leaves.empty() ? nullptr : leaves[0].data(),
                                            leaf_len ? leaves.size() / leaf_len : 0, leaf_len, &h_));
        uint32_t fe = 0;
        int32_t g = 0;
        check(akp_multi_tree_info(h_, &n_, &fe, &height_, &g));
    }
    ~ShardedMerkleTree() { akp_multi_tree_destroy(h_); }  // before its MultiGpu
    ShardedMerkleTree(const ShardedMerkleTree&) = delete;
    ShardedMerkleTree& operator=(const ShardedMerkleTree&) = delete;
    Digest root() const {
        Digest r;
        check(akp_multi_tree_root(h_, r.data()));
        return r;
    }
    size_t height() const { return height_; }
    std::vector<Path<PoseidonFieldConfig>> generate_proofs(const std::vector<uint64_t>& indexes) const {
        const size_t m = indexes.size(), depth = height_ - 2;
        std::vector<FrWire> sib(m), auth(m * depth);
        check(akp_multi_tree_gather_paths(h_, indexes.data(), m, m ? sib[0].data() : nullptr, (m && depth) ? auth[0].data() : nullptr));
        std::vector<Path<PoseidonFieldConfig>> out(m);
        for (size_t i = 0; i < m; ++i) {
            out[i].leaf_index = (size_t)indexes[i];
            out[i].leaf_sibling_hash = sib[i];
            out[i].auth_path.assign(auth.begin() + i * depth, auth.begin() + (i + 1) * depth);
        }
        return out;
    }
    void update_batch(const std::vector<uint64_t>& indexes, const std::vector<FrWire>& new_leaves) {
        check(akp_multi_tree_update_batch(h_, indexes.data(), new_leaves.empty() ? nullptr : new_leaves[0].data(), indexes.size(), leaf_len_));
    }

  private:
    size_t leaf_len_ = 0, n_ = 0, height_ = 0;
    akp_multi_tree* h_ = nullptr;
};

// ---- CanonicalSerialize / CanonicalDeserialize (ark-serialize byte formats; akp_serialize_* / akp_deserialize_*) ---------------
// `compress` is ark-serialize's Compress mode, `validate` its Validate mode (false = deserialize_*_unchecked).  Host only.
namespace serialize {
template <class Call>
inline std::vector<uint8_t> write(Call call) {  // size query, then the real call
    size_t n = 0;
    check(call(nullptr, 0, &n));
    std::vector<uint8_t> out(n);
    check(call(out.data(), out.size(), &n));
    return out;
}
// digests back to back, no length prefix: fe = 1 (Fq: Poseidon / Bowe-Hopwood / the x-only Pedersen hashes), 2 (affine point)
inline std::vector<uint8_t> digests(const uint64_t* wire, size_t n, uint32_t fe, bool compress) {
    return write([&](uint8_t* o, size_t cap, size_t* len) { return akp_serialize_digests(wire, n, fe, compress ? 1 : 0, o, cap, len); });
}
inline std::vector<uint8_t> poseidon_config(const PoseidonConfig& cfg) {
    return write([&](uint8_t* o, size_t cap, size_t* len) { return akp_serialize_poseidon_config(cfg.get(), o, cap, len); });
}
// Parameters { generators }: [num_windows][window_size] affine points (x || y wire format), as akp_te_params_create takes them
inline std::vector<uint8_t> te_parameters(const std::vector<FrWire>& generators_affine, uint32_t window_size, uint32_t num_windows, bool compress) {
    return write([&](uint8_t* o, size_t cap, size_t* len) {
        return akp_serialize_te_parameters(generators_affine.empty() ? nullptr : generators_affine[0].data(), window_size, num_windows, compress ? 1 : 0, o, cap,
                                           len);
    });
}
struct TeGenerators {
    std::vector<FrWire> generators_affine;
    uint32_t window_size = 0, num_windows = 0;
};
inline TeGenerators read_te_parameters(const std::vector<uint8_t>& in, bool compress, bool validate = true) {
    TeGenerators g;
    check(akp_deserialize_te_parameters(in.data(), in.size(), compress ? 1 : 0, 0, nullptr, 0, &g.window_size, &g.num_windows));
    g.generators_affine.resize((size_t)g.window_size * g.num_windows * 2);
    if (!g.generators_affine.empty())
        check(akp_deserialize_te_parameters(in.data(), in.size(), compress ? 1 : 0, validate ? 1 : 0, g.generators_affine[0].data(),
                                            (size_t)g.window_size * g.num_windows, &g.window_size, &g.num_windows));
    return g;
}
// Path<PoseidonFieldConfig> (field digests: both modes write the same bytes)
inline std::vector<uint8_t> path(const Path<PoseidonFieldConfig>& p, bool compress = false) {
    return write([&](uint8_t* o, size_t cap, size_t* len) {
        return akp_serialize_path(p.leaf_sibling_hash.data(), p.auth_path.empty() ? nullptr : p.auth_path[0].data(), p.auth_path.size(), p.leaf_index, 1,
                                  compress ? 1 : 0, o, cap, len);
    });
}
inline Path<PoseidonFieldConfig> read_path(const std::vector<uint8_t>& in, bool compress = false, bool validate = true) {
    Path<PoseidonFieldConfig> p;
    size_t depth = 0;
    uint64_t idx = 0;
    check(akp_deserialize_path(in.data(), in.size(), 1, compress ? 1 : 0, 0, nullptr, nullptr, 0, &depth, &idx));
    p.auth_path.resize(depth);
    check(akp_deserialize_path(in.data(), in.size(), 1, compress ? 1 : 0, validate ? 1 : 0, p.leaf_sibling_hash.data(),
                               depth ? p.auth_path[0].data() : nullptr, depth, &depth, &idx));
    p.leaf_index = (size_t)idx;
    return p;
}
// MultiPath in the flat form of akp_merkle_multipath_encode (field digests)
struct FlatMultiPath {
    std::vector<FrWire> leaf_siblings_hashes, suffixes;
    std::vector<uint64_t> prefix_lengths, suffix_lengths, leaf_indexes;
};
inline std::vector<uint8_t> multi_path(const FlatMultiPath& m, bool compress = false) {
    return write([&](uint8_t* o, size_t cap, size_t* len) {
        return akp_serialize_multipath(m.leaf_siblings_hashes.empty() ? nullptr : m.leaf_siblings_hashes[0].data(), m.prefix_lengths.data(),
                                       m.suffix_lengths.data(), m.suffixes.empty() ? nullptr : m.suffixes[0].data(), m.leaf_indexes.data(),
                                       m.leaf_indexes.size(), 0, 1, compress ? 1 : 0, o, cap, len);
    });
}
inline FlatMultiPath read_multi_path(const std::vector<uint8_t>& in, bool compress = false, bool validate = true) {
    FlatMultiPath m;
    size_t n = 0, ns = 0;
    check(akp_deserialize_multipath(in.data(), in.size(), 1, compress ? 1 : 0, 0, &n, &ns, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0));
    m.leaf_siblings_hashes.resize(n ? n : 1);
    m.prefix_lengths.resize(n ? n : 1);
    m.suffix_lengths.resize(n ? n : 1);
    m.leaf_indexes.resize(n ? n : 1);
    m.suffixes.resize(ns ? ns : 1);
    check(akp_deserialize_multipath(in.data(), in.size(), 1, compress ? 1 : 0, validate ? 1 : 0, &n, &ns, m.leaf_siblings_hashes[0].data(),
                                    m.prefix_lengths.data(), m.suffix_lengths.data(), m.suffixes[0].data(), m.leaf_indexes.data(), n, ns));
    m.leaf_siblings_hashes.resize(n);
    m.prefix_lengths.resize(n);
    m.suffix_lengths.resize(n);
    m.leaf_indexes.resize(n);
    m.suffixes.resize(ns);
    return m;
}
}  // namespace serialize

}  // namespace akp
