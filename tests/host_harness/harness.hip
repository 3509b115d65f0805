// Test-only harness: runs the PRODUCT's per-item device logic (the __host__ __device__ functions of
// crypto_primitives_amd/csrc/*.hpp: round loop, sponge collapse, table construction, digit
// accumulation, shared-inversion finalisation, digest serialisation) on the CPU, so that the
// `-m "not gpu"` suite can compare it with the oracle without a GPU.  It is NOT part of the
// product library and is never used as a fallback; the GPU kernels call the same functions.
// (On the host pass fr_mul is the portable multiplier; the inline-asm multiplier is covered by
// the GPU parity tests.)
#include <hip/hip_runtime.h>
#include <type_traits>
#include <vector>
#include "../../crypto_primitives_amd/csrc/fr.hpp"
#include "../../crypto_primitives_amd/csrc/f29.hpp"
#include "../../crypto_primitives_amd/csrc/poseidon_kernels.hpp"
#include "../../crypto_primitives_amd/csrc/poseidon_opt.hpp"
#include "../../crypto_primitives_amd/csrc/te_kernels.hpp"
using namespace akp;

struct HostFile {
    FP* slots;
    FP load(u32 s) const { return slots[s]; }
    void store(u32 s, const FP& v) const { slots[s] = v; }
};
// wire-format parameter arrays -> internal form (what poseidon_convert_params_kernel does on the device)
static std::vector<F29Pad> to29(const Fr* in, size_t n) {
    std::vector<F29Pad> out(n);
    for (size_t i = 0; i < n; ++i) f29_store_pad(&out[i], f29_balance(f29_from_wire<AKP_PS>(in[i])));  // as poseidon_convert_params_kernel
    return out;
}
// force_generic: 0 = product default (t == 3: register path, else LDS-file path; sparse partial rounds),
//                1 = generic file path with sparse partial rounds even for t == 3,
//                2 = dense partial rounds (t == 3: register path, else file path)
struct T3Host {  // constants in internal form for any t (name kept from the t = 3 path)
    std::vector<F29Pad> ark, mds, mpre, sparse, sbox0, mpre_w, sparse_w, ark_f, fmats_f, sparse_f, sbox0_f;
    bool has_lane1 = false, has_full = false;
    PoseidonConsts c;     // what the wave-per-lane kernels get (lane-0 form)
    PoseidonConsts cfile; // what the one-lane-per-item kernels get: full form, else lane-1 form, else c (as capi_poseidon.hip does)
    PoseidonConsts creg;  // == cfile (kept for the t = 3 register-path call sites)
    T3Host(uint32_t t, uint32_t rf, uint32_t rp, uint64_t alpha, const Fr* a, const Fr* m, bool sparse_form) {
        std::vector<Fr> av(a, a + (size_t)(rf + rp) * t), mv(m, m + (size_t)t * t);
        PoseidonOpt o;
        PoseidonOpt ow;
        PoseidonFullForm ff;
        bool have_w = false, have_f = false;
        if (sparse_form) {
            o = poseidon_optimize(t, rf, rp, av, mv);
            ow = o;
            have_f = poseidon_full_form(o, t, rf, rp, alpha, mv, ff);
            poseidon_rescale_sparse(o, t, rp, alpha);
            have_w = poseidon_rescale_sparse_lane1(ow, t, rp, alpha);
        }
        mds = to29(mv.data(), mv.size());
        if (o.ok) { ark = to29(o.ark_mod.data(), o.ark_mod.size()); mpre = to29(o.mpre.data(), o.mpre.size()); sparse = to29(o.sparse.data(), o.sparse.size());
                    c = PoseidonConsts{ark.data(), mds.data(), mpre.data(), sparse.data(), nullptr, o.scaled ? 1u : 0u}; }
        else { ark = to29(av.data(), av.size()); c = PoseidonConsts{ark.data(), mds.data(), nullptr, nullptr, nullptr, 0u}; }
        if (rf >= 2) {  // as capi_poseidon.hip does: from the round keys the kernels use
            const std::vector<Fr> s0 = poseidon_sbox0(o.ok ? o.ark_mod : av, t, alpha);
            sbox0 = to29(s0.data(), s0.size());
            c.sbox0 = sbox0.data();
        }
        cfile = c;
        if (have_w) {
            mpre_w = to29(ow.mpre.data(), ow.mpre.size());
            sparse_w = to29(ow.sparse.data(), ow.sparse.size());
            cfile.mpre = mpre_w.data();
            cfile.sparse = sparse_w.data();
            cfile.scaled = 2u;
        }
        has_lane1 = have_w;
        has_full = have_f;
        if (have_f) {
            ark_f = to29(ff.ark.data(), ff.ark.size());
            fmats_f = to29(ff.fmats.data(), ff.fmats.size());
            sparse_f = to29(ff.sparse.data(), ff.sparse.size());
            const std::vector<Fr> s0f = poseidon_sbox0(ff.ark, t, alpha);
            sbox0_f = to29(s0f.data(), s0f.size());
            cfile = PoseidonConsts{ark_f.data(), fmats_f.data(), nullptr, sparse_f.data(), sbox0_f.data(), 3u};
        }
        creg = cfile;
    }
};
static PoseidonDims mk(uint32_t rf, uint32_t rp, uint64_t alpha, uint32_t rate, uint32_t cap) {
    return PoseidonDims{rate + cap, rate, cap, rf, rp, alpha};
}
extern "C" {
void hh_fr_mul(const Fr* a, const Fr* b, Fr* o) { *o = fr_mul(*a, *b); }
void hh_fr_add(const Fr* a, const Fr* b, Fr* o) { *o = fr_add(*a, *b); }
void hh_fr_sub(const Fr* a, const Fr* b, Fr* o) { *o = fr_sub(*a, *b); }
void hh_fr_inv(const Fr* a, Fr* o) { *o = fr_inv(*a); }
void hh_fr_pow(const Fr* a, uint64_t e, Fr* o) { *o = fr_pow_small(*a, e); }
// radix-2^29 core, both flavours: out = [a*b, a^2, (a+b)*(c+d) (FU), (a-b)*c (FS), (a-b)*weak_norm(c+d) (FS), a^-1, a^17,
//                                       dot3(a+c, b, b+d, c, c, d), canonical integer of a]
void hh_f29_ops(const Fr* a, const Fr* b, const Fr* c, const Fr* d, Fr* o) {
    const FU ua = f29_from_wire<false>(*a), ub = f29_from_wire<false>(*b), uc = f29_from_wire<false>(*c), ud = f29_from_wire<false>(*d);
    const FS sa = f29_from_wire<true>(*a), sb = f29_from_wire<true>(*b), sc = f29_from_wire<true>(*c), sd = f29_from_wire<true>(*d);
    o[0] = f29_to_wire(f29_mul(ua, ub));
    o[1] = f29_to_wire(f29_sqr(sa));
    o[2] = f29_to_wire(f29_mul(f29_add(ua, ub), f29_add(uc, ud)));
    o[3] = f29_to_wire(f29_mul(f29_sub(sa, sb), sc));
    o[4] = f29_to_wire(f29_mul(f29_sub(sa, sb), f29_weak_norm(f29_add(sc, sd))));
    o[5] = f29_to_wire(f29_inv(sa));
    o[6] = f29_to_wire(f29_pow_small(ua, 17));
    o[7] = f29_to_wire(f29_dot3(f29_add(ua, uc), ub, f29_add(ub, ud), uc, uc, ud));
    o[8] = f29_to_canonical_int(sa);
    o[9] = f29_to_wire(f29_sqr(f29_add(ua, ub)));
    o[10] = f29_to_wire(f29_neg(f29_sub(sa, sb)));
}
// inverses: out = [safegcd (FS), safegcd (FU), a^(p-2) (FS), safegcd of the lazy value a - b (FS)]
void hh_f29_inv(const Fr* a, const Fr* b, Fr* o) {
    const FS sa = f29_from_wire<true>(*a), sb = f29_from_wire<true>(*b);
    o[0] = f29_to_wire(f29_inv(sa));
    o[1] = f29_to_wire(f29_inv(f29_from_wire<false>(*a)));
    o[2] = f29_to_wire(f29_inv_fermat(sa));
    o[3] = f29_to_wire(f29_inv(f29_sub(sa, sb)));
}
// worst-case limb patterns fed straight into the multipliers (internal limbs, not via the wire):
// returns a*b/2^261 mod p canonical for FU (limbs given), and for FS.
void hh_f29_raw_mul(const uint32_t* al, const uint32_t* bl, int flags, Fr* o) {
    // flags: bit 0 signed flavour, bit 1 skip the square, bit 2 skip the three-term dot product -- a routine is only run on
    // operands inside ITS documented limb bounds, so that the sanitizer build (make ubsan: signed-integer-overflow) flags every
    // accumulator overflow as the bug it would be
    const bool sqr = !(flags & 2), dot = !(flags & 4);
    if (flags & 1) { FS x, y; for (int i = 0; i < 9; ++i) { x.l[i] = (int32_t)al[i]; y.l[i] = (int32_t)bl[i]; }
        o[0] = f29_canonical_pack(f29_mul(x, y)); if (sqr) o[1] = f29_canonical_pack(f29_sqr(x)); if (dot) o[2] = f29_canonical_pack(f29_dot3(x, y, x, y, x, y)); }
    else { FU x, y; for (int i = 0; i < 9; ++i) { x.l[i] = al[i]; y.l[i] = bl[i]; }
        o[0] = f29_canonical_pack(f29_mul(x, y)); if (sqr) o[1] = f29_canonical_pack(f29_sqr(x)); if (dot) o[2] = f29_canonical_pack(f29_dot3(x, y, x, y, x, y)); }
}

// four / five-term dot products at their limb bounds (state limbs <= 2^29 + 2, balanced constant digits |d| <= 2^28), and the
// digit balancing itself: out = [dot4(x,y,...), dot5(x,y,...), canonical(balance(y))]
void hh_f29_dotn(const uint32_t* al, const uint32_t* bl, Fr* o) {
    FS x, y;
    for (int i = 0; i < 9; ++i) { x.l[i] = (int32_t)al[i]; y.l[i] = (int32_t)bl[i]; }
    o[0] = f29_canonical_pack<true, true>(f29_dot4(x, y, x, y, x, y, x, y));
    o[1] = f29_canonical_pack<true, true>(f29_dot5(x, y, x, y, x, y, x, y, x, y));
    const FS b = f29_balance(y);
    bool ok = true;
    for (int i = 0; i < 8; ++i) ok = ok && b.l[i] >= -(1 << 28) && b.l[i] < (1 << 28);
    o[2] = f29_canonical_pack<true, true>(b);
    o[3] = Fr{{ok ? 1u : 0u, 0, 0, 0, 0, 0, 0, 0}};
}

void hh_poseidon_permute(uint32_t rf, uint32_t rp, uint64_t alpha, uint32_t rate, uint32_t cap, const Fr* ark, const Fr* mds,
                         Fr* states, size_t n, int force_generic) {
    PoseidonDims D = mk(rf, rp, alpha, rate, cap);
    std::vector<FP> buf(2 * D.t);
    HostFile f{buf.data()};
    T3Host* th = new T3Host(D.t, rf, rp, alpha, ark, mds, force_generic != 2);
    const bool reg_path = D.t == 3 && force_generic != 1;
    for (size_t i = 0; i < n; ++i) {
        if (reg_path) {  // the register-resident fast path
            FP s0 = f29_from_wire<AKP_PS>(states[i * 3]), s1 = f29_from_wire<AKP_PS>(states[i * 3 + 1]), s2 = f29_from_wire<AKP_PS>(states[i * 3 + 2]);
            if (th->creg.scaled == 3u) {  // as poseidon_permute_t3_kernel<true>: the lanes are the wire values
                s0 = f29_unpack<AKP_PS>(states[i * 3]); s1 = f29_unpack<AKP_PS>(states[i * 3 + 1]); s2 = f29_unpack<AKP_PS>(states[i * 3 + 2]);
                poseidon_permute_t3<true>(D, th->creg, s0, s1, s2);
                states[i * 3] = f29_canonical_pack(s0); states[i * 3 + 1] = f29_canonical_pack(s1); states[i * 3 + 2] = f29_canonical_pack(s2);
                continue;
            }
            poseidon_permute_t3<false>(D, th->creg, s0, s1, s2);
            states[i * 3] = f29_to_wire(s0); states[i * 3 + 1] = f29_to_wire(s1); states[i * 3 + 2] = f29_to_wire(s2);
            continue;
        }
        // t = 4, 5 with the full / lane-1 form: the register-resident path (as capi_poseidon.hip routes large batches)
        if ((D.t >= 4 && D.t <= 9) && force_generic == 0 && (th->cfile.scaled == 3u || th->cfile.scaled == 2u) && th->cfile.sparse) {
            auto run = [&](auto tag, auto ff) {
                constexpr u32 T = decltype(tag)::value;
                constexpr bool FF = decltype(ff)::value;
                FP s[T];
                for (u32 e = 0; e < T; ++e) s[e] = reg_load<T, FF>(&states[i * T + e]);
                poseidon_permute_reg<T, FF>(D, th->cfile, s);
                for (u32 e = 0; e < T; ++e) states[i * T + e] = reg_store<T, FF>(s[e]);
            };
            const bool ff = th->cfile.scaled == 3u;
            switch (D.t) {
#define AKP_RUN(TT) case TT: if (ff) run(std::integral_constant<u32, TT>{}, std::true_type{}); else run(std::integral_constant<u32, TT>{}, std::false_type{}); break;
                AKP_RUN(4) AKP_RUN(5) AKP_RUN(6) AKP_RUN(7) AKP_RUN(8) AKP_RUN(9)
#undef AKP_RUN
            }
            continue;
        }
        const bool wire = th->cfile.scaled == 3u;  // as poseidon_permute_kernel
        for (u32 e = 0; e < D.t; ++e) f.store(e, wire ? f29_unpack<AKP_PS>(states[i * D.t + e]) : f29_from_wire<AKP_PS>(states[i * D.t + e]));
        poseidon_permute_file(D, th->cfile, f);
        for (u32 e = 0; e < D.t; ++e) states[i * D.t + e] = wire ? f29_canonical_pack<AKP_PS, true>(f.load(e)) : f29_to_wire(f.load(e));
    }
    delete th;
}
// which constant forms exist for a parameter set: bit 0 sparse, bit 1 lane-0 rescaling, bit 2 lane-1 form, bit 3 full form
int hh_poseidon_forms(uint32_t rf, uint32_t rp, uint64_t alpha, uint32_t rate, uint32_t cap, const Fr* ark, const Fr* mds) {
    T3Host th(rate + cap, rf, rp, alpha, ark, mds, true);
    return (th.c.sparse ? 1 : 0) | (th.c.scaled == 1u ? 2 : 0) | (th.has_lane1 ? 4 : 0) | (th.has_full ? 8 : 0);
}
void hh_poseidon_crh(uint32_t rf, uint32_t rp, uint64_t alpha, uint32_t rate, uint32_t cap, const Fr* ark, const Fr* mds,
                     const Fr* in0, const Fr* in1, size_t k, Fr* out, size_t n, int force_generic) {
    PoseidonDims D = mk(rf, rp, alpha, rate, cap);
    std::vector<FP> buf(2 * D.t);
    HostFile f{buf.data()};
    T3Host* th = new T3Host(D.t, rf, rp, alpha, ark, mds, force_generic != 2);
    const bool reg_path = D.t == 3 && force_generic != 1;
    const bool reg45 = (D.t >= 4 && D.t <= 9) && force_generic == 0 && (th->cfile.scaled == 3u || th->cfile.scaled == 2u) && th->cfile.sparse;
    for (size_t i = 0; i < n; ++i) {
        if (reg45) {
            const bool ff = th->cfile.scaled == 3u;
            switch (D.t) {
#define AKP_RUN(TT) case TT: out[i] = ff ? poseidon_crh_item_reg<TT, true>(D, th->cfile, in0, in1, k, i) : poseidon_crh_item_reg<TT, false>(D, th->cfile, in0, in1, k, i); break;
                AKP_RUN(4) AKP_RUN(5) AKP_RUN(6) AKP_RUN(7) AKP_RUN(8) AKP_RUN(9)
#undef AKP_RUN
            }
            continue;
        }
        out[i] = reg_path ? (th->creg.scaled == 3u ? poseidon_crh_item_t3<true>(D, th->creg, in0, in1, k, i) : poseidon_crh_item_t3<false>(D, th->creg, in0, in1, k, i)) : poseidon_crh_item(D, th->cfile, f, in0, in1, k, i);
    }
    delete th;
}
// LUT construction exactly as capi_te.hip does it: kind 0 -> Pedersen with digit width D (lut: [ceil(n_gen/D)][2^D]);
// kind 1 -> Bowe-Hopwood single table lut1 [n_gen][4] and, when group > 1, group table lut [n_gen/G][2^(3G-1)]
// (hh_te_crh then takes D = group for kind 1).
void hh_te_build_lut(int kind, const Fr* gens, uint32_t W, uint32_t N, uint32_t D, uint32_t group, TeEntry* lut, TeEntry* lut1) {
    const u32 n_gen = W * N;
    if (kind == 2) {  // Pedersen, signed-subset table: lut [n_digits][2^(D-1)], lut1 = cprefix [n_digits + 1]
        std::vector<NielsPad> half(n_gen);
        for (u32 g = 0; g < n_gen; ++g) {
            Niels h;
            (void)te_half_generator(gens, g, h);
            store_niels(&half[g], h);
        }
        const u32 n_digits = (n_gen + D - 1) / D;
        for (u32 i = 0; i < (n_digits << (D - 1)); ++i) store_niels(lut + i, te_pedersen_slut_entry(half.data(), n_gen, D, i));
        for (u32 k = 0; k <= n_digits; ++k) store_niels(lut1 + k, te_pedersen_cprefix_entry(half.data(), n_gen, D, k));
        return;
    }
    if (kind == 0) {
        const u32 entries = ((n_gen + D - 1) / D) << D;
        for (u32 i = 0; i < entries; ++i) store_niels(lut + i, te_pedersen_lut_entry(gens, n_gen, D, i));
        return;
    }
    for (u32 i = 0; i < n_gen * 4; ++i) store_niels(lut1 + i, te_bh_lut_entry(gens, i));
    if (group > 1)
        for (u32 i = 0; i < ((n_gen / group) << (3 * group - 1)); ++i) store_niels(lut + i, te_bh_lutg_entry(gens, group, i));
}
void hh_te_crh(int kind, const TeEntry* lut, const TeEntry* lut1, const uint8_t* msgs, size_t n, size_t msg_len, uint32_t D,
               uint32_t groups, uint32_t steps, size_t lanes, Fr* out) {
    std::vector<F29Pad> xyz(n * 3), prefix(n);
    for (size_t i = 0; i < n; ++i) {
        Ext a = kind == 0 ? te_accumulate_item<0>(lut, lut1, msgs + i * msg_len, msg_len, D, groups, steps)
                          : (kind == 2 ? te_accumulate_item<2>(lut, lut1, msgs + i * msg_len, msg_len, D, groups, steps)
                                       : te_accumulate_item<1>(lut, lut1, msgs + i * msg_len, msg_len, D, groups, steps));
        f29_store_pad(&xyz[3 * i], a.X); f29_store_pad(&xyz[3 * i + 1], a.Y); f29_store_pad(&xyz[3 * i + 2], a.Z);
    }
    for (size_t l = 0; l < lanes && l < n; ++l) {
        if (kind != 1) te_finalize_lane<0>(xyz.data(), prefix.data(), out, n, lanes, l);
        else te_finalize_lane<1>(xyz.data(), prefix.data(), out, n, lanes, l);
    }
}
// the small-batch kernel's arithmetic (te_crh_small_kernel) on the CPU: `split` strided partial sums, a binary tree of
// full additions, one inversion per message
void hh_te_crh_split(int kind, const TeEntry* lut, const TeEntry* lut1, const uint8_t* msgs, size_t n, size_t msg_len, uint32_t D,
                     uint32_t groups, uint32_t steps, uint32_t split, Fr* out) {
    for (size_t i = 0; i < n; ++i) {
        std::vector<Ext> part(split);
        for (u32 j = 0; j < split; ++j)
            part[j] = kind == 0 ? te_accumulate_strided<0>(lut, lut1, msgs + i * msg_len, msg_len, D, groups, steps, j, split)
                                : (kind == 2 ? te_accumulate_strided<2>(lut, lut1, msgs + i * msg_len, msg_len, D, groups, steps, j, split)
                                             : te_accumulate_strided<1>(lut, lut1, msgs + i * msg_len, msg_len, D, groups, steps, j, split));
        for (u32 stride = 1; stride < split; stride <<= 1)
            for (u32 j = 0; j + stride < split; j += 2 * stride) part[j] = te_add_ext(part[j], part[j + stride]);
        const FS zi = f29_inv(part[0].Z);
        if (kind != 1) { out[2 * i] = f29_to_wire(f29_mul(part[0].X, zi)); out[2 * i + 1] = f29_to_wire(f29_mul(part[0].Y, zi)); }
        else out[i] = f29_to_wire(f29_mul(part[0].X, zi));
    }
}
// 1 when the generator is in the prime-order subgroup (2 * (G / 2) == G)
int hh_te_in_subgroup(const Fr* gen_affine) {
    Niels h;
    return te_half_generator(gen_affine, 0, h) ? 1 : 0;
}
void hh_te_serialize_pairs(const Fr* left, const Fr* right, uint32_t fe, size_t buflen, uint8_t* buf, size_t n) {
    for (size_t t = 0; t < n * 2 * fe; ++t) te_serialize_pair_fe(left, right, fe, buflen, buf, t);
}
}
