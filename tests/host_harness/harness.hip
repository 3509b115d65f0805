// Test-only harness: runs the PRODUCT's per-item device logic (the __host__ __device__ functions of
// crypto_primitives_amd/csrc/*.hpp: round loop, sponge collapse, table construction, digit
// accumulation, shared-inversion finalisation, digest serialisation) on the CPU, so that the
// `-m "not gpu"` suite can compare it with the oracle without a GPU.  It is NOT part of the
// product library and is never used as a fallback; the GPU kernels call the same functions.
// (On the host pass fr_mul is the portable multiplier; the inline-asm multiplier is covered by
// the GPU parity tests.)
#include "harness_common.hpp"

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
}
