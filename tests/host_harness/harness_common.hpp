// shared by harness.hip and harness_crh.hip: host-side constant preparation exactly as capi_poseidon.hip does it (test infrastructure)
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include <vector>
#include "../../crypto_primitives_amd/csrc/fr.hpp"
#include "../../crypto_primitives_amd/csrc/f29.hpp"
#include "../../crypto_primitives_amd/csrc/poseidon_kernels.hpp"
#include "../../crypto_primitives_amd/csrc/poseidon_opt.hpp"
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
