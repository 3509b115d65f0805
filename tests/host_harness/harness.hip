// Test-only harness: runs the PRODUCT's per-item device logic (the __host__ __device__ functions of
// crypto_primitives_amd/csrc/*.hpp: round loop, sponge collapse, table construction, digit
// accumulation, shared-inversion finalisation, digest serialisation) on the CPU, so that the
// `-m "not gpu"` suite can compare it with the oracle without a GPU.  It is NOT part of the
// product library and is never used as a fallback; the GPU kernels call the same functions.
// (On the host pass fr_mul is the portable multiplier; the inline-asm multiplier is covered by
// the GPU parity tests.)
#include <hip/hip_runtime.h>
#include <vector>
#include "../../crypto_primitives_amd/csrc/fr.hpp"
#include "../../crypto_primitives_amd/csrc/poseidon_kernels.hpp"
#include "../../crypto_primitives_amd/csrc/te_kernels.hpp"
using namespace akp;

struct HostFile {
    Fr* slots;
    Fr load(u32 s) const { return slots[s]; }
    void store(u32 s, const Fr& v) const { slots[s] = v; }
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

void hh_poseidon_permute(uint32_t rf, uint32_t rp, uint64_t alpha, uint32_t rate, uint32_t cap, const Fr* ark, const Fr* mds,
                         Fr* states, size_t n) {
    PoseidonDims D = mk(rf, rp, alpha, rate, cap);
    std::vector<Fr> buf(2 * D.t);
    HostFile f{buf.data()};
    for (size_t i = 0; i < n; ++i) {
        u32 cur = 0;
        for (u32 e = 0; e < D.t; ++e) f.store(e, states[i * D.t + e]);
        poseidon_permute_file(D, ark, mds, f, cur);
        for (u32 e = 0; e < D.t; ++e) states[i * D.t + e] = f.load(cur * D.t + e);
    }
}
void hh_poseidon_crh(uint32_t rf, uint32_t rp, uint64_t alpha, uint32_t rate, uint32_t cap, const Fr* ark, const Fr* mds,
                     const Fr* in0, const Fr* in1, size_t k, Fr* out, size_t n) {
    PoseidonDims D = mk(rf, rp, alpha, rate, cap);
    std::vector<Fr> buf(2 * D.t);
    HostFile f{buf.data()};
    for (size_t i = 0; i < n; ++i) out[i] = poseidon_crh_item(D, ark, mds, f, in0, in1, k, i);
}
// returns number of LUT entries written
size_t hh_te_build_lut(int kind, const Fr* gens, uint32_t W, uint32_t N, Niels* lut) {
    if (kind == 0) {
        const u32 subs = (W + 3) / 4, n_sub = N * subs;
        for (u32 i = 0; i < n_sub * 16; ++i) lut[i] = te_pedersen_lut_entry(gens, W, subs, i);
        return (size_t)n_sub * 16;
    }
    for (u32 i = 0; i < W * N * 4; ++i) lut[i] = te_bh_lut_entry(gens, i);
    return (size_t)W * N * 4;
}
void hh_te_crh(int kind, const Niels* lut, const uint8_t* msgs, size_t n, size_t msg_len, uint32_t W, uint32_t subs,
               uint32_t steps, size_t lanes, Fr* out) {
    std::vector<Fr> xyz(n * 3), prefix(n);
    for (size_t i = 0; i < n; ++i) {
        Ext a = kind == 0 ? te_accumulate_item<0>(lut, msgs + i * msg_len, msg_len, W, subs, steps)
                          : te_accumulate_item<1>(lut, msgs + i * msg_len, msg_len, W, subs, steps);
        xyz[3 * i] = a.X; xyz[3 * i + 1] = a.Y; xyz[3 * i + 2] = a.Z;
    }
    for (size_t l = 0; l < lanes && l < n; ++l) {
        if (kind == 0) te_finalize_lane<0>(xyz.data(), prefix.data(), out, n, lanes, l);
        else te_finalize_lane<1>(xyz.data(), prefix.data(), out, n, lanes, l);
    }
}
void hh_te_serialize_pairs(const Fr* left, const Fr* right, uint32_t fe, size_t buflen, uint8_t* buf, size_t n) {
    for (size_t t = 0; t < n * 2 * fe; ++t) te_serialize_pair_fe(left, right, fe, buflen, buf, t);
}
}
