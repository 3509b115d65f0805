// Test-only harness, Poseidon CRH half (see harness.hip): the fixed-length sponge collapse of every kernel flavour on the CPU.
// A unit of its own so that the template instantiations of the register paths (t = 4 ... 9, both constant forms) compile in
// parallel with the permutation half; all units are linked into harness.so.
#include "harness_common.hpp"

extern "C" {
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
}
