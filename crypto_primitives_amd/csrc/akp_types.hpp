// akp_types.hpp -- the plain types the kernel headers and the host units of libakp.so share (capi_internal.hpp includes this
// instead of the kernel headers, so that a unit compiles only the kernel family it launches).
#pragma once
#include "f29.hpp"

namespace akp {

// PoseidonConfig dimensions (sponge/poseidon/mod.rs:27-45), t = rate + capacity
struct PoseidonDims {
    u32 t, rate, capacity, full_rounds, partial_rounds;
    u64 alpha;
};

AKP_HD Fr load_fr_global(const Fr* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    const uint4 lo = q[0], hi = q[1];
    return Fr{{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w}};
}
AKP_HD void store_fr_global(Fr* p, const Fr& v) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}

// Table entry: the 27 limbs back to back in ONE 128-byte cache line (w[0..8] = (y+x)/2, w[9..17] = (y-x)/2, w[18..26] = dxy,
// 5 dwords of padding): a lane's entry is two 64-byte L2 sectors and seven 16-byte loads.  (Round 1-2 layout: three
// 48-byte padded elements = 144 B, 3.3 sectors and twelve loads per entry; the gather was 19 % of the Pedersen kernel.)
struct alignas(128) NielsPad {
    u32 w[32];
};
// (a packed 96-byte entry was measured in round 3 and lost: profiles/HISTORY.md)
typedef NielsPad TeEntry;

}  // namespace akp
