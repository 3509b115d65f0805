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
// Packed alternative (round 3 A/B, `make packed96`): the three values as CANONICAL 256-bit integers (8 dwords each, no
// padding) = 96 bytes.  The 16-bit signed Pedersen table shrinks from 268 MB to 201 MB -- inside the 256 MB Infinity Cache --
// and an entry is six 16-byte loads instead of seven, at the price of re-limbing 3 x 256 bits into 3 x 9 limbs of 29 bits in
// registers (~50 VALU instructions per step) and of entries that straddle two 128-byte lines.  Canonical values are
// non-negative with limbs < 2^29: they satisfy every operand bound the 128-byte form does.
struct alignas(32) Niels96 {
    u32 w[24];
};
#if defined(AKP_TE_PACKED96)
typedef Niels96 TeEntry;
#else
typedef NielsPad TeEntry;
#endif

}  // namespace akp
