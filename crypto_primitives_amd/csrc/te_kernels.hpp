// te_kernels.hpp -- Pedersen / Bowe-Hopwood CRH over Jubjub (ark_ed_on_bls12_381) on gfx950.
//
// Replaces, batch-wide:
//   pedersen::CRH::evaluate        crh/pedersen/mod.rs:76-129   (hot loop :112-124)
//   bowe_hopwood::CRH::evaluate    crh/bowe_hopwood/mod.rs:114-186 (hot loop :161-181)
// The reference walks the message bit by bit doing conditional projective additions of
// `generators[i][j]`.  Here the hash is evaluated as a FIXED-BASE windowed multi-scalar sum:
//   Pedersen:  message bit g selects generators[g / W][g % W], i.e. FLAT generator index g, so the window
//              structure is irrelevant to evaluation and the digit width D is a free tuning parameter:
//              H(m) = sum over digits u of LUT[u][digit_u],  digit_u = message bits [uD, uD + D),
//              LUT[u][v] = sum_b v_b * G[uD + b]   (valid for arbitrary generators; with the signed-subset table below and
//              24-bit digits a 4x256 hash is 43 mixed additions instead of the reference's ~512 conditional + 255 window adds)
//   Bowe-Hopwood: chunk c (3 bits) uses flat generator G[c]; digit = (1 + b0 + 2 b1) * (-1)^b2 (zero chunk = +g, :167).
//              G chunks per step (G up to 8): sum_i (-1)^{s_i} (k_i+1) G[Gu+i] = (-1)^{s_0} * LUTG[u][k_0..k_{G-1}, s_i^s_0]
//              (2^(3G-1) entries per group); the < G chunks left over at the end of a message use LUT1[c][k] = (k+1) G[c].
// LUT entries are precomputed once per parameter set in halved "Niels" form ((y+x)/2, (y-x)/2, d*x*y),
// so one step is a 7-product mixed addition (madd-2008-hwcd-3, a = -1, complete on Jubjub because d is a
// non-square; every coordinate comes out scaled by 1/4, which the projective form absorbs and which
// removes the doubling of Z).  One message per lane; the tables are sized for HBM (capi_te.hip: tens of GB when the
// table budget allows, 268 / 237 MB when it is set to keep them in the 256 MB Infinity Cache) -- per step a wavefront gathers
// 64 x 128 B (the entry of step u+1 is fetched before the addition of step u) against ~1480 VALU instructions, and the path
// stays integer-ALU bound at either size (VALUBusy 88-95 %).
// Arithmetic: signed lazy radix-2^29 form (f29.hpp, FS): subtraction is limb-wise, no reduction anywhere.
// The projective -> affine conversion (crh/pedersen/mod.rs:128, bowe_hopwood/mod.rs:185) is one field
// inversion per message in the reference; here a separate pass shares one inversion among up to 64 messages
// per lane (Montgomery's trick), ~5 products per message.
//
// The north-star text suggests LDS bucket accumulation; buckets belong to variable-base MSM (Pippenger).
// With fixed bases the table method needs no buckets and no cross-lane reduction (DESIGN.md "Pedersen").
#pragma once
#include "f29.hpp"
#include "akp_types.hpp"
#include "te_shape.hpp"

namespace akp {

AKP_F29_CONST(f29_inv2, 0x1fffffddu, 0x00000117u, 0x0e5b08c0u, 0x05272e00u, 0x177458d1u, 0x1b73576fu, 0x1c83fc5du, 0x1429eefbu, 0x0026821fu)

struct Niels {
    FS ypx, ymx, dxy;  // (y + x)/2, (y - x)/2, d*x*y   -- normalised
};
struct Ext {
    FS X, Y, Z, T;  // x = X/Z, y = Y/Z, T = XY/Z  -- normalised (product outputs)
};

AKP_HD Ext ext_identity() { return Ext{f29_zero<true>(), f29_one<true>(), f29_one<true>(), f29_zero<true>()}; }
AKP_HD Niels niels_from_affine(const FS& x, const FS& y) {
    const FS h = f29_inv2<true>();
    return Niels{f29_mul(f29_add(y, x), h), f29_mul(f29_sub(y, x), h), f29_mul(f29_mul(x, y), f29_te_d<true>())};
}
AKP_HD Niels niels_neg(const Niels& q) { return Niels{q.ymx, q.ypx, f29_neg(q.dxy)}; }

// P + Q, Q affine in halved Niels form (7 products).  Limb bounds (FS rule |a_i|*|b_j| <= 2^59.4):
//   Y-X, Y+X: < 2^30 against normalised constants; e, f in (-2^29, 2^29); g, h in [0, 2^30):
//   g*h would be 2^60, so g is renormalised first.
AKP_HD Ext te_madd(const Ext& p, const Niels& q) {
    const FS a = f29_mul(f29_sub(p.Y, p.X), q.ymx);
    const FS b = f29_mul(f29_add(p.Y, p.X), q.ypx);
    const FS c = f29_mul(p.T, q.dxy);
    const FS e = f29_sub(b, a), f = f29_sub(p.Z, c), g = f29_weak_norm(f29_add(p.Z, c)), h = f29_add(b, a);
    return Ext{f29_mul(e, f), f29_mul(g, h), f29_mul(f, g), f29_mul(e, h)};
}

// P + Q, both extended (9 products; unified, complete for a = -1, d non-square).  Used to combine partial sums.
// Limb bounds as above: one operand of every product is (re)normalised.
AKP_HD Ext te_add_ext(const Ext& p, const Ext& q) {
    const FS a = f29_mul(f29_weak_norm(f29_sub(p.Y, p.X)), f29_sub(q.Y, q.X));
    const FS b = f29_mul(f29_weak_norm(f29_add(p.Y, p.X)), f29_add(q.Y, q.X));
    const FS c = f29_mul(f29_mul(p.T, q.T), f29_dbl(f29_te_d<true>()));
    const FS zz = f29_mul(p.Z, q.Z);
    const FS d = f29_dbl(zz);
    const FS e = f29_sub(b, a), f = f29_sub(d, c), g = f29_weak_norm(f29_add(d, c)), h = f29_add(b, a);
    return Ext{f29_mul(e, f), f29_mul(g, h), f29_mul(f, g), f29_mul(e, h)};
}

// the affine point of a table entry as an extended point (1 product instead of a 7-product addition to the identity)
AKP_HD Ext ext_from_niels(const Niels& q) {
    const FS x = f29_sub(q.ypx, q.ymx), y = f29_weak_norm(f29_add(q.ypx, q.ymx));  // (y+x)/2 -+ (y-x)/2
    return Ext{x, y, f29_one<true>(), f29_mul(x, y)};
}

AKP_HD Fr load_fr_g(const Fr* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    const uint4 lo = q[0], hi = q[1];
    return Fr{{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w}};
}
AKP_HD void store_fr_g(Fr* p, const Fr& v) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    q[1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}
AKP_HD Niels load_niels(const NielsPad* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    const uint4 v0 = q[0], v1 = q[1], v2 = q[2], v3 = q[3], v4 = q[4], v5 = q[5], v6 = q[6];
    const u32 w[28] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w, v3.x, v3.y,
                       v3.z, v3.w, v4.x, v4.y, v4.z, v4.w, v5.x, v5.y, v5.z, v5.w, v6.x, v6.y, v6.z, v6.w};
    Niels r;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        r.ypx.l[i] = (int32_t)w[i];
        r.ymx.l[i] = (int32_t)w[9 + i];
        r.dxy.l[i] = (int32_t)w[18 + i];
    }
    return r;
}
AKP_HD void store_niels(NielsPad* p, const Niels& n) {
    u32 w[32];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        w[i] = (u32)n.ypx.l[i];
        w[9 + i] = (u32)n.ymx.l[i];
        w[18 + i] = (u32)n.dxy.l[i];
    }
#pragma unroll
    for (int i = 27; i < 32; ++i) w[i] = 0;
    uint4* q = reinterpret_cast<uint4*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) q[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}
AKP_HD Niels niels_of_ext(const Ext& acc) {  // affine point of acc, as a table entry
    const FS zi = f29_inv(acc.Z);
    return niels_from_affine(f29_mul(acc.X, zi), f29_mul(acc.Y, zi));
}

// ---- table construction (one-off per parameter set) ------------------------------------------
AKP_HD Niels te_niels_of_gen(const Fr* __restrict__ gens_affine, size_t g) {
    return niels_from_affine(f29_from_wire<true>(load_fr_g(gens_affine + 2 * g)), f29_from_wire<true>(load_fr_g(gens_affine + 2 * g + 1)));
}
// Pedersen: digit u covers flat generators [uD, uD + D) (clipped to n_gen); entry v in [0, 2^D):
// sum of the generators selected by the bits of v (v = 0 -> identity).
AKP_HD Niels te_pedersen_lut_entry(const Fr* __restrict__ gens_affine /*[N*W][2] wire*/, u32 n_gen, u32 D, u32 idx) {
    const u32 u = idx >> D, v = idx & ((1u << D) - 1u);
    Ext acc = ext_identity();
#pragma unroll 1
    for (u32 b = 0; b < D; ++b) {
        const u32 g = u * D + b;
        if (((v >> b) & 1u) && g < n_gen) acc = te_madd(acc, te_niels_of_gen(gens_affine, g));
    }
    return niels_of_ext(acc);
}
__global__ void te_build_pedersen_lut(const Fr* __restrict__ gens_affine, u32 n_gen, u32 D, u32 n_entries,
                                      TeEntry* __restrict__ lut /*[n_digits][2^D]*/) {
    const u32 idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_entries) return;
    store_niels(lut + idx, te_pedersen_lut_entry(gens_affine, n_gen, D, idx));
}
// Bowe-Hopwood single-chunk table: entry k in [0,4): (k+1) * G[c].
AKP_HD Niels te_bh_lut_entry(const Fr* __restrict__ gens_affine /*[N*W][2] wire*/, u32 idx) {
    const u32 c = idx >> 2, k = idx & 3u;
    const Niels gn = te_niels_of_gen(gens_affine, c);
    Ext acc = ext_identity();
#pragma unroll 1
    for (u32 j = 0; j <= k; ++j) acc = te_madd(acc, gn);
    return niels_of_ext(acc);
}
__global__ void te_build_bh_lut(const Fr* __restrict__ gens_affine, u32 n_gen, TeEntry* __restrict__ lut) {
    const u32 idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_gen * 4u) return;
    store_niels(lut + idx, te_bh_lut_entry(gens_affine, idx));
}
// Bowe-Hopwood group table (G = 2..4 chunks per step): index = k_0 | k_1 << 2 | ... | r_1 << 2G | r_2 << (2G+1) ...
// with r_i = s_i ^ s_0:   (k_0+1) G[Gu] + sum_{i>=1} (-1)^{r_i} (k_i+1) G[Gu+i];   2^(3G-1) entries per group.
AKP_HD Niels te_bh_lutg_entry(const Fr* __restrict__ gens_affine, u32 G, u32 idx) {
    const u32 bits = 3u * G - 1u;
    const u32 u = idx >> bits, v = idx & ((1u << bits) - 1u);
    Ext acc = ext_identity();
#pragma unroll 1
    for (u32 i = 0; i < G; ++i) {
        Niels gn = te_niels_of_gen(gens_affine, (size_t)G * u + i);
        if (i > 0 && ((v >> (2u * G + i - 1u)) & 1u)) gn = niels_neg(gn);
        const u32 k = (v >> (2 * i)) & 3u;
#pragma unroll 1
        for (u32 j = 0; j <= k; ++j) acc = te_madd(acc, gn);
    }
    return niels_of_ext(acc);
}
__global__ void te_build_bh_lutg(const Fr* __restrict__ gens_affine, u32 G, u32 n_entries, TeEntry* __restrict__ lut) {
    const u32 idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_entries) return;
    store_niels(lut + idx, te_bh_lutg_entry(gens_affine, G, idx));
}

// ---- Pedersen, signed-subset table (round 2) ----------------------------------------------------
// A digit of D message bits selects a SUBSET sum of D generators.  With H_g = G_g / 2 (the half of a point of the odd-order
// subgroup: ((r + 1) / 2) * G_g) the subset sum is   sum_b v_b G_b = sum_b H_b + sum_b (2 v_b - 1) H_b = C_u + S_u(v),
// and S_u(~v) = -S_u(v): only the 2^(D-1) entries whose top bit is set are stored, the other half is their negation (swap
// the Niels components).  For the same table size the digit is one bit wider: 74 steps instead of 79 for 4x256.  The
// constants C_u of the windows a message of n_steps digits touches are summed once per possible n_steps (`cprefix`) and are
// what the sum starts from.  Needs every generator in the prime-order subgroup (what setup produces: ark-ec's `rand`
// clears the cofactor); te_halve_generators checks 2 H == G and the host falls back to the plain table otherwise.
AKP_HD u32 te_half_order_bit(u32 i) {  // bit i of (r + 1) / 2, r = order of the prime subgroup of Jubjub
    constexpr u32 K[8] = {0x6b7b965cu, 0x684b872fu, 0xe6640841u, 0x53341049u, 0x809a1d80u, 0x83339d80u, 0x3299d7d4u, 0x073eda75u};
    return (K[i >> 5] >> (i & 31u)) & 1u;
}
// H = G / 2 as a table entry; false when G is not in the prime-order subgroup (2 H != G)
AKP_HD bool te_half_generator(const Fr* __restrict__ gens_affine, size_t g, Niels& out) {
    const FS x = f29_from_wire<true>(load_fr_g(gens_affine + 2 * g)), y = f29_from_wire<true>(load_fr_g(gens_affine + 2 * g + 1));
    const Niels gn = niels_from_affine(x, y);
    Ext acc = ext_identity();
#pragma unroll 1
    for (int i = 250; i >= 0; --i) {  // (r + 1) / 2 has 251 bits
        acc = te_add_ext(acc, acc);
        if (te_half_order_bit((u32)i)) acc = te_madd(acc, gn);
    }
    const Ext dbl = te_add_ext(acc, acc);  // must be G: X = x Z, Y = y Z
    const Fr lx = f29_to_wire(dbl.X), rx = f29_to_wire(f29_mul(x, dbl.Z)), ly = f29_to_wire(dbl.Y), ry = f29_to_wire(f29_mul(y, dbl.Z));
    bool same = true;
#pragma unroll
    for (int i = 0; i < 8; ++i) same = same && lx.l[i] == rx.l[i] && ly.l[i] == ry.l[i];
    out = niels_of_ext(acc);
    return same;
}
__global__ void te_halve_generators(const Fr* __restrict__ gens_affine, u32 n_gen, NielsPad* __restrict__ half, u32* __restrict__ not_in_subgroup) {
    const u32 g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_gen) return;
    Niels h;
    if (!te_half_generator(gens_affine, g, h)) atomicAdd(not_in_subgroup, 1u);
    store_niels(half + g, h);
}
// entry v' in [0, 2^(D-1)) of digit u: S_u(v) for v = v' | 2^(D-1):  sum_b (v_b ? +H : -H)[uD + b] over the generators present
AKP_HD Niels te_pedersen_slut_entry(const NielsPad* __restrict__ half, u32 n_gen, u32 D, u32 idx) {
    const u32 u = idx >> (D - 1u), v = (idx & ((1u << (D - 1u)) - 1u)) | (1u << (D - 1u));
    Ext acc = ext_identity();
#pragma unroll 1
    for (u32 b = 0; b < D; ++b) {
        const u32 g = u * D + b;
        if (g >= n_gen) break;
        const Niels h = load_niels(half + g);
        acc = te_madd(acc, ((v >> b) & 1u) ? h : niels_neg(h));
    }
    return niels_of_ext(acc);
}
__global__ void te_build_pedersen_slut(const NielsPad* __restrict__ half, u32 n_gen, u32 D, u32 n_entries, TeEntry* __restrict__ lut) {
    const u32 idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_entries) return;
    store_niels(lut + idx, te_pedersen_slut_entry(half, n_gen, D, idx));
}
// cprefix[k] = sum of H_g over the generators of digits 0 .. k-1 (k = 0: the identity)
AKP_HD Niels te_pedersen_cprefix_entry(const NielsPad* __restrict__ half, u32 n_gen, u32 D, u32 k) {
    const u32 upto = (k * D < n_gen) ? k * D : n_gen;
    Ext acc = ext_identity();
#pragma unroll 1
    for (u32 g = 0; g < upto; ++g) acc = te_madd(acc, load_niels(half + g));
    return niels_of_ext(acc);
}
__global__ void te_build_pedersen_cprefix(const NielsPad* __restrict__ half, u32 n_gen, u32 D, u32 n_digits, TeEntry* __restrict__ cprefix) {
    const u32 k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > n_digits) return;
    store_niels(cprefix + k, te_pedersen_cprefix_entry(half, n_gen, D, k));
}

// ---- two-part construction of the wide tables (round 4) -----------------------------------------------------------
// The per-entry builders above spend D additions and one inversion on every entry: ~500 field products, 2 s for the 46 GB
// table of 24-bit digits.  A wide entry is the sum of two NARROW ones -- the digit's low bits and its high bits select
// independent subsets of the same generators -- so the wide table is built from two small part tables per digit / group
// (a few thousand entries each, built entry by entry as before) with ONE mixed addition per entry, and the conversion to
// affine shares one inversion among the AKP_TE_BUILD_RUN entries of a lane (Montgomery's trick; the extended point waits
// in the entry's own 128-byte slot meanwhile): ~17 products + 1/16 inversion per entry.
//   Pedersen, signed-subset table of D-bit digits: entry index v' has D - 1 bits (the top bit of the digit is set);
//     lo part: bits [0, k_lo) of the digit, 2^k_lo entries;   hi part: bits [k_lo, D), 2^(D-1-k_lo) entries.
//   Bowe-Hopwood, groups of G chunks: lo part: chunks [0, G_lo) (magnitudes k_0.. and relative signs r_1..: 2^(3 G_lo - 1)
//     entries);   hi part: chunks [G_lo, G) with a relative sign each: 2^(3 (G - G_lo)) entries.
constexpr u32 AKP_TE_BUILD_RUN = 16;
// sum over the bits b in [b0, b1) of digit u of (v_b ? +H : -H)[uD + b]  (generators past n_gen are absent)
AKP_HD Niels te_pedersen_spart_entry(const NielsPad* __restrict__ half, u32 n_gen, u32 D, u32 u, u32 b0, u32 b1, u32 v) {
    Ext acc = ext_identity();
#pragma unroll 1
    for (u32 b = b0; b < b1; ++b) {
        const u32 g = u * D + b;
        if (g >= n_gen) break;
        const Niels h = load_niels(half + g);
        acc = te_madd(acc, ((v >> b) & 1u) ? h : niels_neg(h));
    }
    return niels_of_ext(acc);
}
// lo[u][w]: bits [0, k_lo) = w;   hi[u][w]: bits [k_lo, D - 1) = w, bit D - 1 set
AKP_HD void te_pedersen_sparts_item(const NielsPad* __restrict__ half, u32 n_gen, u32 D, u32 n_digits, u32 k_lo, TeEntry* __restrict__ lo,
                                    TeEntry* __restrict__ hi, u32 idx) {
    const u32 n_lo = n_digits << k_lo, k_hi = D - 1u - k_lo, n_hi = n_digits << k_hi;
    if (idx < n_lo) {
        const u32 u = idx >> k_lo, w = idx & ((1u << k_lo) - 1u);
        store_niels(lo + idx, te_pedersen_spart_entry(half, n_gen, D, u, 0u, k_lo, w));
    } else if (idx - n_lo < n_hi) {
        const u32 j = idx - n_lo, u = j >> k_hi, w = j & ((1u << k_hi) - 1u);
        store_niels(hi + j, te_pedersen_spart_entry(half, n_gen, D, u, k_lo, D, (w << k_lo) | (1u << (D - 1u))));
    }
}
__global__ void te_build_pedersen_sparts(const NielsPad* __restrict__ half, u32 n_gen, u32 D, u32 n_digits, u32 k_lo, TeEntry* __restrict__ lo,
                                         TeEntry* __restrict__ hi) {
    te_pedersen_sparts_item(half, n_gen, D, n_digits, k_lo, lo, hi, blockIdx.x * blockDim.x + threadIdx.x);
}
// sum_{i < cnt} (-1)^{r_i} (k_i + 1) G[first + i],  k_i = bits [2i, 2i + 2) of kbits, r_i = bit i of rbits
AKP_HD Niels te_bh_part_entry(const Fr* __restrict__ gens_affine, size_t first, u32 cnt, u32 kbits, u32 rbits) {
    Ext acc = ext_identity();
#pragma unroll 1
    for (u32 i = 0; i < cnt; ++i) {
        Niels gn = te_niels_of_gen(gens_affine, first + i);
        if ((rbits >> i) & 1u) gn = niels_neg(gn);
        const u32 k = (kbits >> (2u * i)) & 3u;
#pragma unroll 1
        for (u32 j = 0; j <= k; ++j) acc = te_madd(acc, gn);
    }
    return niels_of_ext(acc);
}
// parts of the group table: chunks [Gu, Gu + G_lo) (lo; chunk 0 carries no sign bit) and the G - G_lo chunks after them (hi)
AKP_HD void te_bh_parts_item(const Fr* __restrict__ gens_affine, u32 G, u32 G_lo, u32 n_groups, TeEntry* __restrict__ lo, TeEntry* __restrict__ hi, u32 idx) {
    const u32 G_hi = G - G_lo, lo_bits = 3u * G_lo - 1u, hi_bits = 3u * G_hi;
    const u32 n_lo = n_groups << lo_bits, n_hi = n_groups << hi_bits;
    if (idx < n_lo) {
        const u32 u = idx >> lo_bits, w = idx & ((1u << lo_bits) - 1u);
        store_niels(lo + idx, te_bh_part_entry(gens_affine, (size_t)G * u, G_lo, w & ((1u << (2u * G_lo)) - 1u), (w >> (2u * G_lo)) << 1));
    } else if (idx - n_lo < n_hi) {
        const u32 j = idx - n_lo, u = j >> hi_bits, w = j & ((1u << hi_bits) - 1u);
        store_niels(hi + j, te_bh_part_entry(gens_affine, (size_t)G * u + G_lo, G_hi, w & ((1u << (2u * G_hi)) - 1u), w >> (2u * G_hi)));
    }
}
__global__ void te_build_bh_parts(const Fr* __restrict__ gens_affine, u32 G, u32 G_lo, u32 n_groups, TeEntry* __restrict__ lo, TeEntry* __restrict__ hi) {
    te_bh_parts_item(gens_affine, G, G_lo, n_groups, lo, hi, blockIdx.x * blockDim.x + threadIdx.x);
}
// remainder table of the r chunks starting at chunk `first`: entry[bits] = sum_i (-1)^{s_i} (k_i + 1) G[first + i] (+ tail), the
// chunk bits in message order (k_i = bits [3i, 3i + 2), s_i = bit 3i + 2).  `tail`: the constant of the zero-padded chunks
// behind the data (te_bh_tail_kernel) or NULL.  Entry by entry: at most 2^21 entries, once per message shape.
AKP_HD Niels te_bh_remainder_entry(const Fr* __restrict__ gens_affine, u32 first, u32 r, const TeEntry* __restrict__ tail, u32 idx) {
    Ext acc = ext_identity();
#pragma unroll 1
    for (u32 i = 0; i < r; ++i) {
        Niels gn = te_niels_of_gen(gens_affine, (size_t)first + i);
        if ((idx >> (3u * i + 2u)) & 1u) gn = niels_neg(gn);
        const u32 k = (idx >> (3u * i)) & 3u;
#pragma unroll 1
        for (u32 j = 0; j <= k; ++j) acc = te_madd(acc, gn);
    }
    if (tail) acc = te_madd(acc, load_niels(tail));
    return niels_of_ext(acc);
}
__global__ void te_build_bh_remainder(const Fr* __restrict__ gens_affine, u32 first, u32 r, const TeEntry* __restrict__ tail, u32 n_entries,
                                      TeEntry* __restrict__ out) {
    const u32 idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_entries) return;
    store_niels(out + idx, te_bh_remainder_entry(gens_affine, first, r, tail, idx));
}
// indices of the two parts of wide entry `idx`.  KIND 2: Pedersen signed-subset (W = D, k_lo bits in the lo part);
// KIND 1: Bowe-Hopwood group table (W = G, k_lo = G_lo chunks in the lo part; index layout of te_bh_lutg_entry)
template <int KIND>
AKP_HD void te_build_split(u32 W, u32 k_lo, u32 idx, u32* lo_idx, u32* hi_idx) {
    if (KIND == 2) {
        const u32 u = idx >> (W - 1u), v = idx & ((1u << (W - 1u)) - 1u);
        *lo_idx = (u << k_lo) | (v & ((1u << k_lo) - 1u));
        *hi_idx = (u << (W - 1u - k_lo)) | (v >> k_lo);
    } else {
        const u32 G = W, G_lo = k_lo, G_hi = G - G_lo, bits = 3u * G - 1u;
        const u32 u = idx >> bits, v = idx & ((1u << bits) - 1u);
        const u32 kk = v & ((1u << (2u * G)) - 1u), rr = v >> (2u * G);  // rr: r_1 .. r_{G-1}
        *lo_idx = (u << (3u * G_lo - 1u)) | (kk & ((1u << (2u * G_lo)) - 1u)) | ((rr & ((1u << (G_lo - 1u)) - 1u)) << (2u * G_lo));
        *hi_idx = (u << (3u * G_hi)) | (kk >> (2u * G_lo)) | ((rr >> (G_lo - 1u)) << (2u * G_hi));
    }
}
// lut[idx] = hi part + lo part, affine.  One lane builds AKP_TE_BUILD_RUN entries, `stride` apart starting at `base` (the kernel:
// stride 256 = the workgroup, coalesced across the lanes).
template <int KIND>
AKP_HD void te_build_combine_lane(const TeEntry* __restrict__ lo, const TeEntry* __restrict__ hi, u32 W, u32 k_lo, size_t n_entries,
                                  TeEntry* __restrict__ lut, size_t base, size_t stride) {
    FS pre[AKP_TE_BUILD_RUN];
    FS run = f29_one<true>();
    u32 cnt = 0;
#pragma unroll 1
    for (u32 j = 0; j < AKP_TE_BUILD_RUN; ++j) {
        const size_t e = base + (size_t)j * stride;
        if (e >= n_entries) break;
        u32 li, hi_i;
        te_build_split<KIND>(W, k_lo, (u32)e, &li, &hi_i);
        const Ext s = te_madd(ext_from_niels(load_niels(hi + hi_i)), load_niels(lo + li));
        store_niels(lut + e, Niels{s.X, s.Y, s.Z});  // parked in its own slot until the shared inversion is known
        pre[j] = run;
        run = f29_mul(run, s.Z);
        cnt = j + 1u;
    }
    if (cnt == 0) return;
    FS inv = f29_inv(run);  // Z != 0 always (complete formulas)
#pragma unroll 1
    for (u32 j = cnt; j-- > 0;) {
        const size_t e = base + (size_t)j * stride;
        // (computing the sum a second time instead of parking it -- one write per entry instead of two writes and a read -- builds an idle
        // device's 46 GB table in 58.3 instead of 64.4 ms, but a build that runs BESIDE hashing then competes for the vector ALUs the
        // hashing is bound by: batches beside it took 17 instead of 7-9 ms and the first one 120 ms: profiles/r06_s26, r06_s30)
        const Niels xyz = load_niels(lut + e);
        const FS zi = f29_mul(inv, pre[j]);
        inv = f29_mul(inv, xyz.dxy);
        store_niels(lut + e, niels_from_affine(f29_mul(xyz.ypx, zi), f29_mul(xyz.ymx, zi)));
    }
}
// entries [first, n_entries) in tiles of 256 x AKP_TE_BUILD_RUN; workgroup b takes the tiles b, b + gridDim.x, ...: a grid of one
// workgroup per tile is the fast build, a grid of a few workgroups the POLITE one (round 6: a table built in the background beside
// hashing occupies that many wave slots and no more -- capi_te.hip te_build_wide)
template <int KIND>
__global__ void __launch_bounds__(256) te_build_combine_kernel(const TeEntry* __restrict__ lo, const TeEntry* __restrict__ hi, u32 W, u32 k_lo,
                                                              size_t first, size_t n_entries, TeEntry* __restrict__ lut) {
    constexpr size_t tile = 256u * AKP_TE_BUILD_RUN;
    for (size_t at = first + (size_t)blockIdx.x * tile; at < n_entries; at += (size_t)gridDim.x * tile)
        te_build_combine_lane<KIND>(lo, hi, W, k_lo, n_entries, lut, at + threadIdx.x, 256u);
}
// test build: the wide table against the per-entry definition (canonical values), mismatches counted
template <int KIND>
__global__ void te_check_table_kernel(const void* __restrict__ src /* NielsPad* half (KIND 2) or Fr* generators (KIND 1) */, u32 n_gen, u32 W,
                                      size_t n_entries, size_t first, size_t step, const TeEntry* __restrict__ lut, u32* __restrict__ bad) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t e = first + i * step;
    if (e >= n_entries) return;
    const Niels want = KIND == 2 ? te_pedersen_slut_entry(reinterpret_cast<const NielsPad*>(src), n_gen, W, (u32)e)
                                 : te_bh_lutg_entry(reinterpret_cast<const Fr*>(src), W, (u32)e);
    const Niels got = load_niels(lut + e);
    const Fr a = f29_to_wire(want.ypx), b = f29_to_wire(want.ymx), c = f29_to_wire(want.dxy);
    const Fr x = f29_to_wire(got.ypx), y = f29_to_wire(got.ymx), z = f29_to_wire(got.dxy);
    bool same = true;
#pragma unroll
    for (int k = 0; k < 8; ++k) same = same && a.l[k] == x.l[k] && b.l[k] == y.l[k] && c.l[k] == z.l[k];
    if (!same) atomicAdd(bad, 1u);
}

// ---- message bit access -----------------------------------------------------------------------
// bits [o, o+w) (w <= 24 + 1) of a message of `len` bytes, LSB-first per byte (crh/pedersen/mod.rs:200-209);
// bits past the end read as zero (Pedersen zero padding :91-99 / Bowe-Hopwood chunk padding :131-138).
// Done in two halves for the software pipeline of te_accumulate_item: msg_load issues ONE unconditional, unaligned
// 32-bit load that covers the window (so that nothing waits for it here), msg_combine extracts the bits -- it runs one
// curve addition later.  The load address is pulled back so that the four bytes end inside the message (messages of
// fewer than four bytes are padded by the host, te_crh_dev); a window of w <= 25 bits (24-bit digits, groups of 8 chunks; static_assert
// below) starting at bit (o & 7) of its first
// byte fits the 32 bits, and when the address was pulled back the window reaches past the end of the message, where the
// bits are zero -- exactly what the right shift of the 32-bit word shifts in.
struct MsgRaw {
    u32 word;
};
AKP_HD size_t msg_word_addr(size_t len, size_t o) {
    if (len < 4) return 0;  // host harness only (the device never sees such a length)
    const size_t byte = o >> 3, last = len - 4;
    return byte < last ? byte : last;
}
AKP_HD MsgRaw msg_load(const uint8_t* __restrict__ msg, size_t len, size_t o) {
    if (len == 0) return MsgRaw{0u};
    u32 v;
#if !defined(__HIP_DEVICE_COMPILE__)
    if (len < 4) {  // tests/host_harness calls the per-item code with any length; te_crh_dev pads such messages for the device
        v = 0;
        for (size_t k = 0; k < len; ++k) v |= (u32)msg[k] << (8 * k);
        return MsgRaw{v};
    }
#endif
    __builtin_memcpy(&v, msg + msg_word_addr(len, o), 4);
    return MsgRaw{v};
}
AKP_HD u32 msg_combine(const MsgRaw& r, size_t len, size_t o, u32 w) {
    if (len == 0 || o >= len * 8) return 0u;
    const u32 shift = (u32)(o - 8 * msg_word_addr(len, o));  // 0..7, or up to 31 next to the end of the message
    return (r.word >> shift) & ((1u << w) - 1u);
}
#if defined(__HIPCC__)
// The same message read from the workgroup's LDS image of its messages (te_accumulate_lds_kernel): `img` is a byte-for-byte
// copy of the 16-byte-aligned global range that covers the workgroup's messages, with one pad dword after every 32 dwords
// (dword d of the range lives at img[d + (d >> 5)]) so that lanes whose messages lie 128 bytes apart read different
// banks; `at` is the byte offset of this lane's message inside the range.  The unaligned 32-bit window is two dword
// reads and a funnel shift.
struct MsgLds {
    const u32* img;
    u32 at;
};
__device__ __forceinline__ MsgRaw msg_load(const MsgLds& m, size_t len, size_t o) {
    if (len == 0) return MsgRaw{0u};
    const u32 a = m.at + (u32)msg_word_addr(len, o);
    const u32 d = a >> 2, sh = (a & 3u) * 8u;
    const u32 lo = m.img[d + (d >> 5)], hi = m.img[(d + 1u) + ((d + 1u) >> 5)];
    return MsgRaw{__builtin_amdgcn_alignbit(hi, lo, sh)};
}
#endif

// A message of ANY length, also 1..3 bytes (the ragged batches: every lane has a length of its own, nothing was padded by the host).
// Short messages are assembled from byte loads; everything else is the 32-bit window of msg_load above.
struct MsgAny {
    const uint8_t* p;
};
AKP_HD MsgRaw msg_load(const MsgAny& m, size_t len, size_t o) {
    if (len == 0) return MsgRaw{0u};
    u32 v = 0;
    if (len < 4) {
        for (size_t k = 0; k < len; ++k) v |= (u32)m.p[k] << (8 * k);
        return MsgRaw{v};
    }
    __builtin_memcpy(&v, m.p + msg_word_addr(len, o), 4);
    return MsgRaw{v};
}
constexpr u32 te_shape_max_window_bits = te_shape::MAX_DIGIT > 3 * te_shape::MAX_GROUP ? te_shape::MAX_DIGIT : 3 * te_shape::MAX_GROUP;
// the invariant every window extraction above rests on: a step's bits start at most 7 bits into the 32-bit word
static_assert(te_shape_max_window_bits + 7 <= 32, "a table step's message bits must fit one 32-bit window that starts inside a byte");

// ---- accumulate: one message per lane ------------------------------------------------------------
// A step is split into three stages so that no load is consumed in the stage that issues it (the compiler places the
// wait where the first use is, in program order):
//   te_step_bits   message bits of the step (up to three byte loads)
//   te_step_fetch  table index from those bits, the 128-byte entry load, and the sign of the step -- nothing is computed
//                  on the loaded limbs
//   niels_apply    branch-free negation of the entry (swap y+x / y-x, negate dxy), at the point of use
//  kind 0 (Pedersen): digit = message bits [u*D, u*D + D), entry lut[u << D | digit].
//  kind 1 (Bowe-Hopwood): steps [0, n_groups) are groups of D (= G) chunks from lut (2^(3G-1) entries each),
//                         negated by s_0; steps [n_groups, n_steps) are the left-over single chunks from lut1.
//  kind 2 (Pedersen, signed-subset table): lut[u << (D-1) | ...]; lut1 is the cprefix table the sum starts from.
struct NielsSel {
    Niels q;  // the entry as stored
    u32 neg;  // 1: the step adds -q
};
AKP_HD Niels niels_apply(const NielsSel& s) {
    const int32_t m = -(int32_t)(s.neg & 1u);  // 0 or all ones
    Niels r;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int32_t sw = (s.q.ypx.l[i] ^ s.q.ymx.l[i]) & m;
        r.ypx.l[i] = s.q.ypx.l[i] ^ sw;
        r.ymx.l[i] = s.q.ymx.l[i] ^ sw;
        r.dxy.l[i] = (s.q.dxy.l[i] ^ m) - m;
    }
    return r;
}
// kind 1: `D` packs the group size G (bits 0-7) and the size R of the remainder group (bits 8-15).  R = 0: the chunks a message
// leaves after its last full group are single steps from lut1.  R >= 1: they are ONE step, and lut1 points to the table of
// that remainder: 2^(3R) entries indexed by the 3R message bits as they are (no sign symmetry: the constant of a
// zero-padded tail is folded into every entry, so the step also replaces the tail addition; te_build_bh_remainder).
AKP_HD u32 te_bh_group(u32 D) { return D & 0xffu; }
AKP_HD u32 te_bh_rem(u32 D) { return (D >> 8) & 0xffu; }
// table index of a group of G chunks from its 3G message bits (layout of te_bh_lutg_entry); *s0 = sign bit of chunk 0
AKP_HD u32 te_bh_group_index(u32 bits, u32 G, u32* s0) {
    const u32 s = (bits >> 2) & 1u;
    u32 idx = bits & 3u;
#pragma unroll 1
    for (u32 i = 1; i < G; ++i) {
        idx |= ((bits >> (3u * i)) & 3u) << (2u * i);
        idx |= (((bits >> (3u * i + 2u)) & 1u) ^ s) << (2u * G + i - 1u);
    }
    *s0 = s;
    return idx;
}
// bit offset and width of step u's message bits
template <int KIND>
AKP_HD size_t te_step_offset(u32 D, u32 n_groups, u32 u, u32* width) {
    if (KIND != 1) {
        *width = D;
        return (size_t)u * D;
    }
    const u32 G = te_bh_group(D), R = te_bh_rem(D);  // chunks per group step; chunks of the remainder step
    if (u < n_groups) {
        *width = 3u * G;
        return (size_t)u * 3u * G;
    }
    if (R) {
        *width = 3u * R;
        return (size_t)G * n_groups * 3u;
    }
    *width = 3u;
    return (size_t)(G * n_groups + (u - n_groups)) * 3u;
}
// P + (-1)^neg Q without touching the entry's limbs (they are consumed by the products as loaded): for -Q the roles of
// Y - X and Y + X are exchanged, and e and c change sign:
//   a* = (neg ? Y + X : Y - X) * ymx,  b* = (neg ? Y - X : Y + X) * ypx,  e = +-(b* - a*),  h = b* + a*,  c = +-T * dxy
// SEL = false: plain P + Q.
template <bool SEL>
AKP_HD Ext te_madd_signed(const Ext& p, const Niels& q, u32 neg) {
    const int32_t m = -(int32_t)(neg & 1u);  // 0 or all ones
    FS d = f29_sub(p.Y, p.X), t = f29_add(p.Y, p.X);
    if (SEL) {
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int32_t sw = (d.l[i] ^ t.l[i]) & m;
            d.l[i] ^= sw;
            t.l[i] ^= sw;
        }
    }
    const FS a = f29_mul(d, q.ymx);
    const FS b = f29_mul(t, q.ypx);
    FS c = f29_mul(p.T, q.dxy);
    FS e = f29_sub(b, a);
    if (SEL) {
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            c.l[i] = (c.l[i] ^ m) - m;
            e.l[i] = (e.l[i] ^ m) - m;
        }
    }
    const FS f = f29_sub(p.Z, c), g = f29_weak_norm(f29_add(p.Z, c)), h = f29_add(b, a);
    Ext r;
    r.X = f29_mul(e, f);
    r.Y = f29_mul(g, h);
    r.Z = f29_mul(f, g);
    r.T = f29_mul(e, h);
    return r;
}
// A table entry on its way in: address, sign of the step, and the seven 16-byte pieces of the 128-byte line.
constexpr int AKP_TE_ENTRY_VEC = 7;  // 16-byte pieces of a 128-byte entry that carry data
struct NielsFetch {
    const TeEntry* base;  // lut or lut1 (kept as it is: a pointer that went through an asm statement would lose its
    u32 idx;              // address space and turn the loads into flat loads), entry index
    u32 neg;
    uint4 v[AKP_TE_ENTRY_VEC];
};
AKP_HD void te_fetch_all(NielsFetch& f) {
    const uint4* q = reinterpret_cast<const uint4*>(f.base + f.idx);
#pragma unroll
    for (int k = 0; k < AKP_TE_ENTRY_VEC; ++k) f.v[k] = q[k];
}
AKP_HD Niels niels_of_fetch(const NielsFetch& f) {
    const u32 w[28] = {f.v[0].x, f.v[0].y, f.v[0].z, f.v[0].w, f.v[1].x, f.v[1].y, f.v[1].z, f.v[1].w, f.v[2].x, f.v[2].y,
                       f.v[2].z, f.v[2].w, f.v[3].x, f.v[3].y, f.v[3].z, f.v[3].w, f.v[4].x, f.v[4].y, f.v[4].z, f.v[4].w,
                       f.v[5].x, f.v[5].y, f.v[5].z, f.v[5].w, f.v[6].x, f.v[6].y, f.v[6].z, f.v[6].w};
    Niels r;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        r.ypx.l[i] = (int32_t)w[i];
        r.ymx.l[i] = (int32_t)w[9 + i];
        r.dxy.l[i] = (int32_t)w[18 + i];
    }
    return r;
}
template <int KIND, class M = const uint8_t*>
AKP_HD MsgRaw te_step_bits(const M& msg, size_t msg_len, u32 D, u32 n_groups, u32 u) {
    u32 w;
    return msg_load(msg, msg_len, te_step_offset<KIND>(D, n_groups, u, &w));
}
template <int KIND>
AKP_HD NielsSel te_step_fetch(const TeEntry* __restrict__ lut, const TeEntry* __restrict__ lut1, const MsgRaw& raw, size_t msg_len, u32 D,
                              u32 n_groups, u32 u) {
    u32 width;
    const size_t off = te_step_offset<KIND>(D, n_groups, u, &width);
    const u32 bits = msg_combine(raw, msg_len, off, width);
    if (KIND == 0) return NielsSel{load_niels(lut + (((size_t)u << D) | bits)), 0u};
    if (KIND == 2) {  // signed-subset table: top bit set -> entry as stored, clear -> the entry of the complement, negated
        const u32 half_mask = (1u << (D - 1u)) - 1u;
        const u32 top = (bits >> (D - 1u)) & 1u;
        return NielsSel{load_niels(lut + (((size_t)u << (D - 1u)) | ((top ? bits : ~bits) & half_mask))), top ^ 1u};
    }
    const u32 G = te_bh_group(D), R = te_bh_rem(D);
    if (u < n_groups) {
        u32 s0;
        const u32 idx = te_bh_group_index(bits, G, &s0);
        return NielsSel{load_niels(lut + ((size_t)u << (3u * G - 1u)) + idx), s0};
    }
    if (R) return NielsSel{load_niels(lut1 + bits), 0u};
    const u32 c = G * n_groups + (u - n_groups);
    return NielsSel{load_niels(lut1 + (size_t)c * 4u + (bits & 3u)), (bits >> 2) & 1u};
}
// the same index computation without the load: address of the entry and sign of the step
template <int KIND>
AKP_HD void te_step_address(const TeEntry* __restrict__ lut, const TeEntry* __restrict__ lut1, const MsgRaw& raw, size_t msg_len, u32 D,
                            u32 n_groups, u32 u, NielsFetch& f) {
    u32 width;
    const size_t off = te_step_offset<KIND>(D, n_groups, u, &width);
    const u32 bits = msg_combine(raw, msg_len, off, width);
    f.base = lut;
    if (KIND == 0) {
        f.idx = (u << D) | bits;
        f.neg = 0u;
    } else if (KIND == 2) {
        const u32 half_mask = (1u << (D - 1u)) - 1u;
        const u32 top = (bits >> (D - 1u)) & 1u;
        f.idx = (u << (D - 1u)) | ((top ? bits : ~bits) & half_mask);
        f.neg = top ^ 1u;
    } else {
        const u32 G = te_bh_group(D), R = te_bh_rem(D);
        if (u < n_groups) {
            u32 s0;
            f.idx = (u << (3u * G - 1u)) + te_bh_group_index(bits, G, &s0);
            f.neg = s0;
        } else if (R) {
            f.base = lut1;
            f.idx = bits;
            f.neg = 0u;
        } else {
            const u32 c = G * n_groups + (u - n_groups);
            f.base = lut1;
            f.idx = c * 4u + (bits & 3u);
            f.neg = (bits >> 2) & 1u;
        }
    }
}
template <int KIND, class M = const uint8_t*>
AKP_HD Niels te_step_entry(const TeEntry* __restrict__ lut, const TeEntry* __restrict__ lut1, const M& msg,
                           size_t msg_len, u32 D, u32 n_groups, u32 u) {
    return niels_apply(te_step_fetch<KIND>(lut, lut1, te_step_bits<KIND>(msg, msg_len, D, n_groups, u), msg_len, D, n_groups, u));
}
// Scheduling fence of the pipeline: the loaded message bytes may not be touched before the addition that was issued after
// them has produced `after` (the scheduler would otherwise hoist the cheap bit extraction -- and with it the wait for the
// loads -- above the addition).  Emits no instruction.
AKP_HD void te_consume_after(MsgRaw& r, const Ext& after) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(r.word) : "v"(after.X.l[8]), "v"(after.Y.l[8]), "v"(after.Z.l[8]), "v"(after.T.l[8]));
#else
    (void)r;
    (void)after;
#endif
}
template <int KIND, class M = const uint8_t*>
AKP_HD Ext te_accumulate_item(const TeEntry* __restrict__ lut, const TeEntry* __restrict__ lut1, const M& msg,
                              size_t msg_len, u32 D, u32 n_groups, u32 n_steps) {
    // Software pipeline, two steps per iteration (two entry buffers, no register copies to rotate them).  Before the
    // addition of step u starts, the seven pieces of step u + 1's table line and the message bytes of step u + 2 are
    // requested; they are consumed one whole addition (~1400 instructions) later.  (Spreading the loads over the products
    // of the addition was measured slower than this burst: 3.51 vs 3.36 ms for 2^20 Pedersen hashes, profiles/r02_s24.)  The sum starts as a table entry (no addition): the first step's entry, or (kind 2)
    // cprefix[n_steps].
    Ext acc;
    u32 u;
    if (KIND == 2) {
        acc = ext_from_niels(load_niels(lut1 + n_steps));
        u = 0;
        if (n_steps == 0) return acc;
    } else {
        if (n_steps == 0) return ext_identity();
        acc = ext_from_niels(te_step_entry<KIND>(lut, lut1, msg, msg_len, D, n_groups, 0));
        u = 1;
        if (n_steps == 1) return acc;
    }
    const u32 last = n_steps - 1u;
    NielsFetch f0, f1;
    te_step_address<KIND>(lut, lut1, te_step_bits<KIND>(msg, msg_len, D, n_groups, u), msg_len, D, n_groups, u, f0);
    te_fetch_all(f0);
    MsgRaw nb = te_step_bits<KIND>(msg, msg_len, D, n_groups, u + 1 <= last ? u + 1 : last);
#pragma unroll 1
    for (; u + 2 <= n_steps; u += 2) {
        const u32 u2 = u + 2 <= last ? u + 2 : last, u3 = u + 3 <= last ? u + 3 : last;
        te_step_address<KIND>(lut, lut1, nb, msg_len, D, n_groups, u + 1, f1);  // u + 1 <= last here
        te_fetch_all(f1);
        nb = te_step_bits<KIND>(msg, msg_len, D, n_groups, u2);
        acc = te_madd_signed<KIND != 0>(acc, niels_of_fetch(f0), f0.neg);
        te_consume_after(nb, acc);
        te_step_address<KIND>(lut, lut1, nb, msg_len, D, n_groups, u2, f0);
        te_fetch_all(f0);
        nb = te_step_bits<KIND>(msg, msg_len, D, n_groups, u3);
        acc = te_madd_signed<KIND != 0>(acc, niels_of_fetch(f1), f1.neg);
        te_consume_after(nb, acc);
    }
    if (u < n_steps) acc = te_madd_signed<KIND != 0>(acc, niels_of_fetch(f0), f0.neg);
    return acc;
}
// partial sum over steps first, first + stride, ... (the whole message for first = 0, stride = 1)
template <int KIND>
AKP_HD Ext te_accumulate_strided(const TeEntry* __restrict__ lut, const TeEntry* __restrict__ lut1, const uint8_t* __restrict__ msg,
                                 size_t msg_len, u32 D, u32 n_groups, u32 n_steps, u32 first, u32 stride) {
    Ext acc;
    u32 start;
    if (KIND == 2 && first == 0) {  // the partial sum that starts at step 0 also carries the constant of the signed table
        acc = ext_from_niels(load_niels(lut1 + n_steps));
        start = 0;
    } else {
        if (first >= n_steps) return ext_identity();
        acc = ext_from_niels(te_step_entry<KIND>(lut, lut1, msg, msg_len, D, n_groups, first));  // first term: no addition
        start = first + stride;
    }
    if (start >= n_steps) return acc;
    NielsSel q0 = te_step_fetch<KIND>(lut, lut1, te_step_bits<KIND>(msg, msg_len, D, n_groups, start), msg_len, D, n_groups, start);
#pragma unroll 1
    for (u32 u = start; u < n_steps; u += stride) {
        const u32 nxt = (u + stride < n_steps) ? u + stride : u;
        const NielsSel q1 = te_step_fetch<KIND>(lut, lut1, te_step_bits<KIND>(msg, msg_len, D, n_groups, nxt), msg_len, D, n_groups, nxt);  // fetched ahead of the addition
        acc = te_madd(acc, niels_apply(q0));
        q0 = q1;
    }
    return acc;
}
// writes the extended-coordinate sum (X, Y, Z), internal form, to xyz[idx*3 ..]
#ifndef AKP_TE_MIN_WAVES
#define AKP_TE_MIN_WAVES 1  // waves per SIMD the register allocation must allow.  4 (<= 128 VGPRs; the signed-table kernel then keeps 21 loop-invariant values in scratch) measured the same as 1 (155 VGPRs, 3 waves): profiles/r02_s23.  Build-time A/B: make EXTRA=-DAKP_TE_MIN_WAVES=4
#endif
// `stride`: bytes between consecutive messages (>= msg_len; the bytes past msg_len are never read).  `tail` (may be null):
// one more table entry every sum ends with -- the constant contribution of a zero-padded Bowe-Hopwood tail (a zero chunk
// selects +g, crh/bowe_hopwood/mod.rs:167), see te_crh_dev.
template <int KIND>
__global__ void __launch_bounds__(256, AKP_TE_MIN_WAVES) te_accumulate_kernel(const TeEntry* __restrict__ lut, const TeEntry* __restrict__ lut1,
                                                           const uint8_t* __restrict__ msgs, size_t msg_len, size_t stride, u32 D, u32 n_groups,
                                                           u32 n_steps, const TeEntry* __restrict__ tail, F29Pad* __restrict__ xyz, size_t n) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    Ext acc = te_accumulate_item<KIND>(lut, lut1, msgs + idx * stride, msg_len, D, n_groups, n_steps);
    if (tail) acc = te_madd(acc, load_niels(tail));
    f29_store_pad(xyz + idx * 3, acc.X);
    f29_store_pad(xyz + idx * 3 + 1, acc.Y);
    f29_store_pad(xyz + idx * 3 + 2, acc.Z);
}
// ---- ragged batches: every item has its own length (round 5) ---------------------------------------------------------------------
// The reference hashes each input with ITS length (crh/pedersen/mod.rs:82-99 pads each input to the window; crh/bowe_hopwood/
// mod.rs:131-138 pads each input to a multiple of 3 bits; MerkleTree::new maps LeafHash::evaluate over any iterator of leaves,
// merkle_tree/mod.rs:411-422).  Item i is bytes [offsets[i], offsets[i+1]) of `msgs`.  Its table steps follow from its length
// exactly as te_shape.hpp computes them for a uniform batch -- Bowe-Hopwood: full groups from the wide table, the chunks left
// over as single steps from the one-chunk table (no remainder tables: those belong to ONE length) -- and te_accumulate_item runs
// with per-lane step counts.  `order` (may be null): the launch order of the items, sorted by step count so that the 64 lanes of
// a wave finish together (ragged_sort.hpp); results are stored by ITEM index either way.  `units_built`: digits / groups the wide
// table covers (the host built it for the longest item; the clamp keeps a lying max_len from reading past the table).
template <int KIND>
AKP_HD void te_item_steps(u32 n_gen, u32 D, size_t len, u32 units_built, u32* groups, u32* steps) {
    if (KIND != 1) {
        const size_t used = len * 8 < n_gen ? len * 8 : n_gen;
        u32 st = (u32)((used + D - 1) / D);
        if (KIND == 2 && st > units_built) st = units_built;
        *groups = 0;
        *steps = st;
        return;
    }
    const size_t ch = (len * 8 + 2) / 3;
    const u32 chunks = (u32)(ch < n_gen ? ch : n_gen);
    const u32 G = te_bh_group(D);
    if (G > 1) {
        u32 g = chunks / G;
        if (g > units_built) g = units_built;
        *groups = g;
        *steps = g + (chunks - g * G < G ? chunks - g * G : chunks % G);
    } else {
        *groups = 0;
        *steps = chunks;
    }
}
template <int KIND>
__global__ void __launch_bounds__(256, AKP_TE_MIN_WAVES) te_accumulate_ragged_kernel(const TeEntry* __restrict__ lut, const TeEntry* __restrict__ lut1,
                                                           const uint8_t* __restrict__ msgs, const uint64_t* __restrict__ offsets,
                                                           const u32* __restrict__ order, u32 D, u32 n_gen, u32 units_built,
                                                           u32 max_len, F29Pad* __restrict__ xyz, size_t n) {
    const size_t slot = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= n) return;
    const size_t idx = order ? order[slot] : slot;
    const uint64_t off = offsets[idx], end = offsets[idx + 1];
    // device-resident offsets are the caller's: a pair that DECREASES is the empty message (not a length of ~2^64), an item longer than
    // the caller's own bound is cut at that bound -- the bytes read are [off, off + min(len, max_len)) and nothing else
    const size_t len = end <= off ? 0 : (end - off < max_len ? (size_t)(end - off) : (size_t)max_len);
    u32 groups, steps;
    te_item_steps<KIND>(n_gen, D, len, units_built, &groups, &steps);
    const MsgAny m{msgs + off};
    const Ext acc = te_accumulate_item<KIND>(lut, lut1, m, len, D, groups, steps);
    f29_store_pad(xyz + idx * 3, acc.X);
    f29_store_pad(xyz + idx * 3 + 1, acc.Y);
    f29_store_pad(xyz + idx * 3 + 2, acc.Z);
}
#if defined(__HIPCC__)
// The same with the workgroup's messages staged through LDS (round 4).  The kernel above reads the message bits of a step with
// one 32-bit load per lane at a `stride`-byte pitch: 64 distinct cache lines per load instruction, 64 times per message,
// competing with the table lines for the vector L1 -- and impossible over PCIe.  Here the workgroup copies the byte range
// that holds its blockDim.x messages ONCE, with coalesced 16-byte loads (the range is widened to 16-byte boundaries: never
// across a page), into an LDS image (layout: MsgLds) and every later bit window is two ds_read_b32.  Because each message
// byte is now read exactly once, `msgs` may be the device alias of pinned / registered HOST memory: the host-pointer
// entry point launches this kernel directly on the caller's buffer (zero copy, no staging copy, no second stream).
// Dynamic LDS: te_lds_image_bytes(blockDim.x, msg_len, stride).
AKP_HD size_t te_lds_image_bytes(size_t block, size_t msg_len, size_t stride) {
    const size_t chunks = (15 + (block - 1) * stride + msg_len + 15) / 16;  // worst misalignment of the range's first byte
    const size_t dwords = chunks * 4;
    return (dwords + (dwords >> 5) + 4) * 4;
}
// GATED (round 5, the pinned host path): the launch covers the WHOLE batch while its messages are still arriving by DMA, chunk after
// chunk.  Workgroup b belongs to chunk chunk_of[b / wg_per_granule]; before it touches its messages, thread 0 polls gate_flags[chunk] -- a word in
// FINE-GRAINED DEVICE memory that hipStreamWriteValue32 on the copy stream sets to `epoch` behind the chunk's copy -- and when its
// digests are stored the workgroup writes `epoch` to done[b] in pinned HOST memory (a plain posted write; the host thread releases the
// chunk's copy-out when all its words are there).  Measured preconditions (tools/persist_probe.hip,
// profiles/r05_s8): the polls must stay on the device (polling host memory competes over PCIe with the very copies the workgroups wait
// for) and the kernel must leave wave slots free (a copy / write-value needs one: with every slot spinning nothing arrives) --
// this kernel holds 3 waves per SIMD.  The spin is bounded: on a timeout the workgroup reports through *gate_err and leaves, the
// host falls back to the chunked launches.
struct TeGate {
    const u32* flags;   // [n_chunks], fine-grained device memory
    u32* done;          // [grid], pinned host memory (device alias)
    u32* err;           // pinned host word (device alias): set by a workgroup that gave up
    u32 epoch, wg_per_granule, spin_limit;
    uint8_t chunk_of[64];  // chunk of every granule of wg_per_granule workgroups: the chunks of one launch differ in size (small ones first and last)
    u32 poll_sleep;     // extra s_sleep(127) (~4 us each at 2 GHz) between two polls of a waiting workgroup
    // the workgroup also finishes its digests (projective -> affine with ONE inversion per workgroup) and writes them to
    // `out` (`fe` Fr per digest: x, or x and y).  The 9 x 512-dword product tree takes the place of the message image, which is dead
    // by then: the launch needs max(image, 18 KB) of LDS
    Fr* out;
    u32 fe;
    unsigned long long* stamps;  // test build: [2 * grid] wall_clock64 at gate-open and at the end of every workgroup (nullptr: none)
};
// One inversion for the 256 sums of a workgroup: a product tree over the Z coordinates in LDS (up-sweep: 8 levels of pairwise products),
// one inversion of the root by lane 0, and the down-sweep that turns every node's product into its inverse (inv_left = inv_parent *
// prod_right and vice versa), in place -- a child's slot is read by its parent's lane only and rewritten by that lane, its own
// children still hold their products when their turn comes.  Node n, limb i lives at tree[i * 512 + n]: consecutive lanes touch
// consecutive banks.  Cost per workgroup: wave 0 ~ 8 + 16 products + the 16 k-instruction inversion, the other waves wait at the
// barriers (their SIMDs run the other workgroups of the CU); +10 % on the accumulate work against +15 % for the eight separate
// finalize passes of a pinned batch -- and no second kernel that has to fight the accumulate kernel for issue slots.
#if defined(__HIPCC__)
__device__ __forceinline__ FS te_tree_load(const u32* tree, u32 node) {
    FS r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = (int32_t)tree[i * 512 + node];
    return r;
}
__device__ __forceinline__ void te_tree_store(u32* tree, u32 node, const FS& v) {
#pragma unroll
    for (int i = 0; i < 9; ++i) tree[i * 512 + node] = (u32)v.l[i];
}
// all 256 threads of the workgroup call this; returns 1 / z of the calling lane's z (in the form f29_mul expects, as f29_inv does)
__device__ __forceinline__ FS te_workgroup_inverse(u32* tree, const FS& z) {
    te_tree_store(tree, 255u + threadIdx.x, z);
    __syncthreads();
    // the tree work (24 products and the inversion, all on the lowest lanes) rotates over the four waves with the workgroup index: with
    // wave 0 always in charge, the SIMD that hosts the wave 0 of every workgroup did 84 k instructions per workgroup against 64 k
    // on the others and set the pace of the CU (0.45 instead of 0.35 ms per 2^17-message chunk, profiles/r05_s13)
    const u32 tid = (threadIdx.x + ((blockIdx.x & 3u) << 6)) & 255u;
#pragma unroll 1
    for (u32 count = 128; count >= 1; count >>= 1) {
        if (tid < count) {
            const u32 node = count - 1u + tid;
            te_tree_store(tree, node, f29_mul(te_tree_load(tree, 2u * node + 1u), te_tree_load(tree, 2u * node + 2u)));
        }
        __syncthreads();
    }
    if (tid == 0) te_tree_store(tree, 0u, f29_inv(te_tree_load(tree, 0u)));
    __syncthreads();
#pragma unroll 1
    for (u32 count = 1; count <= 128; count <<= 1) {
        if (tid < count) {
            const u32 node = count - 1u + tid;
            const FS inv = te_tree_load(tree, node), a = te_tree_load(tree, 2u * node + 1u), b = te_tree_load(tree, 2u * node + 2u);
            te_tree_store(tree, 2u * node + 1u, f29_mul(inv, b));
            te_tree_store(tree, 2u * node + 2u, f29_mul(inv, a));
        }
        __syncthreads();
    }
    return te_tree_load(tree, 255u + threadIdx.x);
}
#endif
// the image of `cnt` messages starting at `g0` (pitch `stride`), padded against bank conflicts; returns the misalignment of g0
__device__ __forceinline__ u32 te_load_msg_image(u32* image, const uint8_t* g0, size_t cnt, size_t msg_len, size_t stride) {
    const u32 mis = (u32)((uintptr_t)g0 & 15u);
    const uint4* a0 = reinterpret_cast<const uint4*>(g0 - mis);
    const u32 chunks = (u32)((mis + (cnt - 1) * stride + msg_len + 15) >> 4);
    for (u32 k = threadIdx.x; k < chunks; k += blockDim.x) {
        const uint4 v = a0[k];
        u32* w = image + 4u * k + (k >> 3);  // dwords 4k .. 4k+3 share one 32-dword group: one pad offset
        w[0] = v.x;
        w[1] = v.y;
        w[2] = v.z;
        w[3] = v.w;
    }
    return mis;
}
// FUSED (gated launches only): the workgroup hashes TE_FUSED_ITEMS x 256 messages -- lane t takes messages first + t and
// first + 256 + t, one image after the other -- and finishes all of them with ONE inversion: the lane multiplies its Z coordinates,
// the product tree inverts the 256 products, 1 / Z1 = inv * Z2 and 1 / Z2 = inv * Z1.  The inversion (16 k instructions on one lane) and
// the tree (24 products) are +9 % on a workgroup of 256 points and +4.5 % on one of 512 (profiles/r05_s16).
#ifndef TE_FUSED_ITEMS
#define TE_FUSED_ITEMS 2  // 1: one point per lane (build-time A/B arm)
#endif
template <int KIND, bool GATED>  // GATED: the gated launch of the pinned host path, which also finishes its digests
__device__ __forceinline__ void te_accumulate_lds_body(const TeEntry* __restrict__ lut, const TeEntry* __restrict__ lut1,
                                                       const uint8_t* __restrict__ msgs, size_t msg_len, size_t stride, u32 D, u32 n_groups, u32 n_steps,
                                                       const TeEntry* __restrict__ tail, F29Pad* __restrict__ xyz, size_t n, const TeGate& gate) {
    extern __shared__ u32 te_msg_image[];
    if (GATED) {
        __shared__ u32 gate_open;
        if (threadIdx.x == 0) {
            const u32 chunk = gate.chunk_of[blockIdx.x / gate.wg_per_granule];
            u32 it = 0;
            // RELAXED polls and ONE acquire fence behind the last: an acquire load at system scope is followed by a cache invalidate
            // (buffer_inv sc0 sc1) -- issued every half microsecond by some 700 waiting workgroups it took the table lines of the
            // RUNNING workgroups with it (profiles/r05_s16: 3.64 -> 3.47 ms per 2^20 pinned Pedersen hashes)
            while (__hip_atomic_load(gate.flags + chunk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != gate.epoch && ++it < gate.spin_limit) {
                __builtin_amdgcn_s_sleep(16);
                for (u32 q = 0; q < gate.poll_sleep; ++q) __builtin_amdgcn_s_sleep(127);
            }
            __atomic_thread_fence(__ATOMIC_ACQUIRE);  // system scope: the chunk's bytes, written by the copy engine, are what the loads below see
            gate_open = it < gate.spin_limit;
            if (!gate_open) __hip_atomic_store(gate.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (gate.stamps) gate.stamps[2 * (size_t)blockIdx.x] = wall_clock64();
        }
        __syncthreads();
        if (!gate_open) return;
    }
    if (GATED) {
        const size_t base = (size_t)blockIdx.x * (TE_FUSED_ITEMS * 256);
        Ext acc[TE_FUSED_ITEMS];
#pragma unroll
        for (int h = 0; h < TE_FUSED_ITEMS; ++h) {
            const size_t first = base + (size_t)h * 256;
            // wave priority follows the progress of the workgroup (waiting 0 < first image 1 < second image 2 < the inversion 3): the
            // workgroup closest to its completion word goes first on its SIMD, the chunks' copy-outs leave earlier (1-4 % of a pinned
            // call, profiles/r06_s45)
            if (h == 0) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(2);
            acc[h] = ext_identity();  // lanes past the end of the batch take part in the product tree with Z = 1
            if (first < n) {          // uniform over the workgroup
                if (h) __syncthreads();  // every wave has read its last byte of the previous image
                const size_t cnt = n - first < 256 ? n - first : 256;
                const u32 mis = te_load_msg_image(te_msg_image, msgs + first * stride, cnt, msg_len, stride);
                __syncthreads();
                if (first + threadIdx.x < n) {
                    const MsgLds m{te_msg_image, mis + (u32)(threadIdx.x * stride)};
                    acc[h] = te_accumulate_item<KIND>(lut, lut1, m, msg_len, D, n_groups, n_steps);
                    if (tail) acc[h] = te_madd(acc[h], load_niels(tail));
                }
            }
        }
        static_assert(TE_FUSED_ITEMS == 1 || TE_FUSED_ITEMS == 2, "the pairing below is written for two points per lane");
        __syncthreads();  // the image is dead: it becomes the product tree
        __builtin_amdgcn_s_setprio(3);
        const FS zi = te_workgroup_inverse(te_msg_image, TE_FUSED_ITEMS == 2 ? f29_mul(acc[0].Z, acc[TE_FUSED_ITEMS - 1].Z) : acc[0].Z);
#pragma unroll
        for (int h = 0; h < TE_FUSED_ITEMS; ++h) {
            const size_t idx = base + (size_t)h * 256 + threadIdx.x;
            if (idx < n) {
                const FS zh = TE_FUSED_ITEMS == 2 ? f29_mul(zi, acc[TE_FUSED_ITEMS - 1 - h].Z) : zi;  // 1 / Z of point h
                store_fr_g(gate.out + idx * gate.fe, f29_to_wire(f29_mul(acc[h].X, zh)));
                if (gate.fe == 2) store_fr_g(gate.out + idx * 2 + 1, f29_to_wire(f29_mul(acc[h].Y, zh)));
            }
        }
    } else {
        const size_t first = (size_t)blockIdx.x * blockDim.x;
        const size_t cnt = n - first < blockDim.x ? n - first : blockDim.x;  // the grid covers n: cnt >= 1
        const u32 mis = te_load_msg_image(te_msg_image, msgs + first * stride, cnt, msg_len, stride);
        __syncthreads();
        const size_t idx = first + threadIdx.x;
        if (idx < n) {
            const MsgLds m{te_msg_image, mis + (u32)(threadIdx.x * stride)};
            Ext acc = te_accumulate_item<KIND>(lut, lut1, m, msg_len, D, n_groups, n_steps);
            if (tail) acc = te_madd(acc, load_niels(tail));
            f29_store_pad(xyz + idx * 3, acc.X);
            f29_store_pad(xyz + idx * 3 + 1, acc.Y);
            f29_store_pad(xyz + idx * 3 + 2, acc.Z);
        }
    }
    if (GATED) {
        __threadfence();  // the results of this workgroup are visible device-wide (a copy engine, or another kernel on another XCD, reads them next)
        __syncthreads();
        if (threadIdx.x == 0) {
            if (gate.stamps) gate.stamps[2 * (size_t)blockIdx.x + 1] = wall_clock64();
            __hip_atomic_store(gate.done + blockIdx.x, gate.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
template <int KIND>
__global__ void __launch_bounds__(256, AKP_TE_MIN_WAVES) te_accumulate_lds_kernel(const TeEntry* __restrict__ lut, const TeEntry* __restrict__ lut1,
                                                           const uint8_t* __restrict__ msgs, size_t msg_len, size_t stride, u32 D, u32 n_groups,
                                                           u32 n_steps, const TeEntry* __restrict__ tail, F29Pad* __restrict__ xyz, size_t n) {
    te_accumulate_lds_body<KIND, false>(lut, lut1, msgs, msg_len, stride, D, n_groups, n_steps, tail, xyz, n, TeGate{});
}
template <int KIND>
__global__ void __launch_bounds__(256, AKP_TE_MIN_WAVES) te_accumulate_lds_gated_fused_kernel(const TeEntry* __restrict__ lut, const TeEntry* __restrict__ lut1,
                                                           const uint8_t* __restrict__ msgs, size_t msg_len, size_t stride, u32 D, u32 n_groups,
                                                           u32 n_steps, const TeEntry* __restrict__ tail, size_t n, TeGate gate) {
    te_accumulate_lds_body<KIND, true>(lut, lut1, msgs, msg_len, stride, D, n_groups, n_steps, tail, nullptr, n, gate);
}
#endif
// sum of the single-chunk entries 1 * G[c], c in [from, to): the constant of a zero tail (one thread; once per parameter set)
__global__ void te_bh_tail_kernel(const TeEntry* __restrict__ lut1, u32 from, u32 to, TeEntry* __restrict__ out) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    Ext acc = ext_from_niels(load_niels(lut1 + (size_t)from * 4u));
#pragma unroll 1
    for (u32 c = from + 1; c < to; ++c) acc = te_madd(acc, load_niels(lut1 + (size_t)c * 4u));
    store_niels(out, niels_of_ext(acc));
}

// ---- projective -> affine with shared inversions ---------------------------------------------------
// Lane l handles elements l, l + L, l + 2L, ... (L = total lanes, coalesced): forward prefix products of Z
// into `prefix`, one inversion, backward pass emitting x (and y) in wire format.
// out: Pedersen n x (x, y); Bowe-Hopwood n x (x).
template <int KIND>
AKP_HD void te_finalize_lane(const F29Pad* __restrict__ xyz, F29Pad* __restrict__ prefix, Fr* __restrict__ out, size_t n,
                             size_t lanes, size_t l) {
    // Both passes are chains of dependent products with one wave per SIMD, so a load that is issued where its value is needed
    // costs its full latency 2 x (elements per lane) times: every load is issued one iteration ahead of its use.
    FS run = f29_one<true>();
    size_t cnt = 0;
    if (l >= n) return;
    FS z = f29_load_pad<true>(xyz + l * 3 + 2);
#pragma unroll 1
    for (size_t e = l; e < n; e += lanes, ++cnt) {
        FS zn = z;
        if (e + lanes < n) zn = f29_load_pad<true>(xyz + (e + lanes) * 3 + 2);
        f29_store_pad(prefix + e, run);  // product of the earlier Z's of this lane
        run = f29_mul(run, z);
        z = zn;
    }
    FS inv = f29_inv(run);  // Z != 0 always (complete formulas)
    size_t k = cnt - 1, e = l + k * lanes;
    FS p = f29_load_pad<true>(prefix + e), x = f29_load_pad<true>(xyz + e * 3), y = x;
    z = f29_load_pad<true>(xyz + e * 3 + 2);
    if (KIND != 1) y = f29_load_pad<true>(xyz + e * 3 + 1);
#pragma unroll 1
    for (;;) {
        FS pn = p, zn = z, xn = x, yn = y;
        if (k > 0) {
            const size_t en = e - lanes;
            pn = f29_load_pad<true>(prefix + en);
            zn = f29_load_pad<true>(xyz + en * 3 + 2);
            xn = f29_load_pad<true>(xyz + en * 3);
            if (KIND != 1) yn = f29_load_pad<true>(xyz + en * 3 + 1);
        }
        const FS zi = f29_mul(inv, p);
        inv = f29_mul(inv, z);
        if (KIND != 1) {
            store_fr_g(out + e * 2, f29_to_wire(f29_mul(x, zi)));
            store_fr_g(out + e * 2 + 1, f29_to_wire(f29_mul(y, zi)));
        } else {
            store_fr_g(out + e, f29_to_wire(f29_mul(x, zi)));
        }
        if (k == 0) break;
        --k;
        e -= lanes;
        p = pn;
        z = zn;
        x = xn;
        y = yn;
    }
}
template <int KIND>
__global__ void __launch_bounds__(256) te_finalize_kernel(const F29Pad* __restrict__ xyz, F29Pad* __restrict__ prefix,
                                                         Fr* __restrict__ out, size_t n, size_t lanes) {
    const size_t l = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= lanes || l >= n) return;
#if defined(__HIP_DEVICE_COMPILE__)
    // a pass of few waves and long dependent chains: when it shares its SIMDs with an accumulate kernel (the chunks of the pinned host
    // path) its instructions go first -- beside three accumulate waves per SIMD a chunk's pass took 1.1 ms instead of 0.2 ms and the
    // eight serialised passes were the floor of the call (profiles/r05_s8/timeline_gated1_hbm.txt); alone it makes no difference
    __builtin_amdgcn_s_setprio(3);
#endif
    te_finalize_lane<KIND>(xyz, prefix, out, n, lanes, l);
}

#if defined(__HIPCC__)
// ---- latency variant for small batches (the upper levels of a byte-digest tree) ---------------------------------
// One workgroup of S waves per 64 messages: wave j sums the table entries of steps j, j + S, ..., the S partial sums
// are combined by a log2(S)-deep tree of full additions through LDS, and wave 0 converts to affine with its own
// inversion.  Depth ceil(steps/S) mixed additions + log2(S) full additions instead of `steps` mixed additions, and
// one launch instead of two.
#define AKP_TE_SPLIT 8
// KIND: table kind (as te_accumulate_kernel); XONLY: digest = x coordinate (Bowe-Hopwood, Pedersen with TECompressor)
template <int KIND, bool XONLY>
__global__ void __launch_bounds__(64 * AKP_TE_SPLIT) te_crh_small_kernel(const TeEntry* __restrict__ lut, const TeEntry* __restrict__ lut1,
                                                                        const uint8_t* __restrict__ msgs, size_t msg_len, size_t msg_stride, u32 D,
                                                                        u32 n_groups, u32 n_steps, const TeEntry* __restrict__ tail,
                                                                        Fr* __restrict__ out, size_t n) {
    __shared__ u32 part[AKP_TE_SPLIT - 1][36][64];
    const u32 j = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const u32 lane = threadIdx.x & 63u;
    const size_t item = (size_t)blockIdx.x * 64 + lane;
    const size_t idx = item < n ? item : n - 1;
    Ext acc = te_accumulate_strided<KIND>(lut, lut1, msgs + idx * msg_stride, msg_len, D, n_groups, n_steps, j, AKP_TE_SPLIT);
    if (tail && j == AKP_TE_SPLIT - 1) acc = te_madd(acc, load_niels(tail));  // the wave with the fewest steps takes the constant
#pragma unroll 1
    for (u32 stride = 1; stride < AKP_TE_SPLIT; stride <<= 1) {
        if ((j & (2 * stride - 1)) == stride) {  // each wave j > 0 publishes exactly once, in slot j - 1
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                part[j - 1][i][lane] = (u32)acc.X.l[i];
                part[j - 1][9 + i][lane] = (u32)acc.Y.l[i];
                part[j - 1][18 + i][lane] = (u32)acc.Z.l[i];
                part[j - 1][27 + i][lane] = (u32)acc.T.l[i];
            }
        }
        __syncthreads();
        if ((j & (2 * stride - 1)) == 0) {
            Ext o;
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                o.X.l[i] = (int32_t)part[j + stride - 1][i][lane];
                o.Y.l[i] = (int32_t)part[j + stride - 1][9 + i][lane];
                o.Z.l[i] = (int32_t)part[j + stride - 1][18 + i][lane];
                o.T.l[i] = (int32_t)part[j + stride - 1][27 + i][lane];
            }
            acc = te_add_ext(acc, o);
        }
    }
    if (j != 0 || item >= n) return;
    const FS zi = f29_inv(acc.Z);  // Z != 0 always (complete formulas)
    if (!XONLY) {
        store_fr_g(out + item * 2, f29_to_wire(f29_mul(acc.X, zi)));
        store_fr_g(out + item * 2 + 1, f29_to_wire(f29_mul(acc.Y, zi)));
    } else {
        store_fr_g(out + item, f29_to_wire(f29_mul(acc.X, zi)));
    }
}
#endif

// ---- helpers for TwoToOneCRH / ByteDigestConverter ---------------------------------------------------
// buffer[i] = zeros(buflen); buffer[i][0..] = left[i] || right[i], zip-truncated
// (crh/pedersen/mod.rs:172-179, crh/bowe_hopwood/mod.rs:219-224).
__global__ void te_concat_bytes_kernel(const uint8_t* __restrict__ left, const uint8_t* __restrict__ right, size_t half_len,
                                       size_t buflen, uint8_t* __restrict__ buf, size_t n) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * buflen) return;
    const size_t i = t / buflen, b = t % buflen;
    uint8_t v = 0;
    if (b < half_len) v = left[i * half_len + b];
    else if (b < 2 * half_len) v = right[i * half_len + (b - half_len)];
    buf[t] = v;
}
// Same, but the two halves are digests (fe_per_digest Fr each, wire format) serialised with ark-serialize's
// uncompressed canonical little-endian encoding (macros.rs:3-13; merkle_tree/mod.rs:67-78).
// pairs: left digest = d[2i], right = d[2i+1] when right == nullptr (Merkle level), else left[i], right[i].
AKP_HD void te_serialize_pair_fe(const Fr* __restrict__ left, const Fr* __restrict__ right, u32 fe_per_digest,
                                 size_t buflen, uint8_t* __restrict__ buf, size_t t) {
    const size_t per = 2 * (size_t)fe_per_digest;  // field elements per pair
    const size_t i = t / per, k = t % per;
    const Fr* src;
    if (right == nullptr) src = left + i * per + k;
    else src = (k < fe_per_digest) ? (left + i * fe_per_digest + k) : (right + i * fe_per_digest + (k - fe_per_digest));
    const Fr c = f29_to_canonical_int(f29_from_wire<true>(load_fr_g(src)));
    uint8_t* dst = buf + i * buflen;
#pragma unroll
    for (u32 w = 0; w < 8; ++w) {
#pragma unroll
        for (u32 b = 0; b < 4; ++b) {
            const size_t pos = k * 32 + w * 4 + b;
            if (pos < buflen) dst[pos] = (uint8_t)(c.l[w] >> (8 * b));
        }
    }
}
__global__ void te_serialize_pairs_kernel(const Fr* __restrict__ left, const Fr* __restrict__ right, u32 fe_per_digest,
                                          size_t buflen, uint8_t* __restrict__ buf, size_t n) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * 2 * (size_t)fe_per_digest) return;
    te_serialize_pair_fe(left, right, fe_per_digest, buflen, buf, t);
}
// The same for buffers that hold the two digests completely (2 * fe_per_digest * 32 <= buflen): only the data bytes are written, at a
// pitch of exactly 64 * fe_per_digest bytes and with two 16-byte stores per field element -- the hash kernels are told the pitch and
// never read the zero padding (its contribution is a constant).  (The byte-wise kernel above took 1.7 ms of a 2^23-leaf
// Bowe-Hopwood tree's 18.8 ms: profiles/r04_s11.)
__global__ void __launch_bounds__(256) te_serialize_pairs_vec_kernel(const Fr* __restrict__ left, const Fr* __restrict__ right, u32 fe_per_digest,
                                                                    uint8_t* __restrict__ buf, size_t n) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t per = 2 * (size_t)fe_per_digest;
    if (t >= n * per) return;
    const size_t i = t / per, k = t % per;
    const Fr* src;
    if (right == nullptr) src = left + t;
    else src = (k < fe_per_digest) ? (left + i * fe_per_digest + k) : (right + i * fe_per_digest + (k - fe_per_digest));
    const Fr c = f29_to_canonical_int(f29_from_wire<true>(load_fr_g(src)));
    uint4* dst = reinterpret_cast<uint4*>(buf + t * 32);
    dst[0] = make_uint4(c.l[0], c.l[1], c.l[2], c.l[3]);
    dst[1] = make_uint4(c.l[4], c.l[5], c.l[6], c.l[7]);
}
// messages of 1..3 bytes, zero-padded to four (the accumulate kernels read message bits with one 32-bit load)
__global__ void te_pad4_kernel(const uint8_t* __restrict__ msgs, size_t stride, u32 len, uint8_t* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 v = 0;
    for (u32 k = 0; k < len; ++k) v |= (u32)msgs[i * stride + k] << (8u * k);
    reinterpret_cast<u32*>(out)[i] = v;
}
__global__ void te_zero_tail_kernel(uint8_t* __restrict__ buf, size_t buflen, size_t used, size_t n) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (used >= buflen) return;
    const size_t tail = buflen - used;
    if (t >= n * tail) return;
    buf[(t / tail) * buflen + used + (t % tail)] = 0;
}

}  // namespace akp
