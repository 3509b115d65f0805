// te_kernels.hpp -- Pedersen / Bowe-Hopwood CRH over Jubjub (ark_ed_on_bls12_381) on gfx950.
//
// Replaces, batch-wide:
//   pedersen::CRH::evaluate        crh/pedersen/mod.rs:76-129   (hot loop :112-124)
//   bowe_hopwood::CRH::evaluate    crh/bowe_hopwood/mod.rs:114-186 (hot loop :161-181)
// The reference walks the message bit by bit doing conditional projective additions of
// `generators[i][j]`.  Here the hash is evaluated as a FIXED-BASE windowed multi-scalar sum:
//   Pedersen:  H(m) = sum over sub-windows u of  LUT[u][digit_u],  digit = up to 4 message bits,
//              LUT[u][v] = sum_j v_j * generators[i][4s + j]   (valid for arbitrary generators)
//   Bowe-Hopwood: H(m) = x( sum over 3-bit chunks c of (-1)^b2 * LUT[c][b0 + 2 b1] ),
//              LUT[c][k] = (k + 1) * generators[c / W][c % W]  (zero chunk contributes +g, :167)
// LUT entries are precomputed once per parameter set, in "Niels" form (y+x, y-x, 2d*x*y), so one
// step is a 7-multiplication mixed addition (madd-2008-hwcd-3, a = -1, complete on Jubjub because
// d is a non-square).  One message per lane; the LUT (<= 400 KB) is read through L1/L2 -- each step
// all 64 lanes gather from the same <= 1.5 KB line group, and a step is ~7 Montgomery products
// (~10^4 cycles per wave) so the path is integer-ALU bound, not memory bound.
// The projective -> affine conversion (crh/pedersen/mod.rs:128, bowe_hopwood/mod.rs:185) is one field
// inversion per message in the reference; here it is a separate pass that shares one inversion
// among `chain` messages per lane (Montgomery's trick), so its cost is ~5 products per message.
//
// The north-star text suggests LDS bucket accumulation; buckets belong to variable-base MSM
// (Pippenger).  With fixed bases the table method needs no buckets and no cross-lane reduction
// (see DESIGN.md "Pedersen").
#pragma once
#include "fr.hpp"

namespace akp {

// 2d (Montgomery), d = -(10240/10241): ark_ed_on_bls12_381::EdwardsConfig::COEFF_D
AKP_HD Fr te_2d() {
    return Fr{{0x72e9ed5fu, 0x54a448acu, 0x1b373967u, 0xa51befdbu, 0x7b4a799eu, 0xc0d81f21u, 0xd27ecf14u, 0x3c0445feu}};
}

struct Niels {
    Fr ypx, ymx, t2d;  // y + x, y - x, 2d*x*y
};
struct Ext {
    Fr X, Y, Z, T;  // x = X/Z, y = Y/Z, T = XY/Z
};

AKP_HD Ext ext_identity() { return Ext{fr_zero(), fr_one(), fr_one(), fr_zero()}; }
AKP_HD Niels niels_identity() { return Niels{fr_one(), fr_one(), fr_zero()}; }
AKP_HD Niels niels_from_affine(const Fr& x, const Fr& y) {
    return Niels{fr_add(y, x), fr_sub(y, x), fr_mul(fr_mul(x, y), te_2d())};
}
AKP_HD Niels niels_neg(const Niels& q) { return Niels{q.ymx, q.ypx, fr_neg(q.t2d)}; }

// P + Q, Q affine in Niels form (7 products)
AKP_HD Ext te_madd(const Ext& p, const Niels& q) {
    const Fr a = fr_mul(fr_sub(p.Y, p.X), q.ymx);
    const Fr b = fr_mul(fr_add(p.Y, p.X), q.ypx);
    const Fr c = fr_mul(p.T, q.t2d);
    const Fr d = fr_dbl(p.Z);
    const Fr e = fr_sub(b, a), f = fr_sub(d, c), g = fr_add(d, c), h = fr_add(b, a);
    return Ext{fr_mul(e, f), fr_mul(g, h), fr_mul(f, g), fr_mul(e, h)};
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
AKP_HD Niels load_niels(const Niels* p) {
    const Fr* f = reinterpret_cast<const Fr*>(p);
    return Niels{load_fr_g(f), load_fr_g(f + 1), load_fr_g(f + 2)};
}

// ---- table construction (one-off per parameter set) ------------------------------------------
// Pedersen: sub-window u = (window i, nibble s) covers generators[i][4s .. 4s+w), w = min(4, W-4s).
// entry v in [0,16): sum of the generators selected by the bits of v (v = 0 -> identity).
AKP_HD Niels te_pedersen_lut_entry(const Fr* __restrict__ gens_affine /*[N][W][2]*/, u32 W, u32 subs_per_window, u32 idx) {
    const u32 u = idx >> 4, v = idx & 15u;
    const u32 i = u / subs_per_window, s = u % subs_per_window;
    const u32 w = (W - 4u * s) < 4u ? (W - 4u * s) : 4u;
    Ext acc = ext_identity();
#pragma unroll 1
    for (u32 j = 0; j < w; ++j) {
        if ((v >> j) & 1u) {
            const Fr* g = gens_affine + ((size_t)i * W + 4u * s + j) * 2;
            acc = te_madd(acc, niels_from_affine(load_fr_g(g), load_fr_g(g + 1)));
        }
    }
    const Fr zi = fr_inv(acc.Z);
    return niels_from_affine(fr_mul(acc.X, zi), fr_mul(acc.Y, zi));
}
__global__ void te_build_pedersen_lut(const Fr* __restrict__ gens_affine, u32 W, u32 subs_per_window, u32 n_sub,
                                      Niels* __restrict__ lut /*[n_sub][16]*/) {
    const u32 idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_sub * 16u) return;
    lut[idx] = te_pedersen_lut_entry(gens_affine, W, subs_per_window, idx);
}
// Bowe-Hopwood: chunk c uses generators flat index c; entry k in [0,4): (k+1) * g.
AKP_HD Niels te_bh_lut_entry(const Fr* __restrict__ gens_affine /*[N*W][2]*/, u32 idx) {
    const u32 c = idx >> 2, k = idx & 3u;
    const Fr* g = gens_affine + (size_t)c * 2;
    const Niels gn = niels_from_affine(load_fr_g(g), load_fr_g(g + 1));
    Ext acc = ext_identity();
#pragma unroll 1
    for (u32 j = 0; j <= k; ++j) acc = te_madd(acc, gn);
    const Fr zi = fr_inv(acc.Z);
    return niels_from_affine(fr_mul(acc.X, zi), fr_mul(acc.Y, zi));
}
__global__ void te_build_bh_lut(const Fr* __restrict__ gens_affine, u32 n_gen, Niels* __restrict__ lut) {
    const u32 idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_gen * 4u) return;
    lut[idx] = te_bh_lut_entry(gens_affine, idx);
}

// ---- message bit access -----------------------------------------------------------------------
// bits [o, o+w) of a message of `len` bytes, LSB-first per byte (crh/pedersen/mod.rs:200-209);
// bits past the end read as zero (Pedersen zero padding :91-99 / Bowe-Hopwood chunk padding :131-138).
AKP_HD u32 msg_bits(const uint8_t* __restrict__ msg, size_t len, size_t o, u32 w) {
    const size_t byte = o >> 3;
    u32 v = 0;
    if (byte < len) v = msg[byte];
    if (byte + 1 < len) v |= (u32)msg[byte + 1] << 8;
    return (v >> (o & 7)) & ((1u << w) - 1u);
}

// ---- accumulate: one message per lane ------------------------------------------------------------
// kind 0 (Pedersen): n_steps sub-windows, digit = msg_bits(i*W + 4s, w), entry lut[u*16 + digit].
// kind 1 (Bowe-Hopwood): n_steps chunks, entry lut[c*4 + (b0 + 2 b1)], negated when b2.
// Writes the extended-coordinate sum (X, Y, Z) to xyz[idx*3 ..].
template <int KIND>
AKP_HD Ext te_accumulate_item(const Niels* __restrict__ lut, const uint8_t* __restrict__ msg, size_t msg_len, u32 W,
                              u32 subs_per_window, u32 n_steps) {
    Ext acc = ext_identity();
#pragma unroll 1
    for (u32 u = 0; u < n_steps; ++u) {
        Niels q;
        if (KIND == 0) {
            const u32 i = u / subs_per_window, s = u % subs_per_window;
            const u32 w = (W - 4u * s) < 4u ? (W - 4u * s) : 4u;
            const u32 digit = msg_bits(msg, msg_len, (size_t)i * W + 4u * s, w);
            q = load_niels(lut + (size_t)u * 16u + digit);
        } else {
            const u32 bits = msg_bits(msg, msg_len, (size_t)u * 3u, 3u);
            q = load_niels(lut + (size_t)u * 4u + (bits & 3u));
            if (bits & 4u) q = niels_neg(q);
        }
        acc = te_madd(acc, q);
    }
    return acc;
}
template <int KIND>
__global__ void __launch_bounds__(256) te_accumulate_kernel(const Niels* __restrict__ lut, const uint8_t* __restrict__ msgs,
                                                           size_t msg_len, u32 W, u32 subs_per_window, u32 n_steps,
                                                           Fr* __restrict__ xyz, size_t n) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const Ext acc = te_accumulate_item<KIND>(lut, msgs + idx * msg_len, msg_len, W, subs_per_window, n_steps);
    store_fr_g(xyz + idx * 3, acc.X);
    store_fr_g(xyz + idx * 3 + 1, acc.Y);
    store_fr_g(xyz + idx * 3 + 2, acc.Z);
}

// ---- projective -> affine with shared inversions ---------------------------------------------------
// Lane l handles elements l, l + L, l + 2L, ... (L = total lanes, coalesced), at most `chain` of them:
// forward prefix products of Z into `prefix`, one inversion, backward pass emitting x (and y).
// out: Pedersen n x (x, y); Bowe-Hopwood n x (x).
template <int KIND>
AKP_HD void te_finalize_lane(const Fr* __restrict__ xyz, Fr* __restrict__ prefix, Fr* __restrict__ out, size_t n,
                             size_t lanes, size_t l) {
    Fr run = fr_one();
    size_t cnt = 0;
#pragma unroll 1
    for (size_t e = l; e < n; e += lanes, ++cnt) {
        store_fr_g(prefix + e, run);  // product of the earlier Z's of this lane
        run = fr_mul(run, load_fr_g(xyz + e * 3 + 2));
    }
    Fr inv = fr_inv(run);  // Z != 0 always (complete formulas)
#pragma unroll 1
    for (size_t k = cnt; k-- > 0;) {
        const size_t e = l + k * lanes;
        const Fr zi = fr_mul(inv, load_fr_g(prefix + e));
        inv = fr_mul(inv, load_fr_g(xyz + e * 3 + 2));
        if (KIND == 0) {
            store_fr_g(out + e * 2, fr_mul(load_fr_g(xyz + e * 3), zi));
            store_fr_g(out + e * 2 + 1, fr_mul(load_fr_g(xyz + e * 3 + 1), zi));
        } else {
            store_fr_g(out + e, fr_mul(load_fr_g(xyz + e * 3), zi));
        }
    }
}
template <int KIND>
__global__ void __launch_bounds__(256) te_finalize_kernel(const Fr* __restrict__ xyz, Fr* __restrict__ prefix,
                                                         Fr* __restrict__ out, size_t n, size_t lanes) {
    const size_t l = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= lanes || l >= n) return;
    te_finalize_lane<KIND>(xyz, prefix, out, n, lanes, l);
}

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
// Same, but the two halves are digests (fe_per_digest Fr each, Montgomery) serialised with
// ark-serialize's uncompressed canonical little-endian encoding (macros.rs:3-13; merkle_tree/mod.rs:67-78).
// pairs: left digest = d[2i], right = d[2i+1] when right == nullptr (Merkle level), else left[i], right[i].
AKP_HD void te_serialize_pair_fe(const Fr* __restrict__ left, const Fr* __restrict__ right, u32 fe_per_digest,
                                 size_t buflen, uint8_t* __restrict__ buf, size_t t) {
    const size_t per = 2 * (size_t)fe_per_digest;  // field elements per pair
    const size_t i = t / per, k = t % per;
    const Fr* src;
    if (right == nullptr) src = left + i * per + k;
    else src = (k < fe_per_digest) ? (left + i * fe_per_digest + k) : (right + i * fe_per_digest + (k - fe_per_digest));
    const Fr c = fr_from_mont(load_fr_g(src));
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
__global__ void te_zero_tail_kernel(uint8_t* __restrict__ buf, size_t buflen, size_t used, size_t n) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (used >= buflen) return;
    const size_t tail = buflen - used;
    if (t >= n * tail) return;
    buf[(t / tail) * buflen + used + (t % tail)] = 0;
}

}  // namespace akp
