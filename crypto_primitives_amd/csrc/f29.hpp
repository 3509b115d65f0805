// f29.hpp -- the arithmetic core of the gfx950 kernels: BLS12-381 Fr in radix 2^29, 9 limbs, lazy reduction.
//
// Why this representation (measured on MI355X, profiles/r01_s1_microbench_instruction_rates.txt):
// v_mad_u64_u32 / v_mad_i64_i32 issue at the SAME rate as v_addc_co_u32, v_lshl_add_u64 or v_fma_f64
// (~4.4 cycles per wave64); only plain e32 adds/ands are cheaper (~2.6).  The cost of a field product is
// therefore its instruction COUNT, and a saturated radix-2^32 multiplier spends more than half of its
// ~324 instructions on carries (one v_addc per partial product) and on the conditional subtraction.
// With 29-bit limbs a partial product is < 2^58 (< 2^60 for lazily added operands), so a whole column of a
// 9x9 product -- and of a 3-term dot product -- accumulates in ONE 64-bit register with no carry handling:
// one v_mad per partial product, two instructions per column for the carry (and + 64-bit shift).
//   product:      81 + 81 mads + ~45 others  ~ 205 instructions   (was 324)
//   square:       45 + 81 mads + ~55 others  ~ 180
//   3-term dot:  243 + 81 mads + ~45 others  ~ 370                (was 3 x 324 + adds)
//   add / sub:     9 plain adds, no reduction
// No conditional subtraction is ever needed between operations: the Montgomery radix is R' = 2^261, so
// p / R' = 2^-6 and every product lands in (-1.1p, 2.1p) for operands below 2^258 in magnitude.
//
// Internal form of a field element x:  any integer v == x * 2^261 (mod p), |v| < 2^258,
//   v = sum l[i] * 2^(29 i).   "Normalised": l[0..7] in [0, 2^29), l[8] small (signed in the signed flavour).
// Flavours:  FU (unsigned limbs, Poseidon: only + and *),  FS (signed limbs, Jubjub: also -).
// Headroom rules (checked in DESIGN.md "F29 bounds"):
//   FU: 9*A*B + 9*2^58 < 2^64 -> product operands may have limbs < 2^30 (one lazy add each); a 3-term dot
//       may take state limbs < 2^30 against normalised constants (27 * 2^59 + 9 * 2^58 < 2^64).
//   FS: every column value must stay inside (-2^63, 2^63).  The reduction terms only subtract (>= -8 * 2^58), so the
//       bound is on the sum of the operand products:
//       product   9*A*B < 2^63 -> A*B <= 2^59.8: one operand may be a lazy sum/difference (< 2^30), the other
//                 normalised; anything larger goes through f29_weak_norm first;
//       square    of an operand with non-negative limbs 0..7 <= 2^30 - 1 and a top limb < 2^26 (|value| < 2^258: a normalised value plus a
//                 normalised round key): the doubled limbs stay below 2^31, the widest column without the top limb is
//                 column 7: 8 * (2^30 - 1)^2 + carry (< 7 * 2^31) = 2^63 - 2^31 + small < 2^63; columns 8, 9 have two
//                 terms with the small top limb;
//       dot2..5   Poseidon constants carry balanced digits (f29_balance, |digit| <= 2^28): a product is < 2^57 for a
//                 normalised state limb, the bound is two-sided (products of either sign, the reduction subtracts up to
//                 16 * 2^57): 3 terms over lanes with one lazy addition (45 + 16) * 2^57, 4 / 5 terms over normalised
//                 lanes (36 + 16) / (45 + 16) * 2^57 < 2^63.
// Wire format <-> internal: one Montgomery product with a constant each way (x*2^256 <-> x*2^261).
#pragma once
#include <stdint.h>

#include "fr.hpp"

namespace akp {

#define AKP_MASK29 0x1fffffffu

template <bool SIGNED>
struct F29T;
template <>
struct F29T<false> {
    typedef uint32_t L;
    typedef uint64_t W;
    L l[9];
};
template <>
struct F29T<true> {
    typedef int32_t L;
    typedef int64_t W;
    L l[9];
};
typedef F29T<false> FU;
typedef F29T<true> FS;
// Flavour of the Poseidon kernels.  Round 2: signed.  Poseidon only adds and multiplies non-negative limbs, but the
// signed routines reduce subtractively, which costs two instructions per reduction column instead of four (-18
// instructions per routine, -8 % of a permutation).  Price: half the accumulator headroom, see "Headroom rules".
#define AKP_PS true
typedef F29T<AKP_PS> FP;

AKP_HD u32 p29(int i) {
    constexpr u32 P[9] = {0x00000001u, 0x1ffffff8u, 0x1f96ffbfu, 0x1b4805ffu, 0x1d80553bu, 0x0c0404d0u, 0x1520cce7u, 0x0a6533afu, 0x0073eda7u};
#if defined(__HIP_DEVICE_COMPILE__)
    // p[1] = 2^29 - 8: left visible, hipcc turns m * p[1] into two 64-bit shift/subtract instructions; one
    // v_mad_u64_u32 with the constant in an SGPR is cheaper (same issue cost per instruction on gfx950).
    if (i == 1) {
        u32 c = P[1];
        asm("" : "+s"(c));
        return c;
    }
#endif
    return P[i];
}
#define AKP_F29_CONST(name, ...)                       \
    template <bool S>                                  \
    AKP_HD F29T<S> name() {                            \
        constexpr u32 C[9] = {__VA_ARGS__};            \
        F29T<S> r;                                     \
        _Pragma("unroll") for (int i = 0; i < 9; ++i) r.l[i] = (typename F29T<S>::L)C[i]; \
        return r;                                      \
    }
// R' mod p  (the field's one), 2^266 mod p (wire -> internal), 2^256 mod p (internal -> wire), d*R' (Jubjub d)
AKP_F29_CONST(f29_one, 0x1fffffbau, 0x0000022fu, 0x1cb61180u, 0x0a4e5c00u, 0x0ee8b1a2u, 0x16e6aedfu, 0x1907f8bbu, 0x0853ddf7u, 0x004d043fu)
AKP_F29_CONST(f29_k_in, 0x1ffff72bu, 0x000046a7u, 0x1f5f3540u, 0x0ce3021cu, 0x118f3661u, 0x008176cbu, 0x054e487cu, 0x102e8190u, 0x001e092eu)
AKP_F29_CONST(f29_k_out, 0x1ffffffeu, 0x0000000fu, 0x00d20080u, 0x096ff400u, 0x04ff5588u, 0x07f7f65eu, 0x15be6631u, 0x0b3598a0u, 0x001824b1u)
AKP_F29_CONST(f29_te_d, 0x0e9ed5e8u, 0x12245679u, 0x002d9f52u, 0x03bb3367u, 0x0d9bfb3du, 0x18ebb3ccu, 0x1c29ceccu, 0x0a7b6020u, 0x0020d725u)
// 2^522 mod p: plain integer -> internal representation in one product
AKP_F29_CONST(f29_r2, 0x0a71b3c0u, 0x1d32207eu, 0x1663d999u, 0x1c5abc93u, 0x03b58c44u, 0x0be37438u, 0x0829f771u, 0x1660139eu, 0x0027fd91u)
AKP_F29_CONST(f29_16p, 0x00000010u, 0x1fffff80u, 0x196ffbffu, 0x14805fffu, 0x180553bdu, 0x00404d0eu, 0x120cce76u, 0x06533afau, 0x073eda75u)
AKP_F29_CONST(f29_8p, 0x00000008u, 0x1fffffc0u, 0x1cb7fdffu, 0x1a402fffu, 0x0c02a9deu, 0x00202687u, 0x0906673bu, 0x13299d7du, 0x039f6d3au)
AKP_F29_CONST(f29_4p, 0x00000004u, 0x1fffffe0u, 0x1e5bfeffu, 0x0d2017ffu, 0x160154efu, 0x10101343u, 0x1483339du, 0x0994cebeu, 0x01cfb69du)
AKP_F29_CONST(f29_2p, 0x00000002u, 0x1ffffff0u, 0x1f2dff7fu, 0x16900bffu, 0x1b00aa77u, 0x180809a1u, 0x0a4199ceu, 0x14ca675fu, 0x00e7db4eu)
AKP_F29_CONST(f29_p, 0x00000001u, 0x1ffffff8u, 0x1f96ffbfu, 0x1b4805ffu, 0x1d80553bu, 0x0c0404d0u, 0x1520cce7u, 0x0a6533afu, 0x0073eda7u)

template <bool S>
AKP_HD F29T<S> f29_zero() {
    F29T<S> r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = 0;
    return r;
}
template <bool S>
AKP_HD F29T<S> f29_add(const F29T<S>& a, const F29T<S>& b) {
    F29T<S> r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = a.l[i] + b.l[i];
    return r;
}
AKP_HD FS f29_sub(const FS& a, const FS& b) {
    FS r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = a.l[i] - b.l[i];
    return r;
}
AKP_HD FS f29_neg(const FS& a) {
    FS r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = -a.l[i];
    return r;
}
template <bool S>
AKP_HD F29T<S> f29_dbl(const F29T<S>& a) {
    return f29_add(a, a);
}
// one parallel carry step: limbs back to [0, 2^29 + small) (top limb absorbs the rest)
template <bool S>
AKP_HD F29T<S> f29_weak_norm(const F29T<S>& a) {
    typedef typename F29T<S>::L L;
    F29T<S> r;
    r.l[0] = (L)((u32)a.l[0] & AKP_MASK29);
#pragma unroll
    for (int i = 1; i < 8; ++i) r.l[i] = (L)((u32)a.l[i] & AKP_MASK29) + (a.l[i - 1] >> 29);
    r.l[8] = a.l[8] + (a.l[7] >> 29);
    return r;
}

// ---- Montgomery reduction tail shared by mul / sqr / dot: columns 0..8 have been folded into m[] ----
// Unsigned flavour (additive): m = -acc mod 2^29, acc += m * p (p[0] = 1), result (ab + mp) / 2^261 in [0, 2.1p).
// Signed flavour (subtractive): m = acc mod 2^29, acc -= m * p; the low 29 bits cancel, so the step is just the
// arithmetic shift (floor), and the result (ab - mp) / 2^261 lies in (-2.1p, 1.1p) -- same magnitude bound, two
// instructions per column instead of four.
#define AKP_F29_MSTEP()                               \
    if constexpr (S) {                                \
        m[k] = (u32)acc & AKP_MASK29;                 \
        acc >>= 29;                                   \
    } else {                                          \
        m[k] = (0u - (u32)acc) & AKP_MASK29;          \
        acc += (W)(u64)m[k];                          \
        acc >>= 29;                                   \
    }
#define AKP_F29_RED(mi, pj)                                  \
    if constexpr (S) acc -= (W)((u64)(mi) * (u64)(pj));      \
    else acc += (W)((u64)(mi) * (u64)(pj));

#if defined(__HIP_DEVICE_COMPILE__) && defined(AKP_F29_ASM)
// single-accumulator-chain inline assembly for the field routines (generated, asm/gen_inline.py)
#include "asm/f29_asm.inc"
#endif

// a * b / 2^261 (mod p).  Output normalised.
template <bool S>
AKP_HD F29T<S> f29_mul(const F29T<S>& a, const F29T<S>& b) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(AKP_F29_ASM)
    return f29_mul_asm(a, b);
#endif
    typedef typename F29T<S>::L L;
    typedef typename F29T<S>::W W;
    W acc = 0;
    u32 m[9];
    F29T<S> t;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
#pragma unroll
        for (int i = 0; i <= k; ++i) acc += (W)a.l[i] * (W)b.l[k - i];
#pragma unroll
        for (int i = 0; i < k; ++i) { AKP_F29_RED(m[i], p29(k - i)) }
        AKP_F29_MSTEP()
    }
#pragma unroll
    for (int k = 9; k < 17; ++k) {
#pragma unroll
        for (int i = k - 8; i < 9; ++i) acc += (W)a.l[i] * (W)b.l[k - i];
#pragma unroll
        for (int i = k - 8; i < 9; ++i) { AKP_F29_RED(m[i], p29(k - i)) }
        t.l[k - 9] = (L)((u32)acc & AKP_MASK29);
        acc >>= 29;
    }
    t.l[8] = (L)acc;
    return t;
}

// a * c / 2^261 for a wave-uniform c (a constant): same value as f29_mul; the assembly version keeps c in SGPRs
template <bool S>
AKP_HD F29T<S> f29_mulc(const F29T<S>& a, const F29T<S>& c) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(AKP_F29_ASM)
    return f29_mulc_asm(a, c);
#else
    return f29_mul(a, c);
#endif
}

// a^2 / 2^261: off-diagonal products use the doubled operand (45 products instead of 81)
template <bool S>
AKP_HD F29T<S> f29_sqr(const F29T<S>& a) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(AKP_F29_ASM)
    return f29_sqr_asm(a);
#endif
    typedef typename F29T<S>::L L;
    typedef typename F29T<S>::W W;
    L a2[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) a2[i] = a.l[i] + a.l[i];
    W acc = 0;
    u32 m[9];
    F29T<S> t;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
#pragma unroll
        for (int i = 0; 2 * i < k; ++i) acc += (W)a2[i] * (W)a.l[k - i];
        if ((k & 1) == 0) acc += (W)a.l[k / 2] * (W)a.l[k / 2];
#pragma unroll
        for (int i = 0; i < k; ++i) { AKP_F29_RED(m[i], p29(k - i)) }
        AKP_F29_MSTEP()
    }
#pragma unroll
    for (int k = 9; k < 17; ++k) {
#pragma unroll
        for (int i = k - 8; 2 * i < k; ++i) acc += (W)a2[i] * (W)a.l[k - i];
        if ((k & 1) == 0) acc += (W)a.l[k / 2] * (W)a.l[k / 2];
#pragma unroll
        for (int i = k - 8; i < 9; ++i) { AKP_F29_RED(m[i], p29(k - i)) }
        t.l[k - 9] = (L)((u32)acc & AKP_MASK29);
        acc >>= 29;
    }
    t.l[8] = (L)acc;
    return t;
}

// (a0*b0 + a1*b1 + a2*b2) / 2^261: one reduction for three products (the MDS row of a t = 3 state).
// FU: a limbs < 2^30, b limbs < 2^29  =>  27 * 2^59 + 9 * 2^58 + carry < 2^64.
// FS: a limbs <= 2^29 + small, b limbs < 2^29  =>  27 * 2^58 < 2^63.
// On the device b0..b2 must be wave-uniform (constants): the assembly version takes them in SGPRs.
template <bool S>
AKP_HD F29T<S> f29_dot3(const F29T<S>& a0, const F29T<S>& b0, const F29T<S>& a1, const F29T<S>& b1, const F29T<S>& a2, const F29T<S>& b2) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(AKP_F29_ASM)
    return f29_dot3_asm(a0, b0, a1, b1, a2, b2);
#endif
    typedef typename F29T<S>::L L;
    typedef typename F29T<S>::W W;
    W acc = 0;
    u32 m[9];
    F29T<S> t;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
#pragma unroll
        for (int i = 0; i <= k; ++i) {
            acc += (W)a0.l[i] * (W)b0.l[k - i];
            acc += (W)a1.l[i] * (W)b1.l[k - i];
            acc += (W)a2.l[i] * (W)b2.l[k - i];
        }
#pragma unroll
        for (int i = 0; i < k; ++i) { AKP_F29_RED(m[i], p29(k - i)) }
        AKP_F29_MSTEP()
    }
#pragma unroll
    for (int k = 9; k < 17; ++k) {
#pragma unroll
        for (int i = k - 8; i < 9; ++i) {
            acc += (W)a0.l[i] * (W)b0.l[k - i];
            acc += (W)a1.l[i] * (W)b1.l[k - i];
            acc += (W)a2.l[i] * (W)b2.l[k - i];
        }
#pragma unroll
        for (int i = k - 8; i < 9; ++i) { AKP_F29_RED(m[i], p29(k - i)) }
        t.l[k - 9] = (L)((u32)acc & AKP_MASK29);
        acc >>= 29;
    }
    t.l[8] = (L)acc;
    return t;
}

// (a0*b0 + a1*b1) / 2^261: two products, one reduction (same operand bounds as f29_dot3)
template <bool S>
AKP_HD F29T<S> f29_dot2(const F29T<S>& a0, const F29T<S>& b0, const F29T<S>& a1, const F29T<S>& b1) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(AKP_F29_ASM)
    return f29_dot2_asm(a0, b0, a1, b1);
#endif
    typedef typename F29T<S>::L L;
    typedef typename F29T<S>::W W;
    W acc = 0;
    u32 m[9];
    F29T<S> t;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
#pragma unroll
        for (int i = 0; i <= k; ++i) {
            acc += (W)a0.l[i] * (W)b0.l[k - i];
            acc += (W)a1.l[i] * (W)b1.l[k - i];
        }
#pragma unroll
        for (int i = 0; i < k; ++i) { AKP_F29_RED(m[i], p29(k - i)) }
        AKP_F29_MSTEP()
    }
#pragma unroll
    for (int k = 9; k < 17; ++k) {
#pragma unroll
        for (int i = k - 8; i < 9; ++i) {
            acc += (W)a0.l[i] * (W)b0.l[k - i];
            acc += (W)a1.l[i] * (W)b1.l[k - i];
        }
#pragma unroll
        for (int i = k - 8; i < 9; ++i) { AKP_F29_RED(m[i], p29(k - i)) }
        t.l[k - 9] = (L)((u32)acc & AKP_MASK29);
        acc >>= 29;
    }
    t.l[8] = (L)acc;
    return t;
}

// Balanced digits for CONSTANTS (round keys, matrix entries): limbs 0..7 in [-2^28, 2^28), same value.  A product of a state
// limb (<= 2^29 + small) with such a digit is below 2^57 in magnitude, so FOUR or FIVE terms (36 / 45 products per column)
// fit one signed 64-bit column where normalised digits (< 2^29) allow three -- one Montgomery reduction per row of a t = 4, 5
// state instead of two.  Applied once when a parameter set is converted (poseidon_convert_params_kernel).
AKP_HD FS f29_balance(const FS& a) {
    FS r;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        int32_t x = a.l[i] + c;                      // |x| < 2^30
        int32_t lo = (int32_t)((u32)x & AKP_MASK29);  // [0, 2^29)
        c = x >> 29;
        if (lo >= (1 << 28)) {
            lo -= (1 << 29);
            c += 1;
        }
        r.l[i] = lo;
    }
    r.l[8] = a.l[8] + c;
    return r;
}
// sum_{k < N} a_k * b_k / 2^261 with one reduction, N = 4, 5.  b_k: balanced constants (wave-uniform on the device).
template <int N>
AKP_HD FS f29_dotn_cpp(const FS* const* a, const FS* const* b) {
    typedef int64_t W;
    const bool S = true;
    (void)S;
    W acc = 0;
    u32 m[9];
    FS t;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
#pragma unroll
        for (int i = 0; i <= k; ++i)
#pragma unroll
            for (int q = 0; q < N; ++q) acc += (W)a[q]->l[i] * (W)b[q]->l[k - i];
#pragma unroll
        for (int i = 0; i < k; ++i) acc -= (W)((u64)m[i] * (u64)p29(k - i));
        m[k] = (u32)acc & AKP_MASK29;
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; ++k) {
#pragma unroll
        for (int i = k - 8; i < 9; ++i)
#pragma unroll
            for (int q = 0; q < N; ++q) acc += (W)a[q]->l[i] * (W)b[q]->l[k - i];
#pragma unroll
        for (int i = k - 8; i < 9; ++i) acc -= (W)((u64)m[i] * (u64)p29(k - i));
        t.l[k - 9] = (int32_t)((u32)acc & AKP_MASK29);
        acc >>= 29;
    }
    t.l[8] = (int32_t)acc;
    return t;
}
AKP_HD FS f29_dot4(const FS& a0, const FS& b0, const FS& a1, const FS& b1, const FS& a2, const FS& b2, const FS& a3, const FS& b3) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(AKP_F29_ASM)
    return f29_dot4_asm(a0, b0, a1, b1, a2, b2, a3, b3);
#else
    const FS* a[4] = {&a0, &a1, &a2, &a3};
    const FS* b[4] = {&b0, &b1, &b2, &b3};
    return f29_dotn_cpp<4>(a, b);
#endif
}
AKP_HD FS f29_dot5(const FS& a0, const FS& b0, const FS& a1, const FS& b1, const FS& a2, const FS& b2, const FS& a3, const FS& b3, const FS& a4,
                   const FS& b4) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(AKP_F29_ASM)
    return f29_dot5_asm(a0, b0, a1, b1, a2, b2, a3, b3, a4, b4);
#else
    const FS* a[5] = {&a0, &a1, &a2, &a3, &a4};
    const FS* b[5] = {&b0, &b1, &b2, &b3, &b4};
    return f29_dotn_cpp<5>(a, b);
#endif
}

// x^e, small public exponent (S-box).  MSB-first from x; equals ark-ff Field::pow for e >= 1.
template <bool S>
AKP_HD F29T<S> f29_pow_small(const F29T<S>& x, u64 e) {
    if (e == 0) return f29_one<S>();
    int top = 63 - __builtin_clzll(e);
    F29T<S> r = x;
#if defined(__HIP_DEVICE_COMPILE__) && defined(AKP_F29_ASM)
    // in-place assembly steps: the chain r = r * r, r = r * x needs no register copies
#pragma unroll 1
    for (int i = top - 1; i >= 0; --i) {
        f29_sqr_ip_asm(r);
        if ((e >> i) & 1) f29_mul_ip_asm(r, x);
    }
    return r;
#endif
#pragma unroll 1
    for (int i = top - 1; i >= 0; --i) {
        r = f29_sqr(r);
        if ((e >> i) & 1) r = f29_mul(r, x);
    }
    return r;
}
// a^(p-2); a == 0 -> 0.  255 squarings + 127 products; kept as the cross-check of f29_inv
template <bool S>
AKP_HD F29T<S> f29_inv_fermat(const F29T<S>& a) {
    const u32 E[8] = {0xffffffffu, 0xfffffffeu, AKP_P2, AKP_P3, AKP_P4, AKP_P5, AKP_P6, AKP_P7};  // p - 2
    F29T<S> r = a;  // bit 254 of p-2 is set
#pragma unroll 1
    for (int i = 253; i >= 0; --i) {
        r = f29_sqr(r);
        if ((E[i >> 5] >> (i & 31)) & 1) r = f29_mul(r, a);
    }
    return r;
}

// ---- wire format (8 x u32, x * 2^256 mod p, canonical) <-> internal ------------------------------------
template <bool S>
AKP_HD F29T<S> f29_unpack(const Fr& w) {  // plain re-limbing of a 256-bit integer
    typedef typename F29T<S>::L L;
    F29T<S> r;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int bit = 29 * i, wi = bit >> 5, sh = bit & 31;
        u32 v = w.l[wi] >> sh;
        if (sh + 29 > 32 && wi + 1 < 8) v |= w.l[wi + 1] << (32 - sh);
        r.l[i] = (L)(v & AKP_MASK29);
    }
    return r;
}
template <bool S>
AKP_HD F29T<S> f29_from_wire(const Fr& w) {
    return f29_mulc(f29_unpack<S>(w), f29_k_in<S>());
}
// canonical representative in [0, p) of a value with |v| < 4p (FS) / 0 <= v < 8p (FU), as 8 x u32.
// WIDE: |v| < 16p (FS) / 0 <= v < 32p (FU) -- sums of several reduced terms (the rows of a wide linear layer).
template <bool S, bool WIDE = false>
AKP_HD Fr f29_canonical_pack(const F29T<S>& a) {
    // 1. make non-negative, 2. exact carry propagation, 3. subtract 16p, 8p (wide), 4p, 2p, p where possible, 4. re-limb
    int32_t v[9];
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        int32_t x = (int32_t)a.l[i] + c;
        if (S) x += (int32_t)(WIDE ? f29_16p<false>().l[i] : f29_4p<false>().l[i]);
        if (i < 8) {
            v[i] = (int32_t)((u32)x & AKP_MASK29);
            c = x >> 29;
        } else {
            v[i] = x;
        }
    }
#pragma unroll
    for (int step = WIDE ? 0 : 2; step < 5; ++step) {
        const FU sub = step == 0 ? f29_16p<false>() : (step == 1 ? f29_8p<false>() : (step == 2 ? f29_4p<false>() : (step == 3 ? f29_2p<false>() : f29_p<false>())));
        int32_t d[9];
        int32_t bw = 0;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            int32_t x = v[i] - (int32_t)sub.l[i] + bw;
            if (i < 8) {
                d[i] = (int32_t)((u32)x & AKP_MASK29);
                bw = x >> 29;
            } else {
                d[i] = x;
            }
        }
        const bool ge = d[8] >= 0;
#pragma unroll
        for (int i = 0; i < 9; ++i) v[i] = ge ? d[i] : v[i];
    }
    Fr o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int bit = 32 * j, li = bit / 29, sh = bit % 29;
        u32 x = (u32)v[li] >> sh;
        if (li + 1 < 9) x |= (u32)v[li + 1] << (29 - sh);
        if (58 - sh < 32 && li + 2 < 9) x |= (u32)v[li + 2] << (58 - sh);
        o.l[j] = x;
    }
    return o;
}
template <bool S>
AKP_HD Fr f29_to_wire(const F29T<S>& a) {
    return f29_canonical_pack(f29_mulc(a, f29_k_out<S>()));
}
// canonical little-endian integer of the field element (ark-serialize's encoding of Fq)
template <bool S>
AKP_HD Fr f29_to_canonical_int(const F29T<S>& a) {
    F29T<S> one = f29_zero<S>();
    one.l[0] = 1;
    return f29_canonical_pack(f29_mulc(a, one));  // a / R' = x
}
AKP_HD FS f29_to_signed(const FU& a) {
    FS r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = (int32_t)a.l[i];
    return r;
}

// ---- modular inverse by batched division steps (Bernstein-Yang "safegcd", the half-delta variant) --------------
// The 29-bit signed-limb layout is the natural one for it: 29 division steps on the low words of (f, g) give a 2x2
// transition matrix with entries |.| <= 2^29, one limb; applying it to (f, g) and to the Bezout pair (d, e) is a
// 9-limb by 1-limb multiply-accumulate whose division by 2^29 is a limb shift, and p = 1 (mod 2^29) makes the
// "add a multiple of p so the low limb vanishes" step a negation.  21 batches = 609 >= 590 steps, enough for any
// 256-bit input.  ~16 k cheap VALU instructions with no data-dependent branch, against ~84 k for a^(p-2).
struct F29Trans {
    int32_t u, v, q, r;
};
AKP_HD int32_t f29_divsteps29(int32_t zeta, u32 f, u32 g, F29Trans& t) {
    u32 u = 1, v = 0, q = 0, r = 1;
#pragma unroll
    for (int i = 0; i < 29; ++i) {
        u32 c1 = (u32)(zeta >> 31);  // zeta < 0
        const u32 c2 = 0u - (g & 1u);  // g odd
        const u32 x = (f ^ c1) - c1, y = (u ^ c1) - c1, z = (v ^ c1) - c1;  // (f, u, v) or their negatives
        g += x & c2;
        q += y & c2;
        r += z & c2;
        c1 &= c2;                               // swap case: zeta < 0 and g odd
        zeta = (int32_t)((u32)zeta ^ c1) - 1;   // -zeta - 2 or zeta - 1
        f += g & c1;
        u += q & c1;
        v += r & c1;
        g >>= 1;
        u <<= 1;
        v <<= 1;
    }
    t.u = (int32_t)u;
    t.v = (int32_t)v;
    t.q = (int32_t)q;
    t.r = (int32_t)r;
    return zeta;
}
// (f, g) <- t * (f, g) / 2^29 (exact).  Limbs 0..7 in [0, 2^29), limb 8 carries the sign.
AKP_HD void f29_update_fg(int32_t* f, int32_t* g, const F29Trans& t) {
    int64_t cf = (int64_t)t.u * f[0] + (int64_t)t.v * g[0];
    int64_t cg = (int64_t)t.q * f[0] + (int64_t)t.r * g[0];
    cf >>= 29;
    cg >>= 29;
#pragma unroll
    for (int i = 1; i < 9; ++i) {
        cf += (int64_t)t.u * f[i] + (int64_t)t.v * g[i];
        cg += (int64_t)t.q * f[i] + (int64_t)t.r * g[i];
        f[i - 1] = (int32_t)((u32)cf & AKP_MASK29);
        g[i - 1] = (int32_t)((u32)cg & AKP_MASK29);
        cf >>= 29;
        cg >>= 29;
    }
    f[8] = (int32_t)cf;
    g[8] = (int32_t)cg;
}
// (d, e) <- t * (d, e) / 2^29 (mod p), both kept in (-2p, p)
AKP_HD void f29_update_de(int32_t* d, int32_t* e, const F29Trans& t) {
    const int32_t sd = d[8] >> 31, se = e[8] >> 31;
    int32_t md = (t.u & sd) + (t.v & se);
    int32_t me = (t.q & sd) + (t.r & se);
    int64_t cd = (int64_t)t.u * d[0] + (int64_t)t.v * e[0];
    int64_t ce = (int64_t)t.q * d[0] + (int64_t)t.r * e[0];
    md -= (int32_t)(((u32)cd + (u32)md) & AKP_MASK29);  // p^-1 = 1 (mod 2^29): cd + p * md = 0 (mod 2^29)
    me -= (int32_t)(((u32)ce + (u32)me) & AKP_MASK29);
    cd += (int64_t)md;  // p limb 0 is 1
    ce += (int64_t)me;
    cd >>= 29;
    ce >>= 29;
#pragma unroll
    for (int i = 1; i < 9; ++i) {
        const int64_t pi = (int64_t)p29(i);
        cd += (int64_t)t.u * d[i] + (int64_t)t.v * e[i] + pi * md;
        ce += (int64_t)t.q * d[i] + (int64_t)t.r * e[i] + pi * me;
        d[i - 1] = (int32_t)((u32)cd & AKP_MASK29);
        e[i - 1] = (int32_t)((u32)ce & AKP_MASK29);
        cd >>= 29;
        ce >>= 29;
    }
    d[8] = (int32_t)cd;
    e[8] = (int32_t)ce;
}
// 1/a in the internal (x * 2^261) representation; a == 0 -> 0
template <bool S>
AKP_HD F29T<S> f29_inv(const F29T<S>& a) {
    typedef typename F29T<S>::L L;
    const F29T<false> x = f29_unpack<false>(f29_to_canonical_int(a));  // the plain integer in [0, p)
    int32_t f[9], g[9], d[9], e[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        f[i] = (int32_t)p29(i);
        g[i] = (int32_t)x.l[i];
        d[i] = 0;
        e[i] = 0;
    }
    e[0] = 1;
    int32_t zeta = -1;
#pragma unroll 1
    for (int it = 0; it < 21; ++it) {
        F29Trans t;
        zeta = f29_divsteps29(zeta, (u32)f[0] | ((u32)f[1] << 29), (u32)g[0] | ((u32)g[1] << 29), t);
        f29_update_de(d, e, t);
        f29_update_fg(f, g, t);
    }
    // f = +-1 (or p when a == 0): the inverse is sign(f) * d, brought from (-2p, p) to [0, p)
    const int32_t neg = f[8] >> 31;
    int32_t c = 0;
    const int32_t add0 = d[8] >> 31;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        int32_t v = d[i] + ((int32_t)p29(i) & add0);
        v = (v ^ neg) - neg;
        v += c;
        if (i < 8) {
            d[i] = (int32_t)((u32)v & AKP_MASK29);
            c = v >> 29;
        } else {
            d[i] = v;
        }
    }
    c = 0;
    const int32_t add1 = d[8] >> 31;
    F29T<S> r;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        int32_t v = d[i] + ((int32_t)p29(i) & add1) + c;
        if (i < 8) {
            r.l[i] = (L)((u32)v & AKP_MASK29);
            c = v >> 29;
        } else {
            r.l[i] = (L)v;
        }
    }
    return f29_mulc(r, f29_r2<S>());  // plain 1/x -> (1/x) * 2^261
}

// 12-dword padded storage (three 16-byte vectors) for tables / scratch in global memory
struct F29Pad {
    u32 w[12];
};
template <bool S>
AKP_HD F29T<S> f29_load_pad(const F29Pad* p) {
    typedef typename F29T<S>::L L;
    const uint4* q = reinterpret_cast<const uint4*>(p);
    const uint4 a = q[0], b = q[1], c = q[2];
    F29T<S> r;
    r.l[0] = (L)a.x; r.l[1] = (L)a.y; r.l[2] = (L)a.z; r.l[3] = (L)a.w;
    r.l[4] = (L)b.x; r.l[5] = (L)b.y; r.l[6] = (L)b.z; r.l[7] = (L)b.w;
    r.l[8] = (L)c.x;
    return r;
}
template <bool S>
AKP_HD void f29_store_pad(F29Pad* p, const F29T<S>& v) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4((u32)v.l[0], (u32)v.l[1], (u32)v.l[2], (u32)v.l[3]);
    q[1] = make_uint4((u32)v.l[4], (u32)v.l[5], (u32)v.l[6], (u32)v.l[7]);
    q[2] = make_uint4((u32)v.l[8], 0u, 0u, 0u);
}

}  // namespace akp
