// fr.hpp -- BLS12-381 scalar field Fr (= Jubjub base field) for gfx950 and for the host side
// of the C ABI.  Product code: independent of oracle/.
//
// Representation: 8 x u32 little-endian limbs, Montgomery form x*2^256 mod p, fully reduced.
// Byte-identical to ark-ff's Fp256 in-memory layout (4 x u64 LE limbs, Montgomery, R = 2^256)
// which is what every reference call site on the path manipulates
// (sponge/poseidon/mod.rs:70,81,90-91 -- Fp::pow / add_assign / mul).
//
// Why 32-bit limbs: CDNA4 has no 64x64 multiplier; the widest integer multiply is
// v_mad_u64_u32 (32x32+64 -> 64, carry-out in VCC).  p = 1 (mod 2^32), so the Montgomery
// quotient digit is m = -t0 (no multiply) and p[0]*m is a plain add.  Multiplication is
// product-scanning (Comba) with the reduction interleaved (FIPS): per column one 64-bit
// accumulator + one overflow word, i.e. one v_mad_u64_u32 + one v_addc per partial product.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define AKP_HD __host__ __device__ __forceinline__
#define AKP_D __device__ __forceinline__
#else
#define AKP_HD inline
#define AKP_D inline
#endif

namespace akp {

typedef uint32_t u32;
typedef uint64_t u64;

struct Fr {
    u32 l[8];
};

// p = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
#define AKP_P0 0x00000001u
#define AKP_P1 0xffffffffu
#define AKP_P2 0xfffe5bfeu
#define AKP_P3 0x53bda402u
#define AKP_P4 0x09a1d805u
#define AKP_P5 0x3339d808u
#define AKP_P6 0x299d7d48u
#define AKP_P7 0x73eda753u

AKP_HD u32 fr_p_limb(int i) {
    constexpr u32 P[8] = {AKP_P0, AKP_P1, AKP_P2, AKP_P3, AKP_P4, AKP_P5, AKP_P6, AKP_P7};
    return P[i];
}
// R mod p (Montgomery one), R^2 mod p
AKP_HD Fr fr_one() {
    return Fr{{0xfffffffeu, 0x00000001u, 0x00034802u, 0x5884b7fau, 0xecbc4ff5u, 0x998c4fefu, 0xacc5056fu, 0x1824b159u}};
}
AKP_HD Fr fr_r2() {
    return Fr{{0xf3f29c6du, 0xc999e990u, 0x87925c23u, 0x2b6cedcbu, 0x7254398fu, 0x05d31496u, 0x9f59ff11u, 0x0748d9d9u}};
}
AKP_HD Fr fr_zero() { return Fr{{0, 0, 0, 0, 0, 0, 0, 0}}; }

AKP_HD bool fr_is_zero(const Fr& a) {
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) o |= a.l[i];
    return o == 0;
}
AKP_HD bool fr_eq(const Fr& a, const Fr& b) {
    u32 o = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) o |= a.l[i] ^ b.l[i];
    return o == 0;
}

// r = (t >= p || hi) ? t - p : t   (t < 2p + hi*2^256 assumed so one subtraction suffices)
AKP_HD Fr fr_cond_sub_p(const u32 (&t)[8], u32 hi) {
    u32 r[8];
    u64 bw = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        u64 d = (u64)t[j] - fr_p_limb(j) - bw;
        r[j] = (u32)d;
        bw = (d >> 32) & 1;
    }
    const bool ge = (hi != 0) || (bw == 0);
    Fr o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o.l[j] = ge ? r[j] : t[j];
    return o;
}

AKP_HD Fr fr_add(const Fr& a, const Fr& b) {
    u32 t[8];
    u64 c = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        c += (u64)a.l[j] + b.l[j];
        t[j] = (u32)c;
        c >>= 32;
    }
    return fr_cond_sub_p(t, 0);  // a + b < 2p < 2^256
}
AKP_HD Fr fr_dbl(const Fr& a) { return fr_add(a, a); }

AKP_HD Fr fr_sub(const Fr& a, const Fr& b) {
    u32 t[8];
    u64 bw = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        u64 d = (u64)a.l[j] - b.l[j] - bw;
        t[j] = (u32)d;
        bw = (d >> 32) & 1;
    }
    const u32 mask = (u32)0 - (u32)bw;  // add p back on borrow
    u64 c = 0;
    Fr o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        c += (u64)t[j] + (fr_p_limb(j) & mask);
        o.l[j] = (u32)c;
        c >>= 32;
    }
    return o;
}
AKP_HD Fr fr_neg(const Fr& a) { return fr_sub(fr_zero(), a); }

// ---------------------------------------------------------------------------------------
// Montgomery multiplication, portable form (host + device).  acc = {c2 : acc64}.
#define AKP_MAC(x, y)                         \
    {                                         \
        const u64 p_ = (u64)(x) * (u64)(y);   \
        const u64 s_ = acc + p_;              \
        c2 += (u32)(s_ < acc);                \
        acc = s_;                             \
    }

AKP_HD Fr fr_mul_portable(const Fr& a, const Fr& b) {
    u32 m[8], t[8];
    u64 acc = 0;
    u32 c2 = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
#pragma unroll
        for (int i = 0; i <= k; ++i) AKP_MAC(a.l[i], b.l[k - i]);
#pragma unroll
        for (int i = 0; i < k; ++i) AKP_MAC(m[i], fr_p_limb(k - i));
        m[k] = 0u - (u32)acc;        // -p^{-1} = -1 (mod 2^32)
        AKP_MAC(m[k], AKP_P0);       // low word becomes 0
        acc = (acc >> 32) | ((u64)c2 << 32);
        c2 = 0;
    }
#pragma unroll
    for (int k = 8; k < 16; ++k) {
#pragma unroll
        for (int i = k - 7; i < 8; ++i) AKP_MAC(a.l[i], b.l[k - i]);
#pragma unroll
        for (int i = k - 7; i < 8; ++i) AKP_MAC(m[i], fr_p_limb(k - i));
        t[k - 8] = (u32)acc;
        acc = (acc >> 32) | ((u64)c2 << 32);
        c2 = 0;
    }
    return fr_cond_sub_p(t, (u32)acc);
}

#if defined(__HIP_DEVICE_COMPILE__)
// ---------------------------------------------------------------------------------------
// gfx950 form: the same FIPS schedule, but each partial product is exactly
//   v_mad_u64_u32 acc, vcc, x, y, acc ; v_addc_co_u32 c2, vcc, 0, c2, vcc
// (hipcc does not use the carry-out of v_mad_u64_u32 on its own: it emits a 64-bit compare,
// a cndmask and hazard nops per product -- see DESIGN.md "Fr multiply").
AKP_D void mac_vv(u64& acc, u32& c2, u32 x, u32 y) {
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc"
        : "+v"(acc), "+v"(c2)
        : "v"(x), "v"(y)
        : "vcc");
}
// first product of a column: {c2:acc} < 2^37 so the 64-bit add cannot carry
AKP_D void mac_vv_nc(u64& acc, u32 x, u32 y) {
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y) : "vcc");
}
AKP_D void mac_vs(u64& acc, u32& c2, u32 x, u32 s) {
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc"
        : "+v"(acc), "+v"(c2)
        : "v"(x), "s"(s)
        : "vcc");
}
// m = -lo(acc); acc = (acc >> 32) + (lo != 0) with c2 shifted in: the p[0] = 1 column step
AKP_D void mont_step(u64& acc, u32& c2, u32& m) {
    u32 lo = (u32)acc, hi = (u32)(acc >> 32), nlo, nhi;
    asm("v_sub_co_u32_e32 %0, vcc, 0, %3\n\t"
        "v_addc_co_u32_e32 %1, vcc, 0, %4, vcc\n\t"
        "v_addc_co_u32_e32 %2, vcc, 0, %5, vcc"
        : "=&v"(m), "=&v"(nlo), "=&v"(nhi)
        : "v"(lo), "v"(hi), "v"(c2)
        : "vcc");
    acc = ((u64)nhi << 32) | nlo;
    c2 = 0;
}

AKP_D Fr fr_mul(const Fr& a, const Fr& b) {
    // p[1] = 0xffffffff is the inline constant -1; p[2..7] sit in SGPRs (VOP3 on gfx9 takes no literal)
    const u32 p2 = AKP_P2, p3 = AKP_P3, p4 = AKP_P4, p5 = AKP_P5, p6 = AKP_P6, p7 = AKP_P7;
    const u32 PS[8] = {AKP_P0, AKP_P1, p2, p3, p4, p5, p6, p7};
    u32 m[8], t[8];
    u64 acc = 0;
    u32 c2 = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        mac_vv_nc(acc, a.l[0], b.l[k]);
#pragma unroll
        for (int i = 1; i <= k; ++i) mac_vv(acc, c2, a.l[i], b.l[k - i]);
#pragma unroll
        for (int i = 0; i < k; ++i) mac_vs(acc, c2, m[i], PS[k - i]);
        mont_step(acc, c2, m[k]);
    }
#pragma unroll
    for (int k = 8; k < 16; ++k) {
        bool first = true;
#pragma unroll
        for (int i = k - 7; i < 8; ++i) {
            if (first) { mac_vv_nc(acc, a.l[i], b.l[k - i]); first = false; }
            else mac_vv(acc, c2, a.l[i], b.l[k - i]);
        }
#pragma unroll
        for (int i = k - 7; i < 8; ++i) mac_vs(acc, c2, m[i], PS[k - i]);
        t[k - 8] = (u32)acc;
        acc = (acc >> 32) | ((u64)c2 << 32);
        c2 = 0;
    }
    return fr_cond_sub_p(t, (u32)acc);
}
// host overload seen by the device pass (test harness / host-side parameter code)
__host__ inline Fr fr_mul(const Fr& a, const Fr& b) { return fr_mul_portable(a, b); }
#else
AKP_HD Fr fr_mul(const Fr& a, const Fr& b) { return fr_mul_portable(a, b); }
#endif

AKP_HD Fr fr_sqr(const Fr& a) { return fr_mul(a, a); }

// x^e for a small public exponent (the Poseidon S-box, sponge/poseidon/mod.rs:66-77).
// MSB-first square-and-multiply starting from x (the leading one costs nothing); the
// value equals ark-ff's Field::pow for every e >= 1.
AKP_HD Fr fr_pow_small(const Fr& x, u64 e) {
    if (e == 0) return fr_one();
    int top = 63;
    while (!((e >> top) & 1)) --top;
    Fr r = x;
    for (int i = top - 1; i >= 0; --i) {
        r = fr_sqr(r);
        if ((e >> i) & 1) r = fr_mul(r, x);
    }
    return r;
}

// a^(p-2).  255 squarings + 4-bit fixed window.  a = 0 -> 0.
AKP_HD Fr fr_inv(const Fr& a) {
    // exponent p - 2, 32-bit words little-endian
    const u32 E[8] = {0xffffffffu, 0xfffffffeu, AKP_P2, AKP_P3, AKP_P4, AKP_P5, AKP_P6, AKP_P7};  // p - 2 (borrow into word 1)
    Fr r = fr_one();
    for (int i = 254; i >= 0; --i) {
        r = fr_sqr(r);
        if ((E[i >> 5] >> (i & 31)) & 1) r = fr_mul(r, a);
    }
    return r;
}

AKP_HD Fr fr_to_mont(const Fr& canon) { return fr_mul(canon, fr_r2()); }
AKP_HD Fr fr_from_mont(const Fr& a) {
    Fr one = fr_zero();
    one.l[0] = 1;
    return fr_mul(a, one);
}

}  // namespace akp
