// poseidon_kernels.hpp -- batched Poseidon permutation / fixed-length sponge CRH on gfx950.
//
// Replaces, level-wide, what the reference does one hash at a time:
//   PoseidonSponge::permute            sponge/poseidon/mod.rs:98-121
//   poseidon::CRH::evaluate            crh/poseidon/mod.rs:30-40
//   poseidon::TwoToOneCRH::compress    crh/poseidon/mod.rs:66-79
//   MerkleTree level loops             merkle_tree/mod.rs:458-515 (one launch per level)
//
// Mapping: one sponge instance per lane (64 per wavefront).  The path is integer-ALU bound
// (~626 Montgomery products per 192 algorithmic bytes), so the design goal is issue
// efficiency of v_mad_u64_u32, a small instruction footprint (the whole round loop stays
// inside the instruction cache) and >= 2 waves per SIMD -- not HBM bandwidth.
//
// Generic kernel ("LDS register file"): the t-word sponge state of each lane lives in LDS in
// a lane-interleaved layout (slot s, half h, lane l -> uint4 index (2s+h)*BLOCK + l, i.e.
// conflict-free ds_read/write_b128), so loops over state elements are real loops (dynamic
// slot index) and there is exactly one inlined multiplier body per use site.  Round keys and
// the MDS matrix are wave-uniform: they are fetched through the scalar cache (s_load_dwordx8)
// straight into SGPR operands of the multiplier.
#pragma once
#include "fr.hpp"

namespace akp {

struct PoseidonDims {
    u32 t, rate, capacity, full_rounds, partial_rounds;
    u64 alpha;
};

template <int BLOCK>
struct LdsFile {
    uint4* base;  // [slot][2][BLOCK]
    AKP_D Fr load(u32 slot) const {
        const uint4 lo = base[(2 * slot) * BLOCK + threadIdx.x];
        const uint4 hi = base[(2 * slot + 1) * BLOCK + threadIdx.x];
        return Fr{{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w}};
    }
    AKP_D void store(u32 slot, const Fr& v) const {
        base[(2 * slot) * BLOCK + threadIdx.x] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
        base[(2 * slot + 1) * BLOCK + threadIdx.x] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
    }
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

// S-box x^alpha for a wave-uniform runtime alpha (sponge/poseidon/mod.rs:66-77)
AKP_HD Fr sbox_runtime(const Fr& x, u64 alpha) {
    if (alpha == 0) return fr_one();
    int top = 63 - __builtin_clzll(alpha);
    Fr r = x;
#pragma unroll 1
    for (int i = top - 1; i >= 0; --i) {
        r = fr_sqr(r);
        if ((alpha >> i) & 1) r = fr_mul(r, x);
    }
    return r;
}

// One full permutation of the state held in buffer `cur` of the state file `f`
// (slots [cur*t, cur*t + t)).  On return the state is in buffer `cur` (updated).
// `File` is LdsFile<BLOCK> on the device; tests/host_harness instantiates it with a plain array
// so the very same round code runs on the CPU against the oracle.
template <class File>
AKP_HD void poseidon_permute_file(const PoseidonDims& D, const Fr* __restrict__ ark, const Fr* __restrict__ mds,
                                  const File& f, u32& cur) {
    const u32 T = D.t;
    const u32 half = D.full_rounds / 2;
    const u32 R = D.full_rounds + D.partial_rounds;
#pragma unroll 1
    for (u32 r = 0; r < R; ++r) {
        const bool full = (r < half) || (r >= half + D.partial_rounds);
        const u32 nsbox = full ? T : 1u;
        const Fr* arkr = ark + (size_t)r * T;
        const u32 src = cur * T, dst = (cur ^ 1u) * T;
        // ARK (:79-83) fused with the S-box (:66-77)
#pragma unroll 1
        for (u32 e = 0; e < T; ++e) {
            Fr x = fr_add(f.load(src + e), arkr[e]);
            if (e < nsbox) x = sbox_runtime(x, D.alpha);
            f.store(src + e, x);
        }
        // MDS (:85-96): new[i] = sum_j state[j] * mds[i][j]
#pragma unroll 1
        for (u32 i = 0; i < T; ++i) {
            const Fr* row = mds + (size_t)i * T;
            Fr acc = fr_mul(f.load(src), row[0]);
#pragma unroll 1
            for (u32 j = 1; j < T; ++j) acc = fr_add(acc, fr_mul(f.load(src + j), row[j]));
            f.store(dst + i, acc);
        }
        cur ^= 1u;
    }
}

// Fixed-length sponge CRH of item `idx`: squeeze1(absorb(in[0..k))) on a fresh sponge.
// Element e of item idx is in0[idx*k + e] (in1 == nullptr), or in0[idx] / in1[idx] for e = 0 / 1.
template <class File>
AKP_HD Fr poseidon_crh_item(const PoseidonDims& D, const Fr* __restrict__ ark, const Fr* __restrict__ mds, const File& f,
                            const Fr* __restrict__ in0, const Fr* __restrict__ in1, size_t k, size_t idx) {
    u32 cur = 0;
#pragma unroll 1
    for (u32 e = 0; e < D.t; ++e) f.store(e, fr_zero());  // PoseidonSponge::new :223-234
    // absorb_internal from index 0 (:124-153) followed by the squeeze permutation (:324-344):
    // ceil(k/rate) permutations, or one permutation of the zero state when k == 0 (:238-240).
    size_t done = 0;
    do {
        const size_t take = (k - done) < D.rate ? (k - done) : D.rate;
#pragma unroll 1
        for (size_t j = 0; j < take; ++j) {
            const size_t e = done + j;
            const Fr* src = (in1 == nullptr) ? (in0 + idx * k + e) : (e == 0 ? in0 + idx : in1 + idx);
            const u32 slot = cur * D.t + D.capacity + (u32)j;
            f.store(slot, fr_add(f.load(slot), load_fr_global(src)));
        }
        done += take;
        poseidon_permute_file(D, ark, mds, f, cur);
    } while (done < k);
    return f.load(cur * D.t + D.capacity);  // squeeze_internal(0, 1) :156-186
}

// ---- kernels --------------------------------------------------------------------------
// states: [n][t] Fr, permuted in place.
template <int BLOCK>
__global__ void __launch_bounds__(BLOCK) poseidon_permute_kernel(PoseidonDims D, const Fr* __restrict__ ark,
                                                               const Fr* __restrict__ mds, Fr* states, size_t n) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    LdsFile<BLOCK> f{reinterpret_cast<uint4*>(smem)};
    const size_t idx = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (idx >= n) return;  // lanes never exchange data: no barriers anywhere
    Fr* st = states + idx * D.t;
    u32 cur = 0;
#pragma unroll 1
    for (u32 e = 0; e < D.t; ++e) f.store(e, load_fr_global(st + e));
    poseidon_permute_file(D, ark, mds, f, cur);
#pragma unroll 1
    for (u32 e = 0; e < D.t; ++e) store_fr_global(st + e, f.load(cur * D.t + e));
}

// Fixed-length sponge CRH: out[i] = squeeze1(absorb(in_i[0..k))).
// Element j of input i is  (j < split ? in0 : in1)[i * stride_j ...]:
//   CRH batch / Merkle levels: in0 = inputs, k elements contiguous per item  (in1 = nullptr)
//   two-to-one batch:          in0 = left, in1 = right, k = 2
template <int BLOCK>
__global__ void __launch_bounds__(BLOCK) poseidon_crh_kernel(PoseidonDims D, const Fr* __restrict__ ark,
                                                           const Fr* __restrict__ mds, const Fr* __restrict__ in0,
                                                           const Fr* __restrict__ in1, size_t k, Fr* __restrict__ out,
                                                           size_t n) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    LdsFile<BLOCK> f{reinterpret_cast<uint4*>(smem)};
    const size_t idx = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (idx >= n) return;
    store_fr_global(out + idx, poseidon_crh_item(D, ark, mds, f, in0, in1, k, idx));
}

}  // namespace akp
