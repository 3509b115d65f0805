// poseidon_kernels.hpp -- batched Poseidon permutation / fixed-length sponge CRH on gfx950.
//
// Replaces, level-wide, what the reference does one hash at a time:
//   PoseidonSponge::permute            sponge/poseidon/mod.rs:98-121
//   poseidon::CRH::evaluate            crh/poseidon/mod.rs:30-40
//   poseidon::TwoToOneCRH::compress    crh/poseidon/mod.rs:66-79
//   MerkleTree level loops             merkle_tree/mod.rs:458-515 (one launch per level)
//
// Mapping: one sponge instance per lane (64 per wavefront).  The path is integer-ALU bound (hundreds of
// 255-bit modular products per 192 algorithmic bytes), so the design goal is a minimal VALU instruction
// count per product (f29.hpp), a round body that stays inside the instruction cache and >= 2 waves per
// SIMD to cover the dependent v_mad chains -- not HBM bandwidth.
//
// Arithmetic runs in the lazy radix-2^29 form of f29.hpp (unsigned flavour: Poseidon only adds and
// multiplies).  Round keys and the MDS matrix are converted to that form once per parameter set and are
// wave-uniform: they come through the scalar cache (s_load) straight into SGPR operands of v_mad_u64_u32.
//   t == 3, large batches (the rate-2 headline instance): one item per lane, state in 27 VGPRs, S-box x^alpha by
//          square-and-multiply, each linear-layer row one dot product with a single Montgomery reduction.
//   any other t <= 16, large batches: one item per lane, the state lives in an LDS "register file"
//          (lane-interleaved dwords, conflict-free) so loops over state elements are real loops; rows are chunks
//          of 3-term dots; partial rounds run in the sparse form (in place) as well.
//   small batches, any t (tree tops, single sponges): one WAVE per state lane, t waves per 64 items (bottom of
//          this file): 40 % lower latency, lower throughput.
// The host derives algebraically equivalent constant sets that remove products (poseidon_opt.hpp: sparse partial
// rounds, then the lane-0 / lane-1 / full re-parameterisations); `PoseidonConsts::scaled` says which one a kernel got.
// Values cross HBM in the ABI's wire format (ark-ff Montgomery, R = 2^256); one product with a constant converts
// on load / store, except in the full form, which computes on the wire values directly.
#pragma once
#include "f29.hpp"
#include "akp_types.hpp"

namespace akp {

AKP_HD FP ldc(const F29Pad* p) { return f29_load_pad<AKP_PS>(p); }  // wave-uniform address -> scalar loads

// wire-format parameter array -> internal form (run once per parameter set)
__global__ void poseidon_convert_params_kernel(const Fr* __restrict__ in, F29Pad* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    // balanced digits: constants are only ever multiplied or added, and four / five-term row sums rely on |digit| <= 2^28
    if (i < n) f29_store_pad(out + i, f29_balance(f29_from_wire<AKP_PS>(load_fr_global(in + i))));
}

// =============================== t == 3: register-resident state ===================================
// One full permutation (sponge/poseidon/mod.rs:98-121): per round ARK (:79-83), S-box x^alpha on every
// lane (full rounds) or lane 0 only (partial rounds) (:66-77), then state = MDS * state (:85-96).
// When `sparse` is given the partial rounds run in the algebraically identical sparse form derived on the
// host (poseidon_opt.hpp): one key add, one S-box, row 0 as a 3-term dot, lanes 1,2 += w_i * s; the full
// round before the block applies `mpre`, and `ark` already carries the folded key residue.
// Limb bounds (signed flavour, f29.hpp "Headroom rules").  Constants (round keys, matrix entries) carry BALANCED digits,
// |digit| <= 2^28 (f29_balance).  dot / product outputs are normalised (<= 2^29 - 1), f29_weak_norm of a short sum is
// <= 2^29 + 2; + round key -> |limb| <= 1.5 * 2^29 + 2: the first squaring of an S-box sees columns of at most
// 9 * 2.25 * 2^58 = 2^62.3; every later routine of the S-box sees normalised operands.  Linear layers: a state limb
// (<= 2^30 + 1 after ONE lazy addition) times a balanced digit is < 2^58.01 in magnitude, so the three-term row of the
// t = 3 kernel over lanes that were renormalised in this or the previous round stays within +-(45 + 16) * 2^57 (products
// may have either sign now; the reduction subtracts up to 16 * 2^57): lanes 1,2 are renormalised every SECOND round.
struct PoseidonConsts {
    const F29Pad* ark;     // [R][t] round keys (with the residue folded in when sparse != nullptr)
    const F29Pad* mds;     // [t][t]
    const F29Pad* mpre;    // [t][t] or nullptr
    const F29Pad* sparse;  // [RP][2t] = q0, a00, u_1..u_{t-1}, w_1..w_{t-1}  or nullptr (dense partial rounds)
    const F29Pad* sbox0;   // [t] (0 + ark[0][i])^alpha: round-0 S-box output of a lane that enters as zero (the capacity
                           // lane and the unused rate lanes of a fresh sponge); nullptr when round 0 is not a full round
    u32 scaled;            // 1: the sparse constants are in the rescaled form (poseidon_rescale_sparse): in every partial
                           // round but the last the S-box output enters lane 0 with coefficient 1;
                           // 2: lane-1 form (poseidon_rescale_sparse_lane1): lane 1 takes the S-box output with
                           // coefficient 1 (one-lane-per-item kernels);
                           // 3: full form (poseidon_full_form, t = 3 register kernels only): lane-1 form + one matrix
                           // per full round in `mds` ([RF][t][t]) with unit diagonal except row 0 of the round before
                           // the partial block and the last round; a00 = 1 in the last partial round; `ark` scaled.  Lanes
                           // enter and leave scaled by 2^-5, i.e. as the wire value x * 2^256: no conversion products
};
typedef PoseidonConsts PoseidonT3Consts;
// zero_lanes: bit i set = lane i is known to be zero on entry (uniform over the batch): its first S-box is a constant.
// need_lanes: bit i set = lane i of the result is used; the other rows of the last linear layer are skipped.
// FULLFORM: the constants are in the full form (C.scaled == 3) -- a compile-time switch so that each instantiation
// carries only the row variants it uses (the kernel has to stay inside the 64 KB instruction cache)
template <bool FULLFORM>
AKP_HD void poseidon_permute_t3(const PoseidonDims& D, const PoseidonT3Consts& C, FP& s0, FP& s1, FP& s2, u32 zero_lanes = 0,
                                u32 need_lanes = 7u) {
    const u32 form = FULLFORM ? 3u : (C.scaled == 3u ? 0u : C.scaled);
    const u32 half = D.full_rounds / 2;
    const u32 R = D.full_rounds + D.partial_rounds;
    const bool opt = C.sparse != nullptr;
#pragma unroll 1
    for (u32 r = 0; r < R; ++r) {
        const bool full = (r < half) || (r >= half + D.partial_rounds);
        if (full || !opt) {
            const F29Pad* a = C.ark + (size_t)r * 3;
            const u32 z = (r == 0 && full && C.sbox0 != nullptr) ? zero_lanes : 0u;
            if (z & 1u) s0 = ldc(C.sbox0);
            else s0 = f29_pow_small(f29_add(s0, ldc(a)), D.alpha);
            s1 = f29_add(s1, ldc(a + 1));
            s2 = f29_add(s2, ldc(a + 2));
            if (full) {
                if (z & 2u) s1 = ldc(C.sbox0 + 1);
                else s1 = f29_pow_small(s1, D.alpha);
                if (z & 4u) s2 = ldc(C.sbox0 + 2);
                else s2 = f29_pow_small(s2, D.alpha);
            }
            const F29Pad* m = (opt && r + 1 == half) ? C.mpre : C.mds;
            bool unit0 = false, unit12 = false;  // full form: the diagonal coefficient of the row is 1
            if (FULLFORM) {
                m = C.mds + 9 * (size_t)(r < half ? r : r - D.partial_rounds);
                unit12 = r + 1 != R;
                unit0 = unit12 && r + 1 != half;
            }
            const u32 need = (r + 1 == R) ? need_lanes : 7u;
            FP n0 = s0, n1 = s1, n2 = s2;
            if (unit0) {
                if (need & 1u) n0 = f29_weak_norm(f29_add(s0, f29_dot2(s1, ldc(m + 1), s2, ldc(m + 2))));
            } else if (need & 1u) n0 = f29_dot3(s0, ldc(m + 0), s1, ldc(m + 1), s2, ldc(m + 2));
            if (unit12) {
                if (need & 2u) n1 = f29_weak_norm(f29_add(s1, f29_dot2(s0, ldc(m + 3), s2, ldc(m + 5))));
                if (need & 4u) n2 = f29_weak_norm(f29_add(s2, f29_dot2(s0, ldc(m + 6), s1, ldc(m + 7))));
            } else {
                if (need & 2u) n1 = f29_dot3(s0, ldc(m + 3), s1, ldc(m + 4), s2, ldc(m + 5));
                if (need & 4u) n2 = f29_dot3(s0, ldc(m + 6), s1, ldc(m + 7), s2, ldc(m + 8));
            }
            s0 = n0;
            s1 = n1;
            s2 = n2;
        } else {
            const u32 j = r - half;
            const F29Pad* sp = C.sparse + (size_t)j * 6;
            const FP s = f29_pow_small(f29_add(s0, ldc(sp)), D.alpha);
            // the S-box output enters lane 0 with coefficient 1: lane-0 form in all partial rounds but the last, full form
            // in the last one
            if ((form == 1u && j + 1 < D.partial_rounds) || (FULLFORM && j + 1 == D.partial_rounds))
                s0 = f29_weak_norm(f29_add(s, f29_dot2(s1, ldc(sp + 2), s2, ldc(sp + 3))));
            else s0 = f29_dot3(s, ldc(sp + 1), s1, ldc(sp + 2), s2, ldc(sp + 3));
            if (form >= 2u) s1 = f29_add(s1, s);
            else s1 = f29_add(s1, f29_mulc(s, ldc(sp + 4)));
            s2 = f29_add(s2, f29_mulc(s, ldc(sp + 5)));
            if ((j & 31u) == 31u) {  // lanes 1,2 gain < 2.1p per round and are never reduced mod p: fold them back
                s1 = f29_mulc(s1, f29_one<AKP_PS>());  // every 32 rounds so the top limb stays far below 2^32 for any RP
                s2 = f29_mulc(s2, f29_one<AKP_PS>());
            } else if ((j & 1u) || j + 1 == D.partial_rounds) {
                s1 = f29_weak_norm(s1);
                s2 = f29_weak_norm(s2);
            }
        }
    }
}
// absorb: lane[slot] += input, renormalised so that the following ARK add stays below 2^30.  Value selects per limb
// (a pointer select on `slot` would force the three lanes into scratch memory).
AKP_HD void t3_add_slot(FP& s0, FP& s1, FP& s2, u32 slot, const FP& v) {
    const FP n0 = f29_weak_norm(f29_add(s0, v)), n1 = f29_weak_norm(f29_add(s1, v)), n2 = f29_weak_norm(f29_add(s2, v));
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        s0.l[i] = slot == 0 ? n0.l[i] : s0.l[i];
        s1.l[i] = slot == 1 ? n1.l[i] : s1.l[i];
        s2.l[i] = slot == 2 ? n2.l[i] : s2.l[i];
    }
}
// Fixed-length sponge CRH of item idx on a fresh sponge: squeeze1(absorb(in[0..k))).
// Element e of item idx is in0[idx*k + e] (in1 == nullptr), or in0[idx] / in1[idx] for e = 0 / 1.
// absorb_internal from index 0 (:124-153) + the squeeze permutation (:324-344): ceil(k/rate)
// permutations, or one permutation of the zero state when k == 0 (:238-240).
template <bool FULLFORM>
AKP_HD Fr poseidon_crh_item_t3(const PoseidonDims& D, const PoseidonT3Consts& C, const Fr* __restrict__ in0,
                               const Fr* __restrict__ in1, size_t k, size_t idx) {
    FP s0 = f29_zero<AKP_PS>(), s1 = s0, s2 = s0;  // PoseidonSponge::new :223-234
    size_t done = 0;
    do {
        const size_t take = (k - done) < D.rate ? (k - done) : D.rate;
#pragma unroll 1
        for (size_t j = 0; j < take; ++j) {
            const size_t e = done + j;
            const Fr* src = (in1 == nullptr) ? (in0 + idx * k + e) : (e == 0 ? in0 + idx : in1 + idx);
            t3_add_slot(s0, s1, s2, D.capacity + (u32)j, FULLFORM ? f29_unpack<AKP_PS>(load_fr_global(src)) : f29_from_wire<AKP_PS>(load_fr_global(src)));
        }
        // fresh sponge: every lane outside [capacity, capacity + take) is still zero in the first permutation
        const u32 zero_lanes = done == 0 ? (7u & ~(((1u << take) - 1u) << D.capacity)) : 0u;
        done += take;
        // only state[capacity] of the last permutation is squeezed (:156-186)
        poseidon_permute_t3<FULLFORM>(D, C, s0, s1, s2, zero_lanes, done < k ? 7u : (1u << D.capacity));
        // (full form: the lanes leave the permutation as they entered it, as wire values, so the next block is absorbed
        // the same way)
    } while (done < k);
    // squeeze_internal(0, 1) :156-186 -- limb-wise selects (an array select would go through scratch)
    FP out;
#pragma unroll
    for (int i = 0; i < 9; ++i) out.l[i] = D.capacity == 0 ? s0.l[i] : (D.capacity == 1 ? s1.l[i] : s2.l[i]);
    if (FULLFORM) return f29_canonical_pack(out);  // the lane already holds x * 2^256
    return f29_to_wire(out);
}

// STAGED (round 4, the zero-copy host path): the workgroup's 256 states (24 KB, contiguous) go through an LDS tile with
// coalesced 16-byte loads and stores -- 1 KB per wave instruction -- instead of six 16-byte accesses per lane at a 96-byte pitch.
// In HBM the two are equivalent (counter traffic 1.004 x algorithmic either way); over PCIe the per-lane form splits every
// 128-byte line over several instructions.  Tile: 25 dwords per lane (24 + 1 pad: conflict-free per-lane reads).  Needs a
// 16-byte aligned `states`.
template <bool FULLFORM, bool STAGED>
AKP_D void poseidon_permute_t3_body(const PoseidonDims& D, const PoseidonT3Consts& C, Fr* states, size_t n) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    Fr* st = states + idx * 3;
    Fr w0, w1, w2;
#if defined(__HIPCC__)
    extern __shared__ u32 t3_tile[];
    const size_t first = (size_t)blockIdx.x * blockDim.x;
    const u32 cnt = (u32)(n - first < blockDim.x ? n - first : blockDim.x);
    u32* mine = t3_tile + threadIdx.x * 25u;
    if (STAGED) {
        const uint4* g = reinterpret_cast<const uint4*>(states + first * 3);
        for (u32 k = threadIdx.x; k < cnt * 6u; k += blockDim.x) {
            const uint4 v = g[k];
            u32* w = t3_tile + (k / 6u) * 25u + (k % 6u) * 4u;
            w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
        }
        __syncthreads();
        if (idx < n) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { w0.l[i] = mine[i]; w1.l[i] = mine[8 + i]; w2.l[i] = mine[16 + i]; }
        }
    } else
#endif
    {
        if (idx >= n) return;
        w0 = load_fr_global(st);
        w1 = load_fr_global(st + 1);
        w2 = load_fr_global(st + 2);
    }
    if (idx < n) {
        FP s0, s1, s2;
        if (FULLFORM) {  // the full form takes the wire value x * 2^256 as it is
            s0 = f29_unpack<AKP_PS>(w0);
            s1 = f29_unpack<AKP_PS>(w1);
            s2 = f29_unpack<AKP_PS>(w2);
        } else {
            s0 = f29_from_wire<AKP_PS>(w0);
            s1 = f29_from_wire<AKP_PS>(w1);
            s2 = f29_from_wire<AKP_PS>(w2);
        }
        poseidon_permute_t3<FULLFORM>(D, C, s0, s1, s2);
        if (FULLFORM) {  // the lanes already hold x * 2^256
            w0 = f29_canonical_pack(s0);
            w1 = f29_canonical_pack(s1);
            w2 = f29_canonical_pack(s2);
        } else {
            w0 = f29_to_wire(s0);
            w1 = f29_to_wire(s1);
            w2 = f29_to_wire(s2);
        }
    }
#if defined(__HIPCC__)
    if (STAGED) {
        if (idx < n) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { mine[i] = w0.l[i]; mine[8 + i] = w1.l[i]; mine[16 + i] = w2.l[i]; }
        }
        __syncthreads();
        uint4* g = reinterpret_cast<uint4*>(states + first * 3);
        for (u32 k = threadIdx.x; k < cnt * 6u; k += blockDim.x) {
            const u32* w = t3_tile + (k / 6u) * 25u + (k % 6u) * 4u;
            g[k] = make_uint4(w[0], w[1], w[2], w[3]);
        }
        return;
    }
#endif
    store_fr_global(st, w0);
    store_fr_global(st + 1, w1);
    store_fr_global(st + 2, w2);
}
template <bool FULLFORM>
__global__ void __launch_bounds__(256) poseidon_permute_t3_kernel(PoseidonDims D, PoseidonT3Consts C, Fr* states, size_t n) {
    poseidon_permute_t3_body<FULLFORM, false>(D, C, states, n);
}
template <bool FULLFORM>
__global__ void __launch_bounds__(256) poseidon_permute_t3_staged_kernel(PoseidonDims D, PoseidonT3Consts C, Fr* states, size_t n) {
    poseidon_permute_t3_body<FULLFORM, true>(D, C, states, n);
}
template <bool FULLFORM>
__global__ void __launch_bounds__(256) poseidon_crh_t3_kernel(PoseidonDims D, PoseidonT3Consts C, const Fr* __restrict__ in0,
                                                             const Fr* __restrict__ in1, size_t k, Fr* __restrict__ out, size_t n) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    store_fr_global(out + idx, poseidon_crh_item_t3<FULLFORM>(D, C, in0, in1, k, idx));
}

// Ragged batch (round 5): item i = elements [offsets[i], offsets[i+1]) of `in` -- poseidon::CRH::evaluate takes any &[F]
// (crh/poseidon/mod.rs:30-40) and MerkleTree::new hashes every leaf with its own length (merkle_tree/mod.rs:411-422).  The item's
// sponge runs ceil(k / rate) permutations (one for the empty slice); `order` (may be null) is the launch order sorted by that
// count, longest first (ragged_sort.hpp), results are stored by item index.
template <bool FULLFORM>
__global__ void __launch_bounds__(256) poseidon_crh_ragged_t3_kernel(PoseidonDims D, PoseidonT3Consts C, const Fr* __restrict__ in,
                                                                    const uint64_t* __restrict__ offsets, const u32* __restrict__ order,
                                                                    Fr* __restrict__ out, size_t n) {
    const size_t slot = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= n) return;
    const size_t idx = order ? order[slot] : slot;
    const uint64_t off = offsets[idx], end = offsets[idx + 1];
    // (a pair of device-resident offsets that DECREASES is the empty input, not a length of ~2^64: the lane would never end)
    store_fr_global(out + idx, poseidon_crh_item_t3<FULLFORM>(D, C, in + off, nullptr, end > off ? (size_t)(end - off) : 0, 0));
}

// Path::verify (merkle_tree/mod.rs:172-212) with each lane walking its OWN path: leaf hash, then depth + 1 two-to-one hashes
// of (current, sibling) ordered by the index bit of the level (select_left_right_child :367-381), compared with the root.
// One launch for a whole batch of paths instead of one hash launch + one select launch per level: the chain of a lane
// is the same depth + 2 permutations, but no level waits for the slowest wave of the previous one and nothing but the final
// flag goes back to memory.  t = 3, rate 2, capacity 1, 1 or 2 leaf elements (each hash is ONE permutation of a fresh sponge:
// sponge/poseidon/mod.rs:124-153,324-344); the host keeps the level-by-level form for everything else.  The leaf stage and the
// two-to-one stages share one copy of the permutation (the constants are selected per stage; wave-uniform).
template <bool FULLFORM>
__global__ void __launch_bounds__(256) poseidon_verify_paths_t3_kernel(PoseidonDims DL, PoseidonT3Consts CL, PoseidonDims DT, PoseidonT3Consts CT,
                                                                       const Fr* __restrict__ leaves, u32 leaf_len, const uint64_t* __restrict__ idx,
                                                                       const Fr* __restrict__ sibs, const Fr* __restrict__ auth, u32 depth,
                                                                       const Fr* __restrict__ root, uint8_t* __restrict__ ok, size_t m) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const uint64_t li = idx[i];
    Fr cur = load_fr_global(leaves + i * leaf_len);
    Fr other = leaf_len > 1 ? load_fr_global(leaves + i * leaf_len + 1) : Fr{{0, 0, 0, 0, 0, 0, 0, 0}};
    u32 other_first = 0;  // 1: `other` is the left input
#pragma unroll 1
    for (u32 stage = 0; stage < depth + 2; ++stage) {
        const bool leaf = stage == 0;
        const PoseidonDims D = leaf ? DL : DT;
        const PoseidonT3Consts C = leaf ? CL : CT;
        const Fr a = other_first ? other : cur, b = other_first ? cur : other;
        FP s0 = f29_zero<AKP_PS>();
        FP s1 = f29_weak_norm(FULLFORM ? f29_unpack<AKP_PS>(a) : f29_from_wire<AKP_PS>(a));
        FP s2 = (leaf && leaf_len == 1) ? f29_zero<AKP_PS>() : f29_weak_norm(FULLFORM ? f29_unpack<AKP_PS>(b) : f29_from_wire<AKP_PS>(b));
        poseidon_permute_t3<FULLFORM>(D, C, s0, s1, s2, (leaf && leaf_len == 1) ? 5u : 1u, 2u);
        cur = FULLFORM ? f29_canonical_pack(s1) : f29_to_wire(s1);
        if (stage <= depth) {  // the sibling of the level the NEXT stage hashes: step = stage (0 = leaf level)
            other = load_fr_global(stage == 0 ? sibs + i : auth + i * depth + (depth - stage));
            other_first = (u32)((li >> stage) & 1u);
        }
    }
    ok[i] = fr_eq(cur, load_fr_global(root)) ? 1 : 0;
}

// =============================== any t: LDS "register file" ========================================
// slot s, limb i, lane l -> dword (s*9 + i)*BLOCK + l : consecutive lanes hit consecutive banks.
template <int BLOCK>
struct LdsFile29 {
    u32* base;
    AKP_D FP load(u32 slot) const {
        FP r;
#pragma unroll
        for (int i = 0; i < 9; ++i) r.l[i] = base[(slot * 9 + i) * BLOCK + threadIdx.x];
        return r;
    }
    AKP_D void store(u32 slot, const FP& v) const {
#pragma unroll
        for (int i = 0; i < 9; ++i) base[(slot * 9 + i) * BLOCK + threadIdx.x] = v.l[i];
    }
};
// `File` is LdsFile29<BLOCK> on the device; tests/host_harness instantiates it with a plain array.
// Row sums of the wide kernels: sum_j state[src + j] * row[j] over j in [0, T); `first` (if non-null) replaces the lane-0
// operand (the freshly S-boxed element of a sparse partial round).  poseidon_row_dot returns a weakly normalised value.
// Sums of several normalised terms in signed 32-bit limbs: at most three terms (3 * 2^29 < 2^31) may be pending, so the
// running sum is renormalised before a third term is added (`pending` counts the terms since the last carry step).
#define AKP_ROW_ADD(term)                          \
    do {                                           \
        if (pending == 2u) {                       \
            acc = f29_weak_norm(acc);              \
            pending = 1u;                          \
        }                                          \
        acc = f29_add(acc, (term));                \
        ++pending;                                 \
    } while (0)
// sum_{k < count} get(k) * coefficient co(k): chunks of FIVE terms per Montgomery reduction (balanced constant digits, f29.hpp),
// then one routine for the 4 / 3 / 2 / 1 terms left.  Operands must be normalised lanes (<= 2^29 + 2): (45 + 16) * 2^57 < 2^63.
// Returns the running sum with at most two reduced terms pending (NOT normalised).
template <class Get, class Co>
AKP_HD FP poseidon_terms_sum(u32 count, Get get, Co co) {
    FP acc = f29_zero<AKP_PS>();
    u32 pending = 0;
    u32 k = 0;
#pragma unroll 1
    for (; k + 5 <= count; k += 5)
        AKP_ROW_ADD(f29_dot5(get(k), ldc(co(k)), get(k + 1), ldc(co(k + 1)), get(k + 2), ldc(co(k + 2)), get(k + 3), ldc(co(k + 3)), get(k + 4), ldc(co(k + 4))));
    const u32 left = count - k;
    if (left == 4) AKP_ROW_ADD(f29_dot4(get(k), ldc(co(k)), get(k + 1), ldc(co(k + 1)), get(k + 2), ldc(co(k + 2)), get(k + 3), ldc(co(k + 3))));
    else if (left == 3) AKP_ROW_ADD(f29_dot3(get(k), ldc(co(k)), get(k + 1), ldc(co(k + 1)), get(k + 2), ldc(co(k + 2))));
    else if (left == 2) AKP_ROW_ADD(f29_dot2(get(k), ldc(co(k)), get(k + 1), ldc(co(k + 1))));
    else if (left == 1) AKP_ROW_ADD(f29_mulc(get(k), ldc(co(k))));
    return acc;
}
template <class File>
AKP_HD FP poseidon_row_dot(const File& f, u32 src, u32 T, const F29Pad* __restrict__ row, const FP* first) {
    return f29_weak_norm(poseidon_terms_sum(
        T, [&](u32 k) -> FP { return (first && k == 0) ? *first : f.load(src + k); }, [&](u32 k) { return row + k; }));  // <= 2^29 + 2
}
// the same sum without term `skip` (full form: that coefficient is 1 and the element is added by the caller).
// `first` (if non-null) stands for state[0], as above.  Result NOT normalised: at most two terms pending (<= 2^30 + 2), the
// caller adds one more and normalises.
template <class File>
AKP_HD FP poseidon_row_dot_skip(const File& f, u32 T, const F29Pad* __restrict__ row, u32 skip, const FP* first) {
    return poseidon_terms_sum(
        T - 1,
        [&](u32 c) -> FP {
            const u32 j = c + (c >= skip);
            return (first && j == 0) ? *first : f.load(j);
        },
        [&](u32 c) { return row + (c + (c >= skip)); });
}
#define AKP_POSEIDON_MAX_T 16
// The state occupies slots [0, t) of the file.  Sparse partial rounds update it in place; a dense layer needs all
// old lanes for every new lane, so its t results are parked in a small private (scratch-memory) array and copied
// back -- that happens in the 8 full rounds only, and keeps the LDS footprint at t (not 2t) slots per lane, which
// is what bounds occupancy for wide states.
template <class File>
AKP_HD void poseidon_permute_file(const PoseidonDims& D, const PoseidonConsts& C, const File& f) {
    const u32 T = D.t;
    const u32 half = D.full_rounds / 2;
    const u32 R = D.full_rounds + D.partial_rounds;
    const bool opt = C.sparse != nullptr;
    FP tmp[AKP_POSEIDON_MAX_T];
#pragma unroll 1
    for (u32 r = 0; r < R; ++r) {
        const bool full = (r < half) || (r >= half + D.partial_rounds);
        if (full || !opt) {
            const u32 nsbox = full ? T : 1u;
            const F29Pad* arkr = C.ark + (size_t)r * T;
#pragma unroll 1
            for (u32 e = 0; e < T; ++e) {  // ARK fused with the S-box
                // the lane is a weakly normalised row sum (<= 2^29 + 2), the key has balanced digits: |limb| <= 1.5 * 2^29 + 2
                FP x = f29_add(f.load(e), ldc(arkr + e));
                if (e < nsbox) x = f29_pow_small(x, D.alpha);
                else x = f29_weak_norm(x);  // dense partial round: the lane enters a five-term chunk as it is
                f.store(e, x);
            }
            const bool ff = C.scaled == 3u;  // full form: one matrix per full round, unit diagonal where flagged
            const F29Pad* m = ff ? C.mds + (size_t)(r < half ? r : r - D.partial_rounds) * T * T : ((opt && r + 1 == half) ? C.mpre : C.mds);
            const bool unit_rest = ff && r + 1 != R, unit0 = unit_rest && r + 1 != half;
#pragma unroll 1
            for (u32 i = 0; i < T; ++i) {  // new[i] = sum_j state[j] * m[i][j]
                if (i == 0 ? unit0 : unit_rest) tmp[i] = f29_weak_norm(f29_add(f.load(i), poseidon_row_dot_skip(f, T, m + (size_t)i * T, i, nullptr)));
                else tmp[i] = poseidon_row_dot(f, 0, T, m + (size_t)i * T, nullptr);
            }
#pragma unroll 1
            for (u32 i = 0; i < T; ++i) f.store(i, tmp[i]);
        } else {
            // sparse partial round, in place: lane 0 <- a00*s + u . lanes;  lane i <- lane i + w_i * s
            const u32 j = r - half;
            const F29Pad* sp = C.sparse + (size_t)j * 2 * T;
            const FP sb = f29_pow_small(f29_add(f.load(0), ldc(sp)), D.alpha);
            // full form, last partial round: a00 = 1
            const FP n0 = (C.scaled == 3u && j + 1 == D.partial_rounds) ? f29_weak_norm(f29_add(sb, poseidon_row_dot_skip(f, T, sp + 1, 0, &sb)))
                                                                        : poseidon_row_dot(f, 0, T, sp + 1, &sb);
            // lanes grow < 2^29 per limb per round; a row chunk may consist of three such lanes, and with signed products the
            // column bound is two-sided (27 * 2^58 + 8 * 2^58 > 2^63 for lazy lanes): renormalise every round
            const bool norm = true;
            const bool refold = (j & 31u) == 31u;  // ... and < 2.1p in value: fold back mod p every 32 rounds
#pragma unroll 1
            for (u32 i = 1; i < T; ++i) {
                // lane-1 form (scaled == 2): lane 1 takes the S-box output with coefficient 1
                FP y = (i == 1 && C.scaled >= 2u) ? f29_add(f.load(1), sb) : f29_add(f.load(i), f29_mulc(sb, ldc(sp + T + i)));
                if (refold) y = f29_mulc(y, f29_one<AKP_PS>());
                else if (norm) y = f29_weak_norm(y);
                f.store(i, y);
            }
            f.store(0, n0);
        }
    }
}
template <class File>
AKP_HD Fr poseidon_crh_item(const PoseidonDims& D, const PoseidonConsts& C, const File& f, const Fr* __restrict__ in0,
                            const Fr* __restrict__ in1, size_t k, size_t idx) {
#pragma unroll 1
    for (u32 e = 0; e < D.t; ++e) f.store(e, f29_zero<AKP_PS>());
    size_t done = 0;
    do {
        const size_t take = (k - done) < D.rate ? (k - done) : D.rate;
#pragma unroll 1
        for (size_t j = 0; j < take; ++j) {
            const size_t e = done + j;
            const Fr* src = (in1 == nullptr) ? (in0 + idx * k + e) : (e == 0 ? in0 + idx : in1 + idx);
            const u32 slot = D.capacity + (u32)j;
            // the rate lane holds a weakly normalised MDS output (or zero): keep it that way
            // full form: the lanes hold wire values (x * 2^256), inputs are taken as they are
            const FP in = C.scaled == 3u ? f29_unpack<AKP_PS>(load_fr_global(src)) : f29_from_wire<AKP_PS>(load_fr_global(src));
            f.store(slot, f29_weak_norm(f29_add(f.load(slot), in)));
        }
        done += take;
        poseidon_permute_file(D, C, f);
    } while (done < k);
    // full form: the lane is a row sum of up to six reduced terms (|v| < 16p): the wide canonicalisation
    return C.scaled == 3u ? f29_canonical_pack<AKP_PS, true>(f.load(D.capacity)) : f29_to_wire(f.load(D.capacity));
}

template <int BLOCK>
__global__ void __launch_bounds__(BLOCK) poseidon_permute_kernel(PoseidonDims D, PoseidonConsts C, Fr* states, size_t n) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    LdsFile29<BLOCK> f{reinterpret_cast<u32*>(smem)};
    const size_t idx = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (idx >= n) return;  // lanes never exchange data: no barriers anywhere
    Fr* st = states + idx * D.t;
#pragma unroll 1
    for (u32 e = 0; e < D.t; ++e) f.store(e, C.scaled == 3u ? f29_unpack<AKP_PS>(load_fr_global(st + e)) : f29_from_wire<AKP_PS>(load_fr_global(st + e)));
    poseidon_permute_file(D, C, f);
#pragma unroll 1
    for (u32 e = 0; e < D.t; ++e) store_fr_global(st + e, C.scaled == 3u ? f29_canonical_pack<AKP_PS, true>(f.load(e)) : f29_to_wire(f.load(e)));
}
template <int BLOCK>
__global__ void __launch_bounds__(BLOCK) poseidon_crh_kernel(PoseidonDims D, PoseidonConsts C, const Fr* __restrict__ in0,
                                                           const Fr* __restrict__ in1, size_t k, Fr* __restrict__ out, size_t n) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    LdsFile29<BLOCK> f{reinterpret_cast<u32*>(smem)};
    const size_t idx = (size_t)blockIdx.x * BLOCK + threadIdx.x;
    if (idx >= n) return;
    store_fr_global(out + idx, poseidon_crh_item(D, C, f, in0, in1, k, idx));
}

// =============================== t = 4 .. 9: register-resident state (round 2) ==================================
// The default rate-3 / rate-4 instances (alpha = 5, 8 + 56 rounds) spend 56 of 64 rounds in the sparse partial round,
// which touches fixed lanes: with T a compile-time constant that round is straight-line code on registers.  The eight
// dense rounds would need T row sums and T S-boxes unrolled (too much code for the 64 KB instruction cache at T = 5),
// so they keep ONE copy of the S-box and ONE of each row shape and ROTATE the state through position 0 instead:
//   S-boxes:  T times { S-box on s[0] with key e; rotate s }                       -> s back in order
//   rows:     T times { row i = sum_k s[k] * m[i][(i + k) mod T]; push into n; rotate s }   (s[0] is lane i: a unit diagonal
//             is the STATIC operand 0, the coefficients are scalar loads at a uniform, rotating address)
// 2 * 9 * T register moves per iteration against 350-620 instructions of row arithmetic, in 8 of 64 rounds.
// Constants: the full form (C.scaled == 3, FF) or the lane-1 form (C.scaled == 2) of poseidon_opt.hpp -- the same arrays
// the LDS-file kernels read, so the digests are identical to theirs by construction of the same sums.
template <u32 T>
AKP_HD void reg_rotate(FP (&s)[T]) {
    const FP first = s[0];
#pragma unroll
    for (u32 k = 0; k + 1 < T; ++k) s[k] = s[k + 1];
    s[T - 1] = first;
}
AKP_HD u32 reg_col(u32 i, u32 k, u32 T) {
    const u32 c = i + k;
    return c >= T ? c - T : c;
}
// sum_{k = K0}^{T - 1} s[k] * row[(i + k) mod T]: three to five terms under one Montgomery reduction (six: + one product).
template <u32 T, u32 K0>
AKP_HD FP reg_row_sum(const FP (&s)[T], const F29Pad* __restrict__ row, u32 i) {
    constexpr u32 N = T - K0;
    static_assert(N >= 3 && N <= 9, "register kernels cover t = 4 .. 9");
    // the constants carry balanced digits (f29_balance), so up to five terms share ONE reduction; six to nine terms are five + the
    // rest under a second reduction, and the two-term sum is carried back to <= 2^29 + 1
#define AKP_RS(k) s[K0 + (k)], ldc(row + reg_col(i, K0 + (k), T))
    if constexpr (N == 3) return f29_dot3(AKP_RS(0), AKP_RS(1), AKP_RS(2));
    else if constexpr (N == 4) return f29_dot4(AKP_RS(0), AKP_RS(1), AKP_RS(2), AKP_RS(3));
    else if constexpr (N == 5) return f29_dot5(AKP_RS(0), AKP_RS(1), AKP_RS(2), AKP_RS(3), AKP_RS(4));
    else {
        const FP head = f29_dot5(AKP_RS(0), AKP_RS(1), AKP_RS(2), AKP_RS(3), AKP_RS(4));
        if constexpr (N == 6) return f29_weak_norm(f29_add(head, f29_mulc(AKP_RS(5))));
        else if constexpr (N == 7) return f29_weak_norm(f29_add(head, f29_dot2(AKP_RS(5), AKP_RS(6))));
        else if constexpr (N == 8) return f29_weak_norm(f29_add(head, f29_dot3(AKP_RS(5), AKP_RS(6), AKP_RS(7))));
        else return f29_weak_norm(f29_add(head, f29_dot4(AKP_RS(5), AKP_RS(6), AKP_RS(7), AKP_RS(8))));
    }
#undef AKP_RS
}
template <u32 T, bool FF>
AKP_HD void poseidon_permute_reg(const PoseidonDims& D, const PoseidonConsts& C, FP (&s)[T], u32 need_lanes = 0xffffu) {
    const u32 half = D.full_rounds / 2;
    const u32 R = D.full_rounds + D.partial_rounds;
#pragma unroll 1
    for (u32 r = 0; r < R; ++r) {
        const bool full = (r < half) || (r >= half + D.partial_rounds);
        if (full) {
            const F29Pad* arkr = C.ark + (size_t)r * T;
#pragma unroll 1
            for (u32 e = 0; e < T; ++e) {  // ARK + S-box, lane e at position 0
                s[0] = f29_pow_small(f29_add(s[0], ldc(arkr + e)), D.alpha);
                reg_rotate<T>(s);
            }
            const F29Pad* m = FF ? C.mds + (size_t)(r < half ? r : r - D.partial_rounds) * T * T : ((r + 1 == half) ? C.mpre : C.mds);
            const bool unit_rest = FF && r + 1 != R, unit0 = unit_rest && r + 1 != half;
            const u32 need = (r + 1 == R) ? need_lanes : 0xffffu;
            FP n[T];
#pragma unroll
            for (u32 k = 0; k < T; ++k) n[k] = s[k];
#pragma unroll 1
            for (u32 i = 0; i < T; ++i) {  // row i; s[k] holds lane (i + k) mod T
                FP v = s[0];
                if ((need >> i) & 1u) {
                    if (i == 0 ? unit0 : unit_rest) v = f29_weak_norm(f29_add(s[0], reg_row_sum<T, 1>(s, m + (size_t)i * T, i)));
                    else v = reg_row_sum<T, 0>(s, m + (size_t)i * T, i);  // one reduction: already normalised
                }
                reg_rotate<T>(n);  // n = (n_1 .. n_{T-1}, new): after T pushes the rows are in order
                n[T - 1] = v;
                reg_rotate<T>(s);
            }
#pragma unroll
            for (u32 k = 0; k < T; ++k) s[k] = n[k];
        } else {
            // sparse partial round (same schedule as poseidon_permute_file): lane 0 <- a00 * sb + u . lanes,
            // lane 1 += sb (lane-1 form), lane i += w_i * sb
            const u32 j = r - half;
            const F29Pad* sp = C.sparse + (size_t)j * 2 * T;
            const FP sb = f29_pow_small(f29_add(s[0], ldc(sp)), D.alpha);
            FP t[T];  // (sb, s_1 .. s_{T-1}): the operands of row 0 against sp[1 .. T]
            t[0] = sb;
#pragma unroll
            for (u32 k = 1; k < T; ++k) t[k] = s[k];
            const FP n0 = (FF && j + 1 == D.partial_rounds) ? f29_weak_norm(f29_add(sb, reg_row_sum<T, 1>(t, sp + 1, 0)))  // a00 = 1
                                                            : reg_row_sum<T, 0>(t, sp + 1, 0);
            const bool refold = (j & 31u) == 31u;  // lanes gain < 2.1p per round: fold back mod p every 32 rounds
#pragma unroll
            for (u32 i = 1; i < T; ++i) {
                FP y = (i == 1) ? f29_add(s[1], sb) : f29_add(s[i], f29_mulc(sb, ldc(sp + T + i)));
                if (refold) y = f29_mulc(y, f29_one<AKP_PS>());
                else y = f29_weak_norm(y);  // every round: four / five terms share a column (36 / 45 products + 16 for the reduction)
                s[i] = y;
            }
            s[0] = n0;
        }
    }
}
template <u32 T, bool FF>
AKP_HD FP reg_load(const Fr* p) {
    return FF ? f29_unpack<AKP_PS>(load_fr_global(p)) : f29_from_wire<AKP_PS>(load_fr_global(p));
}
template <u32 T, bool FF>
AKP_HD Fr reg_store(const FP& v) {  // full form: the wide canonicalisation is kept for safety (a last-round row is one reduced term)
    return FF ? f29_canonical_pack<AKP_PS, true>(v) : f29_to_wire(v);
}
// fixed-length sponge CRH on a fresh sponge (same contract as poseidon_crh_item)
template <u32 T, bool FF>
AKP_HD Fr poseidon_crh_item_reg(const PoseidonDims& D, const PoseidonConsts& C, const Fr* __restrict__ in0, const Fr* __restrict__ in1, size_t k,
                                size_t idx) {
    FP s[T];
#pragma unroll
    for (u32 e = 0; e < T; ++e) s[e] = f29_zero<AKP_PS>();
    size_t done = 0;
    do {
        const size_t take = (k - done) < D.rate ? (k - done) : D.rate;
#pragma unroll
        for (u32 e = 0; e < T; ++e) {  // absorb_internal :124-153: state[capacity + j] += input j of this block
            if (e >= D.capacity && (size_t)(e - D.capacity) < take) {
                const size_t el = done + (e - D.capacity);
                const Fr* src = (in1 == nullptr) ? (in0 + idx * k + el) : (el == 0 ? in0 + idx : in1 + idx);
                s[e] = f29_weak_norm(f29_add(s[e], reg_load<T, FF>(src)));
            }
        }
        done += take;
        // only state[capacity] of the last permutation is squeezed (:156-186)
        poseidon_permute_reg<T, FF>(D, C, s, done < k ? 0xffffu : (1u << D.capacity));
    } while (done < k);
    FP out = s[0];
#pragma unroll
    for (u32 e = 1; e < T; ++e)
#pragma unroll
        for (int i = 0; i < 9; ++i) out.l[i] = D.capacity == e ? s[e].l[i] : out.l[i];
    return reg_store<T, FF>(out);
}
template <u32 T, bool FF>
__global__ void __launch_bounds__(256) poseidon_permute_reg_kernel(PoseidonDims D, PoseidonConsts C, Fr* states, size_t n) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    Fr* st = states + idx * T;
    FP s[T];
#pragma unroll
    for (u32 e = 0; e < T; ++e) s[e] = reg_load<T, FF>(st + e);
    poseidon_permute_reg<T, FF>(D, C, s);
#pragma unroll
    for (u32 e = 0; e < T; ++e) store_fr_global(st + e, reg_store<T, FF>(s[e]));
}
template <u32 T, bool FF>
__global__ void __launch_bounds__(256) poseidon_crh_reg_kernel(PoseidonDims D, PoseidonConsts C, const Fr* __restrict__ in0, const Fr* __restrict__ in1,
                                                              size_t k, Fr* __restrict__ out, size_t n) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    store_fr_global(out + idx, poseidon_crh_item_reg<T, FF>(D, C, in0, in1, k, idx));
}

#if defined(__HIPCC__)
// =============================== any t: one wave per state lane ================================================
// Workgroup of t waves per 64 items; wave w keeps lane w of the 64 states in registers.  Every round publishes one
// value per wave in a double-buffered LDS tile (t x 9 x 64 dwords per buffer) and has one barrier:
//   full round      wave w: S-box of its lane, publish, then its MDS row over the t published values;
//   sparse partial  wave 0: S-box, publish s;  wave i > 0: publish u_i * y_i;  then
//                   wave 0: a00 * s + sum of the published terms;  wave i: y_i += w_i * s;
//   dense partial   (parameter sets poseidon_optimize rejects) like a full round with the S-box on wave 0 only.
// Same constants, same schedule and the same field routines as poseidon_permute_file, so the digests are identical.
// No lane of a wave ever touches another item's data; all operand traffic of the arithmetic is registers + SGPR
// constants, the LDS tile only carries the t values a round exchanges.
struct CoopTile {
    u32* base;  // [2][T][9][64]
    u32 T, lane;
    AKP_D void put(u32 buf, u32 slot, const FP& v) const {
#pragma unroll
        for (int i = 0; i < 9; ++i) base[((buf * T + slot) * 9 + i) * 64 + lane] = v.l[i];
    }
    AKP_D FP get(u32 buf, u32 slot) const {
        FP v;
#pragma unroll
        for (int i = 0; i < 9; ++i) v.l[i] = base[((buf * T + slot) * 9 + i) * 64 + lane];
        return v;
    }
};
// sum_j published[j] * row[j], three terms per Montgomery reduction; result weakly normalised
AKP_D FP coop_row_dot(const CoopTile& tile, u32 buf, const F29Pad* __restrict__ row) {
    return f29_weak_norm(poseidon_terms_sum(
        tile.T, [&](u32 k) -> FP { return tile.get(buf, k); }, [&](u32 k) { return row + k; }));
}
// one permutation; x is lane w of the state (weakly normalised in and out); buf is the tile buffer to use next
AKP_D void poseidon_permute_coop(const PoseidonDims& D, const PoseidonConsts& C, const CoopTile& tile, u32 w, FP& x, u32& buf) {
    const u32 T = D.t;
    const u32 half = D.full_rounds / 2;
    const u32 R = D.full_rounds + D.partial_rounds;
    const bool opt = C.sparse != nullptr;
#pragma unroll 1
    for (u32 r = 0; r < R; ++r) {
        const bool full = (r < half) || (r >= half + D.partial_rounds);
        if (full || !opt) {
            x = f29_add(x, ldc(C.ark + (size_t)r * T + w));  // balanced key digits: |limb| <= 1.5 * 2^29 + small
            if (full || w == 0) x = f29_pow_small(x, D.alpha);
            else x = f29_weak_norm(x);  // dense partial round: published as it is into five-term chunks
            tile.put(buf, w, x);
            __syncthreads();
            const F29Pad* m = ((opt && r + 1 == half) ? C.mpre : C.mds) + (size_t)w * T;
            x = coop_row_dot(tile, buf, m);
        } else {
            const u32 j = r - half;
            const F29Pad* sp = C.sparse + (size_t)j * 2 * T;  // q0, a00, u_1..u_{T-1}, w_1..w_{T-1}
            if (w == 0) {
                x = f29_pow_small(f29_add(x, ldc(sp)), D.alpha);
                tile.put(buf, 0, x);
            } else {
                tile.put(buf, w, f29_mulc(x, ldc(sp + 1 + w)));
            }
            __syncthreads();
            if (w == 0) {
                FP acc = (C.scaled == 1u && j + 1 < D.partial_rounds) ? x : f29_mulc(x, ldc(sp + 1));
                u32 pending = 1;
#pragma unroll 1
                for (u32 i = 1; i < T; ++i) AKP_ROW_ADD(tile.get(buf, i));  // signed limbs: at most three terms pending
                x = f29_weak_norm(acc);
            } else {
                x = f29_add(x, f29_mulc(tile.get(buf, 0), ldc(sp + T + w)));
                if ((j & 31u) == 31u) x = f29_mulc(x, f29_one<AKP_PS>());
                else if ((j & 1u) || j + 1 == D.partial_rounds) x = f29_weak_norm(x);
            }
        }
        buf ^= 1u;
    }
}
__global__ void __launch_bounds__(1024) poseidon_permute_coop_kernel(PoseidonDims D, PoseidonConsts C, Fr* states, size_t n) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const u32 w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const CoopTile tile{reinterpret_cast<u32*>(smem), D.t, threadIdx.x & 63u};
    const size_t item = (size_t)blockIdx.x * 64 + tile.lane;
    const size_t idx = item < n ? item : n - 1;  // every lane walks all barriers; only valid items are stored
    FP x = f29_weak_norm(f29_from_wire<AKP_PS>(load_fr_global(states + idx * D.t + w)));
    u32 buf = 0;
    poseidon_permute_coop(D, C, tile, w, x, buf);
    if (item < n) store_fr_global(states + item * D.t + w, f29_to_wire(x));
}
// fixed-length sponge CRH on a fresh sponge (same contract as poseidon_crh_item)
__global__ void __launch_bounds__(1024) poseidon_crh_coop_kernel(PoseidonDims D, PoseidonConsts C, const Fr* __restrict__ in0,
                                                                const Fr* __restrict__ in1, size_t k, Fr* __restrict__ out, size_t n) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const u32 w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const CoopTile tile{reinterpret_cast<u32*>(smem), D.t, threadIdx.x & 63u};
    const size_t item = (size_t)blockIdx.x * 64 + tile.lane;
    const size_t idx = item < n ? item : n - 1;
    FP x = f29_zero<AKP_PS>();
    u32 buf = 0;
    size_t done = 0;
    do {
        const size_t take = (k - done) < D.rate ? (k - done) : D.rate;
        if (w >= D.capacity && (size_t)(w - D.capacity) < take) {  // absorb_internal :124-153: state[capacity + j] += input
            const size_t e = done + (w - D.capacity);
            const Fr* src = (in1 == nullptr) ? (in0 + idx * k + e) : (e == 0 ? in0 + idx : in1 + idx);
            x = f29_weak_norm(f29_add(x, f29_from_wire<AKP_PS>(load_fr_global(src))));
        }
        done += take;
        poseidon_permute_coop(D, C, tile, w, x, buf);
    } while (done < k);
    if (w == D.capacity && item < n) store_fr_global(out + item, f29_to_wire(x));  // squeeze_internal(0, 1)
}
#endif

}  // namespace akp
