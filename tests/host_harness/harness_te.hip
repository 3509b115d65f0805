// Test-only harness, curve-hash half (see harness.hip): the product's table construction, digit accumulation, shared-inversion
// finalisation and digest serialisation (crypto_primitives_amd/csrc/te_kernels.hpp) on the CPU.  A unit of its own so that the
// two halves compile in parallel; both are linked into harness.so.
#include <hip/hip_runtime.h>
#include <vector>
#include "../../crypto_primitives_amd/csrc/fr.hpp"
#include "../../crypto_primitives_amd/csrc/f29.hpp"
#include "../../crypto_primitives_amd/csrc/te_kernels.hpp"
#include "../../crypto_primitives_amd/csrc/ragged_sort.hpp"
using namespace akp;

extern "C" {
// LUT construction exactly as capi_te.hip does it: kind 0 -> Pedersen with digit width D (lut: [ceil(n_gen/D)][2^D]);
// kind 1 -> Bowe-Hopwood single table lut1 [n_gen][4] and, when group > 1, group table lut [n_gen/G][2^(3G-1)]
// (hh_te_crh then takes D = group for kind 1).
void hh_te_build_lut(int kind, const Fr* gens, uint32_t W, uint32_t N, uint32_t D, uint32_t group, TeEntry* lut, TeEntry* lut1) {
    const u32 n_gen = W * N;
    if (kind == 2) {  // Pedersen, signed-subset table: lut [n_digits][2^(D-1)], lut1 = cprefix [n_digits + 1]
        std::vector<NielsPad> half(n_gen);
        for (u32 g = 0; g < n_gen; ++g) {
            Niels h;
            (void)te_half_generator(gens, g, h);
            store_niels(&half[g], h);
        }
        const u32 n_digits = (n_gen + D - 1) / D;
        for (u32 i = 0; i < (n_digits << (D - 1)); ++i) store_niels(lut + i, te_pedersen_slut_entry(half.data(), n_gen, D, i));
        for (u32 k = 0; k <= n_digits; ++k) store_niels(lut1 + k, te_pedersen_cprefix_entry(half.data(), n_gen, D, k));
        return;
    }
    if (kind == 0) {
        const u32 entries = ((n_gen + D - 1) / D) << D;
        for (u32 i = 0; i < entries; ++i) store_niels(lut + i, te_pedersen_lut_entry(gens, n_gen, D, i));
        return;
    }
    for (u32 i = 0; i < n_gen * 4; ++i) store_niels(lut1 + i, te_bh_lut_entry(gens, i));
    if (group > 1)
        for (u32 i = 0; i < ((n_gen / group) << (3 * group - 1)); ++i) store_niels(lut + i, te_bh_lutg_entry(gens, group, i));
}
void hh_te_crh(int kind, const TeEntry* lut, const TeEntry* lut1, const uint8_t* msgs, size_t n, size_t msg_len, uint32_t D,
               uint32_t groups, uint32_t steps, size_t lanes, Fr* out) {
    std::vector<F29Pad> xyz(n * 3), prefix(n);
    for (size_t i = 0; i < n; ++i) {
        Ext a = kind == 0 ? te_accumulate_item<0>(lut, lut1, msgs + i * msg_len, msg_len, D, groups, steps)
                          : (kind == 2 ? te_accumulate_item<2>(lut, lut1, msgs + i * msg_len, msg_len, D, groups, steps)
                                       : te_accumulate_item<1>(lut, lut1, msgs + i * msg_len, msg_len, D, groups, steps));
        f29_store_pad(&xyz[3 * i], a.X); f29_store_pad(&xyz[3 * i + 1], a.Y); f29_store_pad(&xyz[3 * i + 2], a.Z);
    }
    for (size_t l = 0; l < lanes && l < n; ++l) {
        if (kind != 1) te_finalize_lane<0>(xyz.data(), prefix.data(), out, n, lanes, l);
        else te_finalize_lane<1>(xyz.data(), prefix.data(), out, n, lanes, l);
    }
}
// The two-part construction of the wide tables exactly as capi_te.hip's te_build_wide launches it (part tables, then one addition
// per entry with the inversion shared by the AKP_TE_BUILD_RUN entries of a lane), for `units` digits (kind 2, shape = D) / chunk
// groups (kind 1, shape = G): lut gets units << (D - 1) / units << (3G - 1) entries.
void hh_te_build_wide(int kind, const Fr* gens, uint32_t W, uint32_t N, uint32_t shape, uint32_t units, TeEntry* lut) {
    const u32 n_gen = W * N;
    const u32 k_lo = kind == 2 ? (shape - 1) / 2 : shape / 2;
    const size_t n_lo = kind == 2 ? (size_t)units << k_lo : (size_t)units << (3 * k_lo - 1);
    const size_t n_hi = kind == 2 ? (size_t)units << (shape - 1 - k_lo) : (size_t)units << (3 * (shape - k_lo));
    const size_t entries = kind == 2 ? (size_t)units << (shape - 1) : (size_t)units << (3 * shape - 1);
    std::vector<TeEntry> lo(n_lo), hi(n_hi);
    if (kind == 2) {
        std::vector<NielsPad> half(n_gen);
        for (u32 g = 0; g < n_gen; ++g) {
            Niels h;
            (void)te_half_generator(gens, g, h);
            store_niels(&half[g], h);
        }
        for (size_t i = 0; i < n_lo + n_hi; ++i) te_pedersen_sparts_item(half.data(), n_gen, shape, units, k_lo, lo.data(), hi.data(), (u32)i);
    } else {
        for (size_t i = 0; i < n_lo + n_hi; ++i) te_bh_parts_item(gens, shape, k_lo, units, lo.data(), hi.data(), (u32)i);
    }
    const size_t per_block = 256 * (size_t)AKP_TE_BUILD_RUN;
    for (size_t block = 0; block * per_block < entries; ++block)
        for (size_t lane = 0; lane < 256; ++lane) {
            if (kind == 2) te_build_combine_lane<2>(lo.data(), hi.data(), shape, k_lo, entries, lut, block * per_block + lane, 256);
            else te_build_combine_lane<1>(lo.data(), hi.data(), shape, k_lo, entries, lut, block * per_block + lane, 256);
        }
}
// Bowe-Hopwood remainder table (te_build_bh_remainder): the r chunks from chunk `first`, indexed by the raw message bits, with the
// constant of the zero-padded tail chunks [tail_from, tail_to) folded in (tail_from >= tail_to: none); lut1 = the one-chunk table
void hh_te_build_remainder(const Fr* gens, const TeEntry* lut1, uint32_t first, uint32_t r, uint32_t tail_from, uint32_t tail_to, TeEntry* out) {
    TeEntry tail;
    const bool has_tail = tail_from < tail_to;
    if (has_tail) {
        Ext acc = ext_from_niels(load_niels(lut1 + (size_t)tail_from * 4u));
        for (u32 c = tail_from + 1; c < tail_to; ++c) acc = te_madd(acc, load_niels(lut1 + (size_t)c * 4u));
        store_niels(&tail, niels_of_ext(acc));
    }
    for (u32 i = 0; i < (1u << (3 * r)); ++i) store_niels(out + i, te_bh_remainder_entry(gens, first, r, has_tail ? &tail : nullptr, i));
}
// the small-batch kernel's arithmetic (te_crh_small_kernel) on the CPU: `split` strided partial sums, a binary tree of
// full additions, one inversion per message
void hh_te_crh_split(int kind, const TeEntry* lut, const TeEntry* lut1, const uint8_t* msgs, size_t n, size_t msg_len, uint32_t D,
                     uint32_t groups, uint32_t steps, uint32_t split, Fr* out) {
    for (size_t i = 0; i < n; ++i) {
        std::vector<Ext> part(split);
        for (u32 j = 0; j < split; ++j)
            part[j] = kind == 0 ? te_accumulate_strided<0>(lut, lut1, msgs + i * msg_len, msg_len, D, groups, steps, j, split)
                                : (kind == 2 ? te_accumulate_strided<2>(lut, lut1, msgs + i * msg_len, msg_len, D, groups, steps, j, split)
                                             : te_accumulate_strided<1>(lut, lut1, msgs + i * msg_len, msg_len, D, groups, steps, j, split));
        for (u32 stride = 1; stride < split; stride <<= 1)
            for (u32 j = 0; j + stride < split; j += 2 * stride) part[j] = te_add_ext(part[j], part[j + stride]);
        const FS zi = f29_inv(part[0].Z);
        if (kind != 1) { out[2 * i] = f29_to_wire(f29_mul(part[0].X, zi)); out[2 * i + 1] = f29_to_wire(f29_mul(part[0].Y, zi)); }
        else out[i] = f29_to_wire(f29_mul(part[0].X, zi));
    }
}
// te_accumulate_ragged_kernel's per-item code (round 5): item i = bytes [offsets[i], offsets[i+1]) of msgs, its table steps from its own
// length (te_item_steps), message bytes through MsgAny (1..3-byte messages are NOT padded by anybody), then the shared finalisation.
// `msgs` is exactly offsets[n] bytes long in the tests: under ASan every load past an item's end that leaves the buffer is caught.
void hh_te_crh_ragged(int kind, const TeEntry* lut, const TeEntry* lut1, const uint8_t* msgs, const uint64_t* offsets, size_t n, uint32_t D,
                      uint32_t n_gen, uint32_t units_built, size_t lanes, Fr* out) {
    std::vector<F29Pad> xyz(n * 3), prefix(n);
    for (size_t i = 0; i < n; ++i) {
        const size_t len = (size_t)(offsets[i + 1] - offsets[i]);
        const MsgAny m{msgs + offsets[i]};
        u32 groups, steps;
        Ext a;
        if (kind == 0) { te_item_steps<0>(n_gen, D, len, units_built, &groups, &steps); a = te_accumulate_item<0>(lut, lut1, m, len, D, groups, steps); }
        else if (kind == 2) { te_item_steps<2>(n_gen, D, len, units_built, &groups, &steps); a = te_accumulate_item<2>(lut, lut1, m, len, D, groups, steps); }
        else { te_item_steps<1>(n_gen, D, len, units_built, &groups, &steps); a = te_accumulate_item<1>(lut, lut1, m, len, D, groups, steps); }
        f29_store_pad(&xyz[3 * i], a.X); f29_store_pad(&xyz[3 * i + 1], a.Y); f29_store_pad(&xyz[3 * i + 2], a.Z);
    }
    for (size_t l = 0; l < lanes && l < n; ++l) {
        if (kind != 1) te_finalize_lane<0>(xyz.data(), prefix.data(), out, n, lanes, l);
        else te_finalize_lane<1>(xyz.data(), prefix.data(), out, n, lanes, l);
    }
}
// sort key of an item (ragged_sort.hpp) and the step count the kernel computes for it (te_item_steps): must agree
uint32_t hh_ragged_key(uint32_t mode, uint32_t unit, uint32_t cap, uint64_t len) { return ragged_key_of(RaggedKey{mode, unit, cap}, len); }
uint32_t hh_te_item_steps(int kind, uint32_t n_gen, uint32_t D, size_t len, uint32_t units_built, uint32_t* groups) {
    u32 steps;
    if (kind == 0) te_item_steps<0>(n_gen, D, len, units_built, groups, &steps);
    else if (kind == 2) te_item_steps<2>(n_gen, D, len, units_built, groups, &steps);
    else te_item_steps<1>(n_gen, D, len, units_built, groups, &steps);
    return steps;
}
// the w message bits at bit offset o as a table step reads them: ONE 32-bit window (msg_load) and a shift (msg_combine)
uint32_t hh_te_window(const uint8_t* msg, size_t len, size_t o, uint32_t w) { return msg_combine(msg_load(msg, len, o), len, o, w); }
// 1 when the generator is in the prime-order subgroup (2 * (G / 2) == G)
int hh_te_in_subgroup(const Fr* gen_affine) {
    Niels h;
    return te_half_generator(gen_affine, 0, h) ? 1 : 0;
}
void hh_te_serialize_pairs(const Fr* left, const Fr* right, uint32_t fe, size_t buflen, uint8_t* buf, size_t n) {
    for (size_t t = 0; t < n * 2 * fe; ++t) te_serialize_pair_fe(left, right, fe, buflen, buf, t);
}
}
