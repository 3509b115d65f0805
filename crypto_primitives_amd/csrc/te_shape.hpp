// te_shape.hpp -- the arithmetic that decides the shape of a Pedersen / Bowe-Hopwood table and how many table steps a message
// takes.  Host only, no HIP, no state: capi_te.hip uses it for the handles, tests/cpp/test_te_shape.cpp checks it against brute
// force without a GPU.  (What the shapes cost and buy: the comment above akp_te_params_create_shaped in capi_te.hip.)
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdint>

namespace akp {
namespace te_shape {

constexpr uint32_t MAX_DIGIT = 24, MAX_GROUP = 8;  // a step reads its message bits through one 32-bit window
constexpr size_t ENTRY_BYTES_DEFAULT = 128;
constexpr size_t MAX_ENTRIES = (size_t)1 << 32;    // entry indices are 32-bit in the kernels

// entries of the COMPLETE table: Pedersen signed-subset table of D-bit digits over n_gen generators (2^(D-1) per digit, the last
// digit may be clipped but is stored in full); Bowe-Hopwood table of groups of G chunks (2^(3G-1) per group, whole groups only)
inline size_t pedersen_entries(size_t n_gen, uint32_t D) { return ((n_gen + D - 1) / D) << (D - 1); }
inline size_t bh_entries(size_t n_gen, uint32_t G) { return (n_gen / G) << (3 * G - 1); }

// widest digit whose complete table fits `budget` bytes, then the narrowest digit with the same number of digits (64 generators:
// 22 bits give the 3 steps that 24 bits give); never below 2 bits
inline uint32_t pick_digit(size_t n_gen, size_t budget, size_t entry_bytes = ENTRY_BYTES_DEFAULT) {
    uint32_t D = MAX_DIGIT;
    while (D > 2 && (pedersen_entries(n_gen, D) * entry_bytes > budget || pedersen_entries(n_gen, D) >= MAX_ENTRIES)) --D;
    while (D > 2 && (n_gen + D - 2) / (D - 1) == (n_gen + D - 1) / D) --D;
    return D;
}
// largest group (<= n_gen chunks) whose complete table fits `budget` bytes; 1 = no group table
inline uint32_t pick_group(size_t n_gen, size_t budget, size_t entry_bytes = ENTRY_BYTES_DEFAULT) {
    uint32_t G = MAX_GROUP;
    while (G > 1 && (n_gen < G || bh_entries(n_gen, G) * entry_bytes > budget || bh_entries(n_gen, G) >= MAX_ENTRIES)) --G;
    return G;
}
// table steps of a message of msg_len bytes: Pedersen pads with zero bytes (zero digits select the identity: the sum stops at the
// last digit the message reaches); Bowe-Hopwood stops at ceil(bits / 3) chunks = `groups` full groups + left-over chunks (each a
// step of its own here; capi_te.hip turns them into one remainder step)
inline void pedersen_steps(size_t n_gen, uint32_t D, size_t msg_len, uint32_t* steps) {
    const size_t used = std::min<size_t>(msg_len * 8, n_gen);
    *steps = (uint32_t)((used + D - 1) / D);
}
inline void bh_steps(size_t n_gen, uint32_t G, size_t msg_len, uint32_t* groups, uint32_t* steps) {
    const size_t chunks = std::min<size_t>((msg_len * 8 + 2) / 3, n_gen);
    if (G > 1) {
        *groups = (uint32_t)(chunks / G);
        *steps = (uint32_t)(chunks / G + chunks % G);
    } else {
        *groups = 0;
        *steps = (uint32_t)chunks;
    }
}
// units (digits / groups) to build when a message needs `needed` of `total` and `built` exist: at least twice the old coverage,
// so that messages of slowly growing length rebuild the table O(log) times
inline uint32_t grow_target(uint32_t needed, uint32_t built, uint32_t total) { return std::min(total, std::max(needed, 2 * built)); }

}  // namespace te_shape
}  // namespace akp
