// csrc/te_shape.hpp against brute force (host only): the widest table shape a budget admits, the step counts of a message, the
// growth rule of the lazily built tables.  Built and run by tests/test_cpp_header.py::test_te_shape_arithmetic.
#include "../../crypto_primitives_amd/csrc/te_shape.hpp"
#include <cstdio>
#include <cstdlib>
using namespace akp::te_shape;
#define REQUIRE(c) do { if (!(c)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

int main() {
    // the BASELINE windows on an idle MI355X (72 GiB budget) and with the cache-sized budget of rounds 1-3
    const size_t big = (size_t)72 << 30, small = (size_t)320 << 20;
    REQUIRE(pick_digit(1024, big) == 24 && pedersen_entries(1024, 24) * 128 == (size_t)43 << 30);
    REQUIRE(pick_digit(1024, small) == 16 && pick_digit(1024, (size_t)4 << 30) == 20 && pick_digit(1024, (size_t)16 << 30) == 22);
    REQUIRE(pick_group(567, big) == 8 && bh_entries(567, 8) * 128 == (size_t)70 << 30);
    REQUIRE(pick_group(567, small) == 5 && pick_group(567, (size_t)4 << 30) == 6 && pick_group(567, (size_t)16 << 30) == 7);
    REQUIRE(pick_digit(64, big) == 22 && pick_digit(8, big) == 8 && pick_digit(1, big) == 2 && pick_group(3, big) == 3 && pick_group(1, big) == 1);
    // brute force: the pick fits, one more bit / chunk would not (or would not reduce the digit count), for many windows and budgets
    for (size_t n_gen = 1; n_gen <= 3000; n_gen += (n_gen < 40 ? 1 : 37)) {
        for (int lb = 10; lb <= 38; lb += 2) {
            const size_t budget = (size_t)1 << lb;
            const uint32_t D = pick_digit(n_gen, budget), G = pick_group(n_gen, budget);
            REQUIRE(D >= 2 && D <= MAX_DIGIT && G >= 1 && G <= MAX_GROUP);
            REQUIRE(D == 2 || pedersen_entries(n_gen, D) * 128 <= budget);
            const size_t digits = (n_gen + D - 1) / D;
            for (uint32_t w = D + 1; w <= MAX_DIGIT; ++w)  // a wider digit either does not fit or needs as many digits
                REQUIRE(pedersen_entries(n_gen, w) * 128 > budget || pedersen_entries(n_gen, w) >= MAX_ENTRIES || (n_gen + w - 1) / w == digits);
            REQUIRE(D == 2 || (n_gen + D - 2) / (D - 1) > digits);  // a narrower digit needs more digits
            REQUIRE(G == 1 || (G <= n_gen && bh_entries(n_gen, G) * 128 <= budget));
            for (uint32_t w = G + 1; w <= MAX_GROUP; ++w) REQUIRE(w > n_gen || bh_entries(n_gen, w) * 128 > budget);
        }
    }
    // steps: every message bit / chunk is covered exactly once, nothing past the window
    for (size_t n_gen : {(size_t)65, (size_t)567, (size_t)1024}) {
        for (uint32_t D = 2; D <= MAX_DIGIT; ++D)
            for (size_t L = 0; L <= 140; ++L) {
                uint32_t st;
                pedersen_steps(n_gen, D, L, &st);
                const size_t used = std::min(L * 8, n_gen);
                REQUIRE((size_t)st * D >= used && (st == 0 || (size_t)(st - 1) * D < used));
            }
        for (uint32_t G = 1; G <= MAX_GROUP; ++G)
            for (size_t L = 0; L <= 230; ++L) {
                uint32_t g, st;
                bh_steps(n_gen, G, L, &g, &st);
                const size_t chunks = std::min((L * 8 + 2) / 3, n_gen);
                REQUIRE(G == 1 ? (g == 0 && st == chunks) : ((size_t)g * G + (st - g) == chunks && st - g < G));
            }
    }
    uint32_t g, st;
    bh_steps(567, 8, 64, &g, &st); REQUIRE(g == 21 && st == 24);   // a 63x9 tree node: 171 chunks = 21 groups + 3
    bh_steps(567, 8, 32, &g, &st); REQUIRE(g == 10 && st == 16);   // a 32-byte leaf: 86 chunks = 10 groups + 6
    pedersen_steps(1024, 24, 128, &st); REQUIRE(st == 43);
    // growth: never beyond the table, never below what is needed, doubling in between
    REQUIRE(grow_target(11, 0, 43) == 11 && grow_target(12, 11, 43) == 22 && grow_target(43, 22, 43) == 43 && grow_target(30, 22, 43) == 43);
    uint32_t built = 0, rebuilds = 0;
    for (uint32_t need = 1; need <= 43; ++need)
        if (need > built) { built = grow_target(need, built, 43); ++rebuilds; }
    REQUIRE(built == 43 && rebuilds <= 7);
    std::printf("OK\n");
    return 0;
}
