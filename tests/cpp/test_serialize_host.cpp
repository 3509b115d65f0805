// Host-only check of akp.hpp's serialize namespace (no GPU): reads byte strings written by the ORACLE's serialiser
// (oracle/serialize.py, through tests/test_cpp_header.py), parses each with the C++ wrappers, writes it again and requires the
// same bytes; a few structural facts are printed for the python side to compare.
//   case file: repeated { u32 kind, u32 compress, u32 a, u32 b, u64 len, bytes }   kind 0 Path (fe 1), 1 MultiPath (fe 1),
//   2 Parameters (a = window_size, b = num_windows), 3 PoseidonConfig, 4 truncated Path (must throw code 1)
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iterator>

#include "../../include/akp.hpp"

using namespace akp;
#define REQUIRE(c) do { if (!(c)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    std::ifstream f(argv[1], std::ios::binary);
    std::vector<uint8_t> all((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    size_t at = 0, cases = 0;
    while (at < all.size()) {
        uint32_t hdr[4];
        uint64_t len;
        std::memcpy(hdr, all.data() + at, 16);
        std::memcpy(&len, all.data() + at + 16, 8);
        at += 24;
        const std::vector<uint8_t> bytes(all.begin() + at, all.begin() + at + len);
        at += len;
        const bool compress = hdr[1] != 0;
        switch (hdr[0]) {
        case 0: {
            const auto p = serialize::read_path(bytes, compress);
            REQUIRE(serialize::path(p, compress) == bytes);
            std::printf("path depth %zu index %zu\n", p.auth_path.size(), p.leaf_index);
            break;
        }
        case 1: {
            const auto m = serialize::read_multi_path(bytes, compress);
            REQUIRE(serialize::multi_path(m, compress) == bytes);
            std::printf("multipath m %zu suffix digests %zu\n", m.leaf_indexes.size(), m.suffixes.size());
            break;
        }
        case 2: {
            const auto g = serialize::read_te_parameters(bytes, compress);
            REQUIRE(g.window_size == hdr[2] && g.num_windows == hdr[3]);
            REQUIRE(serialize::te_parameters(g.generators_affine, g.window_size, g.num_windows, compress) == bytes);
            // the other mode and back
            const auto other = serialize::te_parameters(g.generators_affine, g.window_size, g.num_windows, !compress);
            REQUIRE(serialize::read_te_parameters(other, !compress).generators_affine == g.generators_affine);
            std::printf("parameters %u x %u\n", g.window_size, g.num_windows);
            break;
        }
        case 3: {
            const PoseidonConfig c = PoseidonConfig::deserialize(nullptr, bytes);
            REQUIRE(serialize::poseidon_config(c) == bytes);
            std::printf("config rounds %u+%u alpha %llu rate %u capacity %u\n", c.full_rounds, c.partial_rounds, (unsigned long long)c.alpha, c.rate, c.capacity);
            break;
        }
        case 4: {
            try { (void)serialize::read_path(bytes, compress); REQUIRE(false); } catch (const Error& e) { REQUIRE(e.code == AKP_ERR_BAD_LENGTH); }
            break;
        }
        default: return 3;
        }
        ++cases;
    }
    std::printf("OK %zu cases\n", cases);
    return 0;
}
