// gather_probe.hip -- the ceiling of the curve-hash table gather (round 4): every lane reads whole 128-byte lines at random
// line indices of a buffer of B bytes (8 x 16-byte loads per line, as te_fetch_all does), U lines in flight per lane, no
// arithmetic beyond an xor of the loaded words.  Reports lines/s and TB/s for B = 64 MB .. 64 GB and U = 1, 2, 4: the rate the
// memory system delivers RANDOM 128-byte lines at, from the L2 / Infinity Cache sizes up to tables that only HBM holds.
//   hipcc --offload-arch=gfx950 -O3 tools/gather_probe.hip -o tools/gather_probe && tools/gather_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t z) {
    z += 0x9e3779b97f4a7c15ull;
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
// PIECES: 16-byte loads per line (8 = the whole line; 1 = touch one piece of it)
template <int U, int PIECES>
__global__ void __launch_bounds__(256) gather_kernel(const uint4* __restrict__ buf, uint64_t n_lines, int rounds, uint32_t* __restrict__ sink) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t state = mix(tid);
    uint32_t acc = 0;
    for (int r = 0; r < rounds; ++r) {
        uint4 v[U][PIECES];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            state = mix(state);
            const uint4* line = buf + (state % n_lines) * 8;
#pragma unroll
            for (int p = 0; p < PIECES; ++p) v[k][p] = line[p];
        }
#pragma unroll
        for (int k = 0; k < U; ++k)
#pragma unroll
            for (int p = 0; p < PIECES; ++p) acc ^= v[k][p].x ^ v[k][p].y ^ v[k][p].z ^ v[k][p].w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void fill_kernel(uint4* buf, size_t n16) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) buf[i] = make_uint4((uint32_t)i, (uint32_t)(i >> 32), 0x9e3779b9u, (uint32_t)(i * 2654435761u));
}

// streaming read of known size: 16 B per lane, coalesced (the access pattern the guide's x2 FETCH_SIZE correction was calibrated on)
__global__ void __launch_bounds__(256) stream_kernel(const uint4* __restrict__ buf, size_t n16, uint32_t* __restrict__ sink) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        const uint4 v = buf[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
// `gather_probe calib [GB]`: launches of KNOWN traffic for a counter pass (rocprofv3 --pmc FETCH_SIZE): per window (64 MB, 1 GB,
// max) one launch each of the whole-line gather (8 pieces), the 7-piece gather te_fetch_all issues, and a one-piece gather; then a
// 4 GiB streaming read.  Prints one line per launch, in launch order, with the bytes of distinct 128-byte lines it touches.
static int calibrate(size_t max_gb) {
    uint32_t* sink;
    CK(hipMalloc(&sink, 4));
    void* buf;
    CK(hipMalloc(&buf, max_gb << 30));
    fill_kernel<<<4096, 256>>>((uint4*)buf, (max_gb << 30) / 16);
    CK(hipDeviceSynchronize());
    const int grid = 1024, rounds = 128;  // 4 waves per SIMD, U = 2: 2^26 lines = 8.59 GB of lines per launch
    const double lines = (double)grid * 256 * rounds * 2;
    for (size_t mb : {(size_t)64, (size_t)1024, max_gb << 10}) {
        const uint64_t n_lines = (mb << 20) / 128;
        gather_kernel<2, 8><<<grid, 256>>>((const uint4*)buf, n_lines, rounds, sink);
        gather_kernel<2, 7><<<grid, 256>>>((const uint4*)buf, n_lines, rounds, sink);
        gather_kernel<2, 1><<<grid, 256>>>((const uint4*)buf, n_lines, rounds, sink);
        CK(hipDeviceSynchronize());
        for (int pieces : {8, 7, 1})
            printf("CALIB gather_kernel<2,%d> window_MB %zu lines %.0f line_bytes %.0f requested_bytes %.0f\n", pieces, mb, lines, lines * 128, lines * 16 * pieces);
    }
    const size_t n16 = ((size_t)4 << 30) / 16;
    stream_kernel<<<4096, 256>>>((const uint4*)buf, n16, sink);
    CK(hipDeviceSynchronize());
    printf("CALIB stream_kernel window_MB 4096 lines %.0f line_bytes %.0f requested_bytes %.0f\n", (double)n16 / 8, (double)n16 * 16, (double)n16 * 16);
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 1 && !strcmp(argv[1], "calib")) return calibrate(argc > 2 ? (size_t)atol(argv[2]) : 64);
    const size_t max_gb = argc > 1 ? (size_t)atol(argv[1]) : 64;
    uint32_t* sink;
    CK(hipMalloc(&sink, 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    void* buf;
    CK(hipMalloc(&buf, max_gb << 30));
    fill_kernel<<<4096, 256>>>((uint4*)buf, (max_gb << 30) / 16);
    CK(hipDeviceSynchronize());
    const int waves_per_simd[] = {2, 3, 4, 8};
    printf("random 128-byte line gather, MI355X: lanes = 256 CUs x 4 SIMDs x W waves x 64; every lane reads `rounds` x U whole lines\n");
    for (size_t mb : {(size_t)64, (size_t)256, (size_t)1024, (size_t)4096, (size_t)16384, max_gb << 10}) {
        if (mb > (max_gb << 10)) continue;
        const uint64_t n_lines = (mb << 20) / 128;
        for (int W : waves_per_simd) {
            const int grid = 256 * 4 * W / 4;  // workgroups of 256 = 4 waves
            auto run = [&](const char* label, auto kern, int U, int pieces) {
                const int rounds = 256 / U;
                kern<<<grid, 256>>>((const uint4*)buf, n_lines, 4, sink);
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0));
                kern<<<grid, 256>>>((const uint4*)buf, n_lines, rounds, sink);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                const double lines = (double)grid * 256 * rounds * U;
                printf("table %6zu MB  waves/SIMD %d  %-22s %7.2f G lines/s  %6.2f TB/s of lines (%5.2f TB/s requested)\n", mb, W, label, lines / ms / 1e6,
                       lines * 128 / ms / 1e9, lines * 16 * pieces / ms / 1e9);
            };
            run("U=1 whole line", gather_kernel<1, 8>, 1, 8);
            run("U=2 whole line", gather_kernel<2, 8>, 2, 8);
            if (W <= 4) run("U=4 whole line", gather_kernel<4, 8>, 4, 8);
            run("U=4 one 16 B piece", gather_kernel<4, 1>, 4, 1);
        }
    }
    return 0;
}
