// pcie_probe.hip -- how fast can pinned host memory be brought into HBM, and by whom?  (round 4, the pinned-input path of the
// curve hashes.)  hipMemcpyAsync (copy engine) against a plain shader copy kernel that reads the device alias of the pinned
// buffer with 16-byte loads, for several grid sizes and loads in flight per lane; then the same while another stream writes
// digests back to pinned memory from a kernel (the zero-copy output of akp_te_crh_batch).
//   hipcc --offload-arch=gfx950 -O3 tools/pcie_probe.hip -o tools/pcie_probe && tools/pcie_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int U>
__global__ void copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n16; i += U * stride) {
        uint4 v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) v[k] = src[i + k * stride];
#pragma unroll
        for (int k = 0; k < U; ++k) dst[i + k * stride] = v[k];
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}


int main() {
    const size_t big = (size_t)128 << 20;
    void *h_in, *h_out, *h_page, *d_a, *d_b;
    CK(hipHostMalloc(&h_in, big, hipHostMallocDefault));
    CK(hipHostMalloc(&h_out, big, hipHostMallocDefault));
    h_page = malloc(big);
    memset(h_in, 1, big); memset(h_out, 2, big); memset(h_page, 3, big);
    CK(hipMalloc(&d_a, big)); CK(hipMalloc(&d_b, big));
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* label, size_t bytes, int reps, auto fn) {
        fn(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, s1));
        for (int r = 0; r < reps; ++r) fn();
        CK(hipEventRecord(e1, s1));
        CK(hipEventSynchronize(e1)); CK(hipDeviceSynchronize());
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-72s %8.3f ms  %6.1f GB/s\n", label, ms / reps, bytes / (ms / reps) / 1e6);
    };
    char lab[160];
    for (size_t mb : {1, 16, 128}) {
        const size_t b = mb << 20;
        snprintf(lab, sizeof lab, "hipMemcpyAsync H2D pinned   %4zu MiB", mb);
        run(lab, b, mb == 128 ? 5 : 20, [&] { CK(hipMemcpyAsync(d_a, h_in, b, hipMemcpyHostToDevice, s1)); });
        snprintf(lab, sizeof lab, "hipMemcpyAsync D2H pinned   %4zu MiB", mb);
        run(lab, b, mb == 128 ? 5 : 20, [&] { CK(hipMemcpyAsync(h_out, d_a, b, hipMemcpyDeviceToHost, s1)); });
    }
    run("hipMemcpyAsync H2D pageable  128 MiB", big, 5, [&] { CK(hipMemcpyAsync(d_a, h_page, big, hipMemcpyHostToDevice, s1)); });
    run("hipMemcpyAsync H2D pinned 8 x 16 MiB back to back", big, 5, [&] {
        for (int c = 0; c < 8; ++c) CK(hipMemcpyAsync((char*)d_a + ((size_t)c << 24), (char*)h_in + ((size_t)c << 24), (size_t)16 << 20, hipMemcpyHostToDevice, s1));
    });
    const size_t n16 = big / 16;
    for (int grid : {32, 64, 128, 256, 512, 1024, 4096}) {
        snprintf(lab, sizeof lab, "shader copy pinned -> HBM, 128 MiB, grid %4d x 256, 1 load in flight", grid);
        run(lab, big, 5, [&] { hipLaunchKernelGGL(copy_kernel<1>, dim3(grid), dim3(256), 0, s1, (const uint4*)h_in, (uint4*)d_a, n16); });
        snprintf(lab, sizeof lab, "shader copy pinned -> HBM, 128 MiB, grid %4d x 256, 4 loads in flight", grid);
        run(lab, big, 5, [&] { hipLaunchKernelGGL(copy_kernel<4>, dim3(grid), dim3(256), 0, s1, (const uint4*)h_in, (uint4*)d_a, n16); });
        snprintf(lab, sizeof lab, "shader copy pinned -> HBM, 128 MiB, grid %4d x 256, 8 loads in flight", grid);
        run(lab, big, 5, [&] { hipLaunchKernelGGL(copy_kernel<8>, dim3(grid), dim3(256), 0, s1, (const uint4*)h_in, (uint4*)d_a, n16); });
    }
    for (int grid : {64, 256, 1024}) {
        snprintf(lab, sizeof lab, "shader copy HBM -> pinned, 128 MiB, grid %4d x 256, 4 in flight", grid);
        run(lab, big, 5, [&] { hipLaunchKernelGGL(copy_kernel<4>, dim3(grid), dim3(256), 0, s1, (const uint4*)d_a, (uint4*)h_out, n16); });
    }
    // both directions at once: reads of pinned input on s1 (timed), writes to pinned output on s2 (half the bytes, as the hashes do)
    for (int grid : {64, 256}) {
        snprintf(lab, sizeof lab, "shader copy in (grid %d, 8 in flight) WHILE shader copy out 64 MiB on another stream", grid);
        run(lab, big, 5, [&] {
            hipLaunchKernelGGL(copy_kernel<4>, dim3(64), dim3(256), 0, s2, (const uint4*)d_b, (uint4*)h_out, n16 / 2);
            hipLaunchKernelGGL(copy_kernel<8>, dim3(grid), dim3(256), 0, s1, (const uint4*)h_in, (uint4*)d_a, n16);
        });
    }
    run("hipMemcpyAsync H2D pinned 128 MiB WHILE shader copy out 64 MiB on another stream", big, 5, [&] {
        hipLaunchKernelGGL(copy_kernel<4>, dim3(64), dim3(256), 0, s2, (const uint4*)d_b, (uint4*)h_out, n16 / 2);
        CK(hipMemcpyAsync(d_a, h_in, big, hipMemcpyHostToDevice, s1));
    });
    run("hipMemcpyAsync H2D pinned 128 MiB WHILE hipMemcpyAsync D2H pinned 64 MiB on another stream", big, 5, [&] {
        CK(hipMemcpyAsync(h_out, d_b, big / 2, hipMemcpyDeviceToHost, s2));
        CK(hipMemcpyAsync(d_a, h_in, big, hipMemcpyHostToDevice, s1));
    });
    return 0;
}
