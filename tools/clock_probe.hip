// clock_probe.hip -- what counts shader cycles on gfx950?  One wave runs a chain of N dependent v_mad_u64_u32 and reads
// s_memtime (clock64) and s_memrealtime (wall_clock64) before and after; the host times the launch with events.
//   hipcc --offload-arch=gfx950 -O3 tools/clock_probe.hip -o tools/clock_probe && tools/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void chain(unsigned long long* out, unsigned n, unsigned seed) {
    unsigned long long acc = seed + threadIdx.x;
    unsigned a = seed | 1u;
    const unsigned long long t0 = clock64(), w0 = wall_clock64();
#pragma unroll 1
    for (unsigned i = 0; i < n; i += 16) {
#pragma unroll
        for (int k = 0; k < 16; ++k) acc = (unsigned long long)(unsigned)acc * a + acc;  // v_mad_u64_u32, dependent
    }
    const unsigned long long t1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = w1 - w0; out[2] = acc; }
}
int main() {
    unsigned long long *d, h[3];
    CK(hipMalloc(&d, 24));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 6; ++rep) {
        const unsigned n = rep < 3 ? (1u << 20) : (1u << 22);
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, 0, d, n, 12345u + rep);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(h, d, 24, hipMemcpyDeviceToHost));
        printf("n %u  event %.3f ms  d(clock64) %llu  d(wall_clock64) %llu  -> clock64/instr %.3f  clock64 MHz (vs events) %.1f  wall_clock64 MHz %.1f  clock64/wall_clock64 %.3f\n",
               n, ms, h[0], h[1], (double)h[0] / n, h[0] / (ms * 1e3), h[1] / (ms * 1e3), (double)h[0] / (double)h[1]);
    }
    int clk = 0; CK(hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0));
    int wclk = 0; CK(hipDeviceGetAttribute(&wclk, hipDeviceAttributeWallClockRate, 0));
    printf("hipDeviceAttributeClockRate %d kHz, WallClockRate %d kHz\n", clk, wclk);
    return 0;
}
