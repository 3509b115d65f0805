// vram_probe.hip -- round 6: where does the first use of never-touched VRAM cost its time?  (The driver's fresh-box bench of round 5
// saw 1.3-1.6 s for the first 46 / 75 GB curve table where every builder session -- which had run pytest on the same lease first -- saw
// 50-65 ms.)  For each of `rounds` rounds: hipMalloc `slices` x `gb` GB one after the other, time each hipMalloc, the FIRST write pass
// over it (a store kernel), a SECOND write pass, then hipFree everything.  Round 1 on a fresh box touches VRAM nobody has used since
// boot; round 2 gets the blocks round 1 released.  Then the same through the virtual-memory API (hipMemAddressReserve + hipMemCreate +
// hipMemMap per slice into ONE address range): is it usable here, what does a slice cost, and how fast are random 128-byte gathers
// from the mapped range against a hipMalloc range of the same size.
//   hipcc --offload-arch=gfx950 -O2 tools/vram_probe.hip -o tools/vram_probe;  tools/vram_probe [gb_per_slice=16] [slices=8] [rounds=2]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);                 \
            return 1;                                                                          \
        }                                                                                      \
    } while (0)

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void fill_kernel(uint4* p, size_t n16, uint32_t v) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n16; i += stride) p[i] = make_uint4(v, v + 1, v + 2, (uint32_t)i);
}
// every lane gathers `steps` random 128-byte lines (8 x uint4) from a window of `lines` lines
__global__ void gather_kernel(const uint4* p, size_t lines, int steps, uint32_t* sink) {
    uint64_t s = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
    uint32_t acc = 0;
    for (int k = 0; k < steps; ++k) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        const size_t line = (size_t)((s >> 16) % lines);
        const uint4* q = p + line * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += q[j].x ^ q[j].w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
// a long kernel of few workgroups: `wgs` workgroups spin for `cycles` of the constant 100 MHz clock
__global__ void spin_kernel(unsigned long long ticks, uint32_t* sink) {
    const unsigned long long t0 = __builtin_readcyclecounter();
    unsigned long long now_ = t0;
    uint32_t acc = threadIdx.x;
    while (now_ - t0 < ticks) {
        for (int i = 0; i < 256; ++i) acc = acc * 1664525u + 1013904223u;
        now_ = __builtin_readcyclecounter();
    }
    if (acc == 0x12345u) sink[0] = acc;
}
static int time_fill(void* p, size_t bytes, double* ms) {
    const double t0 = now();
    hipLaunchKernelGGL(fill_kernel, dim3(256 * 16), dim3(256), 0, 0, (uint4*)p, bytes / 16, 7u);
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    *ms = (now() - t0) * 1e3;
    return 0;
}
static int time_gather(const void* p, size_t bytes, double* glines_s) {
    uint32_t* sink = nullptr;
    CK(hipMalloc(&sink, 4));
    const int steps = 64;
    const size_t lanes = (size_t)1 << 20;
    for (int rep = 0; rep < 2; ++rep) {
        const double t0 = now();
        hipLaunchKernelGGL(gather_kernel, dim3((unsigned)(lanes / 256)), dim3(256), 0, 0, (const uint4*)p, bytes / 128, steps, sink);
        CK(hipGetLastError());
        CK(hipDeviceSynchronize());
        *glines_s = lanes * steps / (now() - t0) / 1e9;
    }
    CK(hipFree(sink));
    return 0;
}

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);  // a GPU fault aborts the process: nothing may sit in a buffer
    const size_t gb = argc > 1 ? strtoull(argv[1], nullptr, 10) : 16;
    const int slices = argc > 2 ? atoi(argv[2]) : 8;
    const int rounds = argc > 3 ? atoi(argv[3]) : 2;
    const char* mode = argc > 4 ? argv[4] : "malloc,bg,vmm";  // which parts run (a crash in one must not take the others' output)
    auto on = [&](const char* m) { return strstr(mode, m) != nullptr; };
    const size_t bytes = gb << 30;
    CK(hipSetDevice(0));
    size_t free_b = 0, total_b = 0;
    CK(hipMemGetInfo(&free_b, &total_b));
    printf("device memory: %.1f GB free of %.1f GB\n", free_b / 1e9, total_b / 1e9);
    {  // warm the runtime (code object load, first launch) on a small buffer
        void* w = nullptr;
        double ms;
        CK(hipMalloc(&w, 1 << 20));
        if (time_fill(w, 1 << 20, &ms)) return 1;
        CK(hipFree(w));
    }
    // "single": ONE hipMalloc of `gb` GB (the shape of a curve table: 46 GB Pedersen, 22 / 75 GB Bowe-Hopwood), then a second of half the size;
    // alloc / first write / second write / free, `rounds` times in this process
    for (int r = 0; on("single") && r < rounds; ++r) {
        for (size_t sz : {bytes, bytes / 2}) {
            void* p = nullptr;
            double t0 = now();
            CK(hipMalloc(&p, sz));
            const double t_alloc = (now() - t0) * 1e3;
            double w1 = 0, w2 = 0;
            if (time_fill(p, sz, &w1) || time_fill(p, sz, &w2)) return 1;
            t0 = now();
            CK(hipFree(p));
            printf("round %d: ONE hipMalloc of %5.1f GB: %8.2f ms   first write pass %8.2f ms   second %7.2f ms   hipFree %7.2f ms\n", r + 1, sz / 1073741824.0, t_alloc, w1, w2,
                   (now() - t0) * 1e3);
        }
    }
    for (int r = 0; on("malloc") && r < rounds; ++r) {
        std::vector<void*> ptrs;
        printf("round %d: hipMalloc of %d x %zu GB\n", r + 1, slices, gb);
        for (int i = 0; i < slices; ++i) {
            void* p = nullptr;
            const double t0 = now();
            CK(hipMalloc(&p, bytes));
            const double t_alloc = (now() - t0) * 1e3;
            double w1 = 0, w2 = 0;
            if (time_fill(p, bytes, &w1) || time_fill(p, bytes, &w2)) return 1;
            printf("  slice %d: hipMalloc %8.2f ms (%.1f GB/s)   first write pass %7.2f ms   second %7.2f ms (%.0f GB/s)\n", i, t_alloc, bytes / t_alloc / 1e6, w1,
                   w2, bytes / w2 / 1e6);
            ptrs.push_back(p);
        }
        const double t0 = now();
        for (void* p : ptrs) CK(hipFree(p));
        printf("  hipFree of all: %.2f ms\n", (now() - t0) * 1e3);
    }
    // does a hipMalloc on another thread hold up launches of this one?  (a background table build beside foreground hashing)
    if (on("bg")) {
        void* w = nullptr;
        CK(hipMalloc(&w, (size_t)1 << 30));
        double alloc_ms = 0;
        void* big = nullptr;
        std::thread th([&] {
            (void)hipSetDevice(0);
            const double t0 = now();
            if (hipMalloc(&big, (size_t)slices * bytes / 2) != hipSuccess) big = nullptr;
            alloc_ms = (now() - t0) * 1e3;
        });
        double worst = 0, sum = 0;
        int cnt = 0;
        const double t_begin = now();
        while (now() - t_begin < 0.05 || (cnt < 2000 && alloc_ms == 0)) {
            double ms;
            if (time_fill(w, (size_t)1 << 26, &ms)) return 1;
            worst = ms > worst ? ms : worst;
            sum += ms;
            ++cnt;
        }
        th.join();
        printf("background hipMalloc of %zu GB: %.2f ms; %d foreground launches (64 MB fill + sync) meanwhile: mean %.3f ms, worst %.3f ms\n", slices * gb / 2, alloc_ms, cnt,
               sum / cnt, worst);
        if (big) CK(hipFree(big));
        CK(hipFree(w));
    }
    // "stall": the same question when the background hipMalloc really STALLS: free `slices` x `gb` GB (the driver wipes released VRAM by
    // DMA, ~35 GB/s; an allocation made while wipes are queued waits for them -- seen above as one multi-second hipMalloc), then allocate
    // them again on a second thread while this thread keeps launching
    if (on("stall")) {
        std::vector<void*> ptrs;
        for (int i = 0; i < slices; ++i) {
            void* p = nullptr;
            double ms;
            CK(hipMalloc(&p, bytes));
            if (time_fill(p, bytes, &ms)) return 1;
            ptrs.push_back(p);
        }
        void* w = nullptr;
        CK(hipMalloc(&w, (size_t)1 << 30));
        for (void* p : ptrs) CK(hipFree(p));
        volatile int done = 0;
        std::vector<double> alloc_ms(slices, 0.0);
        std::thread th([&] {
            (void)hipSetDevice(0);
            std::vector<void*> q;
            for (int i = 0; i < slices; ++i) {
                void* p = nullptr;
                const double t0 = now();
                if (hipMalloc(&p, bytes) != hipSuccess) break;
                alloc_ms[i] = (now() - t0) * 1e3;
                q.push_back(p);
            }
            for (void* p : q) (void)hipFree(p);
            done = 1;
        });
        double worst = 0, sum = 0, t_worst = 0;
        int cnt = 0;
        const double t_begin = now();
        while (!done) {
            double ms;
            if (time_fill(w, (size_t)1 << 26, &ms)) return 1;
            if (ms > worst) { worst = ms; t_worst = now() - t_begin; }
            sum += ms;
            ++cnt;
        }
        th.join();
        printf("stall: background hipMalloc x %d of %zu GB right after freeing them:", slices, gb);
        for (double a : alloc_ms) printf(" %.1f", a);
        printf(" ms\n       %d foreground launches (64 MB fill + sync) meanwhile, %.2f s: mean %.3f ms, worst %.3f ms (at %.2f s)\n", cnt, now() - t_begin, sum / cnt, worst, t_worst);
        CK(hipFree(w));
    }
    // "beside": does ONE long kernel of few workgroups on another stream hold up short launches of this thread?  (the background table
    // build as 128 workgroups walking the table.)  Stream kinds: lowest priority, default priority (non-blocking), CU mask of 32.
    if (on("beside")) {
        void* w = nullptr;
        uint32_t* sink = nullptr;
        CK(hipMalloc(&w, (size_t)1 << 30));
        CK(hipMalloc(&sink, 4));
        int lo_p = 0, hi_p = 0;
        CK(hipDeviceGetStreamPriorityRange(&lo_p, &hi_p));
        for (int kind = 0; kind < 3; ++kind) {
            hipStream_t bs = nullptr;
            if (kind == 0) CK(hipStreamCreateWithPriority(&bs, hipStreamNonBlocking, lo_p));
            if (kind == 1) CK(hipStreamCreateWithFlags(&bs, hipStreamNonBlocking));
            if (kind == 2) {
                uint32_t mask[8] = {0xffffffffu, 0, 0, 0, 0, 0, 0, 0};
                CK(hipExtStreamCreateWithCUMask(&bs, 8, mask));
            }
            for (int fg = 0; fg < 2; ++fg) {  // foreground on the NULL stream / on a non-blocking stream of its own
                hipStream_t fs = nullptr;
                if (fg) CK(hipStreamCreateWithFlags(&fs, hipStreamNonBlocking));
                double ms;
                for (int i = 0; i < 50; ++i) {
                    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, fs, (uint4*)w, ((size_t)1 << 26) / 16, 7u);
                    CK(hipStreamSynchronize(fs));
                }
                hipLaunchKernelGGL(spin_kernel, dim3(128), dim3(256), 0, bs, 30000000ull /* 0.3 s at 100 MHz */, sink);
                CK(hipGetLastError());
                const double t_begin = now();
                double worst = 0, sum = 0;
                int cnt = 0;
                while (hipStreamQuery(bs) == hipErrorNotReady) {
                    const double t0 = now();
                    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, fs, (uint4*)w, ((size_t)1 << 26) / 16, 7u);
                    CK(hipStreamSynchronize(fs));
                    ms = (now() - t0) * 1e3;
                    worst = ms > worst ? ms : worst;
                    sum += ms;
                    ++cnt;
                }
                (void)hipGetLastError();
                printf("beside: long kernel (128 workgroups, %.0f ms) on a %s stream, short launches on %s: %d launches, mean %.3f ms, worst %.3f ms\n", (now() - t_begin) * 1e3,
                       kind == 0 ? "lowest-priority" : kind == 1 ? "default-priority" : "CU-mask(32)", fg ? "a non-blocking stream" : "the NULL stream", cnt, cnt ? sum / cnt : 0.0, worst);
                if (fs) CK(hipStreamDestroy(fs));
            }
            CK(hipStreamDestroy(bs));
        }
        CK(hipFree(sink));
        CK(hipFree(w));
    }
    if (!on("vmm")) return 0;
    // ---- virtual-memory API: one address range, physical slices mapped one by one -------------------------------------------------
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    hipError_t e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
    if (e != hipSuccess) {
        printf("virtual-memory API: hipMemGetAllocationGranularity: %s -- not usable here\n", hipGetErrorString(e));
        return 0;
    }
    printf("virtual-memory API: recommended granularity %zu bytes\n", gran);
    void* base = nullptr;
    const size_t total = (size_t)slices * bytes;
    double t0 = now();
    CK(hipMemAddressReserve(&base, total, (size_t)1 << 30, nullptr, 0));
    printf("  hipMemAddressReserve of %zu GB: %.2f ms\n", total >> 30, (now() - t0) * 1e3);
    std::vector<hipMemGenericAllocationHandle_t> handles;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    for (int i = 0; i < slices; ++i) {
        hipMemGenericAllocationHandle_t h;
        t0 = now();
        CK(hipMemCreate(&h, bytes, &prop, 0));
        const double t_create = (now() - t0) * 1e3;
        t0 = now();
        CK(hipMemMap((char*)base + (size_t)i * bytes, bytes, 0, h, 0));
        CK(hipMemSetAccess((char*)base + (size_t)i * bytes, bytes, &acc, 1));
        const double t_map = (now() - t0) * 1e3;
        printf("  slice %d: hipMemCreate %8.2f ms   map + access %6.2f ms ...", i, t_create, t_map);
        double w1 = 0, w2 = 0;
        if (time_fill((char*)base + (size_t)i * bytes, bytes, &w1) || time_fill((char*)base + (size_t)i * bytes, bytes, &w2)) return 1;
        printf("   first write pass %7.2f ms   second %7.2f ms\n", w1, w2);
        handles.push_back(h);
    }
    double g_vmm = 0, g_malloc = 0;
    if (time_gather(base, total, &g_vmm)) return 1;
    printf("  random 128-byte gathers over the mapped %zu GB range: %.2f G lines/s\n", total >> 30, g_vmm);
    t0 = now();
    CK(hipMemUnmap(base, total));
    for (auto h : handles) CK(hipMemRelease(h));
    CK(hipMemAddressFree(base, total));
    printf("  unmap + release + address free: %.2f ms\n", (now() - t0) * 1e3);
    void* m = nullptr;
    CK(hipMalloc(&m, total));
    double w = 0;
    if (time_fill(m, total, &w)) return 1;
    if (time_gather(m, total, &g_malloc)) return 1;
    printf("  the same gathers over ONE hipMalloc of %zu GB: %.2f G lines/s\n", total >> 30, g_malloc);
    CK(hipFree(m));
    return 0;
}
