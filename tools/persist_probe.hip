// persist_probe.hip -- feasibility of a PERSISTENT curve-hash kernel fed by DMA-completion flags:
//   (a) do host-to-device copies of pinned memory make progress while a kernel whose workgroups SPIN occupies every CU?  (They do
//       if the runtime moves them with the SDMA engines; they deadlock if it uses blit kernels that need CU slots.)
//   (b) does hipStreamWriteValue32 on the copy stream reach a flag the spinning workgroups poll (host-pinned, uncached)?
//   (c) does hipStreamWaitValue32 on signal memory see a counter that workgroups increment -- i.e. can a later stream operation
//       (finalize kernel, copy-out) be released by PART of a running kernel?
// Every spin has an iteration limit: the probe cannot hang the device.   hipcc --offload-arch=gfx950 -O2 tools/persist_probe.hip -o tools/persist_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// workgroup b belongs to chunk b / wg_per_chunk; thread 0 polls flags[chunk] until it holds `expect` (or `limit` polls went by),
// then the workgroup reads its slice of the chunk's data (so that stale cache lines would show), bumps the chunk's counter
__global__ void __launch_bounds__(256) gated_kernel(const volatile uint32_t* flags, uint32_t expect, uint32_t wg_per_chunk, uint32_t limit,
                                                    const uint32_t* __restrict__ data, size_t words_per_wg, uint32_t want_word,
                                                    uint32_t* const* counters, uint64_t* t_release, uint32_t* bad, uint32_t* timed_out) {
    const uint32_t chunk = blockIdx.x / wg_per_chunk;
    extern __shared__ uint32_t occupancy_pad[];  // dynamic LDS only limits how many workgroups a CU holds (as the real kernel's registers do)
    __shared__ uint32_t ok;
    if (threadIdx.x == 0) {
        uint32_t it = 0;
        while (__hip_atomic_load((const uint32_t*)&flags[chunk], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != expect && ++it < limit) __builtin_amdgcn_s_sleep(32);
        ok = it < limit;
        t_release[blockIdx.x] = wall_clock64();  // s_memtime: shader clock
        if (it >= limit) atomicAdd(timed_out, 1u);
    }
    __syncthreads();
    if (!ok) return;
    const uint32_t* mine = data + (size_t)blockIdx.x * words_per_wg;
    uint32_t wrong = 0;
    for (size_t i = threadIdx.x; i < words_per_wg; i += 256) wrong += mine[i] != want_word + chunk;
    if (wrong) atomicAdd(bad, wrong);
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence_system();
        __hip_atomic_fetch_add(counters[chunk], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);  // an 8-byte signal per chunk: its low word
    }
}
__global__ void stamp_kernel(uint64_t* out, int k) { out[k] = wall_clock64(); }

int main(int argc, char** argv) {
    int can_wait = 0;
    CK(hipDeviceGetAttribute(&can_wait, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can_wait);
    const int chunks = 8, grid = 4096, wg_per_chunk = grid / chunks;
    const size_t chunk_bytes = (size_t)16 << 20, words_per_wg = chunk_bytes / 4 / wg_per_chunk;
    uint32_t *h_flags, *d_flags_alias, *h_src, *d_data, *d_bad, *d_to;
    uint64_t *d_trel, *d_stamp;
    // the flags the workgroups poll.  argv[1] = "host": pinned host memory (every poll is a PCIe read -- round one of this probe: thousands
    // of pollers starve the very copies they wait for); default: FINE-GRAINED DEVICE memory (polls stay on the device)
    const bool host_flags = argc > 1 && !strcmp(argv[1], "host");
    const size_t lds_pad = argc > 2 ? (size_t)atol(argv[2]) : 40960;  // bytes of dynamic LDS per workgroup: 40 KB = 4 workgroups per CU
    if (host_flags) {
        CK(hipHostMalloc(&h_flags, 64 * sizeof(uint32_t), hipHostMallocMapped));
        CK(hipHostGetDevicePointer((void**)&d_flags_alias, h_flags, 0));
    } else {
        CK(hipExtMallocWithFlags((void**)&d_flags_alias, 64 * sizeof(uint32_t), hipDeviceMallocFinegrained));
        CK(hipMemset(d_flags_alias, 0, 64 * sizeof(uint32_t)));
        h_flags = nullptr;
    }
    printf("flags in %s memory, %zu bytes of LDS per workgroup\n", host_flags ? "pinned HOST" : "fine-grained DEVICE", lds_pad);
    CK(hipHostMalloc(&h_src, chunks * chunk_bytes, hipHostMallocDefault));
    CK(hipMalloc(&d_data, chunks * chunk_bytes));
    CK(hipMalloc(&d_bad, 4)); CK(hipMalloc(&d_to, 4));
    CK(hipMalloc(&d_trel, grid * 8)); CK(hipMalloc(&d_stamp, 64 * 8));
    void* sig = nullptr;
    hipError_t es = hipExtMallocWithFlags(&sig, 8 * chunks, hipMallocSignalMemory);
    printf("hipExtMallocWithFlags(hipMallocSignalMemory, %d bytes): %s\n", 8 * chunks, hipGetErrorString(es));
    std::vector<void*> sigs(chunks, nullptr);
    if (es != hipSuccess) {  // one 8-byte signal per allocation?
        (void)hipGetLastError();
        bool all = true;
        for (int k = 0; k < chunks; ++k) all = all && hipExtMallocWithFlags(&sigs[k], 8, hipMallocSignalMemory) == hipSuccess;
        printf("  one 8-byte signal per chunk: %s\n", all ? "ok" : "FAILED");
        if (!all) can_wait = 0;
    } else {
        for (int k = 0; k < chunks; ++k) sigs[k] = (char*)sig + 8 * k;
    }
    hipStream_t comp, cin, fin;
    CK(hipStreamCreateWithFlags(&comp, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&cin, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&fin, hipStreamNonBlocking));
    // the kernel gets the addresses of the chunks' counters: the signal memory when there is one (hipStreamWaitValue32 accepts nothing
    // else), plain device words otherwise
    uint32_t* plain = nullptr;
    if (!can_wait) { CK(hipMalloc(&plain, 8 * chunks)); for (int k = 0; k < chunks; ++k) sigs[k] = (char*)plain + 8 * k; }
    uint32_t** d_sig_ptrs;
    CK(hipMalloc(&d_sig_ptrs, chunks * sizeof(void*)));
    CK(hipMemcpy(d_sig_ptrs, sigs.data(), chunks * sizeof(void*), hipMemcpyHostToDevice));
    for (int round = 0; round < 3; ++round) {
        const uint32_t expect = 100 + round, want = 0x5a5a0000u + 16 * round;
        for (int k = 0; k < chunks; ++k)
            for (size_t i = 0; i < chunk_bytes / 4; ++i) h_src[k * (chunk_bytes / 4) + i] = want + k;
        for (int k = 0; k < chunks; ++k) CK(hipMemsetAsync(sigs[k], 0, 8, comp));
        CK(hipMemsetAsync(d_bad, 0, 4, comp)); CK(hipMemsetAsync(d_to, 0, 4, comp));
        CK(hipMemsetAsync(d_stamp, 0, 64 * 8, comp));
        CK(hipStreamSynchronize(comp));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, comp));
        stamp_kernel<<<1, 1, 0, comp>>>(d_stamp, 0);
        // ~0.3 s of polling at most: 2^20 polls x (s_sleep 32 = 2048 cycles)
        gated_kernel<<<grid, 256, lds_pad, comp>>>(d_flags_alias, expect, wg_per_chunk, 1u << 17, d_data, words_per_wg, want, d_sig_ptrs, d_trel, d_bad, d_to);
        CK(hipEventRecord(e1, comp));
        for (int k = 0; k < chunks; ++k) {
            CK(hipMemcpyAsync((char*)d_data + k * chunk_bytes, (char*)h_src + k * chunk_bytes, chunk_bytes, hipMemcpyHostToDevice, cin));
            hipError_t ew = hipStreamWriteValue32(cin, d_flags_alias + k, expect, 0);
            if (ew != hipSuccess) { printf("hipStreamWriteValue32: %s\n", hipGetErrorString(ew)); (void)hipGetLastError(); }
            if (can_wait) {
                hipError_t e = hipStreamWaitValue32(fin, sigs[k], (uint32_t)wg_per_chunk, hipStreamWaitValueGte, 0xffffffffu);
                if (e != hipSuccess) { printf("hipStreamWaitValue32: %s\n", hipGetErrorString(e)); (void)hipGetLastError(); can_wait = 0; }
                else stamp_kernel<<<1, 1, 0, fin>>>(d_stamp, 1 + k);
            }
        }
        CK(hipStreamSynchronize(cin));
        CK(hipStreamSynchronize(comp));
        if (can_wait) CK(hipStreamSynchronize(fin));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        uint32_t bad = 0, to = 0;
        std::vector<uint64_t> trel(grid), stamp(64);
        CK(hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&to, d_to, 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(trel.data(), d_trel, grid * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(stamp.data(), d_stamp, 64 * 8, hipMemcpyDeviceToHost));
        printf("round %d: gated kernel %.3f ms (8 x 16 MB copies = %.1f GB/s), workgroups timed out %u, stale / wrong words %u\n", round, ms,
               chunks * chunk_bytes / ms / 1e6, to, bad);
        for (int k = 0; k < chunks; ++k) {
            uint64_t lo = ~0ull, hi = 0;
            for (int b = k * wg_per_chunk; b < (k + 1) * wg_per_chunk; ++b) { lo = trel[b] < lo ? trel[b] : lo; hi = trel[b] > hi ? trel[b] : hi; }
            printf("  chunk %d: workgroups released %.3f .. %.3f ms after the kernel's stream started", k, (double)(lo - stamp[0]) / 100e3, (double)(hi - stamp[0]) / 100e3);
            if (can_wait) printf(";  stream behind hipStreamWaitValue32 ran at %.3f ms", (double)(stamp[1 + k] - stamp[0]) / 100e3);
            printf("\n");
        }
    }
    printf("(times from wall_clock64: the 100 MHz constant clock)\n");
    return 0;
}
