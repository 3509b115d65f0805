// vmm2_probe.hip -- round 6: which sub-range mappings of the HIP virtual-memory API are valid on this stack (reserve one range, create + map +
// set access piece by piece).  Result (profiles/r06_s2/README.md): most shapes work, a 1 GiB range with pieces of 16 + 128 MiB fails in
// hipMemSetAccess ("invalid argument") -- the API is not used by the library.   hipcc --offload-arch=gfx950 -O2 tools/vmm2_probe.hip -o tools/vmm2_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
static const char* E(hipError_t e) { return hipGetErrorString(e); }
int trial(size_t va, size_t align, std::vector<size_t> pieces) {
    void* base = nullptr;
    hipError_t e = hipMemAddressReserve(&base, va, align, nullptr, 0);
    printf("reserve %zu MiB align %zu MiB: %s base=%p\n", va >> 20, align >> 20, E(e), base);
    if (e != hipSuccess) { (void)hipGetLastError(); return 1; }
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    size_t at = 0;
    std::vector<hipMemGenericAllocationHandle_t> hs;
    for (size_t sz : pieces) {
        hipMemGenericAllocationHandle_t h;
        e = hipMemCreate(&h, sz, &prop, 0);
        printf("  create %zu MiB: %s;", sz >> 20, E(e));
        if (e == hipSuccess) { e = hipMemMap((char*)base + at, sz, 0, h, 0); printf(" map at +%zu MiB: %s;", at >> 20, E(e)); }
        if (e == hipSuccess) { e = hipMemSetAccess((char*)base + at, sz, &acc, 1); printf(" access: %s;", E(e)); }
        if (e == hipSuccess) { e = hipMemset((char*)base + at, 1, sz); printf(" memset: %s", E(e)); hipDeviceSynchronize(); }
        printf("\n");
        (void)hipGetLastError();
        if (e != hipSuccess) break;
        hs.push_back(h);
        at += sz;
    }
    if (at) printf("  unmap: %s\n", E(hipMemUnmap(base, at)));
    for (auto h : hs) hipMemRelease(h);
    printf("  free: %s\n", E(hipMemAddressFree(base, va)));
    return 0;
}
int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipSetDevice(0);
    const size_t M = 1 << 20;
    trial(144 * M, 2 * M, {16 * M, 128 * M});
    trial(144 * M, 2 * M, {144 * M});
    trial(144 * M, 2 * M, {16 * M, 16 * M, 112 * M});
    trial(144 * M, 1024 * M, {16 * M, 128 * M});
    trial(256 * M, 2 * M, {16 * M, 240 * M});
    trial(1024 * M, 1024 * M, {16 * M, 128 * M, 880 * M});
    trial(146 * M, 2 * M, {18 * M, 128 * M});
    return 0;
}
