// Loads tools/bank_microbench.hsaco (gen_bank_microbench.py) and prints cycles per wave instruction of every pattern at
// 1, 2 and 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double clk = prop.clockRate * 1e3;
    hipModule_t mod;
    const char* path = getenv("AKP_BANK_HSACO") ? getenv("AKP_BANK_HSACO") : "tools/bank_microbench.hsaco";
    CK(hipModuleLoad(&mod, path));
    std::ifstream names("tools/bank_microbench.names");
    void* buf;
    CK(hipMalloc(&buf, 1 << 20));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int iters = 20000;
    printf("%d CUs, %.0f MHz; 128 instructions per iteration, %d iterations\n%-28s %10s %10s %10s\n", cus, clk / 1e6, iters, "pattern", "1 w/SIMD", "2 w/SIMD", "4 w/SIMD");
    std::string nm;
    while (std::getline(names, nm)) {
        if (nm.empty()) continue;
        hipFunction_t fn;
        CK(hipModuleGetFunction(&fn, mod, nm.c_str()));
        printf("%-28s", nm.c_str());
        for (int wps : {1, 2, 4}) {
            struct { void* p; int iters; } args{buf, iters};
            size_t asz = 12;
            void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &asz, HIP_LAUNCH_PARAM_END};
            float best = 1e30f;
            for (int rep = 0; rep < 4; ++rep) {
                CK(hipEventRecord(a));
                CK(hipModuleLaunchKernel(fn, cus * wps, 1, 1, 256, 1, 1, 0, 0, nullptr, cfg));
                CK(hipEventRecord(b));
                CK(hipEventSynchronize(b));
                float ms;
                CK(hipEventElapsedTime(&ms, a, b));
                if (rep && ms < best) best = ms;
            }
            printf(" %10.2f", best * 1e-3 * clk / ((double)iters * 128 * wps));
        }
        printf("\n");
    }
    return 0;
}
