// microbench.hip -- measures the issue rate of the VALU instructions the Fr multiplier can be built
// from on gfx950, and the throughput of the multiplier variants themselves.  Evidence for DESIGN.md
// "Fr multiply".  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench.hip -o tools/microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../crypto_primitives_amd/csrc/fr.hpp"
#include "../crypto_primitives_amd/csrc/f29.hpp"
using namespace akp;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
#define REP8(X) X X X X X X X X
#define REP64(X) REP8(REP8(X))

// 8 independent chains x 8 = 64 instructions per loop iteration
#define KERNEL_U64(name, INSTR)                                                             \
__global__ void __launch_bounds__(256) name(u64* out, int iters, u32 x, u32 y) {              \
    u64 a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    u32 vx = x + threadIdx.x, vy = y ^ threadIdx.x;                                            \
    for (int i = 0; i < iters; ++i) {                                                        \
        REP8(asm volatile(INSTR(%0) "\n\t" INSTR(%1) "\n\t" INSTR(%2) "\n\t" INSTR(%3) "\n\t" INSTR(%4) "\n\t" INSTR(%5) "\n\t" INSTR(%6) "\n\t" INSTR(%7) \
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(vx), "v"(vy) : "vcc");) \
    }                                                                                        \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;       \
}
#define KERNEL_U32(name, INSTR)                                                             \
__global__ void __launch_bounds__(256) name(u64* out, int iters, u32 x, u32 y) {              \
    u32 a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    u32 vx = x + threadIdx.x, vy = y ^ threadIdx.x;                                            \
    for (int i = 0; i < iters; ++i) {                                                        \
        REP8(asm volatile(INSTR(%0) "\n\t" INSTR(%1) "\n\t" INSTR(%2) "\n\t" INSTR(%3) "\n\t" INSTR(%4) "\n\t" INSTR(%5) "\n\t" INSTR(%6) "\n\t" INSTR(%7) \
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(vx), "v"(vy) : "vcc");) \
    }                                                                                        \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;       \
}
#define I_MAD64(r) "v_mad_u64_u32 " #r ", vcc, %8, %9, " #r
#define I_MADI64(r) "v_mad_i64_i32 " #r ", vcc, %8, %9, " #r
#define I_LSHR64(r) "v_lshrrev_b64 " #r ", 29, " #r
#define I_AND32(r) "v_and_b32_e32 " #r ", %8, " #r
#define I_MAD64_SGPRDST(r) "v_mad_u64_u32 " #r ", s[10:11], %8, %9, " #r
#define I_LSHLADD64(r) "v_lshl_add_u64 " #r ", " #r ", 0, " #r
#define I_FMA64(r) "v_fma_f64 " #r ", " #r ", " #r ", " #r
#define I_MULLO(r) "v_mul_lo_u32 " #r ", " #r ", %8"
#define I_MULHI(r) "v_mul_hi_u32 " #r ", " #r ", %8"
#define I_ADD32(r) "v_add_u32_e32 " #r ", %8, " #r
#define I_ADDCO(r) "v_add_co_u32_e32 " #r ", vcc, %8, " #r
#define I_ADDC(r) "v_addc_co_u32_e32 " #r ", vcc, %8, " #r ", vcc"
#define I_MAD24(r) "v_mad_u32_u24 " #r ", %8, %9, " #r
#define I_FMA32(r) "v_fma_f32 " #r ", " #r ", " #r ", " #r
#define I_ADD3(r) "v_add3_u32 " #r ", %8, %9, " #r
#define I_CNDMASK(r) "v_cndmask_b32_e32 " #r ", %8, " #r ", vcc"
#define I_ALIGNBIT(r) "v_alignbit_b32 " #r ", " #r ", %8, 7"
#define I_PKFMA32(r) "v_pk_fma_f32 " #r ", " #r ", " #r ", " #r
#define I_MULF64(r) "v_mul_f64 " #r ", " #r ", " #r
#define I_ADDF64(r) "v_add_f64 " #r ", " #r ", " #r

KERNEL_U64(k_mad64, I_MAD64)
KERNEL_U64(k_lshladd64, I_LSHLADD64)
KERNEL_U64(k_madi64, I_MADI64)
KERNEL_U64(k_lshr64, I_LSHR64)
KERNEL_U32(k_and32, I_AND32)
KERNEL_U64(k_fma64, I_FMA64)
KERNEL_U64(k_mulf64, I_MULF64)
KERNEL_U64(k_addf64, I_ADDF64)
KERNEL_U64(k_pkfma32, I_PKFMA32)
KERNEL_U32(k_mullo, I_MULLO)
KERNEL_U32(k_mulhi, I_MULHI)
KERNEL_U32(k_add32, I_ADD32)
KERNEL_U32(k_addco, I_ADDCO)
KERNEL_U32(k_addc, I_ADDC)
KERNEL_U32(k_mad24, I_MAD24)
KERNEL_U32(k_fma32, I_FMA32)
KERNEL_U32(k_add3, I_ADD3)
KERNEL_U32(k_cndmask, I_CNDMASK)
KERNEL_U32(k_alignbit, I_ALIGNBIT)

// one dependent chain: latency of v_mad_u64_u32 -> v_mad_u64_u32
__global__ void __launch_bounds__(256) k_mad64_dep(u64* out, int iters, u32 x, u32 y) {
    u64 a0 = threadIdx.x;
    u32 vx = x + threadIdx.x, vy = y ^ threadIdx.x;
    for (int i = 0; i < iters; ++i) {
        REP64(asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a0) : "v"(vx), "v"(vy) : "vcc");)
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0;
}
// mad + addc pair as used by the multiplier (dependent through vcc, 4 independent accumulators)
__global__ void __launch_bounds__(256) k_mac_pair(u64* out, int iters, u32 x, u32 y) {
    u64 a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3;
    u32 c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    u32 vx = x + threadIdx.x, vy = y ^ threadIdx.x;
    for (int i = 0; i < iters; ++i) {
        REP8(asm volatile(
            "v_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_addc_co_u32_e32 %4, vcc, 0, %4, vcc\n\t"
            "v_mad_u64_u32 %1, vcc, %8, %9, %1\n\tv_addc_co_u32_e32 %5, vcc, 0, %5, vcc\n\t"
            "v_mad_u64_u32 %2, vcc, %8, %9, %2\n\tv_addc_co_u32_e32 %6, vcc, 0, %6, vcc\n\t"
            "v_mad_u64_u32 %3, vcc, %8, %9, %3\n\tv_addc_co_u32_e32 %7, vcc, 0, %7, vcc"
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(vx), "v"(vy) : "vcc");)
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ c0 ^ c1 ^ c2 ^ c3;
}

// multiplier variants: chain of dependent Montgomery products
__global__ void __launch_bounds__(256) k_frmul_asm(Fr* x, int iters) {
    const size_t i = blockIdx.x * blockDim.x + threadIdx.x;
    Fr a = x[i], b = x[i ^ 1];
    for (int k = 0; k < iters; ++k) a = fr_mul(a, b);
    x[i] = a;
}
__global__ void __launch_bounds__(256) k_frmul_portable(Fr* x, int iters) {
    const size_t i = blockIdx.x * blockDim.x + threadIdx.x;
    Fr a = x[i], b = x[i ^ 1];
    for (int k = 0; k < iters; ++k) a = fr_mul_portable(a, b);
    x[i] = a;
}
__global__ void __launch_bounds__(256) k_fradd(Fr* x, int iters) {
    const size_t i = blockIdx.x * blockDim.x + threadIdx.x;
    Fr a = x[i], b = x[i ^ 1];
    for (int k = 0; k < iters; ++k) { a = fr_add(a, b); b = fr_sub(b, a); }
    x[i] = a;
}

// radix-2^29 core (f29.hpp)
template <bool S>
__global__ void __launch_bounds__(256) k_f29_mul(Fr* x, int iters) {
    const size_t i = blockIdx.x * blockDim.x + threadIdx.x;
    F29T<S> a = f29_from_wire<S>(x[i]), b = f29_from_wire<S>(x[i ^ 1]);
    for (int k = 0; k < iters; ++k) a = f29_mul(a, b);
    x[i] = f29_to_wire(a);
}
__global__ void __launch_bounds__(256) k_f29_sqr(Fr* x, int iters) {
    const size_t i = blockIdx.x * blockDim.x + threadIdx.x;
    FU a = f29_from_wire<false>(x[i]);
    for (int k = 0; k < iters; ++k) a = f29_sqr(a);
    x[i] = f29_to_wire(a);
}
__global__ void __launch_bounds__(256) k_f29_dot3(Fr* x, int iters) {
    const size_t i = blockIdx.x * blockDim.x + threadIdx.x;
    FU a = f29_from_wire<false>(x[i]), b = f29_from_wire<false>(x[i ^ 1]), c = f29_from_wire<false>(x[i ^ 2]);
    const FU m0 = f29_one<false>(), m1 = f29_k_in<false>(), m2 = f29_k_out<false>();
    for (int k = 0; k < iters; ++k) { FU r = f29_dot3(a, m0, b, m1, c, m2); c = b; b = a; a = r; }
    x[i] = f29_to_wire(a);
}

// experiment: empty-asm barriers after every partial product keep ONE accumulator chain (hipcc otherwise splits
// the chain for ILP and pays 15 extra 64-bit adds per product); the price is one s_nop per barrier.
template <int MODE>
__device__ __forceinline__ FU mulv(const FU& a, const FU& b) {
    typedef u32 L; typedef u64 W;
    W acc = 0; u32 m[9]; FU t;
#define BAR_P() if (MODE == 2) asm("" : "+v"(acc));
#pragma unroll
    for (int k = 0; k < 9; ++k) {
#pragma unroll
        for (int i = 0; i <= k; ++i) { acc += (W)a.l[i] * (W)b.l[k - i]; BAR_P() }
#pragma unroll
        for (int i = 0; i < k; ++i) { acc += (W)(L)m[i] * (W)(L)p29(k - i); BAR_P() }
        m[k] = (0u - (u32)acc) & AKP_MASK29; acc += (W)(L)m[k]; acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; ++k) {
#pragma unroll
        for (int i = k - 8; i < 9; ++i) { acc += (W)a.l[i] * (W)b.l[k - i]; BAR_P() }
#pragma unroll
        for (int i = k - 8; i < 9; ++i) { acc += (W)(L)m[i] * (W)(L)p29(k - i); BAR_P() }
        t.l[k - 9] = (L)((u32)acc & AKP_MASK29); acc >>= 29;
    }
    t.l[8] = (L)acc; return t;
}
template <int MODE>
__global__ void __launch_bounds__(256) k_f29_mulv(Fr* x, int iters) {
    const size_t i = blockIdx.x * blockDim.x + threadIdx.x;
    FU a = f29_from_wire<false>(x[i]), b = f29_from_wire<false>(x[i ^ 1]);
    for (int k = 0; k < iters; ++k) a = mulv<MODE>(a, b);
    x[i] = f29_to_wire(a);
}

// ---- hand-written assembly F29 routines (tools/asm_microbench.hsaco) vs the same loops in C++ on raw internal limbs ----
__global__ void k_to_internal(const Fr* w, F29Pad* x, size_t n) {
    const size_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) f29_store_pad(x + i, f29_from_wire<false>(w[i]));
}
__global__ void __launch_bounds__(256) k_raw_mul(F29Pad* x, int iters) {
    const size_t i = blockIdx.x * blockDim.x + threadIdx.x;
    FU a = f29_load_pad<false>(x + i); const FU b = f29_load_pad<false>(x + (i ^ 1));
    for (int k = 0; k < iters; ++k) a = f29_mul(a, b);
    f29_store_pad(x + i, a);
}
__global__ void __launch_bounds__(256) k_raw_sqr(F29Pad* x, int iters) {
    const size_t i = blockIdx.x * blockDim.x + threadIdx.x;
    FU a = f29_load_pad<false>(x + i);
    for (int k = 0; k < iters; ++k) a = f29_sqr(a);
    f29_store_pad(x + i, a);
}
__global__ void __launch_bounds__(256) k_raw_dot3(F29Pad* x, int iters) {
    const size_t i = blockIdx.x * blockDim.x + threadIdx.x;
    FU a = f29_load_pad<false>(x + i), b = f29_load_pad<false>(x + (i ^ 1)), c = f29_load_pad<false>(x + (i ^ 2));
    for (int k = 0; k < iters; ++k) { const FU r = f29_dot3(a, b, b, c, c, b); c = b; b = a; a = r; }
    f29_store_pad(x + i, a);
}

template <class F>
static float time_ms(F&& launch, int reps = 3) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch();
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0));
        launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double clk = prop.clockRate * 1e3;  // Hz
    printf("device %s, %d CUs, clock %.0f MHz\n", prop.name, cus, clk / 1e6);
    u64* d; CK(hipMalloc(&d, sizeof(u64) * cus * 8 * 256));
    Fr* df; CK(hipMalloc(&df, sizeof(Fr) * cus * 8 * 256));
    std::vector<Fr> h(cus * 8 * 256);
    for (size_t i = 0; i < h.size(); ++i) for (int j = 0; j < 8; ++j) h[i].l[j] = (u32)(i * 2654435761u + j * 40503u) & (j == 7 ? 0x3fffffffu : 0xffffffffu);
    CK(hipMemcpy(df, h.data(), sizeof(Fr) * h.size(), hipMemcpyHostToDevice));
    const int iters = 2000;
    printf("%-22s %6s %12s %14s\n", "instr", "w/SIMD", "ms", "cyc/wave-instr");
    struct K { const char* name; void (*fn)(u64*, int, u32, u32); int per_iter; };
    K ks[] = {{"v_mad_u64_u32", k_mad64, 64}, {"v_mad_u64_u32 dep", k_mad64_dep, 64}, {"mad+addc pair(x2)", k_mac_pair, 64},
              {"v_mad_i64_i32", k_madi64, 64}, {"v_lshrrev_b64", k_lshr64, 64}, {"v_and_b32", k_and32, 64}, {"v_mul_lo_u32", k_mullo, 64}, {"v_mul_hi_u32", k_mulhi, 64}, {"v_add_u32", k_add32, 64}, {"v_add_co_u32", k_addco, 64},
              {"v_addc_co_u32", k_addc, 64}, {"v_add3_u32", k_add3, 64}, {"v_cndmask_b32", k_cndmask, 64}, {"v_alignbit_b32", k_alignbit, 64},
              {"v_mad_u32_u24", k_mad24, 64}, {"v_lshl_add_u64", k_lshladd64, 64}, {"v_fma_f32", k_fma32, 64}, {"v_pk_fma_f32", k_pkfma32, 64},
              {"v_fma_f64", k_fma64, 64}, {"v_mul_f64", k_mulf64, 64}, {"v_add_f64", k_addf64, 64}};
    for (auto& k : ks) {
        for (int wps : {1, 2, 4}) {  // waves per SIMD: blocks of 256 threads = 4 waves = 1 per SIMD
            const int blocks = cus * wps;
            float ms = time_ms([&] { hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, d, iters, 12345u, 678u); });
            double cyc = ms * 1e-3 * clk / ((double)iters * k.per_iter * wps);
            printf("%-22s %6d %12.3f %14.2f\n", k.name, wps, ms, cyc);
        }
    }
    struct M { const char* name; void (*fn)(Fr*, int); int mul_per_iter; };
    M ms_[] = {{"f29_mul chain-split (hipcc)", k_f29_mulv<0>, 1}, {"f29_mul single chain+nops", k_f29_mulv<2>, 1}, {"f29_mul unsigned", k_f29_mul<false>, 1}, {"f29_mul signed", k_f29_mul<true>, 1}, {"f29_sqr", k_f29_sqr, 1}, {"f29_dot3 (3 products)", k_f29_dot3, 1},
               {"fr_mul asm", k_frmul_asm, 1}, {"fr_mul portable", k_frmul_portable, 1}, {"fr_add+fr_sub", k_fradd, 2}};
    const int miters = 2000;
    printf("%-22s %6s %12s %14s %16s\n", "field op", "w/SIMD", "ms", "cyc/wave-op", "ops/s (chip)");
    for (auto& k : ms_) {
        for (int wps : {1, 2, 4, 8}) {
            const int blocks = cus * wps;
            float ms = time_ms([&] { hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, df, miters); });
            double ops = (double)blocks * 256 * miters * k.mul_per_iter;
            double cyc = ms * 1e-3 * clk / ((double)miters * k.mul_per_iter * wps);
            printf("%-22s %6d %12.3f %14.1f %16.4g\n", k.name, wps, ms, cyc, ops / (ms * 1e-3));
        }
    }
    // ---- asm vs C++ on raw internal limbs: bit-exact comparison, then timing ----
    {
        hipModule_t mod;
        const char* path = getenv("AKP_ASM_HSACO") ? getenv("AKP_ASM_HSACO") : "tools/asm_microbench.hsaco";
        if (hipModuleLoad(&mod, path) != hipSuccess) { printf("asm microbench: cannot load %s (skipped)\n", path); return 0; }
        const size_t n = (size_t)cus * 8 * 256;
        F29Pad *xa, *xc, *x0;
        CK(hipMalloc(&xa, n * sizeof(F29Pad))); CK(hipMalloc(&xc, n * sizeof(F29Pad))); CK(hipMalloc(&x0, n * sizeof(F29Pad)));
        CK(hipMemcpy(df, h.data(), sizeof(Fr) * h.size(), hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_to_internal, dim3((unsigned)(n / 256)), dim3(256), 0, 0, df, x0, n);
        CK(hipDeviceSynchronize());
        struct AK { const char* name; const char* sym; void (*cpp)(F29Pad*, int); };
        AK aks[] = {{"f29_mul", "asm_f29_mul", k_raw_mul}, {"f29_sqr", "asm_f29_sqr", k_raw_sqr}, {"f29_dot3", "asm_f29_dot3", k_raw_dot3}};
        printf("%-26s (asm vs C++ on raw limbs)\n", "routine");
        for (auto& k : aks) {
            hipFunction_t fn; CK(hipModuleGetFunction(&fn, mod, k.sym));
            int it = 7;  // correctness: a few dependent iterations, every lane compared
            CK(hipMemcpy(xa, x0, n * sizeof(F29Pad), hipMemcpyDeviceToDevice)); CK(hipMemcpy(xc, x0, n * sizeof(F29Pad), hipMemcpyDeviceToDevice));
            struct { void* p; int iters; } args{xa, it};
            size_t asz = 12;
            void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &asz, HIP_LAUNCH_PARAM_END};
            CK(hipModuleLaunchKernel(fn, (unsigned)(n / 256), 1, 1, 256, 1, 1, 0, 0, nullptr, cfg));
            hipLaunchKernelGGL(k.cpp, dim3((unsigned)(n / 256)), dim3(256), 0, 0, xc, it);
            CK(hipDeviceSynchronize());
            std::vector<F29Pad> ra(n), rc(n);
            CK(hipMemcpy(ra.data(), xa, n * sizeof(F29Pad), hipMemcpyDeviceToHost)); CK(hipMemcpy(rc.data(), xc, n * sizeof(F29Pad), hipMemcpyDeviceToHost));
            size_t bad = 0;
            for (size_t i = 0; i < n; ++i) for (int j = 0; j < 9; ++j) if (ra[i].w[j] != rc[i].w[j]) { ++bad; break; }
            printf("%-26s parity asm==C++: %s (%zu of %zu lanes differ)\n", k.name, bad ? "FAIL" : "ok", bad, n);
            for (int wps : {1, 2, 4, 8}) {
                const int blocks = cus * wps;
                args.iters = miters; args.p = xa;
                float ms1 = time_ms([&] { CK(hipModuleLaunchKernel(fn, blocks, 1, 1, 256, 1, 1, 0, 0, nullptr, cfg)); });
                float ms2 = time_ms([&] { hipLaunchKernelGGL(k.cpp, dim3(blocks), dim3(256), 0, 0, xc, miters); });
                printf("%-26s %6d   asm %8.3f ms %8.1f cyc   C++ %8.3f ms %8.1f cyc\n", k.name, wps, ms1, ms1 * 1e-3 * clk / ((double)miters * wps), ms2,
                       ms2 * 1e-3 * clk / ((double)miters * wps));
            }
        }
    }
    return 0;
}
