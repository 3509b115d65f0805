// lane_mul_microbench.hip -- prototype of the LANE-DISTRIBUTED Montgomery product sketched in DESIGN.md section 7.6, and the
// measurement behind its numbers: latency of a chain of dependent field products on ONE wave,
//   (a) today's routine: f29_mul, all nine limbs of an element in one lane (one chain of 188 instructions, 153 multiply-adds);
//   (b) the prototype: one element per wave, limb j in lane j; every column of a*b forms at once (operand shifted across lanes by
//       DPP row_shr, the other operand's limb broadcast by v_readlane), interleaved Montgomery digits broadcast the same way, a
//       value-preserving parallel carry step per digit.
// Not product code (nothing in the library uses it); results are checked against f29_mul lane by lane.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lane_mul_microbench.hip -o tools/lane_mul_microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../crypto_primitives_amd/csrc/fr.hpp"
#include "../crypto_primitives_amd/csrc/f29.hpp"
using namespace akp;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
constexpr u32 MASK29 = (1u << 29) - 1u;

// lane k <- lane k - I of the same 16-lane row (zero shifted in)
template <int I>
__device__ __forceinline__ u32 row_shr(u32 v) {
    if constexpr (I == 0) return v;
    else return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x110 + I, 0xF, 0xF, true);
}
// lane k <- lane k + I
template <int I>
__device__ __forceinline__ u32 row_shl(u32 v) {
    return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x100 + I, 0xF, 0xF, true);
}
__device__ __forceinline__ u32 bcast(u32 v, int lane) { return (u32)__builtin_amdgcn_readlane((int)v, lane); }

struct LaneP {
    u32 sh[9];  // sh[i]: lane k holds p_{k-i} (the modulus shifted right by i lanes), zero elsewhere
};

// value-preserving carry step over the 16 column lanes; the carry that leaves lane 15 goes to `top`
__device__ __forceinline__ void carry_step(u64& acc, u64& top) {
    const u64 c = acc >> 29;
    acc = (u64)((u32)acc & MASK29);
    const u32 cl = row_shr<1>((u32)c), ch = row_shr<1>((u32)(c >> 32));
    acc += ((u64)ch << 32) | cl;
    top += c;  // meaningful in lane 15 only: column 15 -> 16
}

// a * b / 2^261 mod p (lazy: limbs < 2^29 + small, value < 2^257); a, b: limb j in lane j (j = 0..8), zero in lanes 9..15
__device__ __forceinline__ u32 lmul(const u32 a, const u32 b, const LaneP& P, const u32 lane16) {
    u64 acc = 0;
#define PROD(I) acc += (u64)bcast(a, I) * (u64)row_shr<I>(b);
    PROD(0) PROD(1) PROD(2) PROD(3) PROD(4) PROD(5) PROD(6) PROD(7) PROD(8)
#undef PROD
    u64 top = (u64)bcast(a, 8) * (u64)row_shr<7>(b);  // lane 15: column 16 = a_8 * b_8
#define RED(I)                                                          \
    {                                                                   \
        const u32 m = bcast((0u - (u32)acc) & MASK29, I);               \
        acc += (u64)m * (u64)P.sh[I];                                   \
        if (I == 8) top += (u64)m * (u64)P.sh[7]; /* lane 15: m_8 * p_8 */ \
        carry_step(acc, top);                                           \
    }
    RED(0) RED(1) RED(2) RED(3) RED(4) RED(5) RED(6) RED(7) RED(8)
#undef RED
    carry_step(acc, top);  // limbs of columns 9..15 below 2^29 + 4
    // columns 9..15 -> lanes 0..6; column 16 (top, lane 15) -> limb 7 (lane 7) and its overflow -> limb 8 (lane 8)
    const u32 lo = row_shl<9>((u32)acc);
    const u32 t7 = row_shl<8>((u32)top & MASK29), t8 = row_shl<7>((u32)(top >> 29));
    return lane16 < 7 ? lo : (lane16 == 7 ? t7 : (lane16 == 8 ? t8 : 0u));
}

__global__ void __launch_bounds__(64) lane_chain_kernel(const u32* in_a, const u32* in_b, const u32* p_limbs, u32* out, int iters) {
    const u32 lane = threadIdx.x & 63u, l16 = lane & 15u;
    LaneP P;
#pragma unroll
    for (int i = 0; i < 9; ++i) P.sh[i] = (l16 >= (u32)i && l16 - i < 9) ? p_limbs[l16 - i] : 0u;
    u32 x = l16 < 9 ? in_a[l16] : 0u;
    const u32 y = l16 < 9 ? in_b[l16] : 0u;
    for (int it = 0; it < iters; ++it) x = lmul(x, y, P, l16);
    if (lane < 9) out[lane] = x;
}

__global__ void __launch_bounds__(64) serial_chain_kernel(const u32* in_a, const u32* in_b, u32* out, int iters) {
    FU x, y;
#pragma unroll
    for (int i = 0; i < 9; ++i) { x.l[i] = in_a[i]; y.l[i] = in_b[i]; }
    for (int it = 0; it < iters; ++it) x = f29_mul(x, y);
    if (threadIdx.x == 0)
        for (int i = 0; i < 9; ++i) out[i] = x.l[i];
}
// canonical representatives of two lazy results (device: f29_canonical_pack)
__global__ void canon_kernel(const u32* a, const u32* b, u32* out) {
    FU x, y;
    for (int i = 0; i < 9; ++i) { x.l[i] = a[i]; y.l[i] = b[i]; }
    const Fr cx = f29_canonical_pack(x), cy = f29_canonical_pack(y);
    for (int i = 0; i < 8; ++i) { out[i] = cx.l[i]; out[8 + i] = cy.l[i]; }
}

int main() {
    std::vector<u32> a(9), b(9), p(9);
    const FU pp = f29_p<false>();
    for (int i = 0; i < 9; ++i) p[i] = pp.l[i];
    srand(7);
    for (int i = 0; i < 9; ++i) { a[i] = ((u32)rand() << 8 ^ (u32)rand()) & MASK29; b[i] = ((u32)rand() << 8 ^ (u32)rand()) & MASK29; }
    a[8] &= (1u << 22) - 1u;  // values below 2^254
    b[8] &= (1u << 22) - 1u;
    u32 *da, *db, *dp, *o1, *o2, *oc;
    CK(hipMalloc(&da, 36)); CK(hipMalloc(&db, 36)); CK(hipMalloc(&dp, 36)); CK(hipMalloc(&o1, 36)); CK(hipMalloc(&o2, 36)); CK(hipMalloc(&oc, 64));
    CK(hipMemcpy(da, a.data(), 36, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), 36, hipMemcpyHostToDevice)); CK(hipMemcpy(dp, p.data(), 36, hipMemcpyHostToDevice));
    bool all_ok = true;
    for (int iters : {1, 2, 5, 33}) {  // parity of the chain after `iters` products
        hipLaunchKernelGGL(lane_chain_kernel, dim3(1), dim3(64), 0, 0, da, db, dp, o1, iters);
        hipLaunchKernelGGL(serial_chain_kernel, dim3(1), dim3(64), 0, 0, da, db, o2, iters);
        hipLaunchKernelGGL(canon_kernel, dim3(1), dim3(1), 0, 0, o1, o2, oc);
        u32 c[16];
        CK(hipMemcpy(c, oc, 64, hipMemcpyDeviceToHost));
        bool ok = true;
        for (int i = 0; i < 8; ++i) ok = ok && c[i] == c[8 + i];
        printf("parity after %2d products: %s\n", iters, ok ? "equal" : "DIFFERENT");
        all_ok = all_ok && ok;
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int N = 20000;
    for (int which = 0; which < 2; ++which) {
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0));
            if (which == 0) hipLaunchKernelGGL(serial_chain_kernel, dim3(1), dim3(64), 0, 0, da, db, o2, N);
            else hipLaunchKernelGGL(lane_chain_kernel, dim3(1), dim3(64), 0, 0, da, db, dp, o1, N);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        printf("%-34s %8.3f ms for %d dependent products = %7.1f ns per product (%.0f cycles at 2.4 GHz)\n",
               which == 0 ? "f29_mul, limbs in one lane" : "lane-distributed product", best, N, best * 1e6 / N, best * 1e6 / N * 2.4);
    }
    return all_ok ? 0 : 1;
}
