// ragged_sort.hpp -- launch order for batches whose items differ in length (round 5): a counting sort of the item indices by a
// small integer key (table steps of a curve hash, permutations of a Poseidon sponge), LONGEST FIRST, so that the 64 lanes of a
// wave run the same number of steps and the long items start before the short ones fill the tail.  Three small kernels on the
// caller's stream, nothing on the host: the offsets may live in device memory only.  The order inside one key is whatever the
// atomics produce -- results are stored by item index, so the output does not depend on it.
//   keys:    key[i] = key_of(offsets[i+1] - offsets[i]),  hist[key]++          (one thread per item)
//   scan:    cursor[k] = number of items with a key > k                        (one workgroup; keys < RAGGED_MAX_KEYS)
//   scatter: order[cursor[key[i]]++] = i                                       (one thread per item)
// Scratch: (n + 2 * RAGGED_MAX_KEYS) u32 for keys, hist, cursor + n u32 for the order.
#pragma once
#include <hip/hip_runtime.h>

#include "akp_types.hpp"

namespace akp {

constexpr u32 RAGGED_MAX_KEYS = 4096;

// what the key of an item is computed from
struct RaggedKey {
    u32 mode;   // 0: ceil(min(8 len, cap) / unit) (Pedersen digits; cap = n_gen, unit = D)
                // 1: chunks = min(ceil(8 len / 3), cap); unit > 1: chunks / unit + chunks % unit, else chunks (Bowe-Hopwood)
                // 2: max(1, ceil(len / unit)) (Poseidon permutations; len in elements, unit = rate)
    u32 unit, cap;
};
AKP_HD u32 ragged_key_of(const RaggedKey& k, uint64_t len) {
    if (len > 0xffffffffu) len = 0;  // offsets that decrease (the item hashes as the empty one): `len * 8` below must not wrap
    u32 v;
    if (k.mode == 0) {
        const uint64_t used = len * 8 < k.cap ? len * 8 : k.cap;
        v = (u32)((used + k.unit - 1) / k.unit);
    } else if (k.mode == 1) {
        const uint64_t ch = (len * 8 + 2) / 3;
        const u32 chunks = (u32)(ch < k.cap ? ch : k.cap);
        v = k.unit > 1 ? chunks / k.unit + chunks % k.unit : chunks;
    } else {
        v = len == 0 ? 1u : (u32)((len + k.unit - 1) / k.unit);
    }
    return v < RAGGED_MAX_KEYS ? v : RAGGED_MAX_KEYS - 1;
}
#if defined(__HIPCC__)
constexpr u32 RAGGED_ITEMS_PER_THREAD = 8;  // a workgroup of 256 threads handles 2048 items: its LDS histogram is zeroed and flushed once
// (one global atomic per item would serialise on a handful of addresses when many items share a key: a uniform batch is ONE key)
static __global__ void __launch_bounds__(256) ragged_keys_kernel(const uint64_t* __restrict__ offsets, size_t n, RaggedKey k, u32* __restrict__ key,
                                                                u32* __restrict__ hist) {
    __shared__ u32 h[RAGGED_MAX_KEYS];
    for (u32 j = threadIdx.x; j < RAGGED_MAX_KEYS; j += 256) h[j] = 0;
    __syncthreads();
    const size_t first = (size_t)blockIdx.x * (256 * RAGGED_ITEMS_PER_THREAD);
    for (u32 r = 0; r < RAGGED_ITEMS_PER_THREAD; ++r) {
        const size_t i = first + r * 256 + threadIdx.x;
        if (i < n) {
            const u32 v = ragged_key_of(k, offsets[i + 1] - offsets[i]);
            key[i] = v;
            atomicAdd(h + v, 1u);
        }
    }
    __syncthreads();
    for (u32 j = threadIdx.x; j < RAGGED_MAX_KEYS; j += 256)
        if (h[j]) atomicAdd(hist + j, h[j]);
}
// cursor[k] = sum of hist[j] for j > k (descending order: the longest items first); one workgroup of 256 threads
static __global__ void __launch_bounds__(256) ragged_scan_kernel(const u32* __restrict__ hist, u32* __restrict__ cursor) {
    __shared__ u32 part[256];
    constexpr u32 per = RAGGED_MAX_KEYS / 256;
    const u32 t = threadIdx.x;
    // thread t owns keys [hi - per + 1, hi], hi = RAGGED_MAX_KEYS - 1 - t * per: thread 0 the largest keys
    const u32 hi = RAGGED_MAX_KEYS - 1u - t * per;
    u32 sum = 0;
    for (u32 j = 0; j < per; ++j) sum += hist[hi - j];
    part[t] = sum;
    __syncthreads();
    u32 before = 0;
    for (u32 j = 0; j < t; ++j) before += part[j];
    for (u32 j = 0; j < per; ++j) {
        cursor[hi - j] = before;
        before += hist[hi - j];
    }
}
// rank inside the workgroup through LDS, ONE global atomic per (workgroup, key) to reserve the key's range
static __global__ void __launch_bounds__(256) ragged_scatter_kernel(const u32* __restrict__ key, size_t n, u32* __restrict__ cursor, u32* __restrict__ order) {
    __shared__ u32 cnt[RAGGED_MAX_KEYS], base[RAGGED_MAX_KEYS];
    for (u32 j = threadIdx.x; j < RAGGED_MAX_KEYS; j += 256) cnt[j] = 0;
    __syncthreads();
    const size_t first = (size_t)blockIdx.x * (256 * RAGGED_ITEMS_PER_THREAD);
    u32 kv[RAGGED_ITEMS_PER_THREAD], rk[RAGGED_ITEMS_PER_THREAD];
#pragma unroll
    for (u32 r = 0; r < RAGGED_ITEMS_PER_THREAD; ++r) {
        const size_t i = first + r * 256 + threadIdx.x;
        kv[r] = 0;
        rk[r] = 0;
        if (i < n) {
            kv[r] = key[i];
            rk[r] = atomicAdd(cnt + kv[r], 1u);
        }
    }
    __syncthreads();
    for (u32 j = threadIdx.x; j < RAGGED_MAX_KEYS; j += 256)
        if (cnt[j]) base[j] = atomicAdd(cursor + j, cnt[j]);
    __syncthreads();
#pragma unroll
    for (u32 r = 0; r < RAGGED_ITEMS_PER_THREAD; ++r) {
        const size_t i = first + r * 256 + threadIdx.x;
        if (i < n) order[base[kv[r]] + rk[r]] = (u32)i;
    }
}
// d_work: (n + 2 * RAGGED_MAX_KEYS) u32, d_order: n u32.  Enqueues on s; returns the first HIP error.
static inline hipError_t ragged_order(const uint64_t* d_offsets, size_t n, const RaggedKey& k, u32* d_work, u32* d_order, hipStream_t s) {
    u32 *key = d_work, *hist = d_work + n, *cursor = hist + RAGGED_MAX_KEYS;
    hipError_t e = hipMemsetAsync(hist, 0, 2 * RAGGED_MAX_KEYS * sizeof(u32), s);
    if (e != hipSuccess) return e;
    const unsigned grid = (unsigned)((n + 256 * RAGGED_ITEMS_PER_THREAD - 1) / (256 * RAGGED_ITEMS_PER_THREAD));
    hipLaunchKernelGGL(ragged_keys_kernel, dim3(grid), dim3(256), 0, s, d_offsets, n, k, key, hist);
    hipLaunchKernelGGL(ragged_scan_kernel, dim3(1), dim3(256), 0, s, hist, cursor);
    hipLaunchKernelGGL(ragged_scatter_kernel, dim3(grid), dim3(256), 0, s, key, n, cursor, d_order);
    return hipGetLastError();
}
#endif

}  // namespace akp
