// capi_ctx.hip -- part of libakp.so (implementation of include/akp.h): errors, contexts, pinned host memory, host field helpers
// Product code.  Never includes, links or calls anything under oracle/; there is no CPU fallback for any compute entry
// point (a missing device is AKP_ERR_HIP).
#include "capi_internal.hpp"

// ------------------------------------------------------------------------------------------
// errors
static thread_local std::string g_last_error;
int32_t fail(int32_t code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}
extern "C" const char* akp_last_error(void) { return g_last_error.c_str(); }
extern "C" int32_t akp_abi_version(void) { return AKP_ABI_VERSION; }
extern "C" int32_t akp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

size_t env_size(const char* name, size_t dflt) {
    const char* e = getenv(name);
    return (e && *e) ? (size_t)strtoull(e, nullptr, 10) : dflt;
}
u32 env_u32(const char* name, u32 dflt, u32 lo, u32 hi) {
    const char* e = getenv(name);
    if (!e) return dflt;
    long v = strtol(e, nullptr, 10);
    return (v < (long)lo || v > (long)hi) ? dflt : (u32)v;
}

// ------------------------------------------------------------------------------------------
// context: device, stream, grow-only scratch slots
void ctx_handle_released(akp_ctx* c) {
    if (c && --c->live_handles == 0 && c->dead) delete c;
}
// Copy streams are high-priority streams (round 6).  The runtime's staged copies, small copies, flag writes and device-to-host copies are
// KERNELS on the copy stream's hardware queue; where a stream's queue lands is the runtime's choice, and a queue that shares a pipe of
// the command processor with a hash kernel's queue is served only at the pipe's time slice (~0.43 ms) while that kernel's grid has
// workgroups left to place -- two normal-priority streams can even share one queue.  High-priority queues come from a pool of their own
// and are served first (profiles/r04_s3, r06_s41 ... s52).
hipError_t ctx_copy_streams(akp_ctx* c) {
    for (int i = 4; i <= 5; ++i)
        if (!c->pipe[i]) {
            int lo = 0, hi = 0;
            hipError_t e = hipDeviceGetStreamPriorityRange(&lo, &hi);
            if (e == hipSuccess) e = hipStreamCreateWithPriority(&c->pipe[i], hipStreamNonBlocking, hi);
            if (e != hipSuccess) {
                c->pipe[i] = nullptr;
                return e;
            }
        }
    return hipSuccess;
}
// scratch slot `slot` with at least `bytes`, to be used on stream `s`: if the previous use was enqueued on a different
// stream, `s` first waits for it (event record + stream wait; nothing blocks on the host)
int32_t ctx_scratch(akp_ctx* c, int slot, size_t bytes, void** out, hipStream_t s) {
    if (bytes == 0) bytes = 16;
    if (c->slot_used[slot] && c->slot_stream[slot] != s) {
        if (!c->slot_event[slot]) HIP_TRY(hipEventCreateWithFlags(&c->slot_event[slot], hipEventDisableTiming));
        HIP_TRY(hipEventRecord(c->slot_event[slot], c->slot_stream[slot]));
        HIP_TRY(hipStreamWaitEvent(s, c->slot_event[slot], 0));
    }
    c->slot_used[slot] = true;
    c->slot_stream[slot] = s;
    if (c->scratch_bytes[slot] < bytes) {
        // grow: a NEW block of at least 1.5 x the old size; the old one is RETIRED (freed with the context) -- work enqueued on other
        // streams may still read it, and draining the device here (round 5) would also wait for whatever else runs on it: the
        // background build of a curve table, other contexts' kernels.  Retired blocks add up to less than twice the largest.
        void* fresh = nullptr;
        const size_t want = std::max(bytes, c->scratch_bytes[slot] + c->scratch_bytes[slot] / 2);
        hipError_t e = hipMalloc(&fresh, want);
        size_t got = want;
        if (e != hipSuccess && want > bytes) {
            (void)hipGetLastError();
            e = hipMalloc(&fresh, bytes);
            got = bytes;
        }
        if (e != hipSuccess) return fail(AKP_ERR_HIP, "scratch of %zu MB: %s", bytes >> 20, hipGetErrorString(e));
        if (c->scratch[slot]) c->scratch_retired.push_back(c->scratch[slot]);
        c->scratch[slot] = fresh;
        c->scratch_bytes[slot] = got;
    }
    *out = c->scratch[slot];
    return AKP_OK;
}
extern "C" int32_t akp_ctx_create(int32_t device_id, akp_ctx** out) {
    if (!out) return fail(AKP_ERR_BAD_PARAMS, "akp_ctx_create: out is NULL");
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device_id < 0 || device_id >= n) return fail(AKP_ERR_HIP, "akp_ctx_create: device %d not present (%d visible)", device_id, n);
    HIP_TRY(hipSetDevice(device_id));
    akp_ctx* c = new akp_ctx();
    c->device = device_id;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete c;
        return fail(AKP_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    *out = c;
    return AKP_OK;
}
extern "C" void akp_ctx_destroy(akp_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    for (void* r : c->scratch_retired) (void)hipFree(r);
    c->scratch_retired.clear();
    for (int i = 0; i < SCR_COUNT; ++i) {
        if (c->scratch[i]) (void)hipFree(c->scratch[i]);
        if (c->slot_event[i]) (void)hipEventDestroy(c->slot_event[i]);
    }
    for (int i = 0; i < 8; ++i)
        if (c->chunk_event[i]) (void)hipEventDestroy(c->chunk_event[i]);
    if (c->pinned) (void)hipHostFree(c->pinned);
    if (c->gate_flags) (void)hipFree(c->gate_flags);
    if (c->gate_done) (void)hipHostFree(c->gate_done);
    c->gate_flags = nullptr;
    c->gate_done = c->gate_done_dev = nullptr;
    c->gate_done_cap = 0;
    for (int i = 0; i < 7; ++i)
        if (c->pipe[i]) (void)hipStreamDestroy(c->pipe[i]);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    c->stream = nullptr;
    for (int i = 0; i < SCR_COUNT; ++i) {
        c->scratch[i] = nullptr;
        c->scratch_bytes[i] = 0;
        c->slot_event[i] = nullptr;
        c->slot_used[i] = false;
    }
    for (int i = 0; i < 8; ++i) c->chunk_event[i] = nullptr;
    for (int i = 0; i < 7; ++i) c->pipe[i] = nullptr;
    c->pinned = nullptr;
    c->pinned_bytes = 0;
    c->last_tree_non_leaf = nullptr;
    if (c->live_handles > 0) {
        c->dead = true;  // freed by ctx_handle_released when the last handle goes
        return;
    }
    delete c;
}
extern "C" int32_t akp_ctx_synchronize(akp_ctx* c) {
    if (!c) return fail(AKP_ERR_BAD_PARAMS, "ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return AKP_OK;
}
extern "C" void* akp_ctx_stream(akp_ctx* c) { return c ? (void*)c->stream : nullptr; }

// ------------------------------------------------------------------------------------------
// effective shader clock (measurement plumbing for bench.py).  One wave runs a chain of dependent
// v_mad_u64_u32 and reads s_memtime (shader-clock cycles on gfx950: 8.25 per dependent multiply-add at any clock,
// profiles/r04_s4/clock_probe.txt) and s_memrealtime (constant 100 MHz) before and after: MHz = 100 * d(s_memtime) / d(s_memrealtime).
// The DPM level sysfs reports ("sclk") is the ceiling of the current power state, not what the ALUs ran at.
__global__ void clock_probe_kernel(uint64_t* __restrict__ out, u32 n, u32 seed) {
    unsigned long long acc = seed + threadIdx.x;
    const u32 a = seed | 1u;
    const unsigned long long t0 = clock64(), w0 = wall_clock64();
#pragma unroll 1
    for (u32 i = 0; i < n; i += 16) {
#pragma unroll
        for (int k = 0; k < 16; ++k) acc = (unsigned long long)(u32)acc * a + acc;
    }
    const unsigned long long t1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) {
        out[0] = t1 - t0;
        out[1] = w1 - w0;
        out[2] = acc;
    }
}
extern "C" int32_t akp_clock_probe_dev(akp_ctx* c, uint32_t chain_len, uint64_t* d_out3, void* stream) {
    if (!c) return fail(AKP_ERR_HIP, "akp_clock_probe_dev: a device context is required");
    if (c->dead) return fail(AKP_ERR_BAD_PARAMS, "akp_clock_probe_dev: the context was destroyed");
    if (!d_out3 || chain_len == 0) return fail(AKP_ERR_BAD_PARAMS, "akp_clock_probe_dev: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, d_out3, (chain_len + 15u) & ~15u, 0x9e3779b9u);
    HIP_TRY(hipGetLastError());
    return AKP_OK;
}

// ------------------------------------------------------------------------------------------
// pinned host memory for callers that want the host-pointer entry points to run at PCIe speed: copies from / to
// pageable memory are staged by the runtime and block the calling thread, pinned (or registered) buffers stream
// asynchronously in both directions at once.  The entry points accept either kind.
extern "C" int32_t akp_host_alloc(size_t bytes, void** out) {
    if (!out) return fail(AKP_ERR_BAD_PARAMS, "out is NULL");
    HIP_TRY(hipHostMalloc(out, bytes ? bytes : 16, hipHostMallocDefault));
    return AKP_OK;
}
extern "C" int32_t akp_host_free(void* p) {
    if (p) HIP_TRY(hipHostFree(p));
    return AKP_OK;
}
extern "C" int32_t akp_host_register(void* p, size_t bytes) {
    if (!p) return fail(AKP_ERR_BAD_PARAMS, "pointer is NULL");
    HIP_TRY(hipHostRegister(p, bytes, hipHostRegisterDefault));
    return AKP_OK;
}
extern "C" int32_t akp_host_unregister(void* p) {
    if (p) HIP_TRY(hipHostUnregister(p));
    return AKP_OK;
}

void* device_alias(const void* host, size_t bytes) {
    if (!host || bytes == 0) return nullptr;
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, host) != hipSuccess) {
        (void)hipGetLastError();  // pageable memory: not an error for the caller
        return nullptr;
    }
    if (a.type != hipMemoryTypeHost || !a.devicePointer) return nullptr;
    hipDeviceptr_t base = nullptr;
    size_t size = 0;
    if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)a.devicePointer) == hipSuccess) {
        if ((const char*)a.devicePointer + bytes > (const char*)base + size) return nullptr;  // only part of the buffer is pinned
    } else {
        (void)hipGetLastError();
    }
    return a.devicePointer;
}

// ------------------------------------------------------------------------------------------
// host field helpers
extern "C" int32_t akp_fr_to_mont(const uint64_t* canonical, uint64_t* mont, size_t n) {
    if ((!canonical || !mont) && n) return fail(AKP_ERR_BAD_PARAMS, "akp_fr_to_mont: NULL buffer");
    for (size_t i = 0; i < n; ++i) {
        if (!fr_words_reduced(canonical + 4 * i)) return fail(AKP_ERR_BAD_PARAMS, "akp_fr_to_mont: element %zu is not < p", i);
        fr_to_words(fr_to_mont(fr_from_words(canonical + 4 * i)), mont + 4 * i);
    }
    return AKP_OK;
}
extern "C" int32_t akp_fr_from_mont(const uint64_t* mont, uint64_t* canonical, size_t n) {
    if ((!canonical || !mont) && n) return fail(AKP_ERR_BAD_PARAMS, "akp_fr_from_mont: NULL buffer");
    for (size_t i = 0; i < n; ++i) fr_to_words(fr_from_mont(fr_from_words(mont + 4 * i)), canonical + 4 * i);
    return AKP_OK;
}

