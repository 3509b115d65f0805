// capi.hip -- implementation of include/akp.h: host-side C++ over the gfx950 kernels.
// Product code.  Never includes, links or calls anything under oracle/; there is no CPU
// fallback for any compute entry point (a missing device is AKP_ERR_HIP).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/akp.h"
#include "fr.hpp"
#include "f29.hpp"
#include "poseidon_kernels.hpp"
#include "poseidon_opt.hpp"
#include "te_kernels.hpp"

using namespace akp;

// ------------------------------------------------------------------------------------------
// errors
static thread_local std::string g_last_error;
static int32_t fail(int32_t code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}
#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return fail(AKP_ERR_HIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

extern "C" const char* akp_last_error(void) { return g_last_error.c_str(); }
extern "C" int32_t akp_abi_version(void) { return AKP_ABI_VERSION; }
extern "C" int32_t akp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static size_t env_size(const char* name, size_t dflt) {
    const char* e = getenv(name);
    return (e && *e) ? (size_t)strtoull(e, nullptr, 10) : dflt;
}
static u32 env_u32(const char* name, u32 dflt, u32 lo, u32 hi) {
    const char* e = getenv(name);
    if (!e) return dflt;
    long v = strtol(e, nullptr, 10);
    return (v < (long)lo || v > (long)hi) ? dflt : (u32)v;
}

// ------------------------------------------------------------------------------------------
// context: device, stream, grow-only scratch slots
enum { SCR_A = 0, SCR_B, SCR_C, SCR_D, SCR_E, SCR_F, SCR_G, SCR_H, SCR_I, SCR_J, SCR_K, SCR_L, SCR_COUNT };
struct akp_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    void* scratch[SCR_COUNT] = {};
    size_t scratch_bytes[SCR_COUNT] = {};
    // stream that last used each slot + an event to order the next use on ANOTHER stream behind it: `_dev` entry points
    // run on the caller's stream while the host-pointer entry points run on `stream` (non-blocking, so no implicit order
    // with the legacy default stream); without this two calls on different streams would race on the shared scratch
    hipStream_t slot_stream[SCR_COUNT] = {};
    bool slot_used[SCR_COUNT] = {};
    hipEvent_t slot_event[SCR_COUNT] = {};
    // pinned staging for small host<->device transfers of the tree / proof entry points
    void* pinned = nullptr;
    size_t pinned_bytes = 0;
    // more streams for the chunked host-pointer batches (copy-in / kernel / copy-out of consecutive chunks overlap)
    hipStream_t pipe[7] = {};
    hipEvent_t chunk_event[8] = {};  // copy-stream -> compute-stream hand-over of leaf chunks (host_tree_build)
    // where the last host-pointer tree build left its inner nodes (heap order, `last_tree_nodes` digests): what the
    // multi-device build reads for the all-gather of the sub-roots and the per-device copy-outs -- an explicit hand-over
    // instead of a convention about scratch slots
    const void* last_tree_non_leaf = nullptr;
    size_t last_tree_nodes = 0;
    // parameter handles created on this context and still alive.  akp_ctx_destroy with live handles releases the device
    // resources and marks the context dead; the struct itself goes with the last handle, whose compute calls fail cleanly
    // until then (handles created on akp_multi_ctx may outlive akp_multi_destroy without touching freed memory)
    int live_handles = 0;
    bool dead = false;
};
static void ctx_handle_released(akp_ctx* c) {
    if (c && --c->live_handles == 0 && c->dead) delete c;
}
// scratch slot `slot` with at least `bytes`, to be used on stream `s`: if the previous use was enqueued on a different
// stream, `s` first waits for it (event record + stream wait; nothing blocks on the host)
static int32_t ctx_scratch(akp_ctx* c, int slot, size_t bytes, void** out, hipStream_t s) {
    if (bytes == 0) bytes = 16;
    if (c->slot_used[slot] && c->slot_stream[slot] != s) {
        if (!c->slot_event[slot]) HIP_TRY(hipEventCreateWithFlags(&c->slot_event[slot], hipEventDisableTiming));
        HIP_TRY(hipEventRecord(c->slot_event[slot], c->slot_stream[slot]));
        HIP_TRY(hipStreamWaitEvent(s, c->slot_event[slot], 0));
    }
    c->slot_used[slot] = true;
    c->slot_stream[slot] = s;
    if (c->scratch_bytes[slot] < bytes) {
        if (c->scratch[slot]) {
            HIP_TRY(hipDeviceSynchronize());  // work enqueued on other streams may still read it
            HIP_TRY(hipFree(c->scratch[slot]));
            c->scratch[slot] = nullptr;
            c->scratch_bytes[slot] = 0;
        }
        HIP_TRY(hipMalloc(&c->scratch[slot], bytes));
        c->scratch_bytes[slot] = bytes;
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
    for (int i = 0; i < SCR_COUNT; ++i) {
        if (c->scratch[i]) (void)hipFree(c->scratch[i]);
        if (c->slot_event[i]) (void)hipEventDestroy(c->slot_event[i]);
    }
    for (int i = 0; i < 8; ++i)
        if (c->chunk_event[i]) (void)hipEventDestroy(c->chunk_event[i]);
    if (c->pinned) (void)hipHostFree(c->pinned);
    for (int i = 0; i < 7; ++i)
        if (c->pipe[i]) (void)hipStreamDestroy(c->pipe[i]);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    c->stream = nullptr;
    for (int i = 0; i < SCR_COUNT; ++i) { c->scratch[i] = nullptr; c->scratch_bytes[i] = 0; c->slot_event[i] = nullptr; c->slot_used[i] = false; }
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
// `_dev` entry points use the caller's stream verbatim (NULL = HIP's legacy default stream, which is what
// torch's default stream is), so event timing and ordering follow the caller's stream semantics.
static inline hipStream_t pick_stream(akp_ctx*, void* s) { return (hipStream_t)s; }

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

// Chunked host-pointer batch: items are cut into chunks of 2^AKP_HOST_CHUNK_LOG2 (default 2^18); chunk i runs copy-in ->
// kernel -> copy-out on stream i mod 3 with its own third of the device buffers, so the three stages of consecutive
// chunks overlap (PCIe is full duplex; the kernel of one chunk hides the copies of its neighbours).  Only for kernels
// without context scratch of their own (the Poseidon batches).
struct HostIn {
    const void* host;
    size_t bytes_per_item;
    int slot;
};
// Device alias of a host buffer the GPU can address directly -- memory from akp_host_alloc / hipHostMalloc or a range
// registered with akp_host_register / hipHostRegister -- or nullptr for ordinary pageable memory.
static void* device_alias(const void* host, size_t bytes) {
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
template <class Launch>
static int32_t pipelined_batch(akp_ctx* c, size_t n, const HostIn* ins, int n_in, void* host_out, size_t out_bytes_per_item, int out_slot,
                               Launch launch /* (void* const* d_in, void* d_out, size_t count, hipStream_t) */) {
    // Zero copy: when every buffer is pinned / registered host memory the kernels read and write it in place over PCIe.
    // Each item is read once and written once, the kernels are compute-bound, and loads and stores of different waves use
    // both directions of the link at the same time -- which the copy engines of this platform do not (opposite copies
    // mostly serialise, profiles/r02_s3): 2^20 permutations 3.79 -> 3.00 ms, 2^22 11.2 -> 10.2 ms (profiles/r02_s33).
    // AKP_HOST_ZERO_COPY=0 keeps the copy pipeline (A/B arm).
    static const bool zero_copy = env_u32("AKP_HOST_ZERO_COPY", 1, 0, 1) != 0;
    if (zero_copy && n) {
        void* di[2] = {nullptr, nullptr};
        bool all = true;
        for (int k = 0; k < n_in && all; ++k) {
            if (ins[k].bytes_per_item == 0) continue;
            di[k] = device_alias(ins[k].host, n * ins[k].bytes_per_item);
            all = di[k] != nullptr;
        }
        void* dout = nullptr;
        if (all) {
            dout = out_slot < 0 ? di[0] : device_alias(host_out, n * out_bytes_per_item);
            all = dout != nullptr;
        }
        if (all) {
            if (int32_t rc = launch(di, dout, n, c->stream)) return rc;
            HIP_TRY(hipStreamSynchronize(c->stream));
            return AKP_OK;
        }
    }
    static const size_t chunk_items = (size_t)1 << env_u32("AKP_HOST_CHUNK_LOG2", 18, 10, 30);
    static const int max_lanes = (int)env_u32("AKP_HOST_LANES", 3, 3, 8);  // >= depth + 1 buffers in flight
    hipStream_t st[8] = {c->stream};
    for (int i = 0; i + 1 < max_lanes; ++i) {
        if (!c->pipe[i]) HIP_TRY(hipStreamCreateWithFlags(&c->pipe[i], hipStreamNonBlocking));
        st[i + 1] = c->pipe[i];
    }
    const size_t chunk = std::min(n, chunk_items);
    const int lanes = n > chunk ? max_lanes : 1;
    void* d_in[2] = {nullptr, nullptr};
    void* d_out = nullptr;
    for (int k = 0; k < n_in; ++k)
        if (int32_t rc = ctx_scratch(c, ins[k].slot, lanes * chunk * ins[k].bytes_per_item, &d_in[k], c->stream)) return rc;
    const bool in_place = out_slot < 0;  // the output overwrites input 0 (permutation)
    if (!in_place)
        if (int32_t rc = ctx_scratch(c, out_slot, lanes * chunk * out_bytes_per_item, &d_out, c->stream)) return rc;
    // Submission order matters: the runtime feeds the copies of all streams to the copy engines in the order they were
    // issued, and a copy-out that still waits for its kernel blocks the copies queued behind it (measured: rocprofv3
    // --memory-copy-trace, profiles/r02_s3).  So the copy-out of chunk i is issued only after the copy-in and kernel of
    // chunk i + 2: by the time the engine reaches it, its kernel has finished.
    const size_t n_chunks = (n + chunk - 1) / chunk;
    const size_t depth = 2;
    void* di[2];
    for (size_t ci = 0; ci < n_chunks + depth; ++ci) {
        if (ci < n_chunks) {
            const size_t done = ci * chunk, cnt = std::min(chunk, n - done);
            const int lane = (int)(ci % lanes);
            hipStream_t s = st[lane];
            for (int k = 0; k < n_in; ++k) {
                di[k] = (char*)d_in[k] + (size_t)lane * chunk * ins[k].bytes_per_item;
                if (ins[k].bytes_per_item)
                    HIP_TRY(hipMemcpyAsync(di[k], (const char*)ins[k].host + done * ins[k].bytes_per_item, cnt * ins[k].bytes_per_item, hipMemcpyHostToDevice, s));
            }
            void* dout = in_place ? di[0] : (char*)d_out + (size_t)lane * chunk * out_bytes_per_item;
            if (int32_t rc = launch(di, dout, cnt, s)) return rc;
        }
        if (ci >= depth) {
            const size_t co = ci - depth, done = co * chunk, cnt = std::min(chunk, n - done);
            const int lane = (int)(co % lanes);
            const void* dout = in_place ? (char*)d_in[0] + (size_t)lane * chunk * ins[0].bytes_per_item : (char*)d_out + (size_t)lane * chunk * out_bytes_per_item;
            HIP_TRY(hipMemcpyAsync((char*)host_out + done * out_bytes_per_item, dout, cnt * out_bytes_per_item, hipMemcpyDeviceToHost, st[lane]));
        }
    }
    for (int i = 0; i < lanes; ++i) HIP_TRY(hipStreamSynchronize(st[i]));
    return AKP_OK;
}

// ------------------------------------------------------------------------------------------
// host field helpers
static inline Fr fr_from_words(const uint64_t* w) {
    Fr f;
    for (int i = 0; i < 4; ++i) {
        f.l[2 * i] = (u32)w[i];
        f.l[2 * i + 1] = (u32)(w[i] >> 32);
    }
    return f;
}
static inline void fr_to_words(const Fr& f, uint64_t* w) {
    for (int i = 0; i < 4; ++i) w[i] = (uint64_t)f.l[2 * i] | ((uint64_t)f.l[2 * i + 1] << 32);
}
static bool fr_words_reduced(const uint64_t* w) {
    const uint64_t P[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL};
    for (int i = 3; i >= 0; --i) {
        if (w[i] < P[i]) return true;
        if (w[i] > P[i]) return false;
    }
    return false;
}
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

// ------------------------------------------------------------------------------------------
// Poseidon parameters
#define AKP_MAX_T 16u
struct akp_poseidon {
    akp_ctx* ctx = nullptr;
    PoseidonDims dims{};
    std::vector<Fr> ark, mds;         // host copies, wire format
    Fr* d_ark = nullptr;              // wire format (export / conversion source)
    Fr* d_mds = nullptr;
    F29Pad* d_ark29 = nullptr;        // internal radix-2^29 form read by the kernels
    F29Pad* d_mds29 = nullptr;
    // sparse-partial-round form (poseidon_opt.hpp); null when not applicable (singular block / no partial rounds)
    F29Pad* d_arkmod29 = nullptr;
    F29Pad* d_mpre29 = nullptr;
    F29Pad* d_sparse29 = nullptr;
    F29Pad* d_sbox0_29 = nullptr;     // (round-0 key)^alpha per lane, see PoseidonConsts::sbox0
    bool scaled = false;              // sparse constants rescaled (poseidon_rescale_sparse)
    F29Pad* d_mpre_w29 = nullptr;     // lane-1 form for the one-lane-per-item kernels (poseidon_rescale_sparse_lane1)
    F29Pad* d_sparse_w29 = nullptr;
    F29Pad* d_ark_f29 = nullptr;      // full form for the t = 3 register kernels (poseidon_full_form)
    F29Pad* d_fmats_f29 = nullptr;
    F29Pad* d_sparse_f29 = nullptr;
    F29Pad* d_sbox0_f29 = nullptr;
};
static int32_t upload_f29(akp_ctx* ctx, const std::vector<Fr>& v, F29Pad** out) {
    Fr* tmp = nullptr;
    HIP_TRY(hipMalloc(&tmp, v.size() * sizeof(Fr)));
    hipError_t e = hipMemcpy(tmp, v.data(), v.size() * sizeof(Fr), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc(out, v.size() * sizeof(F29Pad));
    if (e == hipSuccess) {
        hipLaunchKernelGGL(poseidon_convert_params_kernel, dim3((unsigned)((v.size() + 63) / 64)), dim3(64), 0, ctx->stream, tmp, *out, v.size());
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    }
    (void)hipFree(tmp);
    if (e != hipSuccess) return fail(AKP_ERR_HIP, "uploading optimised Poseidon constants: %s", hipGetErrorString(e));
    return AKP_OK;
}

extern "C" void akp_poseidon_params_destroy(akp_poseidon* p);
extern "C" int32_t akp_poseidon_params_create(akp_ctx* ctx, uint32_t full_rounds, uint32_t partial_rounds, uint64_t alpha,
                                              uint32_t rate, uint32_t capacity, const uint64_t* ark, const uint64_t* mds,
                                              akp_poseidon** out) {
    if (!out || !ark || !mds) return fail(AKP_ERR_BAD_PARAMS, "akp_poseidon_params_create: NULL argument");
    const uint32_t t = rate + capacity;
    if (rate == 0 || t > AKP_MAX_T) return fail(AKP_ERR_BAD_PARAMS, "rate + capacity = %u unsupported (1 <= rate, t <= %u)", t, AKP_MAX_T);
    if (alpha == 0) return fail(AKP_ERR_BAD_PARAMS, "alpha must be >= 1");
    if (full_rounds % 2u) return fail(AKP_ERR_BAD_PARAMS, "full_rounds must be even");
    const size_t na = (size_t)(full_rounds + partial_rounds) * t, nm = (size_t)t * t;
    for (size_t i = 0; i < na; ++i)
        if (!fr_words_reduced(ark + 4 * i)) return fail(AKP_ERR_BAD_PARAMS, "ark[%zu] not reduced", i);
    for (size_t i = 0; i < nm; ++i)
        if (!fr_words_reduced(mds + 4 * i)) return fail(AKP_ERR_BAD_PARAMS, "mds[%zu] not reduced", i);
    akp_poseidon* p = new akp_poseidon();
    p->ctx = ctx;
    if (ctx) ++ctx->live_handles;
    p->dims = PoseidonDims{t, rate, capacity, full_rounds, partial_rounds, alpha};
    p->ark.resize(na);
    p->mds.resize(nm);
    for (size_t i = 0; i < na; ++i) p->ark[i] = fr_from_words(ark + 4 * i);
    for (size_t i = 0; i < nm; ++i) p->mds[i] = fr_from_words(mds + 4 * i);
    if (ctx) {
        hipError_t e = hipSetDevice(ctx->device);
        if (e == hipSuccess) e = hipMalloc(&p->d_ark, std::max<size_t>(na, 1) * sizeof(Fr));
        if (e == hipSuccess) e = hipMalloc(&p->d_mds, nm * sizeof(Fr));
        if (e == hipSuccess && na) e = hipMemcpy(p->d_ark, p->ark.data(), na * sizeof(Fr), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(p->d_mds, p->mds.data(), nm * sizeof(Fr), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMalloc(&p->d_ark29, std::max<size_t>(na, 1) * sizeof(F29Pad));
        if (e == hipSuccess) e = hipMalloc(&p->d_mds29, nm * sizeof(F29Pad));
        if (e == hipSuccess) {
            if (na) hipLaunchKernelGGL(poseidon_convert_params_kernel, dim3((unsigned)((na + 63) / 64)), dim3(64), 0, ctx->stream, p->d_ark, p->d_ark29, na);
            hipLaunchKernelGGL(poseidon_convert_params_kernel, dim3((unsigned)((nm + 63) / 64)), dim3(64), 0, ctx->stream, p->d_mds, p->d_mds29, nm);
            e = hipGetLastError();
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        }
        if (e != hipSuccess) {
            if (p->d_ark) (void)hipFree(p->d_ark);
            if (p->d_mds) (void)hipFree(p->d_mds);
            if (p->d_ark29) (void)hipFree(p->d_ark29);
            if (p->d_mds29) (void)hipFree(p->d_mds29);
            delete p;
            return fail(AKP_ERR_HIP, "uploading Poseidon parameters: %s", hipGetErrorString(e));
        }
        if (!getenv("AKP_POSEIDON_DENSE")) {
            PoseidonOpt opt = poseidon_optimize(t, full_rounds, partial_rounds, p->ark, p->mds);
            PoseidonOpt optw = opt;
            const bool rescale = opt.ok && !getenv("AKP_POSEIDON_NO_RESCALE");
            PoseidonFullForm ff;
            if (rescale && !getenv("AKP_POSEIDON_NO_FULL_FORM") && poseidon_full_form(opt, t, full_rounds, partial_rounds, alpha, p->mds, ff)) {
                int32_t rc = upload_f29(ctx, ff.ark, &p->d_ark_f29);
                if (!rc) rc = upload_f29(ctx, ff.fmats, &p->d_fmats_f29);
                if (!rc) rc = upload_f29(ctx, ff.sparse, &p->d_sparse_f29);
                if (!rc) rc = upload_f29(ctx, poseidon_sbox0(ff.ark, t, alpha), &p->d_sbox0_f29);  // full_rounds >= 2 here
                if (rc) {
                    akp_poseidon_params_destroy(p);
                    return rc;
                }
            }
            if (rescale) poseidon_rescale_sparse(opt, t, partial_rounds, alpha);
            p->scaled = opt.scaled;
            if (rescale && poseidon_rescale_sparse_lane1(optw, t, partial_rounds, alpha)) {
                int32_t rc = upload_f29(ctx, optw.mpre, &p->d_mpre_w29);
                if (!rc) rc = upload_f29(ctx, optw.sparse, &p->d_sparse_w29);
                if (rc) {
                    akp_poseidon_params_destroy(p);
                    return rc;
                }
            }
            if (opt.ok) {
                int32_t rc = upload_f29(ctx, opt.ark_mod, &p->d_arkmod29);
                if (!rc) rc = upload_f29(ctx, opt.mpre, &p->d_mpre29);
                if (!rc) rc = upload_f29(ctx, opt.sparse, &p->d_sparse29);
                if (!rc && full_rounds >= 2) rc = upload_f29(ctx, poseidon_sbox0(opt.ark_mod, t, alpha), &p->d_sbox0_29);
                if (rc) {
                    akp_poseidon_params_destroy(p);
                    return rc;
                }
            }
        }
        if (!p->d_sbox0_29 && full_rounds >= 2) {
            if (int32_t rc = upload_f29(ctx, poseidon_sbox0(p->ark, t, alpha), &p->d_sbox0_29)) {
                akp_poseidon_params_destroy(p);
                return rc;
            }
        }
    }
    *out = p;
    return AKP_OK;
}
extern "C" void akp_poseidon_params_destroy(akp_poseidon* p) {
    if (!p) return;
    if (p->ctx) (void)hipSetDevice(p->ctx->device);
    if (p->d_ark) (void)hipFree(p->d_ark);
    if (p->d_mds) (void)hipFree(p->d_mds);
    if (p->d_ark29) (void)hipFree(p->d_ark29);
    if (p->d_mds29) (void)hipFree(p->d_mds29);
    if (p->d_arkmod29) (void)hipFree(p->d_arkmod29);
    if (p->d_mpre29) (void)hipFree(p->d_mpre29);
    if (p->d_sparse29) (void)hipFree(p->d_sparse29);
    if (p->d_sbox0_29) (void)hipFree(p->d_sbox0_29);
    if (p->d_mpre_w29) (void)hipFree(p->d_mpre_w29);
    if (p->d_sparse_w29) (void)hipFree(p->d_sparse_w29);
    if (p->d_ark_f29) (void)hipFree(p->d_ark_f29);
    if (p->d_fmats_f29) (void)hipFree(p->d_fmats_f29);
    if (p->d_sparse_f29) (void)hipFree(p->d_sparse_f29);
    if (p->d_sbox0_f29) (void)hipFree(p->d_sbox0_f29);
    ctx_handle_released(p->ctx);
    delete p;
}
extern "C" int32_t akp_poseidon_params_dims(const akp_poseidon* p, uint32_t* full_rounds, uint32_t* partial_rounds,
                                            uint64_t* alpha, uint32_t* rate, uint32_t* capacity) {
    if (!p) return fail(AKP_ERR_BAD_PARAMS, "params is NULL");
    if (full_rounds) *full_rounds = p->dims.full_rounds;
    if (partial_rounds) *partial_rounds = p->dims.partial_rounds;
    if (alpha) *alpha = p->dims.alpha;
    if (rate) *rate = p->dims.rate;
    if (capacity) *capacity = p->dims.capacity;
    return AKP_OK;
}
extern "C" int32_t akp_poseidon_params_export(const akp_poseidon* p, uint64_t* ark, uint64_t* mds) {
    if (!p) return fail(AKP_ERR_BAD_PARAMS, "params is NULL");
    if (ark)
        for (size_t i = 0; i < p->ark.size(); ++i) fr_to_words(p->ark[i], ark + 4 * i);
    if (mds)
        for (size_t i = 0; i < p->mds.size(); ++i) fr_to_words(p->mds[i], mds + 4 * i);
    return AKP_OK;
}

// ---- default parameters: Grain LFSR (sponge/poseidon/grain_lfsr.rs:16-181) + Cauchy MDS
//      (sponge/poseidon/traits.rs:105-146), BLS12-381 Fr table (sponge/test.rs:13-31) -----------
namespace {
struct GrainLFSR {
    bool st[80];
    unsigned head = 0;
    unsigned prime_bits;
    GrainLFSR(bool sbox_inverse, unsigned prime_num_bits, unsigned state_len, unsigned rf, unsigned rp) : prime_bits(prime_num_bits) {
        memset(st, 0, sizeof st);
        st[1] = true;
        st[5] = sbox_inverse;
        auto put = [&](int lo, int hi, unsigned v) {
            for (int i = hi; i >= lo; --i) { st[i] = v & 1u; v >>= 1; }
        };
        put(6, 17, prime_num_bits);
        put(18, 29, state_len);
        put(30, 39, rf);
        put(40, 49, rp);
        for (int i = 50; i < 80; ++i) st[i] = true;
        for (int i = 0; i < 160; ++i) update();
    }
    bool update() {
        bool nb = st[(head + 62) % 80] ^ st[(head + 51) % 80] ^ st[(head + 38) % 80] ^ st[(head + 23) % 80] ^ st[(head + 13) % 80] ^ st[head];
        st[head] = nb;
        head = (head + 1) % 80;
        return nb;
    }
    bool next_bit() {  // get_bits :87-107: keep the second bit of a pair iff the first is 1
        bool b = update();
        while (!b) { update(); b = update(); }
        return update();
    }
    // prime_bits bits, most significant first -> canonical 256-bit integer (8 x u32 LE)
    void next_int(u32 (&v)[8]) {
        for (int i = 0; i < 8; ++i) v[i] = 0;
        for (unsigned i = 0; i < prime_bits; ++i) {
            const unsigned pos = prime_bits - 1 - i;
            if (next_bit()) v[pos >> 5] |= 1u << (pos & 31);
        }
    }
};
bool geq_p(const u32 (&v)[8]) {
    for (int i = 7; i >= 0; --i) {
        if (v[i] > fr_p_limb(i)) return true;
        if (v[i] < fr_p_limb(i)) return false;
    }
    return true;
}
void sub_p(u32 (&v)[8]) {
    u64 bw = 0;
    for (int i = 0; i < 8; ++i) {
        u64 d = (u64)v[i] - fr_p_limb(i) - bw;
        v[i] = (u32)d;
        bw = (d >> 32) & 1;
    }
}
Fr lfsr_rejection(GrainLFSR& g) {  // :109-134
    u32 v[8];
    do g.next_int(v); while (geq_p(v));
    Fr c;
    for (int i = 0; i < 8; ++i) c.l[i] = v[i];
    return fr_to_mont(c);
}
Fr lfsr_mod_p(GrainLFSR& g) {  // :136-160 (from_le_bytes_mod_order of a 255-bit value: < 3p)
    u32 v[8];
    g.next_int(v);
    while (geq_p(v)) sub_p(v);
    Fr c;
    for (int i = 0; i < 8; ++i) c.l[i] = v[i];
    return fr_to_mont(c);
}
struct DefaultEntry { unsigned rate, alpha, rf, rp, skip; };
const DefaultEntry kConstraints[7] = {{2, 17, 8, 31, 0}, {3, 5, 8, 56, 0}, {4, 5, 8, 56, 0}, {5, 5, 8, 57, 0},
                                      {6, 5, 8, 57, 0}, {7, 5, 8, 57, 0}, {8, 5, 8, 57, 0}};
const DefaultEntry kWeights[7] = {{2, 257, 8, 13, 0}, {3, 257, 8, 13, 0}, {4, 257, 8, 13, 0}, {5, 257, 8, 13, 0},
                                  {6, 257, 8, 13, 0}, {7, 257, 8, 13, 0}, {8, 257, 8, 13, 0}};
}  // namespace

extern "C" int32_t akp_poseidon_default_params(akp_ctx* ctx, uint32_t rate, int32_t optimized_for_weights, akp_poseidon** out) {
    const DefaultEntry* tab = optimized_for_weights ? kWeights : kConstraints;
    const DefaultEntry* e = nullptr;
    for (int i = 0; i < 7; ++i)
        if (tab[i].rate == rate) e = &tab[i];
    if (!e) return fail(AKP_ERR_BAD_PARAMS, "no default Poseidon parameters for rate %u (reference returns None)", rate);
    const unsigned t = rate + 1;
    GrainLFSR g(false, 255, t, e->rf, e->rp);
    std::vector<Fr> ark((size_t)(e->rf + e->rp) * t), mds((size_t)t * t), xs(t), ys(t);
    for (auto& a : ark) a = lfsr_rejection(g);
    for (unsigned s = 0; s < e->skip; ++s)
        for (unsigned i = 0; i < 2 * t; ++i) (void)lfsr_mod_p(g);
    for (auto& x : xs) x = lfsr_mod_p(g);
    for (auto& y : ys) y = lfsr_mod_p(g);
    for (unsigned i = 0; i < t; ++i)
        for (unsigned j = 0; j < t; ++j) mds[(size_t)i * t + j] = fr_inv(fr_add(xs[i], ys[j]));
    std::vector<uint64_t> aw(ark.size() * 4), mw(mds.size() * 4);
    for (size_t i = 0; i < ark.size(); ++i) fr_to_words(ark[i], &aw[4 * i]);
    for (size_t i = 0; i < mds.size(); ++i) fr_to_words(mds[i], &mw[4 * i]);
    return akp_poseidon_params_create(ctx, e->rf, e->rp, e->alpha, rate, 1, aw.data(), mw.data(), out);
}

// ------------------------------------------------------------------------------------------
// Poseidon launches
static inline unsigned poseidon_block(u32 t) {
    if (const char* e = getenv("AKP_POSEIDON_FILE_BLOCK")) { const unsigned b = (unsigned)atoi(e); if (b == 64 || b == 128 || b == 256) return (36u * t * b <= 65536u) ? b : 64u; }
    // 36*t*B bytes of LDS per block (<= 64 KiB): 256 is fastest while four blocks still fit a CU; from t = 9 on the file
    // is what limits the waves per CU and the finer 64-lane granularity fits one more (tools/gpu_file_block.sh)
    return t <= 7 ? 256u : (t == 8 ? 128u : 64u);
}

// LDS bytes of the generic kernel: t elements of 9 dwords per lane
static inline size_t poseidon_lds(u32 t, unsigned B) { return (size_t)t * 9 * 4 * B; }

static inline PoseidonConsts t3_consts(const akp_poseidon* p) {
    if (p->d_sparse29) return PoseidonConsts{p->d_arkmod29, p->d_mds29, p->d_mpre29, p->d_sparse29, p->d_sbox0_29, p->scaled ? 1u : 0u};
    return PoseidonConsts{p->d_ark29, p->d_mds29, nullptr, nullptr, p->d_sbox0_29, 0u};
}
// constants for the one-lane-per-item kernels (LDS-file kernels, t = 3 register kernels): the full form when it exists,
// else the lane-1 form, else what the wave-per-lane kernels use
static inline PoseidonConsts file_consts(const akp_poseidon* p) {
    if (p->d_sparse_f29) return PoseidonConsts{p->d_ark_f29, p->d_fmats_f29, nullptr, p->d_sparse_f29, p->d_sbox0_f29, 3u};
    if (p->d_sparse_w29) return PoseidonConsts{p->d_arkmod29, p->d_mds29, p->d_mpre_w29, p->d_sparse_w29, p->d_sbox0_29, 2u};
    return t3_consts(p);
}
static inline PoseidonConsts t3_reg_consts(const akp_poseidon* p) { return file_consts(p); }
#define AKP_MAX_BATCH ((size_t)1 << 36)  /* grid.x = n / 256 must stay below 2^31 */
// AKP_POSEIDON_COOP_MAX: largest t = 3 batch routed to the wave-per-lane latency kernels (0 disables them)
static size_t coop_max_items() {
    static const size_t v = [] {
        const char* e = getenv("AKP_POSEIDON_COOP_MAX");
        return (e && *e) ? (size_t)strtoull(e, nullptr, 10) : ((size_t)1 << 15);
    }();
    return v;
}
// generic (t != 3) Poseidon kernels: batches up to AKP_POSEIDON_GENERIC_COOP_MAX (default 2^15) use one wave per state
// lane (2-3x lower latency), larger ones the LDS-file kernel (one lane per item, up to 1.7x the throughput)
static bool generic_coop(size_t n) {
    static const size_t coop_max = [] {
        const char* e = getenv("AKP_POSEIDON_GENERIC_COOP_MAX");
        return (e && *e) ? (size_t)strtoull(e, nullptr, 10) : ((size_t)1 << 15);
    }();
    return n <= coop_max;
}
// t = 4 .. 9 (the default rate-3 .. rate-8 instances): register-resident kernels for large batches when the parameter set has
// the full form or the lane-1 form (AKP_POSEIDON_NO_REG_T=1 keeps the LDS-file kernels: the A/B arm)
static bool reg_t_kernel(const akp_poseidon* p, size_t n, const PoseidonConsts& c) {
    static const bool enabled = !getenv("AKP_POSEIDON_NO_REG_T");
    return enabled && (p->dims.t >= 4 && p->dims.t <= 9) && !generic_coop(n) && (c.scaled == 3u || c.scaled == 2u) && c.sparse != nullptr;
}
template <u32 T>
static void launch_reg_permute(const akp_poseidon* p, const PoseidonConsts& c, Fr* d_states, size_t n, hipStream_t s) {
    const dim3 grid((unsigned)((n + 255) / 256));
    if (c.scaled == 3u) hipLaunchKernelGGL((poseidon_permute_reg_kernel<T, true>), grid, dim3(256), 0, s, p->dims, c, d_states, n);
    else hipLaunchKernelGGL((poseidon_permute_reg_kernel<T, false>), grid, dim3(256), 0, s, p->dims, c, d_states, n);
}
template <u32 T>
static void launch_reg_crh(const akp_poseidon* p, const PoseidonConsts& c, const Fr* in0, const Fr* in1, size_t k, Fr* d_out, size_t n, hipStream_t s) {
    const dim3 grid((unsigned)((n + 255) / 256));
    if (c.scaled == 3u) hipLaunchKernelGGL((poseidon_crh_reg_kernel<T, true>), grid, dim3(256), 0, s, p->dims, c, in0, in1, k, d_out, n);
    else hipLaunchKernelGGL((poseidon_crh_reg_kernel<T, false>), grid, dim3(256), 0, s, p->dims, c, in0, in1, k, d_out, n);
}
static size_t coop_lds(u32 t) {
    const size_t bytes = (size_t)2 * t * 9 * 64 * sizeof(u32);
    if (bytes > 65536) {  // t = 15, 16: above the default 64 KiB of dynamic LDS per workgroup (gfx950 has 160 KiB per CU);
        // set on every such launch: the attribute belongs to the current device's copy of the kernel
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(poseidon_permute_coop_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(poseidon_crh_coop_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    }
    return bytes;
}
// Resident waves per SIMD of the t = 3 register kernels.  Their 78-81 VGPRs would allow six; a dynamic-LDS request of
// 160 KiB / w per workgroup caps it at w workgroups per CU (one wave of each per SIMD).  Four measured 0.5-1 % faster than
// six at every batch size (profiles/r02_s46) and makes the 4096 workgroups of a 2^20-state launch exactly four rounds.
// AKP_POSEIDON_T3_WAVES (3..6, 6 = no cap) for A/B runs.
static unsigned t3_lds_cap() {
    static const unsigned w = env_u32("AKP_POSEIDON_T3_WAVES", 4, 3, 6);
    return w >= 6 ? 0u : ((160u * 1024u / w) & ~1023u);
}
static int32_t launch_permute(akp_poseidon* p, Fr* d_states, size_t n, hipStream_t s) {
    if (n == 0) return AKP_OK;
    if (n > AKP_MAX_BATCH) return fail(AKP_ERR_BAD_PARAMS, "batch of %zu items exceeds the supported 2^36", n);
    if (p->dims.t == 3 && n > coop_max_items()) {
        const PoseidonConsts c = t3_reg_consts(p);
        if (c.scaled == 3u) hipLaunchKernelGGL(poseidon_permute_t3_kernel<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), t3_lds_cap(), s, p->dims, c, d_states, n);
        else hipLaunchKernelGGL(poseidon_permute_t3_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), t3_lds_cap(), s, p->dims, c, d_states, n);
        HIP_TRY(hipGetLastError());
        return AKP_OK;
    }
    if (reg_t_kernel(p, n, file_consts(p))) {
        switch (p->dims.t) {
            case 4: launch_reg_permute<4>(p, file_consts(p), d_states, n, s); break;
            case 5: launch_reg_permute<5>(p, file_consts(p), d_states, n, s); break;
            case 6: launch_reg_permute<6>(p, file_consts(p), d_states, n, s); break;
            case 7: launch_reg_permute<7>(p, file_consts(p), d_states, n, s); break;
            case 8: launch_reg_permute<8>(p, file_consts(p), d_states, n, s); break;
            default: launch_reg_permute<9>(p, file_consts(p), d_states, n, s); break;
        }
        HIP_TRY(hipGetLastError());
        return AKP_OK;
    }
    if (p->dims.t == 3 || generic_coop(n)) {  // t = 3 reaches this point only for small batches (sponge steps, few states)
        hipLaunchKernelGGL(poseidon_permute_coop_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64 * p->dims.t), coop_lds(p->dims.t), s, p->dims, t3_consts(p), d_states, n);
        HIP_TRY(hipGetLastError());
        return AKP_OK;
    }
    const unsigned B = poseidon_block(p->dims.t);
    const size_t lds = poseidon_lds(p->dims.t, B);
    const unsigned grid = (unsigned)((n + B - 1) / B);
    if (B == 256) hipLaunchKernelGGL(poseidon_permute_kernel<256>, dim3(grid), dim3(B), lds, s, p->dims, file_consts(p), d_states, n);
    else if (B == 128) hipLaunchKernelGGL(poseidon_permute_kernel<128>, dim3(grid), dim3(B), lds, s, p->dims, file_consts(p), d_states, n);
    else hipLaunchKernelGGL(poseidon_permute_kernel<64>, dim3(grid), dim3(B), lds, s, p->dims, file_consts(p), d_states, n);
    HIP_TRY(hipGetLastError());
    return AKP_OK;
}
static int32_t launch_crh(akp_poseidon* p, const Fr* in0, const Fr* in1, size_t k, Fr* d_out, size_t n, hipStream_t s) {
    if (n == 0) return AKP_OK;
    if (n > AKP_MAX_BATCH) return fail(AKP_ERR_BAD_PARAMS, "batch of %zu items exceeds the supported 2^36", n);
    if (p->dims.t == 3 && n > coop_max_items()) {
        const PoseidonConsts c = t3_reg_consts(p);
        if (c.scaled == 3u) hipLaunchKernelGGL(poseidon_crh_t3_kernel<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), t3_lds_cap(), s, p->dims, c, in0, in1, k, d_out, n);
        else hipLaunchKernelGGL(poseidon_crh_t3_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), t3_lds_cap(), s, p->dims, c, in0, in1, k, d_out, n);
        HIP_TRY(hipGetLastError());
        return AKP_OK;
    }
    if (reg_t_kernel(p, n, file_consts(p))) {
        switch (p->dims.t) {
            case 4: launch_reg_crh<4>(p, file_consts(p), in0, in1, k, d_out, n, s); break;
            case 5: launch_reg_crh<5>(p, file_consts(p), in0, in1, k, d_out, n, s); break;
            case 6: launch_reg_crh<6>(p, file_consts(p), in0, in1, k, d_out, n, s); break;
            case 7: launch_reg_crh<7>(p, file_consts(p), in0, in1, k, d_out, n, s); break;
            case 8: launch_reg_crh<8>(p, file_consts(p), in0, in1, k, d_out, n, s); break;
            default: launch_reg_crh<9>(p, file_consts(p), in0, in1, k, d_out, n, s); break;
        }
        HIP_TRY(hipGetLastError());
        return AKP_OK;
    }
    // small batches (tree tops, single sponges) are bound by the latency of one permutation: one wave per state lane
    if (p->dims.t == 3 || generic_coop(n)) {
        hipLaunchKernelGGL(poseidon_crh_coop_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64 * p->dims.t), coop_lds(p->dims.t), s, p->dims, t3_consts(p), in0, in1, k, d_out, n);
        HIP_TRY(hipGetLastError());
        return AKP_OK;
    }
    const unsigned B = poseidon_block(p->dims.t);
    const size_t lds = poseidon_lds(p->dims.t, B);
    const unsigned grid = (unsigned)((n + B - 1) / B);
    if (B == 256) hipLaunchKernelGGL(poseidon_crh_kernel<256>, dim3(grid), dim3(B), lds, s, p->dims, file_consts(p), in0, in1, k, d_out, n);
    else if (B == 128) hipLaunchKernelGGL(poseidon_crh_kernel<128>, dim3(grid), dim3(B), lds, s, p->dims, file_consts(p), in0, in1, k, d_out, n);
    else hipLaunchKernelGGL(poseidon_crh_kernel<64>, dim3(grid), dim3(B), lds, s, p->dims, file_consts(p), in0, in1, k, d_out, n);
    HIP_TRY(hipGetLastError());
    return AKP_OK;
}
// which kernel a batch of n items is routed to (the rule of launch_permute / launch_crh), so that a parity probe can
// say which kernel it certified.  The returned string is static.
extern "C" const char* akp_poseidon_kernel_for(const akp_poseidon* p, size_t n, int32_t crh) {
    if (!p || !p->ctx) return "none";
    if (p->dims.t == 3 && n > coop_max_items()) {
        const bool ff = t3_reg_consts(p).scaled == 3u;
        if (crh) return ff ? "poseidon_crh_t3_kernel<true>" : "poseidon_crh_t3_kernel<false>";
        return ff ? "poseidon_permute_t3_kernel<true>" : "poseidon_permute_t3_kernel<false>";
    }
    if (reg_t_kernel(p, n, file_consts(p))) return crh ? "poseidon_crh_reg_kernel" : "poseidon_permute_reg_kernel";
    if (p->dims.t == 3 || generic_coop(n)) return crh ? "poseidon_crh_coop_kernel" : "poseidon_permute_coop_kernel";
    return crh ? "poseidon_crh_kernel" : "poseidon_permute_kernel";
}
#define NEED_DEV(p, what)                                                                                  \
    do {                                                                                                   \
        if (!(p)) return fail(AKP_ERR_BAD_PARAMS, what ": params is NULL");                                \
        if (!(p)->ctx) return fail(AKP_ERR_HIP, what ": parameter handle has no device context (no CPU fallback)"); \
        if ((p)->ctx->dead) return fail(AKP_ERR_BAD_PARAMS, what ": the context of this handle was destroyed");  \
        HIP_TRY(hipSetDevice((p)->ctx->device));                                                           \
    } while (0)

extern "C" int32_t akp_poseidon_permute_batch_dev(akp_poseidon* p, uint64_t* d_states, size_t n, void* stream) {
    NEED_DEV(p, "akp_poseidon_permute_batch_dev");
    return launch_permute(p, reinterpret_cast<Fr*>(d_states), n, pick_stream(p->ctx, stream));
}
extern "C" int32_t akp_poseidon_permute_batch(akp_poseidon* p, uint64_t* states, size_t n) {
    NEED_DEV(p, "akp_poseidon_permute_batch");
    if (n == 0) return AKP_OK;
    if (!states) return fail(AKP_ERR_BAD_PARAMS, "states is NULL");
    const HostIn in[1] = {{states, p->dims.t * sizeof(Fr), SCR_A}};
    return pipelined_batch(p->ctx, n, in, 1, states, p->dims.t * sizeof(Fr), -1,
                           [&](void* const* di, void*, size_t cnt, hipStream_t s) -> int32_t { return launch_permute(p, (Fr*)di[0], cnt, s); });
}
extern "C" int32_t akp_poseidon_crh_batch_dev(akp_poseidon* p, const uint64_t* d_inputs, size_t n, size_t k, uint64_t* d_out, void* stream) {
    NEED_DEV(p, "akp_poseidon_crh_batch_dev");
    return launch_crh(p, (const Fr*)d_inputs, nullptr, k, (Fr*)d_out, n, pick_stream(p->ctx, stream));
}
extern "C" int32_t akp_poseidon_crh_batch(akp_poseidon* p, const uint64_t* inputs, size_t n, size_t k, uint64_t* out) {
    NEED_DEV(p, "akp_poseidon_crh_batch");
    if (n == 0) return AKP_OK;
    if (!out || (!inputs && k)) return fail(AKP_ERR_BAD_PARAMS, "NULL buffer");
    const HostIn in[1] = {{inputs, k * sizeof(Fr), SCR_A}};
    return pipelined_batch(p->ctx, n, in, 1, out, sizeof(Fr), SCR_B, [&](void* const* di, void* dout, size_t cnt, hipStream_t s) -> int32_t {
        return launch_crh(p, (const Fr*)di[0], nullptr, k, (Fr*)dout, cnt, s);
    });
}
extern "C" int32_t akp_poseidon_two_to_one_batch_dev(akp_poseidon* p, const uint64_t* d_left, const uint64_t* d_right, size_t n,
                                                     uint64_t* d_out, void* stream) {
    NEED_DEV(p, "akp_poseidon_two_to_one_batch_dev");
    return launch_crh(p, (const Fr*)d_left, (const Fr*)d_right, 2, (Fr*)d_out, n, pick_stream(p->ctx, stream));
}
extern "C" int32_t akp_poseidon_two_to_one_batch(akp_poseidon* p, const uint64_t* left, const uint64_t* right, size_t n, uint64_t* out) {
    NEED_DEV(p, "akp_poseidon_two_to_one_batch");
    if (n == 0) return AKP_OK;
    if (!left || !right || !out) return fail(AKP_ERR_BAD_PARAMS, "NULL buffer");
    const HostIn in[2] = {{left, sizeof(Fr), SCR_A}, {right, sizeof(Fr), SCR_B}};
    return pipelined_batch(p->ctx, n, in, 2, out, sizeof(Fr), SCR_C, [&](void* const* di, void* dout, size_t cnt, hipStream_t s) -> int32_t {
        return launch_crh(p, (const Fr*)di[0], (const Fr*)di[1], 2, (Fr*)dout, cnt, s);
    });
}

// ------------------------------------------------------------------------------------------
// batched duplex sponge: device state, host-side DuplexSpongeMode bookkeeping
__global__ void sponge_add_kernel(Fr* state, u32 t, u32 lane0, const Fr* elems, size_t k, size_t e0, u32 count, size_t batch) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= batch * count) return;
    const size_t b = i / count, j = i % count;
    Fr* dst = state + b * t + lane0 + j;
    store_fr_global(dst, fr_add(load_fr_global(dst), load_fr_global(elems + b * k + e0 + j)));
}
__global__ void sponge_copy_out_kernel(const Fr* state, u32 t, u32 lane0, Fr* out, size_t n_out, size_t o0, u32 count, size_t batch) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= batch * count) return;
    const size_t b = i / count, j = i % count;
    store_fr_global(out + b * n_out + o0 + j, load_fr_global(state + b * t + lane0 + j));
}
struct akp_sponge {
    akp_poseidon* p = nullptr;
    size_t batch = 0;
    Fr* d_state = nullptr;
    Fr* d_io = nullptr;
    size_t io_elems = 0;
    int mode = 0;      // 0 absorbing, 1 squeezing (sponge/mod.rs:195-206)
    u32 index = 0;
};
extern "C" int32_t akp_sponge_create(akp_poseidon* p, size_t batch, akp_sponge** out) {
    NEED_DEV(p, "akp_sponge_create");
    if (!out || batch == 0) return fail(AKP_ERR_BAD_PARAMS, "akp_sponge_create: bad argument");
    akp_sponge* s = new akp_sponge();
    s->p = p;
    s->batch = batch;
    hipError_t e = hipMalloc(&s->d_state, batch * p->dims.t * sizeof(Fr));
    if (e == hipSuccess) e = hipMemset(s->d_state, 0, batch * p->dims.t * sizeof(Fr));  // new(): all-zero state :223-234
    if (e != hipSuccess) {
        if (s->d_state) (void)hipFree(s->d_state);
        delete s;
        return fail(AKP_ERR_HIP, "akp_sponge_create: %s", hipGetErrorString(e));
    }
    *out = s;
    return AKP_OK;
}
extern "C" void akp_sponge_destroy(akp_sponge* s) {
    if (!s) return;
    (void)hipSetDevice(s->p->ctx->device);
    (void)hipDeviceSynchronize();
    if (s->d_state) (void)hipFree(s->d_state);
    if (s->d_io) (void)hipFree(s->d_io);
    delete s;
}
static int32_t sponge_io(akp_sponge* s, size_t elems) {
    if (s->io_elems < elems) {
        if (s->d_io) {
            HIP_TRY(hipDeviceSynchronize());
            HIP_TRY(hipFree(s->d_io));
            s->d_io = nullptr;
        }
        HIP_TRY(hipMalloc(&s->d_io, elems * sizeof(Fr)));
        s->io_elems = elems;
    }
    return AKP_OK;
}
static int32_t sponge_permute(akp_sponge* s) { return launch_permute(s->p, s->d_state, s->batch, s->p->ctx->stream); }
// absorb_internal (sponge/poseidon/mod.rs:124-153)
static int32_t sponge_absorb_internal(akp_sponge* s, u32 idx, size_t k) {
    const PoseidonDims& D = s->p->dims;
    hipStream_t st = s->p->ctx->stream;
    size_t e0 = 0, remaining = k;
    for (;;) {
        const bool last = idx + remaining <= D.rate;
        const u32 count = last ? (u32)remaining : D.rate - idx;
        if (count) {
            const size_t work = s->batch * count;
            hipLaunchKernelGGL(sponge_add_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, st, s->d_state, D.t,
                               D.capacity + idx, s->d_io, k, e0, count, s->batch);
            HIP_TRY(hipGetLastError());
        }
        if (last) {
            s->mode = 0;
            s->index = idx + (u32)remaining;
            return AKP_OK;
        }
        if (int32_t rc = sponge_permute(s)) return rc;
        e0 += count;
        remaining -= count;
        idx = 0;
    }
}
extern "C" int32_t akp_sponge_absorb(akp_sponge* s, const uint64_t* elems, size_t k) {
    if (!s) return fail(AKP_ERR_BAD_PARAMS, "sponge is NULL");
    if (k == 0) return AKP_OK;  // :238-240
    if (!elems) return fail(AKP_ERR_BAD_PARAMS, "elems is NULL");
    HIP_TRY(hipSetDevice(s->p->ctx->device));
    if (int32_t rc = sponge_io(s, s->batch * k)) return rc;
    hipStream_t st = s->p->ctx->stream;
    HIP_TRY(hipMemcpyAsync(s->d_io, elems, s->batch * k * sizeof(Fr), hipMemcpyHostToDevice, st));
    int32_t rc;
    if (s->mode == 0) {  // :243-250
        u32 idx = s->index;
        if (idx == s->p->dims.rate) {
            if ((rc = sponge_permute(s))) return rc;
            idx = 0;
        }
        rc = sponge_absorb_internal(s, idx, k);
    } else {  // :251-255 no permutation between squeeze and absorb
        rc = sponge_absorb_internal(s, 0, k);
    }
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(st));
    return AKP_OK;
}
// squeeze_internal (sponge/poseidon/mod.rs:156-186)
static int32_t sponge_squeeze_internal(akp_sponge* s, u32 idx, size_t n_out) {
    const PoseidonDims& D = s->p->dims;
    hipStream_t st = s->p->ctx->stream;
    size_t o0 = 0, remaining = n_out;
    for (;;) {
        const bool last = idx + remaining <= D.rate;
        const u32 count = last ? (u32)remaining : D.rate - idx;
        if (count) {
            const size_t work = s->batch * count;
            hipLaunchKernelGGL(sponge_copy_out_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, st, s->d_state, D.t,
                               D.capacity + idx, s->d_io, n_out, o0, count, s->batch);
            HIP_TRY(hipGetLastError());
        }
        if (last) {
            s->mode = 1;
            s->index = idx + (u32)remaining;
            return AKP_OK;
        }
        o0 += count;
        remaining -= count;
        if (remaining != 0)
            if (int32_t rc = sponge_permute(s)) return rc;
        idx = 0;
    }
}
extern "C" int32_t akp_sponge_squeeze(akp_sponge* s, uint64_t* out, size_t n_out) {
    if (!s) return fail(AKP_ERR_BAD_PARAMS, "sponge is NULL");
    if (!out && n_out) return fail(AKP_ERR_BAD_PARAMS, "out is NULL");
    HIP_TRY(hipSetDevice(s->p->ctx->device));
    if (int32_t rc = sponge_io(s, s->batch * std::max<size_t>(n_out, 1))) return rc;
    hipStream_t st = s->p->ctx->stream;
    int32_t rc;
    if (s->mode == 0) {  // :331-334 (permutes even when n_out == 0)
        if ((rc = sponge_permute(s))) return rc;
        rc = sponge_squeeze_internal(s, 0, n_out);
    } else {  // :335-341
        u32 idx = s->index;
        if (idx == s->p->dims.rate) {
            if ((rc = sponge_permute(s))) return rc;
            idx = 0;
        }
        rc = sponge_squeeze_internal(s, idx, n_out);
    }
    if (rc) return rc;
    if (n_out) HIP_TRY(hipMemcpyAsync(out, s->d_io, s->batch * n_out * sizeof(Fr), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return AKP_OK;
}
extern "C" int32_t akp_sponge_get_state(akp_sponge* s, uint64_t* state, int32_t* mode, uint32_t* index) {
    if (!s) return fail(AKP_ERR_BAD_PARAMS, "sponge is NULL");
    HIP_TRY(hipSetDevice(s->p->ctx->device));
    if (state) {
        HIP_TRY(hipStreamSynchronize(s->p->ctx->stream));
        HIP_TRY(hipMemcpy(state, s->d_state, s->batch * s->p->dims.t * sizeof(Fr), hipMemcpyDeviceToHost));
    }
    if (mode) *mode = s->mode;
    if (index) *index = s->index;
    return AKP_OK;
}
extern "C" int32_t akp_sponge_set_state(akp_sponge* s, const uint64_t* state, int32_t mode, uint32_t index) {
    if (!s || !state) return fail(AKP_ERR_BAD_PARAMS, "NULL argument");
    if ((mode != 0 && mode != 1) || index > s->p->dims.rate) return fail(AKP_ERR_BAD_PARAMS, "bad duplex mode");
    HIP_TRY(hipSetDevice(s->p->ctx->device));
    HIP_TRY(hipStreamSynchronize(s->p->ctx->stream));
    HIP_TRY(hipMemcpy(s->d_state, state, s->batch * s->p->dims.t * sizeof(Fr), hipMemcpyHostToDevice));
    s->mode = mode;
    s->index = index;
    return AKP_OK;
}

// ------------------------------------------------------------------------------------------
// Pedersen / Bowe-Hopwood
// largest precomputed table, bytes: the digit width / chunk grouping is reduced until the table fits.  Pedersen: 320 MiB
// admits 16-bit digits for the 4 x 256 window of BASELINE config 4 (64 steps x 2^15 entries x 128 B = 268 MB; measured on
// MI355X, 2^20 x 128 B: 15 bits / 145 MB 3.37 ms, 16 bits 3.25 ms, 17 bits / 512 MB 3.28 ms -- past the 256 MiB Infinity
// Cache the gather costs what the shorter sum saves).  Bowe-Hopwood: 256 MiB admits groups of five chunks for the 63 x 9
// window of config 5 (113 groups x 2^14 entries x 128 B = 237 MB, of which the 64-byte inputs of a tree touch the first
// 73 MB; 2.16 -> 1.78 ms per 2^20 two-to-one hashes against groups of four).  AKP_TE_TABLE_MB overrides both (A/B runs).
static size_t te_table_cap(bool bowe_hopwood = false) {
    static const u32 forced = env_u32("AKP_TE_TABLE_MB", 0, 0, 8192);
    return (size_t)(forced ? forced : (bowe_hopwood ? 256u : 320u)) << 20;
}
struct akp_te_params {
    akp_ctx* ctx = nullptr;
    int kind = 0;
    u32 W = 0, N = 0;
    u32 n_gen = 0;             // W * N flat generators
    u32 digit_bits = 0;        // Pedersen: table digit width D (1..8)
    u32 group = 1;             // Bowe-Hopwood: chunks per table step (1..4)
    TeEntry* d_lut = nullptr;    // Pedersen: [ceil(n_gen/D)][2^D] (signed-subset table: [ceil(n_gen/D)][2^(D-1)]); BH: group table
    TeEntry* d_lut1 = nullptr;  // BH: single-chunk table [n_gen][4]; Pedersen signed-subset: cprefix [n_digits + 1]
    TeEntry* d_tail = nullptr;  // BH: sum of G[c] over the zero-padded tail chunks [tail_from, tail_to) of the last compress shape
    u32 tail_from = 0, tail_to = 0;
    bool signed_subset = false;  // Pedersen: d_lut holds the signed-subset table (te_kernels.hpp), d_lut1 its constants
};
// Pedersen arithmetic (subset-sum tables over W * N generators): the plain hash and the one composed with TECompressor
static inline bool te_is_pedersen(const akp_te_params* p) { return p->kind == AKP_TE_PEDERSEN || p->kind == AKP_TE_PEDERSEN_X; }
static inline u32 te_fe_per_digest(const akp_te_params* p) { return p->kind == AKP_TE_PEDERSEN ? 2u : 1u; }
static inline size_t te_input_bits(const akp_te_params* p) {  // max message bits before the reference panics
    return te_is_pedersen(p) ? (size_t)p->W * p->N : (size_t)p->W * p->N * 3;
}

extern "C" int32_t akp_te_params_create(akp_ctx* ctx, int32_t kind, uint32_t W, uint32_t N, const uint64_t* gens, akp_te_params** out) {
    if (!ctx) return fail(AKP_ERR_HIP, "akp_te_params_create: a device context is required (tables are built on the GPU)");
    if (!out || !gens) return fail(AKP_ERR_BAD_PARAMS, "NULL argument");
    if (kind != AKP_TE_PEDERSEN && kind != AKP_TE_BOWE_HOPWOOD && kind != AKP_TE_PEDERSEN_X) return fail(AKP_ERR_BAD_PARAMS, "unknown kind %d", kind);
    if (W == 0 || N == 0) return fail(AKP_ERR_BAD_PARAMS, "empty window");
    if (kind == AKP_TE_BOWE_HOPWOOD && W > 63) return fail(AKP_ERR_BAD_PARAMS, "Bowe-Hopwood window size %u > 63 (bowe_hopwood/mod.rs:81-101)", W);
    const size_t n_gen = (size_t)W * N;
    if (n_gen > (1u << 22)) return fail(AKP_ERR_BAD_PARAMS, "window %ux%u too large", W, N);
    for (size_t i = 0; i < 2 * n_gen; ++i)
        if (!fr_words_reduced(gens + 4 * i)) return fail(AKP_ERR_BAD_PARAMS, "generator coordinate %zu not reduced", i);
    HIP_TRY(hipSetDevice(ctx->device));
    akp_te_params* p = new akp_te_params();
    p->ctx = ctx; p->kind = kind; p->W = W; p->N = N; p->n_gen = (u32)n_gen;
    ++ctx->live_handles;
    Fr* d_g = nullptr;
    hipError_t e = hipMalloc(&d_g, n_gen * 2 * sizeof(Fr));
    if (e == hipSuccess) e = hipMemcpy(d_g, gens, n_gen * 2 * sizeof(Fr), hipMemcpyHostToDevice);
    if (kind == AKP_TE_PEDERSEN || kind == AKP_TE_PEDERSEN_X) {
        // Signed-subset table (te_kernels.hpp): needs every generator in the prime-order subgroup (checked on the device: 2 (G/2) == G),
        // stores 2^(D-1) entries per digit.  Default D = 15: 4x256 is 69 steps over a 163 MB table.  Measured on MI355X, 2^20 x 128 B
        // (profiles/r02_s9): signed D = 13 / 14 / 15: 2.70 / 2.75 / 2.88e8 hashes/s; plain table D = 13: 2.56e8.
        // AKP_PEDERSEN_PLAIN=1 keeps the plain table (the A/B arm, and the fallback for generators outside the subgroup).
        NielsPad* d_half = nullptr;
        u32* d_bad = nullptr;
        u32 bad = 1;
        if (!getenv("AKP_PEDERSEN_PLAIN")) {
            if (e == hipSuccess) e = hipMalloc(&d_half, n_gen * sizeof(NielsPad));
            if (e == hipSuccess) e = hipMalloc(&d_bad, sizeof(u32));
            if (e == hipSuccess) e = hipMemsetAsync(d_bad, 0, sizeof(u32), ctx->stream);
            if (e == hipSuccess) {
                hipLaunchKernelGGL(te_halve_generators, dim3((unsigned)((n_gen + 63) / 64)), dim3(64), 0, ctx->stream, d_g, (u32)n_gen, d_half, d_bad);
                e = hipGetLastError();
            }
            if (e == hipSuccess) e = hipMemcpyAsync(&bad, d_bad, sizeof(u32), hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        }
        if (e == hipSuccess && bad == 0) {
            u32 D = env_u32("AKP_PEDERSEN_DIGIT_BITS", 16, 2, 17);
            while (D > 2 && ((n_gen + D - 1) / D) * ((size_t)1 << (D - 1)) * sizeof(TeEntry) > te_table_cap()) --D;
            size_t n_digits = (n_gen + D - 1) / D, entries = n_digits << (D - 1);
            e = hipMalloc(&p->d_lut, entries * sizeof(TeEntry));
            while (e == hipErrorOutOfMemory && D > 8) {  // a crowded device: a narrower digit needs half the table
                (void)hipGetLastError();
                --D;
                n_digits = (n_gen + D - 1) / D;
                entries = n_digits << (D - 1);
                e = hipMalloc(&p->d_lut, entries * sizeof(TeEntry));
            }
            p->digit_bits = D;
            p->signed_subset = true;
            if (e == hipSuccess) e = hipMalloc(&p->d_lut1, (n_digits + 1) * sizeof(TeEntry));
            if (e == hipSuccess) {
                hipLaunchKernelGGL(te_build_pedersen_slut, dim3((unsigned)((entries + 63) / 64)), dim3(64), 0, ctx->stream, d_half, (u32)n_gen, D, (u32)entries, p->d_lut);
                hipLaunchKernelGGL(te_build_pedersen_cprefix, dim3((unsigned)((n_digits + 1 + 63) / 64)), dim3(64), 0, ctx->stream, d_half, (u32)n_gen, D, (u32)n_digits, p->d_lut1);
                e = hipGetLastError();
            }
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        } else if (e == hipSuccess) {
            // plain table.  digit width: 13 bits (4x256: 79 steps, 93 MB table, served from the 256 MB Infinity Cache) unless the table
            // would exceed 192 MB; AKP_PEDERSEN_DIGIT_BITS overrides (1..14).  Measured 2^20 x 128 B on MI355X:
            // D = 4: 73 M/s, 8: 151, 10: 177, 12: 199, 13: 209, 14: 220 (175 MB table; round 2: 13 beats 14, profiles/r02_s8).
            u32 D = env_u32("AKP_PEDERSEN_DIGIT_BITS", 13, 1, 14);
            while (D > 1 && ((n_gen + D - 1) / D) * ((size_t)1 << D) * sizeof(TeEntry) > te_table_cap()) --D;
            p->digit_bits = D;
            const size_t entries = ((n_gen + D - 1) / D) << D;
            e = hipMalloc(&p->d_lut, entries * sizeof(TeEntry));
            if (e == hipSuccess) {
                hipLaunchKernelGGL(te_build_pedersen_lut, dim3((unsigned)((entries + 63) / 64)), dim3(64), 0, ctx->stream, d_g, (u32)n_gen, D, (u32)entries, p->d_lut);
                e = hipGetLastError();
            }
        }
        if (d_half) (void)hipFree(d_half);
        if (d_bad) (void)hipFree(d_bad);
    } else {
        u32 G = env_u32("AKP_BH_GROUP", 5, 1, 5);
        while (G > 1 && (n_gen < G || (n_gen / G) * ((size_t)1 << (3 * G - 1)) * sizeof(TeEntry) > te_table_cap(true))) --G;
        p->group = G;
        if (e == hipSuccess) e = hipMalloc(&p->d_lut1, n_gen * 4 * sizeof(TeEntry));
        if (e == hipSuccess) {
            hipLaunchKernelGGL(te_build_bh_lut, dim3((unsigned)((n_gen * 4 + 63) / 64)), dim3(64), 0, ctx->stream, d_g, (u32)n_gen, p->d_lut1);
            e = hipGetLastError();
        }
        if (G > 1) {
            size_t entries = (n_gen / G) << (3 * G - 1);
            if (e == hipSuccess) e = hipMalloc(&p->d_lut, entries * sizeof(TeEntry));
            while (e == hipErrorOutOfMemory && G > 2) {  // a crowded device: a smaller group needs an eighth of the table
                (void)hipGetLastError();
                --G;
                p->group = G;
                entries = (n_gen / G) << (3 * G - 1);
                e = hipMalloc(&p->d_lut, entries * sizeof(TeEntry));
            }
            if (e == hipSuccess) {
                hipLaunchKernelGGL(te_build_bh_lutg, dim3((unsigned)((entries + 63) / 64)), dim3(64), 0, ctx->stream, d_g, G, (u32)entries, p->d_lut);
                e = hipGetLastError();
            }
        }
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (d_g) (void)hipFree(d_g);
    if (e != hipSuccess) {
        if (p->d_lut) (void)hipFree(p->d_lut);
        if (p->d_lut1) (void)hipFree(p->d_lut1);
        delete p;
        return fail(AKP_ERR_HIP, "akp_te_params_create: %s", hipGetErrorString(e));
    }
    *out = p;
    return AKP_OK;
}
extern "C" uint32_t akp_te_entry_bytes(void) { return (uint32_t)sizeof(TeEntry); }
extern "C" void akp_te_params_destroy(akp_te_params* p) {
    if (!p) return;
    (void)hipSetDevice(p->ctx->device);
    (void)hipDeviceSynchronize();
    if (p->d_lut) (void)hipFree(p->d_lut);
    if (p->d_lut1) (void)hipFree(p->d_lut1);
    if (p->d_tail) (void)hipFree(p->d_tail);
    ctx_handle_released(p->ctx);
    delete p;
}

// table steps a message of msg_len bytes touches (later digits are zero / absent): Pedersen pads with zero
// bytes (those digits select the identity); Bowe-Hopwood stops at ceil(bits/3) chunks = `groups` triples +
// left-over singles.
static void te_steps(const akp_te_params* p, size_t msg_len, u32* n_groups, u32* n_steps) {
    const size_t bits = msg_len * 8;
    if (te_is_pedersen(p)) {
        const size_t used = std::min<size_t>(bits, p->n_gen);
        *n_groups = 0;
        *n_steps = (u32)((used + p->digit_bits - 1) / p->digit_bits);
        return;
    }
    const size_t chunks = std::min<size_t>((bits + 2) / 3, (size_t)p->n_gen);
    if (p->group > 1) {
        *n_groups = (u32)(chunks / p->group);
        *n_steps = (u32)(chunks / p->group + chunks % p->group);
    } else {
        *n_groups = 0;
        *n_steps = (u32)chunks;
    }
}
extern "C" int32_t akp_te_params_info(const akp_te_params* p, uint32_t* digit_bits_or_group, int32_t* signed_subset, size_t* table_bytes, size_t msg_len,
                                      uint32_t* steps) {
    if (!p) return fail(AKP_ERR_BAD_PARAMS, "akp_te_params_info: params is NULL");
    const bool ped = te_is_pedersen(p);
    if (digit_bits_or_group) *digit_bits_or_group = ped ? p->digit_bits : p->group;
    if (signed_subset) *signed_subset = ped && p->signed_subset ? 1 : 0;
    if (table_bytes) {
        size_t entries;
        if (ped) {
            const size_t n_digits = (p->n_gen + p->digit_bits - 1) / p->digit_bits;
            entries = p->signed_subset ? (n_digits << (p->digit_bits - 1)) + n_digits + 1 : n_digits << p->digit_bits;
        } else {
            entries = (size_t)p->n_gen * 4 + (p->group > 1 ? ((size_t)(p->n_gen / p->group) << (3 * p->group - 1)) : 0);
        }
        *table_bytes = entries * sizeof(TeEntry);
    }
    if (steps) {
        u32 g = 0, st = 0;
        te_steps(p, msg_len, &g, &st);
        *steps = st;
    }
    return AKP_OK;
}
// accumulate + finalize on device buffers.  scratch: SCR_E (xyz), SCR_F (prefix)
// `data_len` <= msg_len: the bytes [data_len, msg_len) of every message are known to be zero (the padding of a two-to-one
// buffer, crh/bowe_hopwood/mod.rs:219-224) and are not read.  Pedersen: zero bits select nothing, the sum simply stops
// earlier.  Bowe-Hopwood: a zero chunk still adds +g (:167), so the chunks that lie wholly in the padding contribute the
// CONSTANT sum of their generators: one table entry (computed once per shape, te_bh_tail_kernel) added at the end instead
// of one table step per five chunks -- a 63 x 9 inner node (64 bytes of digests in a 70-byte buffer) takes 35 + 1 steps
// instead of 39.
static int32_t te_crh_dev(akp_te_params* p, const uint8_t* d_msgs, size_t n, size_t msg_len, Fr* d_out, hipStream_t s, size_t data_len = (size_t)-1) {
    if (msg_len * 8 > te_input_bits(p))
        return fail(AKP_ERR_BAD_LENGTH, "incorrect input length %zu for window params %ux%u (the reference panics)", msg_len, p->W, p->N);
    if (n == 0) return AKP_OK;
    if (n > ((size_t)1 << 32)) return fail(AKP_ERR_BAD_PARAMS, "batch of %zu messages exceeds the supported 2^32", n);
    const bool tail_on = env_u32("AKP_BH_ZERO_TAIL", 1, 0, 1) != 0;  // 0: walk the padding chunk by chunk (A/B arm; read per call)
    if (data_len > msg_len || !tail_on) data_len = msg_len;
    u32 groups = 0, steps = 0;
    te_steps(p, data_len, &groups, &steps);
    const TeEntry* tail = nullptr;
    if (p->kind == AKP_TE_BOWE_HOPWOOD && data_len < msg_len) {
        const u32 from = (u32)std::min<size_t>((data_len * 8 + 2) / 3, p->n_gen), to = (u32)std::min<size_t>((msg_len * 8 + 2) / 3, p->n_gen);
        if (from < to) {
            if (!p->d_tail) HIP_TRY(hipMalloc(&p->d_tail, sizeof(TeEntry)));
            if (p->tail_from != from || p->tail_to != to) {  // stream-ordered: later launches on other streams go through ctx_scratch-style events below
                HIP_TRY(hipStreamSynchronize(s));           // an earlier shape's constant may still be in use (rare: one shape per parameter set)
                hipLaunchKernelGGL(te_bh_tail_kernel, dim3(1), dim3(64), 0, s, p->d_lut1, from, to, p->d_tail);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipStreamSynchronize(s));
                p->tail_from = from;
                p->tail_to = to;
            }
            tail = p->d_tail;
        }
    }
    size_t stride = msg_len;
    if (data_len > 0 && data_len < 4) {  // the kernels fetch message bits with one 32-bit load: pad 1..3-byte messages to four bytes
        void* pad = nullptr;
        if (int32_t rc = ctx_scratch(p->ctx, SCR_L, n * 4, &pad, s)) return rc;
        hipLaunchKernelGGL(te_pad4_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_msgs, stride, (u32)data_len, (uint8_t*)pad, n);
        HIP_TRY(hipGetLastError());
        d_msgs = (const uint8_t*)pad;
        stride = 4;
        data_len = 4;  // three zero bytes at most: they select nothing (Pedersen) / lie past the steps counted above (Bowe-Hopwood)
    }
    // small batches (tree tops) are bound by the latency of one message: split each one over AKP_TE_SPLIT waves
    static const size_t split_max = env_size("AKP_TE_SPLIT_MAX", (size_t)1 << 14);  // one workgroup per CU
    if (n <= split_max) {
        const unsigned sgrid = (unsigned)((n + 63) / 64);
        const bool xy = p->kind == AKP_TE_PEDERSEN;  // digest = (x, y); otherwise x only
        if (te_is_pedersen(p) && p->signed_subset) {
            if (xy) hipLaunchKernelGGL((te_crh_small_kernel<2, false>), dim3(sgrid), dim3(64 * AKP_TE_SPLIT), 0, s, p->d_lut, p->d_lut1, d_msgs, data_len, stride, p->digit_bits, groups, steps, tail, d_out, n);
            else hipLaunchKernelGGL((te_crh_small_kernel<2, true>), dim3(sgrid), dim3(64 * AKP_TE_SPLIT), 0, s, p->d_lut, p->d_lut1, d_msgs, data_len, stride, p->digit_bits, groups, steps, tail, d_out, n);
        } else if (te_is_pedersen(p)) {
            if (xy) hipLaunchKernelGGL((te_crh_small_kernel<0, false>), dim3(sgrid), dim3(64 * AKP_TE_SPLIT), 0, s, p->d_lut, p->d_lut1, d_msgs, data_len, stride, p->digit_bits, groups, steps, tail, d_out, n);
            else hipLaunchKernelGGL((te_crh_small_kernel<0, true>), dim3(sgrid), dim3(64 * AKP_TE_SPLIT), 0, s, p->d_lut, p->d_lut1, d_msgs, data_len, stride, p->digit_bits, groups, steps, tail, d_out, n);
        } else {
            hipLaunchKernelGGL((te_crh_small_kernel<1, true>), dim3(sgrid), dim3(64 * AKP_TE_SPLIT), 0, s, p->d_lut, p->d_lut1, d_msgs, data_len, stride, p->group, groups, steps, tail, d_out, n);
        }
        HIP_TRY(hipGetLastError());
        return AKP_OK;
    }
    void *xyz = nullptr, *prefix = nullptr;
    if (int32_t rc = ctx_scratch(p->ctx, SCR_E, n * 3 * sizeof(F29Pad), &xyz, s)) return rc;
    if (int32_t rc = ctx_scratch(p->ctx, SCR_F, n * sizeof(F29Pad), &prefix, s)) return rc;
    const unsigned grid = (unsigned)((n + 255) / 256);
    if (te_is_pedersen(p) && p->signed_subset)
        hipLaunchKernelGGL(te_accumulate_kernel<2>, dim3(grid), dim3(256), 0, s, p->d_lut, p->d_lut1, d_msgs, data_len, stride, p->digit_bits, groups, steps, tail, (F29Pad*)xyz, n);
    else if (te_is_pedersen(p))
        hipLaunchKernelGGL(te_accumulate_kernel<0>, dim3(grid), dim3(256), 0, s, p->d_lut, p->d_lut1, d_msgs, data_len, stride, p->digit_bits, groups, steps, tail, (F29Pad*)xyz, n);
    else
        hipLaunchKernelGGL(te_accumulate_kernel<1>, dim3(grid), dim3(256), 0, s, p->d_lut, p->d_lut1, d_msgs, data_len, stride, p->group, groups, steps, tail, (F29Pad*)xyz, n);
    HIP_TRY(hipGetLastError());
    // share one inversion among up to 64 messages per lane, but keep `target` lanes busy when n allows.  Measured at 2^20
    // Pedersen hashes (profiles/r02_s27): 16 K / 32 K / 64 K / 128 K / 256 K / 512 K lanes -> 3.57 / 3.41 / 3.39 | 3.21 / 3.24 /
    // 3.29 / 3.46 ms for accumulate + finalize (two boxes): one wave per SIMD it is.  AKP_TE_FINALIZE_LANES for A/B runs.
    static const size_t target = (size_t)env_u32("AKP_TE_FINALIZE_LANES", 65536, 64, 1u << 24);
    size_t chain = std::min<size_t>(64, std::max<size_t>(1, n / target));
    size_t lanes = (n + chain - 1) / chain;
    const unsigned fgrid = (unsigned)((lanes + 255) / 256);
    if (p->kind == AKP_TE_PEDERSEN)
        hipLaunchKernelGGL(te_finalize_kernel<0>, dim3(fgrid), dim3(256), 0, s, (const F29Pad*)xyz, (F29Pad*)prefix, d_out, n, lanes);
    else
        hipLaunchKernelGGL(te_finalize_kernel<1>, dim3(fgrid), dim3(256), 0, s, (const F29Pad*)xyz, (F29Pad*)prefix, d_out, n, lanes);
    HIP_TRY(hipGetLastError());
    return AKP_OK;
}
#define NEED_TE(p, what)                                                    \
    do {                                                                    \
        if (!(p)) return fail(AKP_ERR_BAD_PARAMS, what ": params is NULL"); \
        if ((p)->ctx->dead) return fail(AKP_ERR_BAD_PARAMS, what ": the context of this handle was destroyed"); \
        HIP_TRY(hipSetDevice((p)->ctx->device));                            \
    } while (0)

extern "C" int32_t akp_te_crh_batch_dev(akp_te_params* p, const uint8_t* d_msgs, size_t n, size_t msg_len, uint64_t* d_out, void* stream) {
    NEED_TE(p, "akp_te_crh_batch_dev");
    return te_crh_dev(p, d_msgs, n, msg_len, (Fr*)d_out, pick_stream(p->ctx, stream));
}
extern "C" int32_t akp_te_crh_batch(akp_te_params* p, const uint8_t* msgs, size_t n, size_t msg_len, uint64_t* out) {
    NEED_TE(p, "akp_te_crh_batch");
    if (msg_len * 8 > te_input_bits(p))
        return fail(AKP_ERR_BAD_LENGTH, "incorrect input length %zu for window params %ux%u (the reference panics)", msg_len, p->W, p->N);
    if (n == 0) return AKP_OK;
    if (!out || (!msgs && msg_len)) return fail(AKP_ERR_BAD_PARAMS, "NULL buffer");
    const size_t fe = te_fe_per_digest(p);
    akp_ctx* c = p->ctx;
    void *dm = nullptr, *dout = nullptr;
    hipStream_t s = c->stream;
    // Large batches: chunks of 2^AKP_TE_HOST_CHUNK_LOG2 messages (default 2^17), double-buffered -- the copy-in of chunk
    // i + 1 and the copy-out of chunk i - 1 run on their own streams under the kernels of chunk i (a 4x256 Pedersen hash
    // moves 128 B in and 64 B out for 3 us of kernel time per 1000 hashes: serial copies would double the call).
    static const size_t chunk = (size_t)1 << env_u32("AKP_TE_HOST_CHUNK_LOG2", 17, 10, 30);
    if (n <= chunk || msg_len == 0) {
        if (int32_t rc = ctx_scratch(c, SCR_A, n * msg_len, &dm, s)) return rc;
        if (int32_t rc = ctx_scratch(c, SCR_B, n * fe * sizeof(Fr), &dout, s)) return rc;
        if (msg_len) HIP_TRY(hipMemcpyAsync(dm, msgs, n * msg_len, hipMemcpyHostToDevice, s));
        if (int32_t rc = te_crh_dev(p, (const uint8_t*)dm, n, msg_len, (Fr*)dout, s)) return rc;
        HIP_TRY(hipMemcpyAsync(out, dout, n * fe * sizeof(Fr), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        return AKP_OK;
    }
    const size_t dig = fe * sizeof(Fr);
    if (int32_t rc = ctx_scratch(c, SCR_A, 2 * chunk * msg_len, &dm, s)) return rc;
    if (int32_t rc = ctx_scratch(c, SCR_B, 2 * chunk * dig, &dout, s)) return rc;
    for (int i = 0; i < 2; ++i)
        if (!c->pipe[i]) HIP_TRY(hipStreamCreateWithFlags(&c->pipe[i], hipStreamNonBlocking));
    for (int i = 0; i < 8; ++i)
        if (!c->chunk_event[i]) HIP_TRY(hipEventCreateWithFlags(&c->chunk_event[i], hipEventDisableTiming));
    hipStream_t cin = c->pipe[0], cout = c->pipe[1];
    hipEvent_t *in_done = c->chunk_event, *comp_done = c->chunk_event + 2, *out_done = c->chunk_event + 4;
    HIP_TRY(hipEventRecord(c->chunk_event[7], s));  // the copy streams start behind whatever used the scratch last
    HIP_TRY(hipStreamWaitEvent(cin, c->chunk_event[7], 0));
    HIP_TRY(hipStreamWaitEvent(cout, c->chunk_event[7], 0));
    const size_t n_chunks = (n + chunk - 1) / chunk;
    for (size_t ci = 0; ci <= n_chunks; ++ci) {
        if (ci < n_chunks) {
            const size_t done = ci * chunk, cnt = std::min(chunk, n - done);
            const int b = (int)(ci & 1);
            uint8_t* d_in = (uint8_t*)dm + (size_t)b * chunk * msg_len;
            Fr* d_o = (Fr*)((char*)dout + (size_t)b * chunk * dig);
            if (ci >= 2) HIP_TRY(hipStreamWaitEvent(cin, comp_done[b], 0));  // the kernels of chunk ci - 2 have read this half
            HIP_TRY(hipMemcpyAsync(d_in, msgs + done * msg_len, cnt * msg_len, hipMemcpyHostToDevice, cin));
            HIP_TRY(hipEventRecord(in_done[b], cin));
            HIP_TRY(hipStreamWaitEvent(s, in_done[b], 0));
            if (ci >= 2) HIP_TRY(hipStreamWaitEvent(s, out_done[b], 0));  // the copy-out of chunk ci - 2 has drained this half
            if (int32_t rc = te_crh_dev(p, d_in, cnt, msg_len, d_o, s)) return rc;
            HIP_TRY(hipEventRecord(comp_done[b], s));
        }
        if (ci >= 1) {  // issued after the copy-in of the next chunk: the copy engines serve the queues in issue order
            const size_t co = ci - 1, done = co * chunk, cnt = std::min(chunk, n - done);
            const int b = (int)(co & 1);
            HIP_TRY(hipStreamWaitEvent(cout, comp_done[b], 0));
            HIP_TRY(hipMemcpyAsync((char*)out + done * dig, (char*)dout + (size_t)b * chunk * dig, cnt * dig, hipMemcpyDeviceToHost, cout));
            HIP_TRY(hipEventRecord(out_done[b], cout));
        }
    }
    HIP_TRY(hipStreamSynchronize(cin));
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(hipStreamSynchronize(cout));
    return AKP_OK;
}
extern "C" int32_t akp_te_two_to_one_batch(akp_te_params* p, const uint8_t* left, const uint8_t* right, size_t n, size_t half_len, uint64_t* out) {
    NEED_TE(p, "akp_te_two_to_one_batch");
    if (n == 0) return AKP_OK;
    if (!out || ((!left || !right) && half_len)) return fail(AKP_ERR_BAD_PARAMS, "NULL buffer");
    const size_t buflen = ((size_t)p->W * p->N) / 8;  // both schemes size the buffer from pedersen's INPUT_SIZE_BITS
    const size_t fe = te_fe_per_digest(p);
    void *dl = nullptr, *dr = nullptr, *dbuf = nullptr, *dout = nullptr;
    if (int32_t rc = ctx_scratch(p->ctx, SCR_A, n * half_len, &dl, p->ctx->stream)) return rc;
    if (int32_t rc = ctx_scratch(p->ctx, SCR_B, n * half_len, &dr, p->ctx->stream)) return rc;
    if (int32_t rc = ctx_scratch(p->ctx, SCR_C, n * buflen, &dbuf, p->ctx->stream)) return rc;
    if (int32_t rc = ctx_scratch(p->ctx, SCR_D, n * fe * sizeof(Fr), &dout, p->ctx->stream)) return rc;
    hipStream_t s = p->ctx->stream;
    if (half_len) {
        HIP_TRY(hipMemcpyAsync(dl, left, n * half_len, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(dr, right, n * half_len, hipMemcpyHostToDevice, s));
    }
    if (buflen) {
        const size_t work = n * buflen;
        hipLaunchKernelGGL(te_concat_bytes_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, s, (const uint8_t*)dl, (const uint8_t*)dr, half_len, buflen, (uint8_t*)dbuf, n);
        HIP_TRY(hipGetLastError());
    }
    // the buffer past left || right is zero padding: te_crh_dev skips it (Pedersen) or adds its constant (Bowe-Hopwood)
    if (int32_t rc = te_crh_dev(p, (const uint8_t*)dbuf, n, buflen, (Fr*)dout, s, std::min(buflen, 2 * half_len))) return rc;
    HIP_TRY(hipMemcpyAsync(out, dout, n * fe * sizeof(Fr), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return AKP_OK;
}
// one level of compress(): serialise digest pairs into per-node buffers (SCR_D), hash into d_out
static int32_t te_compress_dev(akp_te_params* p, const Fr* d_left, const Fr* d_right, size_t n, Fr* d_out, hipStream_t s) {
    if (n == 0) return AKP_OK;
    const size_t buflen = ((size_t)p->W * p->N) / 8;
    const u32 fe = te_fe_per_digest(p);
    const size_t used = std::min<size_t>(buflen, (size_t)2 * fe * 32);
    void* dbuf = nullptr;
    if (int32_t rc = ctx_scratch(p->ctx, SCR_D, n * buflen, &dbuf, s)) return rc;
    const size_t work = n * 2 * fe;
    hipLaunchKernelGGL(te_serialize_pairs_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, s, d_left, d_right, fe, buflen, (uint8_t*)dbuf, n);
    HIP_TRY(hipGetLastError());
    const bool tail_on = env_u32("AKP_BH_ZERO_TAIL", 1, 0, 1) != 0;
    if (used < buflen && !tail_on) {  // with the zero-tail shortcut the padding bytes are never read
        const size_t tw = n * (buflen - used);
        hipLaunchKernelGGL(te_zero_tail_kernel, dim3((unsigned)((tw + 255) / 256)), dim3(256), 0, s, (uint8_t*)dbuf, buflen, used, n);
        HIP_TRY(hipGetLastError());
    }
    return te_crh_dev(p, (const uint8_t*)dbuf, n, buflen, d_out, s, used);
}
extern "C" int32_t akp_te_compress_batch(akp_te_params* p, const uint64_t* left, const uint64_t* right, size_t n, uint64_t* out) {
    NEED_TE(p, "akp_te_compress_batch");
    if (n == 0) return AKP_OK;
    if (!left || !right || !out) return fail(AKP_ERR_BAD_PARAMS, "NULL buffer");
    const size_t fe = te_fe_per_digest(p);
    void *dl = nullptr, *dr = nullptr, *dout = nullptr;
    if (int32_t rc = ctx_scratch(p->ctx, SCR_A, n * fe * sizeof(Fr), &dl, p->ctx->stream)) return rc;
    if (int32_t rc = ctx_scratch(p->ctx, SCR_B, n * fe * sizeof(Fr), &dr, p->ctx->stream)) return rc;
    if (int32_t rc = ctx_scratch(p->ctx, SCR_C, n * fe * sizeof(Fr), &dout, p->ctx->stream)) return rc;
    hipStream_t s = p->ctx->stream;
    HIP_TRY(hipMemcpyAsync(dl, left, n * fe * sizeof(Fr), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(dr, right, n * fe * sizeof(Fr), hipMemcpyHostToDevice, s));
    if (int32_t rc = te_compress_dev(p, (const Fr*)dl, (const Fr*)dr, n, (Fr*)dout, s)) return rc;
    HIP_TRY(hipMemcpyAsync(out, dout, n * fe * sizeof(Fr), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return AKP_OK;
}

// ------------------------------------------------------------------------------------------
// Merkle tree (merkle_tree/mod.rs:411-523): one launch per level, bottom-up; level l of the
// heap-ordered non_leaf array starts at 2^l - 1.
static inline bool pow2_gt1(size_t n) { return n > 1 && (n & (n - 1)) == 0; }


// MerkleTree::new from host leaves: the copy-in of leaf chunk i + 1 (copy stream) overlaps the leaf hashing of chunk i
// (context stream); the inner levels follow on the context stream while the copy stream already returns the leaf digests;
// the inner nodes go back last.  hash_leaves(d_chunk, first_leaf, count, stream), inner(stream).
template <class HashLeaves, class Inner>
static int32_t host_tree_build(akp_ctx* c, const void* leaves, size_t n, size_t leaf_bytes, size_t dig_bytes, void* d_leaves, void* d_ln, void* d_nl,
                               void* h_ln, void* h_nl, void* h_root, HashLeaves hash_leaves, Inner inner) {
    static const size_t chunk_items = (size_t)1 << env_u32("AKP_TREE_CHUNK_LOG2", 20, 12, 30);
    if (!c->pipe[0]) HIP_TRY(hipStreamCreateWithFlags(&c->pipe[0], hipStreamNonBlocking));
    for (int i = 0; i < 8; ++i)
        if (!c->chunk_event[i]) HIP_TRY(hipEventCreateWithFlags(&c->chunk_event[i], hipEventDisableTiming));
    hipStream_t comp = c->stream, copy = c->pipe[0];
    // the scratch regions were acquired for the context stream: let the copy stream start behind whatever used them last
    HIP_TRY(hipEventRecord(c->chunk_event[7], comp));
    HIP_TRY(hipStreamWaitEvent(copy, c->chunk_event[7], 0));
    size_t ci = 0;
    for (size_t done = 0; done < n; done += chunk_items, ++ci) {
        const size_t cnt = std::min(chunk_items, n - done);
        if (leaf_bytes) {
            HIP_TRY(hipMemcpyAsync((char*)d_leaves + done * leaf_bytes, (const char*)leaves + done * leaf_bytes, cnt * leaf_bytes, hipMemcpyHostToDevice, copy));
            hipEvent_t e = c->chunk_event[ci % 6];
            HIP_TRY(hipEventRecord(e, copy));
            HIP_TRY(hipStreamWaitEvent(comp, e, 0));
        }
        if (int32_t rc = hash_leaves((const char*)d_leaves + done * leaf_bytes, done, cnt, comp)) return rc;
    }
    if (h_ln) {  // leaf digests are final: copy them out while the inner levels run
        HIP_TRY(hipEventRecord(c->chunk_event[6], comp));
        HIP_TRY(hipStreamWaitEvent(copy, c->chunk_event[6], 0));
        HIP_TRY(hipMemcpyAsync(h_ln, d_ln, n * dig_bytes, hipMemcpyDeviceToHost, copy));
    }
    if (int32_t rc = inner(comp)) return rc;
    if (h_nl) HIP_TRY(hipMemcpyAsync(h_nl, d_nl, (n - 1) * dig_bytes, hipMemcpyDeviceToHost, comp));
    if (h_root) HIP_TRY(hipMemcpyAsync(h_root, d_nl, dig_bytes, hipMemcpyDeviceToHost, comp));
    HIP_TRY(hipStreamSynchronize(copy));
    HIP_TRY(hipStreamSynchronize(comp));
    c->last_tree_non_leaf = d_nl;
    c->last_tree_nodes = n - 1;
    return AKP_OK;
}

extern "C" int32_t akp_merkle_inner_poseidon_dev(akp_poseidon* two, const uint64_t* d_leaf_nodes, size_t n, uint64_t* d_non_leaf, void* stream) {
    NEED_DEV(two, "akp_merkle_inner_poseidon_dev");
    if (!pow2_gt1(n)) return fail(AKP_ERR_NOT_POW2, "leaves.len() should be power of two and greater than one (got %zu)", n);
    hipStream_t s = pick_stream(two->ctx, stream);
    Fr* nl = (Fr*)d_non_leaf;
    const Fr* child = (const Fr*)d_leaf_nodes;
    for (size_t width = n / 2; width >= 1; width /= 2) {
        const size_t first = width - 1;
        if (int32_t rc = launch_crh(two, child, nullptr, 2, nl + first, width, s)) return rc;
        child = nl + first;
    }
    return AKP_OK;
}
extern "C" int32_t akp_merkle_inner_poseidon(akp_poseidon* two, const uint64_t* leaf_nodes, size_t n, uint64_t* non_leaf) {
    NEED_DEV(two, "akp_merkle_inner_poseidon");
    if (!pow2_gt1(n)) return fail(AKP_ERR_NOT_POW2, "leaves.len() should be power of two and greater than one (got %zu)", n);
    if (!leaf_nodes || !non_leaf) return fail(AKP_ERR_BAD_PARAMS, "NULL buffer");
    akp_ctx* c = two->ctx;
    void *dln = nullptr, *dnl = nullptr;
    if (int32_t rc = ctx_scratch(c, SCR_B, n * sizeof(Fr), &dln, c->stream)) return rc;
    if (int32_t rc = ctx_scratch(c, SCR_C, (n - 1) * sizeof(Fr), &dnl, c->stream)) return rc;
    hipStream_t s = c->stream;
    HIP_TRY(hipMemcpyAsync(dln, leaf_nodes, n * sizeof(Fr), hipMemcpyHostToDevice, s));
    if (int32_t rc = akp_merkle_inner_poseidon_dev(two, (const uint64_t*)dln, n, (uint64_t*)dnl, (void*)s)) return rc;
    HIP_TRY(hipMemcpyAsync(non_leaf, dnl, (n - 1) * sizeof(Fr), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return AKP_OK;
}
extern "C" int32_t akp_merkle_build_poseidon_dev(akp_poseidon* leafp, akp_poseidon* two, const uint64_t* d_leaves, size_t n, size_t leaf_len,
                                                 uint64_t* d_leaf_nodes, uint64_t* d_non_leaf, void* stream) {
    NEED_DEV(leafp, "akp_merkle_build_poseidon_dev");
    NEED_DEV(two, "akp_merkle_build_poseidon_dev");
    if (leafp->ctx != two->ctx) return fail(AKP_ERR_BAD_PARAMS, "leaf and two-to-one parameters belong to different contexts");
    if (!pow2_gt1(n)) return fail(AKP_ERR_NOT_POW2, "leaves.len() should be power of two and greater than one (got %zu)", n);
    hipStream_t s = pick_stream(leafp->ctx, stream);
    if (int32_t rc = launch_crh(leafp, (const Fr*)d_leaves, nullptr, leaf_len, (Fr*)d_leaf_nodes, n, s)) return rc;  // :417-419
    return akp_merkle_inner_poseidon_dev(two, d_leaf_nodes, n, d_non_leaf, (void*)s);
}
extern "C" int32_t akp_merkle_build_poseidon(akp_poseidon* leafp, akp_poseidon* two, const uint64_t* leaves, size_t n, size_t leaf_len,
                                             uint64_t* leaf_nodes, uint64_t* non_leaf, uint64_t* root_out) {
    NEED_DEV(leafp, "akp_merkle_build_poseidon");
    NEED_DEV(two, "akp_merkle_build_poseidon");
    if (!pow2_gt1(n)) return fail(AKP_ERR_NOT_POW2, "leaves.len() should be power of two and greater than one (got %zu)", n);
    if (!leaves && leaf_len) return fail(AKP_ERR_BAD_PARAMS, "leaves is NULL");
    akp_ctx* c = leafp->ctx;
    void *dl = nullptr, *dln = nullptr, *dnl = nullptr;
    if (int32_t rc = ctx_scratch(c, SCR_A, n * leaf_len * sizeof(Fr), &dl, c->stream)) return rc;
    if (int32_t rc = ctx_scratch(c, SCR_B, n * sizeof(Fr), &dln, c->stream)) return rc;
    if (int32_t rc = ctx_scratch(c, SCR_C, (n - 1) * sizeof(Fr), &dnl, c->stream)) return rc;
    if (leafp->ctx != two->ctx) return fail(AKP_ERR_BAD_PARAMS, "leaf and two-to-one parameters belong to different contexts");
    return host_tree_build(
        c, leaves, n, leaf_len * sizeof(Fr), sizeof(Fr), dl, dln, dnl, leaf_nodes, non_leaf, root_out,
        [&](const void* d_chunk, size_t first, size_t cnt, hipStream_t s) -> int32_t {
            return launch_crh(leafp, (const Fr*)d_chunk, nullptr, leaf_len, (Fr*)dln + first, cnt, s);  // :417-419
        },
        [&](hipStream_t s) -> int32_t { return akp_merkle_inner_poseidon_dev(two, (const uint64_t*)dln, n, (uint64_t*)dnl, (void*)s); });
}

extern "C" int32_t akp_merkle_inner_te_dev(akp_te_params* two, const uint64_t* d_leaf_nodes, size_t n, uint64_t* d_non_leaf, void* stream) {
    NEED_TE(two, "akp_merkle_inner_te_dev");
    if (!pow2_gt1(n)) return fail(AKP_ERR_NOT_POW2, "leaves.len() should be power of two and greater than one (got %zu)", n);
    hipStream_t s = pick_stream(two->ctx, stream);
    const u32 fe = te_fe_per_digest(two);
    Fr* nl = (Fr*)d_non_leaf;
    const Fr* child = (const Fr*)d_leaf_nodes;
    for (size_t width = n / 2; width >= 1; width /= 2) {
        const size_t first = width - 1;
        if (int32_t rc = te_compress_dev(two, child, nullptr, width, nl + first * fe, s)) return rc;
        child = nl + first * fe;
    }
    return AKP_OK;
}
extern "C" int32_t akp_merkle_inner_te(akp_te_params* two, const uint64_t* leaf_nodes, size_t n, uint64_t* non_leaf) {
    NEED_TE(two, "akp_merkle_inner_te");
    if (!pow2_gt1(n)) return fail(AKP_ERR_NOT_POW2, "leaves.len() should be power of two and greater than one (got %zu)", n);
    if (!leaf_nodes || !non_leaf) return fail(AKP_ERR_BAD_PARAMS, "NULL buffer");
    akp_ctx* c = two->ctx;
    const size_t fe = te_fe_per_digest(two);
    void *dln = nullptr, *dnl = nullptr;
    if (int32_t rc = ctx_scratch(c, SCR_B, n * fe * sizeof(Fr), &dln, c->stream)) return rc;
    if (int32_t rc = ctx_scratch(c, SCR_C, (n - 1) * fe * sizeof(Fr), &dnl, c->stream)) return rc;
    hipStream_t s = c->stream;
    HIP_TRY(hipMemcpyAsync(dln, leaf_nodes, n * fe * sizeof(Fr), hipMemcpyHostToDevice, s));
    if (int32_t rc = akp_merkle_inner_te_dev(two, (const uint64_t*)dln, n, (uint64_t*)dnl, (void*)s)) return rc;
    HIP_TRY(hipMemcpyAsync(non_leaf, dnl, (n - 1) * fe * sizeof(Fr), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return AKP_OK;
}
extern "C" int32_t akp_merkle_build_te_dev(akp_te_params* leafp, akp_te_params* two, const uint8_t* d_leaves, size_t n, size_t leaf_len,
                                           uint64_t* d_leaf_nodes, uint64_t* d_non_leaf, void* stream) {
    NEED_TE(leafp, "akp_merkle_build_te_dev");
    NEED_TE(two, "akp_merkle_build_te_dev");
    if (leafp->kind != two->kind) return fail(AKP_ERR_BAD_PARAMS, "leaf and two-to-one hashes must be of the same kind");
    if (leafp->ctx != two->ctx) return fail(AKP_ERR_BAD_PARAMS, "parameters belong to different contexts");
    if (!pow2_gt1(n)) return fail(AKP_ERR_NOT_POW2, "leaves.len() should be power of two and greater than one (got %zu)", n);
    hipStream_t s = pick_stream(leafp->ctx, stream);
    if (int32_t rc = te_crh_dev(leafp, d_leaves, n, leaf_len, (Fr*)d_leaf_nodes, s)) return rc;
    return akp_merkle_inner_te_dev(two, d_leaf_nodes, n, d_non_leaf, (void*)s);
}
extern "C" int32_t akp_merkle_build_te(akp_te_params* leafp, akp_te_params* two, const uint8_t* leaves, size_t n, size_t leaf_len,
                                       uint64_t* leaf_nodes, uint64_t* non_leaf, uint64_t* root_out) {
    NEED_TE(leafp, "akp_merkle_build_te");
    NEED_TE(two, "akp_merkle_build_te");
    if (!pow2_gt1(n)) return fail(AKP_ERR_NOT_POW2, "leaves.len() should be power of two and greater than one (got %zu)", n);
    if (leaf_len * 8 > te_input_bits(leafp))
        return fail(AKP_ERR_BAD_LENGTH, "incorrect input length %zu for window params %ux%u (the reference panics)", leaf_len, leafp->W, leafp->N);
    if (!leaves && leaf_len) return fail(AKP_ERR_BAD_PARAMS, "leaves is NULL");
    akp_ctx* c = leafp->ctx;
    const size_t fe = te_fe_per_digest(two);
    void *dl = nullptr, *dln = nullptr, *dnl = nullptr;
    if (int32_t rc = ctx_scratch(c, SCR_A, n * leaf_len, &dl, c->stream)) return rc;
    if (int32_t rc = ctx_scratch(c, SCR_B, n * fe * sizeof(Fr), &dln, c->stream)) return rc;
    if (int32_t rc = ctx_scratch(c, SCR_C, (n - 1) * fe * sizeof(Fr), &dnl, c->stream)) return rc;
    if (leafp->kind != two->kind || leafp->ctx != two->ctx) return fail(AKP_ERR_BAD_PARAMS, "leaf / two-to-one parameters mismatch");
    return host_tree_build(
        c, leaves, n, leaf_len, fe * sizeof(Fr), dl, dln, dnl, leaf_nodes, non_leaf, root_out,
        [&](const void* d_chunk, size_t first, size_t cnt, hipStream_t s) -> int32_t {
            return te_crh_dev(leafp, (const uint8_t*)d_chunk, cnt, leaf_len, (Fr*)dln + first * fe, s);
        },
        [&](hipStream_t s) -> int32_t { return akp_merkle_inner_te_dev(two, (const uint64_t*)dln, n, (uint64_t*)dnl, (void*)s); });
}

// ------------------------------------------------------------------------------------------
// Merkle proofs: gather (generate_proof) and batched verify (Path::verify)
static inline size_t log2_exact(size_t n) { size_t l = 0; while (((size_t)1 << l) < n) ++l; return l; }

// path i, level j (0 = root side): sibling of the ancestor of leaf idx at tree depth j + 1
//   ancestor heap index at depth d (root = depth 0): (2^d - 1) + (idx >> (log2n - d))
AKP_HD size_t merkle_auth_node(size_t log2n, size_t leaf_index, size_t j) {
    const size_t d = j + 1;
    const size_t anc = (((size_t)1 << d) - 1) + (leaf_index >> (log2n - d));
    return (anc & 1) ? anc + 1 : anc - 1;  // sibling(): left children have odd heap indices (:764-771)
}
__global__ void merkle_gather_kernel(const Fr* __restrict__ leaf_nodes, const Fr* __restrict__ non_leaf, size_t log2n, u32 fe,
                                     const uint64_t* __restrict__ idx, size_t m, Fr* __restrict__ sib_out, Fr* __restrict__ auth_out) {
    const size_t depth = log2n - 1;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m * (depth + 1)) return;
    const size_t i = t / (depth + 1), j = t % (depth + 1);
    const size_t li = idx[i];
    for (u32 e = 0; e < fe; ++e) {
        if (j == depth) store_fr_global(sib_out + i * fe + e, load_fr_global(leaf_nodes + (li ^ 1) * fe + e));  // :536-544
        else store_fr_global(auth_out + (i * depth + j) * fe + e, load_fr_global(non_leaf + merkle_auth_node(log2n, li, j) * fe + e));
    }
}
extern "C" int32_t akp_merkle_gather_paths(const uint64_t* leaf_nodes, const uint64_t* non_leaf, size_t n, uint32_t fe, const uint64_t* idx,
                                           size_t m, uint64_t* sib_out, uint64_t* auth_out) {
    if (!pow2_gt1(n)) return fail(AKP_ERR_NOT_POW2, "tree must have a power-of-two number of leaves > 1");
    if (fe != 1 && fe != 2) return fail(AKP_ERR_BAD_PARAMS, "fe_per_digest must be 1 or 2");
    if (m && (!leaf_nodes || !idx || !sib_out || (n > 2 && (!non_leaf || !auth_out)))) return fail(AKP_ERR_BAD_PARAMS, "NULL buffer");
    const size_t log2n = log2_exact(n), depth = log2n - 1, w = 4 * (size_t)fe;
    for (size_t i = 0; i < m; ++i) {
        if (idx[i] >= n) return fail(AKP_ERR_BAD_PARAMS, "leaf index %llu out of range", (unsigned long long)idx[i]);
        memcpy(sib_out + i * w, leaf_nodes + (idx[i] ^ 1) * w, w * 8);
        for (size_t j = 0; j < depth; ++j) memcpy(auth_out + (i * depth + j) * w, non_leaf + merkle_auth_node(log2n, idx[i], j) * w, w * 8);
    }
    return AKP_OK;
}
extern "C" int32_t akp_merkle_gather_paths_dev(akp_ctx* ctx, const uint64_t* d_leaf_nodes, const uint64_t* d_non_leaf, size_t n, uint32_t fe,
                                               const uint64_t* d_idx, size_t m, uint64_t* d_sib, uint64_t* d_auth, void* stream) {
    if (!ctx) return fail(AKP_ERR_HIP, "akp_merkle_gather_paths_dev: a device context is required");
    if (!pow2_gt1(n)) return fail(AKP_ERR_NOT_POW2, "tree must have a power-of-two number of leaves > 1");
    if (fe != 1 && fe != 2) return fail(AKP_ERR_BAD_PARAMS, "fe_per_digest must be 1 or 2");
    if (m == 0) return AKP_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t log2n = log2_exact(n), work = m * log2n;
    hipLaunchKernelGGL(merkle_gather_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const Fr*)d_leaf_nodes,
                       (const Fr*)d_non_leaf, log2n, fe, d_idx, m, (Fr*)d_sib, (Fr*)d_auth);
    HIP_TRY(hipGetLastError());
    return AKP_OK;
}

// select_left_right_child (:367-381) for a whole level: bit `shift` of the leaf index decides the side
__global__ void merkle_select_kernel(const Fr* __restrict__ cur, const Fr* __restrict__ sib, size_t sib_stride, u32 fe,
                                     const uint64_t* __restrict__ idx, u32 shift, Fr* __restrict__ left, Fr* __restrict__ right, size_t m) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m * fe) return;
    const size_t i = t / fe, e = t % fe;
    const bool is_left = ((idx[i] >> shift) & 1) == 0;
    const Fr c = load_fr_global(cur + t), s = load_fr_global(sib + i * sib_stride + e);
    store_fr_global(left + t, is_left ? c : s);
    store_fr_global(right + t, is_left ? s : c);
}
__global__ void merkle_compare_kernel(const Fr* __restrict__ cur, const Fr* __restrict__ root, u32 fe, uint8_t* __restrict__ ok, size_t m) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    bool eq = true;
    for (u32 e = 0; e < fe; ++e) eq = eq && fr_eq(load_fr_global(cur + i * fe + e), load_fr_global(root + e));
    ok[i] = eq ? 1 : 0;
}
// shared driver: `hash_leaves` fills d_cur; `two_to_one(left, right, out)` hashes one level
template <class HashLeaves, class TwoToOne>
static int32_t verify_paths_common(akp_ctx* c, u32 fe, const uint64_t* root, size_t m, const uint64_t* idx, const uint64_t* sibs,
                                   const uint64_t* auth, size_t depth, uint8_t* ok_out, HashLeaves hash_leaves, TwoToOne two_to_one) {
    if (m == 0) return AKP_OK;
    if (!root || !idx || !sibs || !ok_out || (depth && !auth)) return fail(AKP_ERR_BAD_PARAMS, "NULL buffer");
    const size_t dig = (size_t)fe * sizeof(Fr);
    void *d_cur = nullptr, *d_l = nullptr, *d_r = nullptr, *d_idx = nullptr, *d_sib = nullptr, *d_auth = nullptr, *d_misc = nullptr;
    if (int32_t rc = ctx_scratch(c, SCR_G, m * dig, &d_cur, c->stream)) return rc;
    if (int32_t rc = ctx_scratch(c, SCR_H, m * dig, &d_l, c->stream)) return rc;
    if (int32_t rc = ctx_scratch(c, SCR_I, m * dig, &d_r, c->stream)) return rc;
    const size_t idx_bytes = (m * 8 + 15) & ~(size_t)15;  // keep the digests behind the index array 16-byte aligned
    if (int32_t rc = ctx_scratch(c, SCR_J, idx_bytes + m * dig + dig + m, &d_misc, c->stream)) return rc;
    if (int32_t rc = ctx_scratch(c, SCR_K, std::max<size_t>(m * depth * dig, 16), &d_auth, c->stream)) return rc;
    d_idx = d_misc;
    d_sib = (char*)d_misc + idx_bytes;
    void* d_root = (char*)d_sib + m * dig;
    uint8_t* d_ok = (uint8_t*)d_root + dig;
    hipStream_t s = c->stream;
    HIP_TRY(hipMemcpyAsync(d_idx, idx, m * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_sib, sibs, m * dig, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_root, root, dig, hipMemcpyHostToDevice, s));
    if (depth) HIP_TRY(hipMemcpyAsync(d_auth, auth, m * depth * dig, hipMemcpyHostToDevice, s));
    if (int32_t rc = hash_leaves((Fr*)d_cur, s)) return rc;
    const unsigned grid = (unsigned)((m * fe + 255) / 256);
    for (size_t step = 0; step <= depth; ++step) {
        // step 0 pairs the leaf digest with leaf_sibling_hash; step s >= 1 uses auth_path[depth - s]
        const Fr* sib = step == 0 ? (const Fr*)d_sib : (const Fr*)d_auth + (depth - step) * fe;
        const size_t stride = step == 0 ? fe : depth * fe;
        hipLaunchKernelGGL(merkle_select_kernel, dim3(grid), dim3(256), 0, s, (const Fr*)d_cur, sib, stride, fe, (const uint64_t*)d_idx, (u32)step,
                           (Fr*)d_l, (Fr*)d_r, m);
        HIP_TRY(hipGetLastError());
        if (int32_t rc = two_to_one((const Fr*)d_l, (const Fr*)d_r, (Fr*)d_cur, s)) return rc;
    }
    hipLaunchKernelGGL(merkle_compare_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, (const Fr*)d_cur, (const Fr*)d_root, fe, d_ok, m);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(ok_out, d_ok, m, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return AKP_OK;
}
extern "C" int32_t akp_merkle_verify_paths_poseidon(akp_poseidon* leafp, akp_poseidon* two, const uint64_t* root, const uint64_t* leaves, size_t m,
                                                    size_t leaf_len, const uint64_t* idx, const uint64_t* sibs, const uint64_t* auth, size_t depth,
                                                    uint8_t* ok_out) {
    NEED_DEV(leafp, "akp_merkle_verify_paths_poseidon");
    NEED_DEV(two, "akp_merkle_verify_paths_poseidon");
    if (leafp->ctx != two->ctx) return fail(AKP_ERR_BAD_PARAMS, "parameters belong to different contexts");
    if (m && !leaves && leaf_len) return fail(AKP_ERR_BAD_PARAMS, "leaves is NULL");
    akp_ctx* c = leafp->ctx;
    return verify_paths_common(
        c, 1, root, m, idx, sibs, auth, depth, ok_out,
        [&](Fr* d_cur, hipStream_t s) -> int32_t {
            void* dl = nullptr;
            if (int32_t rc = ctx_scratch(c, SCR_A, m * leaf_len * sizeof(Fr), &dl, s)) return rc;
            if (leaf_len) HIP_TRY(hipMemcpyAsync(dl, leaves, m * leaf_len * sizeof(Fr), hipMemcpyHostToDevice, s));
            return launch_crh(leafp, (const Fr*)dl, nullptr, leaf_len, d_cur, m, s);
        },
        [&](const Fr* l, const Fr* r, Fr* out, hipStream_t s) -> int32_t { return launch_crh(two, l, r, 2, out, m, s); });
}
extern "C" int32_t akp_merkle_verify_paths_te(akp_te_params* leafp, akp_te_params* two, const uint64_t* root, const uint8_t* leaves, size_t m,
                                              size_t leaf_len, const uint64_t* idx, const uint64_t* sibs, const uint64_t* auth, size_t depth,
                                              uint8_t* ok_out) {
    NEED_TE(leafp, "akp_merkle_verify_paths_te");
    NEED_TE(two, "akp_merkle_verify_paths_te");
    if (leafp->kind != two->kind || leafp->ctx != two->ctx) return fail(AKP_ERR_BAD_PARAMS, "leaf / two-to-one parameters mismatch");
    if (leaf_len * 8 > te_input_bits(leafp))
        return fail(AKP_ERR_BAD_LENGTH, "incorrect input length %zu for window params %ux%u (the reference panics)", leaf_len, leafp->W, leafp->N);
    if (m && !leaves && leaf_len) return fail(AKP_ERR_BAD_PARAMS, "leaves is NULL");
    akp_ctx* c = leafp->ctx;
    return verify_paths_common(
        c, te_fe_per_digest(two), root, m, idx, sibs, auth, depth, ok_out,
        [&](Fr* d_cur, hipStream_t s) -> int32_t {
            void* dl = nullptr;
            if (int32_t rc = ctx_scratch(c, SCR_A, m * leaf_len, &dl, s)) return rc;
            if (leaf_len) HIP_TRY(hipMemcpyAsync(dl, leaves, m * leaf_len, hipMemcpyHostToDevice, s));
            return te_crh_dev(leafp, (const uint8_t*)dl, m, leaf_len, d_cur, s);
        },
        [&](const Fr* l, const Fr* r, Fr* out, hipStream_t s) -> int32_t { return te_compress_dev(two, l, r, m, out, s); });
}

#include "capi_tree.inc"
#include "capi_multi.inc"
