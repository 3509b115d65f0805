// capi_poseidon.hip -- part of libakp.so (implementation of include/akp.h): Poseidon parameters, kernel routing, batch entry points, the
// batched duplex sponge
// Product code.  Never includes, links or calls anything under oracle/; there is no CPU fallback for any compute entry
// point (a missing device is AKP_ERR_HIP).
#include "capi_internal.hpp"
#include "poseidon_kernels.hpp"
#include "ragged_sort.hpp"
#include "poseidon_opt.hpp"

// ------------------------------------------------------------------------------------------
// Poseidon parameters
static int32_t upload_f29(akp_ctx* ctx, const std::vector<Fr>& v, F29Pad** out) {
    Fr* tmp = nullptr;
    HIP_TRY(hipMalloc(&tmp, v.size() * sizeof(Fr)));
    hipError_t e = hipMemcpy(tmp, v.data(), v.size() * sizeof(Fr), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc(out, v.size() * sizeof(F29Pad));
    if (e == hipSuccess) {
        hipLaunchKernelGGL(poseidon_convert_params_kernel, dim3((unsigned)((v.size() + 63) / 64)), dim3(64), 0, ctx->stream, tmp, *out,
                v.size());
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    }
    (void)hipFree(tmp);
    if (e != hipSuccess) return fail(AKP_ERR_HIP, "uploading optimised Poseidon constants: %s", hipGetErrorString(e));
    return AKP_OK;
}

// A/B arms whose comparison is settled (dense partial rounds, plain sparse form, lane-1 instead of the full form, LDS-file instead of
// the register kernels for t = 4 .. 9): selectable by environment only in the test build (-DAKP_TEST_HOOKS, libakp_testhooks.so);
// the simpler forms stay in libakp.so as the fallbacks of parameter sets that do not admit the faster ones
static inline bool arm_env(const char* name) {
#if defined(AKP_TEST_HOOKS)
    return getenv(name) != nullptr;
#else
    (void)name;
    return false;
#endif
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
            if (na) hipLaunchKernelGGL(poseidon_convert_params_kernel, dim3((unsigned)((na + 63) / 64)), dim3(64), 0, ctx->stream,
                    p->d_ark, p->d_ark29, na);
            hipLaunchKernelGGL(poseidon_convert_params_kernel, dim3((unsigned)((nm + 63) / 64)), dim3(64), 0, ctx->stream, p->d_mds,
                    p->d_mds29, nm);
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
        if (!arm_env("AKP_POSEIDON_DENSE")) {
            PoseidonOpt opt = poseidon_optimize(t, full_rounds, partial_rounds, p->ark, p->mds);
            PoseidonOpt optw = opt;
            const bool rescale = opt.ok && !arm_env("AKP_POSEIDON_NO_RESCALE");
            PoseidonFullForm ff;
            if (rescale && !arm_env("AKP_POSEIDON_NO_FULL_FORM") && poseidon_full_form(opt, t, full_rounds, partial_rounds, alpha, p->mds,
                    ff)) {
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
void poseidon_unpin(akp_poseidon* p) {
    if (p && --p->pins == 0 && p->destroy_pending) akp_poseidon_params_destroy(p);
}
extern "C" void akp_poseidon_params_destroy(akp_poseidon* p) {
    if (!p) return;
    if (p->pins > 0) {  // a tree or a sponge still computes with it: freed by its last unpin
        p->destroy_pending = true;
        return;
    }
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
        bool nb = st[(head + 62) % 80] ^ st[(head + 51) % 80] ^ st[(head + 38) % 80] ^ st[(head + 23) % 80]
                   ^ st[(head + 13) % 80] ^ st[head];
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
    static const bool enabled = !arm_env("AKP_POSEIDON_NO_REG_T");
    return enabled && (p->dims.t >= 4 && p->dims.t <= 9) && !generic_coop(n) && (c.scaled == 3u || c.scaled == 2u) && c.sparse != nullptr;
}
template <u32 T>
static void launch_reg_permute(const akp_poseidon* p, const PoseidonConsts& c, Fr* d_states, size_t n, hipStream_t s) {
    const dim3 grid((unsigned)((n + 255) / 256));
    if (c.scaled == 3u) hipLaunchKernelGGL((poseidon_permute_reg_kernel<T, true>), grid, dim3(256), 0, s, p->dims, c, d_states, n);
    else hipLaunchKernelGGL((poseidon_permute_reg_kernel<T, false>), grid, dim3(256), 0, s, p->dims, c, d_states, n);
}
template <u32 T>
static void launch_reg_crh(const akp_poseidon* p, const PoseidonConsts& c, const Fr* in0, const Fr* in1, size_t k, Fr* d_out, size_t n,
        hipStream_t s) {
    const dim3 grid((unsigned)((n + 255) / 256));
    if (c.scaled == 3u) hipLaunchKernelGGL((poseidon_crh_reg_kernel<T, true>), grid, dim3(256), 0, s, p->dims, c, in0, in1, k, d_out, n);
    else hipLaunchKernelGGL((poseidon_crh_reg_kernel<T, false>), grid, dim3(256), 0, s, p->dims, c, in0, in1, k, d_out, n);
}
static size_t coop_lds(u32 t) {
    const size_t bytes = (size_t)2 * t * 9 * 64 * sizeof(u32);
    if (bytes > 65536) {  // t = 15, 16: above the default 64 KiB of dynamic LDS per workgroup (gfx950 has 160 KiB per CU);
        // set on every such launch: the attribute belongs to the current device's copy of the kernel
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(poseidon_permute_coop_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                (int)bytes);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(poseidon_crh_coop_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                (int)bytes);
    }
    return bytes;
}
// Resident waves per SIMD of the t = 3 register kernels.  Their 78-81 VGPRs would allow six; a dynamic-LDS request of
// 160 KiB / w per workgroup caps it at w workgroups per CU (one wave of each per SIMD).  Four measured 0.5-1 % faster than
// six at every batch size (profiles/r02_s46) and makes the 4096 workgroups of a 2^20-state launch exactly four rounds.
// A compile-time choice since round 3 (the A/B knob of round 2 is gone).
static unsigned t3_lds_cap() {
    constexpr unsigned w = 4;
    return (160u * 1024u / w) & ~1023u;
}
int32_t launch_permute(akp_poseidon* p, Fr* d_states, size_t n, hipStream_t s, bool host_memory) {
    if (n == 0) return AKP_OK;
    if (n > AKP_MAX_BATCH) return fail(AKP_ERR_BAD_PARAMS, "batch of %zu items exceeds the supported 2^36", n);
    if (p->dims.t == 3 && n > coop_max_items()) {
        const PoseidonConsts c = t3_reg_consts(p);
        // states in pinned HOST memory (the zero-copy path of akp_poseidon_permute_batch): whole-line transfers through an LDS tile.
        // A/B against six 16-byte accesses per lane (profiles/r04_s6/poseidon_hostpath_staged_ab.txt): 2^20 states 3.10 -> 3.01 ms,
        // 2^22 states 10.85 -> 10.11 ms (37 -> 40 GB/s in each direction at once: the duplex limit of the link).
        if (host_memory && ((uintptr_t)d_states & 15u) == 0) {
            if (c.scaled == 3u) hipLaunchKernelGGL(poseidon_permute_t3_staged_kernel<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), t3_lds_cap(), s,
                    p->dims, c, d_states, n);
            else hipLaunchKernelGGL(poseidon_permute_t3_staged_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), t3_lds_cap(), s, p->dims, c,
                    d_states, n);
            HIP_TRY(hipGetLastError());
            return AKP_OK;
        }
        if (c.scaled == 3u) hipLaunchKernelGGL(poseidon_permute_t3_kernel<true>, dim3((unsigned)((n + 255) / 256)), dim3(256),
                t3_lds_cap(), s, p->dims, c, d_states, n);
        else hipLaunchKernelGGL(poseidon_permute_t3_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), t3_lds_cap(), s, p->dims,
                c, d_states, n);
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
        hipLaunchKernelGGL(poseidon_permute_coop_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64 * p->dims.t), coop_lds(p->dims.t), s,
                p->dims, t3_consts(p), d_states, n);
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
int32_t launch_crh(akp_poseidon* p, const Fr* in0, const Fr* in1, size_t k, Fr* d_out, size_t n, hipStream_t s) {
    if (n == 0) return AKP_OK;
    if (n > AKP_MAX_BATCH) return fail(AKP_ERR_BAD_PARAMS, "batch of %zu items exceeds the supported 2^36", n);
    if (p->dims.t == 3 && n > coop_max_items()) {
        const PoseidonConsts c = t3_reg_consts(p);
        if (c.scaled == 3u) hipLaunchKernelGGL(poseidon_crh_t3_kernel<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), t3_lds_cap(), s,
                p->dims, c, in0, in1, k, d_out, n);
        else hipLaunchKernelGGL(poseidon_crh_t3_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), t3_lds_cap(), s, p->dims, c,
                in0, in1, k, d_out, n);
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
        hipLaunchKernelGGL(poseidon_crh_coop_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64 * p->dims.t), coop_lds(p->dims.t), s,
                p->dims, t3_consts(p), in0, in1, k, d_out, n);
        HIP_TRY(hipGetLastError());
        return AKP_OK;
    }
    const unsigned B = poseidon_block(p->dims.t);
    const size_t lds = poseidon_lds(p->dims.t, B);
    const unsigned grid = (unsigned)((n + B - 1) / B);
    if (B == 256) hipLaunchKernelGGL(poseidon_crh_kernel<256>, dim3(grid), dim3(B), lds, s, p->dims, file_consts(p), in0, in1, k, d_out, n);
    else if (B == 128) hipLaunchKernelGGL(poseidon_crh_kernel<128>, dim3(grid), dim3(B), lds, s, p->dims, file_consts(p), in0, in1, k,
            d_out, n);
    else hipLaunchKernelGGL(poseidon_crh_kernel<64>, dim3(grid), dim3(B), lds, s, p->dims, file_consts(p), in0, in1, k, d_out, n);
    HIP_TRY(hipGetLastError());
    return AKP_OK;
}
// Path::verify for m paths in ONE launch (poseidon_verify_paths_t3_kernel); *done = false when the shape is outside that kernel
// (t != 3, rate != 2, more than two leaf elements, parameter sets of different constant forms, or a batch small enough for the
// latency kernels to win level by level): the caller then runs the level-by-level form.
int32_t launch_verify_paths_t3(akp_poseidon* leafp, akp_poseidon* two, const Fr* d_leaves, size_t leaf_len, const uint64_t* d_idx, const Fr* d_sibs,
        const Fr* d_auth, size_t depth, const Fr* d_root, uint8_t* d_ok, size_t m, hipStream_t s, bool* done) {
    *done = false;
    // A/B against the level-by-level form, 2^16 paths of a 2^20-leaf tree (profiles/r04_s4): 3.96 ms in this one kernel against
    // 21 x (0.19 ms hash + 0.005 ms select) = 4.2 ms; 4.78 against 5.04 ms device time with the staging copies.  At 2^16 paths the
    // machine holds ONE wave per SIMD either way, and a lone wave issues a dependent multiply-add every 8.25 cycles (4.4 with two
    // waves): the chain of 21 permutations per lane is the floor, not the launches.
    auto fits = [](const akp_poseidon* p) { return p->dims.t == 3 && p->dims.rate == 2 && p->dims.capacity == 1 && p->dims.full_rounds >= 2; };
    if (!fits(leafp) || !fits(two) || leaf_len < 1 || leaf_len > 2 || m <= coop_max_items() || depth > 62) return AKP_OK;
    const PoseidonConsts cl = t3_reg_consts(leafp), ct = t3_reg_consts(two);
    if ((cl.scaled == 3u) != (ct.scaled == 3u)) return AKP_OK;
    const dim3 grid((unsigned)((m + 255) / 256));
    if (cl.scaled == 3u)
        hipLaunchKernelGGL(poseidon_verify_paths_t3_kernel<true>, grid, dim3(256), t3_lds_cap(), s, leafp->dims, cl, two->dims, ct, d_leaves,
                (u32)leaf_len, d_idx, d_sibs, d_auth, (u32)depth, d_root, d_ok, m);
    else
        hipLaunchKernelGGL(poseidon_verify_paths_t3_kernel<false>, grid, dim3(256), t3_lds_cap(), s, leafp->dims, cl, two->dims, ct, d_leaves,
                (u32)leaf_len, d_idx, d_sibs, d_auth, (u32)depth, d_root, d_ok, m);
    HIP_TRY(hipGetLastError());
    *done = true;
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
                           [&](void* const* di, void*, size_t cnt, hipStream_t s) -> int32_t {
                               return launch_permute(p, (Fr*)di[0], cnt, s, device_alias(states, 16) == di[0]);  // true on the zero-copy path
                           });
}
extern "C" int32_t akp_poseidon_crh_batch_dev(akp_poseidon* p, const uint64_t* d_inputs, size_t n, size_t k, uint64_t* d_out,
        void* stream) {
    NEED_DEV(p, "akp_poseidon_crh_batch_dev");
    return launch_crh(p, (const Fr*)d_inputs, nullptr, k, (Fr*)d_out, n, pick_stream(p->ctx, stream));
}
extern "C" int32_t akp_poseidon_crh_batch(akp_poseidon* p, const uint64_t* inputs, size_t n, size_t k, uint64_t* out) {
    NEED_DEV(p, "akp_poseidon_crh_batch");
    if (n == 0) return AKP_OK;
    if (!out || (!inputs && k)) return fail(AKP_ERR_BAD_PARAMS, "NULL buffer");
    const HostIn in[1] = {{inputs, k * sizeof(Fr), SCR_A}};
    return pipelined_batch(p->ctx, n, in, 1, out, sizeof(Fr), SCR_B, [&](void* const* di, void* dout, size_t cnt,
            hipStream_t s) -> int32_t {
        return launch_crh(p, (const Fr*)di[0], nullptr, k, (Fr*)dout, cnt, s);
    });
}
// ---- ragged batches: every input has its own number of elements (round 5; crh/poseidon/mod.rs:30-40, merkle_tree/mod.rs:411-422) ----
// item i = elements [d_offsets[i], d_offsets[i+1]) of d_inputs.  t = 3 parameter sets only (the register kernel with per-lane
// lengths; other widths go through the host entry point, which groups the items by length).  Scratch: SCR_H for the launch order.
int32_t poseidon_crh_ragged_dev(akp_poseidon* p, const Fr* d_inputs, const uint64_t* d_offsets, size_t n, Fr* d_out, hipStream_t s) {
    if (n == 0) return AKP_OK;
    if (p->dims.t != 3) return fail(AKP_ERR_BAD_PARAMS, "ragged Poseidon batches on device buffers need t = 3 (got t = %u): use akp_poseidon_crh_batch_ragged", p->dims.t);
    if (n >= ((size_t)1 << 32)) return fail(AKP_ERR_BAD_PARAMS, "batch of %zu items exceeds the supported 2^32 - 1", n);
    const u32* order = nullptr;
    if (n >= 4096) {
        void* work = nullptr;
        if (int32_t rc = ctx_scratch(p->ctx, SCR_H, (2 * n + 2 * RAGGED_MAX_KEYS) * sizeof(u32), &work, s)) return rc;
        u32* d_order = (u32*)work + n + 2 * RAGGED_MAX_KEYS;
        HIP_TRY(ragged_order(d_offsets, n, RaggedKey{2u, p->dims.rate, 0u}, (u32*)work, d_order, s));
        order = d_order;
    }
    const PoseidonConsts c = t3_reg_consts(p);
    const dim3 grid((unsigned)((n + 255) / 256));
    if (c.scaled == 3u) hipLaunchKernelGGL(poseidon_crh_ragged_t3_kernel<true>, grid, dim3(256), t3_lds_cap(), s, p->dims, c, d_inputs, d_offsets, order, d_out, n);
    else hipLaunchKernelGGL(poseidon_crh_ragged_t3_kernel<false>, grid, dim3(256), t3_lds_cap(), s, p->dims, c, d_inputs, d_offsets, order, d_out, n);
    HIP_TRY(hipGetLastError());
    return AKP_OK;
}
extern "C" int32_t akp_poseidon_crh_batch_ragged_dev(akp_poseidon* p, const uint64_t* d_inputs, const uint64_t* d_offsets, size_t n, uint64_t* d_out,
        void* stream) {
    NEED_DEV(p, "akp_poseidon_crh_batch_ragged_dev");
    if (n && (!d_offsets || !d_out)) return fail(AKP_ERR_BAD_PARAMS, "NULL buffer");
    return poseidon_crh_ragged_dev(p, (const Fr*)d_inputs, d_offsets, n, (Fr*)d_out, pick_stream(p->ctx, stream));
}
extern "C" int32_t akp_poseidon_crh_batch_ragged(akp_poseidon* p, const uint64_t* inputs, const uint64_t* offsets, size_t n, uint64_t* out) {
    NEED_DEV(p, "akp_poseidon_crh_batch_ragged");
    if (n == 0) return AKP_OK;
    if (!offsets || !out) return fail(AKP_ERR_BAD_PARAMS, "NULL buffer");
    for (size_t i = 0; i < n; ++i)
        if (offsets[i + 1] < offsets[i]) return fail(AKP_ERR_BAD_PARAMS, "offsets[%zu] > offsets[%zu]: offsets must not decrease", i, i + 1);
    const size_t total = (size_t)(offsets[n] - offsets[0]);
    if (total && !inputs) return fail(AKP_ERR_BAD_PARAMS, "inputs is NULL");
    akp_ctx* c = p->ctx;
    if (p->dims.t == 3) {
        hipStream_t s = c->stream;
        void *di = nullptr, *doff = nullptr, *dout = nullptr;
        if (int32_t rc = ctx_scratch(c, SCR_A, std::max<size_t>(total, 1) * sizeof(Fr), &di, s)) return rc;
        if (int32_t rc = ctx_scratch(c, SCR_G, (n + 1) * sizeof(uint64_t), &doff, s)) return rc;
        if (int32_t rc = ctx_scratch(c, SCR_B, n * sizeof(Fr), &dout, s)) return rc;
        if (total) HIP_TRY(hipMemcpyAsync(di, inputs + 4 * offsets[0], total * sizeof(Fr), hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(doff, offsets, (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, s));
        if (int32_t rc = poseidon_crh_ragged_dev(p, (const Fr*)di - offsets[0], (const uint64_t*)doff, n, (Fr*)dout, s)) return rc;
        HIP_TRY(hipMemcpyAsync(out, dout, n * sizeof(Fr), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        return AKP_OK;
    }
    // other widths: the items grouped by length on the host, one uniform batch per distinct length (a sponge of rate r sees few)
    std::vector<std::pair<uint64_t, size_t>> by_len(n);
    for (size_t i = 0; i < n; ++i) by_len[i] = {offsets[i + 1] - offsets[i], i};
    std::sort(by_len.begin(), by_len.end());
    std::vector<uint64_t> in_buf, out_buf;
    for (size_t a = 0; a < n;) {
        size_t b = a;
        while (b < n && by_len[b].first == by_len[a].first) ++b;
        const size_t k = (size_t)by_len[a].first, cnt = b - a;
        in_buf.resize(std::max<size_t>(cnt * k * 4, 1));
        out_buf.resize(cnt * 4);
        for (size_t j = 0; j < cnt; ++j)
            if (k) memcpy(in_buf.data() + j * k * 4, inputs + 4 * offsets[by_len[a + j].second], k * sizeof(Fr));
        if (int32_t rc = akp_poseidon_crh_batch(p, in_buf.data(), cnt, k, out_buf.data())) return rc;
        for (size_t j = 0; j < cnt; ++j) memcpy(out + 4 * by_len[a + j].second, out_buf.data() + 4 * j, sizeof(Fr));
        a = b;
    }
    return AKP_OK;
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
    return pipelined_batch(p->ctx, n, in, 2, out, sizeof(Fr), SCR_C, [&](void* const* di, void* dout, size_t cnt,
            hipStream_t s) -> int32_t {
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
    poseidon_pin(p);            // the sponge keeps computing with these parameters ...
    ++p->ctx->live_handles;     // ... on this context: both outlive it
    *out = s;
    return AKP_OK;
}
extern "C" void akp_sponge_destroy(akp_sponge* s) {
    if (!s) return;
    akp_ctx* c = s->p->ctx;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    if (s->d_state) (void)hipFree(s->d_state);
    if (s->d_io) (void)hipFree(s->d_io);
    poseidon_unpin(s->p);
    ctx_handle_released(c);
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
static int32_t sponge_permute(akp_sponge* s, hipStream_t st) { return launch_permute(s->p, s->d_state, s->batch, st); }
// absorb_internal (sponge/poseidon/mod.rs:124-153)
static int32_t sponge_absorb_internal(akp_sponge* s, u32 idx, size_t k, const Fr* d_elems, hipStream_t st) {
    const PoseidonDims& D = s->p->dims;
    size_t e0 = 0, remaining = k;
    for (;;) {
        const bool last = idx + remaining <= D.rate;
        const u32 count = last ? (u32)remaining : D.rate - idx;
        if (count) {
            const size_t work = s->batch * count;
            hipLaunchKernelGGL(sponge_add_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, st, s->d_state, D.t,
                               D.capacity + idx, d_elems, k, e0, count, s->batch);
            HIP_TRY(hipGetLastError());
        }
        if (last) {
            s->mode = 0;
            s->index = idx + (u32)remaining;
            return AKP_OK;
        }
        if (int32_t rc = sponge_permute(s, st)) return rc;
        e0 += count;
        remaining -= count;
        idx = 0;
    }
}
// CryptographicSponge::absorb (sponge/poseidon/mod.rs:236-257) on elements that are already on the device, enqueued on `st`
static int32_t sponge_absorb_core(akp_sponge* s, const Fr* d_elems, size_t k, hipStream_t st) {
    if (s->mode == 0) {  // :243-250
        u32 idx = s->index;
        if (idx == s->p->dims.rate) {
            if (int32_t rc = sponge_permute(s, st)) return rc;
            idx = 0;
        }
        return sponge_absorb_internal(s, idx, k, d_elems, st);
    }
    return sponge_absorb_internal(s, 0, k, d_elems, st);  // :251-255 no permutation between squeeze and absorb
}
extern "C" int32_t akp_sponge_absorb(akp_sponge* s, const uint64_t* elems, size_t k) {
    if (!s) return fail(AKP_ERR_BAD_PARAMS, "sponge is NULL");
    if (k == 0) return AKP_OK;  // :238-240
    if (!elems) return fail(AKP_ERR_BAD_PARAMS, "elems is NULL");
    NEED_DEV(s->p, "akp_sponge_absorb");
    if (int32_t rc = sponge_io(s, s->batch * k)) return rc;
    hipStream_t st = s->p->ctx->stream;
    HIP_TRY(hipMemcpyAsync(s->d_io, elems, s->batch * k * sizeof(Fr), hipMemcpyHostToDevice, st));
    if (int32_t rc = sponge_absorb_core(s, s->d_io, k, st)) return rc;
    HIP_TRY(hipStreamSynchronize(st));
    return AKP_OK;
}
extern "C" int32_t akp_sponge_absorb_dev(akp_sponge* s, const uint64_t* d_elems, size_t k, void* stream) {
    if (!s) return fail(AKP_ERR_BAD_PARAMS, "sponge is NULL");
    if (k == 0) return AKP_OK;
    if (!d_elems) return fail(AKP_ERR_BAD_PARAMS, "d_elems is NULL");
    NEED_DEV(s->p, "akp_sponge_absorb_dev");
    return sponge_absorb_core(s, (const Fr*)d_elems, k, pick_stream(s->p->ctx, stream));
}
// squeeze_internal (sponge/poseidon/mod.rs:156-186)
static int32_t sponge_squeeze_internal(akp_sponge* s, u32 idx, size_t n_out, Fr* d_out, hipStream_t st) {
    const PoseidonDims& D = s->p->dims;
    size_t o0 = 0, remaining = n_out;
    for (;;) {
        const bool last = idx + remaining <= D.rate;
        const u32 count = last ? (u32)remaining : D.rate - idx;
        if (count) {
            const size_t work = s->batch * count;
            hipLaunchKernelGGL(sponge_copy_out_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, st, s->d_state, D.t,
                               D.capacity + idx, d_out, n_out, o0, count, s->batch);
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
            if (int32_t rc = sponge_permute(s, st)) return rc;
        idx = 0;
    }
}
// squeeze_native_field_elements (sponge/poseidon/mod.rs:324-344) into a device buffer, enqueued on `st`
static int32_t sponge_squeeze_core(akp_sponge* s, Fr* d_out, size_t n_out, hipStream_t st) {
    if (s->mode == 0) {  // :331-334 (permutes even when n_out == 0)
        if (int32_t rc = sponge_permute(s, st)) return rc;
        return sponge_squeeze_internal(s, 0, n_out, d_out, st);
    }
    u32 idx = s->index;  // :335-341
    if (idx == s->p->dims.rate) {
        if (int32_t rc = sponge_permute(s, st)) return rc;
        idx = 0;
    }
    return sponge_squeeze_internal(s, idx, n_out, d_out, st);
}
extern "C" int32_t akp_sponge_squeeze(akp_sponge* s, uint64_t* out, size_t n_out) {
    if (!s) return fail(AKP_ERR_BAD_PARAMS, "sponge is NULL");
    if (!out && n_out) return fail(AKP_ERR_BAD_PARAMS, "out is NULL");
    NEED_DEV(s->p, "akp_sponge_squeeze");
    if (int32_t rc = sponge_io(s, s->batch * std::max<size_t>(n_out, 1))) return rc;
    hipStream_t st = s->p->ctx->stream;
    if (int32_t rc = sponge_squeeze_core(s, s->d_io, n_out, st)) return rc;
    if (n_out) HIP_TRY(hipMemcpyAsync(out, s->d_io, s->batch * n_out * sizeof(Fr), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return AKP_OK;
}
extern "C" int32_t akp_sponge_squeeze_dev(akp_sponge* s, uint64_t* d_out, size_t n_out, void* stream) {
    if (!s) return fail(AKP_ERR_BAD_PARAMS, "sponge is NULL");
    if (!d_out && n_out) return fail(AKP_ERR_BAD_PARAMS, "d_out is NULL");
    NEED_DEV(s->p, "akp_sponge_squeeze_dev");
    return sponge_squeeze_core(s, (Fr*)d_out, n_out, pick_stream(s->p->ctx, stream));
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

