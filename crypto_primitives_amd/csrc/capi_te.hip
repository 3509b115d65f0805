// capi_te.hip -- part of libakp.so (implementation of include/akp.h): Pedersen / Bowe-Hopwood over Jubjub: tables, batch entry points
// Product code.  Never includes, links or calls anything under oracle/; there is no CPU fallback for any compute entry
// point (a missing device is AKP_ERR_HIP).
#include "capi_internal.hpp"
#include <chrono>
#include "te_kernels.hpp"
#include "te_shape.hpp"
#include "ragged_sort.hpp"

// ------------------------------------------------------------------------------------------
// Pedersen / Bowe-Hopwood
// Width of the precomputed tables.  A hash is one 7-product curve addition per table step, so the only way to shorten it is
// to make a step cover more message bits -- each extra bit doubles the table.  Rounds 1-3 kept the tables inside the 256 MiB
// Infinity Cache (Pedersen 4x256: 16-bit digits, 64 steps, 268 MB; Bowe-Hopwood 63x9: groups of 5 chunks, 237 MB).  Round 4
// measured what the 288 GB of HBM buy (profiles/r04_s10): a step whose entry comes from HBM instead of the cache costs
// 49 -> 57 us per 2^20 hashes, and there are fewer of them --
//   Pedersen 4x256, 2^20 x 128 B:  D = 16 / 20 / 22 / 24:  3.16 / 2.84 / 2.62 / 2.43 ms  (268 MB / 3.5 / 12.6 / 46 GB)
//   Bowe-Hopwood 63x9, 2^20 x 64 B: G = 5 / 6 / 7 / 8:     1.81 / 1.73 / 1.57 / 1.44 ms  (237 MB / 1.6 / 10.9 / 75 GB)
// The width is the widest the context's TABLE BUDGET admits (akp_ctx_set_table_budget; default 320 MiB = the cache-sized tables), or the
// explicit shape of akp_te_params_create_shaped.  Digits up to 24 bits / groups up to 8 chunks: 25-bit digits (the 32-bit message
// window of a step would admit them: 41 steps, 88 GB) were measured SLOWER, 2.65 against 2.41 ms -- a digit's region of the table is
// then 2 GB and the gather leaves the TLB reach (64.6 against 55.9 us per step; profiles/r04_s10/digit25_rejected.txt, gather_probe.txt).
//
// Round 6 -- who pays for a wide table, and when (profiles/r06_s1 .. r06_s7).  The kernels of a 46 GB build take 62 ms, but the ALLOCATION is
// not ours to bound: the first 46 GB taken from VRAM nobody had used since the box came up cost 1.2 s (the driver's BENCH_r05 and
// profiles/r06_s1; 0.3 ms on every later lease of the same machine), and a hipMalloc that follows a large hipFree waits for the kernel
// driver's wipe of the released memory (4 s after freeing 128 GB, ~35 GB/s: tools/vram_probe.hip) -- while launches of OTHER threads go on
// undisturbed (worst 0.8 ms during a 4 s stall).  Hence:
//   * a handle whose BUDGET admits a table wider than the default starts on the cache-sized table (built in ~1 ms on the caller's thread,
//     as always) and a thread of the library allocates and builds the wide one on a stream of its own; calls switch when it is complete
//     (te_pick).  akp_te_params_prepare is the blocking form.  Explicit shapes (akp_te_params_create_shaped) are built by the caller, as
//     before: that caller asked for exactly this table.
//   * nothing a launch may have been given is freed while a handle is attached: an extension allocates the new table, builds it and RETIRES
//     the old one (as superseded constants are), to be freed with the table object.  A graph captured after `prepare` stays valid
//     (ADVICE r05), and the free-then-allocate of round 5's extension -- exactly the pattern that stalls -- is gone.
static size_t te_device_quarter() {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) {
        (void)hipGetLastError();
        return (size_t)320 << 20;
    }
    return std::max<size_t>(total_b / 4, (size_t)64 << 20);
}
constexpr size_t TE_DEFAULT_BUDGET = (size_t)320 << 20;  // tables that stay inside the 256 MiB Infinity Cache and build in milliseconds
// the budget a handle's SHAPE is derived from: a function of the setting and the device, not of the memory that happens to be free at
// this moment (two workers creating the same parameters must file them under the same shape; whether the table fits is decided when it
// is built -- te_ensure_table narrows it or gives the upgrade up)
static size_t te_table_budget(const akp_ctx* ctx) {
    if (!ctx->table_budget) return TE_DEFAULT_BUDGET;
    if (ctx->table_budget != AKP_TABLE_BUDGET_DEVICE) return ctx->table_budget;
    return te_device_quarter();
}
extern "C" int32_t akp_ctx_set_table_budget(akp_ctx* ctx, size_t bytes) {
    if (!ctx) return fail(AKP_ERR_BAD_PARAMS, "NULL context");
    ctx->table_budget = bytes;
    return AKP_OK;
}
extern "C" size_t akp_ctx_table_budget(const akp_ctx* ctx) {
    if (!ctx || ctx->dead) return 0;
    (void)hipSetDevice(ctx->device);
    return te_table_budget(ctx);
}
constexpr u32 TE_MAX_DIGIT = te_shape::MAX_DIGIT, TE_MAX_GROUP = te_shape::MAX_GROUP;
static inline size_t te_pedersen_entries(size_t n_gen, u32 D) { return te_shape::pedersen_entries(n_gen, D); }
static inline size_t te_bh_entries(size_t n_gen, u32 G) { return te_shape::bh_entries(n_gen, G); }
static inline double te_now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// ---- storage of the wide table -------------------------------------------------------------------------------------------------
// A hipMalloc of the current coverage; an extension allocates the new coverage, builds ALL of it and RETIRES the old block (launches
// already enqueued -- or captured into a graph -- may still read it: it is freed with the table, when no handle is left).  To bound what
// retiring costs, a table is extended at most once: the first build covers what the first messages need, the second covers everything
// (te_ensure_table).
// (Measured and not used, profiles/r06_s2, r06_s8: the virtual-memory API -- one reserved range, physical memory mapped as the table
// grows, so that an extension would build only the new units in place.  Gathers from a mapped range run at the hipMalloc rate (8.65 G
// lines/s) and 46 GB tables built fine, but hipMemSetAccess on the second piece of a range failed with "invalid argument" for some
// range sizes on this stack (tools/vmm2_probe.hip) and one probe run ended in a GPU memory fault: not something to put under every
// hash of a library.)
static hipError_t te_storage_grow(TeTable* t, size_t bytes, TeEntry** previous) {
    TeEntry* fresh = nullptr;
    const hipError_t e = hipMalloc(&fresh, bytes);
    if (e != hipSuccess) return e;
    *previous = t->d_lut;  // retired by the caller once the new block is filled
    t->d_lut = fresh;
    return hipSuccess;
}
// the block just allocated could not be filled: back to the previous one (units_built still describes it); the unfilled block is retired
// (hipFree would wait for the whole device)
static void te_storage_undo(TeTable* t, TeEntry* previous) {
    if (t->d_lut) t->retired.push_back(t->d_lut);
    t->d_lut = previous;
}
// a narrowed shape gives its table up (te_narrow): kept until the table object goes, like everything a launch may have been given
static void te_storage_retire(TeTable* t) {
    if (t->d_lut) t->retired.push_back(t->d_lut);
    t->d_lut = nullptr;
    t->units_built = 0;
}

// workgroups of a background build's combine kernel.  Sweep (profiles/r06_s34; first 2^20-hash Pedersen batch / first 2^23-leaf
// Bowe-Hopwood tree beside the build, default budget 4.6 / 26.1 ms): 32 -> 4.9 / 32.9 ms (wide table in use after 0.57 / 0.28 s),
// 64 -> 3.7 / 34.9 ms, 128 -> 9.7 / 37.4 ms (0.35 / 0.11 s), 256 -> 8.7 / 44.5 ms.  What the build takes in all does not depend on the
// width: a 2^26-leaf tree that saturates the device for 0.17 s beside it takes ~0.03 s longer at every width -- the 22.5 GB table's
// 30 ms of machine time have to come from somewhere (non-temporal stores for the table: no difference, r06_s35).
static const unsigned TE_BG_WGS = env_u32("AKP_TE_BG_WGS", 64, 1, 1u << 20);
// two-part construction of a wide table (te_kernels.hpp): part tables entry by entry (for ALL units up to `units`: kilobytes to
// megabytes, < 0.5 ms), then one addition per wide entry of the units [from, units) (`from` is 0 today: every build fills a fresh block).
// KIND 2: Pedersen signed-subset table of W-bit digits over the halved generators `src`; KIND 1: Bowe-Hopwood table of groups of W chunks.
template <int KIND>
static hipError_t te_build_wide(TeTable* t, const void* src, u32 n_gen, u32 W, u32 from, u32 units, akp_te_build_report* rep) {
    hipStream_t bs = t->active_stream;
    const u32 k_lo = KIND == 2 ? (W - 1) / 2 : W / 2;
    const u32 unit_shift = KIND == 2 ? W - 1 : 3 * W - 1;
    const size_t n_lo = KIND == 2 ? (size_t)units << k_lo : (size_t)units << (3 * k_lo - 1);
    const size_t n_hi = KIND == 2 ? (size_t)units << (W - 1 - k_lo) : (size_t)units << (3 * (W - k_lo));
    const size_t first = (size_t)from << unit_shift, entries = (size_t)units << unit_shift;
    TeEntry *lo = nullptr, *hi = nullptr;
    double t0 = te_now_ms();
    hipError_t e = hipMalloc(&lo, n_lo * sizeof(TeEntry));
    if (e == hipSuccess) e = hipMalloc(&hi, n_hi * sizeof(TeEntry));
    rep->alloc_ms += te_now_ms() - t0;
    if (e == hipSuccess) {
        t0 = te_now_ms();
        const unsigned pgrid = (unsigned)((n_lo + n_hi + 63) / 64);
        if (KIND == 2)
            hipLaunchKernelGGL(te_build_pedersen_sparts, dim3(pgrid), dim3(64), 0, bs, (const NielsPad*)src, n_gen, W, units, k_lo, lo, hi);
        else
            hipLaunchKernelGGL(te_build_bh_parts, dim3(pgrid), dim3(64), 0, bs, (const Fr*)src, W, k_lo, units, lo, hi);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(bs);
        rep->parts_ms += te_now_ms() - t0;
    }
    if (e == hipSuccess) {
        t0 = te_now_ms();
        unsigned cgrid = (unsigned)((entries - first + 256 * AKP_TE_BUILD_RUN - 1) / (256 * AKP_TE_BUILD_RUN));
        // A build that runs BESIDE hashing (the background upgrade) must not take the machine from it.  Stream priority alone does not do
        // it (waves that are resident keep their slots: the first 2^23-leaf tree beside the build took 53 ms instead of 27, profiles/r06_s10);
        // capping the build's occupancy through unused LDS costs every compute unit a share (2^26-leaf tree 234 against 179 ms, r06_s11); a
        // stream with a CU mask did it (r06_s13) -- until the first out-of-memory hipMalloc of the process, which then hung or crashed inside
        // the runtime (r06_s14 .. s16).  So the build simply launches FEW workgroups, each walking many tiles: TE_BG_WGS of them hold that
        // many x 4 of the device's 8192 wave slots.  It takes several times as long; nobody waits for it.
        if (rep->in_background) cgrid = std::min(cgrid, TE_BG_WGS);
        hipLaunchKernelGGL(te_build_combine_kernel<KIND>, dim3(cgrid), dim3(256), 0, bs, lo, hi, W, k_lo, first, entries, t->d_lut);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(bs);
        rep->combine_ms += te_now_ms() - t0;
    }
#if defined(AKP_TEST_HOOKS)
    // AKP_TE_TABLE_CHECK=k: every k-th entry of the table against the per-entry definition
    const size_t step = env_size("AKP_TE_TABLE_CHECK", 0);
    if (e == hipSuccess && step) {
        u32* d_bad = nullptr;
        u32 bad = 0;
        e = hipMalloc(&d_bad, sizeof(u32));
        if (e == hipSuccess) e = hipMemsetAsync(d_bad, 0, sizeof(u32), bs);
        if (e == hipSuccess) {
            const size_t cnt = (entries + step - 1) / step;
            hipLaunchKernelGGL(te_check_table_kernel<KIND>, dim3((unsigned)((cnt + 63) / 64)), dim3(64), 0, bs, src, n_gen, W, entries, (size_t)0, step,
                    t->d_lut, d_bad);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(&bad, d_bad, sizeof(u32), hipMemcpyDeviceToHost, bs);
        if (e == hipSuccess) e = hipStreamSynchronize(bs);
        if (d_bad) t->retired.push_back(d_bad);
        if (e == hipSuccess && bad) {
            fprintf(stderr, "akp: AKP_TE_TABLE_CHECK: %u of the sampled table entries differ from the per-entry definition\n", bad);
            e = hipErrorAssert;
        }
    }
#endif
    // the part tables (megabytes) are RETIRED, not freed: hipFree waits for the whole device -- a caller building its cache-sized table here
    // would wait for the background build of the wide one (first call 112 ms instead of 5: profiles/r06_s31)
    if (lo) t->retired.push_back(lo);
    if (hi) t->retired.push_back(hi);
    return e;
}
// ---- the process-wide table store (TeTable, capi_internal.hpp) ---------------------------------------------------------------
static std::mutex g_store_mu;
static std::vector<TeTable*> g_store;
// The streams table builds run on: ONE pair per device for the life of the process -- `build` (default priority) for a build some caller
// waits for, `bg` (lowest priority) for the background upgrade.  Round 6 first gave every table streams of its own; a test-suite creates
// and drops hundreds of tables, and the stream churn (each new lowest-priority stream can bring a hardware queue into being: a 28 ms
// hiccup beside a first call) bought nothing: builds of different tables are rare enough to share a queue.
struct TeDevStreams {
    hipStream_t build = nullptr, bg = nullptr;
};
static TeDevStreams g_dev_streams[64];
static std::mutex g_dev_streams_mu;
static hipError_t te_dev_streams(int device, bool want_bg, hipStream_t* build, hipStream_t* bg) {  // the device is current
    std::lock_guard<std::mutex> lk(g_dev_streams_mu);
    TeDevStreams& d = g_dev_streams[device & 63];
    if (!d.build) {
        const hipError_t e = hipStreamCreateWithFlags(&d.build, hipStreamNonBlocking);
        if (e != hipSuccess) return e;
    }
    if (want_bg && !d.bg) {
        int lo_prio = 0, hi_prio = 0;
        if (hipDeviceGetStreamPriorityRange(&lo_prio, &hi_prio) != hipSuccess ||
            hipStreamCreateWithPriority(&d.bg, hipStreamNonBlocking, lo_prio) != hipSuccess) {
            (void)hipGetLastError();
            d.bg = nullptr;  // the builder falls back to the build stream
        }
    }
    *build = d.build;
    *bg = d.bg;
    return hipSuccess;
}
static std::vector<TeTable*> g_all_tables;  // every live table (also those that left the store): the exit hook joins their builders
static void te_join_builders_at_exit() {
    // a build thread that is still allocating when the process exits would meet a runtime that is being torn down: wait for it
    std::vector<TeTable*> all;
    {
        std::lock_guard<std::mutex> lk(g_store_mu);
        all = g_all_tables;
    }
    for (TeTable* t : all)
        if (t->builder.joinable()) t->builder.join();
}
static void te_table_free(TeTable* t) {  // refs == 0: no handle, no launch can name it any more
    if (t->builder.joinable()) t->builder.join();
    {
        std::lock_guard<std::mutex> lk(g_store_mu);
        g_all_tables.erase(std::remove(g_all_tables.begin(), g_all_tables.end(), t), g_all_tables.end());
    }
    (void)hipSetDevice(t->device);
    (void)hipDeviceSynchronize();
    if (t->d_lut) (void)hipFree(t->d_lut);
    for (void* p : t->retired) (void)hipFree(p);
    if (t->d_lut1) (void)hipFree(t->d_lut1);
    if (t->d_gens) (void)hipFree(t->d_gens);
    if (t->d_half) (void)hipFree(t->d_half);
    for (int i = 0; i < t->n_tails; ++i)
        if (t->tails[i].d) (void)hipFree(t->tails[i].d);
    for (int i = 0; i < t->n_rem; ++i)
        if (t->rem[i].d) (void)hipFree(t->rem[i].d);
    delete t;
}
static void te_table_release(TeTable* t) {
    if (!t) return;
    {
        std::lock_guard<std::mutex> lk(g_store_mu);
        if (--t->refs > 0) return;
        g_store.erase(std::remove(g_store.begin(), g_store.end(), t), g_store.end());
    }
    te_table_free(t);
}
// everything a fresh table needs before its first hash: kilobytes and a few small kernels (the wide table itself is built by
// te_ensure_table for the message lengths that arrive or that akp_te_params_prepare names).  Caller holds t->mu.
static hipError_t te_table_init(TeTable* t, const uint64_t* gens, u32 shape, size_t budget) {
    const size_t n_gen = t->n_gen;
    constexpr size_t max_entries = (size_t)1 << 32;  // entry indices are 32-bit in the kernels
    Fr* d_g = nullptr;
    // builds a CALLER waits for run on `build_stream` (default priority); the background upgrade uses `bg_stream` (lowest priority: it
    // yields the compute units to the hashing it runs beside) -- `active_stream` is whichever the thread that holds t->mu builds on
    // (the background stream of a wide table is looked up HERE, by the thread that creates the handle, not by the builder beside the
    // handle's first call: the process's first lowest-priority stream brings a hardware queue into being)
    hipError_t e = te_dev_streams(t->device, t->shape_auto && budget > TE_DEFAULT_BUDGET, &t->build_stream, &t->bg_stream);
    t->active_stream = t->build_stream;
    hipStream_t bs = t->build_stream;
    if (e == hipSuccess) e = hipMalloc(&d_g, n_gen * 2 * sizeof(Fr));
    if (e == hipSuccess) e = hipMemcpy(d_g, gens, n_gen * 2 * sizeof(Fr), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (d_g) (void)hipFree(d_g);
        return e;
    }
    if (t->pedersen) {
        // Signed-subset table (te_kernels.hpp): needs every generator in the prime-order subgroup (checked on the device: 2 (G/2) == G),
        // stores 2^(D-1) entries per digit.  The plain table stays as the fallback for generators outside the subgroup; as an A/B
        // arm (AKP_PEDERSEN_PLAIN=1) it is selectable only in the test build (-DAKP_TEST_HOOKS).
        NielsPad* d_half = nullptr;
        u32* d_bad = nullptr;
        u32 bad = 1;
#if defined(AKP_TEST_HOOKS)
        const bool force_plain = getenv("AKP_PEDERSEN_PLAIN") != nullptr;
#else
        const bool force_plain = false;
#endif
        if (!force_plain) {
            if (e == hipSuccess) e = hipMalloc(&d_half, n_gen * sizeof(NielsPad));
            if (e == hipSuccess) e = hipMalloc(&d_bad, sizeof(u32));
            if (e == hipSuccess) e = hipMemsetAsync(d_bad, 0, sizeof(u32), bs);
            if (e == hipSuccess) {
                hipLaunchKernelGGL(te_halve_generators, dim3((unsigned)((n_gen + 63) / 64)), dim3(64), 0, bs, d_g, (u32)n_gen, d_half, d_bad);
                e = hipGetLastError();
            }
            if (e == hipSuccess) e = hipMemcpyAsync(&bad, d_bad, sizeof(u32), hipMemcpyDeviceToHost, bs);
            if (e == hipSuccess) e = hipStreamSynchronize(bs);
        }
        if (e == hipSuccess && bad == 0) {
            const u32 D = shape;
            if (te_pedersen_entries(n_gen, D) >= max_entries) e = hipErrorInvalidValue;
            const size_t n_digits = (n_gen + D - 1) / D;
            t->digit_bits = D;
            t->signed_subset = true;
            t->units_total = (u32)n_digits;
            if (e == hipSuccess) e = hipMalloc(&t->d_lut1, (n_digits + 1) * sizeof(TeEntry));
            if (e == hipSuccess) {
                hipLaunchKernelGGL(te_build_pedersen_cprefix, dim3((unsigned)((n_digits + 1 + 63) / 64)), dim3(64), 0, bs, d_half, (u32)n_gen, D,
                        (u32)n_digits, t->d_lut1);
                e = hipGetLastError();
            }
            if (e == hipSuccess) e = hipStreamSynchronize(bs);
            t->d_half = d_half;
            d_half = nullptr;
        } else if (e == hipSuccess) {
            // plain table (generators outside the prime-order subgroup: never what `setup` produces): entry by entry, digits of
            // at most 14 bits (4x256: 13 bits, 79 steps, 93 MB)
            u32 D = t->shape_auto ? 13 : std::min(shape, 14u);
            while (D > 1 && ((n_gen + D - 1) / D) * ((size_t)1 << D) * sizeof(TeEntry) > std::min(budget, (size_t)192 << 20) && t->shape_auto) --D;
            t->digit_bits = D;
            const size_t entries = ((n_gen + D - 1) / D) << D;
            e = hipMalloc(&t->d_lut, entries * sizeof(TeEntry));
            if (e == hipSuccess) {
                hipLaunchKernelGGL(te_build_pedersen_lut, dim3((unsigned)((entries + 63) / 64)), dim3(64), 0, bs, d_g, (u32)n_gen, D, (u32)entries, t->d_lut);
                e = hipGetLastError();
            }
        }
        if (d_half) t->retired.push_back(d_half);  // (hipFree would wait for the whole device: another table's background build, say)
        if (d_bad) t->retired.push_back(d_bad);
    } else {
        const u32 G = shape;
        if (G > 1 && te_bh_entries(n_gen, G) >= max_entries) e = hipErrorInvalidValue;
        if (e == hipSuccess) e = hipMalloc(&t->d_lut1, n_gen * 4 * sizeof(TeEntry));
        if (e == hipSuccess) {
            hipLaunchKernelGGL(te_build_bh_lut, dim3((unsigned)((n_gen * 4 + 63) / 64)), dim3(64), 0, bs, d_g, (u32)n_gen, t->d_lut1);
            e = hipGetLastError();
        }
        if (G > 1) t->units_total = (u32)(n_gen / G);
        t->group = G;
        t->d_gens = d_g;  // kept: the group table and the remainder tables of later message lengths are built from them
        d_g = nullptr;
    }
    if (e == hipSuccess) e = hipStreamSynchronize(bs);
    if (d_g) t->retired.push_back(d_g);
    return e;
}
// The table of (device, kind, window, generators, shape): found in the store or entered as a PLACEHOLDER and initialised under its own
// lock -- the store's lock is held for the lookup only, so creating handles of unrelated parameters (or on other devices) never waits
// for device work, and two threads asking for the same parameters at the same moment still end up with ONE table (the second waits on
// the table's lock until the first has initialised it).
static int32_t te_store_attach(akp_ctx* ctx, bool ped, u32 W, u32 N, const uint64_t* gens, u32 key_shape, bool shape_auto, size_t budget, TeTable** out) {
    const size_t n_gen = (size_t)W * N, words = n_gen * 8;
    TeTable* t = nullptr;
    bool mine = false;
    std::unique_lock<std::mutex> init_lock;
    {
        std::lock_guard<std::mutex> lk(g_store_mu);
        for (TeTable* c : g_store)
            if (c->device == ctx->device && c->pedersen == ped && c->W == W && c->N == N && c->key_shape == key_shape && c->shape_auto == shape_auto &&
                c->gens.size() == words && memcmp(c->gens.data(), gens, words * sizeof(uint64_t)) == 0) {
                t = c;
                break;
            }
        if (!t) {
            t = new TeTable();
            t->device = ctx->device; t->pedersen = ped; t->W = W; t->N = N; t->n_gen = (u32)n_gen;
            t->shape_auto = shape_auto; t->key_shape = key_shape;
            t->gens.assign(gens, gens + words);
            t->in_store = true;
            g_store.push_back(t);
            static bool exit_hook = false;
            if (!exit_hook) {
                exit_hook = true;
                std::atexit(te_join_builders_at_exit);
            }
            g_all_tables.push_back(t);
            mine = true;
            init_lock = std::unique_lock<std::mutex>(t->mu);  // taken before the store lock goes: nobody sees the placeholder unlocked
        }
        ++t->refs;
    }
    if (mine) {
        const hipError_t e = te_table_init(t, gens, key_shape, budget);
        t->initialised = true;
        if (e != hipSuccess) {
            t->init_rc = AKP_ERR_HIP;
            t->init_err = hipGetErrorString(e);
            (void)hipGetLastError();
        }
        init_lock.unlock();
    } else {
        std::lock_guard<std::mutex> wait_init(t->mu);
    }
    if (t->init_rc) {
        const std::string why = t->init_err;
        {
            std::lock_guard<std::mutex> lk(g_store_mu);  // a table that failed to initialise is not offered again
            if (t->in_store) {
                g_store.erase(std::remove(g_store.begin(), g_store.end(), t), g_store.end());
                t->in_store = false;
            }
        }
        te_table_release(t);
        return fail(AKP_ERR_HIP, "akp_te_params_create: %s", why.c_str());
    }
    *out = t;
    return AKP_OK;
}
extern "C" void akp_te_params_destroy(akp_te_params* p);
extern "C" int32_t akp_te_params_create_shaped(akp_ctx* ctx, int32_t kind, uint32_t W, uint32_t N, const uint64_t* gens, uint32_t shape,
        akp_te_params** out) {
    if (!ctx) return fail(AKP_ERR_HIP, "akp_te_params_create: a device context is required (tables are built on the GPU)");
    if (!out || !gens) return fail(AKP_ERR_BAD_PARAMS, "NULL argument");
    if (ctx->dead) return fail(AKP_ERR_BAD_PARAMS, "akp_te_params_create: the context was destroyed");
    if (kind != AKP_TE_PEDERSEN && kind != AKP_TE_BOWE_HOPWOOD && kind != AKP_TE_PEDERSEN_X) return fail(AKP_ERR_BAD_PARAMS,
            "unknown kind %d", kind);
    if (W == 0 || N == 0) return fail(AKP_ERR_BAD_PARAMS, "empty window");
    if (kind == AKP_TE_BOWE_HOPWOOD && W > 63) return fail(AKP_ERR_BAD_PARAMS,
            "Bowe-Hopwood window size %u > 63 (bowe_hopwood/mod.rs:81-101)", W);
    if (shape > (kind == AKP_TE_BOWE_HOPWOOD ? TE_MAX_GROUP : TE_MAX_DIGIT) || (shape == 1 && kind != AKP_TE_BOWE_HOPWOOD))
        return fail(AKP_ERR_BAD_PARAMS, "table shape %u: Pedersen digits have 2..%u bits, Bowe-Hopwood groups 1..%u chunks (0: from the table budget)",
                shape, TE_MAX_DIGIT, TE_MAX_GROUP);
    const size_t n_gen = (size_t)W * N;
    if (n_gen > (1u << 22)) return fail(AKP_ERR_BAD_PARAMS, "window %ux%u too large", W, N);
    for (size_t i = 0; i < 2 * n_gen; ++i)
        if (!fr_words_reduced(gens + 4 * i)) return fail(AKP_ERR_BAD_PARAMS, "generator coordinate %zu not reduced", i);
    HIP_TRY(hipSetDevice(ctx->device));
#if defined(AKP_TEST_HOOKS)
    if (!shape) shape = kind == AKP_TE_BOWE_HOPWOOD ? env_u32("AKP_BH_GROUP", 0, 1, TE_MAX_GROUP) : env_u32("AKP_PEDERSEN_DIGIT_BITS", 0, 1, TE_MAX_DIGIT);
#endif
    const bool ped = kind != AKP_TE_BOWE_HOPWOOD;
    const bool shape_auto = shape == 0;
    const size_t budget = te_table_budget(ctx);
    auto shape_for = [&](size_t b) {
        u32 k = ped ? te_shape::pick_digit(n_gen, b, sizeof(TeEntry)) : te_shape::pick_group(n_gen, b, sizeof(TeEntry));
        if (!ped && n_gen < k) k = (u32)n_gen;
        return k;
    };
    // the shape the table is filed under: what was asked for, or what the budget admits
    u32 key_shape = shape ? (ped ? std::max(shape, 2u) : shape) : shape_for(budget);
    if (!ped && n_gen < key_shape) key_shape = (u32)n_gen;
    // a budget above the default: the handle starts on the cache-sized table and moves to the wide one when that is complete
    const u32 start_shape = shape_auto ? shape_for(std::min(budget, TE_DEFAULT_BUDGET)) : key_shape;
    akp_te_params* p = new akp_te_params();
    p->ctx = ctx; p->kind = kind; p->W = W; p->N = N; p->n_gen = (u32)n_gen;
    if (int32_t rc = te_store_attach(ctx, ped, W, N, gens, start_shape, shape_auto, std::min(budget, TE_DEFAULT_BUDGET), &p->t)) {
        delete p;
        return rc;
    }
    // (a Pedersen table over generators outside the prime-order subgroup is the plain one: nothing wider exists for it)
    if (shape_auto && key_shape > start_shape && !(ped && !p->t->signed_subset)) {
        if (int32_t rc = te_store_attach(ctx, ped, W, N, gens, key_shape, shape_auto, budget, &p->wide)) {
            te_table_release(p->t);
            delete p;
            return rc;
        }
    }
    ++ctx->live_handles;
    *out = p;
    return AKP_OK;
}
extern "C" int32_t akp_te_params_create(akp_ctx* ctx, int32_t kind, uint32_t W, uint32_t N, const uint64_t* gens, akp_te_params** out) {
    return akp_te_params_create_shaped(ctx, kind, W, N, gens, 0, out);
}
extern "C" uint32_t akp_te_entry_bytes(void) { return (uint32_t)sizeof(TeEntry); }
void te_unpin(akp_te_params* p) {
    if (p && --p->pins == 0 && p->destroy_pending) akp_te_params_destroy(p);
}
extern "C" void akp_te_params_destroy(akp_te_params* p) {
    if (!p) return;
    if (p->pins > 0) {  // a tree still computes with it: freed by its last unpin
        p->destroy_pending = true;
        return;
    }
    te_table_release(p->wide);  // the tables go with the LAST handle attached to them (a running build is waited for, the device drained)
    te_table_release(p->t);
    ctx_handle_released(p->ctx);
    delete p;
}

// table steps a message of msg_len bytes touches (later digits are zero / absent): Pedersen pads with zero
// bytes (those digits select the identity); Bowe-Hopwood stops at ceil(bits/3) chunks = `groups` triples +
// left-over singles.
static void te_steps(const TeTable* t, size_t msg_len, u32* n_groups, u32* n_steps) {
    if (t->pedersen) {
        *n_groups = 0;
        te_shape::pedersen_steps(t->n_gen, t->digit_bits, msg_len, n_steps);
    } else {
        te_shape::bh_steps(t->n_gen, t->group, msg_len, n_groups, n_steps);
    }
}
static size_t te_table_bytes(const TeTable* t) {
    size_t entries;
    if (t->pedersen) {
        const size_t n_digits = (t->n_gen + t->digit_bits - 1) / t->digit_bits;
        entries = t->signed_subset ? ((size_t)t->units_built << (t->digit_bits - 1)) + n_digits + 1 : n_digits << t->digit_bits;
    } else {
        entries = (size_t)t->n_gen * 4 + (t->group > 1 ? ((size_t)t->units_built << (3 * t->group - 1)) : 0);
        for (int i = 0; i < t->n_rem; ++i) entries += (size_t)1 << (3 * t->rem[i].r);
        entries += t->n_tails;
    }
    return entries * sizeof(TeEntry);
}
// workgroup size of te_accumulate_lds_kernel for messages of data_len bytes at a pitch of `stride`: the largest whose LDS image
// fits 40 KB (128-byte pitch: 256 messages = 33.8 KB, four workgroups per CU); 0 = pitch above 640 bytes or an empty / padded
// message: the per-lane global loads of te_accumulate_kernel.  A/B of the two kernels on resident messages
// (profiles/r04_s1): Pedersen 4x256 3.294 -> 3.189 ms, Bowe-Hopwood 64 B 1.824 -> 1.762 ms per 2^20 hashes.
static unsigned te_lds_block(size_t data_len, size_t stride) {
    if (data_len < 4) return 0;
    unsigned block = 256;
    while (block >= 64 && te_lds_image_bytes(block, data_len, stride) > 40960) block /= 2;
    return block >= 64 ? block : 0;
}
// the constant of a zero-padded Bowe-Hopwood tail instead of walking the padding chunk by chunk (-1.0 ms on a 2^23-leaf tree,
// profiles/r02_s31); the other arm (AKP_BH_ZERO_TAIL=0, read per call) exists only in the test build
static inline bool te_zero_tail_on() {
#if defined(AKP_TEST_HOOKS)
    return env_u32("AKP_BH_ZERO_TAIL", 1, 0, 1) != 0;
#else
    return true;
#endif
}
static const size_t te_split_max = env_size("AKP_TE_SPLIT_MAX", (size_t)1 << 14);  // one workgroup per CU
// `pipe` (host-pointer entry point with pinned buffers): the batch runs chunk by chunk -- kernels on `s`, the DMA copy-in of later
// chunks on pipe->cin, the DMA copy-out of chunk i on pipe->side under the kernels of chunk i + 1.  Every buffer (messages,
// xyz, prefix, digests) is sized for the whole batch, so no chunk waits for a buffer and all copies are issued up front.
// Nothing synchronises here; the caller waits for the three streams.
struct TePipe {
    size_t chunk;
    const uint8_t* h_msgs;  // pinned host source, copied to d_msgs chunk-wise by DMA
    void* h_out;            // pinned host destination, filled from d_out chunk-wise by DMA
    hipStream_t cin, side;
    hipEvent_t ev_in, ev_acc;
};
// Table work (build / extend / remainder / tail constant) allocates and waits: legal from any entry point EXCEPT while the caller's
// stream is being captured into a graph -- there the caller must have prepared the table (akp_te_params_prepare) before capture began.
static bool te_capturing(hipStream_t s) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return st == hipStreamCaptureStatusActive;
}
// how a call may treat a table that lacks something: build it (and wait), or report TE_NOT_READY (the caller falls back to the
// cache-sized table of the handle)
enum TeBuild { TE_LOOK = 0, TE_BUILD = 1 };
constexpr int32_t TE_NOT_READY = -1;  // internal: never returned through the C ABI
static int32_t te_missing(TeBuild mode, hipStream_t s, const char* what) {
    if (mode == TE_LOOK) return TE_NOT_READY;
    if (te_capturing(s))
        return fail(AKP_ERR_BAD_PARAMS, "%s needs a table that is not built and the stream is being captured: call akp_te_params_prepare before the capture", what);
    return AKP_OK;
}
// A table whose shape came from the table budget narrows it when the device cannot hold the table after all: one bit / one chunk
// less, everything that depends on the shape rebuilt (the old constants and the old range are RETIRED: a launch may hold them).  The
// narrowed table leaves the store (its shape no longer says what a new handle with this budget would get); the handles attached keep it.
// Caller holds t->mu.
static hipError_t te_narrow(TeTable* t) {
    {
        std::lock_guard<std::mutex> lk(g_store_mu);
        if (t->in_store) {
            g_store.erase(std::remove(g_store.begin(), g_store.end(), t), g_store.end());
            t->in_store = false;
        }
    }
    te_storage_retire(t);
    if (t->pedersen) {
        const u32 D = --t->digit_bits;
        const size_t n_digits = ((size_t)t->n_gen + D - 1) / D;
        t->units_total = (u32)n_digits;
        if (t->d_lut1) t->retired.push_back(t->d_lut1);
        t->d_lut1 = nullptr;
        hipError_t e = hipMalloc(&t->d_lut1, (n_digits + 1) * sizeof(TeEntry));
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(te_build_pedersen_cprefix, dim3((unsigned)((n_digits + 1 + 63) / 64)), dim3(64), 0, t->active_stream, t->d_half, t->n_gen, D,
                (u32)n_digits, t->d_lut1);
        e = hipGetLastError();
        return e == hipSuccess ? hipStreamSynchronize(t->active_stream) : e;
    }
    --t->group;
    t->units_total = t->n_gen / t->group;
    for (int i = 0; i < t->n_rem; ++i)
        if (t->rem[i].d) t->retired.push_back(t->rem[i].d);  // their first chunk follows the group size
    t->n_rem = 0;
    return hipSuccess;
}
// The wide table covers the digits / chunk groups [0, units_built); a message that needs more extends it: a NEW allocation is built
// (te_storage_grow) and the old block retired -- nothing a launch was given is freed, nothing is drained.  The first build covers what the
// first messages need, an extension the complete table: at most one block is ever retired.  62 ms of kernel time for the 46 GB of a whole
// 4x256 table, milliseconds for the cache-sized default or the prefix a tree's 32- and 64-byte nodes use -- plus whatever the
// allocation costs on this box at this moment (the comment at the top).  Returns the table steps of a data_len-byte message (te_steps)
// for the shape the table ends up with.  Caller holds t->mu; the build runs on the table's own stream and is complete when this returns.
static int32_t te_ensure_table(TeTable* t, size_t data_len, u32* groups, u32* steps, hipStream_t s, TeBuild mode, bool background = false) {
    const bool ped = t->pedersen;
    for (;;) {
        te_steps(t, data_len, groups, steps);
        const u32 needed = ped ? *steps : *groups;
        if (needed <= t->units_built || (ped ? !t->signed_subset : t->group <= 1)) return AKP_OK;  // (plain table / single chunks: complete)
        if (int32_t rc = te_missing(mode, s, "the curve hash")) return rc;
        const u32 shape = ped ? t->digit_bits : t->group;
        const bool can_narrow = t->shape_auto && shape > (ped ? 8u : 2u);
        auto bytes_of = [&](u32 units) { return ((size_t)units << (ped ? shape - 1 : 3 * shape - 1)) * sizeof(TeEntry); };
        // the first build covers what this message needs, an extension everything (the old block is retired, not freed: at most once)
        u32 target = t->builds ? t->units_total : te_shape::grow_target(needed, t->units_built, t->units_total);
        akp_te_build_report rep{};
        rep.in_background = background ? 1u : 0u;
        const double t_begin = te_now_ms();
        hipError_t e = hipSuccess;
        if (t->shape_auto) {  // at most half of what is free now
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
                if (bytes_of(target) > free_b / 2) target = needed;
                if (bytes_of(target) > free_b / 2 && can_narrow) e = hipErrorOutOfMemory;
            } else {
                (void)hipGetLastError();
            }
        }
        double t0 = te_now_ms();
        TeEntry* previous = nullptr;
        const bool try_grow = e == hipSuccess;
        if (try_grow) e = te_storage_grow(t, bytes_of(target), &previous);
        const bool grown = try_grow && e == hipSuccess;
        rep.alloc_ms = te_now_ms() - t0;
        if (e == hipErrorOutOfMemory && can_narrow) {
            (void)hipGetLastError();
            e = te_narrow(t);
            if (e == hipSuccess) continue;
        }
        if (e == hipSuccess)
            e = ped ? te_build_wide<2>(t, t->d_half, t->n_gen, t->digit_bits, 0, target, &rep)
                    : te_build_wide<1>(t, t->d_gens, t->n_gen, t->group, 0, target, &rep);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            if (grown) te_storage_undo(t, previous);
            return fail(AKP_ERR_HIP, "curve table of %zu MB (%u of %u %s): %s -- lower akp_ctx_set_table_budget and create the handle again",
                    bytes_of(target) >> 20, target, t->units_total, ped ? "digits" : "chunk groups", hipGetErrorString(e));
        }
        if (previous) t->retired.push_back(previous);  // launches already enqueued (or captured) may still read it: it goes with the table
        const u32 from = 0;
        rep.table_bytes = bytes_of(target);
        rep.shape = shape;
        rep.units_from = from;
        rep.units_to = target;
        rep.units_total = t->units_total;
        rep.in_background = background ? 1u : 0u;
        rep.total_ms = te_now_ms() - t_begin;
        t->last_build = rep;
        t->units_built = target;
        ++t->builds;
        return AKP_OK;
    }
}
// Bowe-Hopwood: table of the r (1 .. 7) chunks starting at chunk `first` -- what a message shape leaves after its last full
// group -- with the constant of the zero-padded tail chunks [tail_from, tail_to) folded into every entry, so that those chunks AND
// the tail are ONE table step instead of r + 1 additions; built on the first use of the shape (2^(3r) entries: at most 268 MB,
// milliseconds).  *out stays NULL when the table's slots are taken or memory is short: the chunks are then single steps
// from the one-chunk table and the tail its own addition, as before round 4.  A 63x9 tree node of 64 data bytes is 21 + 1
// additions instead of 21 + 3 + 1 with groups of eight, a 32-byte leaf 10 + 1 instead of 10 + 6.  Caller holds t->mu.
static int32_t te_bh_remainder(TeTable* t, u32 first, u32 r, u32 tail_from, u32 tail_to, const TeEntry** out, hipStream_t s, TeBuild mode) {
    *out = nullptr;
    if (tail_from >= tail_to) tail_from = tail_to = 0;
    for (int i = 0; i < t->n_rem; ++i)
        if (t->rem[i].first == first && t->rem[i].r == r && t->rem[i].tail_from == tail_from && t->rem[i].tail_to == tail_to) {
            *out = t->rem[i].d;
            return AKP_OK;
        }
    if (t->n_rem == TeTable::MAX_REMAINDERS || !t->d_gens) return AKP_OK;
    if (int32_t rc = te_missing(mode, s, "the Bowe-Hopwood hash of a new message shape")) return rc;
    const double t0 = te_now_ms();
    hipStream_t bs = t->active_stream;
    const size_t entries = (size_t)1 << (3 * r);
    TeEntry *d = nullptr, *d_t = nullptr;
    hipError_t e = hipMalloc(&d, entries * sizeof(TeEntry));
    if (e == hipSuccess && tail_from < tail_to) {
        e = hipMalloc(&d_t, sizeof(TeEntry));
        if (e == hipSuccess) {
            hipLaunchKernelGGL(te_bh_tail_kernel, dim3(1), dim3(64), 0, bs, t->d_lut1, tail_from, tail_to, d_t);
            e = hipGetLastError();
        }
    }
    if (e == hipSuccess) {
        hipLaunchKernelGGL(te_build_bh_remainder, dim3((unsigned)((entries + 63) / 64)), dim3(64), 0, bs, t->d_gens, first, r, d_t, (u32)entries, d);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(bs);
    if (d_t) t->retired.push_back(d_t);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        if (d) t->retired.push_back(d);
        if (e == hipErrorOutOfMemory) return AKP_OK;
        return fail(AKP_ERR_HIP, "Bowe-Hopwood remainder table: %s", hipGetErrorString(e));
    }
    t->rem[t->n_rem++] = TeTable::Remainder{first, r, tail_from, tail_to, d};
    t->last_build.constants_ms += te_now_ms() - t0;
    *out = d;
    return AKP_OK;
}
// Bowe-Hopwood: the constant of the zero-padded tail chunks [from, to) as ONE entry (shapes without a remainder table); one entry per
// shape, computed once and never rewritten.  Caller holds t->mu.
static int32_t te_bh_tail(TeTable* t, u32 from, u32 to, const TeEntry** out, hipStream_t s, TeBuild mode) {
    for (int i = 0; i < t->n_tails; ++i)
        if (t->tails[i].from == from && t->tails[i].to == to) {
            *out = t->tails[i].d;
            return AKP_OK;
        }
    if (int32_t rc = te_missing(mode, s, "the Bowe-Hopwood hash of a new two-to-one shape")) return rc;
    if (t->n_tails == TeTable::MAX_TAILS) {  // more shapes than slots (no real parameter set does this): the old entries are retired (128 bytes each)
        for (int i = 0; i < t->n_tails; ++i) t->retired.push_back(t->tails[i].d);
        t->n_tails = 0;
    }
    TeEntry* d = nullptr;
    HIP_TRY(hipMalloc(&d, sizeof(TeEntry)));
    hipLaunchKernelGGL(te_bh_tail_kernel, dim3(1), dim3(64), 0, t->active_stream, t->d_lut1, from, to, d);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(t->active_stream);
    if (e != hipSuccess) {
        (void)hipFree(d);
        return fail(AKP_ERR_HIP, "Bowe-Hopwood tail constant: %s", hipGetErrorString(e));
    }
    t->tails[t->n_tails++] = TeTable::Tail{from, to, d};
    *out = d;
    return AKP_OK;
}
// What one launch needs from the table for messages of msg_len bytes of which the first data_len carry data: built on demand
// (TE_BUILD) or looked up (TE_LOOK: TE_NOT_READY when something is missing).  Caller holds t->mu and keeps it until the kernels that
// use these pointers are enqueued.  `want_rem` false: the ragged kernels take left-over chunks as single steps (no remainder table,
// no tail: every item has its own length).
struct TeResolved {
    TeTable* t = nullptr;
    u32 shape = 0, groups = 0, steps = 0;
    const TeEntry *lut = nullptr, *lut1 = nullptr, *tail = nullptr;
};
static int32_t te_resolve(TeTable* t, size_t msg_len, size_t data_len, hipStream_t s, TeResolved* r, TeBuild mode, bool want_rem = true, bool background = false) {
    if (int32_t rc = te_ensure_table(t, data_len, &r->groups, &r->steps, s, mode, background)) return rc;
    // what the kernels call D: the digit width, or the chunks per group with the size of the remainder step above it (te_bh_rem)
    r->t = t;
    r->shape = t->pedersen ? t->digit_bits : t->group;
    r->lut = t->d_lut;
    r->lut1 = t->d_lut1;
    r->tail = nullptr;
    if (!want_rem) return AKP_OK;
    u32 tail_from = 0, tail_to = 0;
    if (!t->pedersen && data_len < msg_len) {
        tail_from = (u32)std::min<size_t>((data_len * 8 + 2) / 3, t->n_gen);
        tail_to = (u32)std::min<size_t>((msg_len * 8 + 2) / 3, t->n_gen);
    }
    bool tail_folded = false;
    if (!t->pedersen && t->group > 1 && r->steps > r->groups) {  // chunks after the last full group: one step, tail included
        const TeEntry* rem = nullptr;
        if (int32_t rc = te_bh_remainder(t, t->group * r->groups, r->steps - r->groups, tail_from, tail_to, &rem, s, mode)) return rc;
        if (rem) {
            r->shape |= (r->steps - r->groups) << 8;
            r->lut1 = rem;
            r->steps = r->groups + 1;
            tail_folded = true;
        }
    }
    if (tail_from < tail_to && !tail_folded)
        if (int32_t rc = te_bh_tail(t, tail_from, tail_to, &r->tail, s, mode)) return rc;
    return AKP_OK;
}
// ---- the background upgrade ----------------------------------------------------------------------------------------------------
// One build at a time per table: whoever finds the wide table lacking what its call needs starts the thread (if none is running) and
// hashes on the cache-sized table meanwhile; a request that arrives while a build runs is dropped -- the next call of that shape asks
// again.  The thread holds the table's lock for the whole build: callers TRY the lock (te_pick) and take the cache-sized table when it
// is held.  A failed build (memory) marks the table: no further attempts, the handles stay where they are.
static void te_upgrade_kick(TeTable* w, size_t msg_len, size_t data_len, bool want_rem) {
    if (w->upgrade_failed.load()) return;
    bool idle = false;
    if (!w->building.compare_exchange_strong(idle, true)) return;
    if (w->builder.joinable()) w->builder.join();  // the previous build has ended (`building` was false): only its thread object is left
    try {
    w->builder = std::thread([w, msg_len, data_len, want_rem] {
        (void)hipSetDevice(w->device);
        {
            std::lock_guard<std::mutex> lk(w->mu);
            // lowest priority + few workgroups (te_build_wide); without a low-priority stream: the build stream
            if (w->bg_stream) w->active_stream = w->bg_stream;
            TeResolved r;
            int32_t rc = AKP_OK;
            // a tree announced its inner-node shape (te_tree_prepare): that one FIRST, so that the leaf level's request -- which arrives
            // first and needs fewer units -- does not cause a build and then an extension
            if (w->hint_set.exchange(false)) rc = te_resolve(w, w->hint_msg_len, w->hint_data_len, w->active_stream, &r, TE_BUILD, true, true);
            if (rc == AKP_OK) rc = te_resolve(w, msg_len, data_len, w->active_stream, &r, TE_BUILD, want_rem, true);
            if (rc != AKP_OK) {
                w->upgrade_error = akp_last_error();  // (this thread's: nobody else would see it)
                w->upgrade_failed.store(true);
            }
            w->active_stream = w->build_stream;
        }
        w->building.store(false);
    });
    } catch (...) {  // no thread to be had: the handle stays on the cache-sized table, a later call asks again
        w->building.store(false);
    }
}
// The table this call hashes on, resolved, its lock held in `lk` (until the kernels that use the pointers are enqueued): the wide
// table of the handle when it has everything the call needs, else the table the handle started on -- built on demand as ever -- with
// a request to the builder on the way.
// The request to the builder goes out when the call that found the wide table lacking is DONE with its own allocations and launches
// (the destructor of this object, declared before the table lock in every caller): allocations of two threads stall together when the
// driver is busy wiping or mapping memory, and the call's own -- the cache-sized table on its first use, scratch -- must not queue up
// behind the 46 GB the builder is about to ask for (first call 1.1-1.8 s instead of 5 ms: profiles/r06_s31).
struct TeKickLater {
    TeTable* w = nullptr;
    size_t msg_len = 0, data_len = 0;
    bool want_rem = true;
    ~TeKickLater() {
        if (w) te_upgrade_kick(w, msg_len, data_len, want_rem);
    }
};
static int32_t te_pick(akp_te_params* p, size_t msg_len, size_t data_len, hipStream_t s, TeKickLater& later, std::unique_lock<std::mutex>& lk, TeResolved* r,
        bool want_rem = true) {
    if (TeTable* w = p->wide) {
        lk = std::unique_lock<std::mutex>(w->mu, std::try_to_lock);
        if (lk.owns_lock()) {
            const int32_t rc = te_resolve(w, msg_len, data_len, s, r, TE_LOOK, want_rem);
            if (rc == AKP_OK) return AKP_OK;
            lk.unlock();
            if (rc != TE_NOT_READY) return rc;
            if (!te_capturing(s)) {  // (an allocation on another thread would break a global-mode capture)
                later.w = w;
                later.msg_len = msg_len;
                later.data_len = data_len;
                later.want_rem = want_rem;
            }
        }
    }
    lk = std::unique_lock<std::mutex>(p->t->mu);
    return te_resolve(p->t, msg_len, data_len, s, r, TE_BUILD, want_rem);
}
// blocking form (akp_te_params_prepare): the wide table if the handle has one -- a running background build is waited for, what is
// still missing is built on this thread -- else the handle's only table
static int32_t te_prepare(akp_te_params* p, size_t msg_len, size_t data_len, hipStream_t s) {
    TeResolved r;
    if (TeTable* w = p->wide) {
        std::lock_guard<std::mutex> lk(w->mu);
        const int32_t rc = te_resolve(w, msg_len, data_len, s, &r, TE_BUILD);
        if (rc == AKP_OK) {
            w->upgrade_failed.store(false);
            return AKP_OK;
        }
        w->upgrade_error = akp_last_error();
        w->upgrade_failed.store(true);
        return rc;
    }
    std::lock_guard<std::mutex> lk(p->t->mu);
    return te_resolve(p->t, msg_len, data_len, s, &r, TE_BUILD);
}
extern "C" int32_t akp_te_params_prepare(akp_te_params* p, size_t msg_len) {
    NEED_TE(p, "akp_te_params_prepare");
    if (msg_len * 8 > te_input_bits(p))
        return fail(AKP_ERR_BAD_LENGTH, "incorrect input length %zu for window params %ux%u (the reference panics)", msg_len, p->W, p->N);
    return te_prepare(p, msg_len, msg_len, p->ctx->stream);
}
// ... and what TwoToOneCRH::compress / the inner levels of a tree will need: two serialised digests in the (W*N)/8-byte buffer
static void te_compress_shape(const akp_te_params* p, size_t* buflen, size_t* used) {
    *buflen = ((size_t)p->W * p->N) / 8;
    *used = te_zero_tail_on() ? std::min<size_t>(*buflen, (size_t)2 * te_fe_per_digest(p) * 32) : *buflen;
}
// `s`: the stream the caller is about to enqueue on (only asked whether it is being captured; builds run on the table's stream)
int32_t te_prepare_compress(akp_te_params* p, hipStream_t s) {
    size_t buflen, used;
    te_compress_shape(p, &buflen, &used);
    return te_prepare(p, buflen, used, s);
}
extern "C" int32_t akp_te_params_prepare_compress(akp_te_params* p) {
    NEED_TE(p, "akp_te_params_prepare_compress");
    return te_prepare_compress(p, p->ctx->stream);
}
// A tree hashes its leaves first and its (usually longer) two-to-one buffers second: when both hashes share a table, building it
// for the inner nodes FIRST means one build instead of a build and an extension (the cold first tree: profiles/r05_s2 -> r05_s4).
// The table the handle STARTS on is built here (milliseconds); a budget-chosen wide table gets the same request in the background.
int32_t te_tree_prepare(akp_te_params* leafp, akp_te_params* two, hipStream_t s) {
    size_t buflen, used;
    te_compress_shape(two, &buflen, &used);
    if (two->wide && !te_capturing(s)) {
        // the wide table is asked for by the tree's first hash call, when that call has issued its own work (TeKickLater); what the
        // builder should build FIRST is announced here
        two->wide->hint_msg_len = buflen;
        two->wide->hint_data_len = used;
        two->wide->hint_set.store(true);
    }
    if (leafp->t != two->t) return AKP_OK;
    std::lock_guard<std::mutex> lk(two->t->mu);
    TeResolved r;
    return te_resolve(two->t, buflen, used, s, &r, TE_BUILD);
}
extern "C" int32_t akp_te_params_info(const akp_te_params* p, uint32_t* digit_bits_or_group, int32_t* signed_subset, size_t* table_bytes,
        size_t msg_len,
                                      uint32_t* steps) {
    if (!p) return fail(AKP_ERR_BAD_PARAMS, "akp_te_params_info: params is NULL");
    // the table a call with msg_len-byte messages would use now: the wide one if it covers that length (and no build holds it)
    TeTable* t = p->t;
    std::unique_lock<std::mutex> lk;
    if (p->wide) {
        lk = std::unique_lock<std::mutex>(p->wide->mu, std::try_to_lock);
        u32 g = 0, st = 0;
        if (lk.owns_lock()) {
            te_steps(p->wide, msg_len, &g, &st);
            if (p->wide->units_built && (p->wide->pedersen ? st : g) <= p->wide->units_built) t = p->wide;
            else lk.unlock();
        }
    }
    if (t == p->t) lk = std::unique_lock<std::mutex>(t->mu);
    const bool ped = t->pedersen;
    if (digit_bits_or_group) *digit_bits_or_group = ped ? t->digit_bits : t->group;
    if (signed_subset) *signed_subset = ped && t->signed_subset ? 1 : 0;
    if (table_bytes) *table_bytes = te_table_bytes(t);
    if (steps) {
        u32 g = 0, st = 0;
        te_steps(t, msg_len, &g, &st);
        if (!ped && t->group > 1 && st > g) st = g + 1;  // the chunks after the last full group are one step (te_bh_remainder)
        *steps = st;
    }
    return AKP_OK;
}
extern "C" int32_t akp_te_params_table_info(const akp_te_params* p, uint64_t* table_id, uint32_t* handles_attached, uint64_t* wide_builds,
        akp_te_build_report* last_build) {
    if (!p) return fail(AKP_ERR_BAD_PARAMS, "akp_te_params_table_info: params is NULL");
    TeTable* t = p->wide ? p->wide : p->t;
    if (table_id) *table_id = (uint64_t)(uintptr_t)t;
    if (handles_attached) {
        std::lock_guard<std::mutex> lk(g_store_mu);
        *handles_attached = (uint32_t)t->refs;
    }
    if (wide_builds || last_build) {
        // (a running background build holds the lock for its whole length: report "pending" without waiting for it)
        std::unique_lock<std::mutex> lk(t->mu, std::try_to_lock);
        if (!lk.owns_lock() && !p->wide) lk.lock();
        if (wide_builds) *wide_builds = lk.owns_lock() ? t->builds : 0;
        if (last_build) {
            if (lk.owns_lock()) *last_build = t->last_build;
            else memset(last_build, 0, sizeof *last_build);
            last_build->upgrade_state = !p->wide ? 0u : t->upgrade_failed.load() ? 3u : (lk.owns_lock() && t->units_built && !t->building.load()) ? 2u : 1u;
            memset(last_build->note, 0, sizeof last_build->note);
            if (last_build->upgrade_state == 3 && lk.owns_lock()) snprintf(last_build->note, sizeof last_build->note, "%s", t->upgrade_error.c_str());
        }
    }
    return AKP_OK;
}
// projective -> affine for `cnt` sums.  One inversion is shared among up to 64 messages per lane, but `target` lanes
// stay busy when the range allows.  Measured at 2^20 Pedersen hashes (profiles/r02_s27): 16 K / 32 K / 64 K / 128 K /
// 256 K / 512 K lanes -> 3.57 / 3.41 / 3.39 | 3.21 / 3.24 / 3.29 / 3.46 ms for accumulate + finalize (two boxes): one wave
// per SIMD it is (a compile-time choice since round 3).
// `target`: lanes to keep busy: 65536 = one wave per SIMD.  A pass that runs BESIDE an accumulate kernel (the chunks of the pinned host
// path) is bound by the LATENCY of its dependent chains, not by its instruction count: longer inversion chains on fewer lanes (8192
// / 2048 lanes) measured slower, as did one inversion per message (131072 lanes) -- profiles/r05_s8/gated_sweep.txt.  What helps is
// issue priority: te_finalize_kernel raises its waves' priority (s_setprio), 1.1 -> 0.25 ms per chunk beside the accumulate kernel.
static inline size_t te_chunk_finalize_lanes() {
#if defined(AKP_TEST_HOOKS)
    return env_size("AKP_TE_CHUNK_FINALIZE_LANES", 65536);  // A/B (test build only): 8192 / 2048 lanes (longer inversion chains) measured SLOWER
#else
    return 65536;
#endif
}
static int32_t te_launch_finalize(const akp_te_params* p, const F29Pad* x, F29Pad* pre, Fr* d_out, size_t cnt, hipStream_t st, size_t target = 65536) {
    const size_t chain = std::min<size_t>(64, std::max<size_t>(1, cnt / target));
    const size_t lanes = (cnt + chain - 1) / chain;
    const unsigned fgrid = (unsigned)((lanes + 255) / 256);
    if (p->kind == AKP_TE_PEDERSEN)
        hipLaunchKernelGGL(te_finalize_kernel<0>, dim3(fgrid), dim3(256), 0, st, x, pre, d_out, cnt, lanes);
    else
        hipLaunchKernelGGL(te_finalize_kernel<1>, dim3(fgrid), dim3(256), 0, st, x, pre, d_out, cnt, lanes);
    HIP_TRY(hipGetLastError());
    return AKP_OK;
}
// `pitch`: distance in bytes between consecutive messages in d_msgs (0: msg_len); a pitch below msg_len is legal when the bytes
// past data_len are the implied zero padding (te_compress_dev packs the digest pairs without it)
static int32_t te_crh_run(akp_te_params* p, const uint8_t* d_msgs, size_t n, size_t msg_len, Fr* d_out, hipStream_t s, size_t data_len,
        const TePipe* pipe, size_t pitch = 0);
int32_t te_crh_dev(akp_te_params* p, const uint8_t* d_msgs, size_t n, size_t msg_len, Fr* d_out, hipStream_t s, size_t data_len) {
    return te_crh_run(p, d_msgs, n, msg_len, d_out, s, data_len, nullptr);
}
static int32_t te_crh_run(akp_te_params* p, const uint8_t* d_msgs, size_t n, size_t msg_len, Fr* d_out, hipStream_t s, size_t data_len,
        const TePipe* pipe, size_t pitch) {
    if (msg_len * 8 > te_input_bits(p))
        return fail(AKP_ERR_BAD_LENGTH, "incorrect input length %zu for window params %ux%u (the reference panics)", msg_len, p->W, p->N);
    if (n == 0) return AKP_OK;
    if (n > ((size_t)1 << 32)) return fail(AKP_ERR_BAD_PARAMS, "batch of %zu messages exceeds the supported 2^32", n);
    const bool tail_on = te_zero_tail_on();
    if (data_len > msg_len || !tail_on) data_len = msg_len;
    // the table's pointers are read and the kernels that use them enqueued under its lock (TeTable, capi_internal.hpp): handles of
    // other contexts share the table, and one of them may be extending it right now.  te_pick: the wide table of the handle once it
    // is complete, the cache-sized one until then.
    TeKickLater kick;  // (before the lock: destroyed after it)
    std::unique_lock<std::mutex> table_lock;
    TeResolved rs;
    if (int32_t rc = te_pick(p, msg_len, data_len, s, kick, table_lock, &rs)) return rc;
    TeTable* t = rs.t;
    const u32 shape = rs.shape, groups = rs.groups, steps = rs.steps;
    const TeEntry *lut = rs.lut, *lut1 = rs.lut1, *tail = rs.tail;
    const bool signed_subset = t->signed_subset;
    size_t stride = pitch ? pitch : msg_len;
    if (data_len > 0 && data_len < 4) {  // the kernels fetch message bits with one 32-bit load: pad 1..3-byte messages to four bytes
        void* pad = nullptr;
        if (int32_t rc = ctx_scratch(p->ctx, SCR_L, n * 4, &pad, s)) return rc;
        hipLaunchKernelGGL(te_pad4_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_msgs, stride, (u32)data_len,
                (uint8_t*)pad, n);
        HIP_TRY(hipGetLastError());
        d_msgs = (const uint8_t*)pad;
        stride = 4;
        data_len = 4;  // three zero bytes at most: they select nothing (Pedersen) / lie past the steps counted above (Bowe-Hopwood)
    }
    // small batches (tree tops) are bound by the latency of one message: split each one over AKP_TE_SPLIT waves
    if (n <= te_split_max) {
        const unsigned sgrid = (unsigned)((n + 63) / 64);
        const bool xy = p->kind == AKP_TE_PEDERSEN;  // digest = (x, y); otherwise x only
        if (te_is_pedersen(p) && signed_subset) {
            if (xy) hipLaunchKernelGGL((te_crh_small_kernel<2, false>), dim3(sgrid), dim3(64 * AKP_TE_SPLIT), 0, s, lut, lut1,
                    d_msgs, data_len, stride, shape, groups, steps, tail, d_out, n);
            else hipLaunchKernelGGL((te_crh_small_kernel<2, true>), dim3(sgrid), dim3(64 * AKP_TE_SPLIT), 0, s, lut, lut1,
                    d_msgs, data_len, stride, shape, groups, steps, tail, d_out, n);
        } else if (te_is_pedersen(p)) {
            if (xy) hipLaunchKernelGGL((te_crh_small_kernel<0, false>), dim3(sgrid), dim3(64 * AKP_TE_SPLIT), 0, s, lut, lut1,
                    d_msgs, data_len, stride, shape, groups, steps, tail, d_out, n);
            else hipLaunchKernelGGL((te_crh_small_kernel<0, true>), dim3(sgrid), dim3(64 * AKP_TE_SPLIT), 0, s, lut, lut1,
                    d_msgs, data_len, stride, shape, groups, steps, tail, d_out, n);
        } else {
            hipLaunchKernelGGL((te_crh_small_kernel<1, true>), dim3(sgrid), dim3(64 * AKP_TE_SPLIT), 0, s, lut, lut1, d_msgs,
                    data_len, stride, shape, groups, steps, tail, d_out, n);
        }
        HIP_TRY(hipGetLastError());
        return AKP_OK;
    }
    void *xyz = nullptr, *prefix = nullptr;
    if (int32_t rc = ctx_scratch(p->ctx, SCR_E, n * 3 * sizeof(F29Pad), &xyz, s)) return rc;
    if (int32_t rc = ctx_scratch(p->ctx, SCR_F, n * sizeof(F29Pad), &prefix, s)) return rc;
    const u32 fe = te_fe_per_digest(p);
    // messages [first, first + cnt): extended-coordinate sums into xyz
    auto accumulate = [&](size_t first, size_t cnt, hipStream_t st) -> int32_t {
        const uint8_t* m = d_msgs + first * stride;
        F29Pad* x = (F29Pad*)xyz + first * 3;
        // messages staged through LDS (te_accumulate_lds_kernel): the largest workgroup whose image fits 40 KB (a 128-byte
        // pitch: 256 messages = 33.8 KB, four workgroups per CU); pitches above 640 bytes keep the per-lane global loads
        const unsigned block = te_lds_block(data_len, stride);
        if (block) {
            const unsigned lgrid = (unsigned)((cnt + block - 1) / block);
            const size_t shm = te_lds_image_bytes(block, data_len, stride);
            if (te_is_pedersen(p) && signed_subset)
                hipLaunchKernelGGL(te_accumulate_lds_kernel<2>, dim3(lgrid), dim3(block), shm, st, lut, lut1, m, data_len, stride,
                        shape, groups, steps, tail, x, cnt);
            else if (te_is_pedersen(p))
                hipLaunchKernelGGL(te_accumulate_lds_kernel<0>, dim3(lgrid), dim3(block), shm, st, lut, lut1, m, data_len, stride,
                        shape, groups, steps, tail, x, cnt);
            else
                hipLaunchKernelGGL(te_accumulate_lds_kernel<1>, dim3(lgrid), dim3(block), shm, st, lut, lut1, m, data_len, stride,
                        shape, groups, steps, tail, x, cnt);
            HIP_TRY(hipGetLastError());
            return AKP_OK;
        }
        const unsigned grid = (unsigned)((cnt + 255) / 256);
        if (te_is_pedersen(p) && signed_subset)
            hipLaunchKernelGGL(te_accumulate_kernel<2>, dim3(grid), dim3(256), 0, st, lut, lut1, m, data_len, stride, shape,
                    groups, steps, tail, x, cnt);
        else if (te_is_pedersen(p))
            hipLaunchKernelGGL(te_accumulate_kernel<0>, dim3(grid), dim3(256), 0, st, lut, lut1, m, data_len, stride, shape,
                    groups, steps, tail, x, cnt);
        else
            hipLaunchKernelGGL(te_accumulate_kernel<1>, dim3(grid), dim3(256), 0, st, lut, lut1, m, data_len, stride, shape,
                    groups, steps, tail, x, cnt);
        HIP_TRY(hipGetLastError());
        return AKP_OK;
    };
    // projective -> affine for the same range (te_launch_finalize)
    auto finalize = [&](size_t first, size_t cnt, hipStream_t st) -> int32_t {
        return te_launch_finalize(p, (const F29Pad*)xyz + first * 3, (F29Pad*)prefix + first, d_out + first * fe, cnt, st,
                pipe ? te_chunk_finalize_lanes() : (size_t)65536);
    };
    if (pipe) {
        const size_t dig = fe * sizeof(Fr);
        for (size_t first = 0; first < n; first += pipe->chunk) {
            const size_t cnt = std::min(pipe->chunk, n - first);
            if (pipe->h_msgs) {
                HIP_TRY(hipMemcpyAsync((uint8_t*)d_msgs + first * stride, pipe->h_msgs + first * stride, cnt * stride, hipMemcpyHostToDevice,
                        pipe->cin));
                HIP_TRY(hipEventRecord(pipe->ev_in, pipe->cin));
                HIP_TRY(hipStreamWaitEvent(s, pipe->ev_in, 0));
            }
            if (int32_t rc = accumulate(first, cnt, s)) return rc;
            // the finalize pass stays on `s`: beside an accumulate kernel it takes 0.43 ms instead of 0.055 ms and slows that kernel
            // from 0.40 to 0.58 ms per 2^17 messages (both want the vector ALUs; profiles/r04_s3)
            if (int32_t rc = finalize(first, cnt, s)) return rc;
            HIP_TRY(hipEventRecord(pipe->ev_acc, s));
            HIP_TRY(hipStreamWaitEvent(pipe->side, pipe->ev_acc, 0));
            HIP_TRY(hipMemcpyAsync((char*)pipe->h_out + first * dig, (const char*)d_out + first * dig, cnt * dig, hipMemcpyDeviceToHost,
                    pipe->side));
        }
        return AKP_OK;
    }
    if (int32_t rc = accumulate(0, n, s)) return rc;
    return finalize(0, n, s);
}

// ---- the pinned host path as ONE gated launch (round 5; DESIGN.md section 1, profiles/r05_s8, r05_s13, r05_s16) -----------------------
// Until round 4 a pinned batch ran as eight chunked launches, each behind its chunk's DMA: every launch pays its ramp (0.45-0.5 ms per
// 2^17 messages against 0.29 ms of the resident launch's share) and the call sat at that floor (4.2 ms per 2^20 Pedersen hashes
// against 2.3-3.2 ms resident).  Here the accumulate kernel is launched ONCE over the whole batch while the messages are still
// arriving: workgroups wait on their chunk's arrival flag (te_accumulate_lds_gated_fused_kernel; flag = hipStreamWriteValue32 behind the
// chunk's copy), hash two messages per lane, finish their digests themselves (one inversion per workgroup) and report completion in
// pinned host memory; this thread releases a chunk's copy-out (DMA, side stream) as soon as the chunk's workgroups are done.
// Measured and dropped on the way: separate per-chunk finalize passes behind an accumulate-only gated kernel (4.43 / 3.9 ms per 2^20
// Pedersen hashes when this form took 4.2 / 3.65; they fight the accumulate kernel for issue slots), the digests stored straight
// into the pinned buffer through an LDS staging area (4.6 / 3.9 ms), the flag writes on a stream of their own (same), four
// workgroups per CU instead of three (same): profiles/r05_s13, r05_s16.

// Chunks of one launch differ in size: a quarter chunk first (the workgroups start after 0.08 ms of copying instead of 0.31), small
// ones last (the copy-out behind the last workgroup is short), full ones between.  Sizes in granules of chunk / 4 messages; the kernel
// maps workgroup -> granule -> chunk through TeGate::chunk_of (64 granules at most: larger batches take larger granules).
// Returns the first message of every chunk, then n.
static std::vector<size_t> te_gate_schedule(size_t n, size_t chunk, bool ramp, size_t* granule_out) {
    size_t granule = chunk / 4;
    while ((n + granule - 1) / granule > 64) granule *= 2;
    const size_t n_granules = (n + granule - 1) / granule, full = std::max<size_t>(1, chunk / granule);
    std::vector<size_t> sizes;
    size_t left = n_granules;
    const bool ramped = ramp && full == 4 && n_granules >= 12;
    if (ramped) {
        sizes = {1, 1, 2};
        left -= 8;
    }
    while (left) {
        const size_t k = std::min(left, full);
        sizes.push_back(k);
        left -= k;
    }
    if (ramped) sizes.insert(sizes.end(), {2, 1, 1});
    std::vector<size_t> first;
    size_t at = 0;
    for (size_t k : sizes) {
        first.push_back(at * granule);
        at += k;
    }
    first.push_back(n);
    *granule_out = granule;
    return first;
}
// the context's gate resources: 64 arrival flags in fine-grained device memory, n_wg completion words (+ the error word) in pinned host
// memory, the copy-in and copy-out streams.  false (hipSuccess cleared): this stack cannot gate, the context stops trying
static bool te_gate_resources(akp_ctx* c, size_t n_wg) {
    auto ok = [](hipError_t e) {
        if (e != hipSuccess) (void)hipGetLastError();
        return e == hipSuccess;
    };
    if (!c->gate_flags) {
        if (!ok(hipExtMallocWithFlags((void**)&c->gate_flags, 64 * sizeof(u32), hipDeviceMallocFinegrained))) { c->gate_flags = nullptr; return false; }
        if (!ok(hipMemset(c->gate_flags, 0, 64 * sizeof(u32)))) return false;
    }
    if (c->gate_done_cap < n_wg) {
        if (c->gate_done) {
            if (!ok(hipDeviceSynchronize())) return false;
            (void)hipHostFree(c->gate_done);
            c->gate_done = c->gate_done_dev = nullptr;
            c->gate_done_cap = 0;
        }
        // word 0: set by a workgroup that gave up; the completion words start at word 16
        if (!ok(hipHostMalloc((void**)&c->gate_done, (n_wg + 16) * sizeof(u32), hipHostMallocMapped))) { c->gate_done = nullptr; return false; }
        if (!ok(hipHostGetDevicePointer((void**)&c->gate_done_dev, c->gate_done, 0))) {
            (void)hipHostFree(c->gate_done);
            c->gate_done = c->gate_done_dev = nullptr;
            return false;
        }
        memset(c->gate_done, 0, (n_wg + 16) * sizeof(u32));
        c->gate_done_cap = n_wg;
    }
    // The copy-in stream of the gated launch is a HIGH-priority stream of its own (pipe[5]), like the copy-out stream (pipe[4]).  Its flag
    // writes (and copies below the DMA threshold) are kernels of another hardware queue than the hash kernel's, and where the runtime
    // puts a stream is not ours to choose: at normal priority the two streams sometimes shared ONE queue (the kernel then started behind
    // the last copy: 6.5 instead of 3.9 ms per 2^20 pinned Pedersen hashes) and sometimes two queues on one PIPE of the command
    // processor -- a grid with workgroups still to be placed keeps its pipe, every flag write waited for the pipe's time slice (~0.43 ms)
    // and the copy behind it for the flag: 14 ms, in every call of the process (profiles/r06_s39 ... s41: bench.py's pinned Pedersen
    // leg).  High-priority queues come from a pool of their own and are served first: 4.06 / 3.41 ms in exactly those two placements
    // (profiles/r06_s45).  A grid the device holds at once (workgroups taking tile after tile) also frees the pipe, but costs 12 %
    // where nothing collided (s42 ... s45, patch in profiles/r06_s45).
    if (!ok(ctx_copy_streams(c))) return false;
    if (!c->chunk_event[7] && !ok(hipEventCreateWithFlags(&c->chunk_event[7], hipEventDisableTiming))) return false;
    return true;
}
// *used = false: preconditions not met or the gate failed (the caller runs the chunked launches, which write every digest again)
static int32_t te_crh_gated(akp_te_params* p, const uint8_t* h_msgs, size_t n, size_t msg_len, void* h_out, bool* used) {
    *used = false;
    akp_ctx* c = p->ctx;
    if (c->gate_unavailable) return AKP_OK;
    if (c->gate_skip) {  // backing off after a timeout
        --c->gate_skip;
        return AKP_OK;
    }
    // the test build's switches (tools/gpu_r5_gated.py, gpu_r5_gate_knobs.py, the fallback test); libakp.so reads none of them
    size_t chunk = (size_t)1 << 17, lds_floor = 36864;
    u32 spin_limit = 1u << 15, poll_sleep = 0;
    bool ramp = true, skip_in = false, skip_out = false;
    const char* stamps_path = nullptr;
    (void)stamps_path;
#if defined(AKP_TEST_HOOKS)
    if (env_u32("AKP_TE_GATED", 1, 0, 1) == 0) return AKP_OK;  // A/B against the chunked launches
    chunk = env_size("AKP_TE_PIPE_CHUNK", chunk);
    lds_floor = env_size("AKP_TE_GATE_LDS_FLOOR", lds_floor);
    ramp = env_u32("AKP_TE_GATE_RAMP", 1, 0, 1) != 0;
    spin_limit = env_u32("AKP_TE_GATE_SPIN_LIMIT", spin_limit, 1, 1u << 24);  // 1: every workgroup that has to wait gives up -- exercises the fallback
    poll_sleep = env_u32("AKP_TE_GATE_POLL_SLEEP", 0, 0, 64);
    if (poll_sleep) spin_limit = std::max(1u, spin_limit / (1u + 8u * poll_sleep));  // the same bound in time
    // timing probes: the gated kernel without the copies it normally runs beside (the digests are then those of whatever the device
    // buffer held: the tools read times only), and per-workgroup release / end times written to a file
    skip_in = env_u32("AKP_TE_GATE_SKIP_COPY_IN", 0, 0, 1) != 0;
    skip_out = env_u32("AKP_TE_GATE_SKIP_COPY_OUT", 0, 0, 1) != 0;
    stamps_path = getenv("AKP_TE_GATE_STAMPS");
    if (stamps_path && !*stamps_path) stamps_path = nullptr;
#endif
    if (te_lds_block(msg_len, msg_len) != 256 || (chunk & 2047) || n <= chunk) return AKP_OK;
    size_t granule = 0;
    const std::vector<size_t> chunk_first = te_gate_schedule(n, chunk, ramp, &granule);
    const size_t n_chunks = chunk_first.size() - 1;
    if (n_chunks < 2 || n_chunks > 64) return AKP_OK;
    // a waiting workgroup gives up after the time its chunk would take at 1/10 of the link's rate (two polls per microsecond, 5 GB/s;
    // at least 2^15 polls): batches above 2^23 messages come in chunks of up to 2^-6 of the batch
    if (spin_limit == 1u << 15) spin_limit = (u32)std::min<size_t>(1u << 24, std::max<size_t>(spin_limit, 4 * granule * msg_len * 2 / 5000));
    // ONE gated launch per device at a time: two of them (two host threads with a context each) would hold all eight wave slots of
    // every SIMD with waiting workgroups, and the flag writes they wait for could not run.  A second caller takes the chunked launches.
    static std::mutex gate_busy[64];
    std::unique_lock<std::mutex> gate_turn(gate_busy[c->device & 63], std::try_to_lock);
    if (!gate_turn.owns_lock()) return AKP_OK;
    constexpr size_t per_wg = 256 * TE_FUSED_ITEMS;  // messages per workgroup
    const size_t n_wg = (n + per_wg - 1) / per_wg;
    if (!te_gate_resources(c, n_wg)) {
        c->gate_unavailable = true;
        return AKP_OK;
    }
    if (++c->gate_epoch == 0) ++c->gate_epoch;  // 0 is what fresh memory holds
    const u32 epoch = c->gate_epoch;
    hipStream_t s = c->stream, cin = c->pipe[5], side = c->pipe[4];
    // (`side` has the highest stream priority: device-to-host copies are blit KERNELS on this stack whatever the stream, and at normal
    // priority they wait until the hash kernel thins out -- the copy-outs of a large batch would pile up behind it: profiles/r05_s16)
    const u32 fe = te_fe_per_digest(p);
    const size_t dig = fe * sizeof(Fr);
    void *dm = nullptr, *dout = nullptr, *d_stamps = nullptr;
    if (int32_t rc = ctx_scratch(c, SCR_A, n * msg_len, &dm, s)) return rc;
    if (int32_t rc = ctx_scratch(c, SCR_B, n * dig, &dout, s)) return rc;
#if defined(AKP_TEST_HOOKS)
    if (stamps_path) {
        if (int32_t rc = ctx_scratch(c, SCR_F, 2 * n_wg * sizeof(uint64_t), &d_stamps, s)) return rc;
        HIP_TRY(hipMemsetAsync(d_stamps, 0, 2 * n_wg * sizeof(uint64_t), s));
    }
#endif
    __atomic_store_n(c->gate_done, 0u, __ATOMIC_RELEASE);  // the error word; nothing of this context is running a gated kernel now
    HIP_TRY(hipEventRecord(c->chunk_event[7], s));  // the side streams start behind whatever used the scratch last
    HIP_TRY(hipStreamWaitEvent(cin, c->chunk_event[7], 0));
    HIP_TRY(hipStreamWaitEvent(side, c->chunk_event[7], 0));
    // an error from here on leaves copies of the caller's buffers (and possibly the kernel, which its spin limit ends) in flight:
    // the streams are drained before it is returned
    bool gave_up = false, refused = false;
    const int32_t rc = [&]() -> int32_t {
        {
            TeKickLater kick;
            std::unique_lock<std::mutex> table_lock;
            TeResolved rs;
            if (int32_t rc = te_pick(p, msg_len, msg_len, s, kick, table_lock, &rs)) return rc;
            TeTable* t = rs.t;
            // all copies first, each followed by its flag (a small kernel); nothing of this call has been launched yet if the write-value
            // is refused
            for (size_t k = 0; k < n_chunks; ++k) {
                const size_t first = chunk_first[k], cnt = chunk_first[k + 1] - first;
                if (!skip_in) HIP_TRY(hipMemcpyAsync((uint8_t*)dm + first * msg_len, h_msgs + first * msg_len, cnt * msg_len, hipMemcpyHostToDevice, cin));
                if (hipStreamWriteValue32(cin, c->gate_flags + k, epoch, 0) != hipSuccess) {
                    (void)hipGetLastError();
                    refused = true;
                    return AKP_OK;
                }
            }
            TeGate gate{c->gate_flags, c->gate_done_dev + 16, c->gate_done_dev, epoch, (u32)(granule / per_wg), spin_limit, {}, poll_sleep, (Fr*)dout, fe,
                        (unsigned long long*)d_stamps};
            for (size_t k = 0; k < n_chunks; ++k)
                for (size_t g = chunk_first[k] / granule; g * granule < chunk_first[k + 1]; ++g) gate.chunk_of[g] = (uint8_t)k;
            // LDS: one message image (the product tree takes its place afterwards), and more than 32 KB per workgroup = at most four
            // workgroups (four waves per SIMD) on a CU: the flag writes and, for chunks below the runtime's DMA threshold, the copies
            // themselves are small KERNELS -- with every wave slot held by a spinning workgroup they never run and nothing arrives
            // (measured: tools/persist_probe.hip; a 63x9 batch in 4 MB chunks timed out)
            const size_t image = (te_lds_image_bytes(256, msg_len, msg_len) + 15) & ~(size_t)15;
            const size_t shm = std::max(std::max<size_t>(image, 9 * 512 * sizeof(u32)), lds_floor);
            const dim3 grid((unsigned)n_wg);
            if (t->pedersen && t->signed_subset)
                hipLaunchKernelGGL(te_accumulate_lds_gated_fused_kernel<2>, grid, dim3(256), shm, s, rs.lut, rs.lut1, (const uint8_t*)dm, msg_len, msg_len, rs.shape,
                        rs.groups, rs.steps, rs.tail, n, gate);
            else if (t->pedersen)
                hipLaunchKernelGGL(te_accumulate_lds_gated_fused_kernel<0>, grid, dim3(256), shm, s, rs.lut, rs.lut1, (const uint8_t*)dm, msg_len, msg_len, rs.shape,
                        rs.groups, rs.steps, rs.tail, n, gate);
            else
                hipLaunchKernelGGL(te_accumulate_lds_gated_fused_kernel<1>, grid, dim3(256), shm, s, rs.lut, rs.lut1, (const uint8_t*)dm, msg_len, msg_len, rs.shape,
                        rs.groups, rs.steps, rs.tail, n, gate);
            HIP_TRY(hipGetLastError());
        }
        // release every chunk's copy-out when its workgroups have reported (they finish roughly in launch order)
        const volatile u32* done = c->gate_done + 16;
        auto t0 = std::chrono::steady_clock::now();  // of the last workgroup seen to report: the watchdog is about PROGRESS, batches of any size pass
        for (size_t k = 0; k < n_chunks && !gave_up; ++k) {
            const size_t first = chunk_first[k], cnt = chunk_first[k + 1] - first;
            const size_t wg1 = (first + cnt + per_wg - 1) / per_wg;
            for (size_t b = first / per_wg; b < wg1 && !gave_up; ++b) {
                unsigned spins = 0;
                while (done[b] != epoch)
                    if ((++spins & 0xfff) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 0.25) {
                        gave_up = true;  // the kernel's own spin limit has ended it long before: its error word says why
                        break;
                    }
                if (spins > 0xfff) t0 = std::chrono::steady_clock::now();
            }
            if (!gave_up && !skip_out)
                HIP_TRY(hipMemcpyAsync((char*)h_out + first * dig, (const char*)dout + first * dig, cnt * dig, hipMemcpyDeviceToHost, side));
        }
        return AKP_OK;
    }();
    if (rc || refused) {
        (void)hipStreamSynchronize(cin);
        (void)hipStreamSynchronize(s);
        (void)hipStreamSynchronize(side);
        if (refused) c->gate_unavailable = true;  // hipStreamWriteValue32 is not available on this stack
        return rc;
    }
    HIP_TRY(hipStreamSynchronize(cin));
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(hipStreamSynchronize(side));
    const u32 err = __atomic_load_n(c->gate_done, __ATOMIC_ACQUIRE);  // the kernel has ended: its writes to host memory are there
#if defined(AKP_TEST_HOOKS)
    if (d_stamps) {
        std::vector<uint64_t> st(2 * n_wg);
        HIP_TRY(hipMemcpy(st.data(), d_stamps, st.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
        if (FILE* f = fopen(stamps_path, "wb")) {
            fwrite(st.data(), sizeof(uint64_t), st.size(), f);
            fclose(f);
        }
    }
#endif
    if (gave_up || err) {
        // *used stays false: the caller repeats the batch with the chunked launches.  A timeout is a moment, not a property of the stack
        // (ADVICE r05): back off for 8, 16, ... 1024 pinned calls, then try again; akp_last_error() carries a note after the (successful) call
        ++c->gate_timeouts;
        c->gate_skip = 8u << std::min<u32>(c->gate_timeouts - 1, 7);
        (void)fail(AKP_OK, "note: the gated launch of a pinned curve-hash batch timed out (%u time(s) on this context): this call and the next %u pinned calls use the chunked launches",
                c->gate_timeouts, c->gate_skip);
        return AKP_OK;
    }
    c->gate_timeouts = 0;
    *used = true;
    return AKP_OK;
}

// Gated or chunked?  Measured per context, not assumed.  Where the runtime puts a stream (which hardware queue, which pipe of the command
// processor) decides how well the gated launch's copies, flag writes and copy-outs get past its kernel, and that is not the library's to
// choose: round 6 met placements where the gated call took 5 / 6.5 / 14 ms against 3.4 / 4.9 / 4.9 ms for the chunked launches, and others where it
// takes 3.4 / 3.9 against 3.8 / 4.9 (profiles/r06_s41 ... s47).  So: four calls of either form, in turns (the first of each pays for streams, flags,
// scratch and is not counted; the best of the other three counts), then the faster one by a running mean of its ns per message, with
// every 32nd call given to the other form so that the picture can change (a background table upgrade, another tenant on the device).  Figures
// are kept per shape (tables of the handle, message length), four shapes per context.  Wall time of the whole call: these entry points return when the digests are in the caller's buffer.
constexpr u32 TE_TUNE_FIRST = 4;  // calls of either form before the choice: the first is not counted, the best of the other three is its figure
static akp_ctx::GateTune* te_gate_shape(akp_ctx* c, uint64_t key) {  // key != 0
    akp_ctx::GateTune* lru = &c->gate_tune[0];
    ++c->gate_tune_clock;
    for (akp_ctx::GateTune& g : c->gate_tune) {
        if (g.key == key) {
            g.last_used = c->gate_tune_clock;
            return &g;
        }
        if (g.last_used < lru->last_used) lru = &g;
    }
    *lru = akp_ctx::GateTune{};
    lru->key = key;
    lru->last_used = c->gate_tune_clock;
    return lru;
}
static bool te_gate_choice(akp_ctx::GateTune* g) {
    // AKP_TE_PINNED_FORM=gated | chunked pins the form (include/akp.h; the tests of the gated launch and the tools' A/B arms use it)
    if (const char* form = getenv("AKP_TE_PINNED_FORM")) {
        if (!strcmp(form, "gated")) return true;
        if (!strcmp(form, "chunked")) return false;
    }
    if (g->obs[1] < TE_TUNE_FIRST && g->obs[1] <= g->obs[0]) return true;  // in turns: both forms see the same moments of the process
    if (g->obs[0] < TE_TUNE_FIRST) return false;
    if (g->obs[1] < TE_TUNE_FIRST) return true;
    const bool gated_faster = g->ema[1] <= g->ema[0];
    if ((int)gated_faster != g->form_noted) {  // the choice, and every change of it, is visible: a note behind the (successful) call
        g->form_noted = (int)gated_faster;
        (void)fail(AKP_OK, "note: the pinned curve-hash batches of this context take %s (measured: %.2f ns per message gated, %.2f chunked)",
                gated_faster ? "the gated launch" : "the chunked launches", g->ema[1], g->ema[0]);
    }
    return (++g->calls & 31u) == 0 ? !gated_faster : gated_faster;
}
static void te_gate_observe(akp_ctx::GateTune* g, bool gated, double ns_per_msg) {
    const int i = gated ? 1 : 0;
    if (g->obs[i] == 1) g->ema[i] = ns_per_msg;
    else if (g->obs[i] > 1 && g->obs[i] < TE_TUNE_FIRST) g->ema[i] = std::min(g->ema[i], ns_per_msg);  // what disturbs a call only ever adds
    else if (g->obs[i] >= TE_TUNE_FIRST)  // later: down at once (one probe after a disturbed start is enough to change over), up slowly
        g->ema[i] = ns_per_msg < g->ema[i] ? ns_per_msg : 0.75 * g->ema[i] + 0.25 * ns_per_msg;
    if (g->obs[i] < (1u << 30)) ++g->obs[i];
}

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
    // Large batches: chunks of 2^17 messages, double-buffered -- the copy-in of chunk
    // i + 1 and the copy-out of chunk i - 1 run on their own streams under the kernels of chunk i (a 4x256 Pedersen hash
    // moves 128 B in and 64 B out for 3 us of kernel time per 1000 hashes: serial copies would double the call).
    constexpr size_t chunk = (size_t)1 << 17;
    const size_t dig = fe * sizeof(Fr);
    const bool out_pinned = device_alias(out, n * dig) != nullptr;
    // Pinned / registered buffers on both sides (round 4): everything is asynchronous -- the messages come in by DMA chunk after
    // chunk (all copies issued up front: the device buffers hold the whole batch), the kernels of the chunks run back to back
    // on the context stream, and the digests of chunk i leave by DMA on a side stream under the kernels of chunk i + 1.
    // Measured and dropped on the way (profiles/r04_s2, r04_s3): (1) round 3's finalize pass storing straight into the pinned
    // buffer -- 16-byte stores at a 64-byte pitch cross PCIe at 17 GB/s, 0.47 ms per 2^17 digests against 0.05 ms into HBM
    // + 0.15 ms of DMA; (2) the accumulate kernel reading pinned messages in place (it reads every byte once, through LDS):
    // every resident workgroup waits for PCIe at the same moments, 5.1 ms per 2^20 hashes in one launch and 6.0 ms
    // chunked, against 0.30 ms of DMA per chunk that hides completely; (3) the finalize pass on the side stream.
    // A pageable buffer on either side keeps the double-buffered loop below (its staged copies block this thread, which
    // that loop's issue order is built around).
    if (n > te_split_max && msg_len >= 4 && out_pinned && device_alias(msgs, n * msg_len)) {
        bool gated = false;
        // (the handle's tables and the message length: a handle with an HBM-sized table has figures of its own)
        const uint64_t tune_key = ((uint64_t)(uintptr_t)p->t << 16) ^ ((uint64_t)(uintptr_t)p->wide << 20) ^ (uint64_t)msg_len;
        const auto call_t0 = std::chrono::steady_clock::now();
        akp_ctx::GateTune* const tune = te_gate_shape(c, tune_key | 1u);
        const auto observe = [&](bool was_gated) {
            te_gate_observe(tune, was_gated, std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - call_t0).count() / (double)n);
        };
        const bool try_gate = te_gate_choice(tune);
        if (try_gate)
            if (int32_t rc = te_crh_gated(p, msgs, n, msg_len, out, &gated)) return rc;
#if defined(AKP_TEST_HOOKS)
        if (const char* report = getenv("AKP_TE_GATE_REPORT"))  // one line per pinned call: which form ran it (tools/gpu_r5_gated_stress.py counts them)
            if (FILE* f = fopen(report, "a")) {
                fprintf(f, "%zu %zu %d\n", n, msg_len, gated ? 1 : 0);
                fclose(f);
            }
#endif
        if (gated) {
            observe(true);
            return AKP_OK;
        }
        // the copy streams of the chunked launches are high-priority streams: [4] the copy-out stream -- its copy kernels should not queue
        // behind the hash kernels' workgroups (median 4.41 -> 4.24 ms per 2^20 hashes, profiles/r04_s3) -- and, since round 6, [5] the copy-in
        // stream, shared with the gated launch (te_gate_resources): in the placement where the normal-priority copy-in stream met the hash
        // kernels' queue on one pipe 3.9 -> 3.3 ms (Pedersen, HBM-sized table) and 2.4 -> 1.9 ms (Bowe-Hopwood), no difference elsewhere
        // (profiles/r06_s50, r06_s51)
        HIP_TRY(ctx_copy_streams(c));
        for (int i = 0; i < 8; ++i)
            if (!c->chunk_event[i]) HIP_TRY(hipEventCreateWithFlags(&c->chunk_event[i], hipEventDisableTiming));
        size_t pchunk = chunk;
#if defined(AKP_TEST_HOOKS)
        pchunk = env_size("AKP_TE_PIPE_CHUNK", chunk);
#endif
        const TePipe pipe{pchunk /* 2^16 / 2^18 / 2^19 measured slower, profiles/r04_s3 */, msgs, out, c->pipe[5], c->pipe[4], c->chunk_event[0], c->chunk_event[1]};
        if (int32_t rc = ctx_scratch(c, SCR_A, n * msg_len, &dm, s)) return rc;
        if (int32_t rc = ctx_scratch(c, SCR_B, n * dig, &dout, s)) return rc;
        HIP_TRY(hipEventRecord(c->chunk_event[7], s));  // the side streams start behind whatever used the scratch last
        HIP_TRY(hipStreamWaitEvent(pipe.cin, c->chunk_event[7], 0));
        HIP_TRY(hipStreamWaitEvent(pipe.side, c->chunk_event[7], 0));
        const int32_t rc = te_crh_run(p, (const uint8_t*)dm, n, msg_len, (Fr*)dout, s, msg_len, &pipe);
        HIP_TRY(hipStreamSynchronize(pipe.cin));
        HIP_TRY(hipStreamSynchronize(s));
        HIP_TRY(hipStreamSynchronize(pipe.side));
        if (rc == AKP_OK && !try_gate) observe(false);  // (not the calls that asked for the gate and did not get it: one of them may have waited for a time-out)
        return rc;
    }
    if (n <= chunk || msg_len == 0) {
        if (int32_t rc = ctx_scratch(c, SCR_A, n * msg_len, &dm, s)) return rc;
        if (int32_t rc = ctx_scratch(c, SCR_B, n * dig, &dout, s)) return rc;
        if (msg_len) HIP_TRY(hipMemcpyAsync(dm, msgs, n * msg_len, hipMemcpyHostToDevice, s));
        if (int32_t rc = te_crh_dev(p, (const uint8_t*)dm, n, msg_len, (Fr*)dout, s)) return rc;
        HIP_TRY(hipMemcpyAsync(out, dout, n * dig, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        return AKP_OK;
    }
    if (int32_t rc = ctx_scratch(c, SCR_A, 2 * chunk * msg_len, &dm, s)) return rc;
    if (int32_t rc = ctx_scratch(c, SCR_B, 2 * chunk * dig, &dout, s)) return rc;
    // copy-in and copy-out on the context's two high-priority streams (round 6; the staged copies are kernels on queues of their own: served
    // first, they do not wait for the hash kernels' grids to be placed: 4.57 -> 4.33 ms per 2^20 Pedersen hashes, profiles/r06_s52)
    HIP_TRY(ctx_copy_streams(c));
    for (int i = 0; i < 8; ++i)
        if (!c->chunk_event[i]) HIP_TRY(hipEventCreateWithFlags(&c->chunk_event[i], hipEventDisableTiming));
    hipStream_t cin = c->pipe[5], cout = c->pipe[4];
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
// ---- ragged batches: every message has its own length (round 5; crh/pedersen/mod.rs:82-99, crh/bowe_hopwood/mod.rs:121-138,
// merkle_tree/mod.rs:411-422) ---------------------------------------------------------------------------------------------------
// item i = bytes [offsets[i], offsets[i+1]) of d_msgs; max_len = a bound on the longest item (the wide table is built for it).
// Items launch sorted by their number of table steps (ragged_sort.hpp), longest first; scratch: SCR_E / SCR_F as te_crh_run, SCR_H
// for the sort.  Enqueue only.
static inline bool te_ragged_sort_on(size_t n) {
#if defined(AKP_TEST_HOOKS)
    if (const char* e = getenv("AKP_RAGGED_SORT")) return atoi(e) != 0;  // A/B of the launch order (test build only)
#endif
    return n >= 4096;  // below that the three sort launches cost more than the idle lanes
}
int32_t te_crh_ragged_dev(akp_te_params* p, const uint8_t* d_msgs, const uint64_t* d_offsets, size_t n, size_t max_len, Fr* d_out, hipStream_t s) {
    if (max_len * 8 > te_input_bits(p))
        return fail(AKP_ERR_BAD_LENGTH, "incorrect input length %zu for window params %ux%u (the reference panics)", max_len, p->W, p->N);
    if (n == 0) return AKP_OK;
    if (n >= ((size_t)1 << 32)) return fail(AKP_ERR_BAD_PARAMS, "batch of %zu messages exceeds the supported 2^32 - 1", n);
    akp_ctx* c = p->ctx;
    TeKickLater kick;
    std::unique_lock<std::mutex> table_lock;
    TeResolved rs;
    if (int32_t rc = te_pick(p, max_len, max_len, s, kick, table_lock, &rs, false)) return rc;
    TeTable* t = rs.t;
    const u32 D = t->pedersen ? t->digit_bits : t->group;  // Bowe-Hopwood: no remainder table (R = 0): left-over chunks are single steps
    void *xyz = nullptr, *prefix = nullptr, *work = nullptr;
    if (int32_t rc = ctx_scratch(c, SCR_E, n * 3 * sizeof(F29Pad), &xyz, s)) return rc;
    if (int32_t rc = ctx_scratch(c, SCR_F, n * sizeof(F29Pad), &prefix, s)) return rc;
    const u32* order = nullptr;
    if (te_ragged_sort_on(n)) {
        if (int32_t rc = ctx_scratch(c, SCR_H, (2 * n + 2 * RAGGED_MAX_KEYS) * sizeof(u32), &work, s)) return rc;
        u32* d_order = (u32*)work + n + 2 * RAGGED_MAX_KEYS;
        const RaggedKey key{t->pedersen ? 0u : 1u, D, t->n_gen};
        HIP_TRY(ragged_order(d_offsets, n, key, (u32*)work, d_order, s));
        order = d_order;
    }
    const unsigned grid = (unsigned)((n + 255) / 256);
    const u32 built = t->pedersen && !t->signed_subset ? 0xffffffffu : (t->pedersen || t->group > 1 ? t->units_built : 0u);
    if (t->pedersen && t->signed_subset)
        hipLaunchKernelGGL(te_accumulate_ragged_kernel<2>, dim3(grid), dim3(256), 0, s, t->d_lut, t->d_lut1, d_msgs, d_offsets, order, D, t->n_gen, built,
                (u32)max_len, (F29Pad*)xyz, n);
    else if (t->pedersen)
        hipLaunchKernelGGL(te_accumulate_ragged_kernel<0>, dim3(grid), dim3(256), 0, s, t->d_lut, t->d_lut1, d_msgs, d_offsets, order, D, t->n_gen, built,
                (u32)max_len, (F29Pad*)xyz, n);
    else
        hipLaunchKernelGGL(te_accumulate_ragged_kernel<1>, dim3(grid), dim3(256), 0, s, t->d_lut, t->d_lut1, d_msgs, d_offsets, order, D, t->n_gen, built,
                (u32)max_len, (F29Pad*)xyz, n);
    HIP_TRY(hipGetLastError());
    return te_launch_finalize(p, (const F29Pad*)xyz, (F29Pad*)prefix, d_out, n, s);
}
// offsets[0 .. n] on the host: non-decreasing, every item within the window; *max_len = the longest item
int32_t te_ragged_check_offsets(const akp_te_params* p, const uint64_t* offsets, size_t n, size_t* max_len) {
    size_t mx = 0;
    for (size_t i = 0; i < n; ++i) {
        if (offsets[i + 1] < offsets[i]) return fail(AKP_ERR_BAD_PARAMS, "offsets[%zu] > offsets[%zu]: offsets must not decrease", i, i + 1);
        mx = std::max<size_t>(mx, (size_t)(offsets[i + 1] - offsets[i]));
    }
    if (mx * 8 > te_input_bits(p))
        return fail(AKP_ERR_BAD_LENGTH, "incorrect input length %zu for window params %ux%u (the reference panics)", mx, p->W, p->N);
    *max_len = mx;
    return AKP_OK;
}
extern "C" int32_t akp_te_crh_batch_ragged_dev(akp_te_params* p, const uint8_t* d_msgs, const uint64_t* d_offsets, size_t n, size_t max_len,
        uint64_t* d_out, void* stream) {
    NEED_TE(p, "akp_te_crh_batch_ragged_dev");
    if (n && (!d_offsets || !d_out)) return fail(AKP_ERR_BAD_PARAMS, "NULL buffer");
    return te_crh_ragged_dev(p, d_msgs, d_offsets, n, max_len, (Fr*)d_out, pick_stream(p->ctx, stream));
}
extern "C" int32_t akp_te_crh_batch_ragged(akp_te_params* p, const uint8_t* msgs, const uint64_t* offsets, size_t n, uint64_t* out) {
    NEED_TE(p, "akp_te_crh_batch_ragged");
    if (n == 0) return AKP_OK;
    if (!offsets || !out) return fail(AKP_ERR_BAD_PARAMS, "NULL buffer");
    size_t max_len = 0;
    if (int32_t rc = te_ragged_check_offsets(p, offsets, n, &max_len)) return rc;
    const size_t total = (size_t)(offsets[n] - offsets[0]);
    if (total && !msgs) return fail(AKP_ERR_BAD_PARAMS, "msgs is NULL");
    akp_ctx* c = p->ctx;
    hipStream_t s = c->stream;
    const size_t dig = te_fe_per_digest(p) * sizeof(Fr);
    void *dm = nullptr, *doff = nullptr, *dout = nullptr;
    if (int32_t rc = ctx_scratch(c, SCR_A, total + 4, &dm, s)) return rc;
    if (int32_t rc = ctx_scratch(c, SCR_G, (n + 1) * sizeof(uint64_t), &doff, s)) return rc;
    if (int32_t rc = ctx_scratch(c, SCR_B, n * dig, &dout, s)) return rc;
    if (total) HIP_TRY(hipMemcpyAsync(dm, msgs + offsets[0], total, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(doff, offsets, (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, s));
    // the offsets go up as they are: the device base is moved back by offsets[0] instead (never dereferenced below offsets[0])
    if (int32_t rc = te_crh_ragged_dev(p, (const uint8_t*)dm - offsets[0], (const uint64_t*)doff, n, max_len, (Fr*)dout, s)) return rc;
    HIP_TRY(hipMemcpyAsync(out, dout, n * dig, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return AKP_OK;
}
extern "C" int32_t akp_te_two_to_one_batch(akp_te_params* p, const uint8_t* left, const uint8_t* right, size_t n, size_t half_len,
        uint64_t* out) {
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
        hipLaunchKernelGGL(te_concat_bytes_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, s, (const uint8_t*)dl,
                (const uint8_t*)dr, half_len, buflen, (uint8_t*)dbuf, n);
        HIP_TRY(hipGetLastError());
    }
    // the buffer past left || right is zero padding: te_crh_dev skips it (Pedersen) or adds its constant (Bowe-Hopwood)
    if (int32_t rc = te_crh_dev(p, (const uint8_t*)dbuf, n, buflen, (Fr*)dout, s, std::min(buflen, 2 * half_len))) return rc;
    HIP_TRY(hipMemcpyAsync(out, dout, n * fe * sizeof(Fr), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return AKP_OK;
}
// one level of compress(): serialise digest pairs into per-node buffers (SCR_D), hash into d_out
int32_t te_compress_dev(akp_te_params* p, const Fr* d_left, const Fr* d_right, size_t n, Fr* d_out, hipStream_t s) {
    if (n == 0) return AKP_OK;
    const size_t buflen = ((size_t)p->W * p->N) / 8;
    const u32 fe = te_fe_per_digest(p);
    const size_t used = std::min<size_t>(buflen, (size_t)2 * fe * 32);
    void* dbuf = nullptr;
    if (int32_t rc = ctx_scratch(p->ctx, SCR_D, n * buflen, &dbuf, s)) return rc;
    const size_t work = n * 2 * fe;
    const bool tail_on = te_zero_tail_on();
    if (used == (size_t)2 * fe * 32 && tail_on) {  // both digests fit: packed at a pitch of `used` bytes, the padding is never materialised
        hipLaunchKernelGGL(te_serialize_pairs_vec_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, s, d_left, d_right, fe, (uint8_t*)dbuf, n);
        HIP_TRY(hipGetLastError());
        return te_crh_run(p, (const uint8_t*)dbuf, n, buflen, d_out, s, used, nullptr, used);
    }
    hipLaunchKernelGGL(te_serialize_pairs_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, s, d_left, d_right, fe, buflen,
            (uint8_t*)dbuf, n);
    HIP_TRY(hipGetLastError());
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

