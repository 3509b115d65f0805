// capi_internal.hpp -- what the translation units of libakp.so share: the context, the parameter-handle structs and
// the handful of internal entry points that cross unit boundaries.  Product code (no oracle, no CPU fallback).
//   capi_ctx.hip       errors, contexts and their scratch, pinned host memory, host field helpers
//   capi_poseidon.hip  Poseidon parameters (incl. the Grain-LFSR defaults), kernel routing, batch entry points, sponge
//   capi_te.hip        Pedersen / Bowe-Hopwood tables and batch entry points
//   capi_merkle.hip    tree builds, proofs, verification (+ capi_tree.inc: HBM-resident trees; capi_multi.inc: several GPUs)
// Only the unit that launches a kernel family includes that family's header (poseidon_kernels.hpp / te_kernels.hpp).
// Everything not declared in include/akp.h has hidden visibility (-fvisibility=hidden; akp.h is included under `default`).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#pragma GCC visibility push(default)
#include "../../include/akp.h"
#pragma GCC visibility pop
#include "fr.hpp"
#include "f29.hpp"
#include "akp_types.hpp"

using namespace akp;

// ---- errors, environment knobs (capi_ctx.hip) ---------------------------------------------------------------------------
int32_t fail(int32_t code, const char* fmt, ...);
size_t env_size(const char* name, size_t dflt);
u32 env_u32(const char* name, u32 dflt, u32 lo, u32 hi);
#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return fail(AKP_ERR_HIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// ---- context: device, stream, grow-only scratch slots ---------------------------------------------------------------------
enum { SCR_A = 0, SCR_B, SCR_C, SCR_D, SCR_E, SCR_F, SCR_G, SCR_H, SCR_I, SCR_J, SCR_K, SCR_L, SCR_COUNT };
struct akp_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    void* scratch[SCR_COUNT] = {};
    size_t scratch_bytes[SCR_COUNT] = {};
    std::vector<void*> scratch_retired;  // blocks a slot outgrew (ctx_scratch): freed with the context, never under running work
    // stream that last used each slot + an event to order the next use on ANOTHER stream behind it: `_dev` entry points
    // run on the caller's stream while the host-pointer entry points run on `stream` (non-blocking, so no implicit order
    // with the legacy default stream); without this two calls on different streams would race on the shared scratch
    hipStream_t slot_stream[SCR_COUNT] = {};
    bool slot_used[SCR_COUNT] = {};
    hipEvent_t slot_event[SCR_COUNT] = {};
    // pinned staging for small host<->device transfers of the tree / proof entry points
    void* pinned = nullptr;
    size_t pinned_bytes = 0;
    // more streams for the chunked host-pointer batches (copy-in / kernel / copy-out of consecutive chunks overlap); [4] (copy-out of the
    // pinned curve-hash path) and [5] (copy-in of its gated launch) are high-priority streams
    hipStream_t pipe[7] = {};
    hipEvent_t chunk_event[8] = {};  // copy-stream -> compute-stream hand-over of leaf chunks (host_tree_build)
    // where the last host-pointer tree build left its inner nodes (heap order, `last_tree_nodes` digests): what the
    // multi-device build reads for the all-gather of the sub-roots and the per-device copy-outs -- an explicit hand-over
    // instead of a convention about scratch slots
    const void* last_tree_non_leaf = nullptr;
    size_t last_tree_nodes = 0;
    // handles created on this context and still alive: parameter sets, trees, sponges.  akp_ctx_destroy with live handles
    // releases the device resources and marks the context dead; the struct itself goes with the last handle, whose compute
    // calls fail cleanly until then (handles created on akp_multi_ctx may outlive akp_multi_destroy without touching freed memory)
    int live_handles = 0;
    bool dead = false;
    // gated (persistent) curve-hash launch of the pinned host path (capi_te.hip te_crh_gated): per-chunk arrival flags in FINE-GRAINED
    // device memory (+ one error word), per-workgroup completion words in pinned host memory, the epoch that tags one call's values
    u32* gate_flags = nullptr;   // [64] arrival flags (fine-grained device memory)
    // which form serves this context's pinned curve-hash batches -- the gated launch or the chunked launches -- is MEASURED (capi_te.hip
    // te_gate_choice), per shape (the handle's tables, the message length): ns per message of either form ([0] chunked, [1] gated), how
    // often each was seen, the form the last note in akp_last_error() named (1 gated, 0 chunked, -1 none yet).  Four shapes are
    // remembered (a host that alternates between a few shapes keeps their figures), the least recently used one makes room.
    struct GateTune {
        uint64_t key = 0;
        double ema[2] = {0.0, 0.0};
        u32 obs[2] = {0, 0};
        u32 calls = 0, last_used = 0;
        int form_noted = -1;
    };
    GateTune gate_tune[4];
    u32 gate_tune_clock = 0;
    u32* gate_done = nullptr;    // host pointer: word 0 = a workgroup gave up, completion words from word 16
    u32* gate_done_dev = nullptr;  // its device alias
    size_t gate_done_cap = 0;
    u32 gate_epoch = 0;
    bool gate_unavailable = false;  // STRUCTURAL: an allocation / hipStreamWriteValue32 failed -- this stack cannot gate: the chunked launches from then on
    // a gated launch that TIMED OUT (another context's long kernel held the device, say) is not structural: the batch is repeated through
    // the chunked launches, the next `gate_skip` pinned calls take them too, then the gate is tried again (8, 16, ... 1024 calls)
    u32 gate_skip = 0, gate_timeouts = 0;
    // HBM one precomputed curve table may take (akp_ctx_set_table_budget); 0: 320 MiB (cache-sized tables); AKP_TABLE_BUDGET_DEVICE: a
    // quarter of the device's memory, at most half of what is free
    size_t table_budget = 0;
};
void ctx_handle_released(akp_ctx* c);
// scratch slot `slot` with at least `bytes`, to be used on stream `s` (ordered behind the slot's last use on another stream)
int32_t ctx_scratch(akp_ctx* c, int slot, size_t bytes, void** out, hipStream_t s);
// the context's two HIGH-priority copy streams, created on first use: pipe[5] copy-in, pipe[4] copy-out (capi_ctx.hip)
hipError_t ctx_copy_streams(akp_ctx* c);
// `_dev` entry points use the caller's stream verbatim (NULL = HIP's legacy default stream, which is what
// torch's default stream is), so event timing and ordering follow the caller's stream semantics.
static inline hipStream_t pick_stream(akp_ctx*, void* s) { return (hipStream_t)s; }

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
void* device_alias(const void* host, size_t bytes);
template <class Launch>
static int32_t pipelined_batch(akp_ctx* c, size_t n, const HostIn* ins, int n_in, void* host_out, size_t out_bytes_per_item, int out_slot,
                               Launch launch /* (void* const* d_in, void* d_out, size_t count, hipStream_t) */) {
    // Zero copy: when every buffer is pinned / registered host memory the kernels read and write it in place over PCIe.
    // Each item is read once and written once, the kernels are compute-bound, and loads and stores of different waves use
    // both directions of the link at the same time -- which the copy engines of this platform do not (opposite copies
    // mostly serialise, profiles/r02_s3): 2^20 permutations 3.79 -> 3.00 ms, 2^22 11.2 -> 10.2 ms (profiles/r02_s33).
    if (n) {
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
    constexpr int max_lanes = 3;  // >= depth + 1 buffers in flight (3 / 4 lanes measured alike, profiles/r02_s4)
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
                    HIP_TRY(hipMemcpyAsync(di[k], (const char*)ins[k].host + done * ins[k].bytes_per_item, cnt * ins[k].bytes_per_item,
                            hipMemcpyHostToDevice, s));
            }
            void* dout = in_place ? di[0] : (char*)d_out + (size_t)lane * chunk * out_bytes_per_item;
            if (int32_t rc = launch(di, dout, cnt, s)) return rc;
        }
        if (ci >= depth) {
            const size_t co = ci - depth, done = co * chunk, cnt = std::min(chunk, n - done);
            const int lane = (int)(co % lanes);
            const void* dout = in_place ? (char*)d_in[0] + (size_t)lane * chunk * ins[0].bytes_per_item :
                                        (char*)d_out + (size_t)lane * chunk * out_bytes_per_item;
            HIP_TRY(hipMemcpyAsync((char*)host_out + done * out_bytes_per_item, dout, cnt * out_bytes_per_item, hipMemcpyDeviceToHost,
                    st[lane]));
        }
    }
    for (int i = 0; i < lanes; ++i) HIP_TRY(hipStreamSynchronize(st[i]));
    return AKP_OK;
}

// ---- host field helpers -------------------------------------------------------------------------------------------------
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
static inline bool fr_words_reduced(const uint64_t* w) {
    const uint64_t P[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL};
    for (int i = 3; i >= 0; --i) {
        if (w[i] < P[i]) return true;
        if (w[i] > P[i]) return false;
    }
    return false;
}

// ---- Poseidon parameter handle (capi_poseidon.hip) ----------------------------------------------------------------------
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
    // trees and sponges built on this handle keep using it (akp_merkle_tree::pleaf / ptwo, akp_sponge::p): they pin it, and
    // akp_poseidon_params_destroy on a pinned handle only marks it -- the memory goes with the last pin (a host that drops its
    // parameter objects before its trees, e.g. an LRU handle cache, cannot leave them with a dangling pointer)
    int pins = 0;
    bool destroy_pending = false;
};
static inline void poseidon_pin(akp_poseidon* p) { if (p) ++p->pins; }
void poseidon_unpin(akp_poseidon* p);
#define NEED_DEV(p, what)                                                                                  \
    do {                                                                                                   \
        if (!(p)) return fail(AKP_ERR_BAD_PARAMS, what ": params is NULL");                                \
        if (!(p)->ctx) return fail(AKP_ERR_HIP, what ": parameter handle has no device context (no CPU fallback)"); \
        if ((p)->ctx->dead) return fail(AKP_ERR_BAD_PARAMS, what ": the context of this handle was destroyed");  \
        HIP_TRY(hipSetDevice((p)->ctx->device));                                                           \
    } while (0)
// the batch launchers the tree code shares with the batch entry points: route n items to the register / latency / LDS-file kernels
int32_t launch_permute(akp_poseidon* p, Fr* d_states, size_t n, hipStream_t s, bool host_memory = false);
int32_t launch_crh(akp_poseidon* p, const Fr* in0, const Fr* in1, size_t k, Fr* d_out, size_t n, hipStream_t s);
// ragged CRH batch on device buffers (t = 3): item i = elements [d_offsets[i], d_offsets[i+1]) of d_inputs
int32_t poseidon_crh_ragged_dev(akp_poseidon* p, const Fr* d_inputs, const uint64_t* d_offsets, size_t n, Fr* d_out, hipStream_t s);
int32_t launch_verify_paths_t3(akp_poseidon* leafp, akp_poseidon* two, const Fr* d_leaves, size_t leaf_len, const uint64_t* d_idx, const Fr* d_sibs,
        const Fr* d_auth, size_t depth, const Fr* d_root, uint8_t* d_ok, size_t m, hipStream_t s, bool* done);

// ---- Pedersen / Bowe-Hopwood parameter handle (capi_te.hip) -------------------------------------------------------------
// The precomputed tables of one parameter set on one DEVICE (round 5).  In the reference a `Parameters` value is `Sync` and every
// rayon worker borrows the same one (crh/mod.rs:22, merkle_tree/mod.rs:417,458,494); here every worker thread has a context of its
// own, so the tables -- up to 46 / 75 GB -- cannot belong to a handle: handles created with the same generators, window and table
// shape on contexts of the same device ATTACH to one TeTable (process-wide store in capi_te.hip, reference-counted; the last handle
// frees it).  `mu` serialises everything that reads or changes the table's pointers with the kernel launches that use them.
// Round 6: NOTHING a launch may have been given is freed while a handle is attached -- an extended table and superseded constants
// are retired, not freed -- so a graph captured after akp_te_params_prepare stays valid whatever other handles of the same table hash
// later, and no extension drains the device.
struct TeTable {
    std::mutex mu;
    int device = 0;
    bool pedersen = false;     // Pedersen arithmetic (AKP_TE_PEDERSEN and AKP_TE_PEDERSEN_X share a table), else Bowe-Hopwood
    u32 W = 0, N = 0;
    u32 n_gen = 0;             // W * N flat generators
    u32 digit_bits = 0;        // Pedersen: table digit width D (2..24; plain table 1..14)
    u32 group = 1;             // Bowe-Hopwood: chunks per table step (1..8)
    TeEntry* d_lut = nullptr;    // Pedersen: [ceil(n_gen/D)][2^D] (signed-subset table: [ceil(n_gen/D)][2^(D-1)]); BH: group table
    TeEntry* d_lut1 = nullptr;  // BH: single-chunk table [n_gen][4]; Pedersen signed-subset: cprefix [n_digits + 1]
    // BH: sum of G[c] over the zero-padded tail chunks [from, to) of a two-to-one shape that has no remainder table: one entry per
    // shape, computed once (a table is shared by streams of several contexts: nothing is ever recomputed in place)
    struct Tail {
        u32 from = 0, to = 0;
        TeEntry* d = nullptr;
    };
    static constexpr int MAX_TAILS = 8;
    Tail tails[MAX_TAILS];
    int n_tails = 0;
    // BH: the < G chunks a message shape leaves after its last full group are ONE more table step: a table of 2^(3r) entries for
    // the r chunks starting at chunk `first`, with the constant of the zero-padded tail chunks [tail_from, tail_to) folded in;
    // built on the first use of that shape (a parameter set sees a handful of shapes)
    struct Remainder {
        u32 first = 0, r = 0, tail_from = 0, tail_to = 0;
        TeEntry* d = nullptr;
    };
    static constexpr int MAX_REMAINDERS = 8;
    Remainder rem[MAX_REMAINDERS];
    int n_rem = 0;
    Fr* d_gens = nullptr;  // BH: the generators (affine, wire form), kept for the group / remainder tables built later
    // The wide table is built FOR THE MESSAGE LENGTHS THAT ARRIVE (or that akp_te_params_prepare names): d_lut covers the first
    // `units_built` of `units_total` digits (Pedersen signed-subset table) / chunk groups (Bowe-Hopwood); a longer message extends
    // it (te_ensure_table).  A table that only ever hashes 32- and 64-byte tree nodes holds a third of the 63x9 table.
    u32 units_total = 0, units_built = 0;
    bool shape_auto = true;  // the shape came from the table budget (not from akp_te_params_create_shaped): it may narrow when memory is short
    NielsPad* d_half = nullptr;  // Pedersen signed-subset table: the halved generators it is built from
    bool signed_subset = false;  // Pedersen: d_lut holds the signed-subset table (te_kernels.hpp), d_lut1 its constants
    // store bookkeeping (under the store's mutex): handles attached, the key the table is filed under
    int refs = 0;
    bool in_store = false;
    u32 key_shape = 0;               // the shape it was created with (a narrowed table leaves the store: its shape no longer says what was asked)
    std::vector<uint64_t> gens;      // host copy of the generators (key comparison; 64 KB for a 4x256 window)
    uint64_t builds = 0;             // wide-table builds so far (akp_te_params_table_info: a test can see that eight handles built once)
    // creation (te_store_attach): the table enters the store as a placeholder and is initialised under ITS lock, not the store's
    bool initialised = false;
    int32_t init_rc = 0;
    std::string init_err;
    std::vector<void*> retired;                // device blocks superseded while handles were attached (an extended table, constants of a narrowed
                                               // shape): a launch -- or a captured graph -- may still read them; freed with the table
    // builds run on the DEVICE's two build streams (capi_te.hip te_dev_streams: never behind a caller's work, never destroyed): `build_stream`
    // for a build some caller waits for, `bg_stream` (lowest priority) for the background upgrade; `active_stream`: the one the current
    // holder of `mu` builds on
    hipStream_t build_stream = nullptr, bg_stream = nullptr, active_stream = nullptr;
    // background build of an HBM-sized table (te_upgrade_kick): the thread that is building or built last, whether one is running, what went wrong
    std::thread builder;
    std::atomic<bool> building{false};
    std::atomic<bool> upgrade_failed{false};
    // a tree's inner-node shape, announced before its leaf level asks for the table (te_tree_prepare): the builder takes it first
    std::atomic<bool> hint_set{false};
    size_t hint_msg_len = 0, hint_data_len = 0;
    std::string upgrade_error;
    akp_te_build_report last_build{};          // phases of the last build / extension of the wide table (akp_te_params_table_info)
};
struct akp_te_params {
    akp_ctx* ctx = nullptr;
    int kind = 0;
    u32 W = 0, N = 0;
    u32 n_gen = 0;             // W * N flat generators
    TeTable* t = nullptr;      // shared with every handle of the same parameters on this device: the table this handle was created with
                               // -- or, when the context's table budget admits a wider one, the CACHE-SIZED table it starts hashing on
    TeTable* wide = nullptr;   // ... and the HBM-sized table that takes over once it is built (in the background, or by akp_te_params_prepare)
    int pins = 0;              // as akp_poseidon::pins
    bool destroy_pending = false;
};
static inline void te_pin(akp_te_params* p) { if (p) ++p->pins; }
void te_unpin(akp_te_params* p);
// Pedersen arithmetic (subset-sum tables over W * N generators): the plain hash and the one composed with TECompressor
static inline bool te_is_pedersen(const akp_te_params* p) { return p->kind == AKP_TE_PEDERSEN || p->kind == AKP_TE_PEDERSEN_X; }
static inline u32 te_fe_per_digest(const akp_te_params* p) { return p->kind == AKP_TE_PEDERSEN ? 2u : 1u; }
static inline size_t te_input_bits(const akp_te_params* p) {  // max message bits before the reference panics
    return te_is_pedersen(p) ? (size_t)p->W * p->N : (size_t)p->W * p->N * 3;
}

#define NEED_TE(p, what)                                                    \
    do {                                                                    \
        if (!(p)) return fail(AKP_ERR_BAD_PARAMS, what ": params is NULL"); \
        if ((p)->ctx->dead) return fail(AKP_ERR_BAD_PARAMS, what ": the context of this handle was destroyed"); \
        HIP_TRY(hipSetDevice((p)->ctx->device));                            \
    } while (0)
// n messages of msg_len bytes (device) -> n digests; data_len < msg_len: the bytes past data_len are zero padding (two-to-one buffers)
int32_t te_crh_dev(akp_te_params* p, const uint8_t* d_msgs, size_t n, size_t msg_len, Fr* d_out, hipStream_t s,
        size_t data_len = (size_t)-1);
// ragged batch on device buffers (item i = bytes [d_offsets[i], d_offsets[i+1]) of d_msgs; max_len bounds the longest) and the host-side
// check of an offsets array (monotonic, every item inside the window: AKP_ERR_BAD_LENGTH where the reference panics)
int32_t te_crh_ragged_dev(akp_te_params* p, const uint8_t* d_msgs, const uint64_t* d_offsets, size_t n, size_t max_len, Fr* d_out, hipStream_t s);
int32_t te_ragged_check_offsets(const akp_te_params* p, const uint64_t* offsets, size_t n, size_t* max_len);
// akp_te_params_prepare_compress without the argument checks (the tree builders call it)
int32_t te_prepare_compress(akp_te_params* p, hipStream_t s);
// before a tree build: when leaf and two-to-one hash share a table, build it for the inner nodes first (one build, no extension)
int32_t te_tree_prepare(akp_te_params* leafp, akp_te_params* two, hipStream_t s);
// TwoToOneCRH::compress on device digests (d_right == nullptr: pairs d_left[2i], d_left[2i + 1], a tree level)
int32_t te_compress_dev(akp_te_params* p, const Fr* d_left, const Fr* d_right, size_t n, Fr* d_out, hipStream_t s);

static inline bool pow2_gt1(size_t n) { return n > 1 && (n & (n - 1)) == 0; }
