/*
 * akp.h -- C ABI of the MI355X-native CRH / sponge / Merkle hot path
 *          (drop-in for the native, non-R1CS path of ark-crypto-primitives).
 *
 * The reference has no FFI: its "operator API" is three Rust traits with per-item static
 * functions.  This header is the boundary a Rust shim binds to implement
 * those traits level-wide on the GPU.  Each entry point cites the reference item it replaces
 * (paths relative to /root/reference/crypto-primitives/src).
 *
 * Conventions
 *   - Every call returns int32_t status (AKP_OK == 0).  No exceptions cross the boundary.
 *     akp_last_error() returns a thread-local message for the last failure.
 *   - Field element wire format ("Fr"): 32 bytes = 4 x u64 little-endian limbs, MONTGOMERY form
 *     (x * 2^256 mod p), fully reduced -- byte-identical to ark-ff's in-memory Fp256
 *     (`Fp(BigInt([u64; 4]))`) for BLS12-381 Fr (= Jubjub base field Fq), so a Rust caller
 *     passes `&[Fr]` memory unchanged.  akp_fr_to_mont / akp_fr_from_mont convert canonical
 *     little-endian integers for non-Rust callers.
 *   - Arrays are AoS at the ABI: n x t x 32 B states, n x k x 32 B CRH inputs, n x len byte
 *     messages.  Affine points are x || y (2 x Fr).
 *   - Functions without suffix take HOST pointers (they stage through device scratch owned by
 *     the context and synchronise before returning).  Functions ending in `_dev` take DEVICE
 *     pointers plus a hipStream_t (passed as void*; NULL = HIP's default stream, used verbatim); they
 *     only enqueue work and never synchronise -- this is the zero-copy path used when inputs are
 *     already resident in HBM (bench.py, torch tensors' data_ptr()).
 *   - Ownership: the caller owns every buffer it passes; the library owns parameter handles and
 *     its scratch.  A context (and everything created on it) is not thread-safe: calls that share a
 *     context must be serialised by the caller (the Rust shim keeps one context per thread or a
 *     mutex); distinct contexts are independent and may be used concurrently.
 *     Within one context, calls on DIFFERENT streams are ordered by the library where they share
 *     its scratch (event record + stream wait; nothing blocks on the host).
 *   - Errors mirror the reference: a length the reference would panic on
 *     (crh/pedersen/mod.rs:82-89, crh/bowe_hopwood/mod.rs:121-129) returns AKP_ERR_BAD_LENGTH;
 *     a leaf count that is not a power of two > 1 (merkle_tree/mod.rs:430-433) returns
 *     AKP_ERR_NOT_POW2.  There is NO CPU fallback: without a HIP device every compute call
 *     returns AKP_ERR_HIP.
 */
#ifndef AKP_H
#define AKP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AKP_OK 0
#define AKP_ERR_BAD_LENGTH 1 /* lib.rs:47-52 Error::IncorrectInputLength / length panics */
#define AKP_ERR_BAD_PARAMS 2
#define AKP_ERR_HIP 3
#define AKP_ERR_RCCL 4 /* RCCL could not be loaded / a collective of the multi-device entry points failed */
#define AKP_ERR_NOT_POW2 5

#define AKP_ABI_VERSION 5

typedef struct akp_ctx akp_ctx;
typedef struct akp_poseidon akp_poseidon; /* PoseidonConfig<Fr>, sponge/poseidon/mod.rs:27-45 */
typedef struct akp_te_params akp_te_params; /* pedersen::Parameters / bowe_hopwood::Parameters */
typedef struct akp_sponge akp_sponge;     /* batch of PoseidonSponge<Fr>, sponge/poseidon/mod.rs:54-63 */
typedef struct akp_merkle_tree akp_merkle_tree; /* MerkleTree<P> resident in HBM, merkle_tree/mod.rs:383-396 */
typedef struct akp_multi akp_multi;       /* the GPUs of one node driven from one process + their RCCL communicator */
typedef struct akp_multi_tree akp_multi_tree; /* MerkleTree<P> sharded over the devices of an akp_multi, resident in their HBM */

int32_t akp_abi_version(void);
const char* akp_last_error(void);
/* number of visible HIP devices (0 when there is none or the runtime is unusable) */
int32_t akp_device_count(void);

/* ---- context ---------------------------------------------------------------------------- */
int32_t akp_ctx_create(int32_t device_id, akp_ctx** out);
void akp_ctx_destroy(akp_ctx* ctx);
int32_t akp_ctx_synchronize(akp_ctx* ctx);
/* HBM that ONE precomputed curve table (akp_te_params_create) may occupy on this context's device, bytes.
 *   0 (the default)           320 MiB: the tables stay inside the 256 MiB Infinity Cache (4x256 Pedersen: 16-bit digits, 268 MB;
 *                             63x9 Bowe-Hopwood: groups of 5 chunks, 237 MB) and are built in milliseconds;
 *   AKP_TABLE_BUDGET_DEVICE   a quarter of the device's memory (72 GiB on a 288 GB MI355X: 24-bit digits = 46 GB, groups of 8 chunks
 *                             = 75 GB): -23 % time per hash once the table exists -- for a host that keeps hashing with one parameter set;
 *   any other value           that many bytes.
 * A budget above the default costs NOTHING at the start (round 6): the handle begins to hash on the cache-sized table, the wide one
 * is allocated and built by a thread of the library on a stream of its own (62 ms of kernel time for the 46 GB table; but the FIRST
 * 46 GB allocation on a box whose VRAM nobody has used yet took 1.2 s, and an allocation that follows a large hipFree waits for the
 * driver's wipe of the released memory, ~35 GB/s: profiles/r06_s1 .. r06_s7 -- which is why none of it happens on the caller's thread), and
 * calls switch to it when it is complete.  akp_te_params_prepare waits for it.  If the device cannot hold it the handle stays on the
 * cache-sized table (akp_te_params_table_info says so).  The build is polite -- a few workgroups on a lowest-priority stream: ~0.5 s for
 * the 46 GB table instead of 0.06 s, and the hashing beside it keeps its rate -- but a DEVICE-wide wait of the host (hipDeviceSynchronize)
 * waits for it like for any other work on the device: synchronise streams or events.
 * Handles that exist keep their tables.  akp_ctx_table_budget returns the value the next handle would be created with. */
#define AKP_TABLE_BUDGET_DEVICE ((size_t)-1)
int32_t akp_ctx_set_table_budget(akp_ctx* ctx, size_t bytes);
size_t akp_ctx_table_budget(const akp_ctx* ctx);
/* the hipStream_t (as void*) the HOST-POINTER entry points of this context enqueue their kernels and copies on: lets a caller
 * bracket such a call with its own events (bench.py times the device side of the proof / update entry points this way) or
 * order its own work behind it.  NULL for a NULL context.  The stream belongs to the context. */
void* akp_ctx_stream(akp_ctx* ctx);
/* Effective shader clock (measurement plumbing, no hashing): enqueues ONE wave that runs `chain_len` dependent v_mad_u64_u32 on
 * `stream` and writes three u64 to the device buffer d_out3: [0] shader-clock cycles (s_memtime), [1] ticks of the constant
 * 100 MHz clock (s_memrealtime) over the same interval, [2] the chain's value.  MHz = 100 * [0] / [1]; cycles per dependent
 * multiply-add = [0] / chain_len.  The DPM level sysfs calls "sclk" is a ceiling, not what the ALUs ran at. */
int32_t akp_clock_probe_dev(akp_ctx* ctx, uint32_t chain_len, uint64_t* d_out3, void* stream);

/* ---- pinned host memory (optional) ----------------------------------------------------------- */
/* The host-pointer entry points accept any host memory.  Pageable memory: batches larger than one chunk are cut into chunks
 * whose copy-in / kernel / copy-out overlap on three streams (the copies are staged by the runtime and block the calling
 * thread).  Memory from akp_host_alloc, or registered with akp_host_register:
 *   - the Poseidon batch kernels address it DIRECTLY (zero copy: every item is read once and written once over PCIe, both
 *     directions at the same time) when all buffers of a call are of that kind;
 *   - akp_te_crh_batch, when BOTH `msgs` and `out` are of that kind, runs as ONE gated launch (round 5): the batch is copied in by
 *     asynchronous DMA in chunks of up to 2^17 messages (smaller ones first and last), all copies issued up front, each followed by
 *     an arrival flag; the accumulate kernel is launched once over the whole batch, its workgroups wait for their chunk's flag and
 *     finish their digests themselves (one inversion per workgroup of 512 points); the digests of a chunk leave by DMA as soon as
 *     its workgroups have reported -- never by zero copy: in-place reads make every workgroup wait for PCIe at the same moments and
 *     in-place 16-byte digest stores cross PCIe at 17 GB/s (measured, profiles/r04_s2 .. r04_s3).  Pinned on one side only behaves
 *     like pageable memory; a second caller on the same device while a gated launch is in flight, or a stack on which the gate
 *     cannot work, gets round 4's chunked launches (same digests).  Round 6: which of the two forms is faster depends on where the
 *     runtime happens to put the call's streams (hardware queue, pipe of the command processor: gated 3.4 - 5.1 ms against chunked
 *     3.4 - 5.0 ms per 2^20 Pedersen hashes over the placements measured, profiles/r06_s41 ... s46), so a context MEASURES: four calls
 *     of either form, in turns, per (parameter set, message length) -- four such shapes are remembered --, then the faster one, every 32nd call given to the other.  The environment
 *     variable AKP_TE_PINNED_FORM=gated | chunked pins the form (read at every call; anything else: measure).  The choice, and every later
 *     change of it, is announced behind the successful call that made it: akp_last_error() then starts with "note: the pinned curve-hash batches".
 * Pinned buffers gain 20 % for the Poseidon batches (3.5e8 against 2.9e8 permutations/s) and, since round 5, 15 - 35 % for the curve
 * hashes (per 2^20 hashes, median wall time: Pedersen 4x256 3.4 ms pinned against 4.0 ms pageable with the HBM-sized table, 4.0
 * against 5.0 ms with the default one; Bowe-Hopwood 63x9 64-byte inputs 2.0 - 2.3 ms against 2.8 - 3.5 ms; profiles/r05_s16) -- the
 * resident launch takes 2.3 - 3.3 / 1.4 - 1.9 ms, the rest is PCIe (134 MB in, 67 MB out per 2^20 Pedersen hashes: the copy-in alone
 * takes 2.5 ms) and, with the default table, compute that
 * shares the device with the copies.  Register a buffer as a whole: the runtime rejects copies that straddle registered and unregistered memory. */
int32_t akp_host_alloc(size_t bytes, void** out);
int32_t akp_host_free(void* p);
int32_t akp_host_register(void* p, size_t bytes);
int32_t akp_host_unregister(void* p);

/* ---- field helpers (host side, no device needed) ------------------------------------------ */
/* canonical little-endian integers (must be < p) <-> Montgomery wire format; n elements */
int32_t akp_fr_to_mont(const uint64_t* canonical, uint64_t* mont, size_t n);
int32_t akp_fr_from_mont(const uint64_t* mont, uint64_t* canonical, size_t n);

/* ---- Poseidon parameters ------------------------------------------------------------------ */
/* PoseidonConfig::new (sponge/poseidon/mod.rs:191-217): ark is [full+partial][t], mds is [t][t],
 * t = rate + capacity, all Fr wire format.  ctx may be NULL for a host-only handle (parameter
 * generation / inspection without a GPU); compute calls on such a handle return AKP_ERR_HIP. */
int32_t akp_poseidon_params_create(akp_ctx* ctx, uint32_t full_rounds, uint32_t partial_rounds, uint64_t alpha,
                                   uint32_t rate, uint32_t capacity, const uint64_t* ark, const uint64_t* mds,
                                   akp_poseidon** out);
/* PoseidonDefaultConfigField::get_default_poseidon_parameters for BLS12-381 Fr
 * (sponge/poseidon/traits.rs:69-155: Grain LFSR + Cauchy MDS; per-field table sponge/test.rs:13-31).
 * rate in 2..8.  Returns AKP_ERR_BAD_PARAMS where the reference returns None. */
int32_t akp_poseidon_default_params(akp_ctx* ctx, uint32_t rate, int32_t optimized_for_weights, akp_poseidon** out);
void akp_poseidon_params_destroy(akp_poseidon* p);
/* read back the dimensions; any out pointer may be NULL */
int32_t akp_poseidon_params_dims(const akp_poseidon* p, uint32_t* full_rounds, uint32_t* partial_rounds,
                                 uint64_t* alpha, uint32_t* rate, uint32_t* capacity);
/* copy out ark ([full+partial][t]) and mds ([t][t]) in wire format; either may be NULL */
int32_t akp_poseidon_params_export(const akp_poseidon* p, uint64_t* ark, uint64_t* mds);

/* ---- Poseidon batches ----------------------------------------------------------------------- */
/* PoseidonSponge::permute (sponge/poseidon/mod.rs:98-121) on n states of t Fr each, in place. */
int32_t akp_poseidon_permute_batch(akp_poseidon* p, uint64_t* states, size_t n);
int32_t akp_poseidon_permute_batch_dev(akp_poseidon* p, uint64_t* d_states, size_t n, void* stream);
/* poseidon::CRH::evaluate (crh/poseidon/mod.rs:30-40): n inputs of elems_per_input Fr -> n Fr.
 * elems_per_input may be 0 (hash of the empty slice). */
int32_t akp_poseidon_crh_batch(akp_poseidon* p, const uint64_t* inputs, size_t n, size_t elems_per_input,
                               uint64_t* out);
int32_t akp_poseidon_crh_batch_dev(akp_poseidon* p, const uint64_t* d_inputs, size_t n, size_t elems_per_input,
                                   uint64_t* d_out, void* stream);
/* The same for inputs of DIFFERENT lengths (crh/poseidon/mod.rs:30-40 takes any &[F]): input i = elements
 * [offsets[i], offsets[i+1]) of `inputs` (offsets: n + 1 non-decreasing element indices; an empty input is the hash of the empty
 * slice).  t = 3 parameter sets run one launch with per-lane lengths, the items ordered by their permutation count on the device
 * so that a wave's lanes finish together; other widths are grouped by length on the host (host entry point only). */
int32_t akp_poseidon_crh_batch_ragged(akp_poseidon* p, const uint64_t* inputs, const uint64_t* offsets, size_t n, uint64_t* out);
int32_t akp_poseidon_crh_batch_ragged_dev(akp_poseidon* p, const uint64_t* d_inputs, const uint64_t* d_offsets, size_t n,
                                          uint64_t* d_out, void* stream);
/* poseidon::TwoToOneCRH::{evaluate,compress} (crh/poseidon/mod.rs:58-79): out[i] = H(left[i], right[i]). */
int32_t akp_poseidon_two_to_one_batch(akp_poseidon* p, const uint64_t* left, const uint64_t* right, size_t n,
                                      uint64_t* out);
int32_t akp_poseidon_two_to_one_batch_dev(akp_poseidon* p, const uint64_t* d_left, const uint64_t* d_right, size_t n,
                                          uint64_t* d_out, void* stream);

/* name of the kernel a batch of n items is routed to (crh = 0: permutation, 1: CRH / two-to-one / tree level): the
 * t = 3 register kernels above 2^15 items, the wave-per-lane latency kernels below, the LDS-file kernels for t != 3.
 * Lets a parity check state which kernel it exercised.  Static string. */
const char* akp_poseidon_kernel_for(const akp_poseidon* p, size_t n, int32_t crh);

/* ---- batched duplex sponge (CryptographicSponge / FieldBasedCryptographicSponge) ------------ */
/* `batch` independent PoseidonSponge<Fr> instances that follow the same absorb/squeeze schedule
 * (sponge/poseidon/mod.rs:223-257, 324-344; state machine sponge/mod.rs:195-206).  State lives
 * on the device; the duplex mode bookkeeping is host-side and shared by the batch. */
int32_t akp_sponge_create(akp_poseidon* p, size_t batch, akp_sponge** out);
void akp_sponge_destroy(akp_sponge* s);
/* absorb: elems is [batch][elems_per_instance] Fr (host).  elems_per_instance == 0 is a no-op. */
int32_t akp_sponge_absorb(akp_sponge* s, const uint64_t* elems, size_t elems_per_instance);
/* squeeze_native_field_elements(n): out is [batch][n] Fr (host) */
int32_t akp_sponge_squeeze(akp_sponge* s, uint64_t* out, size_t n_per_instance);
/* the same on DEVICE buffers ([batch][k] Fr in, [batch][n] Fr out), enqueued on `stream` without synchronising: a batch of
 * sponges driven entirely from HBM (the state never leaves the device either way).  Calls on one sponge must be issued in
 * program order on one stream (the duplex bookkeeping is host-side). */
int32_t akp_sponge_absorb_dev(akp_sponge* s, const uint64_t* d_elems, size_t elems_per_instance, void* stream);
int32_t akp_sponge_squeeze_dev(akp_sponge* s, uint64_t* d_out, size_t n_per_instance, void* stream);
/* SpongeExt::{into_state,from_state} (sponge/mod.rs:184-191): state is [batch][t] Fr;
 * mode 0 = Absorbing{index}, 1 = Squeezing{index}. */
int32_t akp_sponge_get_state(akp_sponge* s, uint64_t* state, int32_t* mode, uint32_t* index);
int32_t akp_sponge_set_state(akp_sponge* s, const uint64_t* state, int32_t mode, uint32_t index);

/* ---- Pedersen / Bowe-Hopwood over Jubjub (ark_ed_on_bls12_381) -------------------------------- */
#define AKP_TE_PEDERSEN 0     /* crh/pedersen/mod.rs: digest = affine point x||y (2 Fr) */
#define AKP_TE_BOWE_HOPWOOD 1 /* crh/bowe_hopwood/mod.rs: digest = x coordinate (1 Fr) */
#define AKP_TE_PEDERSEN_X 2   /* crh/injective_map/mod.rs:16-108: PedersenCRHCompressor / PedersenTwoToOneCRHCompressor with
                               * TECompressor -- the Pedersen hash followed by the injective map (x, y) -> x: digest = 1 Fr;
                               * compress serialises the two x coordinates (64 bytes) into the (W*N)/8-byte buffer.  The
                               * leaf / two-to-one hashes of the reference's R1CS Merkle tests (merkle_tree/tests/constraints.rs). */
/* Parameters { generators } (crh/pedersen/mod.rs:28-31, crh/bowe_hopwood/mod.rs:33-37):
 * generators is [num_windows][window_size] affine points (x||y, Fr wire format), used verbatim
 * (no assumption that generators[i][j] is a multiple of generators[i][0]).
 * For AKP_TE_BOWE_HOPWOOD window_size must be <= 63 (setup bound, bowe_hopwood/mod.rs:81-101).
 * SHARED TABLES (the reference's `Parameters: Sync`, crh/mod.rs:22 -- one value borrowed by every rayon worker,
 * merkle_tree/mod.rs:417,458,494): handles created with the same generators, window and table shape on contexts of the SAME
 * device attach to ONE set of precomputed tables in that device's HBM (process-wide, reference-counted, freed with the last
 * handle; thread-safe).  N worker threads with a context each hold one table, built once, whoever hashes first. */
int32_t akp_te_params_create(akp_ctx* ctx, int32_t kind, uint32_t window_size, uint32_t num_windows,
                             const uint64_t* generators_affine, akp_te_params** out);
/* The handle holds a precomputed table in HBM (no counterpart in the reference, which adds generators bit by bit): a hash is one
 * curve addition per table step, a step covers `digit_bits` message bits (Pedersen) / `group` 3-bit chunks (Bowe-Hopwood), and
 * every extra bit doubles the table -- memory for time.  akp_te_params_create picks the widest table the context's table
 * budget admits (akp_ctx_set_table_budget above: by default the 268 MB / 237 MB tables that fit the Infinity Cache; with
 * AKP_TABLE_BUDGET_DEVICE on an idle MI355X 24-bit digits = 46 GB for a 4x256 window, 43 steps per 128-byte message instead of
 * 64, -23 % time; groups of 8 chunks = 75 GB for a 63x9 window).  This form fixes the shape instead: digit_bits 2..24 / group 1..8, 0 = from the budget.  The digests do
 * not depend on the shape.
 * The table is BUILT FOR THE MESSAGE LENGTHS THAT ARRIVE: creation allocates kilobytes, the first hash of a length builds the
 * digits / groups that length touches (a 63x9 handle that only hashes a tree's 32- and 64-byte nodes holds 22.5 of the 75 GB),
 * a longer message later extends the table: ONCE, to the complete table (a new allocation is built; the old table is kept until the
 * last handle of these generators is destroyed, the device is not drained).  Nothing a launch was given is freed while a handle
 * of the table is alive: a HIP graph captured after akp_te_params_prepare stays valid whatever other handles of the same
 * generators hash later.  An explicit shape is built by the call that first needs it (or by akp_te_params_prepare); only shapes
 * chosen by the budget start on the cache-sized table.  akp_te_params_info reports what a call would use at the moment. */
int32_t akp_te_params_create_shaped(akp_ctx* ctx, int32_t kind, uint32_t window_size, uint32_t num_windows,
                                    const uint64_t* generators_affine, uint32_t digit_bits_or_group, akp_te_params** out);
void akp_te_params_destroy(akp_te_params* p);
/* Build NOW what hashing messages of msg_len bytes needs (the wide table up to that length; for Bowe-Hopwood also the remainder
 * table of that length) and wait for it: the host chooses the moment instead of meeting the build inside its first hash (explicit
 * shapes) or hashing on the cache-sized table until the background build has finished (budget-chosen shapes).
 * akp_te_params_prepare_compress does the same for TwoToOneCRH::compress / the inner levels of a tree (two serialised digests in
 * the (W*N)/8-byte buffer).  Without it the table work of an explicit shape happens lazily inside the first call that needs it --
 * also a `_dev` call, which then allocates and waits ONCE (never while its stream is being captured into a graph: that call fails
 * with AKP_ERR_BAD_PARAMS and names this function). */
int32_t akp_te_params_prepare(akp_te_params* p, size_t msg_len);
int32_t akp_te_params_prepare_compress(akp_te_params* p);
/* phases of the last build / extension of a handle's wide table, host wall clock (milliseconds) */
typedef struct akp_te_build_report {
    uint64_t table_bytes;   /* bytes of the wide table after that build */
    uint32_t shape;         /* digit bits (Pedersen) / chunks per group (Bowe-Hopwood) */
    uint32_t units_from, units_to, units_total; /* digits / groups it added: [units_from, units_to) of units_total */
    uint32_t in_background; /* 1: built by the library's thread while the handle hashed on the cache-sized table */
    /* 0: the handle has ONE table (default budget or explicit shape); 1: a wide table is wanted and not complete yet (the handle
     * hashes on the cache-sized one); 2: calls use the wide table; 3: the wide table could not be built, the handle stays on the
     * cache-sized one (akp_last_error of the akp_te_params_prepare that failed, or `note`) */
    uint32_t upgrade_state;
    double alloc_ms;        /* hipMalloc of the table and of its part tables */
    double parts_ms;        /* part-table kernels */
    double combine_ms;      /* te_build_combine_kernel: one curve addition per entry, the first writes to the new memory */
    double constants_ms;    /* Bowe-Hopwood remainder table / tail constant of the message shape */
    double total_ms;
    char note[96];          /* why upgrade_state is 3, else empty */
} akp_te_build_report;
/* the shared table behind a handle (any pointer may be NULL): an identifier that is equal for two handles iff they use the same
 * tables, the number of handles attached to it, how often its wide table has been built or extended so far, and the phases of the
 * last of those builds.  For a handle with a budget-chosen wide table all four describe THAT table. */
int32_t akp_te_params_table_info(const akp_te_params* p, uint64_t* table_id, uint32_t* handles_attached, uint64_t* wide_builds,
                                 akp_te_build_report* last_build);
/* Tuning facts of a handle (any pointer may be NULL): digit width of the Pedersen table / chunks per table step of the
 * Bowe-Hopwood table, whether the Pedersen table is the signed-subset one, bytes of precomputed tables in HBM, and the
 * number of table steps (curve additions + 1) an input of msg_len bytes takes -- of the table a call with msg_len-byte messages
 * would use NOW (the cache-sized one while a budget-chosen wide table is not complete for that length). */
int32_t akp_te_params_info(const akp_te_params* p, uint32_t* digit_bits_or_group, int32_t* signed_subset,
                           size_t* table_bytes, size_t msg_len, uint32_t* steps);
/* bytes one table entry occupies in HBM = what one lane gathers per table step (128: one cache line per entry) */
uint32_t akp_te_entry_bytes(void);
/* pedersen::CRH::evaluate (crh/pedersen/mod.rs:76-129) / bowe_hopwood::CRH::evaluate
 * (crh/bowe_hopwood/mod.rs:114-186): n messages of msg_len bytes each ->
 * n digests (2 Fr for Pedersen, 1 Fr for Bowe-Hopwood).
 * The FIRST call with a longer message than any before may extend the handle's table (see akp_te_params_create_shaped).
 * Bowe-Hopwood: the FIRST call with a new message shape (length, and for two-to-one buffers the length of the zero padding) may
 * build one more small table (the chunks the shape leaves after its last full group, with the padding's constant folded in:
 * <= 268 MB, milliseconds, kept in the handle; up to eight shapes) -- that call waits for the context's own stream once; every
 * later call with the shape only enqueues. */
int32_t akp_te_crh_batch(akp_te_params* p, const uint8_t* msgs, size_t n, size_t msg_len, uint64_t* out);
int32_t akp_te_crh_batch_dev(akp_te_params* p, const uint8_t* d_msgs, size_t n, size_t msg_len, uint64_t* d_out,
                             void* stream);
/* The same for messages of DIFFERENT lengths: the reference hashes every input with ITS length -- Pedersen pads each input with
 * zero bits to the window (crh/pedersen/mod.rs:82-99: padding does not change the digest), Bowe-Hopwood pads each input to a
 * multiple of 3 bits only (crh/bowe_hopwood/mod.rs:131-138: the digest DEPENDS on the length, a host cannot pad).  Message i =
 * bytes [offsets[i], offsets[i+1]) of msgs (offsets: n + 1 non-decreasing byte offsets).  One launch, per-lane step counts; the
 * items are ordered by step count on the device (counting sort, longest first) so that a wave's lanes finish together.  A message
 * longer than the window is AKP_ERR_BAD_LENGTH for the whole call (the reference panics on that item).  `_dev`: max_len is the
 * caller's bound on the longest message (the table is built for it).  The device-resident offsets are the caller's: item i reads the
 * bytes [offsets[i], offsets[i] + min(len_i, max_len)) of d_msgs and nothing else -- an item longer than max_len is cut there (its
 * digest is that of the cut message), a pair of offsets that decreases is the empty message; table reads are clamped to what is built. */
int32_t akp_te_crh_batch_ragged(akp_te_params* p, const uint8_t* msgs, const uint64_t* offsets, size_t n, uint64_t* out);
int32_t akp_te_crh_batch_ragged_dev(akp_te_params* p, const uint8_t* d_msgs, const uint64_t* d_offsets, size_t n, size_t max_len,
                                    uint64_t* d_out, void* stream);
/* TwoToOneCRH::evaluate (crh/pedersen/mod.rs:158-182, crh/bowe_hopwood/mod.rs:202-227):
 * left/right are n x half_len bytes; buffer = (W*N)/8 zero bytes overwritten by left||right
 * (zip-truncated), then CRH::evaluate. */
int32_t akp_te_two_to_one_batch(akp_te_params* p, const uint8_t* left, const uint8_t* right, size_t n,
                                size_t half_len, uint64_t* out);
/* TwoToOneCRH::compress (crh/pedersen/mod.rs:187-197, crh/bowe_hopwood/mod.rs:229-239): inputs are
 * digests (wire format); they are serialised uncompressed (canonical LE) on the device first. */
int32_t akp_te_compress_batch(akp_te_params* p, const uint64_t* left, const uint64_t* right, size_t n, uint64_t* out);

/* ---- Merkle tree ---------------------------------------------------------------------------- */
/* MerkleTree::new (merkle_tree/mod.rs:411-523) with Poseidon leaf CRH + Poseidon TwoToOneCRH and
 * IdentityDigestConverter (config shape merkle_tree/tests/mod.rs:198-206).
 * leaves: n_leaves x leaf_len Fr.  leaf_nodes: n_leaves Fr.  non_leaf_nodes: n_leaves-1 Fr in the
 * reference's heap order (root at 0; level l occupies [2^l - 1, 2^(l+1) - 1)).
 * Either output may be NULL in the host variant (skips that copy); root_out (1 Fr) may be NULL. */
int32_t akp_merkle_build_poseidon(akp_poseidon* leaf_params, akp_poseidon* two_to_one_params, const uint64_t* leaves,
                                  size_t n_leaves, size_t leaf_len, uint64_t* leaf_nodes, uint64_t* non_leaf_nodes,
                                  uint64_t* root_out);
int32_t akp_merkle_build_poseidon_dev(akp_poseidon* leaf_params, akp_poseidon* two_to_one_params,
                                      const uint64_t* d_leaves, size_t n_leaves, size_t leaf_len, uint64_t* d_leaf_nodes,
                                      uint64_t* d_non_leaf_nodes, void* stream);
/* MerkleTree::new_with_leaf_digest (merkle_tree/mod.rs:424-523): inner levels only. */
int32_t akp_merkle_inner_poseidon_dev(akp_poseidon* two_to_one_params, const uint64_t* d_leaf_nodes, size_t n_leaves,
                                      uint64_t* d_non_leaf_nodes, void* stream);
/* host-pointer form of the above: leaf_nodes in (n_leaves Fr), non_leaf_nodes out (n_leaves - 1 Fr) */
int32_t akp_merkle_inner_poseidon(akp_poseidon* two_to_one_params, const uint64_t* leaf_nodes, size_t n_leaves,
                                  uint64_t* non_leaf_nodes);
/* MerkleTree::new_with_leaf_digest for Pedersen / Bowe-Hopwood digests (ByteDigestConverter): leaf_nodes are
 * n_leaves digests (2 Fr / 1 Fr each) */
int32_t akp_merkle_inner_te(akp_te_params* two_to_one_params, const uint64_t* leaf_nodes, size_t n_leaves,
                            uint64_t* non_leaf_nodes);
int32_t akp_merkle_inner_te_dev(akp_te_params* two_to_one_params, const uint64_t* d_leaf_nodes, size_t n_leaves,
                                uint64_t* d_non_leaf_nodes, void* stream);
/* MerkleTree::new over byte leaves with Pedersen or Bowe-Hopwood hashes and ByteDigestConverter
 * (merkle_tree/mod.rs:67-78; config shape merkle_tree/tests/mod.rs:13-33).  Both parameter sets
 * must have the same kind.  Digest = 2 Fr (Pedersen) / 1 Fr (Bowe-Hopwood) per node. */
int32_t akp_merkle_build_te(akp_te_params* leaf_params, akp_te_params* two_to_one_params, const uint8_t* leaves,
                            size_t n_leaves, size_t leaf_len, uint64_t* leaf_nodes, uint64_t* non_leaf_nodes,
                            uint64_t* root_out);
int32_t akp_merkle_build_te_dev(akp_te_params* leaf_params, akp_te_params* two_to_one_params, const uint8_t* d_leaves,
                                size_t n_leaves, size_t leaf_len, uint64_t* d_leaf_nodes, uint64_t* d_non_leaf_nodes,
                                void* stream);

/* MerkleTree::new over leaves of DIFFERENT lengths with every buffer in device memory (the resident handle's form:
 * akp_merkle_tree_build_*_ragged below): leaf i = elements (Poseidon; t = 3 leaf parameters) / bytes (te) [d_offsets[i], d_offsets[i+1])
 * of d_leaves; max_len bounds the longest byte leaf.  Enqueue only. */
int32_t akp_merkle_build_poseidon_ragged_dev(akp_poseidon* leaf_params, akp_poseidon* two_to_one_params, const uint64_t* d_leaves,
                                             const uint64_t* d_offsets, size_t n_leaves, uint64_t* d_leaf_nodes,
                                             uint64_t* d_non_leaf_nodes, void* stream);
int32_t akp_merkle_build_te_ragged_dev(akp_te_params* leaf_params, akp_te_params* two_to_one_params, const uint8_t* d_leaves,
                                       const uint64_t* d_offsets, size_t n_leaves, size_t max_len, uint64_t* d_leaf_nodes,
                                       uint64_t* d_non_leaf_nodes, void* stream);

/* ---- Merkle proofs (merkle_tree/mod.rs:146-213, 536-579) -------------------------------------------- */
/* MerkleTree::generate_proof for m leaf indices at once (get_leaf_sibling_hash :536-544 + compute_auth_path
 * :547-569): pure index arithmetic over the heap-ordered arrays.  fe_per_digest = 1 (Poseidon / Bowe-Hopwood)
 * or 2 (Pedersen).  auth_paths is [m][depth] digests, root side first, depth = log2(n_leaves) - 1.
 * Host variant works on host arrays (no device needed); _dev gathers from a tree resident in HBM. */
int32_t akp_merkle_gather_paths(const uint64_t* leaf_nodes, const uint64_t* non_leaf_nodes, size_t n_leaves,
                                uint32_t fe_per_digest, const uint64_t* leaf_indices, size_t m,
                                uint64_t* leaf_sibling_hashes, uint64_t* auth_paths);
int32_t akp_merkle_gather_paths_dev(akp_ctx* ctx, const uint64_t* d_leaf_nodes, const uint64_t* d_non_leaf_nodes,
                                    size_t n_leaves, uint32_t fe_per_digest, const uint64_t* d_leaf_indices, size_t m,
                                    uint64_t* d_leaf_sibling_hashes, uint64_t* d_auth_paths, void* stream);
/* Path::verify (merkle_tree/mod.rs:172-212) for m paths of equal depth in one call: leaf hash, then depth + 1
 * two-to-one levels, all m paths advancing together on the GPU.  ok_out[i] = 1 iff path i reproduces `root`.
 * leaves: m x leaf_len Fr (Poseidon) / m x leaf_len bytes (te).  All pointers are host pointers. */
int32_t akp_merkle_verify_paths_poseidon(akp_poseidon* leaf_params, akp_poseidon* two_to_one_params, const uint64_t* root,
                                         const uint64_t* leaves, size_t m, size_t leaf_len, const uint64_t* leaf_indices,
                                         const uint64_t* leaf_sibling_hashes, const uint64_t* auth_paths, size_t depth,
                                         uint8_t* ok_out);
/* the same with every buffer in device memory (paths from akp_merkle_gather_paths_dev are verified where they lie); d_ok_out: m
 * bytes; enqueues on `stream`, no synchronisation */
int32_t akp_merkle_verify_paths_poseidon_dev(akp_poseidon* leaf_params, akp_poseidon* two_to_one_params, const uint64_t* d_root,
                                             const uint64_t* d_leaves, size_t m, size_t leaf_len, const uint64_t* d_leaf_indices,
                                             const uint64_t* d_leaf_sibling_hashes, const uint64_t* d_auth_paths, size_t depth,
                                             uint8_t* d_ok_out, void* stream);
int32_t akp_merkle_verify_paths_te(akp_te_params* leaf_params, akp_te_params* two_to_one_params, const uint64_t* root,
                                   const uint8_t* leaves, size_t m, size_t leaf_len, const uint64_t* leaf_indices,
                                   const uint64_t* leaf_sibling_hashes, const uint64_t* auth_paths, size_t depth,
                                   uint8_t* ok_out);


/* ---- Merkle tree resident in HBM (MerkleTree<P>, merkle_tree/mod.rs:383-725) ------------------------------------ */
/* MerkleTree::new (:411-422): the two node vectors stay in device memory (leaf_nodes[n], non_leaf_nodes[n-1] in
 * heap order); the host asks for what it needs (root, proofs, the vectors).  The tree pins its parameter handles:
 * destroying one while the tree lives only defers its release to akp_merkle_tree_destroy (likewise akp_sponge).  leaves: n x leaf_len Fr (Poseidon) / n x leaf_len bytes (Pedersen, Bowe-Hopwood). */
int32_t akp_merkle_tree_build_poseidon(akp_poseidon* leaf_params, akp_poseidon* two_to_one_params, const uint64_t* leaves,
                                       size_t n_leaves, size_t leaf_len, akp_merkle_tree** out);
int32_t akp_merkle_tree_build_te(akp_te_params* leaf_params, akp_te_params* two_to_one_params, const uint8_t* leaves,
                                 size_t n_leaves, size_t leaf_len, akp_merkle_tree** out);
/* MerkleTree::new over leaves of DIFFERENT lengths: the reference maps LeafHash::evaluate over any iterator of leaves
 * (:411-422), each leaf hashed with its own length (akp_poseidon_crh_batch_ragged / akp_te_crh_batch_ragged for the leaf level).
 * Leaf i = elements (Poseidon) / bytes (te) [offsets[i], offsets[i+1]) of `leaves`; offsets[0] need not be 0. */
int32_t akp_merkle_tree_build_poseidon_ragged(akp_poseidon* leaf_params, akp_poseidon* two_to_one_params, const uint64_t* leaves,
                                              const uint64_t* offsets, size_t n_leaves, akp_merkle_tree** out);
int32_t akp_merkle_tree_build_te_ragged(akp_te_params* leaf_params, akp_te_params* two_to_one_params, const uint8_t* leaves,
                                        const uint64_t* offsets, size_t n_leaves, akp_merkle_tree** out);
/* the same from leaves that are already in device memory (synchronises the context stream before returning) */
int32_t akp_merkle_tree_build_poseidon_dev(akp_poseidon* leaf_params, akp_poseidon* two_to_one_params, const uint64_t* d_leaves,
                                           size_t n_leaves, size_t leaf_len, akp_merkle_tree** out);
int32_t akp_merkle_tree_build_te_dev(akp_te_params* leaf_params, akp_te_params* two_to_one_params, const uint8_t* d_leaves,
                                     size_t n_leaves, size_t leaf_len, akp_merkle_tree** out);
/* MerkleTree::new_with_leaf_digest (:424-523); with all-default digests this is MerkleTree::blank (:400-408).
 * leaf_digests: n_leaves digests (1 Fr; 2 Fr for Pedersen). */
int32_t akp_merkle_tree_from_digests_poseidon(akp_poseidon* leaf_params, akp_poseidon* two_to_one_params,
                                              const uint64_t* leaf_digests, size_t n_leaves, akp_merkle_tree** out);
int32_t akp_merkle_tree_from_digests_te(akp_te_params* leaf_params, akp_te_params* two_to_one_params,
                                        const uint64_t* leaf_digests, size_t n_leaves, akp_merkle_tree** out);
void akp_merkle_tree_destroy(akp_merkle_tree* t);
/* n_leaves, Fr per digest, height() (:531-533); any out pointer may be NULL */
int32_t akp_merkle_tree_info(const akp_merkle_tree* t, size_t* n_leaves, uint32_t* fe_per_digest, size_t* height);
/* MerkleTree::root (:526-528) */
int32_t akp_merkle_tree_root(akp_merkle_tree* t, uint64_t* root_out);
/* copies of leaf_nodes / non_leaf_nodes (what GpuMerkleTree<P> hands to code that reads the reference's fields);
 * either may be NULL */
int32_t akp_merkle_tree_export(akp_merkle_tree* t, uint64_t* leaf_nodes, uint64_t* non_leaf_nodes);
/* device addresses of the two vectors (valid until the tree is destroyed), for the `_dev` entry points */
int32_t akp_merkle_tree_device_ptrs(akp_merkle_tree* t, uint64_t** d_leaf_nodes, uint64_t** d_non_leaf_nodes);
/* MerkleTree::generate_proof (:572-579) for m leaf indices: leaf_sibling_hashes [m], auth_paths [m][log2(n) - 1]
 * (root side first) */
int32_t akp_merkle_tree_gather_paths(akp_merkle_tree* t, const uint64_t* leaf_indices, size_t m,
                                     uint64_t* leaf_sibling_hashes, uint64_t* auth_paths);
/* MerkleTree::generate_multi_proof (:592-625) for m SORTED, DISTINCT leaf indexes, prefix_encode_path (:795-805) done on the device: the
 * dense paths never leave HBM, only leaf_sibling_hashes [m], prefix_lengths [m] and the concatenated suffixes come back
 * (*n_suffix_digests of them; suffix_cap = m * (log2(n) - 1) always suffices; a smaller buffer that turns out too small is
 * AKP_ERR_BAD_LENGTH with the needed count in *n_suffix_digests).  The same flat form as akp_merkle_multipath_encode. */
int32_t akp_merkle_tree_multi_proof(akp_merkle_tree* t, const uint64_t* leaf_indices, size_t m, uint64_t* leaf_sibling_hashes,
                                    uint64_t* prefix_lengths, uint64_t* suffixes, size_t suffix_cap, size_t* n_suffix_digests);
/* MerkleTree::update (:692-702), batched: equal to update(leaf_indices[k], new_leaves[k]) for k = 0..m-1 in order (a
 * repeated index keeps its last leaf); every level is one hash launch over the distinct touched nodes.  An index out
 * of range (the reference asserts) is AKP_ERR_BAD_PARAMS and leaves the tree untouched. */
int32_t akp_merkle_tree_update_batch(akp_merkle_tree* t, const uint64_t* leaf_indices, const void* new_leaves, size_t m,
                                     size_t leaf_len);
/* MerkleTree::check_update (:707-725): *ok = 1 and the tree is updated iff the new root equals asserted_new_root;
 * otherwise *ok = 0 and the tree is unchanged. */
int32_t akp_merkle_tree_check_update(akp_merkle_tree* t, uint64_t leaf_index, const void* new_leaf, size_t leaf_len,
                                     const uint64_t* asserted_new_root, int32_t* ok);

/* ---- MultiPath (merkle_tree/mod.rs:215-331, 592-625, 795-817) ------------------------------------------------------ */
/* prefix_encode_path (:795-805) over m dense auth paths [m][depth] digests (in the order of the sorted, de-duplicated
 * leaf indexes of generate_multi_proof): prefix_lengths [m], suffixes concatenated (capacity m * depth digests),
 * *n_suffix_digests = digests written.  Host only. */
int32_t akp_merkle_multipath_encode(const uint64_t* auth_paths, size_t m, size_t depth, uint32_t fe_per_digest,
                                    uint64_t* prefix_lengths, uint64_t* suffixes, size_t* n_suffix_digests);
/* prefix_decode_path (:807-817): the inverse.  Host only. */
int32_t akp_merkle_multipath_decode(const uint64_t* prefix_lengths, const uint64_t* suffixes, size_t n_suffix_digests, size_t m,
                                    size_t depth, uint32_t fe_per_digest, uint64_t* auth_paths);
/* MultiPath::verify (:262-331): leaves in the order of leaf_indexes; depth = auth_paths_suffixes[0].len().  *ok = 1 iff
 * the paths lead to `root`, with the reference's memoisation (a tree node shared by several paths is computed once,
 * from the first path that reaches it); each level is one hash launch over the distinct nodes. */
int32_t akp_merkle_verify_multipath_poseidon(akp_poseidon* leaf_params, akp_poseidon* two_to_one_params, const uint64_t* root,
                                             const uint64_t* leaves, size_t m, size_t leaf_len, const uint64_t* leaf_indexes,
                                             const uint64_t* leaf_siblings_hashes, const uint64_t* prefix_lengths,
                                             const uint64_t* suffixes, size_t n_suffix_digests, size_t depth, int32_t* ok);
int32_t akp_merkle_verify_multipath_te(akp_te_params* leaf_params, akp_te_params* two_to_one_params, const uint64_t* root,
                                       const uint8_t* leaves, size_t m, size_t leaf_len, const uint64_t* leaf_indexes,
                                       const uint64_t* leaf_siblings_hashes, const uint64_t* prefix_lengths,
                                       const uint64_t* suffixes, size_t n_suffix_digests, size_t depth, int32_t* ok);

/* ---- ark-serialize byte formats (host only: no device, no context) ---------------------------------------------------------- */
/* Derived CanonicalSerialize / CanonicalDeserialize of the structs that cross the boundary, fields in declaration order:
 * PoseidonConfig (sponge/poseidon/mod.rs:26-45), pedersen / bowe_hopwood Parameters (crh/pedersen/mod.rs:28-31,
 * crh/bowe_hopwood/mod.rs:33-37), Path (merkle_tree/mod.rs:139-152), MultiPath (:239-254).  usize / u64: 8 bytes LE; Vec<T>: u64
 * length + elements; Fq: 32 bytes LE of the canonical integer; affine point: x || y (compress = 0) or y with the sign of x in
 * the top bit of the last byte (compress = 1); projective points serialise as their affine form.
 * Writers: `out` may be NULL (size query); *out_len always receives the size; a buffer that is too small is
 * AKP_ERR_BAD_LENGTH.  Readers: truncated input / trailing bytes are AKP_ERR_BAD_LENGTH, non-canonical field elements and
 * (validate != 0, ark-serialize's Validate::Yes) points off the curve or outside the prime-order subgroup AKP_ERR_BAD_PARAMS;
 * validate = 0 is deserialize_*_unchecked. */
/* n digests back to back, no length prefix (fe_per_digest 1: Fq -- Poseidon / Bowe-Hopwood; 2: affine point -- Pedersen) */
int32_t akp_serialize_digests(const uint64_t* digests, size_t n, uint32_t fe_per_digest, int32_t compress, uint8_t* out,
                              size_t out_cap, size_t* out_len);
int32_t akp_deserialize_digests(const uint8_t* in, size_t in_len, size_t n, uint32_t fe_per_digest, int32_t compress,
                                int32_t validate, uint64_t* digests);
/* PoseidonConfig: both modes write the same bytes.  The reader builds a handle (ctx may be NULL: host-only handle) and therefore
 * refuses (AKP_ERR_BAD_PARAMS) field values no handle can be built from -- ark rows != full + partial rounds, a row or mds not
 * t = rate + capacity wide, odd full_rounds, alpha 0 -- which the reference's derive reads without complaint and panics on at the
 * first permutation.  Likewise akp_deserialize_multipath wants its four vectors of one length and akp_deserialize_te_parameters
 * windows of one size (tests/test_serialize_fuzz.py pins these differences against the oracle reader). */
int32_t akp_serialize_poseidon_config(const akp_poseidon* p, uint8_t* out, size_t out_cap, size_t* out_len);
int32_t akp_deserialize_poseidon_config(akp_ctx* ctx, const uint8_t* in, size_t in_len, akp_poseidon** out);
/* Parameters { generators: Vec<Vec<C>> }: generators_affine is [num_windows][window_size] points as in akp_te_params_create.
 * Reader: generators_affine == NULL only reports the shape; otherwise up to cap_points points are written. */
int32_t akp_serialize_te_parameters(const uint64_t* generators_affine, uint32_t window_size, uint32_t num_windows,
                                    int32_t compress, uint8_t* out, size_t out_cap, size_t* out_len);
int32_t akp_deserialize_te_parameters(const uint8_t* in, size_t in_len, int32_t compress, int32_t validate,
                                      uint64_t* generators_affine, size_t cap_points, uint32_t* window_size,
                                      uint32_t* num_windows);
/* Path { leaf_sibling_hash, auth_path (root side first), leaf_index }.  Reader: leaf_sibling_hash == NULL only reports
 * *depth and *leaf_index.  fe_per_digest (here and for MultiPath): 1 or 2 when LeafDigest and InnerDigest have the same width (every
 * configuration of the reference's tests), AKP_FE_PAIR(leaf_fe, inner_fe) when a Config gives them different ones (leaf_sibling_hash
 * / leaf_siblings_hashes are then leaf_fe Fr per digest, the authentication paths inner_fe). */
#define AKP_FE_PAIR(leaf_fe, inner_fe) (((uint32_t)(leaf_fe) << 8) | (uint32_t)(inner_fe))
int32_t akp_serialize_path(const uint64_t* leaf_sibling_hash, const uint64_t* auth_path, size_t depth, uint64_t leaf_index,
                           uint32_t fe_per_digest, int32_t compress, uint8_t* out, size_t out_cap, size_t* out_len);
int32_t akp_deserialize_path(const uint8_t* in, size_t in_len, uint32_t fe_per_digest, int32_t compress, int32_t validate,
                             uint64_t* leaf_sibling_hash, uint64_t* auth_path, size_t auth_cap, size_t* depth,
                             uint64_t* leaf_index);
/* MultiPath in the flat form of akp_merkle_multipath_encode: m paths, suffixes concatenated; suffix i holds suffix_lengths[i]
 * digests, or depth - prefix_lengths[i] when suffix_lengths is NULL.  Reader: leaf_siblings_hashes == NULL only reports *m
 * and *n_suffix_digests; the four vectors of the struct must have the same length (the flat form cannot hold anything else). */
int32_t akp_serialize_multipath(const uint64_t* leaf_siblings_hashes, const uint64_t* prefix_lengths,
                                const uint64_t* suffix_lengths, const uint64_t* suffixes, const uint64_t* leaf_indexes, size_t m,
                                size_t depth, uint32_t fe_per_digest, int32_t compress, uint8_t* out, size_t out_cap,
                                size_t* out_len);
int32_t akp_deserialize_multipath(const uint8_t* in, size_t in_len, uint32_t fe_per_digest, int32_t compress, int32_t validate,
                                  size_t* m, size_t* n_suffix_digests, uint64_t* leaf_siblings_hashes, uint64_t* prefix_lengths,
                                  uint64_t* suffix_lengths, uint64_t* suffixes, uint64_t* leaf_indexes, size_t m_cap,
                                  size_t suffix_cap);

/* ---- several GPUs from one process (merkle_tree/mod.rs:411-523 sharded by leaf range) -------- */
/* akp_ctx_create for n_dev devices (a power of two, distinct ids) + ncclCommInitAll over them.  RCCL is loaded with
 * dlopen on first use (librccl.so.1; AKP_RCCL_LIB overrides): AKP_ERR_RCCL when it is missing or fails. */
int32_t akp_multi_create(const int32_t* device_ids, int32_t n_dev, akp_multi** out);
void akp_multi_destroy(akp_multi* m);
int32_t akp_multi_size(const akp_multi* m);
/* context of device slot i (owned by m): create that device's parameter handles on it.  Handles may outlive akp_multi_destroy:
 * their compute calls then fail with AKP_ERR_BAD_PARAMS and destroying them stays valid (the context struct goes with its
 * last handle) */
akp_ctx* akp_multi_ctx(akp_multi* m, int32_t i);
/* phase breakdown of the last akp_merkle_build_sharded_* call on m, milliseconds, maximum over the devices (ms_out[5]):
 * [0] leaf copy-in + sub-tree build (host wall clock of the slowest device thread), [1] the ncclAllGather of the sub-roots,
 * [2] the top G - 1 nodes, [3] copy-out of inner nodes / root (device time between events on each device's stream),
 * [4] the whole call (host wall clock) */
int32_t akp_multi_last_phases(const akp_multi* m, double* ms_out);
/* MerkleTree::new over all devices of m: device r hashes leaves [r n/G, (r+1) n/G) into its own sub-tree, ONE
 * ncclAllGather moves the G sub-roots (xGMI), every device computes the top G-1 nodes.  leaf_params[r] and
 * two_to_one_params[r] are handles created on akp_multi_ctx(m, r).  Host buffers and outputs exactly as
 * akp_merkle_build_poseidon / akp_merkle_build_te (global heap order; outputs may be NULL). */
int32_t akp_merkle_build_sharded_poseidon(akp_multi* m, akp_poseidon* const* leaf_params, akp_poseidon* const* two_to_one_params,
                                          const uint64_t* leaves, size_t n_leaves, size_t leaf_len, uint64_t* leaf_nodes,
                                          uint64_t* non_leaf_nodes, uint64_t* root_out);
int32_t akp_merkle_build_sharded_te(akp_multi* m, akp_te_params* const* leaf_params, akp_te_params* const* two_to_one_params,
                                    const uint8_t* leaves, size_t n_leaves, size_t leaf_len, uint64_t* leaf_nodes,
                                    uint64_t* non_leaf_nodes, uint64_t* root_out);


/* ---- the tree sharded over several GPUs, RESIDENT (MerkleTree<P> as an object: merkle_tree/mod.rs:383-396, 526-579, 629-702) --- */
/* Device r of G keeps the sub-tree over leaves [r n/G, (r+1) n/G) in its own HBM (an ordinary akp_merkle_tree on akp_multi_ctx(m, r));
 * the top of the tree -- heap nodes 0 .. 2G-2: the G-1 nodes above the sub-roots and the sub-roots -- is replicated on every device
 * and on the host, refreshed after the build and after every update by ONE all-gather of the sub-roots + the top levels.  No node
 * array ever crosses a link or PCIe unless akp_multi_tree_export asks for it.  Parameter handle arrays as for
 * akp_merkle_build_sharded_*; the tree pins them.  akp_multi_last_phases reports the phases of the last build. */
/* MerkleTree::new (:411-422) from host leaves (global order; device r's range is copied in by its own host thread) */
int32_t akp_multi_tree_build_poseidon(akp_multi* m, akp_poseidon* const* leaf_params, akp_poseidon* const* two_to_one_params,
                                      const uint64_t* leaves, size_t n_leaves, size_t leaf_len, akp_multi_tree** out);
int32_t akp_multi_tree_build_te(akp_multi* m, akp_te_params* const* leaf_params, akp_te_params* const* two_to_one_params,
                                const uint8_t* leaves, size_t n_leaves, size_t leaf_len, akp_multi_tree** out);
/* leaves of different lengths (akp_merkle_tree_build_*_ragged per device): offsets has n_leaves + 1 entries into the one host array */
int32_t akp_multi_tree_build_poseidon_ragged(akp_multi* m, akp_poseidon* const* leaf_params, akp_poseidon* const* two_to_one_params,
                                             const uint64_t* leaves, const uint64_t* offsets, size_t n_leaves, akp_multi_tree** out);
int32_t akp_multi_tree_build_te_ragged(akp_multi* m, akp_te_params* const* leaf_params, akp_te_params* const* two_to_one_params,
                                       const uint8_t* leaves, const uint64_t* offsets, size_t n_leaves, akp_multi_tree** out);
/* the same from leaves that are already resident: d_leaves[r] = device r's n_leaves / G leaves, in ITS memory */
int32_t akp_multi_tree_build_poseidon_dev(akp_multi* m, akp_poseidon* const* leaf_params, akp_poseidon* const* two_to_one_params,
                                          const uint64_t* const* d_leaves, size_t n_leaves, size_t leaf_len, akp_multi_tree** out);
int32_t akp_multi_tree_build_te_dev(akp_multi* m, akp_te_params* const* leaf_params, akp_te_params* const* two_to_one_params,
                                    const uint8_t* const* d_leaves, size_t n_leaves, size_t leaf_len, akp_multi_tree** out);
/* The tree keeps its akp_multi alive: akp_multi_destroy(m) with live trees is deferred to the last akp_multi_tree_destroy. */
void akp_multi_tree_destroy(akp_multi_tree* t);
int32_t akp_multi_tree_info(const akp_multi_tree* t, size_t* n_leaves, uint32_t* fe_per_digest, size_t* height, int32_t* n_dev);
/* MerkleTree::root (:526-528), from the replicated top (no device access) */
int32_t akp_multi_tree_root(akp_multi_tree* t, uint64_t* root_out);
/* the sub-tree handle of device slot r (owned by t): its device pointers (akp_merkle_tree_device_ptrs) for local `_dev` work */
akp_merkle_tree* akp_multi_tree_shard(akp_multi_tree* t, int32_t r);
/* MerkleTree::generate_proof (:572-579) for m GLOBAL leaf indices: each index is served by the device that owns it; outputs as
 * akp_merkle_tree_gather_paths (auth_paths [m][log2(n) - 1], root side first: log2 G top siblings, then the local path) */
int32_t akp_multi_tree_gather_paths(akp_multi_tree* t, const uint64_t* leaf_indices, size_t m, uint64_t* leaf_sibling_hashes,
                                    uint64_t* auth_paths);
/* MerkleTree::update (:692-702), batched as akp_merkle_tree_update_batch: per-shard updates, then the exchange.  If a device
 * fails after others have updated, the replicated top is refreshed all the same (root and proofs describe the shards as they are)
 * and the error says so; if that refresh fails too the tree is marked unusable and every later call returns AKP_ERR_BAD_PARAMS. */
int32_t akp_multi_tree_update_batch(akp_multi_tree* t, const uint64_t* leaf_indices, const void* new_leaves, size_t m,
                                    size_t leaf_len);
/* MerkleTree::check_update (:707-725) on the sharded tree: *ok = 1 and the update is kept iff the new root equals asserted_new_root */
int32_t akp_multi_tree_check_update(akp_multi_tree* t, uint64_t leaf_index, const void* new_leaf, size_t leaf_len,
                                    const uint64_t* asserted_new_root, int32_t* ok);
/* MerkleTree::new_with_leaf_digest (:424-523) / blank (:400-408) over the devices: n_leaves digests in global order (host) */
int32_t akp_multi_tree_from_digests_poseidon(akp_multi* m, akp_poseidon* const* leaf_params, akp_poseidon* const* two_to_one_params,
                                             const uint64_t* leaf_digests, size_t n_leaves, akp_multi_tree** out);
int32_t akp_multi_tree_from_digests_te(akp_multi* m, akp_te_params* const* leaf_params, akp_te_params* const* two_to_one_params,
                                       const uint64_t* leaf_digests, size_t n_leaves, akp_multi_tree** out);
/* the reference's two vectors in GLOBAL heap order (checkpoints, tests); either pointer may be NULL */
int32_t akp_multi_tree_export(akp_multi_tree* t, uint64_t* leaf_nodes, uint64_t* non_leaf_nodes);

#ifdef __cplusplus
}
#endif
#endif /* AKP_H */
