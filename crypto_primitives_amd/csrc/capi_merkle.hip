// capi_merkle.hip -- part of libakp.so (implementation of include/akp.h): Merkle trees -- builds, proofs, verification; capi_tree.inc
// (HBM-resident trees), capi_multi.inc (several GPUs), capi_multi_tree.inc (the tree sharded over several GPUs, resident)
// Product code.  Never includes, links or calls anything under oracle/; there is no CPU fallback for any compute entry
// point (a missing device is AKP_ERR_HIP).
#include "capi_internal.hpp"

// ------------------------------------------------------------------------------------------
// Merkle tree (merkle_tree/mod.rs:411-523): one launch per level, bottom-up; level l of the
// heap-ordered non_leaf array starts at 2^l - 1.


// MerkleTree::new from host leaves: the copy-in of leaf chunk i + 1 (copy stream) overlaps the leaf hashing of chunk i
// (context stream); the inner levels follow on the context stream while the copy stream already returns the leaf digests;
// the inner nodes go back last.  hash_leaves(d_chunk, first_leaf, count, stream), inner(stream).
template <class HashLeaves, class Inner>
static int32_t host_tree_build(akp_ctx* c, const void* leaves, size_t n, size_t leaf_bytes, size_t dig_bytes, void* d_leaves, void* d_ln,
        void* d_nl,
                               void* h_ln, void* h_nl, void* h_root, HashLeaves hash_leaves, Inner inner) {
    constexpr size_t chunk_items = (size_t)1 << 20;  // leaves per copy-in chunk (round 2: 2^18 .. 2^22 measured alike; no knob)
    // the copy stream is a high-priority stream (pipe[5], round 6): the staged copies of pageable leaves are small kernels on a queue of
    // their own, and a queue that shares a pipe of the command processor with the hash kernels' queue is served late while a grid
    // has workgroups left to place (profiles/r06_s41); neutral where the queues do not meet (profiles/r06_s52)
    HIP_TRY(ctx_copy_streams(c));
    for (int i = 0; i < 8; ++i)
        if (!c->chunk_event[i]) HIP_TRY(hipEventCreateWithFlags(&c->chunk_event[i], hipEventDisableTiming));
    hipStream_t comp = c->stream, copy = c->pipe[5];
    // the scratch regions were acquired for the context stream: let the copy stream start behind whatever used them last
    HIP_TRY(hipEventRecord(c->chunk_event[7], comp));
    HIP_TRY(hipStreamWaitEvent(copy, c->chunk_event[7], 0));
    size_t ci = 0;
    for (size_t done = 0; done < n; done += chunk_items, ++ci) {
        const size_t cnt = std::min(chunk_items, n - done);
        if (leaf_bytes) {
            HIP_TRY(hipMemcpyAsync((char*)d_leaves + done * leaf_bytes, (const char*)leaves + done * leaf_bytes, cnt * leaf_bytes,
                    hipMemcpyHostToDevice, copy));
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

extern "C" int32_t akp_merkle_inner_poseidon_dev(akp_poseidon* two, const uint64_t* d_leaf_nodes, size_t n, uint64_t* d_non_leaf,
        void* stream) {
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
extern "C" int32_t akp_merkle_build_poseidon_dev(akp_poseidon* leafp, akp_poseidon* two, const uint64_t* d_leaves, size_t n,
        size_t leaf_len,
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

// the same for the Poseidon field tree (leaf i = elements [d_offsets[i], d_offsets[i+1]) of d_leaves; t = 3 leaf parameters)
extern "C" int32_t akp_merkle_build_poseidon_ragged_dev(akp_poseidon* leafp, akp_poseidon* two, const uint64_t* d_leaves, const uint64_t* d_offsets,
        size_t n, uint64_t* d_leaf_nodes, uint64_t* d_non_leaf, void* stream) {
    NEED_DEV(leafp, "akp_merkle_build_poseidon_ragged_dev");
    NEED_DEV(two, "akp_merkle_build_poseidon_ragged_dev");
    if (leafp->ctx != two->ctx) return fail(AKP_ERR_BAD_PARAMS, "leaf and two-to-one parameters belong to different contexts");
    if (!pow2_gt1(n)) return fail(AKP_ERR_NOT_POW2, "leaves.len() should be power of two and greater than one (got %zu)", n);
    if (!d_offsets || !d_leaf_nodes || !d_non_leaf) return fail(AKP_ERR_BAD_PARAMS, "NULL buffer");
    hipStream_t s = pick_stream(leafp->ctx, stream);
    if (int32_t rc = poseidon_crh_ragged_dev(leafp, (const Fr*)d_leaves, d_offsets, n, (Fr*)d_leaf_nodes, s)) return rc;
    return akp_merkle_inner_poseidon_dev(two, d_leaf_nodes, n, d_non_leaf, (void*)s);
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
    if (int32_t rc = te_tree_prepare(leafp, two, s)) return rc;
    if (int32_t rc = te_crh_dev(leafp, d_leaves, n, leaf_len, (Fr*)d_leaf_nodes, s)) return rc;
    return akp_merkle_inner_te_dev(two, d_leaf_nodes, n, d_non_leaf, (void*)s);
}
// MerkleTree::new over leaves of DIFFERENT lengths, everything in device memory (round 5; merkle_tree/mod.rs:411-422): leaf i = bytes
// [d_offsets[i], d_offsets[i+1]) of d_leaves; max_len bounds the longest leaf (the table is built for it).  Enqueue only.
extern "C" int32_t akp_merkle_build_te_ragged_dev(akp_te_params* leafp, akp_te_params* two, const uint8_t* d_leaves, const uint64_t* d_offsets, size_t n,
        size_t max_len, uint64_t* d_leaf_nodes, uint64_t* d_non_leaf, void* stream) {
    NEED_TE(leafp, "akp_merkle_build_te_ragged_dev");
    NEED_TE(two, "akp_merkle_build_te_ragged_dev");
    if (leafp->kind != two->kind || leafp->ctx != two->ctx) return fail(AKP_ERR_BAD_PARAMS, "leaf / two-to-one parameters mismatch");
    if (!pow2_gt1(n)) return fail(AKP_ERR_NOT_POW2, "leaves.len() should be power of two and greater than one (got %zu)", n);
    if (!d_offsets || !d_leaf_nodes || !d_non_leaf) return fail(AKP_ERR_BAD_PARAMS, "NULL buffer");
    hipStream_t s = pick_stream(leafp->ctx, stream);
    if (int32_t rc = te_tree_prepare(leafp, two, s)) return rc;
    if (int32_t rc = te_crh_ragged_dev(leafp, d_leaves, d_offsets, n, max_len, (Fr*)d_leaf_nodes, s)) return rc;
    return akp_merkle_inner_te_dev(two, d_leaf_nodes, n, d_non_leaf, (void*)s);
}
extern "C" int32_t akp_merkle_build_te(akp_te_params* leafp, akp_te_params* two, const uint8_t* leaves, size_t n, size_t leaf_len,
                                       uint64_t* leaf_nodes, uint64_t* non_leaf, uint64_t* root_out) {
    NEED_TE(leafp, "akp_merkle_build_te");
    NEED_TE(two, "akp_merkle_build_te");
    if (!pow2_gt1(n)) return fail(AKP_ERR_NOT_POW2, "leaves.len() should be power of two and greater than one (got %zu)", n);
    if (leaf_len * 8 > te_input_bits(leafp))
        return fail(AKP_ERR_BAD_LENGTH, "incorrect input length %zu for window params %ux%u (the reference panics)", leaf_len, leafp->W,
                leafp->N);
    if (!leaves && leaf_len) return fail(AKP_ERR_BAD_PARAMS, "leaves is NULL");
    akp_ctx* c = leafp->ctx;
    const size_t fe = te_fe_per_digest(two);
    void *dl = nullptr, *dln = nullptr, *dnl = nullptr;
    if (int32_t rc = ctx_scratch(c, SCR_A, n * leaf_len, &dl, c->stream)) return rc;
    if (int32_t rc = ctx_scratch(c, SCR_B, n * fe * sizeof(Fr), &dln, c->stream)) return rc;
    if (int32_t rc = ctx_scratch(c, SCR_C, (n - 1) * fe * sizeof(Fr), &dnl, c->stream)) return rc;
    if (leafp->kind != two->kind || leafp->ctx != two->ctx) return fail(AKP_ERR_BAD_PARAMS, "leaf / two-to-one parameters mismatch");
    if (int32_t rc = te_tree_prepare(leafp, two, c->stream)) return rc;
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
extern "C" int32_t akp_merkle_gather_paths_dev(akp_ctx* ctx, const uint64_t* d_leaf_nodes, const uint64_t* d_non_leaf, size_t n,
        uint32_t fe,
                                               const uint64_t* d_idx, size_t m, uint64_t* d_sib, uint64_t* d_auth, void* stream) {
    if (!ctx) return fail(AKP_ERR_HIP, "akp_merkle_gather_paths_dev: a device context is required");
    if (!pow2_gt1(n)) return fail(AKP_ERR_NOT_POW2, "tree must have a power-of-two number of leaves > 1");
    if (fe != 1 && fe != 2) return fail(AKP_ERR_BAD_PARAMS, "fe_per_digest must be 1 or 2");
    if (m == 0) return AKP_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t log2n = log2_exact(n), work = m * log2n;
    hipLaunchKernelGGL(merkle_gather_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
            (const Fr*)d_leaf_nodes,
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
// `walk(d_idx, d_sib, d_auth, d_root, d_ok, stream, &done)`: a one-launch form that may decline (done = false)
template <class HashLeaves, class TwoToOne, class Walk>
static int32_t verify_paths_common(akp_ctx* c, u32 fe, const uint64_t* root, size_t m, const uint64_t* idx, const uint64_t* sibs,
                                   const uint64_t* auth, size_t depth, uint8_t* ok_out, HashLeaves hash_leaves, TwoToOne two_to_one, Walk walk) {
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
    bool walked = false;
    if (int32_t rc = walk((const uint64_t*)d_idx, (const Fr*)d_sib, (const Fr*)d_auth, (const Fr*)d_root, d_ok, s, &walked)) return rc;
    if (walked) {
        HIP_TRY(hipMemcpyAsync(ok_out, d_ok, m, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        return AKP_OK;
    }
    if (int32_t rc = hash_leaves((Fr*)d_cur, s)) return rc;
    const unsigned grid = (unsigned)((m * fe + 255) / 256);
    for (size_t step = 0; step <= depth; ++step) {
        // step 0 pairs the leaf digest with leaf_sibling_hash; step s >= 1 uses auth_path[depth - s]
        const Fr* sib = step == 0 ? (const Fr*)d_sib : (const Fr*)d_auth + (depth - step) * fe;
        const size_t stride = step == 0 ? fe : depth * fe;
        hipLaunchKernelGGL(merkle_select_kernel, dim3(grid), dim3(256), 0, s, (const Fr*)d_cur, sib, stride, fe, (const uint64_t*)d_idx,
                (u32)step,
                           (Fr*)d_l, (Fr*)d_r, m);
        HIP_TRY(hipGetLastError());
        if (int32_t rc = two_to_one((const Fr*)d_l, (const Fr*)d_r, (Fr*)d_cur, s)) return rc;
    }
    hipLaunchKernelGGL(merkle_compare_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, (const Fr*)d_cur, (const Fr*)d_root, fe,
            d_ok, m);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(ok_out, d_ok, m, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return AKP_OK;
}
extern "C" int32_t akp_merkle_verify_paths_poseidon(akp_poseidon* leafp, akp_poseidon* two, const uint64_t* root, const uint64_t* leaves,
        size_t m,
        size_t leaf_len, const uint64_t* idx, const uint64_t* sibs, const uint64_t* auth, size_t depth,
                                                    uint8_t* ok_out) {
    NEED_DEV(leafp, "akp_merkle_verify_paths_poseidon");
    NEED_DEV(two, "akp_merkle_verify_paths_poseidon");
    if (leafp->ctx != two->ctx) return fail(AKP_ERR_BAD_PARAMS, "parameters belong to different contexts");
    if (m && !leaves && leaf_len) return fail(AKP_ERR_BAD_PARAMS, "leaves is NULL");
    akp_ctx* c = leafp->ctx;
    bool leaves_uploaded = false;  // the one-launch attempt below uploads the leaves into SCR_A; when it declines (small batch, shape), the level-by-level path reuses them
    return verify_paths_common(
        c, 1, root, m, idx, sibs, auth, depth, ok_out,
        [&](Fr* d_cur, hipStream_t s) -> int32_t {
            void* dl = nullptr;
            if (int32_t rc = ctx_scratch(c, SCR_A, m * leaf_len * sizeof(Fr), &dl, s)) return rc;
            if (leaf_len && !leaves_uploaded) HIP_TRY(hipMemcpyAsync(dl, leaves, m * leaf_len * sizeof(Fr), hipMemcpyHostToDevice, s));
            return launch_crh(leafp, (const Fr*)dl, nullptr, leaf_len, d_cur, m, s);
        },
        [&](const Fr* l, const Fr* r, Fr* out, hipStream_t s) -> int32_t { return launch_crh(two, l, r, 2, out, m, s); },
        [&](const uint64_t* d_idx, const Fr* d_sib, const Fr* d_auth, const Fr* d_root, uint8_t* d_ok, hipStream_t s, bool* done) -> int32_t {
            *done = false;
            if (leaf_len < 1 || leaf_len > 2) return AKP_OK;
            void* dl = nullptr;
            if (int32_t rc = ctx_scratch(c, SCR_A, m * leaf_len * sizeof(Fr), &dl, s)) return rc;
            HIP_TRY(hipMemcpyAsync(dl, leaves, m * leaf_len * sizeof(Fr), hipMemcpyHostToDevice, s));
            leaves_uploaded = true;
            return launch_verify_paths_t3(leafp, two, (const Fr*)dl, leaf_len, d_idx, d_sib, d_auth, depth, d_root, d_ok, m, s, done);
        });
}
// the same with every buffer already in device memory (proofs produced by akp_merkle_gather_paths_dev can be checked where they
// lie): enqueues on `stream`, no synchronisation.  d_ok_out: m bytes.  Shapes outside the one-launch kernel (launch_verify_paths_t3)
// run level by level on context scratch.
extern "C" int32_t akp_merkle_verify_paths_poseidon_dev(akp_poseidon* leafp, akp_poseidon* two, const uint64_t* d_root, const uint64_t* d_leaves,
        size_t m, size_t leaf_len, const uint64_t* d_idx, const uint64_t* d_sibs, const uint64_t* d_auth, size_t depth, uint8_t* d_ok_out,
        void* stream) {
    NEED_DEV(leafp, "akp_merkle_verify_paths_poseidon_dev");
    NEED_DEV(two, "akp_merkle_verify_paths_poseidon_dev");
    if (leafp->ctx != two->ctx) return fail(AKP_ERR_BAD_PARAMS, "parameters belong to different contexts");
    if (m == 0) return AKP_OK;
    if (!d_root || !d_idx || !d_sibs || !d_ok_out || (depth && !d_auth) || (leaf_len && !d_leaves)) return fail(AKP_ERR_BAD_PARAMS, "NULL buffer");
    akp_ctx* c = leafp->ctx;
    hipStream_t s = pick_stream(c, stream);
    bool walked = false;
    if (int32_t rc = launch_verify_paths_t3(leafp, two, (const Fr*)d_leaves, leaf_len, d_idx, (const Fr*)d_sibs, (const Fr*)d_auth, depth, (const Fr*)d_root,
            d_ok_out, m, s, &walked))
        return rc;
    if (walked) return AKP_OK;
    void *d_cur = nullptr, *d_l = nullptr, *d_r = nullptr;
    if (int32_t rc = ctx_scratch(c, SCR_G, m * sizeof(Fr), &d_cur, s)) return rc;
    if (int32_t rc = ctx_scratch(c, SCR_H, m * sizeof(Fr), &d_l, s)) return rc;
    if (int32_t rc = ctx_scratch(c, SCR_I, m * sizeof(Fr), &d_r, s)) return rc;
    if (int32_t rc = launch_crh(leafp, (const Fr*)d_leaves, nullptr, leaf_len, (Fr*)d_cur, m, s)) return rc;
    const unsigned grid = (unsigned)((m + 255) / 256);
    for (size_t step = 0; step <= depth; ++step) {
        const Fr* sib = step == 0 ? (const Fr*)d_sibs : (const Fr*)d_auth + (depth - step);
        hipLaunchKernelGGL(merkle_select_kernel, dim3(grid), dim3(256), 0, s, (const Fr*)d_cur, sib, step == 0 ? (size_t)1 : depth, 1u, d_idx, (u32)step,
                (Fr*)d_l, (Fr*)d_r, m);
        HIP_TRY(hipGetLastError());
        if (int32_t rc = launch_crh(two, (const Fr*)d_l, (const Fr*)d_r, 2, (Fr*)d_cur, m, s)) return rc;
    }
    hipLaunchKernelGGL(merkle_compare_kernel, dim3(grid), dim3(256), 0, s, (const Fr*)d_cur, (const Fr*)d_root, 1u, d_ok_out, m);
    HIP_TRY(hipGetLastError());
    return AKP_OK;
}
extern "C" int32_t akp_merkle_verify_paths_te(akp_te_params* leafp, akp_te_params* two, const uint64_t* root, const uint8_t* leaves,
        size_t m,
        size_t leaf_len, const uint64_t* idx, const uint64_t* sibs, const uint64_t* auth, size_t depth,
                                              uint8_t* ok_out) {
    NEED_TE(leafp, "akp_merkle_verify_paths_te");
    NEED_TE(two, "akp_merkle_verify_paths_te");
    if (leafp->kind != two->kind || leafp->ctx != two->ctx) return fail(AKP_ERR_BAD_PARAMS, "leaf / two-to-one parameters mismatch");
    if (leaf_len * 8 > te_input_bits(leafp))
        return fail(AKP_ERR_BAD_LENGTH, "incorrect input length %zu for window params %ux%u (the reference panics)", leaf_len, leafp->W,
                leafp->N);
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
        [&](const Fr* l, const Fr* r, Fr* out, hipStream_t s) -> int32_t { return te_compress_dev(two, l, r, m, out, s); },
        [](const uint64_t*, const Fr*, const Fr*, const Fr*, uint8_t*, hipStream_t, bool* done) -> int32_t {
            *done = false;  // curve hashes: level by level (a lane-walked path would gather table lines for 21 dependent hashes)
            return AKP_OK;
        });
}

#include "capi_tree.inc"
#include "capi_multi.inc"
#include "capi_multi_tree.inc"

