"""Leaf-range sharding of MerkleTree::new across the GPUs of one node.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI, "gloo" in CPU tests).
Rank r of G owns leaves [r*n/G, (r+1)*n/G) and builds the height-log2(n/G)+1 sub-tree locally; its
sub-root is global heap node (G-1)+r.  The only exchange is ONE all-gather of the G sub-roots
(32 B each for Poseidon / Bowe-Hopwood digests, 64 B for Pedersen points -- latency-bound, link
bandwidth irrelevant); every rank then computes the top G-1 nodes redundantly.  There is no
data-path collective besides that, so the permutation / CRH batches shard with none at all.

The hashing backend is injected so the control flow can be exercised on CPU (gloo) in tests with a
test double; the product backend is `GpuPoseidonBackend` below (device pointers, no host copies).
"""
import numpy as np


def shard_range(n_leaves: int, rank: int, world: int):
    assert n_leaves % world == 0 and world & (world - 1) == 0, "world size must be a power of two dividing n"
    per = n_leaves // world
    assert per >= 2, "each rank needs at least two leaves (sub-tree of height >= 2)"
    return rank * per, (rank + 1) * per


def global_node_slices(n_leaves: int, rank: int, world: int):
    """[(level, global_start, count, local_start)] mapping this rank's local non_leaf heap array onto
    the global heap array: at global level l >= log2(world) the rank owns the contiguous slice
    [2^l - 1 + r*2^l/G, 2^l - 1 + (r+1)*2^l/G)."""
    out = []
    g = world.bit_length() - 1
    levels = (n_leaves.bit_length() - 1)  # non-leaf levels 0 .. log2(n)-1
    for l in range(g, levels):
        cnt = (1 << l) // world
        local_level = l - g
        out.append((l, (1 << l) - 1 + rank * cnt, cnt, (1 << local_level) - 1))
    return out


def combine_top(two_to_one_compress, sub_roots):
    """top log2(G) levels from the G gathered sub-roots; returns heap-ordered array of the top G-1
    nodes (root first).  `two_to_one_compress(left[m], right[m]) -> [m]`."""
    level = np.asarray(sub_roots)
    levels = []
    while level.shape[0] > 1:
        level = np.asarray(two_to_one_compress(level[0::2], level[1::2]))
        levels.append(level)
    if not levels:
        return level[:0]
    return np.concatenate(list(reversed(levels)), axis=0)


def build_sharded(backend, local_leaves, n_leaves_global: int, dist=None):
    """Sharded MerkleTree::new.  `backend` provides
         build_subtree(local_leaves) -> (leaf_nodes, non_leaf_nodes, sub_root as numpy [digest...])
         two_to_one_compress(left, right) -> numpy
       `dist` is torch.distributed (initialised) or None for a single process.
       Returns dict(root, top_nodes (G-1 heap-ordered), leaf_nodes, non_leaf_nodes (local))."""
    if dist is None or not dist.is_initialized():
        leaf_nodes, non_leaf, root = backend.build_subtree(local_leaves)
        return {"root": root, "top_nodes": np.asarray(root)[None][:0], "leaf_nodes": leaf_nodes, "non_leaf_nodes": non_leaf}
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    shard_range(n_leaves_global, rank, world)  # validates the partition
    if hasattr(backend, "combine_top_tensor"):
        # tensor-resident flow (the GPU backends): the sub-root is gathered straight from the node array and the top
        # G-1 nodes are one more inner-level build over the gathered digests -- no host copy before the root
        leaf_nodes, non_leaf = backend.build_subtree_tensors(local_leaves)
        subs = torch.empty((world,) + tuple(non_leaf.shape[1:]), dtype=non_leaf.dtype, device=non_leaf.device)
        dist.all_gather_into_tensor(subs, non_leaf[0:1].contiguous())  # the one collective of the tree build
        top_t = backend.combine_top_tensor(subs)
        top = top_t.cpu().numpy().view(np.uint64)
        root = top[0] if len(top) else subs[0].cpu().numpy().view(np.uint64)
        return {"root": root, "top_nodes": top, "leaf_nodes": leaf_nodes, "non_leaf_nodes": non_leaf}
    leaf_nodes, non_leaf, sub_root = backend.build_subtree(local_leaves)
    sub_root = np.ascontiguousarray(sub_root, dtype=np.uint64)
    dev = backend.comm_device()
    mine = torch.from_numpy(sub_root.view(np.int64).reshape(-1)).to(dev)
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)  # the one collective of the tree build
    subs = np.stack([g.cpu().numpy().view(np.uint64).reshape(sub_root.shape) for g in gathered])
    top = combine_top(backend.two_to_one_compress, subs)
    root = top[0] if len(top) else subs[0]
    return {"root": root, "top_nodes": top, "leaf_nodes": leaf_nodes, "non_leaf_nodes": non_leaf}


class GpuPoseidonBackend:
    """Poseidon leaf CRH + Poseidon two-to-one on this rank's GPU; leaves/nodes stay in HBM as torch
    int64 tensors viewed as Fr wire format (4 x u64 per element)."""

    def __init__(self, leaf_params, two_params, leaf_len=1, device=None):
        import torch
        from ._lib import default_context
        self.torch = torch
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.ctx = default_context(self.device.index or 0)
        self.leaf_h = leaf_params.handle(self.ctx)
        self.two_h = two_params.handle(self.ctx)
        self.two_params = two_params
        self.leaf_len = leaf_len

    def comm_device(self):
        return self.device

    def build_subtree_tensors(self, d_leaves):
        """d_leaves: int64 cuda tensor [n_local, leaf_len, 4]; or, for leaves of DIFFERENT lengths, a tuple (flat int64 tensor [total, 4],
        offsets int64 tensor [n_local + 1] in elements): akp_merkle_build_poseidon_ragged_dev."""
        from ._lib import lib, check
        torch = self.torch
        if isinstance(d_leaves, tuple):
            flat, offs = d_leaves[0], d_leaves[1]
            n = offs.shape[0] - 1
            leaf_nodes = torch.empty((n, 4), dtype=torch.int64, device=self.device)
            non_leaf = torch.empty((n - 1, 4), dtype=torch.int64, device=self.device)
            stream = torch.cuda.current_stream(self.device).cuda_stream
            check(lib.akp_merkle_build_poseidon_ragged_dev(self.leaf_h.h, self.two_h.h, flat.data_ptr(), offs.data_ptr(), n,
                                                           leaf_nodes.data_ptr(), non_leaf.data_ptr(), stream))
            return leaf_nodes, non_leaf
        n = d_leaves.shape[0]
        leaf_nodes = torch.empty((n, 4), dtype=torch.int64, device=self.device)
        non_leaf = torch.empty((n - 1, 4), dtype=torch.int64, device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        check(lib.akp_merkle_build_poseidon_dev(self.leaf_h.h, self.two_h.h, d_leaves.data_ptr(), n, self.leaf_len,
                                                leaf_nodes.data_ptr(), non_leaf.data_ptr(), stream))
        return leaf_nodes, non_leaf

    def build_subtree(self, d_leaves):
        leaf_nodes, non_leaf = self.build_subtree_tensors(d_leaves)
        return leaf_nodes, non_leaf, non_leaf[0].cpu().numpy().view(np.uint64)

    def combine_top_tensor(self, subs):
        """subs: [G, 4] gathered sub-roots on this device -> the top G-1 nodes, heap order (root first): the inner
        levels of a tree whose leaf digests are the sub-roots (merkle_tree/mod.rs:441-515)."""
        from ._lib import lib, check
        torch = self.torch
        g = subs.shape[0]
        top = torch.empty((g - 1, 4), dtype=torch.int64, device=self.device)
        if g > 1:
            stream = torch.cuda.current_stream(self.device).cuda_stream
            check(lib.akp_merkle_inner_poseidon_dev(self.two_h.h, subs.data_ptr(), g, top.data_ptr(), stream))
        return top

    def two_to_one_compress(self, left, right):
        from .crh.poseidon import TwoToOneCRH
        return TwoToOneCRH.compress_batch(self.two_params, np.ascontiguousarray(left), np.ascontiguousarray(right))


class GpuTeBackend:
    """Pedersen / Bowe-Hopwood leaf + two-to-one hashes with ByteDigestConverter on this rank's GPU (BASELINE
    config 5: Bowe-Hopwood tree over byte leaves).  Leaves are a uint8 cuda tensor [n_local, leaf_len]; digests
    are int64 tensors viewed as Fr wire format (1 Fr per node for Bowe-Hopwood, 2 for Pedersen)."""

    def __init__(self, leaf_params, two_params, device=None):
        import torch
        from ._lib import default_context
        self.torch = torch
        self.device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.ctx = default_context(self.device.index or 0)
        self.leaf_h = leaf_params.handle(self.ctx)
        self.two_h = two_params.handle(self.ctx)
        self.two_params = two_params
        if self.leaf_h.fe_per_digest != self.two_h.fe_per_digest:
            raise TypeError("leaf and two-to-one parameters produce digests of different widths")
        self.fe = self.two_h.fe_per_digest  # what libakp writes per node for this handle kind

    def comm_device(self):
        return self.device

    def build_subtree_tensors(self, d_leaves):
        """d_leaves: uint8 cuda tensor [n_local, leaf_len]; or, for leaves of DIFFERENT lengths, a tuple (flat uint8 tensor, offsets
        int64 tensor [n_local + 1] into it, max_len): akp_merkle_build_te_ragged_dev"""
        from ._lib import lib, check
        torch = self.torch
        if isinstance(d_leaves, tuple):
            flat, offs, max_len = d_leaves
            n = offs.shape[0] - 1
            leaf_nodes = torch.empty((n, self.fe * 4), dtype=torch.int64, device=self.device)
            non_leaf = torch.empty((n - 1, self.fe * 4), dtype=torch.int64, device=self.device)
            stream = torch.cuda.current_stream(self.device).cuda_stream
            check(lib.akp_merkle_build_te_ragged_dev(self.leaf_h.h, self.two_h.h, flat.data_ptr(), offs.data_ptr(), n, int(max_len),
                                                     leaf_nodes.data_ptr(), non_leaf.data_ptr(), stream))
            return leaf_nodes, non_leaf
        n, L = d_leaves.shape[0], d_leaves.shape[1]
        leaf_nodes = torch.empty((n, self.fe * 4), dtype=torch.int64, device=self.device)
        non_leaf = torch.empty((n - 1, self.fe * 4), dtype=torch.int64, device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        check(lib.akp_merkle_build_te_dev(self.leaf_h.h, self.two_h.h, d_leaves.data_ptr(), n, L, leaf_nodes.data_ptr(),
                                          non_leaf.data_ptr(), stream))
        return leaf_nodes, non_leaf

    def build_subtree(self, d_leaves):
        leaf_nodes, non_leaf = self.build_subtree_tensors(d_leaves)
        return leaf_nodes, non_leaf, non_leaf[0].cpu().numpy().view(np.uint64)

    def combine_top_tensor(self, subs):
        from ._lib import lib, check
        torch = self.torch
        g = subs.shape[0]
        top = torch.empty((g - 1, self.fe * 4), dtype=torch.int64, device=self.device)
        if g > 1:
            stream = torch.cuda.current_stream(self.device).cuda_stream
            check(lib.akp_merkle_inner_te_dev(self.two_h.h, subs.data_ptr(), g, top.data_ptr(), stream))
        return top

    def two_to_one_compress(self, left, right):
        from .crh import pedersen, bowe_hopwood, injective_map
        from ._lib import TE_PEDERSEN, TE_BOWE_HOPWOOD
        cls = {TE_PEDERSEN: pedersen.TwoToOneCRH, TE_BOWE_HOPWOOD: bowe_hopwood.TwoToOneCRH}.get(self.two_h.kind, injective_map.PedersenTwoToOneCRHCompressor)
        out = cls.compress_batch(self.two_params, np.ascontiguousarray(left), np.ascontiguousarray(right))
        return np.ascontiguousarray(out).reshape(len(out), -1)
