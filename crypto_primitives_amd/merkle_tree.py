"""merkle_tree::{Config, MerkleTree, Path, MultiPath} (merkle_tree/mod.rs) with GPU level-wide hashing.

The build (`MerkleTree.new`, :411-523) is the hot path: one leaf-hash launch and one two-to-one
launch per level on the GPU, results landing in the reference's heap layout (`leaf_nodes[n]`,
`non_leaf_nodes[n-1]`, root at 0).  Proof generation (:547-625) is index arithmetic over those
arrays (host); verification (:172-212, :262-331) re-hashes through the same GPU batch entry points.

Config objects bundle what the reference's `Config` trait (:83-122) fixes at type level.
"""
import numpy as np

from ._lib import lib, check, NotPowerOfTwo
from .crh import poseidon as _pos, pedersen as _ped, bowe_hopwood as _bh, injective_map as _inj


# ---- Config implementations ---------------------------------------------------------------------
class PoseidonFieldConfig:
    """Leaf = [Fr], LeafDigest = InnerDigest = Fr, IdentityDigestConverter, poseidon CRH / TwoToOneCRH
    (the FieldMTConfig of merkle_tree/tests/mod.rs:198-206)."""
    LeafHash = _pos.CRH
    TwoToOneHash = _pos.TwoToOneCRH
    digest_shape = (4,)

    @staticmethod
    def build(leaf_params, two_params, leaves):
        x = np.ascontiguousarray(leaves, dtype=np.uint64)
        n = x.shape[0]
        k = x.size // (4 * n) if n else 0
        leaf_nodes = np.empty((n, 4), dtype=np.uint64)
        non_leaf = np.empty((max(n - 1, 0), 4), dtype=np.uint64)
        check(lib.akp_merkle_build_poseidon(leaf_params.handle().h, two_params.handle().h, x.ctypes.data, n, k,
                                            leaf_nodes.ctypes.data, non_leaf.ctypes.data, None))
        return leaf_nodes, non_leaf

    @staticmethod
    def build_inner(two_params, leaf_digests):
        ln = np.ascontiguousarray(leaf_digests, dtype=np.uint64).reshape(-1, 4)
        non_leaf = np.empty((max(len(ln) - 1, 0), 4), dtype=np.uint64)
        check(lib.akp_merkle_inner_poseidon(two_params.handle().h, ln.ctypes.data, len(ln), non_leaf.ctypes.data))
        return ln, non_leaf

    default_leaf_digest = staticmethod(lambda: np.zeros(4, dtype=np.uint64))  # Fr::default() == 0

    @staticmethod
    def hash_leaves(leaf_params, leaves):
        if _pos.ragged_fr(leaves) is not None:
            return _pos.CRH.evaluate_batch(leaf_params, leaves)
        x = np.ascontiguousarray(leaves, dtype=np.uint64)
        n = x.shape[0]
        return _pos.CRH.evaluate_batch(leaf_params, x.reshape(n, -1, 4))

    @staticmethod
    def two_to_one_evaluate(two_params, left_digests, right_digests):  # IdentityDigestConverter (:53-63)
        return _pos.TwoToOneCRH.evaluate_batch(two_params, left_digests, right_digests)

    @staticmethod
    def two_to_one_compress(two_params, left, right):
        return _pos.TwoToOneCRH.compress_batch(two_params, left, right)

    @staticmethod
    def verify_abi(leaf_params, two_params, root, leaves, idx, sibs, auth, depth):
        x = np.ascontiguousarray(leaves, dtype=np.uint64)
        m = len(idx)
        k = x.size // (4 * m)
        ok = np.zeros(m, dtype=np.uint8)
        check(lib.akp_merkle_verify_paths_poseidon(leaf_params.handle().h, two_params.handle().h, root.ctypes.data, x.ctypes.data, m, k,
                                                   idx.ctypes.data, sibs.ctypes.data, auth.ctypes.data if depth else None, depth, ok.ctypes.data))
        return ok


def _leaf_handle(config, params, ctx=None):
    """handle of the LEAF hash parameters as this config's LeafHash class uses them (the class fixes the kernel kind and with
    it the Fr per digest libakp writes -- crh/pedersen.py te_handle)"""
    return params.handle(ctx) if config is PoseidonFieldConfig else _ped.te_handle(params, config.LeafHash, ctx)


def _two_handle(config, params, ctx=None):
    return params.handle(ctx) if config is PoseidonFieldConfig else _ped.te_handle(params, config.TwoToOneHash._crh, ctx)


class _ByteConfig:
    """Leaf = [u8], ByteDigestConverter (:67-78): digests are serialised uncompressed before the
    two-to-one hash (config shape merkle_tree/tests/mod.rs:24-33)."""

    @classmethod
    def build(cls, leaf_params, two_params, leaves):
        m, n, L = _ped._as_msgs(leaves)
        fe = cls.LeafHash._FE
        leaf_nodes = np.empty((n, fe, 4), dtype=np.uint64)
        non_leaf = np.empty((max(n - 1, 0), fe, 4), dtype=np.uint64)
        check(lib.akp_merkle_build_te(_leaf_handle(cls, leaf_params).h, _two_handle(cls, two_params).h, m.ctypes.data if m.size else None,
                                      n, L, leaf_nodes.ctypes.data, non_leaf.ctypes.data, None))
        shp = cls.digest_shape
        return leaf_nodes.reshape((n,) + shp), non_leaf.reshape((max(n - 1, 0),) + shp)

    @classmethod
    def build_inner(cls, two_params, leaf_digests):
        ln = np.ascontiguousarray(leaf_digests, dtype=np.uint64).reshape((-1,) + cls.digest_shape)
        non_leaf = np.empty((max(len(ln) - 1, 0),) + cls.digest_shape, dtype=np.uint64)
        check(lib.akp_merkle_inner_te(_two_handle(cls, two_params).h, ln.ctypes.data, len(ln), non_leaf.ctypes.data))
        return ln, non_leaf

    @classmethod
    def default_leaf_digest(cls):
        # Default of an affine TE point is the identity (0, 1); of an Fq digest it is 0
        from . import field
        return field.fr([0, 1]).reshape(2, 4) if cls.digest_shape == (2, 4) else np.zeros(4, dtype=np.uint64)

    @classmethod
    def hash_leaves(cls, leaf_params, leaves):
        return cls.LeafHash.evaluate_batch(leaf_params, leaves)

    @classmethod
    def two_to_one_evaluate(cls, two_params, left_digests, right_digests):
        # convert() = uncompressed bytes, then TwoToOneHash::evaluate == compress on the digests
        return cls.TwoToOneHash.compress_batch(two_params, left_digests, right_digests)

    @classmethod
    def two_to_one_compress(cls, two_params, left, right):
        return cls.TwoToOneHash.compress_batch(two_params, left, right)

    @classmethod
    def verify_abi(cls, leaf_params, two_params, root, leaves, idx, sibs, auth, depth):
        mm, m, L = _ped._as_msgs(leaves)
        ok = np.zeros(m, dtype=np.uint8)
        check(lib.akp_merkle_verify_paths_te(_leaf_handle(cls, leaf_params).h, _two_handle(cls, two_params).h, root.ctypes.data, mm.ctypes.data if mm.size else None,
                                             m, L, idx.ctypes.data, sibs.ctypes.data, auth.ctypes.data if depth else None, depth, ok.ctypes.data))
        return ok


class PedersenByteConfig(_ByteConfig):
    LeafHash = _ped.CRH
    TwoToOneHash = _ped.TwoToOneCRH
    digest_shape = (2, 4)


class BoweHopwoodByteConfig(_ByteConfig):
    LeafHash = _bh.CRH
    TwoToOneHash = _bh.TwoToOneCRH
    digest_shape = (4,)


class PedersenXByteConfig(_ByteConfig):
    """JubJubMerkleTreeParams of merkle_tree/tests/constraints.rs: LeafHash = PedersenCRHCompressor<JubJub, TECompressor, W>,
    TwoToOneHash = PedersenTwoToOneCRHCompressor<...>, digests = Fq, ByteDigestConverter"""
    LeafHash = _inj.PedersenCRHCompressor
    TwoToOneHash = _inj.PedersenTwoToOneCRHCompressor
    digest_shape = (4,)


def _ragged_leaves(config, leaves):
    """(flat, offsets) when `leaves` is a list whose items differ in length, else None (crh.poseidon.ragged_fr / crh.pedersen.ragged_bytes)"""
    return _pos.ragged_fr(leaves) if config is PoseidonFieldConfig else _ped.ragged_bytes(leaves)


# ---- index helpers (:730-786) ---------------------------------------------------------------------
def tree_height(num_leaves):
    return 1 if num_leaves == 1 else (num_leaves.bit_length() - 1) + 1


def left_child(i):
    return 2 * i + 1


def right_child(i):
    return 2 * i + 2


def parent(i):
    return (i - 1) >> 1 if i > 0 else None


def sibling(i):
    if i == 0:
        return None
    return i + 1 if i % 2 == 1 else i - 1


def is_left_child(i):
    return i % 2 == 1


def convert_index_to_last_level(index, height):
    return index + (1 << (height - 1)) - 1


def _eq(a, b):
    return np.array_equal(np.asarray(a), np.asarray(b))


class Path:
    """merkle_tree::Path (:146-213)."""

    def __init__(self, config, leaf_sibling_hash, auth_path, leaf_index):
        self.config = config
        self.leaf_sibling_hash = leaf_sibling_hash
        self.auth_path = auth_path  # root-side first, excludes the root
        self.leaf_index = leaf_index

    def verify(self, leaf_hash_params, two_to_one_params, root_hash, leaf) -> bool:
        """:172-212."""
        return verify_paths(self.config, leaf_hash_params, two_to_one_params, root_hash, [self], [leaf])[0]


def verify_paths(config, leaf_params, two_params, root_hash, paths, leaves):
    """Batched Path::verify (:172-212): one ABI call; all paths advance one level per GPU launch."""
    n = len(paths)
    if n == 0:
        return []
    depth = len(paths[0].auth_path)
    assert all(len(p.auth_path) == depth for p in paths)
    if isinstance(leaves, np.ndarray):
        leaves = [leaves[i] for i in range(n)]
    if _ragged_leaves(config, list(leaves)) is not None:  # leaves of different lengths: one ABI call per distinct length
        lens = [len(x) if isinstance(x, (bytes, bytearray)) else len(np.asarray(x).reshape(-1, 4) if config is PoseidonFieldConfig else np.asarray(x).reshape(-1)) for x in leaves]
        out = [False] * n
        for L in sorted(set(lens)):
            sel = [i for i in range(n) if lens[i] == L]
            for i, v in zip(sel, verify_paths(config, leaf_params, two_params, root_hash, [paths[i] for i in sel], [leaves[i] for i in sel])):
                out[i] = v
        return out
    idx = np.array([p.leaf_index for p in paths], dtype=np.uint64)
    sibs = np.ascontiguousarray(np.stack([np.asarray(p.leaf_sibling_hash) for p in paths]), dtype=np.uint64)
    if depth:
        auth = np.ascontiguousarray(np.stack([np.stack([np.asarray(a) for a in p.auth_path]) for p in paths]), dtype=np.uint64)
    else:
        auth = np.zeros(0, dtype=np.uint64)
    root = np.ascontiguousarray(root_hash, dtype=np.uint64)
    if isinstance(leaves[0], (bytes, bytearray)):
        lv = [bytes(x) for x in leaves]
    else:
        lv = np.stack([np.ascontiguousarray(x, dtype=np.uint64 if config is PoseidonFieldConfig else np.uint8) for x in leaves])
    ok = config.verify_abi(leaf_params, two_params, root, lv, idx, sibs, auth, depth)
    return [bool(v) for v in ok]


class MultiPath:
    """merkle_tree::MultiPath (:239-351): prefix-encoded authentication paths."""

    def __init__(self, config, leaf_siblings_hashes, auth_paths_prefix_lenghts, auth_paths_suffixes, leaf_indexes):
        self.config = config
        self.leaf_siblings_hashes = leaf_siblings_hashes
        self.auth_paths_prefix_lenghts = auth_paths_prefix_lenghts
        self.auth_paths_suffixes = auth_paths_suffixes
        self.leaf_indexes = leaf_indexes

    def decode_paths(self):
        """prefix_decode_path (:807-817) applied incrementally (:283-292)."""
        paths = []
        prev = list(self.auth_paths_suffixes[0])
        for i, leaf_index in enumerate(self.leaf_indexes):
            k = self.auth_paths_prefix_lenghts[i]
            auth = (prev[:k] if k else []) + list(self.auth_paths_suffixes[i])
            prev = auth
            paths.append(Path(self.config, self.leaf_siblings_hashes[i], auth, leaf_index))
        return paths

    def verify(self, leaf_hash_params, two_to_one_params, root_hash, leaves) -> bool:
        """:262-331 through akp_merkle_verify_multipath_*: the reference's memoisation (a node shared by several paths
        is computed once, from the first path that reaches it), one hash launch per level over the distinct nodes."""
        import ctypes as C
        cfg = self.config
        leaves = list(leaves)
        m = len(self.leaf_indexes)
        assert len(leaves) >= m, "one leaf per index"
        depth = len(self.auth_paths_suffixes[0])
        shp = cfg.digest_shape
        idx = np.array(self.leaf_indexes, dtype=np.uint64)
        sibs = np.ascontiguousarray(np.stack([np.asarray(x) for x in self.leaf_siblings_hashes]), dtype=np.uint64)
        pre = np.array(self.auth_paths_prefix_lenghts, dtype=np.uint64)
        flat = [np.asarray(d, dtype=np.uint64).reshape(shp) for sfx in self.auth_paths_suffixes for d in sfx]
        suf = np.ascontiguousarray(np.stack(flat), dtype=np.uint64) if flat else np.zeros((0,) + shp, dtype=np.uint64)
        root = np.ascontiguousarray(root_hash, dtype=np.uint64)
        ok = C.c_int32(0)
        if cfg is PoseidonFieldConfig:
            lv = np.stack([np.ascontiguousarray(x, dtype=np.uint64) for x in leaves[:m]])
            k = lv.size // (4 * m)
            check(lib.akp_merkle_verify_multipath_poseidon(leaf_hash_params.handle().h, two_to_one_params.handle().h, root.ctypes.data,
                                                           lv.ctypes.data, m, k, idx.ctypes.data, sibs.ctypes.data, pre.ctypes.data,
                                                           suf.ctypes.data if len(flat) else None, len(flat), depth, C.byref(ok)))
        else:
            mm, _, L = _ped._as_msgs([bytes(x) if isinstance(x, (bytes, bytearray)) else np.asarray(x, dtype=np.uint8).tobytes() for x in leaves[:m]])
            check(lib.akp_merkle_verify_multipath_te(_leaf_handle(cfg, leaf_hash_params).h, _two_handle(cfg, two_to_one_params).h, root.ctypes.data,
                                                     mm.ctypes.data if mm.size else None, m, L, idx.ctypes.data, sibs.ctypes.data,
                                                     pre.ctypes.data, suf.ctypes.data if len(flat) else None, len(flat), depth, C.byref(ok)))
        return bool(ok.value)

    def verify_each(self, leaf_hash_params, two_to_one_params, root_hash, leaves):
        """every decoded path verified independently (no memoisation): batched Path::verify over the decoded paths"""
        leaves = list(leaves)
        return verify_paths(self.config, leaf_hash_params, two_to_one_params, root_hash, self.decode_paths(), leaves)


class MerkleTree:
    """merkle_tree::MerkleTree<P> (:383-726)."""

    def __init__(self, config, leaf_hash_param, two_to_one_hash_param, leaf_nodes, non_leaf_nodes):
        self.config = config
        self.leaf_hash_param = leaf_hash_param
        self.two_to_one_hash_param = two_to_one_hash_param
        self.leaf_nodes = leaf_nodes
        self.non_leaf_nodes = non_leaf_nodes
        self._height = tree_height(len(leaf_nodes))

    @classmethod
    def new(cls, config, leaf_hash_param, two_to_one_hash_param, leaves):
        """MerkleTree::new (:411-422): leaves.len() must be a power of two greater than one.  Leaves of different lengths (a list)
        are hashed each with its own length, as the reference's map over the leaf iterator does."""
        n = len(leaves)
        if n < 2 or n & (n - 1):
            raise NotPowerOfTwo(5, "`leaves.len() should be power of two and greater than one")
        if _ragged_leaves(config, leaves) is not None:
            return GpuMerkleTree.new(config, leaf_hash_param, two_to_one_hash_param, leaves).to_host()
        leaf_nodes, non_leaf = config.build(leaf_hash_param, two_to_one_hash_param, leaves)
        return cls(config, leaf_hash_param, two_to_one_hash_param, leaf_nodes, non_leaf)

    @classmethod
    def new_with_leaf_digest(cls, config, leaf_hash_param, two_to_one_hash_param, leaf_digests):
        """MerkleTree::new_with_leaf_digest (:424-523): inner levels only, from given leaf digests."""
        n = len(leaf_digests)
        if n < 2 or n & (n - 1):
            raise NotPowerOfTwo(5, "`leaves.len() should be power of two and greater than one")
        leaf_nodes, non_leaf = config.build_inner(two_to_one_hash_param, leaf_digests)
        return cls(config, leaf_hash_param, two_to_one_hash_param, leaf_nodes.copy(), non_leaf)

    @classmethod
    def blank(cls, config, leaf_hash_param, two_to_one_hash_param, height):
        """MerkleTree::blank (:400-408): all leaf digests are `LeafDigest::default()`."""
        d = config.default_leaf_digest()
        return cls.new_with_leaf_digest(config, leaf_hash_param, two_to_one_hash_param, np.stack([d] * (1 << (height - 1))))

    def root(self):
        return self.non_leaf_nodes[0].copy()

    def height(self):
        return self._height

    def get_leaf_sibling_hash(self, index):  # :536-544
        return self.leaf_nodes[index + 1 if index & 1 == 0 else index - 1].copy()

    def compute_auth_path(self, index):  # :547-569
        cur = parent(convert_index_to_last_level(index, self._height))
        path = []
        while cur != 0:
            path.append(self.non_leaf_nodes[sibling(cur)].copy())
            cur = parent(cur)
        path.reverse()
        return path

    def generate_proof(self, index) -> Path:  # :572-579
        return Path(self.config, self.get_leaf_sibling_hash(index), self.compute_auth_path(index), index)

    def generate_proofs(self, indexes):
        """generate_proof for many leaves in one ABI call (akp_merkle_gather_paths)."""
        idx = np.ascontiguousarray(list(indexes), dtype=np.uint64)
        m = len(idx)
        shp = self.config.digest_shape
        fe = 2 if shp == (2, 4) else 1
        depth = self._height - 2
        sibs = np.empty((m,) + shp, dtype=np.uint64)
        auth = np.empty((m, max(depth, 0)) + shp, dtype=np.uint64)
        ln = np.ascontiguousarray(self.leaf_nodes, dtype=np.uint64)
        nl = np.ascontiguousarray(self.non_leaf_nodes, dtype=np.uint64)
        check(lib.akp_merkle_gather_paths(ln.ctypes.data, nl.ctypes.data, len(ln), fe, idx.ctypes.data, m, sibs.ctypes.data,
                                          auth.ctypes.data if depth > 0 else None))
        return [Path(self.config, sibs[i], [auth[i, j] for j in range(max(depth, 0))], int(idx[i])) for i in range(m)]

    def generate_multi_proof(self, indexes) -> MultiPath:  # :592-625
        idxs = sorted(set(indexes))
        prefix_lens, suffixes, sibs = [], [], []
        prev = []
        for i in idxs:
            sibs.append(self.get_leaf_sibling_hash(i))
            path = self.compute_auth_path(i)
            k = 0
            while k < min(len(prev), len(path)) and _eq(prev[k], path[k]):  # prefix_encode_path (:795-805)
                k += 1
            prefix_lens.append(k)
            suffixes.append(path[k:])
            prev = path
        return MultiPath(self.config, sibs, prefix_lens, suffixes, idxs)

    def _updated_path(self, index, new_leaf):  # :629-677
        cfg = self.config
        new_hash = cfg.hash_leaves(self.leaf_hash_param, [new_leaf])[0]
        if index & 1 == 0:
            l, r = new_hash, self.leaf_nodes[index + 1]
        else:
            l, r = self.leaf_nodes[index - 1], new_hash
        cur = cfg.two_to_one_evaluate(self.two_to_one_hash_param, l[None], r[None])[0]
        path = [cur]
        prev_index = parent(convert_index_to_last_level(index, self._height))
        while prev_index != 0:
            sib = self.non_leaf_nodes[sibling(prev_index)]
            l, r = (cur, sib) if is_left_child(prev_index) else (sib, cur)
            cur = cfg.two_to_one_compress(self.two_to_one_hash_param, l[None], r[None])[0]
            path.append(cur)
            prev_index = parent(prev_index)
        return new_hash, path  # bottom-to-top

    def update(self, index, new_leaf):  # :692-702
        assert index < len(self.leaf_nodes), "index out of range"
        new_hash, path = self._updated_path(index, new_leaf)
        self._apply(index, new_hash, path)

    def check_update(self, index, new_leaf, asserted_new_root) -> bool:  # :707-725
        assert index < len(self.leaf_nodes), "index out of range"
        new_hash, path = self._updated_path(index, new_leaf)
        if not _eq(path[-1], asserted_new_root):
            return False
        self._apply(index, new_hash, path)
        return True

    def update_batch(self, indices, new_leaves):
        """Batched form of `update` (extension): equal to calling update(i, leaf) for each pair in
        order (a repeated index keeps its last leaf), but every level is ONE batched hash call over the distinct
        touched nodes: log2(n) launches in total instead of log2(n) per leaf."""
        idx = np.asarray(indices, dtype=np.int64).reshape(-1)
        assert idx.size == len(new_leaves), "one leaf per index"
        if idx.size == 0:
            return
        assert idx.min() >= 0 and idx.max() < len(self.leaf_nodes), "index out of range"
        cfg = self.config
        hashes = cfg.hash_leaves(self.leaf_hash_param, new_leaves)
        last = {}
        for k, i in enumerate(idx.tolist()):  # last write wins, as with sequential updates
            last[i] = k
        keep = np.fromiter(last.values(), dtype=np.int64)
        self.leaf_nodes[idx[keep]] = hashes[keep]
        if self._height < 2:
            return
        # bottom inner level: parents of leaf pairs (evaluate on converted leaf digests, :643-654)
        pairs = np.unique(idx >> 1)
        cur = cfg.two_to_one_evaluate(self.two_to_one_hash_param, np.ascontiguousarray(self.leaf_nodes[2 * pairs]),
                                      np.ascontiguousarray(self.leaf_nodes[2 * pairs + 1]))
        nodes = pairs + (1 << (self._height - 2)) - 1  # heap indices of those parents
        self.non_leaf_nodes[nodes] = cur
        while nodes[0] != 0:  # upper levels: compress (:656-672); all nodes of a round sit on one level
            nodes = np.unique((nodes - 1) >> 1)
            cur = cfg.two_to_one_compress(self.two_to_one_hash_param, np.ascontiguousarray(self.non_leaf_nodes[2 * nodes + 1]),
                                          np.ascontiguousarray(self.non_leaf_nodes[2 * nodes + 2]))
            self.non_leaf_nodes[nodes] = cur

    def _apply(self, index, new_hash, path_bottom_to_top):
        self.leaf_nodes[index] = new_hash
        cur = convert_index_to_last_level(index, self._height)
        for node in path_bottom_to_top:
            cur = parent(cur)
            self.non_leaf_nodes[cur] = node


class GpuMerkleTree:
    """MerkleTree<P> (:383-726) RESIDENT IN HBM: the akp_merkle_tree handle of include/akp.h.  leaf_nodes / non_leaf_nodes
    stay on the device; proofs, updates and the root are served from there (what a `GpuMerkleTree<P>` of the Rust shim
    wraps).  `to_host()` materialises the reference's two vectors as a MerkleTree."""

    def __init__(self, config, leaf_hash_param, two_to_one_hash_param, handle):
        self.config = config
        self.leaf_hash_param = leaf_hash_param
        self.two_to_one_hash_param = two_to_one_hash_param
        self._h = handle
        import ctypes as C
        n, fe, h = C.c_size_t(), C.c_uint32(), C.c_size_t()
        check(lib.akp_merkle_tree_info(self._h, C.byref(n), C.byref(fe), C.byref(h)))
        self.n_leaves, self._fe, self._height = n.value, fe.value, h.value
        want = int(np.prod(config.digest_shape)) // 4
        if self._fe != want:  # every host buffer below is sized from config.digest_shape
            raise TypeError("tree handle holds %d Fr per digest, %s expects %d" % (self._fe, config.__name__, want))

    @staticmethod
    def _leaf_array(config, leaves):
        if config is PoseidonFieldConfig:
            x = np.ascontiguousarray(leaves, dtype=np.uint64)
            n = x.shape[0]
            return x, n, (x.size // (4 * n) if n else 0)
        return _ped._as_msgs(leaves)

    @classmethod
    def new(cls, config, leaf_hash_param, two_to_one_hash_param, leaves):
        import ctypes as C
        h = C.c_void_p()
        rg = _ragged_leaves(config, leaves)
        if rg is not None:  # leaves of different lengths: akp_merkle_tree_build_*_ragged
            flat, offs = rg
            fn = lib.akp_merkle_tree_build_poseidon_ragged if config is PoseidonFieldConfig else lib.akp_merkle_tree_build_te_ragged
            check(fn(_leaf_handle(config, leaf_hash_param).h, _two_handle(config, two_to_one_hash_param).h, flat.ctypes.data if flat.size else None,
                     offs.ctypes.data, len(offs) - 1, C.byref(h)))
            return cls(config, leaf_hash_param, two_to_one_hash_param, h)
        x, n, k = cls._leaf_array(config, leaves)
        fn = lib.akp_merkle_tree_build_poseidon if config is PoseidonFieldConfig else lib.akp_merkle_tree_build_te
        check(fn(_leaf_handle(config, leaf_hash_param).h, _two_handle(config, two_to_one_hash_param).h, x.ctypes.data if x.size else None, n, k, C.byref(h)))
        return cls(config, leaf_hash_param, two_to_one_hash_param, h)

    @classmethod
    def new_with_leaf_digest(cls, config, leaf_hash_param, two_to_one_hash_param, leaf_digests):
        import ctypes as C
        d = np.ascontiguousarray(leaf_digests, dtype=np.uint64).reshape((-1,) + config.digest_shape)
        h = C.c_void_p()
        fn = lib.akp_merkle_tree_from_digests_poseidon if config is PoseidonFieldConfig else lib.akp_merkle_tree_from_digests_te
        check(fn(_leaf_handle(config, leaf_hash_param).h, _two_handle(config, two_to_one_hash_param).h, d.ctypes.data, len(d), C.byref(h)))
        return cls(config, leaf_hash_param, two_to_one_hash_param, h)

    @classmethod
    def blank(cls, config, leaf_hash_param, two_to_one_hash_param, height):
        d = config.default_leaf_digest()
        return cls.new_with_leaf_digest(config, leaf_hash_param, two_to_one_hash_param, np.stack([d] * (1 << (height - 1))))

    def close(self):
        if getattr(self, "_h", None):
            lib.akp_merkle_tree_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def root(self):
        r = np.empty(self.config.digest_shape, dtype=np.uint64)
        check(lib.akp_merkle_tree_root(self._h, r.ctypes.data))
        return r

    def height(self):
        return self._height

    def to_host(self) -> MerkleTree:
        shp = self.config.digest_shape
        ln = np.empty((self.n_leaves,) + shp, dtype=np.uint64)
        nl = np.empty((self.n_leaves - 1,) + shp, dtype=np.uint64)
        check(lib.akp_merkle_tree_export(self._h, ln.ctypes.data, nl.ctypes.data))
        return MerkleTree(self.config, self.leaf_hash_param, self.two_to_one_hash_param, ln, nl)

    def generate_proofs(self, indexes):
        idx = np.ascontiguousarray(list(indexes), dtype=np.uint64)
        m, shp, depth = len(idx), self.config.digest_shape, self._height - 2
        sibs = np.empty((m,) + shp, dtype=np.uint64)
        auth = np.empty((m, max(depth, 0)) + shp, dtype=np.uint64)
        check(lib.akp_merkle_tree_gather_paths(self._h, idx.ctypes.data, m, sibs.ctypes.data, auth.ctypes.data if depth > 0 else None))
        return [Path(self.config, sibs[i], [auth[i, j] for j in range(max(depth, 0))], int(idx[i])) for i in range(m)]

    def generate_proof(self, index) -> Path:
        return self.generate_proofs([index])[0]

    def generate_multi_proof(self, indexes) -> MultiPath:
        """:592-625: sorted, de-duplicated indexes; gather, prefix_encode_path and suffix compaction on the device (round 5)."""
        import ctypes as C
        idxs = sorted(set(int(i) for i in indexes))
        idx = np.array(idxs, dtype=np.uint64)
        m, shp, depth = len(idx), self.config.digest_shape, self._height - 2
        sibs = np.empty((m,) + shp, dtype=np.uint64)
        pre = np.zeros(m, dtype=np.uint64)
        suf = np.empty((m * max(depth, 0),) + shp, dtype=np.uint64)
        cnt = C.c_size_t(0)
        # gather + prefix_encode_path + compaction on the device (akp_merkle_tree_multi_proof): only the suffixes cross PCIe
        check(lib.akp_merkle_tree_multi_proof(self._h, idx.ctypes.data, m, sibs.ctypes.data, pre.ctypes.data, suf.ctypes.data if depth > 0 else None,
                                              m * max(depth, 0), C.byref(cnt)))
        suffixes, o = [], 0
        for i in range(m):
            k = max(depth, 0) - int(pre[i])
            suffixes.append([suf[o + j].copy() for j in range(k)])
            o += k
        assert o == cnt.value
        return MultiPath(self.config, [sibs[i] for i in range(m)], [int(x) for x in pre], suffixes, idxs)

    def update_batch(self, indices, new_leaves):
        """update(indices[k], new_leaves[k]) for k in order (a repeated index keeps its LAST leaf), one hash launch per level.  New
        leaves of different lengths: the repeated indexes are resolved here (last one wins), then one call per distinct length --
        the touched paths of different calls only meet in nodes that the later call recomputes from the tree's current state."""
        idx_list = [int(i) for i in indices]
        if _ragged_leaves(self.config, list(new_leaves)) is not None:
            assert len(idx_list) == len(new_leaves), "one leaf per index"
            last = {}
            for k, i in enumerate(idx_list):
                last[i] = k
            keep = sorted(last.values())
            by_len = {}
            for k in keep:
                leaf = new_leaves[k]
                L = len(leaf) if isinstance(leaf, (bytes, bytearray)) else len(np.asarray(leaf).reshape(-1, 4) if self.config is PoseidonFieldConfig else np.asarray(leaf).reshape(-1))
                by_len.setdefault(L, []).append(k)
            for L in sorted(by_len):
                ks = by_len[L]
                self.update_batch([idx_list[k] for k in ks], [new_leaves[k] for k in ks] if self.config is not PoseidonFieldConfig
                                  else np.stack([np.asarray(new_leaves[k], dtype=np.uint64).reshape(-1, 4) for k in ks]))
            return
        idx = np.ascontiguousarray(idx_list, dtype=np.uint64)
        x, n, k = self._leaf_array(self.config, new_leaves)
        assert n == len(idx), "one leaf per index"
        check(lib.akp_merkle_tree_update_batch(self._h, idx.ctypes.data, x.ctypes.data if x.size else None, n, k))

    def update(self, index, new_leaf):  # :692-702
        self.update_batch([index], [new_leaf] if self.config is not PoseidonFieldConfig else np.asarray(new_leaf, dtype=np.uint64)[None])

    def check_update(self, index, new_leaf, asserted_new_root) -> bool:  # :707-725
        import ctypes as C
        x, _, k = self._leaf_array(self.config, [new_leaf] if self.config is not PoseidonFieldConfig else np.asarray(new_leaf, dtype=np.uint64)[None])
        root = np.ascontiguousarray(asserted_new_root, dtype=np.uint64)
        ok = C.c_int32(0)
        check(lib.akp_merkle_tree_check_update(self._h, int(index), x.ctypes.data if x.size else None, k, root.ctypes.data, C.byref(ok)))
        return bool(ok.value)


class MultiGpu:
    """akp_multi: the GPUs of one node driven from one process, with the RCCL communicator inside the product library."""

    def __init__(self, device_ids):
        import ctypes as C
        ids = (C.c_int32 * len(device_ids))(*device_ids)
        h = C.c_void_p()
        check(lib.akp_multi_create(ids, len(device_ids), C.byref(h)))
        self._h = h
        self._ctxs = {}
        self._param_owners = []
        self._trees = []
        self.size = lib.akp_multi_size(h)

    def ctx(self, i):
        from ._lib import Context
        import ctypes as C
        if i not in self._ctxs:
            c = Context.__new__(Context)
            c.h = C.c_void_p(lib.akp_multi_ctx(self._h, i))
            c.device_id = None
            c.close = lambda: None  # owned by the akp_multi handle
            self._ctxs[i] = c
        return self._ctxs[i]

    def build_sharded(self, config, leaf_hash_param, two_to_one_hash_param, leaves, want_nodes=True):
        """MerkleTree::new (:411-523) over all devices: akp_merkle_build_sharded_{poseidon,te}.  Returns
        (leaf_nodes, non_leaf_nodes, root) in the reference's global heap order (nodes None when want_nodes is False)."""
        la, ta = self._handles(config, leaf_hash_param, two_to_one_hash_param)
        shp = config.digest_shape
        if config is PoseidonFieldConfig:
            x = np.ascontiguousarray(leaves, dtype=np.uint64)
            n = x.shape[0]
            k = x.size // (4 * n) if n else 0
            fn = lib.akp_merkle_build_sharded_poseidon
        else:
            x, n, k = _ped._as_msgs(leaves)
            fn = lib.akp_merkle_build_sharded_te
        ln = np.empty((n,) + shp, dtype=np.uint64) if want_nodes else None
        nl = np.empty((max(n - 1, 0),) + shp, dtype=np.uint64) if want_nodes else None
        root = np.empty(shp, dtype=np.uint64)
        check(fn(self._h, la, ta, x.ctypes.data if x.size else None, n, k, ln.ctypes.data if want_nodes else None,
                 nl.ctypes.data if want_nodes else None, root.ctypes.data))
        return ln, nl, root

    def _handles(self, config, leaf_hash_param, two_to_one_hash_param):
        import ctypes as C
        G = self.size
        lh = [_leaf_handle(config, leaf_hash_param, self.ctx(r)) for r in range(G)]
        th = [_two_handle(config, two_to_one_hash_param, self.ctx(r)) for r in range(G)]
        for cfg in (leaf_hash_param, two_to_one_hash_param):  # destroyed with this object, before their contexts
            if not any(cfg is c for c in self._param_owners):
                self._param_owners.append(cfg)
        return (C.c_void_p * G)(*[h.h for h in lh]), (C.c_void_p * G)(*[h.h for h in th])

    def build_tree(self, config, leaf_hash_param, two_to_one_hash_param, leaves=None, device_leaf_ptrs=None, n_leaves=None, leaf_len=None):
        """MerkleTree::new (:411-422) sharded over all devices and RESIDENT there (akp_multi_tree_build_*): a ShardedMerkleTree.
        `leaves`: host leaves in global order; or `device_leaf_ptrs` = one device address per slot (that slot's n / G leaves,
        already in its memory) with `n_leaves` and `leaf_len`."""
        import ctypes as C
        la, ta = self._handles(config, leaf_hash_param, two_to_one_hash_param)
        h = C.c_void_p()
        pos = config is PoseidonFieldConfig
        rg = _ragged_leaves(config, leaves) if device_leaf_ptrs is None else None
        if rg is not None:
            flat, offs = rg
            fn = lib.akp_multi_tree_build_poseidon_ragged if pos else lib.akp_multi_tree_build_te_ragged
            check(fn(self._h, la, ta, flat.ctypes.data if flat.size else None, offs.ctypes.data, len(offs) - 1, C.byref(h)))
        elif device_leaf_ptrs is None:
            x, n, k = GpuMerkleTree._leaf_array(config, leaves)
            fn = lib.akp_multi_tree_build_poseidon if pos else lib.akp_multi_tree_build_te
            check(fn(self._h, la, ta, x.ctypes.data if x.size else None, n, k, C.byref(h)))
        else:
            ptrs = (C.c_void_p * self.size)(*[int(p) for p in device_leaf_ptrs])
            fn = lib.akp_multi_tree_build_poseidon_dev if pos else lib.akp_multi_tree_build_te_dev
            check(fn(self._h, la, ta, ptrs, n_leaves, leaf_len, C.byref(h)))
        t = ShardedMerkleTree(self, config, h)
        self._trees.append(t)
        return t

    def build_tree_with_leaf_digest(self, config, leaf_hash_param, two_to_one_hash_param, leaf_digests):
        """MerkleTree::new_with_leaf_digest (:424-523) sharded and resident; with all-default digests: MerkleTree::blank (:400-408)"""
        import ctypes as C
        la, ta = self._handles(config, leaf_hash_param, two_to_one_hash_param)
        d = np.ascontiguousarray(leaf_digests, dtype=np.uint64).reshape((-1,) + config.digest_shape)
        h = C.c_void_p()
        fn = lib.akp_multi_tree_from_digests_poseidon if config is PoseidonFieldConfig else lib.akp_multi_tree_from_digests_te
        check(fn(self._h, la, ta, d.ctypes.data, len(d), C.byref(h)))
        t = ShardedMerkleTree(self, config, h)
        self._trees.append(t)
        return t

    def last_phases(self):
        """phase breakdown of the last build_sharded / build_tree call (akp_multi_last_phases), milliseconds, maximum over the devices"""
        import ctypes as C
        ms = (C.c_double * 5)()
        check(lib.akp_multi_last_phases(self._h, ms))
        return {"copy_in_and_subtree_ms": ms[0], "allgather_ms": ms[1], "top_levels_ms": ms[2], "copy_out_ms": ms[3], "whole_call_ms": ms[4]}

    def close(self):
        if getattr(self, "_h", None):
            for t in self._trees:  # resident trees first (they pin their parameter handles)
                t.close()
            self._trees.clear()
            for cfg in self._param_owners:  # parameter handles created on our contexts go first
                ours = {id(c) for c in self._ctxs.values()}
                for key in [k for k in cfg._handles if (k[0] if isinstance(k, tuple) else k) in ours]:  # te handles are keyed (ctx, kind)
                    cfg._handles.pop(key).__del__()
            self._param_owners.clear()
            self._ctxs.clear()
            lib.akp_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ShardedMerkleTree:
    """MerkleTree<P> sharded over the devices of a MultiGpu and resident there (akp_multi_tree, include/akp.h): every device keeps
    the sub-tree of its leaf range in its HBM, the top log2(G) levels are replicated.  root / generate_proofs / update_batch
    follow GpuMerkleTree; `to_host()` gathers the reference's two vectors in global heap order."""

    def __init__(self, multi, config, handle):
        import ctypes as C
        self._multi, self.config, self._h = multi, config, handle
        n, fe, h, g = C.c_size_t(), C.c_uint32(), C.c_size_t(), C.c_int32()
        check(lib.akp_multi_tree_info(self._h, C.byref(n), C.byref(fe), C.byref(h), C.byref(g)))
        self.n_leaves, self._fe, self._height, self.n_dev = n.value, fe.value, h.value, g.value

    def close(self):
        if getattr(self, "_h", None):
            lib.akp_multi_tree_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def root(self):
        r = np.empty(self.config.digest_shape, dtype=np.uint64)
        check(lib.akp_multi_tree_root(self._h, r.ctypes.data))
        return r

    def height(self):
        return self._height

    def generate_proofs(self, indexes):
        idx = np.ascontiguousarray(list(indexes), dtype=np.uint64)
        m, shp, depth = len(idx), self.config.digest_shape, self._height - 2
        sibs = np.empty((m,) + shp, dtype=np.uint64)
        auth = np.empty((m, max(depth, 0)) + shp, dtype=np.uint64)
        check(lib.akp_multi_tree_gather_paths(self._h, idx.ctypes.data, m, sibs.ctypes.data, auth.ctypes.data if depth > 0 else None))
        return [Path(self.config, sibs[i], [auth[i, j] for j in range(max(depth, 0))], int(idx[i])) for i in range(m)]

    def generate_proof(self, index) -> Path:
        return self.generate_proofs([index])[0]

    def update_batch(self, indices, new_leaves):
        idx = np.ascontiguousarray(list(indices), dtype=np.uint64)
        x, n, k = GpuMerkleTree._leaf_array(self.config, new_leaves)
        assert n == len(idx), "one leaf per index"
        check(lib.akp_multi_tree_update_batch(self._h, idx.ctypes.data, x.ctypes.data if x.size else None, n, k))

    def update(self, index, new_leaf):  # :692-702
        self.update_batch([index], [new_leaf] if self.config is not PoseidonFieldConfig else np.asarray(new_leaf, dtype=np.uint64)[None])

    def check_update(self, index, new_leaf, asserted_new_root) -> bool:  # :707-725
        import ctypes as C
        x, _, k = GpuMerkleTree._leaf_array(self.config, [new_leaf] if self.config is not PoseidonFieldConfig else np.asarray(new_leaf, dtype=np.uint64)[None])
        root = np.ascontiguousarray(asserted_new_root, dtype=np.uint64)
        ok = C.c_int32(0)
        check(lib.akp_multi_tree_check_update(self._h, int(index), x.ctypes.data if x.size else None, k, root.ctypes.data, C.byref(ok)))
        return ok.value == 1

    def generate_multi_proof(self, indexes) -> MultiPath:
        """:592-625: sorted, de-duplicated indexes; paths gathered from the owning shards, prefix_encode_path through the ABI"""
        import ctypes as C
        idxs = sorted(set(int(i) for i in indexes))
        idx = np.array(idxs, dtype=np.uint64)
        m, shp, depth = len(idx), self.config.digest_shape, max(self._height - 2, 0)
        sibs = np.empty((m,) + shp, dtype=np.uint64)
        auth = np.empty((m, depth) + shp, dtype=np.uint64)
        check(lib.akp_multi_tree_gather_paths(self._h, idx.ctypes.data, m, sibs.ctypes.data, auth.ctypes.data if depth else None))
        pre = np.zeros(m, dtype=np.uint64)
        suf = np.empty((m * depth,) + shp, dtype=np.uint64)
        cnt = C.c_size_t(0)
        check(lib.akp_merkle_multipath_encode(auth.ctypes.data if depth else None, m, depth, self._fe, pre.ctypes.data, suf.ctypes.data if depth else None, C.byref(cnt)))
        suffixes, o = [], 0
        for i in range(m):
            k = depth - int(pre[i])
            suffixes.append([suf[o + j].copy() for j in range(k)])
            o += k
        return MultiPath(self.config, [sibs[i] for i in range(m)], [int(x) for x in pre], suffixes, idxs)

    def to_host(self, leaf_hash_param=None, two_to_one_hash_param=None) -> MerkleTree:
        shp = self.config.digest_shape
        ln = np.empty((self.n_leaves,) + shp, dtype=np.uint64)
        nl = np.empty((self.n_leaves - 1,) + shp, dtype=np.uint64)
        check(lib.akp_multi_tree_export(self._h, ln.ctypes.data, nl.ctypes.data))
        return MerkleTree(self.config, leaf_hash_param, two_to_one_hash_param, ln, nl)
