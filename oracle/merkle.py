"""Oracle restatement of the reference MerkleTree (test infrastructure).

Follows merkle_tree/mod.rs: new :411-422, new_with_leaf_digest :424-523,
proofs :547-625, Path::verify :172-212, MultiPath::verify :262-331,
update :629-702, index helpers :730-817.
The tree is generic over three callables (the Config trait, :83-122):
    leaf_hash(leaf) -> leaf_digest
    two_to_one_evaluate(conv(l), conv(r)) -> inner digest   (bottom level, after `convert`)
    two_to_one_compress(l, r) -> inner digest               (upper levels)
    convert(leaf_digest) -> two-to-one input                (DigestConverter :48-78)
"""


def tree_height(n):  # :730-737
    return 1 if n == 1 else (n.bit_length() - 1) + 1


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


def convert_index_to_last_level(index, height):  # :783-786
    return index + (1 << (height - 1)) - 1


class Path:
    def __init__(self, leaf_sibling_hash, auth_path, leaf_index):
        self.leaf_sibling_hash = leaf_sibling_hash
        self.auth_path = auth_path
        self.leaf_index = leaf_index


class MerkleTree:
    def __init__(self, leaf_hash, two_to_one_evaluate, two_to_one_compress, convert, leaves=None, leaf_digests=None):
        self.leaf_hash = leaf_hash
        self.t_eval = two_to_one_evaluate
        self.t_comp = two_to_one_compress
        self.convert = convert
        if leaf_digests is None:
            leaf_digests = [leaf_hash(l) for l in leaves]  # :417-419
        n = len(leaf_digests)
        assert n > 1 and (n & (n - 1)) == 0, "leaves.len() should be power of two and greater than one"
        self.height = tree_height(n)
        non_leaf = [None] * (n - 1)
        # level start indices 0,1,3,7,... (:446-451)
        idx, level_indices = 0, []
        for _ in range(self.height - 1):
            level_indices.append(idx)
            idx = left_child(idx)
        start = level_indices.pop()
        upper = left_child(start)
        for i in range(start, upper):  # bottom non-leaf level (:454-483)
            l = left_child(i) - upper
            r = right_child(i) - upper
            non_leaf[i] = self.t_eval(convert(leaf_digests[l]), convert(leaf_digests[r]))
        for start in reversed(level_indices):  # upper levels (:486-515)
            upper = left_child(start)
            for i in range(start, upper):
                non_leaf[i] = self.t_comp(non_leaf[left_child(i)], non_leaf[right_child(i)])
        self.leaf_nodes = list(leaf_digests)
        self.non_leaf_nodes = non_leaf

    def root(self):
        return self.non_leaf_nodes[0]

    def get_leaf_sibling_hash(self, index):  # :536-544
        return self.leaf_nodes[index + 1] if index & 1 == 0 else self.leaf_nodes[index - 1]

    def compute_auth_path(self, index):  # :547-569
        h = tree_height(len(self.leaf_nodes))
        cur = parent(convert_index_to_last_level(index, h))
        path = []
        while cur != 0:
            path.append(self.non_leaf_nodes[sibling(cur)])
            cur = parent(cur)
        path.reverse()
        return path

    def generate_proof(self, index):  # :572-579
        return Path(self.get_leaf_sibling_hash(index), self.compute_auth_path(index), index)

    def generate_multi_proof(self, indexes):  # :592-625
        idxs = sorted(set(indexes))
        prefix_lens, suffixes, sib = [], [], []
        prev = []
        for i in idxs:
            sib.append(self.get_leaf_sibling_hash(i))
            path = self.compute_auth_path(i)
            k = 0
            while k < min(len(prev), len(path)) and prev[k] == path[k]:
                k += 1
            prefix_lens.append(k)
            suffixes.append(path[k:])
            prev = path
        return {"leaf_indexes": idxs, "auth_paths_prefix_lenghts": prefix_lens,
                "auth_paths_suffixes": suffixes, "leaf_siblings_hashes": sib}

    def verify(self, path: Path, root, leaf):  # Path::verify :172-212
        claimed = self.leaf_hash(leaf)
        if path.leaf_index & 1 == 0:
            l, r = claimed, path.leaf_sibling_hash
        else:
            l, r = path.leaf_sibling_hash, claimed
        cur = self.t_eval(self.convert(l), self.convert(r))
        index = path.leaf_index >> 1
        for level in range(len(path.auth_path) - 1, -1, -1):
            if index & 1 == 0:
                l, r = cur, path.auth_path[level]
            else:
                l, r = path.auth_path[level], cur
            cur = self.t_comp(l, r)
            index >>= 1
        return cur == root

    def verify_multi(self, mp, root, leaves):  # MultiPath::verify :262-331
        prev = list(mp["auth_paths_suffixes"][0])
        for i, leaf_index in enumerate(mp["leaf_indexes"]):
            k = mp["auth_paths_prefix_lenghts"][i]
            auth = (prev[:k] if k else []) + list(mp["auth_paths_suffixes"][i])
            prev = auth
            if not self.verify(Path(mp["leaf_siblings_hashes"][i], auth, leaf_index), root, leaves[i]):
                return False
        return True

    def verify_multi_memo(self, mp, root, leaves):
        """MultiPath::verify exactly as written (:262-331), INCLUDING its hash look-up table keyed by the node's index in
        the tree: a node reached by several paths is computed once, from the first of them, and later paths reuse that
        value without hashing their own children."""
        tree_height = len(mp["auth_paths_suffixes"][0]) + 2
        lut = {}
        prev = list(mp["auth_paths_suffixes"][0])
        for i, leaf_index in enumerate(mp["leaf_indexes"]):
            k = mp["auth_paths_prefix_lenghts"][i]
            auth = (prev[:k] if k else []) + list(mp["auth_paths_suffixes"][i])  # prefix_decode_path :807-817
            prev = auth
            claimed = self.leaf_hash(leaves[i])
            sib = mp["leaf_siblings_hashes"][i]
            l, r = (claimed, sib) if leaf_index & 1 == 0 else (sib, claimed)
            index = leaf_index >> 1
            index_in_tree = parent(convert_index_to_last_level(leaf_index, tree_height))
            if index_in_tree not in lut:
                lut[index_in_tree] = self.t_eval(self.convert(l), self.convert(r))
            cur = lut[index_in_tree]
            for level in range(len(auth) - 1, -1, -1):
                l, r = (cur, auth[level]) if index & 1 == 0 else (auth[level], cur)
                index >>= 1
                index_in_tree = parent(index_in_tree)
                if index_in_tree not in lut:
                    lut[index_in_tree] = self.t_comp(l, r)
                cur = lut[index_in_tree]
            if cur != root:
                return False
        return True

    def check_update(self, index, new_leaf, asserted_new_root):  # :707-725
        import copy
        trial = copy.copy(self)
        trial.leaf_nodes, trial.non_leaf_nodes = list(self.leaf_nodes), list(self.non_leaf_nodes)
        trial.update(index, new_leaf)
        if trial.non_leaf_nodes[0] != asserted_new_root:
            return False
        self.leaf_nodes, self.non_leaf_nodes = trial.leaf_nodes, trial.non_leaf_nodes
        return True

    def update(self, index, new_leaf):  # :629-702
        assert index < len(self.leaf_nodes)
        new_hash = self.leaf_hash(new_leaf)
        self.leaf_nodes[index] = new_hash
        if index & 1 == 0:
            l, r = new_hash, self.leaf_nodes[index + 1]
        else:
            l, r = self.leaf_nodes[index - 1], new_hash
        cur_idx = parent(convert_index_to_last_level(index, self.height))
        cur = self.t_eval(self.convert(l), self.convert(r))
        self.non_leaf_nodes[cur_idx] = cur
        while cur_idx != 0:
            sib = sibling(cur_idx)
            if cur_idx % 2 == 1:
                cur = self.t_comp(cur, self.non_leaf_nodes[sib])
            else:
                cur = self.t_comp(self.non_leaf_nodes[sib], cur)
            cur_idx = parent(cur_idx)
            self.non_leaf_nodes[cur_idx] = cur
