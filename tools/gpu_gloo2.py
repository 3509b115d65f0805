"""Two ranks sharing GPU 0 (gloo for the collective, the real GPU backends for the hashing): the N > 1 control flow of
build_sharded with GpuPoseidonBackend / GpuTeBackend, checked against a single-process build.
  python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/gpu_gloo2.py"""
import os, sys
sys.path.insert(0, ".")
import numpy as np, torch, torch.distributed as dist
import crypto_primitives_amd as cpa
from crypto_primitives_amd import field, params
from crypto_primitives_amd.crh import bowe_hopwood
from crypto_primitives_amd.distributed import GpuPoseidonBackend, GpuTeBackend, build_sharded, shard_range

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
cfg = cpa.get_default_poseidon_parameters(2, False)
n = 1 << 12
leaves = field.random_fr(n, seed=5).reshape(n, 1, 4)
lo, hi = shard_range(n, rank, world)
be = GpuPoseidonBackend(cfg, cfg, leaf_len=1, device=dev)
d_all = torch.from_numpy(leaves.view(np.int64)).to(dev)
res = build_sharded(be, d_all[lo:hi], n, dist)
full = build_sharded(be, d_all, n, None)
nl = full["non_leaf_nodes"].cpu().numpy().view(np.uint64)
assert np.array_equal(res["root"], nl[0]) and np.array_equal(res["top_nodes"], nl[: world - 1]), "poseidon top mismatch"
B = bowe_hopwood.Parameters(params.bowe_hopwood_generators(0xA5A50005, 63, 9))
tb = GpuTeBackend(B, B, device=dev)
bl = torch.from_numpy(np.random.default_rng(1).integers(0, 256, size=(n, 32), dtype=np.uint8)).to(dev)
res = build_sharded(tb, bl[lo:hi], n, dist)
full = build_sharded(tb, bl, n, None)
nl = full["non_leaf_nodes"].cpu().numpy().view(np.uint64)
assert np.array_equal(res["root"], nl[0]) and np.array_equal(res["top_nodes"], nl[: world - 1]), "bh top mismatch"
dist.barrier(); dist.destroy_process_group()
print("rank %d ok" % rank)
