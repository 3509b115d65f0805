"""`merkle`: BASELINE configs[2] -- MerkleTree::new, Poseidon leaf + two-to-one, 2^24 1-Fr leaves in total (strong scaling over the
ranks: leaf-range shards + ONE all-gather of the sub-roots), plus the same tree through the C ABI's one-process multi-device
entry points in a child process."""
import json
import os
import subprocess
import sys
import time

from .common import HBM_PEAK_GBS


def one_process_leg(log2_leaves, bh_log2_per_gpu=0):
    """Child process of the Merkle leg: the same 2^k-leaf Poseidon tree -- and the Bowe-Hopwood tree of BASELINE configs[4] at
    2^j leaves per device -- through the C ABI's single-process multi-device entry points (what a Rust host calls): all visible
    GPUs (a power of two, at most 8), leaves in pageable host memory, one host thread per device, RCCL all-gather of the
    sub-roots inside libakp.so.  PCIe-inclusive, with the per-phase breakdown the library records (akp_multi_last_phases), so
    that the first run on a multi-GPU node yields copy-in + sub-tree / all-gather / top / copy-out, not one number.  Also the
    sharded RESIDENT tree (akp_multi_tree_*): built from the same leaves, 2^12 proofs and 2^10 updates served from the shards.
    Prints one JSON line."""
    import numpy as np
    import torch
    import crypto_primitives_amd as cpa
    from crypto_primitives_amd import field
    g = 1
    while g * 2 <= min(torch.cuda.device_count(), 8):
        g *= 2
    total = 1 << log2_leaves
    cfg = cpa.get_default_poseidon_parameters(2, False)
    leaves = field.random_fr(total, seed=0xA5A50003).reshape(total, 1, 4)
    mg = cpa.MultiGpu(list(range(g)))

    def timed(config, lp, tp, lv):
        mg.build_sharded(config, lp, tp, lv[: 1 << 12], want_nodes=False)  # handles, tables, scratch, RCCL warm-up
        mg.build_sharded(config, lp, tp, lv, want_nodes=False)
        best = None
        for _ in range(3):
            m0 = time.perf_counter()
            _, _, mroot = mg.build_sharded(config, lp, tp, lv, want_nodes=False)
            sec = time.perf_counter() - m0
            if best is None or sec < best[0]:
                best = (sec, mg.last_phases(), mroot)
        return best
    secs, phases, mroot = timed(cpa.PoseidonFieldConfig, cfg, cfg, leaves)
    res = {"entry_point": "akp_merkle_build_sharded_poseidon", "devices": g, "seconds": secs, "phases_ms": phases,
           "includes": "copy-in of the leaves from pageable memory over PCIe (one host thread per device)",
           "collective": "ncclAllGather of %d sub-roots" % g, "root_limb0": int(np.asarray(mroot).reshape(-1)[0])}
    # the sharded RESIDENT tree: nothing but the root, the requested proofs and the new leaves crosses PCIe
    t0 = time.perf_counter()
    st = mg.build_tree(cpa.PoseidonFieldConfig, cfg, cfg, leaves)
    build_s = time.perf_counter() - t0
    bphases = mg.last_phases()
    root_same = bool(np.array_equal(np.asarray(mroot).reshape(-1), np.asarray(st.root()).reshape(-1)))
    rng = np.random.default_rng(0xA5A50021)
    idx = rng.integers(0, total, size=1 << 12).astype(np.uint64)
    import ctypes as C
    from crypto_primitives_amd._lib import lib, check
    depth = log2_leaves - 1
    sib, auth = np.empty((len(idx), 4), np.uint64), np.empty((len(idx), depth, 4), np.uint64)
    check(lib.akp_multi_tree_gather_paths(st._h, idx.ctypes.data, 16, sib.ctypes.data, auth.ctypes.data))
    t0 = time.perf_counter()  # the C call alone (building 4096 Python Path objects costs ten times as much)
    check(lib.akp_multi_tree_gather_paths(st._h, idx.ctypes.data, len(idx), sib.ctypes.data, auth.ctypes.data))
    proof_s = time.perf_counter() - t0
    proofs = st.generate_proofs(idx[:64])
    assert C.sizeof(C.c_uint64) == 8 and np.array_equal(np.asarray(proofs[5].auth_path).reshape(depth, 4), auth[5])
    ok = all(cpa.merkle_tree.verify_paths(cpa.PoseidonFieldConfig, cfg, cfg, st.root(), proofs[:64], [leaves[int(i)] for i in idx[:64]]))
    upd = rng.integers(0, total, size=1 << 10).astype(np.uint64)
    new = field.random_fr(len(upd), seed=0xA5A50022).reshape(len(upd), 1, 4)
    t0 = time.perf_counter()
    st.update_batch(upd, new)
    upd_s = time.perf_counter() - t0
    res["resident_tree"] = {"entry_points": "akp_multi_tree_build_poseidon / _gather_paths / _update_batch", "build_seconds": build_s, "build_phases_ms": bphases,
                            "root_before_updates_matches_the_sharded_build": root_same,
                            "proofs": len(idx), "proofs_ms": proof_s * 1e3, "sampled_proofs_verify": bool(ok), "updates": len(upd), "update_ms": upd_s * 1e3,
                            "bytes_moved_off_the_devices": "root + proofs only (the 2^%d-leaf node arrays stay in HBM)" % log2_leaves}
    st.close()
    if bh_log2_per_gpu:
        from crypto_primitives_amd import params as cparams
        from crypto_primitives_amd.crh import bowe_hopwood
        B = bowe_hopwood.Parameters(cparams.bowe_hopwood_generators(0xA5A50005, 63, 9))
        nb = g << bh_log2_per_gpu
        per = 1 << bh_log2_per_gpu
        lv = np.concatenate([np.random.default_rng(0xA5A50005 + r).integers(0, 256, size=(per, 32), dtype=np.uint8) for r in range(g)])  # rank r's shard of the torchrun leg
        bsecs, bphases, broot = timed(cpa.BoweHopwoodByteConfig, B, B, lv)
        res["bowe_hopwood"] = {"entry_point": "akp_merkle_build_sharded_te", "leaves": nb, "leaves_per_device": per, "seconds": bsecs, "leaves_per_s": nb / bsecs,
                               "phases_ms": bphases, "root_limb0": int(np.asarray(broot).reshape(-1)[0])}
    mg.close()
    print(json.dumps(res))
    return 0


def run(env):
    args, np, torch = env.args, env.np, env.torch
    if not args.merkle_log2:
        return None
    total = 1 << args.merkle_log2
    per = total // env.world
    leaves = env.field.random_fr(per, seed=0xA5A50003 + env.rank).reshape(per, 1, 4)
    d_leaves = torch.from_numpy(leaves.view(np.int64)).to(env.dev)
    backend = env.GpuPoseidonBackend(env.cfg, env.cfg, leaf_len=1, device=env.dev)
    env.build_sharded(backend, d_leaves, total, env.dist)  # untimed full-size warm-up build (allocations, RCCL, clocks)
    env.barrier()
    m0 = time.perf_counter()
    res = env.build_sharded(backend, d_leaves, total, env.dist)
    env.barrier()
    msec = env.max_over_ranks(time.perf_counter() - m0)
    merkle = {"config": "BASELINE configs[2]: MerkleTree::new, Poseidon leaf + two-to-one, 1-Fr leaves", "leaves": total, "seconds": msec,
              "leaves_per_s": total / msec, "scaling": "strong", "permutations": 2 * total - 1,
              "root_limb0": int(np.asarray(res["root"]).reshape(-1)[0]), "algorithmic_GBps": 160.0 * total / msec / 1e9,
              "hbm_frac": 160.0 * total / msec / 1e9 / HBM_PEAK_GBS}
    if env.rank == 0:
        # sampled parity on rank 0's sub-tree: leaf digests, and inner nodes recomputed by the oracle from their children
        ln = res["leaf_nodes"].cpu().numpy().view(np.uint64).reshape(per, 4)
        nl = res["non_leaf_nodes"].cpu().numpy().view(np.uint64).reshape(per - 1, 4)
        si = np.unique(np.linspace(0, per - 1, 257).astype(np.int64))
        ok = np.array_equal(ln[si], env.ora.crh_batch(np.ascontiguousarray(leaves[si]), 1, threads=env.ora_threads))
        ni = np.unique(np.concatenate([np.arange(0, min(64, per - 1)), np.linspace(0, per - 2, 257).astype(np.int64)]))

        def child(ix):  # heap children: inner nodes below per - 1, then the leaf digests
            return np.where((ix < per - 1)[:, None], nl[np.clip(ix, 0, per - 2)], ln[np.clip(ix - (per - 1), 0, per - 1)])
        ok = ok and np.array_equal(nl[ni], env.ora.two_to_one_batch(np.ascontiguousarray(child(2 * ni + 1)), np.ascontiguousarray(child(2 * ni + 2)),
                                                                    threads=env.ora_threads))
        merkle["sampled_parity_bit_exact"] = bool(ok)
        if not ok:
            raise SystemExit("Merkle leg: sampled nodes differ from the oracle")
    # the same tree through the C ABI's single-process multi-device entry point (what a Rust host calls): all visible GPUs
    # (a power of two), leaves in host memory, RCCL all-gather of the sub-roots inside libakp.so.  PCIe-inclusive.  Only when
    # this is the one process of the run; any failure is reported, not fatal (n_dev > 1 cannot be tested on a one-GPU box).
    if env.world == 1 and not env.shared_gpu and os.environ.get("AKP_BENCH_NO_MULTI") != "1":
        # in a child process with a time limit: a first-ever n_dev > 1 RCCL bring-up must not be able to take the headline
        # measurement down with it (a crash or a hang there is reported here, nothing else)
        cmd = [sys.executable, env.bench_path, "--one-process-leg", str(args.merkle_log2), "--one-process-bh", str(args.bh_merkle_log2)]
        try:
            cp = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
            line = [l for l in cp.stdout.splitlines() if l.startswith("{")]
            if cp.returncode == 0 and line:
                leg = json.loads(line[-1])
                leg["root_matches"] = leg.pop("root_limb0", None) == merkle["root_limb0"]
                merkle["one_process_c_abi"] = leg
            else:
                merkle["one_process_c_abi"] = {"error": "exit %d: %s" % (cp.returncode, (cp.stderr or cp.stdout)[-300:])}
        except subprocess.TimeoutExpired:  # pragma: no cover
            merkle["one_process_c_abi"] = {"error": "no result within 300 s (child process stopped)"}
        except Exception as exc:  # pragma: no cover
            merkle["one_process_c_abi"] = {"error": repr(exc)[:300]}
    return merkle
