"""`sweep`: north_star's "synthetic 2^20 - 2^26 leaf batches" on ONE GPU -- permutations/s, Poseidon-tree leaves/s and Bowe-Hopwood
63x9-tree leaves/s (32-byte leaves) at 2^20, 2^22, 2^24, 2^26 with the HBM fraction of each point.  Inputs are generated on the device (a 2^20-element random block tiled: the
kernels are data-independent; parity is what the headline probe, the merkle leg and the test-suite establish at these sizes),
three launches per point after ~40 ms of untimed launches of the same kind (clock settle), device time between events on the launch stream."""
from .common import ALGO_BYTES_PER_PERM, HBM_PEAK_GBS


def run(env, sizes=(20, 22, 24, 26)):
    np, torch, lib, check = env.np, env.torch, env.lib, env.check
    if env.rank != 0 or env.args.no_sweep:
        return None
    t = env.cfg.t
    block = torch.from_numpy(env.field.random_fr((1 << 20) * t, seed=0xA5A50031).reshape(-1, t, 4).view(np.int64)).to(env.dev)
    out = {"inputs": "device-generated (a random 2^20-state block tiled)", "launches_per_point": 3, "settle": "20 / 5 / 1 / 1 untimed launches before the timed ones at 2^20 / 2^22 / 2^24 / 2^26 (clock ramp)", "points": {}}

    def timed(fn, settle=1):
        # `settle` untimed launches first: from an idle device the first launches run on ramping clocks (the 2^20 point of round 4 sat 17 %
        # under the headline for that reason); ~40 ms of the same launch before the three timed ones
        for _ in range(settle):
            fn()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(3)]
        for a, b in evs:
            a.record()
            fn()
            b.record()
        torch.cuda.synchronize(env.dev)
        ms = sorted(a.elapsed_time(b) for a, b in evs)
        return ms[1]
    hb = None
    if env.args.bh_merkle_log2:  # BASELINE configs[4]'s hash pair over the same sizes (2^26 = its stated size, here on ONE GPU)
        from crypto_primitives_amd import params as cparams
        from crypto_primitives_amd.crh import bowe_hopwood
        hb = bowe_hopwood.Parameters(cparams.bowe_hopwood_generators(0xA5A50005, 63, 9)).handle(env.ctx)
        bblock = torch.from_numpy(np.random.default_rng(0xA5A50032).integers(0, 256, size=(1 << 20, 32), dtype=np.uint8)).to(env.dev)
    for lg in sizes:
        n = 1 << lg
        point = {}
        settle = max(1, 20 >> max(0, lg - 20))  # 20 launches at 2^20, 5 at 2^22, 1 from 2^24
        try:
            states = block.repeat(n >> 20, 1, 1) if lg > 20 else block.clone()
            ms = timed(lambda: check(lib.akp_poseidon_permute_batch_dev(env.ph.h, states.data_ptr(), n, env.stream)), settle)
            point["permutations_per_s"] = n / (ms / 1e3)
            point["permute_ms"] = ms
            point["permute_hbm_frac"] = ALGO_BYTES_PER_PERM * n / (ms / 1e3) / 1e9 / HBM_PEAK_GBS
            del states
            leaves = block[:, 0, :].contiguous().repeat(n >> 20, 1) if lg > 20 else block[:, 0, :].contiguous()
            ln = torch.empty((n, 4), dtype=torch.int64, device=env.dev)
            nl = torch.empty((n - 1, 4), dtype=torch.int64, device=env.dev)
            ms = timed(lambda: check(lib.akp_merkle_build_poseidon_dev(env.ph.h, env.ph.h, leaves.data_ptr(), n, 1, ln.data_ptr(), nl.data_ptr(), env.stream)), max(1, settle // 2))
            point["tree_leaves_per_s"] = n / (ms / 1e3)
            point["tree_ms"] = ms
            point["tree_hbm_frac"] = 160.0 * n / (ms / 1e3) / 1e9 / HBM_PEAK_GBS  # 32 B leaf in, 2 x 32 B digests out, 2 x 32 B re-read per inner node
            del leaves
            if hb is not None:
                bl = bblock.repeat(n >> 20, 1) if lg > 20 else bblock
                ms = timed(lambda: check(lib.akp_merkle_build_te_dev(hb.h, hb.h, bl.data_ptr(), n, 32, ln.data_ptr(), nl.data_ptr(), env.stream)), max(1, settle // 2))
                point["bh_tree_leaves_per_s"] = n / (ms / 1e3)
                point["bh_tree_ms"] = ms
                point["bh_tree_hbm_frac"] = 160.0 * n / (ms / 1e3) / 1e9 / HBM_PEAK_GBS
                del bl
            del ln, nl
        except Exception as exc:  # pragma: no cover - a crowded device: report what fitted
            point["error"] = repr(exc)[:200]
            torch.cuda.empty_cache()
        out["points"]["2^%d" % lg] = point
    return out
