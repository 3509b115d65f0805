"""`bh_merkle`: BASELINE configs[4] -- MerkleTree::new, Bowe-Hopwood 63 x 9 over Jubjub, 2^23 leaves of 32 bytes PER GPU (weak
scaling: 2^26 on 8 GPUs), ByteDigestConverter"""
import time

from .common import HBM_PEAK_GBS, MADS_PER_PRODUCT, VALU_PEAK_WAVE_INSTR, te_counters, te_pmc


def run(env):
    args, np, torch = env.args, env.np, env.torch
    if not args.bh_merkle_log2:
        return None
    from crypto_primitives_amd import params as cparams
    from crypto_primitives_amd.crh import bowe_hopwood
    per = 1 << args.bh_merkle_log2
    total = per * env.world
    gens = cparams.bowe_hopwood_generators(0xA5A50005, 63, 9)
    leaves = np.random.default_rng(0xA5A50005 + env.rank).integers(0, 256, size=(per, 32), dtype=np.uint8)
    d_leaves = torch.from_numpy(leaves).to(env.dev)
    from crypto_primitives_amd._lib import TABLE_BUDGET_DEVICE
    budget_before = env.ctx.table_budget_setting

    def fresh(budget, g):
        env.ctx.set_table_budget(budget)
        try:
            Bp = bowe_hopwood.Parameters(g)
            return Bp, env.GpuTeBackend(Bp, Bp, device=env.dev)
        finally:
            env.ctx.set_table_budget(budget_before)

    def warm_allocator(leaves_):
        """the node arrays of one tree, allocated and handed back to torch's caching allocator: the timed builds below then reuse those
        blocks.  A hipMalloc issued while the driver is still mapping a 22 GB table allocated a moment ago (the background build) takes
        tens of ms instead of 0.1 (profiles/r06_s21: first tree 59 instead of 25 ms when its node arrays are fresh allocations) -- the
        cold figures are about the library's start, not about the allocator's"""
        tmp = [torch.empty((leaves_, 4), dtype=torch.int64, device=env.dev), torch.empty((leaves_ - 1, 4), dtype=torch.int64, device=env.dev)]
        del tmp

    def one_table(budget):
        """a FRESH handle under `budget`: the first tree from nothing (tables for 32- and 64-byte nodes + scratch + RCCL warm-up + the
        tree), then the warm build time"""
        Bp, tb = fresh(budget, gens)
        assert Bp.handle(env.ctx).table_info()["wide_builds"] == 0
        warm_allocator(per)
        env.barrier()
        c0 = time.perf_counter()
        env.build_sharded(tb, d_leaves, total, env.dist)
        torch.cuda.current_stream(env.dev).synchronize()  # the launch stream: a device-wide synchronize would also wait for the background build
        cold_ms = env.max_over_ranks(time.perf_counter() - c0) * 1e3
        # a budget above the default: that first tree ran (wholly or partly) on the cache-sized tables while a thread of the library builds
        # the wide ones; keep building trees until calls use them, and note when that was
        hh0 = Bp.handle(env.ctx)
        ready_ms, trees_before = None, 0
        if budget:
            while hh0.table_info()["last_build"]["upgrade_state"] == 1 and time.perf_counter() - c0 < 60.0:
                env.build_sharded(tb, d_leaves, per, None)  # (this rank's shard as a tree of its own: no collective inside the wait)
                torch.cuda.current_stream(env.dev).synchronize()
                trees_before += 1
            ready_ms = (time.perf_counter() - c0) * 1e3
            hh0.prepare(32, compress=True)
        env.barrier()
        reps = 3
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        m0 = time.perf_counter()
        for a, b in evs:
            a.record()
            res = env.build_sharded(tb, d_leaves, total, env.dist)
            b.record()
        env.barrier()
        bsec = env.max_over_ranks(time.perf_counter() - m0) / reps
        hh = Bp.handle(env.ctx)
        return {"B": Bp, "tb": tb, "res": res, "bsec": bsec, "dev_ms": sum(a.elapsed_time(b) for a, b in evs) / reps,
                "rec": {"group": hh.info()["digit_bits_or_group"], "table_bytes": hh.info(32)["table_bytes"], "steps_leaf": hh.info(32)["steps"],
                        "steps_inner": hh.info(64)["steps"], "cold_first_tree_ms": cold_ms, "warm_seconds": bsec, "warm_leaves_per_s": total / bsec,
                        "upgrade_ready_after_ms": ready_ms, "trees_built_meanwhile": trees_before + 1 if budget else None, "last_build": hh.table_info()["last_build"]}}
    cache = one_table(0)
    hbm = hbm_error = None
    if not env.shared_gpu:
        try:  # the opt-in tables are figures BESIDE the leg's own: if they cannot be built here (memory), the leg stands without them
            hbm = one_table(TABLE_BUDGET_DEVICE)
        except Exception as exc:  # noqa: BLE001
            hbm_error = repr(exc)[:300]
    env.keepalive.append((cache, hbm))
    main = cache  # `seconds`, `leaves_per_s` and `roofline` of this leg are the LIBRARY DEFAULT's; the HBM-sized tables stand beside them
    B, res, bsec, dev_ms = main["B"], main["res"], main["bsec"], main["dev_ms"]
    tables = {"cache_sized": cache["rec"], "library_default": "cache_sized (akp_ctx_set_table_budget 0 = 320 MiB)",
              "cold_first_tree_ms_means": "fresh handles: tables for the leaf and inner-node lengths + scratch allocation + one tree of %d leaves "
                                          "per GPU, host wall clock, max over ranks; the cache-sized handle is measured first and also pays the "
                                          "context's first scratch allocation (and, at N > 1, the first collective).  With the HBM-sized budget the "
                                          "first trees run on the cache-sized tables while the wide ones are built in the background" % per,
              "headline_table": "cache_sized (the library default)"}
    if hbm_error:
        tables["hbm_sized_error"] = hbm_error
    if hbm:
        tables["hbm_sized"] = hbm["rec"]
        d_cold = (hbm["rec"]["cold_first_tree_ms"] - cache["rec"]["cold_first_tree_ms"]) / 1e3
        d_tree = cache["rec"]["warm_seconds"] - hbm["rec"]["warm_seconds"]
        tables["break_even_trees"] = 1 + max(d_cold, 0.0) / d_tree if d_tree > 0 else None
    # configs[4] as ONE tree of 2^26 leaves on one GPU, from nothing, for both tables (generators of their own: nothing is built yet)
    if env.world == 1 and not env.shared_gpu and not args.no_sweep and args.sweep_max_log2 >= 26 and args.bh_merkle_log2 >= 20:
        big = 1 << 26
        d_big = d_leaves.repeat(big // per, 1) if big > per else d_leaves[:big]
        g2 = cparams.bowe_hopwood_generators(0xA5A50105, 63, 9)
        # the context's scratch and torch's allocator at the size of a 2^26-leaf tree BEFORE the cold figures (a tree on the tables the leg
        # has already built): `cold` below is about the tables -- a multi-GB hipMalloc inside the timed tree costs whatever the driver is
        # busy with at that moment (0.17 -> 0.41 s for the cache-sized variant right after the 22 GB tables above were allocated, r06_s32)
        r0 = env.build_sharded(cache["tb"], d_big, big, None)
        torch.cuda.current_stream(env.dev).synchronize()
        del r0
        single, alive = {}, []  # (`alive`: nothing of this block is freed before both variants are measured -- a hipMalloc that follows a
        # large hipFree waits for the driver's wipe of the released memory, seconds for tens of GB: profiles/r06_s2, r06_s20)
        for name, budget in (("cache_sized", 0), ("hbm_sized", TABLE_BUDGET_DEVICE)):
            Bp, tb2 = fresh(budget, g2)
            warm_allocator(big)
            torch.cuda.synchronize(env.dev)
            c0 = time.perf_counter()
            r2 = env.build_sharded(tb2, d_big, big, None)
            torch.cuda.current_stream(env.dev).synchronize()
            cold = time.perf_counter() - c0
            if budget:
                Bp.handle(env.ctx).prepare(32, compress=True)  # (the background build of the wide tables, waited for: the warm tree uses them)
            c0 = time.perf_counter()
            r2 = env.build_sharded(tb2, d_big, big, None)
            torch.cuda.synchronize(env.dev)
            single[name] = {"cold_first_tree_s": cold, "warm_tree_s": time.perf_counter() - c0, "table_bytes": Bp.handle(env.ctx).info(32)["table_bytes"]}
            alive.append((tb2, Bp))
            del r2
        tables["single_tree_2p26_one_gpu"] = single
        env.keepalive.append(alive)
        del d_big
    h = B.handle(env.ctx)
    pmc = te_pmc(h.info(64)["table_bytes"])
    grp = h.info()["digit_bits_or_group"]
    # additions per inner node: the table steps of the 64 data bytes; the constant of the zero-padded tail (a zero chunk adds +g) is
    # folded into the remainder step when the 171 data chunks leave one (groups of 8: 21 + 1), else it is one more addition
    inner_adds = h.info(64)["steps"] + (0 if grp > 1 and 171 % grp else 1)
    bh_merkle = {"config": "BASELINE configs[4]: MerkleTree::new, Bowe-Hopwood 63x9 over Jubjub, 32-byte leaves, ByteDigestConverter",
                 "leaves": total, "leaves_per_gpu": per, "tables": tables, "seconds": bsec, "leaves_per_s": total / bsec, "scaling": "weak",
                 "hbm_table_seconds": hbm["rec"]["warm_seconds"] if hbm else None,
                 "roofline": {"bound": "hbm", "kernels": "te_accumulate_lds_kernel<1> + te_finalize_kernel<1> + te_serialize_pairs_kernel per level",
                              "algorithmic_bytes_per_leaf": 160, "device_ms_per_build": dev_ms, "achieved": 160.0 * per / (dev_ms / 1e3) / 1e9,
                              "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": 160.0 * per / (dev_ms / 1e3) / 1e9 / HBM_PEAK_GBS,
                              "table": h.info(32),
                              "traffic": te_counters("bh_32B", per, h.info(32)["steps"], pmc)["traffic"] + te_counters("bh_70B", per - 1, inner_adds, pmc)["traffic"],
                              "traffic_measured_in_this_run": False, "traffic_calibration": pmc["calibration"] + " (FETCH_SIZE x 2 holds for random 128-byte-line gathers: x2 = 1.045 x the distinct line bytes)",
                             "traffic_static_from": pmc["source"] + " (FETCH_SIZE x 2 + WRITE_SIZE of te_accumulate_lds_kernel<1> + te_finalize_kernel<1> at 2^20 x 32 B "
                                                     "for the leaf level and 2^20 x 70 B scaled to the inner nodes' table steps; levels of <= 2^14 nodes run the split kernel; "
                                                     "NOT measured in this run)",
                              "valu": {"table_steps_per_leaf_hash": h.info(32)["steps"],
                                       "table_steps_per_inner_node": inner_adds,
                                       "inner_node_note": "64 bytes of digests in a 70-byte buffer: the table steps of the 64 data bytes, the last of them a "
                                                          "remainder entry with the constant of the zero-padded tail folded in (a zero chunk adds +g); "
                                                          "%d steps if the padding is walked" % h.info(70)["steps"],
                                       "field_products_per_step": 7}}}
    rfb = bh_merkle["roofline"]
    rfb["traffic_over_algorithmic"] = rfb["traffic"] / (160.0 * per)
    rfb["hbm_bytes_moved_per_leaf"] = rfb["traffic"] / per
    rfb["moved_GBps"] = rfb["traffic"] / (dev_ms / 1e3) / 1e9
    rfb["moved_frac_of_hbm_peak"] = rfb["moved_GBps"] / HBM_PEAK_GBS
    bh_mads = (per * (rfb["valu"]["table_steps_per_leaf_hash"] * 7 + 6) + (per - 1) * (rfb["valu"]["table_steps_per_inner_node"] * 7 + 6)) * MADS_PER_PRODUCT
    rfb["valu"]["v_mad_per_s"] = bh_mads / (dev_ms / 1e3)
    rfb["valu"]["frac_of_mad_issue_peak"] = bh_mads / (dev_ms / 1e3) / (VALU_PEAK_WAVE_INSTR * 64)
    if env.rank == 0:
        from oracle import cref
        cur = cref.CurveParams(63, 9, gens)
        ln = res["leaf_nodes"].cpu().numpy().view(np.uint64).reshape(per, 4)
        nl = res["non_leaf_nodes"].cpu().numpy().view(np.uint64).reshape(per - 1, 4)
        si = np.unique(np.linspace(0, per - 1, 129).astype(np.int64))
        ok = np.array_equal(ln[si], cur.bh_crh_batch(np.ascontiguousarray(leaves[si]), len(si), 32, threads=env.ora_threads))
        # inner nodes from their children: buffer = LE(left) || LE(right) zero-padded to (63 * 9) / 8 = 70 bytes
        ni = np.unique(np.concatenate([np.arange(0, min(32, per - 1)), np.linspace(0, per - 2, 97).astype(np.int64)]))

        def child(ix):
            return np.where((ix < per - 1)[:, None], nl[np.clip(ix, 0, per - 2)], ln[np.clip(ix - (per - 1), 0, per - 1)])
        buf = np.zeros((len(ni), 70), np.uint8)
        buf[:, :32] = cref.from_mont(np.ascontiguousarray(child(2 * ni + 1))).view(np.uint8).reshape(len(ni), 32)
        buf[:, 32:64] = cref.from_mont(np.ascontiguousarray(child(2 * ni + 2))).view(np.uint8).reshape(len(ni), 32)
        ok = ok and np.array_equal(nl[ni], cur.bh_crh_batch(buf, len(ni), 70, threads=env.ora_threads))
        bh_merkle["sampled_parity_bit_exact"] = bool(ok)
        if not ok:
            raise SystemExit("Bowe-Hopwood leg: sampled nodes differ from the oracle")
    return bh_merkle
