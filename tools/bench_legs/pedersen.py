"""`pedersen`: BASELINE configs[3] -- pedersen::CRH over Jubjub, window 4 x 256, 2^20 messages of 128 bytes per GPU, resident"""
import time

from .common import HBM_PEAK_GBS, MADS_PER_PRODUCT, VALU_PEAK_WAVE_INSTR, ClockProbe, gpu_clock_mhz, gpu_sensors, te_counters, te_pmc, valu_peak_wave_instr


def run(env):
    args, np, torch, lib, check = env.args, env.np, env.torch, env.lib, env.check
    if not args.pedersen_log2:
        return None
    from crypto_primitives_amd import params as cparams
    from crypto_primitives_amd.crh import pedersen as cped
    npd = 1 << args.pedersen_log2
    gens = cparams.pedersen_generators(0xA5A50004, 4, 256)
    msgs = np.random.default_rng(0xA5A50004 + env.rank).integers(0, 256, size=(npd, 128), dtype=np.uint8)
    d_msgs = torch.from_numpy(msgs).to(env.dev)
    d_out = torch.empty((npd, 8), dtype=torch.int64, device=env.dev)
    from crypto_primitives_amd._lib import TABLE_BUDGET_DEVICE
    budget_before = env.ctx.table_budget_setting
    state = {}

    def ped_step():
        check(lib.akp_te_crh_batch_dev(state["h"].h, d_msgs.data_ptr(), npd, 128, d_out.data_ptr(), env.stream))

    def one_table(budget):
        """a FRESH handle under `budget` (nothing else in this process holds these generators with that shape, so the table is built
        here): the first call from nothing (table build + scratch + hash), then the warm rate"""
        env.ctx.set_table_budget(budget)
        try:
            P = cped.Parameters(gens)
            state["h"] = h = P.handle(env.ctx)
        finally:
            env.ctx.set_table_budget(budget_before)
        assert h.table_info()["wide_builds"] == 0
        torch.cuda.synchronize(env.dev)
        c0 = time.perf_counter()
        ped_step()
        torch.cuda.current_stream(env.dev).synchronize()  # the launch stream: a device-wide synchronize would also wait for the background build
        cold_ms = (time.perf_counter() - c0) * 1e3
        # a budget above the default: that first batch ran on the cache-sized table while a thread of the library builds the wide one;
        # keep hashing until calls use it (what a host does), and note when that was
        ready_ms, calls_before = None, 0
        if budget:
            while h.table_info()["last_build"]["upgrade_state"] == 1 and time.perf_counter() - c0 < 60.0:
                ped_step()
                torch.cuda.current_stream(env.dev).synchronize()
                calls_before += 1
            ready_ms = (time.perf_counter() - c0) * 1e3
            h.prepare(128)
        for _ in range(3):
            ped_step()
        env.barrier()
        reps = 10
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        p0 = time.perf_counter()
        for a, b in evs:
            a.record()
            ped_step()
            b.record()
        env.barrier()
        psec = env.max_over_ranks(time.perf_counter() - p0)
        kms = sorted(a.elapsed_time(b) for a, b in evs)
        info = h.info(128)
        return {"P": P, "h": h, "reps": reps, "psec": psec, "kavg": sum(kms) / len(kms) / 1e3,
                "rec": {"digit_bits": info["digit_bits_or_group"], "table_bytes": info["table_bytes"], "steps": info["steps"],
                        "cold_first_call_ms": cold_ms, "warm_ms_per_batch": psec / reps * 1e3, "warm_hashes_per_s": npd * env.world * reps / psec,
                        "upgrade_ready_after_ms": ready_ms, "batches_hashed_on_the_cache_sized_table_meanwhile": calls_before + 1 if budget else None,
                        "last_build": h.table_info()["last_build"]}}
    # the library's default first (cache-sized table), then the HBM-sized table a host opts into -- both from nothing
    cache = one_table(0)
    hbm = runner_hbm = None
    if not env.shared_gpu:
        try:  # the opt-in table is a figure BESIDE the leg's own: if it cannot be built here (memory), the leg stands without it
            hbm = one_table(TABLE_BUDGET_DEVICE)
        except Exception as exc:  # noqa: BLE001
            runner_hbm = repr(exc)[:300]
    env.keepalive.append((cache, hbm))
    main = cache  # the leg's `hashes_per_s`, `roofline` and `sustained` are the LIBRARY DEFAULT's (the opt-in is not the headline)
    hP, reps, psec, kavg = main["h"], main["reps"], main["psec"], main["kavg"]
    state["h"] = hP
    pinfo = hP.info(128)
    psteps = pinfo["steps"]
    pmc = te_pmc(pinfo["table_bytes"])
    tc = te_counters("pedersen_128B", npd, psteps, pmc)
    tables = {"cache_sized": cache["rec"], "library_default": "cache_sized (akp_ctx_set_table_budget 0 = 320 MiB)",
              "headline_table": "cache_sized (the library default)",
              "cold_first_call_ms_means": "fresh handle: table build + scratch allocation + one batch of %d hashes, host wall clock; the cache-sized "
                                          "handle is measured first and also pays the context's first scratch allocation.  With the HBM-sized budget the "
                                          "first batches run on the cache-sized table while the wide one is built in the background "
                                          "(`upgrade_ready_after_ms`: when the handle switched)" % npd}
    if runner_hbm:
        tables["hbm_sized_error"] = runner_hbm
    if hbm:
        tables["hbm_sized"] = hbm["rec"]
        d_cold = hbm["rec"]["cold_first_call_ms"] - cache["rec"]["cold_first_call_ms"]
        d_hash = (cache["rec"]["warm_ms_per_batch"] - hbm["rec"]["warm_ms_per_batch"]) / npd
        tables["break_even_hashes"] = npd + max(d_cold, 0.0) / d_hash if d_hash > 0 else None
    pedersen = {"config": "BASELINE configs[3]: pedersen::CRH, Jubjub, window 4x256, 128-byte messages", "messages_per_gpu": npd, "tables": tables,
                "hashes_per_s": npd * env.world * reps / psec, "ms_per_batch": psec / reps * 1e3,
                "hbm_table_hashes_per_s": hbm["rec"]["warm_hashes_per_s"] if hbm else None,
                "roofline": {"bound": "hbm", "kernels": "te_accumulate_lds_kernel<2> + te_finalize_kernel<0>", "algorithmic_bytes_per_hash": 192,
                             "kernel_avg_ms": kavg * 1e3, "achieved": 192.0 * npd / kavg / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": 192.0 * npd / kavg / 1e9 / HBM_PEAK_GBS,
                             "traffic": tc["traffic"], "traffic_over_algorithmic": tc["traffic"] / (192.0 * npd),
                             "traffic_measured_in_this_run": False, "traffic_calibration": pmc["calibration"] + " (FETCH_SIZE x 2 holds for random 128-byte-line gathers: x2 = 1.045 x the distinct line bytes)",
                             "traffic_static_from": pmc["source"] + " (FETCH_SIZE x 2 + WRITE_SIZE of te_accumulate_lds_kernel<2> + te_finalize_kernel<0>; NOT measured in this run)",
                             "hbm_bytes_moved_per_hash": tc["traffic"] / npd, "moved_GBps": tc["traffic"] / kavg / 1e9,
                             "moved_frac_of_hbm_peak": tc["traffic"] / kavg / 1e9 / HBM_PEAK_GBS,
                             "table_bytes_gathered_per_hash": psteps * int(lib.akp_te_entry_bytes()),
                             "gather_over_algorithmic": psteps * int(lib.akp_te_entry_bytes()) / 192.0,
                             "table": pinfo,
                             "valu": {"table_steps_per_hash": psteps, "field_products_per_step": 7,
                                      "valu_instructions_per_hash": te_counters("pedersen_128B", 1, psteps, pmc)["valu_instr"],
                                      "v_mad_per_s": (psteps * 7 + 6) * MADS_PER_PRODUCT * npd / kavg,
                                      "frac_of_mad_issue_peak": (psteps * 7 + 6) * MADS_PER_PRODUCT * npd / kavg / (VALU_PEAK_WAVE_INSTR * 64),
                                      "v_mad_note": "7 products per table step + ~6 per hash in the shared-inversion pass, 153 multiply-adds each; peak at the nominal 2.4 GHz "
                                                    "(`sustained.frac_of_mad_issue_peak_at_effective_sclk`: at the clock the board actually holds under this kernel)",
                                      "note": "VALU-issue bound like the permutation (VALUBusy 88-95 %, 4.4-4.6 cycles per wave instruction at the effective clock: "
                                              "profiles/r04_s10, r04_s11): one mixed addition of 7 products per table step, so the lever is the NUMBER of steps -- the table "
                                              "is sized for HBM, not for the Infinity Cache (24-bit digits, 46 GB: 43 steps instead of the 64 of the 268 MB table; "
                                              "3.16 -> 2.43 ms per 2^20 hashes although a step from HBM takes 49 -> 55 us, half of that a 4 % lower clock at the power "
                                              "cap).  Message bits come from an LDS image of the workgroup's messages; one 128-byte line per table entry"}}}
    if args.sustain_seconds > 0:  # clock / power under the gather-heavy kernel (the permutation's figures are in `sustained`)
        count = int(min(2000, max(8, 0.5 * args.sustain_seconds / max(kavg, 1e-4))))
        probe = ClockProbe(env)  # effective shader clock under THIS kernel (one wave on a side stream beside every 1/8 of the loop)
        torch.cuda.synchronize(env.dev)
        s0 = time.perf_counter()
        for i in range(count):
            if i % max(1, count // 8) == max(1, count // 16):
                probe.launch()
            ped_step()
        time.sleep(min(0.25, 0.25 * count * kavg))  # sample while the queue is still draining
        cmid, sens = gpu_clock_mhz(env.local_rank), gpu_sensors()
        torch.cuda.synchronize(env.dev)
        ssec = time.perf_counter() - s0
        mhz = [p["mhz"] for p in probe.read()]
        eff = sum(mhz) / len(mhz) if mhz else None
        mads = (psteps * 7 + 6) * MADS_PER_PRODUCT * npd * count / ssec
        pedersen["sustained"] = {"launches": count, "seconds": ssec, "hashes_per_s": npd * count / ssec, "sclk_level_mhz_during": cmid,
                                 "effective_sclk_mhz": eff, "effective_sclk_mhz_min_max": [min(mhz), max(mhz)] if mhz else None,
                                 "frac_of_mad_issue_peak_at_effective_sclk": mads / (valu_peak_wave_instr(eff) * 64) if eff else None,
                                 "power_w_during": sens["power_w"], "power_cap_w": sens["power_cap_w"], "temp_c_max_during": sens["temp_c_max"],
                                 "note": "the curve kernels hold the board at its power cap: the effective clock under them is ~2.0 GHz against ~2.3 GHz under the "
                                         "permutation kernel (profiles/r04_s10/te_clock_vs_table.txt)"}
    if env.rank == 0:
        from oracle import cref
        cur = cref.CurveParams(4, 256, gens)
        si = np.unique(np.concatenate([np.arange(64), np.linspace(0, npd - 1, 193).astype(np.int64)]))
        got = d_out.cpu().numpy().view(np.uint64).reshape(npd, 2, 4)[si]
        ok = bool(np.array_equal(got, cur.pedersen_crh_batch(np.ascontiguousarray(msgs[si]), len(si), 128, threads=env.ora_threads)))
        pedersen["sampled_parity_bit_exact"] = ok
        pedersen["parity_samples"] = int(len(si))
        if not ok:
            raise SystemExit("Pedersen leg: sampled digests differ from the oracle")
    return pedersen
