"""`host_path`: the host-pointer entry points (what a Rust host calls), PCIe-inclusive -- never `value`.  Wall time per call,
MEDIAN of `reps` calls (round 4: one call in ~14 of the three-stream curve-hash pipeline took +6 ms in a runtime wait, profiles/r04_s3)."""
import ctypes as C
import time


def _median_call(fn, reps):
    fn()  # warm-up (scratch, streams)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2], ts[0], ts[-1]


def _form_note(lib):
    """which form the context settled on for the pinned batches just timed: the library announces its choice in akp_last_error() (round 6)"""
    note = (lib.akp_last_error() or b"").decode(errors="replace")
    return note if note.startswith("note: the pinned") else "no note (AKP_TE_PINNED_FORM pins the form, or the gate is unavailable)"


def run(env, host_states, n):
    args, np, lib, check = env.args, env.np, env.lib, env.check
    if args.no_host_path or env.rank != 0:
        return None
    ph = env.ph
    host_path = {"states": n, "pcie_bound_note": "96 B in + 96 B out per permutation; PCIe Gen5 x16 is 63 GB/s per direction (spec)",
                 "statistic": "median wall time of the calls after one warm-up call (min and max beside it)"}
    work = host_states.copy()
    for label in ("pageable", "pinned"):
        if label == "pinned":
            pp = C.c_void_p()
            check(lib.akp_host_alloc(work.nbytes, C.byref(pp)))
            arr = np.ctypeslib.as_array((C.c_uint64 * (work.size)).from_address(pp.value))
            arr[:] = host_states.reshape(-1)
            ptr = pp
        else:
            ptr = work.ctypes.data
        hs, lo, hi = _median_call(lambda: check(lib.akp_poseidon_permute_batch(ph.h, ptr, n)), 7)
        host_path[label] = {"permutations_per_s": n / hs, "ms_per_batch": hs * 1e3, "ms_min": lo * 1e3, "ms_max": hi * 1e3, "GBps_each_direction": 96.0 * n / hs / 1e9,
                            "mode": "zero copy: the kernel addresses the pinned buffer over PCIe" if label == "pinned"
                                    else "chunked copy-in / kernel / copy-out on three streams (runtime-staged copies)"}
        if label == "pinned":
            check(lib.akp_host_free(pp))
    if args.pedersen_log2:  # config 4 through the host-pointer entry point: 128 B in, 64 B out per hash
        from crypto_primitives_amd import params as cparams2
        from crypto_primitives_amd.crh import pedersen as cped2
        nph = 1 << args.pedersen_log2
        hPh = cped2.Parameters(cparams2.pedersen_generators(0xA5A50004, 4, 256)).handle(env.ctx)
        hm = np.random.default_rng(0xA5A50014).integers(0, 256, size=(nph, 128), dtype=np.uint8)
        ho = np.empty((nph, 8), dtype=np.uint64)
        hs, lo, hi = _median_call(lambda: check(lib.akp_te_crh_batch(hPh.h, hm.ctypes.data, nph, 128, ho.ctypes.data)), 9)
        host_path["pedersen_pageable"] = {"hashes_per_s": nph / hs, "ms_per_batch": hs * 1e3, "ms_min": lo * 1e3, "ms_max": hi * 1e3, "GBps_in": 128.0 * nph / hs / 1e9,
                                          "GBps_out": 64.0 * nph / hs / 1e9,
                                          "mode": "double-buffered chunks of 2^17 messages: copy-in / kernels / copy-out on three streams (runtime-staged copies)"}
        # the same with pinned buffers (akp_host_alloc) on both sides: asynchronous DMA in and out, kernels back to back
        pm, po = C.c_void_p(), C.c_void_p()
        check(lib.akp_host_alloc(hm.nbytes, C.byref(pm)))
        check(lib.akp_host_alloc(ho.nbytes, C.byref(po)))
        np.ctypeslib.as_array((C.c_uint8 * hm.size).from_address(pm.value))[:] = hm.reshape(-1)
        hs2, lo2, hi2 = _median_call(lambda: check(lib.akp_te_crh_batch(hPh.h, pm, nph, 128, po)), 19)
        pinned_out = np.ctypeslib.as_array((C.c_uint64 * ho.size).from_address(po.value)).reshape(ho.shape)
        same = bool(np.array_equal(pinned_out, ho))
        host_path["pedersen_pinned"] = {"hashes_per_s": nph / hs2, "ms_per_batch": hs2 * 1e3, "ms_min": lo2 * 1e3, "ms_max": hi2 * 1e3, "GBps_in": 128.0 * nph / hs2 / 1e9,
                                        "GBps_out": 64.0 * nph / hs2 / 1e9, "digests_equal_the_pageable_call": same,
                                        "mode": "pinned buffers on both sides; the context measures which form is faster where the runtime put its streams (round 6, "
                                                "profiles/r06_s41 ... s46) -- ONE gated launch (round 5: DMA copy-in in chunks of up to 2^17 messages issued up front with an "
                                                "arrival flag behind each, the accumulate kernel launched once over the whole batch, workgroups wait on their chunk's flag, "
                                                "finish their digests themselves, copy-out of a chunk released by the host thread as its workgroups report) or round 4's "
                                                "chunked launches -- and keeps it: of the 20 calls of this leg four go to either form (in turns), twelve to the faster one"}
        host_path["pedersen_pinned"]["form"] = _form_note(lib)
        if not env.shared_gpu:  # the same pinned call with the HBM-sized table (opt-in budget; prepared first, not built in the background beside the calls)
            from crypto_primitives_amd._lib import TABLE_BUDGET_DEVICE
            env.ctx.set_table_budget(TABLE_BUDGET_DEVICE)
            try:
                Pw = cped2.Parameters(cparams2.pedersen_generators(0xA5A50004, 4, 256))
                hw = Pw.handle(env.ctx)
            finally:
                env.ctx.set_table_budget(0)
            hw.prepare(128)
            hs3, lo3, hi3 = _median_call(lambda: check(lib.akp_te_crh_batch(hw.h, pm, nph, 128, po)), 19)
            host_path["pedersen_pinned_hbm_table"] = {"hashes_per_s": nph / hs3, "ms_per_batch": hs3 * 1e3, "ms_min": lo3 * 1e3, "ms_max": hi3 * 1e3,
                                                      "digests_equal_the_pageable_call": bool(np.array_equal(pinned_out, ho)), "table": hw.info(128)}
            host_path["pedersen_pinned_hbm_table"]["form"] = _form_note(lib)
            same = same and host_path["pedersen_pinned_hbm_table"]["digests_equal_the_pageable_call"]
            env.keepalive.append((hw, Pw))
        check(lib.akp_host_free(pm))
        check(lib.akp_host_free(po))
        if not same:
            raise SystemExit("host path: the pinned Pedersen call differs from the pageable one")
    if args.merkle_log2:
        ntree = 1 << min(args.merkle_log2, 22)
        lv = env.field.random_fr(ntree, seed=0xA5A50013).reshape(ntree, 1, 4)
        root = np.empty(4, np.uint64)
        check(lib.akp_merkle_build_poseidon(ph.h, ph.h, lv.ctypes.data, ntree, 1, None, None, root.ctypes.data))
        h0 = time.perf_counter()
        check(lib.akp_merkle_build_poseidon(ph.h, ph.h, lv.ctypes.data, ntree, 1, None, None, root.ctypes.data))
        host_path["merkle_root_only"] = {"leaves": ntree, "seconds": time.perf_counter() - h0}
    return host_path
