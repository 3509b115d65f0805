"""`predicted_scaling`: what the design implies for the 2 / 4 / 8-GPU runs of the two tree configurations, from quantities measured
in THIS run on one GPU -- so that the first real SCALE run has numbers to be checked against (VERDICT r03 #4d).  Not a measurement.

Model of a tree of N leaves over G devices (leaf-range shards + ONE all-gather of the sub-roots):
    t(G) = t_wide(N / G) + t_narrow(N / G) + t_exchange + t_top(G)
  t_wide    levels of more than 2^15 nodes: throughput-bound, (hashes in those levels) / (hash rate measured at N on one GPU)
  t_narrow  levels of <= 2^15 nodes: each costs the latency of ONE hash launch whatever its width (profiles/r03_s7): 16 levels per device
  t_exchange the all-gather of G digests (latency-bound; taken from akp_multi_last_phases of the one-process leg when present, else 0.03 ms)
  t_top     log2 G more latency-bound levels
Strong scaling (fixed N, configs[2]): efficiency(G) = t(1) / (G t(G)).  Weak scaling (N = G x per-GPU share, configs[4]):
efficiency(G) = t(1) / t(G) with the per-device share fixed."""


def _tree_time(per_device_leaves, hash_rate, narrow_ms, exchange_ms, log2g):
    wide_hashes = 0
    width = per_device_leaves  # leaf level first, then the inner levels
    narrow_levels = 0
    while width >= 1:
        if width > (1 << 15):
            wide_hashes += width
        else:
            narrow_levels += 1
        width //= 2
    wide_s = wide_hashes / hash_rate if hash_rate > 0 else 0.0  # trees of <= 2^15 leaves per device have no wide level
    return wide_s + (narrow_levels + log2g) * narrow_ms / 1e3 + (exchange_ms / 1e3 if log2g else 0.0)


def predict(merkle, bh_merkle, narrow_poseidon_ms=0.098, narrow_bh_ms=0.092):
    out = {"model": "t(G) = wide levels at the measured one-GPU hash rate + 16 narrow levels per device + log2 G top levels at one launch latency each "
                    "+ one latency-bound all-gather; tools/bench_legs/scaling.py",
           "narrow_level_ms": {"poseidon": narrow_poseidon_ms, "bowe_hopwood": narrow_bh_ms, "source": "Poseidon: profiles/r03_s7/level_gaps.txt; Bowe-Hopwood: 73 us split kernel + 19 us pair serialisation per level, "
                                                "profiles/r04_s13/rocprof_kernel_stats_bench_py.csv"}}
    if merkle and merkle.get("seconds"):
        n = merkle["leaves"]
        leg = merkle.get("one_process_c_abi") or {}
        exch = (leg.get("phases_ms") or {}).get("allgather_ms") or 0.03
        # calibrate the wide-level hash rate so that the model reproduces THIS run's one-device time
        narrow = 16 * narrow_poseidon_ms / 1e3
        wide_hashes = sum(w for w in (n >> k for k in range(0, n.bit_length())) if w > (1 << 15))
        rate = wide_hashes / max(merkle["seconds"] - narrow, 1e-6)
        t1 = _tree_time(n, rate, narrow_poseidon_ms, exch, 0)
        out["merkle_strong"] = {"leaves": n, "calibrated_wide_hash_rate_per_s": rate, "all_gather_ms": exch}
        for g, lg in ((2, 1), (4, 2), (8, 3)):
            if n // g < 2:
                continue
            tg = _tree_time(n // g, rate, narrow_poseidon_ms, exch, lg)
            out["merkle_strong"]["%d_gpus" % g] = {"seconds": tg, "leaves_per_s": n / tg, "efficiency": t1 / (g * tg)}
    if bh_merkle and bh_merkle.get("seconds"):
        per = bh_merkle["leaves_per_gpu"]
        narrow = 16 * narrow_bh_ms / 1e3
        wide_hashes = sum(w for w in (per >> k for k in range(0, per.bit_length())) if w > (1 << 15))
        rate = wide_hashes / max(bh_merkle["seconds"] - narrow, 1e-6)
        t1 = _tree_time(per, rate, narrow_bh_ms, 0.03, 0)
        out["bh_merkle_weak"] = {"leaves_per_gpu": per, "calibrated_wide_hash_rate_per_s": rate}
        for g, lg in ((2, 1), (4, 2), (8, 3)):
            tg = _tree_time(per, rate, narrow_bh_ms, 0.03, lg)
            out["bh_merkle_weak"]["%d_gpus" % g] = {"seconds": tg, "leaves_per_s": g * per / tg, "efficiency": t1 / tg}
    out["permutation_weak"] = "the headline batch shards with no exchange: the model is efficiency 1.0 at every G (what the driver's SCALE run measures is clock / power variation between devices)"
    return out
