"""The ONE line bench.py prints, cut from the full record.

The driver keeps the tail of stdout and parses its last line; round 4's 22 KB line fell out of that window (BENCH_r04.parsed == null).
`compact(full)` keeps the contract keys, the `roofline` and `cpu_baseline` objects with scalar fields only, and ONE scalar per side
leg; everything else (sweep, predicted_scaling, proofs, host_path, the clock-probe method text, per-leg roofline blocks) stays in
`bench_full.json`, which the line names under "full".  `LIMIT` is asserted here and in tests/ (CPU: on the committed full records
under profiles/; GPU: on the live line)."""
import json

LIMIT = 6000  # bytes; the driver's window lies between 8 KB (lost) and 15.7 KB (parsed) -- stay well inside the smaller


def _get(d, *path, default=None):
    for k in path:
        if not isinstance(d, dict) or k not in d or d[k] is None:
            return default
        d = d[k]
    return d


def _r(x, digits=6):
    """floats to `digits` significant digits: the line is for reading and for the driver's consistency check, the full record keeps
    every bit"""
    if isinstance(x, float):
        return float("%.*g" % (digits, x))
    return x


def compact(full, full_name="bench_full.json"):
    rf = full["roofline"]
    valu = rf.get("valu") or {}
    line = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline")}
    line["dtype"] = full["dtype"].split(" ")[0]  # "i32": 9 x 29-bit limbs in 32-bit registers, 64-bit multiply-add accumulation (full record: dtype_detail)
    line["data"] = full["data"]
    line["config"] = {k: full["config"][k] for k in ("workload", "states_per_gpu", "parallelism")}
    par = full["parity"]
    line["parity"] = {"bit_exact": par["bit_exact"], "states_checked": par["timed_buffer_states_checked"], "of_states": par.get("of_states"), "kernel": par["probe_kernel"]}
    line["curve_parity"] = full["curve_parity"].split(":")[0][:160]
    line["curve_parity_emitter"] = (full.get("curve_parity_emitter") or "")[:120]
    line["roofline"] = {"bound": rf["bound"], "kernel": rf["kernel"], "achieved": _r(rf["achieved"]), "peak": rf["peak"], "unit": rf["unit"],
                        "frac": _r(rf["achieved"] / rf["peak"]), "traffic": rf.get("traffic"), "traffic_measured_in_this_run": bool(rf.get("traffic_measured_in_this_run", False)),
                        "traffic_source": rf.get("traffic_source"),
                        "kernel_avg_ms": _r(rf["kernel_avg_ms"]), "algorithmic_bytes_per_launch": rf["algorithmic_bytes_per_launch"],
                        "effective_sclk_mhz": _r(rf.get("effective_sclk_mhz")), "frac_of_mad_issue_peak": _r(valu.get("frac_of_mad_issue_peak"))}
    cb = full.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {"value": _r(cb["value"]), "unit": cb["unit"], "cores": cb["cores"], "cores_basis": cb.get("cores_basis"),
                                "kind": cb["kind"], "threads_used": cb.get("threads_used"), "cpu_model": cb.get("cpu_model"),
                                "sample": (cb.get("sample_short") or cb.get("sample", ""))[:200]}
        line["gpu_over_cpu"] = _r(full.get("gpu_over_cpu"))
    else:
        line["cpu_baseline"] = None  # N > 1 (rank 0 at N = 1 only) or --no-cpu-baseline
    # ---- one scalar per side leg (absent leg -> null) --------------------------------------------------------------------------
    scal = {
        "merkle_leaves": _get(full, "merkle", "leaves"),
        "merkle_s": _get(full, "merkle", "seconds"),                       # configs[2]: 2^24 Poseidon leaves, all ranks
        "merkle_leaves_per_s": _get(full, "merkle", "leaves_per_s"),
        "merkle_hbm_frac": _get(full, "merkle", "hbm_frac"),
        "pedersen_hashes_per_s": _get(full, "pedersen", "hashes_per_s"),   # configs[3], warm, the library's default (cache-sized) table
        "pedersen_hbm_table_hashes_per_s": _get(full, "pedersen", "tables", "hbm_sized", "warm_hashes_per_s"),   # opt-in: AKP_TABLE_BUDGET_DEVICE
        "pedersen_hbm_frac": _get(full, "pedersen", "roofline", "frac"),
        "pedersen_moved_frac_of_hbm_peak": _get(full, "pedersen", "roofline", "moved_frac_of_hbm_peak"),  # counter traffic (calibrated, profiles/r05_s6) / 8 TB/s
        "pedersen_cold_first_call_ms": _get(full, "pedersen", "tables", "cache_sized", "cold_first_call_ms"),          # library default, from nothing
        "pedersen_cold_first_call_ms_hbm_table": _get(full, "pedersen", "tables", "hbm_sized", "cold_first_call_ms"),
        "pedersen_hbm_table_ready_after_ms": _get(full, "pedersen", "tables", "hbm_sized", "upgrade_ready_after_ms"),
        "bh_leaves": _get(full, "bh_merkle", "leaves"),
        "bh_s": _get(full, "bh_merkle", "seconds"),                        # configs[4] share: 2^23 leaves per GPU, warm, the library's default tables
        "bh_hbm_table_s": _get(full, "bh_merkle", "tables", "hbm_sized", "warm_seconds"),                               # opt-in
        "bh_leaves_per_s": _get(full, "bh_merkle", "leaves_per_s"),
        "bh_hbm_frac": _get(full, "bh_merkle", "roofline", "frac"),
        "bh_moved_frac_of_hbm_peak": _get(full, "bh_merkle", "roofline", "moved_frac_of_hbm_peak"),
        "bh_cold_first_tree_ms": _get(full, "bh_merkle", "tables", "cache_sized", "cold_first_tree_ms"),                # library default, from nothing
        "bh_cold_first_tree_ms_hbm_table": _get(full, "bh_merkle", "tables", "hbm_sized", "cold_first_tree_ms"),
        "bh_hbm_table_ready_after_ms": _get(full, "bh_merkle", "tables", "hbm_sized", "upgrade_ready_after_ms"),
        "bh_2p26_s": _get(full, "sweep", "points", "2^26", "bh_tree_ms"),                                               # ONE GPU, warm, default tables
        "bh_2p26_cold_s": _get(full, "bh_merkle", "tables", "single_tree_2p26_one_gpu", "cache_sized", "cold_first_tree_s"),
        "bh_2p26_cold_s_hbm_table": _get(full, "bh_merkle", "tables", "single_tree_2p26_one_gpu", "hbm_sized", "cold_first_tree_s"),
        "ragged_bh_hashes_per_s": _get(full, "ragged", "bowe_hopwood_63x9", "hashes_per_s"),          # 2^20 items of 0 .. 64 bytes, one launch
        "ragged_pedersen_hashes_per_s": _get(full, "ragged", "pedersen_4x256", "hashes_per_s"),
        "ragged_poseidon_hashes_per_s": _get(full, "ragged", "poseidon_rate2", "hashes_per_s"),
        "verify_paths_hashes_per_s": _get(full, "proofs", "poseidon", "verify_all_leaves_dev", "hashes_per_s_device"),
        "multi_proof_proofs_per_s": _get(full, "proofs", "poseidon", "generate_multi_proof", "proofs_per_s"),
        "update_2p10_leaves_per_s": _get(full, "proofs", "poseidon", "update_batch", "2^10", "leaves_per_s"),
        "host_pinned_perm_per_s": _get(full, "host_path", "pinned", "permutations_per_s"),
        "host_pageable_perm_per_s": _get(full, "host_path", "pageable", "permutations_per_s"),
        "pedersen_pinned_hashes_per_s": _get(full, "host_path", "pedersen_pinned", "hashes_per_s"),                  # default table
        "pedersen_pinned_hbm_table_hashes_per_s": _get(full, "host_path", "pedersen_pinned_hbm_table", "hashes_per_s"),
        "sustained_perm_per_s": _get(full, "sustained", "2^%d" % (full["config"]["states_per_gpu"].bit_length() - 1), "permutations_per_s"),
        "cpu_pedersen_hashes_per_s": _get(full, "cpu_baseline", "pedersen", "value"),
        "cpu_bh_leaves_per_s": _get(full, "cpu_baseline", "bh_merkle", "value"),
        "predicted_8gpu_merkle_s": _get(full, "predicted_scaling", "merkle_strong", "8_gpus", "seconds"),
        "predicted_8gpu_bh_s": _get(full, "predicted_scaling", "bh_merkle_weak", "8_gpus", "seconds"),
        "predicted_Ngpu_merkle_s": _get(full, "predicted_scaling", "merkle_strong", "predicted", "seconds"),
        "predicted_Ngpu_bh_s": _get(full, "predicted_scaling", "bh_merkle_weak", "predicted", "seconds"),
    }
    if scal["bh_2p26_s"] is not None:
        scal["bh_2p26_s"] /= 1e3
    line["legs"] = {k: _r(v) for k, v in scal.items() if v is not None}
    errs = full.get("leg_errors") or {}
    line["legs_failed"] = sorted(errs)  # a failed side leg does not void the headline: it is named here, its scalars are absent above
    if errs:
        line["leg_errors"] = {k: str(v)[:160] for k, v in sorted(errs.items())[:8]}
    line["launch"] = {"ranks": _get(full, "launch", "ranks"), "backend": _get(full, "launch", "backend")}
    line["full"] = full_name
    s = json.dumps(line)
    if len(s) >= LIMIT:  # cannot happen with the fields above; a loud failure beats a line the driver drops
        raise RuntimeError("bench line is %d bytes (limit %d)" % (len(s), LIMIT))
    return line
