"""`cpu_baseline`: the oracle's reference-shaped C restatement (oracle/c/akp_oracle.c, "kind": "port") timed on this box's host
cores: the permutation over a bounded sample of the same states, and the Pedersen / Bowe-Hopwood-tree legs beside it"""
import time

from .common import cpu_info


def run(env, out, host_states, n, value, pedersen, bh_merkle):
    args, np = env.args, env.np
    from oracle import cref
    ora = env.ora
    hw = cref.hardware_threads()
    info = cpu_info()
    # what this process can really use: affinity mask, cgroup CPU quota and hardware threads, whichever is smallest
    bounds = {"hardware threads": hw, "affinity mask": info["affinity_cpus"], "cgroup quota": info["cgroup_cpu_quota"]}
    basis, eff_cores = min(((k, v) for k, v in bounds.items() if v), key=lambda kv: kv[1])
    cands = sorted({hw, max(1, hw // 2), max(1, hw // 4), max(1, hw // 8), min(hw, 16), min(hw, 8), max(1, int(round(eff_cores)))}, reverse=True)

    def cpu_leg(run_, total, seconds, cal, quant=lambda k: k):
        """`run_(k, threads)` processes the first k items; returns (rate at the best thread count, threads, 1-thread rate, items
        timed).  Whole passes over min(total, rate * seconds) items until `seconds` have been spent."""
        c0 = time.perf_counter()
        run_(quant(max(2, cal // 8)), 1)
        rate1 = quant(max(2, cal // 8)) / (time.perf_counter() - c0)
        best = (rate1, 1)
        for cand in cands:
            c0 = time.perf_counter()
            run_(cal, cand)
            r = cal / (time.perf_counter() - c0)
            if r > best[0]:
                best = (r, cand)
        rate, threads = best
        sample = quant(int(min(total, max(cal, rate * seconds))))
        passes = 0
        c0 = time.perf_counter()
        while True:
            run_(sample, threads)
            passes += 1
            cpu_s = time.perf_counter() - c0
            if sample < total or cpu_s >= seconds:
                break
        return sample * passes / cpu_s, threads, rate1, sample * passes
    rate_n, threads, rate1, sample = cpu_leg(lambda k, th: ora.permute_batch(host_states[:k], threads=th), n, args.cpu_seconds, 8192)
    out["cpu_baseline"] = {"value": rate_n, "unit": "permutations/s", "cores": eff_cores, "cores_basis": basis, "kind": "port",
                           "threads_used": threads, "rate_1_thread": rate1, "effective_cores": rate_n / rate1,
                           "hardware_threads": hw, **info,
                           "sample_short": "%d permutations of the timed 2^%d states, oracle/c/akp_oracle.c (reference-shaped port), %d pthreads" % (sample, args.log2_states, threads),
                           "sample": "%d permutations over the same 2^%d states, reference-shaped C restatement (oracle/c/akp_oracle.c: "
                                     "dense MDS, square-and-multiply, one permutation per call as the reference), %d pthreads (best of a "
                                     "thread-count sweep); `cores` = min(affinity, cgroup quota, hardware threads) = what this container "
                                     "may use; `effective_cores` = that rate / the 1-thread rate: what it really delivered" % (sample, args.log2_states, threads)}
    out["gpu_over_cpu"] = value / rate_n
    curve_seconds = max(2.0, args.cpu_seconds / 3.0)
    if pedersen:  # BASELINE configs[3] on the CPU: bit-by-bit conditional additions as crh/pedersen/mod.rs:112-124
        from crypto_primitives_amd import params as cparams3
        cur = cref.CurveParams(4, 256, cparams3.pedersen_generators(0xA5A50004, 4, 256))
        cm = np.random.default_rng(0xA5A50004).integers(0, 256, size=(1 << 16, 128), dtype=np.uint8)
        r_n, th, r1, smp = cpu_leg(lambda k, t_: cur.pedersen_crh_batch(cm[:k], k, 128, threads=t_), len(cm), curve_seconds, 2048)
        out["cpu_baseline"]["pedersen"] = {"value": r_n, "unit": "hashes/s", "threads_used": th, "rate_1_thread": r1, "effective_cores": r_n / r1,
                                           "sample": "%d Pedersen 4x256 hashes of 128-byte messages (orc_pedersen_crh_batch)" % smp,
                                           "gpu_over_cpu": pedersen["hashes_per_s"] / r_n}
    if bh_merkle:  # BASELINE configs[4] on the CPU: the whole tree (leaf hashes + inner levels, barrier per level), 2^k leaves
        from crypto_primitives_amd import params as cparams4
        curb = cref.CurveParams(63, 9, cparams4.bowe_hopwood_generators(0xA5A50005, 63, 9))
        cl = np.random.default_rng(0xA5A50005).integers(0, 256, size=(1 << 16, 32), dtype=np.uint8)

        def pow2(k):
            return 1 << max(1, int(k).bit_length() - 1)  # the largest power of two <= k (a tree needs one)
        r_n, th, r1, smp = cpu_leg(lambda k, t_: curb.merkle_build(1, curb, cl[:k], k, 32, threads=t_), len(cl), curve_seconds, 2048, pow2)
        out["cpu_baseline"]["bh_merkle"] = {"value": r_n, "unit": "leaves/s", "threads_used": th, "rate_1_thread": r1, "effective_cores": r_n / r1,
                                            "sample": "Bowe-Hopwood 63x9 trees over %d leaves of 32 bytes in total (orc_curve_merkle_build, power-of-two trees)" % smp,
                                            "gpu_over_cpu": bh_merkle["leaves_per_s"] / r_n}
