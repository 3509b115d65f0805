"""`sustained`: the headline launch looped for seconds at 2^20 and 2^24 states (rank 0's device): rate, launch-time spread, clock
level, power and temperature while the queue drains"""
import time

from .common import gpu_clock_mhz, gpu_sensors


def run(env, d_states, n, kern_avg_s):
    args, np, torch = env.args, env.np, env.torch
    if not (args.sustain_seconds > 0 and env.rank == 0):
        return None
    t = env.cfg.t
    sustained = {}
    for lg in sorted({args.log2_states, args.sustain_log2_big} - {0}):
        ns = 1 << lg
        if ns == n:
            buf = d_states
        else:
            try:
                buf = torch.from_numpy(env.field.random_fr(min(ns, 1 << 20) * t, seed=0xA5A50012).reshape(-1, t, 4).view(np.int64)).to(env.dev).repeat(max(1, ns >> 20), 1, 1)
            except Exception:
                continue
        per_launch = max(kern_avg_s * ns / n, 1e-4)
        count = int(min(4000, max(8, args.sustain_seconds / per_launch)))
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(count)]
        c0 = gpu_clock_mhz(env.local_rank)
        torch.cuda.synchronize(env.dev)
        s0 = time.perf_counter()
        for a, b in evs:
            a.record()
            env.check(env.lib.akp_poseidon_permute_batch_dev(env.ph.h, buf.data_ptr(), ns, env.stream))
            b.record()
        cmid, sens = gpu_clock_mhz(env.local_rank), gpu_sensors()
        torch.cuda.synchronize(env.dev)
        secs = time.perf_counter() - s0
        ms = sorted(a.elapsed_time(b) for a, b in evs)
        sustained["2^%d" % lg] = {"launches": count, "seconds": secs, "permutations_per_s": ns * count / secs,
                                  "launch_ms_min": ms[0], "launch_ms_median": ms[len(ms) // 2], "launch_ms_max": ms[-1],
                                  "sclk_level_mhz_before": c0, "sclk_level_mhz_during": cmid, "sclk_level_mhz_after": gpu_clock_mhz(env.local_rank),
                                  "power_w_during": sens["power_w"], "power_cap_w": sens["power_cap_w"], "temp_c_max_during": sens["temp_c_max"]}
        if buf is not d_states:
            del buf
    return sustained
