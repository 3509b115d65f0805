"""constants, sensors and plumbing shared by bench.py and its legs"""
import glob
import json
import os
import subprocess

ALGO_BYTES_PER_PERM = 192        # 96 B read + 96 B write (t = 3)
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec
MODMUL_PER_PERM_REF = 626        # reference-shaped count
# HBM bytes per permutation from PMC passes of an earlier session of THIS round (profiles/r06_s25 = r05_s7 to four digits; NOT measured in this run):
# (2 * 49 585.3 KB + 98 304 KB) * 1024 per 2^20 permutations = 192.8 B  (algorithmic: 192 B); FETCH_SIZE 49 582.8 / 49 587.8 KB,
# WRITE_SIZE 98 304 KB, SQ_INSTS_VALU 1 224 736 768, VALUBusy 93.6-93.9 % (tools/gpu_pmc_r4.sh; round 3: the same to four digits)
PMC_TRAFFIC_BYTES_PER_PERM = (2 * 49584.7 + 98304.0) * 1024 / (1 << 20)
PMC_TRAFFIC_SOURCE = "profiles/r06_s25/pmc_poseidon.txt"
NOMINAL_SCLK_MHZ = 2400.0
CYCLES_PER_WAVE_MAD = 4.0        # one v_mad (wave64) per ~4 cycles per SIMD (profiles/r01_s1_microbench*)
SIMDS = 256 * 4


def valu_peak_wave_instr(sclk_mhz=NOMINAL_SCLK_MHZ):
    """wave-level v_mad instructions per second the chip can issue at `sclk_mhz`"""
    return SIMDS * sclk_mhz * 1e6 / CYCLES_PER_WAVE_MAD


VALU_PEAK_WAVE_INSTR = valu_peak_wave_instr()
# curve-hash kernels, per 2^20-hash launch, KB / instructions (profiles/r06_s25/pmc_te_hbm.txt = r05_s7: rocprofv3 --pmc, one counter per pass, the
# kernels with the HBM-sized tables -- 24-bit Pedersen digits, 8-chunk Bowe-Hopwood groups + remainder step; NOT measured
# in this run).  `steps` is the table-step count of that launch: a run whose handles got another shape (smaller table budget)
# scales the gather-proportional parts by its own step count.  FETCH x 2 as MI355X_MICROARCH.md prescribes for gfx950 (128-byte
# requests tallied at 64 B): with it the accumulate kernels fetch ~1.0 x the table bytes they gather -- every table line comes
# from HBM, the L2 only serves the second half of a line.  (The 268 MB / 237 MB tables of rounds 1-3: profiles/r04_s8/pmc_te.txt.)
PMC_TE = {"source": "profiles/r06_s25/pmc_te_hbm.txt",
          # FETCH_SIZE x 2 for RANDOM 128-byte-line gathers: calibrated in profiles/r05_s6 (x2 = 1.045 x the distinct line bytes; every
          # fabric request is a whole line tallied at 64 B, as for streaming reads)
          "calibration": "profiles/r05_s6/README.md",
          "pedersen_128B": {"fetch_kb": 2882390 + 164740, "write_kb": 147560 + 114860, "valu_instr": 1040370000 + 48292900, "steps": 43},  # accumulate_lds<2> + finalize<0>
          "bh_32B": {"fetch_kb": 734823 + 164800, "write_kb": 147483 + 81920, "valu_instr": 257327000 + 39591900, "steps": 11},          # accumulate_lds<1> + finalize<1>
          "bh_70B": {"fetch_kb": 1547435 + 164780, "write_kb": 147495 + 81920, "valu_instr": 583819000 + 39591900, "steps": 24}}
MADS_PER_PRODUCT = 153           # multiply-adds of one field product (81 limb products + 72 reduction products)
MADS_PER_PERM = 55 * (4 * 117 + 153) + (20 * 234 + 4 * 315) + (30 * (315 + 153) + (234 + 153))  # multiply-adds per permutation
# (45 / 81 / 162 / 243 limb products + 72 reduction products for a square / product / 2-term / 3-term dot) in the full form:
# 55 S-boxes; full-round rows: 20 with unit diagonal (dot2), 4 dot3; partial rounds: dot3 + one product (lane-1 form), the
# last one dot2 + one product; no conversion products.


# the same kernels with the library's DEFAULT tables (cache-sized: 16-bit Pedersen digits, 64 steps; Bowe-Hopwood groups of 5: 18 steps per
# 32-byte leaf, 36 per inner node) -- profiles/r06_s25/pmc_te.txt (= r05_s7).  FETCH_SIZE counts the Infinity Cache's hits as well: for these tables
# `traffic` is what crosses the fabric into the L2, most of it served by the 256 MiB cache, not by HBM.
PMC_TE_DEFAULT = {"source": "profiles/r06_s25/pmc_te.txt", "calibration": PMC_TE["calibration"],
                  "pedersen_128B": {"fetch_kb": 3777575 + 164640, "write_kb": 147490 + 114832, "valu_instr": 1545830000 + 48292900, "steps": 64},
                  "bh_32B": {"fetch_kb": 883960 + 163455, "write_kb": 147576 + 81920, "valu_instr": 426050000 + 39591900, "steps": 18},
                  "bh_70B": {"fetch_kb": 1937305 + 164866, "write_kb": 147525 + 81920, "valu_instr": 921133000 + 39591900, "steps": 36}}


def te_pmc(table_bytes):
    """the PMC record that belongs to a handle's table: the HBM-sized one above 1 GiB, else the cache-sized default"""
    return PMC_TE if table_bytes and table_bytes > (1 << 30) else PMC_TE_DEFAULT


def te_counters(key, hashes, steps=None, pmc=None):
    """bytes / instructions for `hashes` hashes from the per-2^20 PMC figures of `pmc` (PMC_TE: HBM-sized tables, the default here for the
    callers of round 5; PMC_TE_DEFAULT: cache-sized); the gather-proportional parts are scaled to `steps` table steps per hash when the
    run's table shape differs from the profiled one"""
    c = (pmc or PMC_TE)[key]
    steps_scale = 1.0 if steps is None else steps / float(c["steps"])
    per = hashes / float(1 << 20)
    return {"traffic": (2.0 * c["fetch_kb"] * steps_scale + c["write_kb"]) * 1024.0 * per, "valu_instr": c["valu_instr"] * steps_scale * per}


class Env:
    """what a leg needs: modules (torch, np, cpa, field, lib, check), the device / context / stream, the default Poseidon config and
    its handle, rank plumbing (rank, world, local_rank, dist, shared_gpu, barrier(), max_over_ranks()), the oracle handle of rank 0
    (`ora`, checker only) and the parsed arguments"""


def _card_dirs():
    return sorted(glob.glob("/sys/class/drm/card*/device"))


def gpu_clock_mhz(index=0):
    """shader clock LEVEL of the current power state: the entry pp_dpm_sclk marks with '*' (highest over the cards that expose one
    -- a box may list an idle integrated device first), else `rocm-smi --showclocks --json`; None when unreadable.  A ceiling, not
    the effective clock: that is `roofline.effective_sclk_mhz` (akp_clock_probe_dev)."""
    best = None
    try:
        for d in _card_dirs():
            path = os.path.join(d, "pp_dpm_sclk")
            if not os.path.exists(path):
                continue
            for line in open(path).read().splitlines():
                if line.strip().endswith("*"):
                    v = float(line.split(":")[1].strip().split("M")[0])
                    best = v if best is None else max(best, v)
    except Exception:
        pass
    if best is not None and best > 200.0:
        return best
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--json"], capture_output=True, text=True, timeout=10).stdout
        vals = []
        for card in json.loads(out).values():
            for k, v in card.items():
                if "sclk" in k.lower() and "(" in str(v):
                    vals.append(float(str(v).split("(")[1].split("M")[0]))
        if vals:
            return max(vals + ([best] if best else []))
    except Exception:
        pass
    return best


def gpu_sensors():
    """socket power (W), its cap (W) and the hottest temperature (C) of the BUSIEST card, from hwmon: power1_input (what the MI355X
    boxes expose, label PPT; round 3 only looked for power1_average and reported None everywhere) or power1_average, power1_cap,
    temp*_input.  A box shows every GPU of its node; the one this process loads is the one that draws the most.  Values None when
    unreadable."""
    best = None
    for d in _card_dirs():
        for hw in glob.glob(os.path.join(d, "hwmon", "hwmon*")):
            def rd(name, scale):
                try:
                    return float(open(os.path.join(hw, name)).read().strip()) / scale
                except Exception:
                    return None
            w = rd("power1_input", 1e6)
            if w is None:
                w = rd("power1_average", 1e6)
            if w is None:
                continue
            temps = [t for t in (rd(os.path.basename(p), 1e3) for p in glob.glob(os.path.join(hw, "temp*_input"))) if t is not None]
            rec = {"power_w": w, "power_cap_w": rd("power1_cap", 1e6), "temp_c_max": max(temps) if temps else None, "source": "hwmon " + os.path.basename(hw)}
            if best is None or w > best["power_w"]:
                best = rec
    return best or {"power_w": None, "power_cap_w": None, "temp_c_max": None, "source": None}


def gpu_power_w():
    return gpu_sensors()["power_w"]


def cpu_info():
    model = None
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if q == "max" else float(q) / float(per)
    except Exception:
        pass
    try:
        aff = len(os.sched_getaffinity(0))
    except Exception:
        aff = None
    return {"cpu_model": model, "logical_cpus": os.cpu_count(), "affinity_cpus": aff, "cgroup_cpu_quota": quota}


def measure_hbm_copy(torch, dev, nbytes=1 << 30, reps=10):
    """Read+write GB/s of a plain 1 GiB device-to-device copy on this box (the measured HBM rate beside the
    vendor 8 TB/s).  Measurement plumbing only -- not part of the hashed path."""
    try:
        a = torch.empty(nbytes // 8, dtype=torch.int64, device=dev)
        b = torch.empty_like(a)
        a.zero_()
        for _ in range(2):
            b.copy_(a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            b.copy_(a)
        e1.record()
        torch.cuda.synchronize()
        secs = e0.elapsed_time(e1) / 1e3 / reps
        del a, b
        return 2 * nbytes / secs / 1e9
    except Exception:  # pragma: no cover - measurement is best effort
        return None


class ClockProbe:
    """effective shader clock through akp_clock_probe_dev: one wave, a chain of dependent multiply-adds, shader-clock cycles
    against the constant 100 MHz clock.  `launch()` enqueues a probe on its own stream (non-blocking: it may run BESIDE the
    timed kernels, one wave among ~16 000); `read()` returns the list of MHz values of the probes that have finished."""

    def __init__(self, env, chain_len=1 << 20, slots=32):
        self.env, self.chain_len = env, chain_len
        self.stream = env.torch.cuda.Stream(device=env.dev)
        self.pool = env.torch.zeros((slots, 3), dtype=env.torch.int64, device=env.dev)  # allocated and cleared BEFORE any timed region
        env.torch.cuda.synchronize(env.dev)
        self.used = 0
        self.first = 0

    def launch(self):
        if self.used >= self.pool.shape[0]:
            return
        t = self.pool[self.used]
        self.env.check(self.env.lib.akp_clock_probe_dev(self.env.ctx.h, self.chain_len, t.data_ptr(), self.stream.cuda_stream))
        self.used += 1

    def read(self):
        """the probes launched since the last read"""
        self.stream.synchronize()
        rows = self.pool[self.first:self.used].cpu().numpy().view("uint64")
        self.first = self.used
        out = []
        for c, w, _ in rows:
            c, w = int(c), int(w)
            if w:
                out.append({"mhz": 100.0 * c / w, "cycles_per_dependent_mad": c / float(self.chain_len), "ms": w / 1e5})
        return out
