"""`ragged`: batches whose items differ in length (the reference hashes every input with its own length: merkle_tree/mod.rs:411-422,
crh/bowe_hopwood/mod.rs:131-138, crh/pedersen/mod.rs:82-99, crh/poseidon/mod.rs:30-40) -- one launch with per-lane step counts, the
items ordered by step count on the device (tools: ragged_sort.hpp).  Resident inputs, device time between events on the launch
stream; each figure next to the uniform batch of the MEAN length (what the same bytes would cost if they were equally long) and,
in the test build, next to the unsorted launch (AKP_RAGGED_SORT=0).

Standalone (A/B of the launch order; needs libakp_testhooks.so):  AKP_LIB=.../libakp_testhooks.so python tools/bench_legs/ragged.py"""
import os
import sys
import time


def _timed(torch, dev, fn, reps=5):
    fn()
    torch.cuda.synchronize(dev)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize(dev)
    ms = sorted(a.elapsed_time(b) for a, b in evs)
    return ms[len(ms) // 2]


def measure(np, torch, cpa, dev, stream, ctx, log2n, ora_threads=8, parity=True):
    from crypto_primitives_amd import params as cparams, field
    from crypto_primitives_amd.crh import bowe_hopwood, pedersen
    lib, check = cpa.lib, cpa._lib.check
    n = 1 << log2n
    rng = np.random.default_rng(0xA5A50051)
    out = {"items": n, "timing": "median of 5 launches, device time between events on the launch stream, inputs / offsets / digests resident in HBM"}

    def offsets_of(lens):
        o = np.zeros(n + 1, np.uint64)
        o[1:] = np.cumsum(lens.astype(np.uint64))
        return o

    def te_case(name, handle, fe, max_len, uniform_len, cref_fn):
        lens = rng.integers(0, max_len + 1, size=n)
        offs = offsets_of(lens)
        flat = rng.integers(0, 256, size=max(int(offs[-1]), 1), dtype=np.uint8)
        d_flat, d_offs = torch.from_numpy(flat).to(dev), torch.from_numpy(offs.view(np.int64)).to(dev)
        d_out = torch.empty((n, 4 * fe), dtype=torch.int64, device=dev)
        handle.prepare(max_len)
        rec = {"lengths": "uniform random 0 .. %d bytes (mean %.1f)" % (max_len, float(lens.mean()))}
        ms = _timed(torch, dev, lambda: check(lib.akp_te_crh_batch_ragged_dev(handle.h, d_flat.data_ptr(), d_offs.data_ptr(), n, max_len, d_out.data_ptr(), stream)))
        rec["ms"] = ms
        rec["hashes_per_s"] = n / (ms / 1e3)
        rec["message_bytes_per_s"] = float(offs[-1]) / (ms / 1e3)
        if cpa._lib.LIB_PATH.endswith("libakp_testhooks.so"):  # the launch order A/B exists only in the test build
            os.environ["AKP_RAGGED_SORT"] = "0"
            try:
                rec["ms_unsorted_launch"] = _timed(torch, dev, lambda: check(lib.akp_te_crh_batch_ragged_dev(handle.h, d_flat.data_ptr(), d_offs.data_ptr(), n, max_len, d_out.data_ptr(), stream)))
            finally:
                del os.environ["AKP_RAGGED_SORT"]
        uni = torch.from_numpy(rng.integers(0, 256, size=(n, uniform_len), dtype=np.uint8)).to(dev)
        rec["ms_uniform_batch_of_%d_bytes" % uniform_len] = _timed(torch, dev, lambda: check(lib.akp_te_crh_batch_dev(handle.h, uni.data_ptr(), n, uniform_len, d_out.data_ptr(), stream)))
        uni_max = torch.from_numpy(rng.integers(0, 256, size=(n, max_len), dtype=np.uint8)).to(dev)
        rec["ms_uniform_batch_of_%d_bytes" % max_len] = _timed(torch, dev, lambda: check(lib.akp_te_crh_batch_dev(handle.h, uni_max.data_ptr(), n, max_len, d_out.data_ptr(), stream)))
        if parity:
            check(lib.akp_te_crh_batch_ragged_dev(handle.h, d_flat.data_ptr(), d_offs.data_ptr(), n, max_len, d_out.data_ptr(), stream))
            torch.cuda.synchronize(dev)
            got = d_out.cpu().numpy().view(np.uint64).reshape(n, fe, 4)
            si = np.unique(np.concatenate([np.arange(32), np.linspace(0, n - 1, 161).astype(np.int64)]))
            ok = True
            for L in np.unique(lens[si]):
                sel = si[lens[si] == L]
                idx = offs[sel].astype(np.int64)[:, None] + np.arange(L)[None, :]
                arr = np.ascontiguousarray(flat[idx]) if L else np.zeros((len(sel), 0), np.uint8)
                ok = ok and np.array_equal(got[sel], cref_fn(arr, len(sel), int(L)).reshape(len(sel), fe, 4))
            rec["sampled_parity_bit_exact"] = bool(ok)
            rec["parity_samples"] = int(len(si))
        out[name] = rec

    from oracle import cref
    gb = cparams.bowe_hopwood_generators(0xA5A50005, 63, 9)
    B = bowe_hopwood.Parameters(gb)
    cb = cref.CurveParams(63, 9, gb)
    te_case("bowe_hopwood_63x9", B.handle(ctx), 1, 64, 32, lambda a, k, L: cb.bh_crh_batch(a, k, L, threads=ora_threads))
    gp = cparams.pedersen_generators(0xA5A50004, 4, 256)
    P = pedersen.Parameters(gp)
    cp = cref.CurveParams(4, 256, gp)
    te_case("pedersen_4x256", P.handle(ctx), 2, 128, 64, lambda a, k, L: cp.pedersen_crh_batch(a, k, L, threads=ora_threads))
    # Poseidon: 0 .. 8 elements per input (1 .. 4 permutations at rate 2)
    cfg = cpa.get_default_poseidon_parameters(2, False)
    ph = cfg.handle(ctx)
    lens = rng.integers(0, 9, size=n)
    offs = offsets_of(lens)
    elems = field.random_fr(1 << 16, seed=0xA5A50052).reshape(-1, 4)
    d_e = torch.from_numpy(np.ascontiguousarray(np.tile(elems, (int(offs[-1]) // len(elems) + 1, 1))[: int(offs[-1])]).view(np.int64)).to(dev)
    d_o = torch.from_numpy(offs.view(np.int64)).to(dev)
    d_out = torch.empty((n, 4), dtype=torch.int64, device=dev)
    rec = {"lengths": "uniform random 0 .. 8 elements (mean %.2f): %.2f permutations per input on average" % (float(lens.mean()), float(np.maximum(1, -(-lens // 2)).mean()))}
    rec["ms"] = _timed(torch, dev, lambda: check(lib.akp_poseidon_crh_batch_ragged_dev(ph.h, d_e.data_ptr(), d_o.data_ptr(), n, d_out.data_ptr(), stream)))
    rec["hashes_per_s"] = n / (rec["ms"] / 1e3)
    rec["permutations_per_s"] = float(np.maximum(1, -(-lens // 2)).sum()) / (rec["ms"] / 1e3)
    uni = torch.from_numpy(np.ascontiguousarray(np.tile(elems, (4 * n // len(elems) + 1, 1))[: 4 * n]).view(np.int64)).to(dev)
    rec["ms_uniform_batch_of_4_elements"] = _timed(torch, dev, lambda: check(lib.akp_poseidon_crh_batch_dev(ph.h, uni.data_ptr(), n, 4, d_out.data_ptr(), stream)))
    if parity:
        check(lib.akp_poseidon_crh_batch_ragged_dev(ph.h, d_e.data_ptr(), d_o.data_ptr(), n, d_out.data_ptr(), stream))
        torch.cuda.synchronize(dev)
        got = d_out.cpu().numpy().view(np.uint64)
        host_e = d_e.cpu().numpy().view(np.uint64).reshape(-1, 4)
        po = cref.Poseidon(cfg.full_rounds, cfg.partial_rounds, cfg.alpha, cfg.rate, cfg.capacity, cfg.ark, cfg.mds)
        si = np.unique(np.concatenate([np.arange(32), np.linspace(0, n - 1, 161).astype(np.int64)]))
        ok = True
        for L in np.unique(lens[si]):
            sel = si[lens[si] == L]
            idx = offs[sel].astype(np.int64)[:, None] + np.arange(L)[None, :]
            exp = po.crh_batch(np.ascontiguousarray(host_e[idx]).reshape(len(sel), int(L), 4), int(L), threads=ora_threads) if L else np.tile(po.crh_empty(), (len(sel), 1))
            ok = ok and np.array_equal(got[sel], np.asarray(exp).reshape(len(sel), 4))
        rec["sampled_parity_bit_exact"] = bool(ok)
    out["poseidon_rate2"] = rec
    return out


def run(env):
    if env.rank != 0 or env.world != 1 or not env.args.ragged_log2:
        return None
    rec = measure(env.np, env.torch, env.cpa, env.dev, env.stream, env.ctx, env.args.ragged_log2, env.ora_threads)
    for k in ("bowe_hopwood_63x9", "pedersen_4x256", "poseidon_rate2"):
        if not rec[k].get("sampled_parity_bit_exact", True):
            raise SystemExit("ragged leg (%s): sampled digests differ from the oracle" % k)
    return rec


if __name__ == "__main__":
    ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, ROOT)
    import json
    import numpy as np
    import torch
    import crypto_primitives_amd as cpa
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    ctx = cpa.default_context(0)
    if len(sys.argv) > 2 and sys.argv[2] == "hbm":
        ctx.set_table_budget(cpa._lib.TABLE_BUDGET_DEVICE)
    t0 = time.time()
    r = measure(np, torch, cpa, dev, torch.cuda.current_stream(dev).cuda_stream, ctx, int(sys.argv[1]) if len(sys.argv) > 1 else 20)
    r["library"] = os.path.basename(cpa._lib.LIB_PATH)
    r["seconds"] = time.time() - t0
    print(json.dumps(r, indent=1))
