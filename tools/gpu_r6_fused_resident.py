"""round 6: the resident curve-hash launch as ONE fused kernel (te_accumulate_lds_fused_kernel) against accumulate + finalize -- run once as
shipped and once with AKP_TE_FUSED_MIN=999999999999 (the two kernels at every size); device ms between events, median of 15, + a digest checksum."""
import hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import crypto_primitives_amd as cpa
from crypto_primitives_amd import params as cparams
from crypto_primitives_amd.crh import pedersen, bowe_hopwood
lib, check = cpa.lib, cpa._lib.check
dev = torch.device("cuda", 0)
ctx = cpa.default_context(0)
st = torch.cuda.current_stream(dev).cuda_stream
out = {"AKP_TE_FUSED_MIN": os.environ.get("AKP_TE_FUSED_MIN", "default (2^18)")}
for name, cls, gens, L, fe in (("pedersen_4x256_128B", pedersen, cparams.pedersen_generators(0xA5A50004, 4, 256), 128, 2),
                               ("pedersen_4x256_32B", pedersen, cparams.pedersen_generators(0xA5A50004, 4, 256), 32, 2),
                               ("bowe_hopwood_63x9_64B", bowe_hopwood, cparams.bowe_hopwood_generators(0xA5A50005, 63, 9), 64, 1),
                               ("bowe_hopwood_63x9_32B", bowe_hopwood, cparams.bowe_hopwood_generators(0xA5A50005, 63, 9), 32, 1)):
    for table in ("cache_sized", "hbm_sized"):
        ctx.set_table_budget(0 if table == "cache_sized" else cpa._lib.TABLE_BUDGET_DEVICE)
        h = cls.Parameters(gens).handle(ctx)
        ctx.set_table_budget(0)
        h.prepare(L)
        for lg in (18, 20, 22):
            n = 1 << lg
            d_m = torch.from_numpy(np.random.default_rng(7).integers(0, 256, size=(1 << 18, L), dtype=np.uint8)).to(dev).repeat(n >> 18, 1)
            d_o = torch.empty((n, 4 * fe), dtype=torch.int64, device=dev)
            for _ in range(4):
                check(lib.akp_te_crh_batch_dev(h.h, d_m.data_ptr(), n, L, d_o.data_ptr(), st))
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(15)]
            for a, b in evs:
                a.record(); check(lib.akp_te_crh_batch_dev(h.h, d_m.data_ptr(), n, L, d_o.data_ptr(), st)); b.record()
            torch.cuda.synchronize(dev)
            ms = sorted(a.elapsed_time(b) for a, b in evs)
            out["%s/%s/2^%d" % (name, table, lg)] = {"ms_median": round(ms[7], 3), "ms_min": round(ms[0], 3),
                                                    "sha": hashlib.sha256(d_o[: 1 << 18].cpu().numpy().tobytes()).hexdigest()[:12]}
        del h
print(json.dumps(out, indent=0))
