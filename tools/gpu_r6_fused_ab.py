"""round 6: would folding te_finalize_kernel into the accumulate kernel's tail pay on the RESIDENT path?  The gated
kernel of the pinned host path (te_accumulate_lds_gated_fused_kernel) IS that fusion -- two messages per lane, one inversion per 512 points
through an LDS product tree, digests written by the kernel -- so its pace on data that is already there is the answer.  Test build
(AKP_LIB=.../libakp_testhooks.so): the pinned call with the copy-in and copy-out switched off (the arrival flags are written at once: no
workgroup waits) against the resident launch (te_accumulate_lds_kernel + te_finalize_kernel, device time between events), 2^20 Pedersen
4x256 hashes of 128 bytes and 2^20 Bowe-Hopwood 63x9 hashes of 64 bytes, both table sizes."""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import crypto_primitives_amd as cpa
from crypto_primitives_amd import params as cparams
from crypto_primitives_amd.crh import pedersen, bowe_hopwood
lib, check = cpa.lib, cpa._lib.check
assert os.environ.get("AKP_LIB", "").endswith("libakp_testhooks.so"), "needs the test build"
dev = torch.device("cuda", 0)
ctx = cpa.default_context(0)
n = 1 << 20
out = {"hashes": n}
for name, cls, gens, L, fe in (("pedersen_4x256_128B", pedersen, cparams.pedersen_generators(0xA5A50004, 4, 256), 128, 2),
                               ("bowe_hopwood_63x9_64B", bowe_hopwood, cparams.bowe_hopwood_generators(0xA5A50005, 63, 9), 64, 1)):
    for table in ("cache_sized", "hbm_sized"):
        ctx.set_table_budget(0 if table == "cache_sized" else cpa._lib.TABLE_BUDGET_DEVICE)
        h = cls.Parameters(gens).handle(ctx)
        ctx.set_table_budget(0)
        h.prepare(L)
        msgs = np.random.default_rng(7).integers(0, 256, size=(n, L), dtype=np.uint8)
        d_m = torch.from_numpy(msgs).to(dev)
        d_o = torch.empty((n, 4 * fe), dtype=torch.int64, device=dev)
        st = torch.cuda.current_stream(dev).cuda_stream
        for _ in range(5):
            check(lib.akp_te_crh_batch_dev(h.h, d_m.data_ptr(), n, L, d_o.data_ptr(), st))
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(15)]
        for a, b in evs:
            a.record(); check(lib.akp_te_crh_batch_dev(h.h, d_m.data_ptr(), n, L, d_o.data_ptr(), st)); b.record()
        torch.cuda.synchronize(dev)
        res_ms = sorted(a.elapsed_time(b) for a, b in evs)
        pm, po = C.c_void_p(), C.c_void_p()
        check(lib.akp_host_alloc(msgs.nbytes, C.byref(pm))); check(lib.akp_host_alloc(n * 32 * fe, C.byref(po)))
        np.ctypeslib.as_array((C.c_uint8 * msgs.size).from_address(pm.value))[:] = msgs.reshape(-1)
        os.environ["AKP_TE_GATE_SKIP_COPY_IN"] = "1"; os.environ["AKP_TE_GATE_SKIP_COPY_OUT"] = "1"
        ts = []
        for i in range(17):
            t0 = time.perf_counter(); check(lib.akp_te_crh_batch(h.h, pm, n, L, po)); ts.append((time.perf_counter() - t0) * 1e3)
        del os.environ["AKP_TE_GATE_SKIP_COPY_IN"], os.environ["AKP_TE_GATE_SKIP_COPY_OUT"]
        ts = sorted(ts[2:])
        out["%s/%s" % (name, table)] = {"resident_accumulate_plus_finalize_ms_median": round(res_ms[len(res_ms) // 2], 3), "resident_ms_min": round(res_ms[0], 3),
                                        "fused_kernel_alone_wall_ms_median": round(ts[len(ts) // 2], 3), "fused_ms_min": round(ts[0], 3), "steps": h.info(L)["steps"]}
        check(lib.akp_host_free(pm)); check(lib.akp_host_free(po))
        del h
print(json.dumps(out, indent=1))
