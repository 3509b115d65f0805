#!/bin/bash
# round 6: the pinned Pedersen / Bowe-Hopwood host path (gated launch) with and without the Python mirror's fixed mmap threshold,
# in the tool's own loop and inside bench.py's host_path leg (profiles/r06_s39 showed 14 ms per 2^20 hashes there against 3.9 ms before).
O=gpurun_out/r06_s40; mkdir -p $O
T=crypto_primitives_amd/lib/libakp_testhooks.so
AKP_LIB=$PWD/$T timeout 300 python tools/gpu_r5_gated.py > $O/gated_default.json 2> $O/gated_default.err
AKP_KEEP_MALLOC=1 AKP_LIB=$PWD/$T timeout 300 python tools/gpu_r5_gated.py > $O/gated_keep_malloc.json 2> $O/gated_keep_malloc.err
timeout 400 python bench.py --no-sweep --sustain-seconds 0 --cpu-seconds 2 > $O/bench_default.json 2> $O/bench_default.err; cp bench_full.json $O/bench_full_default.json
AKP_KEEP_MALLOC=1 timeout 400 python bench.py --no-sweep --sustain-seconds 0 --cpu-seconds 2 > $O/bench_keep_malloc.json 2> $O/bench_keep_malloc.err; cp bench_full.json $O/bench_full_keep_malloc.json
python - <<'P'
import json
for k in ("default","keep_malloc"):
    try:
        j=json.load(open(f"gpurun_out/r06_s40/bench_full_{k}.json"))["host_path"]
        print(k, {a:round(b["ms_per_batch"],2) for a,b in j.items() if isinstance(b,dict) and "ms_per_batch" in b})
    except Exception as e: print(k,"ERR",e)
    try:
        j=json.load(open(f"gpurun_out/r06_s40/gated_{k}.json"))
        for t in ("cache_sized","hbm_sized"):
            for n,r in j[t].items(): print(k,t,n,round(r["chunked"]["ms_median"],2),round(r["gated"]["ms_median"],2))
    except Exception as e: print(k,"ERR",e)
P
