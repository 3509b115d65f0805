#!/bin/bash
# round 6: the gated launch with a RESIDENT grid (workgroups walk the tiles) against one workgroup per tile (AKP_TE_GATE_PERSIST=0, test build),
# where the copy-in stream's queue shares a command-processor pipe with the kernel's (bench.py: profiles/r06_s41) and where it does not (the tool).
O=gpurun_out/r06_s43; mkdir -p $O
T=$PWD/crypto_primitives_amd/lib/libakp_testhooks.so
export AKP_BENCH_FULL=$PWD/$O/full.json
show() { python - "$1" <<'P'
import json,sys
j=json.load(open("gpurun_out/r06_s43/full.json"))["host_path"]
print(sys.argv[1], {a:round(b["ms_per_batch"],2) for a,b in j.items() if isinstance(b,dict) and "ms_per_batch" in b})
P
}
timeout 600 python -m pytest tests/test_gpu_host_path_gated.py -x -q -m gpu 2>&1 | tail -3 | tee $O/pytest_gated.txt
MIN="--merkle-log2 0 --bh-merkle-log2 0 --proofs-log2 0 --ragged-log2 0 --no-sweep --sustain-seconds 0 --no-cpu-baseline"
AKP_LIB=$T AKP_TE_GATE_PERSIST=1 timeout 300 python bench.py $MIN > $O/P1.json 2> $O/P1.err; show bench_minimal_resident_grid
AKP_LIB=$T AKP_TE_GATE_PERSIST=0 timeout 300 python bench.py $MIN > $O/P0.json 2> $O/P0.err; show bench_minimal_grid_per_tile
for p in 1 0; do
AKP_LIB=$T AKP_TE_GATE_PERSIST=$p timeout 300 python tools/gpu_r5_gated.py > $O/tool_persist$p.json 2> $O/tool_persist$p.err
python - $p <<'P'
import json,sys
j=json.load(open(f"gpurun_out/r06_s43/tool_persist{sys.argv[1]}.json"))
for t in ("cache_sized","hbm_sized"):
    for n,r in j[t].items(): print("tool persist="+sys.argv[1],t,n,"chunked %.2f gated %.2f (min %.2f) resident %.2f"%(r["chunked"]["ms_median"],r["gated"]["ms_median"],r["gated"]["ms_min"],r["resident_ms"]), r["gated"]["digests_equal_the_pageable_call"])
P
done
