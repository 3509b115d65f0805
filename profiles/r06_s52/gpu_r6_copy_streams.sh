#!/bin/bash
# round 6: the copy streams of the pageable curve-hash call and of the host tree build at high priority (test build: AKP_COPY_STREAMS_HIGH),
# in bench.py's placement, twice each
O=gpurun_out/r06_s52; mkdir -p $O
export AKP_LIB=$PWD/crypto_primitives_amd/lib/libakp_testhooks.so AKP_BENCH_FULL=$PWD/$O/full.json
MIN="--bh-merkle-log2 0 --proofs-log2 0 --ragged-log2 0 --no-sweep --sustain-seconds 0 --no-cpu-baseline"
for rep in 1 2; do for H in 0 1; do
AKP_COPY_STREAMS_HIGH=$H timeout 300 python bench.py $MIN > $O/line_$H_$rep.json 2> $O/err_$H_$rep.txt
python - $H $rep <<'P'
import json,sys
j=json.load(open("gpurun_out/r06_s52/full.json"))["host_path"]
print("copy streams high =",sys.argv[1],"rep",sys.argv[2],{a:round(b["ms_per_batch"],2) for a,b in j.items() if isinstance(b,dict) and "ms_per_batch" in b}, "tree 2^22 from host leaves: %.2f ms"%(j["merkle_root_only"]["seconds"]*1e3))
P
done; done | tee $O/summary.txt
