#!/bin/bash
# round 6: the kept form of the gated launch (high-priority copy-in stream of its own, wave priorities) over the stream placements that
# stalled it (K extra streams), the gated tests, and bench.py, whose pinned Pedersen leg was the 14 ms case
O=gpurun_out/r06_s48; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_host_path_gated.py -x -q -m gpu 2>&1 | tail -3 | tee $O/pytest_gated.txt
timeout 400 python bench.py > $O/bench_line.json 2> $O/bench.err; cp bench_full.json $O/bench_full.json
python - <<'P'
import json
j=json.load(open("gpurun_out/r06_s48/bench_full.json"))["host_path"]
print("bench host_path", {a:round(b["ms_per_batch"],2) for a,b in j.items() if isinstance(b,dict) and "ms_per_batch" in b})
P
for K in 0 1 2 4; do timeout 200 python tools/gpu_r6_gate_grid.py $K 2>/dev/null | tee -a $O/placements.jsonl; done
for K in 0 1 2 4; do timeout 200 python tools/gpu_r6_gate_grid.py $K hbm 2>/dev/null | tee -a $O/placements.jsonl; done
