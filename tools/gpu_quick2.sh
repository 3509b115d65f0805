#!/bin/bash
TAG=${1:-q}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest gpu ==";  timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -5 $OUT/pytest_gpu.log
echo "== bench_all ==";   timeout 1500 python tools/bench_all.py --max-log2 22 > $OUT/bench_all.jsonl 2> $OUT/bench_all.err; grep -E "pedersen|bowe" $OUT/bench_all.jsonl; tail -3 $OUT/bench_all.err
echo "== D=4 / group=1 arms =="; AKP_PEDERSEN_DIGIT_BITS=4 AKP_BH_GROUP=1 timeout 900 python tools/bench_all.py --max-log2 12 2>/dev/null | grep -E "pedersen_crh|bowe_hopwood_crh" | grep -v cpu_baseline
