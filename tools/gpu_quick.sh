#!/bin/bash
TAG=${1:-q}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest gpu ==";  timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -5 $OUT/pytest_gpu.log
echo "== microbench (field ops) =="; timeout 300 tools/microbench > $OUT/microbench.log 2>&1; grep -A40 "field op" $OUT/microbench.log | grep -E "f29|field"
echo "== bench ==";       timeout 600 python bench.py --no-cpu-baseline > $OUT/bench.log 2>&1; tail -1 $OUT/bench.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_avg_ms'], d['merkle']['seconds'])"
AKP_POSEIDON_DENSE=1 timeout 600 python bench.py --no-cpu-baseline --merkle-log2 0 > $OUT/bench_dense.log 2>&1; tail -1 $OUT/bench_dense.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dense partial rounds:', d['value'], d['roofline']['kernel_avg_ms'])"
