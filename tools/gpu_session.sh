#!/bin/bash
# One GPU-box session: microbench, smoke, GPU parity tests, bench, rocprof kernel trace.
# Usage (from the repo root on the GPU box): bash tools/gpu_session.sh [tag]
TAG=${1:-s1}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
[ -z "$GRAFT_REPO_ROOT" ] && OUT=$(pwd)/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== rocminfo ==" > $OUT/env.log; rocminfo | grep -E "Marketing Name|gfx|Compute Unit|Max Clock" | head -12 >> $OUT/env.log 2>&1
lscpu | grep -E "Model name|^CPU\(s\)|Thread|Socket" >> $OUT/env.log 2>&1
echo "== microbench ==";  timeout 300 tools/microbench > $OUT/microbench.log 2>&1; tail -5 $OUT/microbench.log
echo "== smoke ==";       timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -3 $OUT/smoke.log
echo "== pytest gpu ==";  timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -15 $OUT/pytest_gpu.log
echo "== bench ==";       timeout 900 python bench.py > $OUT/bench.log 2>&1; tail -3 $OUT/bench.log
echo "== rocprof ==";     (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python $OLDPWD/bench.py --steps 5 --warmup 1 --merkle-log2 20 --no-cpu-baseline > $OUT/rocprof.log 2>&1); tail -3 $OUT/rocprof.log
ls -R $OUT/prof 2>/dev/null | head -20
for f in $(find $OUT/prof -name "*kernel_stats.csv" 2>/dev/null); do echo "--- $f"; head -12 $f; done
# keep only the small summaries (<= 64 MiB merge limit)
find $OUT/prof -name "*.csv" -size +8M -delete 2>/dev/null
