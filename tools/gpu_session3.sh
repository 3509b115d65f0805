#!/bin/bash
TAG=${1:-s3}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest gpu ==";  timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -5 $OUT/pytest_gpu.log
echo "== bench_all ==";   timeout 1500 python tools/bench_all.py > $OUT/bench_all.jsonl 2> $OUT/bench_all.err; cat $OUT/bench_all.jsonl; tail -3 $OUT/bench_all.err
echo "== bench ==";       timeout 600 python bench.py > $OUT/bench.log 2>&1; tail -1 $OUT/bench.log
for C in FETCH_SIZE WRITE_SIZE; do
  echo "== pmc $C =="
  (cd /tmp && timeout 600 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -o pmc -- python $OLDPWD/bench.py --steps 3 --warmup 1 --merkle-log2 0 --no-cpu-baseline > $OUT/pmc_$C.log 2>&1); tail -2 $OUT/pmc_$C.log
  for f in $(find $OUT/pmc_$C -name "*counter_collection.csv"); do head -3 $f; grep -c . $f; done
done
find $OUT -name "*.csv" -size +4M -delete 2>/dev/null
