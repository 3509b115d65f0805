#!/bin/bash
# full evidence session: GPU tests, bench_all, bench (+rocprof kernel trace), PMC passes
TAG=${1:-full}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest gpu ==";  timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -4 $OUT/pytest_gpu.log
echo "== bench_all ==";   timeout 1500 python tools/bench_all.py > $OUT/bench_all.jsonl 2> $OUT/bench_all.err; cat $OUT/bench_all.jsonl | cut -c1-220
echo "== bench ==";       timeout 600 python bench.py > $OUT/bench.log 2>&1; tail -1 $OUT/bench.log
echo "== rocprof ==";     (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- python $OLDPWD/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/rocprof.log 2>&1); head -5 $OUT/prof/trace_kernel_stats.csv | cut -c1-260
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -o pmc -- python $OLDPWD/bench.py --steps 3 --warmup 1 --merkle-log2 0 --no-cpu-baseline > $OUT/pmc_$C.log 2>&1)
  grep "permute_t3" $OUT/pmc_$C/pmc_counter_collection.csv | tail -2 | awk -F, '{print $(NF-3), $(NF-2)}'
done
echo "== latency sweeps =="
timeout 300 python tools/gpu_coop.py > $OUT/latency_poseidon_t3.txt 2>&1; AKP_POSEIDON_COOP_MAX=0 timeout 300 python tools/gpu_coop.py >> $OUT/latency_poseidon_t3.txt 2>&1
timeout 300 python tools/gpu_te_latency.py > $OUT/latency_te.txt 2>&1; AKP_TE_SPLIT_MAX=0 timeout 300 python tools/gpu_te_latency.py 2>&1 | sed 's/^/split-kernel off: /' >> $OUT/latency_te.txt
timeout 600 bash tools/gpu_generic.sh > $OUT/generic_rates.txt 2>&1
timeout 120 python tools/gpu_ramp.py > $OUT/clock_ramp.txt 2>&1
grep -h "n=2^10 \|n=2^14 " $OUT/latency_poseidon_t3.txt $OUT/latency_te.txt | cut -c1-90
find $OUT -name "*kernel_trace.csv" -size +1M -delete
