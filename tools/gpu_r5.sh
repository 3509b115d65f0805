#!/bin/bash
# round 5: the validation session -- GPU test-suite, smoke, bench.py (line + full record), rocprofv3 --kernel-trace --stats of the same
# bench command (side loops that launch the headline kernel at OTHER sizes switched off, so that its average is the 2^20-state launch), the
# persist_probe raw output.   bash tools/gpu_r5.sh <session dir under gpurun_out> [parts: test smoke bench stats probe]
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05_sX}; mkdir -p $OUT
PARTS=${2:-"test smoke bench stats probe"}
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for P in $PARTS; do case $P in
  test)  timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest_gpu.txt ;;
  smoke) timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.txt ;;
  bench) timeout 900 python bench.py > $OUT/bench_line.json 2> $OUT/bench_stderr.txt; echo "bench rc=$?"; cp bench_full.json $OUT/; wc -c $OUT/bench_line.json; cat $OUT/bench_line.json ;;
  stats) (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-host-path --sustain-seconds 0 --no-sweep --proofs-log2 0 --ragged-log2 0 > $OUT/bench_under_rocprof.txt 2>&1)
         F=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $OUT/rocprof_kernel_stats_bench_py.csv; rm -rf $OUT/prof; head -12 $OUT/rocprof_kernel_stats_bench_py.csv ;;
  probe) timeout 120 tools/persist_probe host 0 > $OUT/persist_probe_host_flags.txt 2>&1; timeout 120 tools/persist_probe dev 0 > $OUT/persist_probe_device_flags_full_occupancy.txt 2>&1
         timeout 120 tools/persist_probe dev 40960 > $OUT/persist_probe_device_flags_4wg_per_cu.txt 2>&1; tail -4 $OUT/persist_probe_device_flags_4wg_per_cu.txt ;;
esac; done
