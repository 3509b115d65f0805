#!/bin/bash
# round 6: GPU sessions.   bash tools/gpu_r6.sh <session dir under gpurun_out> [parts]
#   fresh   `python bench.py` as the FIRST GPU process of the lease (what the driver does), then a second run on the same lease
#   vram    tools/vram_probe: hipMalloc / first write / second write per 16 GB slice, two rounds; background hipMalloc beside launches;
#           the virtual-memory API (reserve + create + map per slice)
#   single  tools/vram_probe in `single` mode, two processes: ONE 46 GB hipMalloc (+ one of 23 GB), twice per process
#   test    pytest -m gpu        smoke   __graft_entry__.smoke()        bench   one more bench.py (line + full record)
#   stats   rocprofv3 --kernel-trace --stats of the bench command (side loops off)
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r06_sX}; mkdir -p $OUT
PARTS=${2:-"fresh vram test"}
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for P in $PARTS; do case $P in
  fresh) timeout 900 python bench.py > $OUT/bench_line_first_process.json 2> $OUT/bench_stderr_first.txt; echo "bench(first) rc=$?"; cp bench_full.json $OUT/bench_full_first_process.json
         timeout 900 python bench.py > $OUT/bench_line_second_process.json 2> $OUT/bench_stderr_second.txt; echo "bench(second) rc=$?"; cp bench_full.json $OUT/bench_full_second_process.json
         cat $OUT/bench_line_first_process.json; cat $OUT/bench_line_second_process.json ;;
  vram)  for M in malloc bg vmm; do timeout 300 tools/vram_probe ${VRAM_GB:-16} ${VRAM_SLICES:-8} 2 $M > $OUT/vram_probe_$M.txt 2>&1; echo "vram $M rc=$?"; cat $OUT/vram_probe_$M.txt; done ;;
  single) for I in 1 2; do timeout 300 tools/vram_probe 46 1 2 single > $OUT/vram_probe_single_process$I.txt 2>&1; echo "single $I rc=$?"; cat $OUT/vram_probe_single_process$I.txt; done ;;
  test)  timeout 1500 python -m pytest tests -m gpu -q --timeout=600 > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest_gpu.txt ;;
  smoke) timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.txt ;;
  bench) timeout 900 python bench.py > $OUT/bench_line.json 2> $OUT/bench_stderr.txt; echo "bench rc=$?"; cp bench_full.json $OUT/; wc -c $OUT/bench_line.json; cat $OUT/bench_line.json ;;
  stats) (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-host-path --sustain-seconds 0 --no-sweep --proofs-log2 0 --ragged-log2 0 > $OUT/bench_under_rocprof.txt 2>&1)
         F=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $OUT/rocprof_kernel_stats_bench_py.csv; rm -rf $OUT/prof; head -12 $OUT/rocprof_kernel_stats_bench_py.csv ;;
  *)     echo "unknown part $P" ;;
esac; done
