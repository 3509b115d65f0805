#!/bin/bash
# one GPU session of round 2: tools/gpu_r2.sh <tag> [steps...]   steps: test smoke bench stats pmc pmcte avail
TAG=${1:-r02_s1}; shift
STEPS=${@:-test smoke bench stats pmc}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocm-smi --showclocks --showpower > $OUT/smi_before.txt 2>&1
for S in $STEPS; do case $S in
test)  echo "== pytest gpu =="; timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; tail -15 $OUT/pytest_gpu.log;;
smoke) echo "== smoke =="; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -3 $OUT/smoke.log;;
bench) echo "== bench =="; timeout 900 python bench.py > $OUT/bench.log 2> $OUT/bench.err; tail -c 6000 $OUT/bench.log; tail -5 $OUT/bench.err;;
bench2) echo "== bench --gpus 2 (shared GPU hook) =="; AKP_BENCH_SHARED_GPU=1 timeout 900 python bench.py --gpus 2 --no-cpu-baseline --no-host-path --merkle-log2 20 --bh-merkle-log2 16 --sustain-seconds 0 > $OUT/bench2.log 2> $OUT/bench2.err; tail -c 1500 $OUT/bench2.log; tail -5 $OUT/bench2.err;;
benchprof) echo "== rocprofv3 --kernel-trace --stats of bench.py itself =="
   (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/benchprof -o trace -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-host-path --sustain-seconds 0 > $OUT/benchprof_bench.log 2> $OUT/benchprof.err); tail -c 300 $OUT/benchprof_bench.log
   F=$(find $OUT/benchprof -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $OUT/rocprof_kernel_stats_bench_py.csv && head -12 $OUT/rocprof_kernel_stats_bench_py.csv | cut -c1-200
   find $OUT/benchprof -name "*kernel_trace.csv" -delete;;
stats) echo "== rocprofv3 kernel stats =="
   (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o trace -- python $GRAFT_REPO_ROOT/tools/prof_driver.py > $OUT/stats.log 2>&1); tail -2 $OUT/stats.log
   F=$(find $OUT/stats -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $OUT/rocprof_kernel_stats.csv && head -30 $OUT/rocprof_kernel_stats.csv
   find $OUT/stats -name "*kernel_trace.csv" -delete;;
pmc) for C in SQ_INSTS_VALU VALUBusy GRBM_GUI_ACTIVE SQ_INSTS_SALU FETCH_SIZE WRITE_SIZE; do
     echo "== pmc $C (poseidon) =="
     (cd /tmp && PROF_REPS=2 timeout 600 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$C -o pmc -- python $GRAFT_REPO_ROOT/tools/prof_driver.py poseidon > $OUT/pmc_$C.log 2>&1); tail -1 $OUT/pmc_$C.log
     F=$(find $OUT/pmc_$C -name "*counter_collection.csv" | head -1)
     [ -n "$F" ] && python - "$F" "$C" >> $OUT/pmc_counters.txt <<'PY'
import csv, sys, collections
f, c = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    acc[(r.get("Kernel_Name", "?")[:60], r.get("Counter_Name", c))].append(float(r.get("Counter_Value", 0)))
for (k, n), v in sorted(acc.items()):
    print(k, n, "launches", len(v), "last", v[-1], "mean", sum(v) / len(v))
PY
   done; cat $OUT/pmc_counters.txt; find $OUT -name "*counter_collection.csv" -size +2M -delete;;
pmcte) for C in SQ_INSTS_VALU VALUBusy FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum; do
     echo "== pmc $C (te) =="
     (cd /tmp && PROF_REPS=2 timeout 600 rocprofv3 --pmc $C --output-format csv -d $OUT/pmcte_$C -o pmc -- python $GRAFT_REPO_ROOT/tools/prof_driver.py te > $OUT/pmcte_$C.log 2>&1); tail -1 $OUT/pmcte_$C.log
     F=$(find $OUT/pmcte_$C -name "*counter_collection.csv" | head -1)
     [ -n "$F" ] && python - "$F" "$C" >> $OUT/pmcte_counters.txt <<'PY'
import csv, sys, collections
f, c = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    acc[(r.get("Kernel_Name", "?")[:60], r.get("Counter_Name", c))].append(float(r.get("Counter_Value", 0)))
for (k, n), v in sorted(acc.items()):
    print(k, n, "launches", len(v), "last", v[-1], "mean", sum(v) / len(v))
PY
   done; cat $OUT/pmcte_counters.txt; find $OUT -name "*counter_collection.csv" -size +2M -delete;;
hostpath) echo "== host path probe =="; timeout 600 python tools/host_path_probe.py > $OUT/host_path_probe.txt 2>&1; cat $OUT/host_path_probe.txt;;
hosttrace) echo "== host path timeline =="
   (cd /tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/hosttrace -o t -- python $GRAFT_REPO_ROOT/tools/host_trace_driver.py > $OUT/hosttrace.log 2>&1); tail -4 $OUT/hosttrace.log
   python - $OUT/hosttrace <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "permute" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", r.get("Stream_Id", "?")))
for f in glob.glob(sys.argv[1] + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", "copy")[:14], r.get("Stream_Id", "?")))
rows.sort()
rows = rows[-40:]
t0 = rows[0][0] if rows else 0
for a, b, k, s in rows:
    print("%9.3f ms  +%7.3f ms  %-14s stream %s" % ((a - t0) / 1e6, (b - a) / 1e6, k, s))
PY
   ;;
lanes) for L in 3 4; do for CH in 16 17 18; do AKP_HOST_LANES=$L AKP_HOST_CHUNK_LOG2=$CH python tools/host_path_probe.py child 20 | sed "s/^/lanes $L /"; done; done 2>&1 | grep -v amdgpu.ids | tee $OUT/host_lanes.txt;;
rates) echo "== generic rates (register kernels for t = 4, 5 vs the LDS-file arm) =="
   (python tools/gpu_rates2.py; AKP_POSEIDON_NO_REG_T=1 python tools/gpu_rates2.py) 2>&1 | grep -v amdgpu.ids | tee $OUT/generic_rates.txt;;
digits) echo "== Pedersen table digit width =="
   (for D in 13 14 15; do AKP_PEDERSEN_DIGIT_BITS=$D python tools/gpu_pedersen_digits.py; done; AKP_PEDERSEN_PLAIN=1 AKP_PEDERSEN_DIGIT_BITS=13 python tools/gpu_pedersen_digits.py) 2>&1 | grep -v amdgpu.ids | tee $OUT/pedersen_digits.txt;;
latency) echo "== small-batch latencies =="
   (AKP_POSEIDON_COOP_MAX=0 python tools/gpu_coop.py; AKP_POSEIDON_COOP_MAX=1000000000 python tools/gpu_coop.py) 2>&1 | grep -v amdgpu.ids > $OUT/latency_poseidon_t3.txt; tail -25 $OUT/latency_poseidon_t3.txt
   python tools/gpu_te_latency.py 2>&1 | grep -v amdgpu.ids > $OUT/latency_te.txt; tail -20 $OUT/latency_te.txt;;
benchall) echo "== bench_all =="; timeout 900 python tools/bench_all.py --cpu-seconds 1 > $OUT/bench_all.jsonl 2> $OUT/bench_all.err; cut -c1-230 $OUT/bench_all.jsonl; tail -3 $OUT/bench_all.err;;
avail) rocprofv3 --list-avail 2>/dev/null | grep -i -o "TCC_[A-Z0-9_]*\|MALL[A-Z0-9_]*\|TCP_[A-Z0-9_]*" | sort -u > $OUT/avail_cache_counters.txt; wc -l $OUT/avail_cache_counters.txt;;
esac; done
rocm-smi --showclocks --showpower > $OUT/smi_after.txt 2>&1
du -sh $OUT
