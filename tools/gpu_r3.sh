#!/bin/bash
# one GPU session of round 3: tools/gpu_r3.sh <tag> [steps...]
#   steps: test (pytest -m gpu)  bench  proofsprof (rocprofv3 kernel stats of tools/bench_proofs.py)
#          teab (A/B of the curve-table entry layout: 128-byte line vs 96-byte packed; probe + kernel stats + PMC passes per build)
TAG=${1:-r03_s1}; shift
STEPS=${@:-test bench}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
pmc_pass() {  # $1 = counter, $2 = label, rest = command; appends per-kernel sums to $OUT/pmc_$2.txt
  local C=$1 L=$2; shift 2
  (cd /tmp && PROF_REPS=2 timeout 600 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_${L}_$C -o pmc -- "$@" > $OUT/pmc_${L}_$C.log 2>&1)
  local F=$(find $OUT/pmc_${L}_$C -name "*counter_collection.csv" | head -1)
  [ -n "$F" ] && python - "$F" "$C" >> $OUT/pmc_$L.txt <<'PY'
import csv, sys, collections
f, c = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    acc[(r.get("Kernel_Name", "?")[:60], r.get("Counter_Name", c))].append(float(r.get("Counter_Value", 0)))
for (k, n), v in sorted(acc.items()):
    print(k, n, "launches", len(v), "last", v[-1], "mean", sum(v) / len(v))
PY
  find $OUT/pmc_${L}_$C -name "*counter_collection.csv" -size +2M -delete
}
for S in $STEPS; do case $S in
test)  echo "== pytest gpu =="; timeout 2400 python -m pytest tests -m gpu -q --durations=10 > $OUT/pytest_gpu.log 2>&1; tail -25 $OUT/pytest_gpu.log;;
smoke) echo "== smoke =="; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -3 $OUT/smoke.log;;
bench) echo "== bench =="; timeout 1200 python bench.py > $OUT/bench.json 2> $OUT/bench.err; tail -c 3000 $OUT/bench.json; tail -5 $OUT/bench.err;;
benchprof) echo "== rocprofv3 --kernel-trace --stats of bench.py itself =="
   (cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/benchprof -o trace -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-host-path --sustain-seconds 0 > $OUT/bench_under_rocprofv3.json 2> $OUT/benchprof.err); tail -c 300 $OUT/bench_under_rocprofv3.json
   F=$(find $OUT/benchprof -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $OUT/rocprof_kernel_stats_bench_py.csv && head -14 $OUT/rocprof_kernel_stats_bench_py.csv | cut -c1-200
   rm -rf $OUT/benchprof;;
proofsprof) echo "== rocprofv3 kernel stats of tools/bench_proofs.py =="
   for CFG in poseidon bh; do
     (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pp_$CFG -o trace -- python $GRAFT_REPO_ROOT/tools/bench_proofs.py --config $CFG > $OUT/proofs_${CFG}_under_rocprofv3.json 2> $OUT/pp_$CFG.err)
     F=$(find $OUT/pp_$CFG -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $OUT/proofs_${CFG}_kernel_stats.csv && head -16 $OUT/proofs_${CFG}_kernel_stats.csv | cut -c1-180
     rm -rf $OUT/pp_$CFG
   done
   python tools/bench_proofs.py > $OUT/proofs.json 2> $OUT/proofs.err; tail -3 $OUT/proofs.err;;
pmc) for C in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU VALUBusy GRBM_GUI_ACTIVE; do pmc_pass $C poseidon python $GRAFT_REPO_ROOT/tools/prof_driver.py poseidon; done
   grep "permute_t3\|crh_t3" $OUT/pmc_poseidon.txt | cut -c1-200;;
gaps) echo "== launch gaps of the narrow tree levels =="
   (cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/gaps -o t -- python $GRAFT_REPO_ROOT/tools/gpu_level_gaps.py run > $OUT/gaps_run.log 2>&1)
   python tools/gpu_level_gaps.py analyse $OUT/gaps > $OUT/level_gaps.txt 2>&1; grep "==\|total" $OUT/level_gaps.txt; rm -rf $OUT/gaps;;
finab) for V in default splitfin; do
     if [ $V = splitfin ]; then export AKP_LIB=$GRAFT_REPO_ROOT/crypto_primitives_amd/lib/libakp_splitfin.so; else unset AKP_LIB; fi
     echo "== finalize overlap: $V (AKP_LIB=${AKP_LIB:-default}) =="
     for R in 1 2; do python tools/gpu_te_gather_probe.py 2>&1 | grep "random" ; done | tee $OUT/te_probe_$V.txt
     python bench.py --steps 3 --warmup 1 --merkle-log2 0 --proofs-log2 0 --no-cpu-baseline --no-host-path --sustain-seconds 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench legs: pedersen %.4g hashes/s (%.3f ms), bh tree 2^23 %.2f ms' % (d['pedersen']['hashes_per_s'], d['pedersen']['ms_per_batch'], d['bh_merkle']['seconds']*1e3))" | tee -a $OUT/te_probe_$V.txt
     if [ $V = splitfin ]; then timeout 900 python -m pytest tests/test_gpu_curves.py tests/test_gpu_canaries.py -m gpu -q 2>&1 | tail -2 | tee -a $OUT/te_probe_$V.txt; fi
   done; unset AKP_LIB;;
teab) for V in line128 packed96; do
     if [ $V = packed96 ]; then export AKP_LIB=$GRAFT_REPO_ROOT/crypto_primitives_amd/lib/libakp_packed96.so; else unset AKP_LIB; fi
     echo "== te entry layout: $V (AKP_LIB=${AKP_LIB:-default}) =="
     python tools/gpu_te_gather_probe.py 2>&1 | grep -v amdgpu.ids > $OUT/te_gather_probe_$V.txt; cat $OUT/te_gather_probe_$V.txt
     (cd /tmp && PROF_REPS=6 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st_$V -o trace -- python $GRAFT_REPO_ROOT/tools/prof_driver.py te > $OUT/st_$V.log 2>&1)
     F=$(find $OUT/st_$V -name "*kernel_stats.csv" | head -1); [ -n "$F" ] && cp $F $OUT/te_kernel_stats_$V.csv && grep -i "accumulate\|finalize\|small" $OUT/te_kernel_stats_$V.csv | cut -c1-200
     python - $OUT/st_$V <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(set)
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "accumulate" in r["Kernel_Name"] or "small" in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:48]].add((r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("SGPR_Count"), r.get("Scratch_Size"), r.get("LDS_Block_Size")))
for k, v in sorted(agg.items()):
    print("registers (VGPR, AGPR, SGPR, scratch, LDS):", k, sorted(v))
PY
     rm -rf $OUT/st_$V
     for C in FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum TCC_EA_RDREQ_sum SQ_INSTS_VALU VALUBusy; do pmc_pass $C te_$V python $GRAFT_REPO_ROOT/tools/prof_driver.py te; done
     grep -i "accumulate" $OUT/pmc_te_$V.txt | cut -c1-200
   done; unset AKP_LIB;;
esac; done
du -sh $OUT
