#!/bin/bash
# round 6 (= round 5's script, new session names): HBM traffic / VALU counters of the hot kernels, one counter per rocprofv3 pass (separate --pmc passes, no trace domains), on
# tools/prof_driver.py: the permutation / CRH kernels, the curve-hash kernels with the library's default (cache-sized) tables and with
# the HBM-sized tables (PROF_TABLES=hbm).  Usage: bash tools/gpu_r6_pmc.sh <session dir under gpurun_out>
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r06_pmc}; mkdir -p $OUT
export TMPDIR=/tmp
for W in poseidon te te_hbm; do
  : > $OUT/pmc_$W.txt
  DRV=$W; [ $W = te_hbm ] && DRV=te
  for C in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU VALUBusy; do
    (cd /tmp && PROF_REPS=2 PROF_TABLES=$([ $W = te_hbm ] && echo hbm) timeout 600 rocprofv3 --pmc $C --output-format csv -d $OUT/p_${W}_$C -o pmc -- python $GRAFT_REPO_ROOT/tools/prof_driver.py $DRV > $OUT/p_${W}_$C.log 2>&1)
    F=$(find $OUT/p_${W}_$C -name "*counter_collection.csv" | head -1)
    [ -n "$F" ] && python - "$F" "$C" >> $OUT/pmc_$W.txt <<'PY'
import csv, sys, collections
f, c = sys.argv[1], sys.argv[2]
acc = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    name = r.get("Kernel_Name", "?")
    name = name[name.find("akp::"):] if "akp::" in name else name
    name = name.split("(")[0]
    if "akp::" not in name or "build" in name or "halve" in name or "convert" in name:
        continue
    acc.setdefault((name, r.get("Grid_Size", "?")), []).append(float(r.get("Counter_Value", 0)))
for (k, g), v in acc.items():
    print("%-12s %-44s grid %-9s launches %d  values %s" % (c, k, g, len(v), " ".join("%.6g" % x for x in v[-4:])))
PY
    rm -rf $OUT/p_${W}_$C $OUT/p_${W}_$C.log
  done
  echo "== $W"; cat $OUT/pmc_$W.txt
done
