#!/bin/bash
# round 5: calibration of FETCH_SIZE for RANDOM 128-byte-line gathers (tools/gather_probe.hip calib) next to a streaming read -- the
# guide's x2 correction is for 16 B / lane streaming; the curve-hash kernels gather whole table lines.  One counter per pass.
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r05_calib}; mkdir -p $OUT
export TMPDIR=/tmp
$GRAFT_REPO_ROOT/tools/gather_probe calib 64 > $OUT/calib_launches.txt 2>&1
for C in FETCH_SIZE TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_MISS_sum TCC_HIT_sum; do
  (cd /tmp && timeout 600 rocprofv3 --pmc $C --output-format csv -d $OUT/p_$C -o pmc -- $GRAFT_REPO_ROOT/tools/gather_probe calib 64 > $OUT/p_$C.log 2>&1)
  F=$(find $OUT/p_$C -name "*counter_collection.csv" | head -1)
  if [ -n "$F" ]; then
    python - "$F" "$C" >> $OUT/calib_counters.txt <<'PY'
import csv, sys
f, c = sys.argv[1], sys.argv[2]
for r in csv.DictReader(open(f)):
    name = r.get("Kernel_Name", "?").split("(")[0]
    if "fill" in name:
        continue
    print("%-24s %-40s %s" % (c, name[-40:], r.get("Counter_Value")))
PY
  else
    echo "$C: no counter file" >> $OUT/calib_counters.txt; tail -3 $OUT/p_$C.log >> $OUT/calib_counters.txt
  fi
  rm -rf $OUT/p_$C
done
cat $OUT/calib_launches.txt $OUT/calib_counters.txt
