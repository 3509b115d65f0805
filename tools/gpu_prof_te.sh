#!/bin/bash
TAG=${1:-p}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_all -o trace -- python $OLDPWD/tools/bench_all.py --max-log2 20 > $OUT/bench_all.jsonl 2> $OUT/err.log)
head -20 $OUT/prof_all/trace_kernel_stats.csv | cut -c1-200
python - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/prof_all/trace_kernel_trace.csv")))
import collections
agg=collections.defaultdict(list)
for r in rows:
    d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6
    if d>0.5: agg[(r["Kernel_Name"][:60], r["Grid_Size"], r.get("VGPR_Count"), r.get("SGPR_Count"), r.get("Scratch_Size"), r.get("LDS_Block_Size"))].append(d)
for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1]))[:25]:
    print("%.3f ms x%d  %s" % (min(v), len(v), k))
PY
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -o trace -- python $OLDPWD/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench.log 2>&1)
head -6 $OUT/prof_bench/trace_kernel_stats.csv | cut -c1-220
find $OUT -name "*kernel_trace.csv" -size +2M -delete
