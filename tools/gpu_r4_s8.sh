OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s8; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for combo in "pinned pinned"; do
  tag=$(echo $combo | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/tr_$tag -o t -- python $GRAFT_REPO_ROOT/tools/te_host_trace_driver.py $combo > $OUT/trace_$tag.log 2>&1
  tail -3 $OUT/trace_$tag.log
  python $GRAFT_REPO_ROOT/tools/trace_timeline.py $OUT/tr_$tag 40 > $OUT/timeline_$tag.txt; cat $OUT/timeline_$tag.txt
  rm -rf $OUT/tr_$tag
done
cd $GRAFT_REPO_ROOT; for i in 1 2 3; do python tools/te_host_trace_driver.py pinned pinned 2>&1 | tail -2; done
