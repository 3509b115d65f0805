OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s6; mkdir -p $OUT
timeout 600 python tools/gpu_te_msg_lds.py 2>&1 | grep "host path" > $OUT/te_hostpath.txt; cat $OUT/te_hostpath.txt
cd /tmp && export TMPDIR=/tmp
for combo in "pinned pinned"; do
  tag=$(echo $combo | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/tr_$tag -o t -- python $GRAFT_REPO_ROOT/tools/te_host_trace_driver.py $combo > $OUT/trace_$tag.log 2>&1
  tail -3 $OUT/trace_$tag.log
  python $GRAFT_REPO_ROOT/tools/trace_timeline.py $OUT/tr_$tag 36 > $OUT/timeline_$tag.txt; cat $OUT/timeline_$tag.txt
  rm -rf $OUT/tr_$tag
done
cd $GRAFT_REPO_ROOT && timeout 600 python -m pytest tests/test_gpu_canaries.py tests/test_gpu_curves.py -m gpu -x -q 2>&1 | tail -3
