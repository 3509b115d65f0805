OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s16; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_multi_slots.py tests/test_gpu_tree_handle.py tests/test_gpu_lifetimes.py -m gpu -x -q 2>&1 | tail -8 | tee $OUT/pytest.txt
cd /tmp && export TMPDIR=/tmp
for W in 0 1; do
  AKP_VERIFY_WALK=$W timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_walk$W -o p -- python $GRAFT_REPO_ROOT/tools/bench_proofs.py --config poseidon --log2-m 16 > $OUT/proofs_walk$W.json 2>/dev/null
  f=$(find $OUT/prof_walk$W -name "*kernel_stats.csv" | head -1); cp $f $OUT/proofs_poseidon_kernel_stats_walk$W.csv; head -12 $f
  rm -rf $OUT/prof_walk$W
done
