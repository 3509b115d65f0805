OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s13; mkdir -p $OUT
tools/clock_probe | tee $OUT/clock_probe.txt
timeout 1500 python -m pytest tests/test_gpu_multi_slots.py tests/test_gpu_tree_handle.py tests/test_gpu_lifetimes.py -m gpu -x -q 2>&1 | tail -25 | tee $OUT/pytest.txt
