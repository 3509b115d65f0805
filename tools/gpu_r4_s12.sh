OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s12; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_multi_slots.py tests/test_gpu_tree_handle.py -m gpu -x -q 2>&1 | tail -25 | tee $OUT/pytest.txt
