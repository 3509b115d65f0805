OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s7; mkdir -p $OUT
timeout 600 python tools/gpu_te_msg_lds.py 2>&1 | grep "host path" > $OUT/te_hostpath.txt; cat $OUT/te_hostpath.txt
timeout 900 python -m pytest tests/test_gpu_lifetimes.py tests/test_gpu_canaries.py tests/test_gpu_tree_handle.py tests/test_gpu_poseidon.py -m gpu -x -q 2>&1 | tail -8 | tee $OUT/pytest.txt
