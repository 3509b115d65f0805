OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s18; mkdir -p $OUT
python tools/gpu_te_dma_interference.py 2>&1 | grep -v amdgpu.ids | tee $OUT/te_dma_interference.txt
for c in "pinned pinned" "pageable pageable" "pinned pinned" "pageable pageable"; do python tools/te_host_calls.py $c 14 2>&1 | tail -1; done | tee $OUT/te_host_calls.txt
timeout 1500 python -m pytest tests/test_gpu_bench_contract.py -m gpu -x -q 2>&1 | tail -6 | tee $OUT/pytest_contract.txt
