OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s22; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" | tail -8 | tee $OUT/pytest_gpu_full.txt
