OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s23; mkdir -p $OUT
for A in 0 1; do AKP_POSEIDON_STAGED_IO=$A python tools/gpu_poseidon_hostpath.py 2>&1 | grep -v amdgpu.ids; done | tee $OUT/poseidon_hostpath_staged_ab.txt
timeout 600 python -m pytest tests/test_gpu_poseidon.py tests/test_gpu_canaries.py -m gpu -x -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" | tail -3
