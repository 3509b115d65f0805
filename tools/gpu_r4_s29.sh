OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s29; mkdir -p $OUT
AKP_LIB=$GRAFT_REPO_ROOT/crypto_primitives_amd/lib/libakp_testhooks.so timeout 900 python -m pytest tests/test_gpu_multi_slots.py -m gpu -x -q -p no:cacheprovider --durations=15 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" | tail -24 | tee $OUT/pytest_hooks_child.txt
