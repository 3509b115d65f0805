OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s28; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" | tail -20 | tee $OUT/pytest_gpu_full.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $OUT/smoke.txt
