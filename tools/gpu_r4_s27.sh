OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s27; mkdir -p $OUT
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -2 $OUT/bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r04_s27/bench.json') if l.startswith('{')][-1])
print('value %.4g' % d['value']); print(json.dumps(d['sweep']['points'])[:1500]); print(json.dumps(d['merkle']['one_process_c_abi']['resident_tree'])[:500])
PY
timeout 900 python -m pytest tests/test_gpu_tree_handle.py tests/test_gpu_bench_contract.py -m gpu -x -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" | tail -3
