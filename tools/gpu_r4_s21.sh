OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s21; mkdir -p $OUT
timeout 600 python tools/bench_proofs.py --config poseidon > $OUT/proofs.json 2>$OUT/proofs.err; tail -2 $OUT/proofs.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_s21/proofs.json').read().strip().splitlines()[-1])['poseidon']
print(json.dumps(d['verify_all_leaves_dev'])); print(json.dumps(d['verify_paths']))
PY
timeout 900 python -m pytest tests/test_abi.py tests/test_gpu_tree_handle.py tests/test_gpu_multi_slots.py tests/test_gpu_poseidon.py -m gpu -x -q 2>&1 | tail -4
