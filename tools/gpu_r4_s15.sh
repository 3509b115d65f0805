OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s15; mkdir -p $OUT
export AKP_LIB=$GRAFT_REPO_ROOT/crypto_primitives_amd/lib/libakp_testhooks.so
echo "== all, faulthandler"; PYTHONFAULTHANDLER=1 timeout 600 python -X dev -m pytest tests/test_gpu_multi_slots.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -40; echo "rc=$?"
echo "== build_logic + poseidon"; PYTHONFAULTHANDLER=1 timeout 600 python -m pytest tests/test_gpu_multi_slots.py -m gpu -x -q -p no:cacheprovider -k "sharded_build_logic or resident_tree_poseidon" 2>&1 | tail -5
echo "== poseidon + bytes"; PYTHONFAULTHANDLER=1 timeout 600 python -m pytest tests/test_gpu_multi_slots.py -m gpu -x -q -p no:cacheprovider -k "byte_digests or resident_tree_poseidon" 2>&1 | tail -5
echo "== build_logic + bytes"; PYTHONFAULTHANDLER=1 timeout 600 python -m pytest tests/test_gpu_multi_slots.py -m gpu -x -q -p no:cacheprovider -k "byte_digests or sharded_build_logic" 2>&1 | tail -5
