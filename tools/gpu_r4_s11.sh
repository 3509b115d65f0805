OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s11; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_tree_handle.py -m gpu -x -q -k "lane_walk or poseidon_vs_oracle" 2>&1 | tail -5 | tee $OUT/pytest.txt
for W in 0 1; do for M in 16 18; do
  AKP_VERIFY_WALK=$W timeout 300 python tools/bench_proofs.py --config poseidon --log2-m $M 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])['poseidon']['verify_paths']
print('walk=$W m=2^$M  device %.3f ms  wall %.3f ms  hashes/s(device) %.4g  all_accepted %s neg_control %s' % (d['device_ms'], d['wall_ms'], d['hashes_per_s_device'], d['all_accepted'], d['negative_control_rejected_only_the_wrong_leaf']))"
done; done | tee $OUT/verify_walk_ab.txt
