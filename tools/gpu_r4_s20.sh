OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s20; mkdir -p $OUT
for c in none torch torch,heat; do python tools/te_host_calls_ctx.py $c 2>&1 | grep conditions; done | tee $OUT/te_host_calls_conditions.txt
