OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s9; mkdir -p $OUT
for L in 16 17 18 19; do for i in 1 2; do echo "chunk 2^$L: $(AKP_TE_PIPE_CHUNK_LOG2=$L python tools/gpu_te_msg_lds.py 2>&1 | grep 'pinned in/out')"; done; done | tee $OUT/pipe_chunk_sweep.txt
