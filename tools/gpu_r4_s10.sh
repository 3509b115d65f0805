OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s10; mkdir -p $OUT
(for P in 1 0; do for L in 17 18; do echo "prio=$P chunk=2^$L: $(AKP_TE_PIPE_PRIO=$P AKP_TE_PIPE_CHUNK_LOG2=$L python tools/te_host_calls.py pinned pinned 14 2>&1 | tail -1)"; done; done
python tools/te_host_calls.py pageable pageable 14 2>&1 | tail -1
python tools/te_host_calls.py pinned pageable 14 2>&1 | tail -1
python tools/te_host_calls.py pageable pinned 14 2>&1 | tail -1) | tee $OUT/te_host_calls.txt
