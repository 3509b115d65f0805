#!/bin/bash
TAG=${1:-a}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
timeout 300 tools/microbench > $OUT/microbench.log 2>&1; grep -A30 "asm vs C++" $OUT/microbench.log; tail -3 $OUT/microbench.log
