#!/bin/bash
# round 6: the arms of gpu_r6_gate_grid.py
# last-first in processes that follow a process with HBM-sized tables (is the gated arm slow there, or only early?)
O=gpurun_out/r06_s51; mkdir -p $O
export AKP_LIB=$PWD/crypto_primitives_amd/lib/libakp_testhooks.so
timeout 200 python tools/gpu_r6_gate_grid.py 0 hbm 2>/dev/null | tee -a $O/order.jsonl
GATE_GRID_REVERSE=1 timeout 200 python tools/gpu_r6_gate_grid.py 1 hbm 2>/dev/null | tee -a $O/order.jsonl
timeout 200 python tools/gpu_r6_gate_grid.py 1 hbm 2>/dev/null | tee -a $O/order.jsonl
GATE_GRID_REVERSE=1 timeout 200 python tools/gpu_r6_gate_grid.py 2 hbm 2>/dev/null | tee -a $O/order.jsonl
