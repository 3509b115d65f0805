# round 4 session 1: parity of the LDS-staged accumulate kernel, then its A/B
mkdir -p gpurun_out/r04_s1
timeout 900 python -m pytest tests/test_gpu_curves.py tests/test_gpu_canaries.py tests/test_gpu_merkle.py tests/test_gpu_features.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r04_s1/pytest_curves.txt
cat gpurun_out/r04_s1/pytest_curves.txt
for A in 0 1; do AKP_TE_MSG_LDS=$A timeout 600 python tools/gpu_te_msg_lds.py > gpurun_out/r04_s1/te_msg_lds_arm$A.txt 2>&1; cat gpurun_out/r04_s1/te_msg_lds_arm$A.txt; done
