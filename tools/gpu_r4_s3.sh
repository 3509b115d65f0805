mkdir -p gpurun_out/r04_s3
timeout 900 python -m pytest tests/test_gpu_curves.py tests/test_gpu_canaries.py tests/test_gpu_features.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r04_s3/pytest_curves.txt
cat gpurun_out/r04_s3/pytest_curves.txt
AKP_TE_MSG_LDS=1 timeout 600 python tools/gpu_te_msg_lds.py > gpurun_out/r04_s3/te_msg_lds_arm1.txt 2>&1; grep "host path" gpurun_out/r04_s3/te_msg_lds_arm1.txt
