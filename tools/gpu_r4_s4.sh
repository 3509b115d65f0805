mkdir -p gpurun_out/r04_s4
for Z in 0 1; do echo "# AKP_TE_ZERO_COPY_IN=$Z"; AKP_TE_ZERO_COPY_IN=$Z timeout 600 python tools/gpu_te_msg_lds.py 2>&1 | grep "host path"; done > gpurun_out/r04_s4/te_hostpath_ab.txt 2>&1
cat gpurun_out/r04_s4/te_hostpath_ab.txt
