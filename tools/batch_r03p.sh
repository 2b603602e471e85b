#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03p; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -5
for n in 900 1024 2048; do SDPSR_DEBUG=1 python tools/eig_only.py $n 0 random 2>&1 | grep -E "done|syev n|resid |orth" | tail -7; done
python tools/eig_only.py 4096 0 random 2>&1 | grep -E "syev n|resid |orth" | tail -4
SDPSR_DEBUG=1 timeout 600 python tools/config_times.py 2>&1 | grep -E "config|eigen_dec|blockDiag" | tail -12
