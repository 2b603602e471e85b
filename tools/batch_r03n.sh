#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03n; mkdir -p $O; cd $R
timeout 600 python tools/stedc_check.py 200 777 1500 > $O/stedc_check.txt 2>&1
grep "driver 0" $O/stedc_check.txt
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -5
cd /tmp && export TMPDIR=/tmp
SDPSR_TOOL_FLAGS=256 rocprofv3 --kernel-trace --stats --output-format csv -d $O/eig1024 -o eig -- python3 $R/tools/eig_only.py 1024 0 random > $O/eig1024.log 2>&1
grep -E "dc_|gemm|bt_" $O/eig1024/eig_kernel_stats.csv | cut -c1-70,150-250 | head -20
