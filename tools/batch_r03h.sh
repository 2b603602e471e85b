#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03h; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -6 > $O/gpu_tests.log
for f in 0 12288; do echo "== flags=$f" >> $O/config_times.txt; SDPSR_TOOL_FLAGS=$f timeout 600 python tools/config_times.py >> $O/config_times.txt 2>&1; done
cat $O/gpu_tests.log $O/config_times.txt
