#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03o; mkdir -p $O; cd $R
timeout 600 python tools/stedc_check.py 200 777 1024 1500 > $O/stedc_check.txt 2>&1
grep "driver 0" $O/stedc_check.txt | awk '{print $2,$3,$4,$5,$7,$9,$11,$13}' | column -t
cd /tmp && export TMPDIR=/tmp
SDPSR_TOOL_FLAGS=256 rocprofv3 --kernel-trace --stats --output-format csv -d $O/eig1024 -o eig -- python3 $R/tools/eig_only.py 1024 0 random > $O/eig1024.log 2>&1
grep -E "dc_|bt_" $O/eig1024/eig_kernel_stats.csv | cut -c1-60,100-190 | head -20
cd $R; SDPSR_DEBUG=1 python tools/eig_only.py 1024 0 random 2>&1 | grep -E "solver done|syev n" | tail -4
