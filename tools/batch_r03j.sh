#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03j; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/adm -o adm -- python3 $R/tools/config1_adm_trace.py > $O/adm.log 2>&1
cat $O/adm.log | tail -8
