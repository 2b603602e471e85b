#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03g; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "syev or eigen or block_diag" 2>&1 | tail -6 > $O/gpu_tests.log
SDPSR_TOOL_FLAGS=0 timeout 300 python tools/sytrd_time.py 512 1024 2048 4096 >> $O/sytrd_time.txt 2>&1
for n in 300 900 1024 2048; do for k in random degenerate; do
  echo "== n=$n $k" >> $O/eig.txt; timeout 300 python tools/eig_only.py $n 0 $k 2>&1 | tail -9 >> $O/eig.txt
done; done
cat $O/gpu_tests.log $O/sytrd_time.txt; grep -E "==|syev n|resid |eigval" $O/eig.txt
