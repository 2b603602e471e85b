#!/bin/bash
# row form of the tridiagonalisation (n <= 2048): parity, then timings against the panel form (flag 4096)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03e; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "syev or eigen or block_diag" 2>&1 | tail -6 > $O/gpu_tests.log
for n in 300 900 1024 2048; do for k in random degenerate; do
  for f in 0 4096; do echo "== n=$n $k flags=$f" >> $O/eig.txt; SDPSR_TOOL_FLAGS=$f timeout 300 python tools/eig_only.py $n 0 $k 2>&1 | tail -9 >> $O/eig.txt; done
done; done
for f in 0 4096; do echo "flags=$f" >> $O/sytrd_time.txt; SDPSR_TOOL_FLAGS=$f timeout 300 python tools/sytrd_time.py 512 1024 2048 >> $O/sytrd_time.txt 2>&1; done
cat $O/gpu_tests.log; cat $O/sytrd_time.txt; grep -E "==|syev n|resid |eigval" $O/eig.txt
