#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03m; mkdir -p $O; cd $R
timeout 600 python tools/stedc_check.py 200 777 1024 > $O/stedc_check.txt 2>&1
cat $O/stedc_check.txt | tail -60
for n in 900 1024 2048; do for k in random degenerate; do
  echo "== n=$n $k" >> $O/eig.txt; SDPSR_DEBUG=1 timeout 300 python tools/eig_only.py $n 0 $k 2>&1 | tail -16 >> $O/eig.txt
done; done
grep -E "==|syev n|resid |eigval|orth|solver done" $O/eig.txt | tail -50
