#!/bin/bash
# hybrid panel/row tridiagonalisation at N = 4096, per-kernel profile of the row form, class-list insert with 4 candidates
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03f; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "syev or eigen" 2>&1 | tail -6 > $O/gpu_tests.log
for f in 0 4096; do echo "flags=$f" >> $O/sytrd_time.txt; SDPSR_TOOL_FLAGS=$f timeout 300 python tools/sytrd_time.py 1024 2048 3072 4096 >> $O/sytrd_time.txt 2>&1; done
for w in theta_c32xk128 theta_er7xk72; do
  for f in 0 2048 0 2048; do
    timeout 300 python bench.py --steps 30 --warmup 5 --cpu-n 0 --skip-roofline --workload $w --flags $f 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w flags=$f', d['value'], d['ms_per_step'])" >> $O/ab.txt
  done
done
cd /tmp && export TMPDIR=/tmp
SDPSR_TOOL_FLAGS=256 rocprofv3 --kernel-trace --stats --output-format csv -d $O/eig1024 -o eig -- python3 $R/tools/eig_only.py 1024 0 random > $O/eig1024.log 2>&1
SDPSR_TOOL_FLAGS=256 rocprofv3 --kernel-trace --stats --output-format csv -d $O/eig2048 -o eig -- python3 $R/tools/eig_only.py 2048 0 random > $O/eig2048.log 2>&1
cat $O/gpu_tests.log $O/sytrd_time.txt $O/ab.txt
