#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03l; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or refine" 2>&1 | tail -3
for w in theta_c32xk128 theta_er7xk72; do
  for f in 0 2048 0 2048; do
    timeout 300 python bench.py --steps 30 --warmup 5 --cpu-n 0 --skip-roofline --workload $w --flags $f 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w flags=$f', d['value'], d['ms_per_step'])"
  done
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_theta -o bench -- python3 $R/bench.py --steps 20 --warmup 3 --skip-roofline --workload theta_c32xk128 > $O/bench_theta_under_rocprof.json 2> /dev/null
grep -E "refine_insert" $O/bench_theta/bench_kernel_stats.csv | cut -c1-60,150-260
