#!/bin/bash
# one-barrier insert pass: parity, A/B against the barrier form (flag 2048); configs[1] admissible_subspace after the overflow fix
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03k; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -6 > $O/gpu_tests.log
cat $O/gpu_tests.log
for w in closed_scheme theta_c32xk128 theta_er7xk72; do
  for f in 0 2048 0 2048; do
    timeout 300 python bench.py --steps 30 --warmup 5 --cpu-n 0 --skip-roofline --workload $w --flags $f 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w flags=$f', d['value'], d['ms_per_step'])" >> $O/ab.txt
  done
done
cat $O/ab.txt
timeout 300 python tools/config1_adm_trace.py 2>&1 | grep admissible
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_theta -o bench -- python3 $R/bench.py --steps 20 --warmup 3 --skip-roofline --workload theta_c32xk128 > $O/bench_theta_under_rocprof.json 2> /dev/null
grep -E "refine_insert" $O/bench_theta/bench_kernel_stats.csv | cut -c1-60,150-260
