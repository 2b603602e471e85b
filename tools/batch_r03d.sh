#!/bin/bash
# class-list insert pass: parity tests, then A/B against the hash pass (flag 2048 = SDPSR_FLAG_REFINE_NO_CLASSLIST)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03d; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -6 > $O/gpu_tests.log
for w in closed_scheme theta_c32xk128 theta_er7xk72; do
  for f in 0 2048 0 2048; do
    timeout 300 python bench.py --steps 30 --warmup 5 --cpu-n 0 --skip-roofline --workload $w --flags $f 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w flags=$f', d['value'], d['ms_per_step'], d.get('phase_ms'))" >> $O/ab.txt
  done
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_theta -o bench -- python3 $R/bench.py --steps 20 --warmup 3 --skip-roofline --workload theta_c32xk128 > $O/bench_theta_under_rocprof.json 2> /dev/null
cat $O/gpu_tests.log; cat $O/ab.txt
