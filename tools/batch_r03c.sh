#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03c; mkdir -p $O; cd $R
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or verify or refine or sort or trajectory or channel" 2>&1 | tail -4 > $O/gpu_tests.log
python bench.py --steps 30 --warmup 5 --cpu-n 0 > $O/bench_default.json 2> $O/bench_default.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o bench -- python3 $R/bench.py --steps 30 --warmup 5 --skip-roofline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_theta -o bench -- python3 $R/bench.py --steps 20 --warmup 3 --skip-roofline --workload theta_c32xk128 > $O/bench_theta_under_rocprof.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_er7 -o bench -- python3 $R/bench.py --steps 20 --warmup 3 --skip-roofline --workload theta_er7xk72 > $O/bench_er7_under_rocprof.json 2> /dev/null
cd $R
for n in 900 1024 2048; do for drv in 0 1 2; do python tools/eig_only.py $n $drv random 2>&1 | grep "syev n=" | tail -1 >> $O/eig_drivers.txt; done; done
tail -3 $O/gpu_tests.log
