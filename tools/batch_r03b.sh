#!/bin/bash
# round-3 measurement batch (GPU box): tests, default bench, insert-pass tuning, power traces, config times
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03b; mkdir -p $O; cd $R
python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/gpu_tests.log
python bench.py --steps 30 --warmup 5 --cpu-n 0 > $O/bench_default.json 2> $O/bench_default.err
for W in 1 2 3 4 5 6; do
  for WL in theta_c32xk128 closed_scheme; do
    python bench.py --steps 20 --warmup 3 --skip-roofline --workload $WL --insert-wgs $W > $O/tune_${WL}_w$W.json 2> /dev/null
  done
done
python bench.py --steps 20 --warmup 3 --skip-roofline --flags 1024 > $O/bench_full_basis_image.json 2> /dev/null
python bench.py --steps 20 --warmup 3 --skip-roofline --flags 512 > $O/bench_no_verify.json 2> /dev/null
python bench.py --steps 20 --warmup 3 --skip-roofline --flags 128 > $O/bench_small_eigen_device.json 2> /dev/null
python bench.py --steps 20 --warmup 3 --skip-roofline --channels 4 > $O/bench_channels4.json 2> /dev/null
python tools/power_trace.py 0 4096 102 3 > $O/power_i8_product_launch.txt 2>&1
python tools/power_trace.py 0 4096 104 3 > $O/power_i8_4ch.txt 2>&1
python tools/power_trace.py 0 8192 4 3 > $O/power_i8_n8192.txt 2>&1
python tools/power_trace.py 1 8192 1 3 > $O/power_f32_n8192.txt 2>&1
python tools/power_trace.py 2 4096 1 3 > $O/power_f64.txt 2>&1
python tools/config_times.py > $O/config_times.txt 2>&1
tail -3 $O/gpu_tests.log
