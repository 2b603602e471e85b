#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel statistics + separate PMC passes of this
# round, written under gpurun_out/r02/ (copied into profiles/ by tools/pmc_summarize.py + cp).
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# (a) default bench command, timed steps only (clean kernel statistics)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o bench -- python3 $R/bench.py --steps 30 --warmup 5 --skip-roofline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
# (b) the other two instances
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_theta -o bench -- python3 $R/bench.py --steps 20 --warmup 3 --skip-roofline --workload theta_c32xk128 > $O/bench_theta_under_rocprof.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_er7 -o bench -- python3 $R/bench.py --steps 20 --warmup 3 --skip-roofline --workload theta_er7xk72 > $O/bench_er7_under_rocprof.json 2> /dev/null
# (c) dense driver (the launch sequence of the tridiagonalisation is graph-replayed in production;
#     rocprofv3 on this image crashes on 8000-node graphs, so the profile runs the direct launches)
export SDPSR_NO_GRAPH=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/dense -o dense -- python3 $R/bench.py --steps 3 --warmup 1 --skip-roofline --eig-driver 4 > $O/dense_under_rocprof.json 2> /dev/null
unset SDPSR_NO_GRAPH
# (d) N = 8192 squares
rocprofv3 --kernel-trace --stats --output-format csv -d $O/sq8192 -o sq -- python3 $R/tools/kernel_roofline.py 8192 > $O/sq8192.json 2> /dev/null
# (e) PMC passes (separate runs per counter, kernel-trace only)
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_i8tri_$C -o p -- python3 $R/tools/pmc_probe.py 0 4096 104 > /dev/null 2>&1
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_symv_$C -o p -- python3 $R/tools/pmc_probe.py 5 4096 > /dev/null 2>&1
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_f32_8192_$C -o p -- python3 $R/tools/pmc_probe.py 1 8192 1 > /dev/null 2>&1
done
ls -R $O | head -60
# (f) the default bench command itself (full JSON line with roofline + cpu_baseline)
cd $R && python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
# (g) clocks under the hot kernels, the pure-MFMA probe, the int8 diagnostic builds, configs[1..2]
python3 tools/clock_under_kernels.py > $O/clock_under_kernels.txt 2>&1
hipcc -O3 --offload-arch=gfx950 tools/probes/mfma_clock_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe > $O/mfma_clock_probe.txt 2>&1
for d in 0 1 2 3; do echo "SDPSR_W4_DIAG=$d" >> $O/i8_w4_diag.txt; SDPSR_GEMM_I8_W4=1 SDPSR_W4_DIAG=$d python3 tools/i8_tri_time.py >> $O/i8_w4_diag.txt 2>&1; done
python3 tools/config_times.py > $O/config_times.txt 2>&1
python3 tools/config2_bd_phases.py > $O/config2_bd_phases.txt 2>&1
python3 tools/small_syev_time.py > $O/small_syev_time.txt 2>&1
# (h) the label product on the matrix cores: per shape, A/B against the VALU form, shader clock, the fp64 MFMA probe
python3 tools/label_product_time.py > $O/label_product.txt 2>&1
SDPSR_SPMM_VALU=1 python3 tools/label_product_time.py >> $O/label_product.txt 2>&1
python3 tools/label_product_clock.py >> $O/label_product.txt 2>&1
hipcc -O3 --offload-arch=gfx950 tools/probes/mfma_f64_probe.hip -o /tmp/mfma_f64_probe 2> /dev/null && /tmp/mfma_f64_probe > $O/mfma_f64_probe.txt 2>&1
