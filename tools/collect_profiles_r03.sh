#!/bin/bash
# Round 3, on the GPU box (through gpurun): rocprofv3 kernel statistics of the bench instances, the
# dense driver, separate PMC passes (HBM traffic of the int8 launch; instruction / LDS / stall counters of
# the refinement's insert pass), power + clock traces.  Output under gpurun_out/r03/; tools/copy_profiles_r03.py
# turns it into profiles/r03_*.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
WHAT=${1:-all}
if [ "$WHAT" = all ] || [ "$WHAT" = stats ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o bench -- python3 $R/bench.py --steps 30 --warmup 5 --skip-roofline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_theta -o bench -- python3 $R/bench.py --steps 20 --warmup 3 --skip-roofline --workload theta_c32xk128 > $O/bench_theta_under_rocprof.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_er7 -o bench -- python3 $R/bench.py --steps 20 --warmup 3 --skip-roofline --workload theta_er7xk72 > $O/bench_er7_under_rocprof.json 2> /dev/null
# dense driver: per-kernel statistics need the launches one by one (--no-graph = SDPSR_FLAG_NO_GRAPH)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/dense -o dense -- python3 $R/bench.py --steps 3 --warmup 1 --skip-roofline --eig-driver 4 --no-graph > $O/dense_under_rocprof.json 2> /dev/null
fi
if [ "$WHAT" = all ] || [ "$WHAT" = pmc ]; then
rocprofv3 -L > $O/counters_list.txt 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_i8tri_$C -o p -- python3 $R/tools/pmc_probe.py 0 4096 102 > /dev/null 2>&1
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_insert_$C -o p -- python3 $R/bench.py --steps 4 --warmup 2 --skip-roofline --workload theta_c32xk128 > /dev/null 2>&1
  for NN in 1024 2048; do SDPSR_TOOL_FLAGS=256 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_rows${NN}_$C -o p -- python3 $R/tools/eig_only.py $NN 0 random > /dev/null 2>&1; done
done
# the insert pass of the refinement (refine_insert_kernel<SrcJoint<2,2>,8,1024> in theta_c32xk128): instruction mix and stalls
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/pmc_insert_A -o p -- python3 $R/bench.py --steps 4 --warmup 2 --skip-roofline --workload theta_c32xk128 > $O/pmc_insert_A.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_insert_B -o p -- python3 $R/bench.py --steps 4 --warmup 2 --skip-roofline --workload theta_c32xk128 > $O/pmc_insert_B.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_insert_G -o p -- python3 $R/bench.py --steps 4 --warmup 2 --skip-roofline --workload theta_c32xk128 > /dev/null 2>&1
fi
if [ "$WHAT" = all ] || [ "$WHAT" = power ]; then
cd $R
( echo "# socket power (hwmon power1_input of the GPU in use, by PCI address) and sclk while ONE kernel runs in a ~3 s loop"; echo "# tools/power_trace.py <kind> <n> <aux>: kind 0 int8 square (aux 102 = the product launch: 2 channels, lower-triangle tiles; 104 = 4 channels; n 8192 aux 4 = full squares), 1 fp32 square, 2 fp64 GEMM" ) > $O/power_under_kernels.txt
for A in "0 4096 102" "0 4096 104" "0 8192 4" "1 8192 1" "2 4096 1"; do python3 tools/power_trace.py $A 3 >> $O/power_under_kernels.txt 2>&1; done
python3 tools/clock_under_kernels.py > $O/clock_under_kernels.txt 2>&1
fi
if [ "$WHAT" = all ] || [ "$WHAT" = bench ]; then
cd $R
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
python3 bench.py --steps 20 --warmup 3 --skip-roofline --flags 1024 > $O/ab_full_basis_image.json 2> /dev/null
python3 bench.py --steps 20 --warmup 3 --skip-roofline --flags 512 > $O/ab_no_verify_shortcut.json 2> /dev/null
python3 bench.py --steps 20 --warmup 3 --skip-roofline --flags 128 > $O/ab_small_eigen_on_device.json 2> /dev/null
python3 bench.py --steps 20 --warmup 3 --skip-roofline --channels 4 > $O/ab_channels4.json 2> /dev/null
python3 bench.py --steps 20 --warmup 3 --skip-roofline --channels 4 --workload theta_c32xk128 > $O/ab_channels4_theta.json 2> /dev/null
python3 tools/config_times.py > $O/config_times.txt 2>&1
python3 tools/config2_bd_phases.py > $O/config2_bd_phases.txt 2>&1
python3 tools/stedc_check.py 200 777 1024 > $O/stedc_check.txt 2>&1
for f in 0 4096; do echo "sdpsr_opts.flags = $f (4096 = SDPSR_FLAG_SYTRD_PANELS: the panel form at every order)" >> $O/sytrd_time.txt; SDPSR_TOOL_FLAGS=$f python3 tools/sytrd_time.py 512 1024 2048 3072 4096 >> $O/sytrd_time.txt 2>&1; done
( cd /tmp; SDPSR_TOOL_FLAGS=256 rocprofv3 --kernel-trace --stats --output-format csv -d $O/dense1024 -o eig -- python3 $R/tools/eig_only.py 1024 0 random > $O/dense1024.log 2>&1 )
for n in 900 1024 2048 4096; do for drv in 0 5 1; do python3 tools/eig_only.py $n $drv random 2>&1 | grep "syev n=" | tail -1 >> $O/eig_drivers.txt; done; done
fi
if [ "$WHAT" = all ] || [ "$WHAT" = seeds ]; then
cd $R
python3 tools/stress_seeds.py 300 > $O/stress_seeds.txt 2>&1
python3 tools/big_instance_seeds.py 60 1 > $O/big_instance_seeds.txt 2>&1
python3 tools/big_instance_seeds.py 30 0 >> $O/big_instance_seeds.txt 2>&1
python3 tools/big_instance_seeds.py 20 1 1 >> $O/big_instance_seeds.txt 2>&1
python3 tools/big_instance_seeds.py 20 1 512 >> $O/big_instance_seeds.txt 2>&1
python3 tools/big_instance_seeds.py 20 1 0 4 >> $O/big_instance_seeds.txt 2>&1
( python3 tools/bd_failure_compare.py device 8 5000 4 0; python3 tools/bd_failure_compare.py device 8 5000 4 64; python3 tools/bd_failure_compare.py device 8 5000 6 0; python3 tools/bd_failure_compare.py device 72 3000 0 0; python3 tools/bd_failure_rate.py 4096 2000 commutative 2>&1 | tail -3; python3 tools/bd_failure_rate.py 4104 3000 er7 2>&1 | tail -3 ) > $O/bd_failure_rates.txt 2>&1
fi
ls $O | head -50
