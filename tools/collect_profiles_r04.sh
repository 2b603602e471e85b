#!/bin/bash
# Round 4, on the GPU box (through gpurun): counter passes under the two squares (int8 product launch, fp32 N = 8192),
# kernel statistics of the bench instances at N = 4096 and N = 8192, the dense driver, the many-classes refinement,
# seed sweeps.  Output under gpurun_out/r04/; tools/copy_profiles_r04.py turns it into profiles/r04_*.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
WHAT=${1:-all}
pmc_pass() {  # tag, counters, program args...
  local tag=$1; shift; local ctr=$1; shift
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_$tag -o p -- python3 "$@" > $O/pmc_$tag.log 2>&1
}
if [ "$WHAT" = all ] || [ "$WHAT" = pmc_squares ]; then
  # int8 product launch (2 channels, lower triangle): the persistent kernel (aux 102) and the 128 x 128 tiles it replaced
  # (aux 202); the fp32 square at N = 8192.  MFMA group, LDS / stall group, clock, HBM traffic -- every group its own pass
  for K in "i8sym 0 4096 102" "i8tri128 0 4096 202" "f32n8192 1 8192 1"; do
    set -- $K; T=$1; shift
    pmc_pass ${T}_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CU_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA" $R/tools/pmc_probe.py "$@"
    pmc_pass ${T}_lds "SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY" $R/tools/pmc_probe.py "$@"
    pmc_pass ${T}_grbm "GRBM_GUI_ACTIVE" $R/tools/pmc_probe.py "$@"
    pmc_pass ${T}_FETCH_SIZE "FETCH_SIZE" $R/tools/pmc_probe.py "$@"
    pmc_pass ${T}_WRITE_SIZE "WRITE_SIZE" $R/tools/pmc_probe.py "$@"
  done
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/squares8192 -o sq -- python3 $R/tools/pmc_probe.py 1 8192 1 > $O/squares8192.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/squares_i8sym -o sq -- python3 $R/tools/pmc_probe.py 0 4096 102 > $O/squares_i8sym.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/refine_bucket -o rb -- python3 $R/tools/pmc_probe.py 3 4096 8388608 > $O/refine_bucket.log 2>&1
  for C in FETCH_SIZE WRITE_SIZE; do pmc_pass refine_bucket_$C $C $R/tools/pmc_probe.py 3 4096 8388608; done
fi
if [ "$WHAT" = all ] || [ "$WHAT" = stats ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o bench -- python3 $R/bench.py --steps 30 --warmup 5 --skip-roofline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_theta -o bench -- python3 $R/bench.py --steps 20 --warmup 3 --skip-roofline --workload theta_c32xk128 > $O/bench_theta_under_rocprof.json 2> /dev/null
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_er7 -o bench -- python3 $R/bench.py --steps 20 --warmup 3 --skip-roofline --workload theta_er7xk72 > $O/bench_er7_under_rocprof.json 2> /dev/null
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/dense -o dense -- python3 $R/bench.py --steps 3 --warmup 1 --skip-roofline --eig-driver 4 --no-graph > $O/dense_under_rocprof.json 2> /dev/null
fi
if [ "$WHAT" = all ] || [ "$WHAT" = n8192 ]; then
  cd $R
  python3 bench.py --n 8192 --steps 10 --warmup 2 --cpu-n 0 > $O/bench_n8192.json 2> $O/bench_n8192.err
  ( cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench8192 -o bench -- python3 $R/bench.py --n 8192 --steps 10 --warmup 2 --skip-roofline > $O/bench_n8192_under_rocprof.json 2> /dev/null )
fi
if [ "$WHAT" = all ] || [ "$WHAT" = bench ]; then
  cd $R
  python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
  python3 bench.py --steps 20 --warmup 3 --skip-roofline --square-kernel 1 > $O/ab_square_kernel_128tiles.json 2> /dev/null
  python3 bench.py --steps 20 --warmup 3 --skip-roofline --square-kernel 1 --workload theta_c32xk128 > $O/ab_square_kernel_128tiles_theta.json 2> /dev/null
  python3 bench.py --steps 20 --warmup 3 --skip-roofline --flags 1024 > $O/ab_full_basis_image.json 2> /dev/null
  python3 bench.py --steps 20 --warmup 3 --skip-roofline --restarts-per-gpu 2 > $O/restarts_per_gpu_2.json 2> /dev/null
  python3 tools/config_times.py > $O/config_times.txt 2>&1
  python3 tools/stedc_check.py 200 777 1024 > $O/stedc_check.txt 2>&1
  SDPSR_TOOL_FLAGS=0 python3 tools/sytrd_time.py 512 1024 2048 3072 4096 > $O/sytrd_time.txt 2>&1
  for n in 1024 2048 4096; do for drv in 0 1; do python3 tools/eig_only.py $n $drv random 2>&1 | grep "syev n=" | tail -1 >> $O/eig_drivers.txt; done; done
fi
if [ "$WHAT" = all ] || [ "$WHAT" = seeds ]; then
  cd $R
  python3 tools/stress_seeds.py 300 > $O/stress_seeds.txt 2>&1
  python3 tools/big_instance_seeds.py 60 1 > $O/big_instance_seeds.txt 2>&1
  python3 tools/big_instance_seeds.py 30 0 >> $O/big_instance_seeds.txt 2>&1
  ( python3 tools/bd_failure_compare.py device 8 3000 4 0; python3 tools/bd_failure_compare.py device 8 3000 6 0; python3 tools/bd_failure_rate.py 4096 2000 commutative 2>&1 | tail -3; python3 tools/bd_failure_rate.py 4104 2000 er7 2>&1 | tail -3 ) > $O/bd_failure_rates.txt 2>&1
fi
if [ "$WHAT" = sytrd ]; then
  # after the rework of the tridiagonalisation's latency chains: the default line again (kernels.sytrd_*), the dense driver's
  # kernel statistics, every form of the tridiagonalisation against LAPACK + its time, the eigensolver beside rocSOLVER's
  cd $R
  python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
  SDPSR_TOOL_FLAGS=0 python3 tools/sytrd_time.py 512 1024 2048 3072 4096 8192 > $O/sytrd_time.txt 2>&1
  python3 tools/sytrd_forms_check.py > $O/sytrd_forms_check.txt 2>&1
  rm -f $O/eig_drivers.txt
  for n in 1024 2048 4096; do for drv in 0 1; do python3 tools/eig_only.py $n $drv random 2>&1 | grep "syev n=" | tail -1 >> $O/eig_drivers.txt; done; done
  python3 tools/config_times.py > $O/config_times.txt 2>&1
  ( cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $O/dense -o dense -- python3 $R/bench.py --steps 3 --warmup 1 --skip-roofline --eig-driver 4 --no-graph > $O/dense_under_rocprof.json 2> /dev/null )
  tail -n 8 $O/sytrd_time.txt $O/eig_drivers.txt $O/sytrd_forms_check.txt
fi
ls $O | head -80
