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
done
# the insert pass of the refinement (refine_insert_kernel<SrcJoint<2,2>,8,1024> in theta_c32xk128): instruction mix and stalls
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/pmc_insert_A -o p -- python3 $R/bench.py --steps 4 --warmup 2 --skip-roofline --workload theta_c32xk128 > $O/pmc_insert_A.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_insert_B -o p -- python3 $R/bench.py --steps 4 --warmup 2 --skip-roofline --workload theta_c32xk128 > $O/pmc_insert_B.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_insert_G -o p -- python3 $R/bench.py --steps 4 --warmup 2 --skip-roofline --workload theta_c32xk128 > /dev/null 2>&1
fi
if [ "$WHAT" = all ] || [ "$WHAT" = power ]; then
cd $R
python3 tools/power_trace.py 0 4096 102 3 > $O/power_i8_product_launch.txt 2>&1
python3 tools/power_trace.py 0 4096 104 3 > $O/power_i8_4ch.txt 2>&1
python3 tools/power_trace.py 0 8192 4 3 > $O/power_i8_n8192.txt 2>&1
python3 tools/power_trace.py 1 4096 1 3 > $O/power_f32.txt 2>&1
python3 tools/power_trace.py 2 4096 1 3 > $O/power_f64.txt 2>&1
python3 tools/clock_under_kernels.py > $O/clock_under_kernels.txt 2>&1
fi
ls $O | head -50
