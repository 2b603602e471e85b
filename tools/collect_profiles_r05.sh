#!/bin/bash
# Round 5, on the GPU box (through gpurun): everything profiles/r05_* is made from.  Output under gpurun_out/r05/; the
# files are copied into profiles/ by hand afterwards (names: r05_<file>).  Sections: bench | stats | refine | n8192 | configs
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05
mkdir -p $O
cd $R && export TMPDIR=/tmp
WHAT=${1:-all}
stats() {  # tag, bench args...
  local tag=$1; shift
  rm -rf /tmp/st_$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st_$tag -o bench -- python3 bench.py --skip-roofline "$@" > $O/bench_${tag}_under_rocprof.json 2> /dev/null
  cp $(find /tmp/st_$tag -name "*kernel_stats.csv" | head -1) $O/bench_${tag}_kernel_stats.csv
}
if [ "$WHAT" = all ] || [ "$WHAT" = bench ]; then
  python3 bench.py > $O/bench_n4096.json 2> $O/bench_n4096.err
fi
if [ "$WHAT" = all ] || [ "$WHAT" = stats ]; then
  stats n4096 --steps 30 --warmup 5
  stats theta_c32xk128 --steps 20 --warmup 3 --workload theta_c32xk128
  stats theta_er7xk72 --steps 20 --warmup 3 --workload theta_er7xk72
  stats dense_driver --steps 3 --warmup 1 --eig-driver 4 --no-graph
fi
if [ "$WHAT" = all ] || [ "$WHAT" = refine ]; then
  python3 tools/refine_probe.py both 4096 34 3000 30000 262144 1000000 8388608 > $O/refine_probes.txt 2>&1
  for T in "cold 8388608" "warm 8388608" "warm 3000" "warm 262144" "warm 34"; do
    set -- $T
    rm -rf /tmp/rp_$1_$2
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$1_$2 -o out -- python3 tools/refine_probe.py $1 4096 $2 > $O/refine_$1_$2.log 2>&1
    cp $(find /tmp/rp_$1_$2 -name "*kernel_stats.csv" | head -1) $O/refine_$1_$2_classes_kernel_stats.csv
  done
  bash tools/gpu/pmc_refine.sh 4096 8388608 > $O/refine_pmc_8388608.txt 2>&1
  cp gpurun_out/pmc_refine_FETCH_SIZE.csv $O/ 2>/dev/null; cp gpurun_out/pmc_refine_WRITE_SIZE.csv $O/ 2>/dev/null
fi
if [ "$WHAT" = all ] || [ "$WHAT" = n8192 ]; then
  python3 bench.py --n 8192 --steps 10 --warmup 2 > $O/bench_n8192.json 2> $O/bench_n8192.err
  stats n8192 --n 8192 --steps 10 --warmup 2
fi
if [ "$WHAT" = all ] || [ "$WHAT" = configs ]; then
  python3 tools/config_times.py > $O/config_times.txt 2>&1
  python3 tools/host_waits.py > $O/host_waits.txt 2>&1
  python3 tools/band_chase.py 1024 2048 4096 > $O/band_chase.txt 2>&1
  python3 tools/band_reduce.py 1024 2048 4096 > $O/band_reduce.txt 2>&1
  bash tools/gpu/pmc_round.sh > $O/pmc_round.txt 2>&1
fi
ls -la $O
