#!/bin/bash
# quick rocprofv3 kernel statistics of one bench workload (timed steps only): tools/quick_prof.sh [workload] [extra bench args]
R=$GRAFT_REPO_ROOT
W=${1:-closed_scheme}
shift
O=$R/gpurun_out/quick_$W
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o q -- python3 $R/bench.py --steps 20 --warmup 3 --skip-roofline --cpu-n 0 --workload $W "$@" > $O/bench.json 2> $O/err.txt
find $O -name "*kernel_stats.csv" -exec cp {} $O/stats.csv \;
