#!/bin/bash
# rocprofv3 kernel statistics of one tools/*.py script: tools/prof_tool.sh <script.py> [args]; summary in gpurun_out/prof_tool/stats.csv
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_tool
rm -rf $O; mkdir -p $O
S=$R/$1
shift
cd $R && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o q -- python3 $S "$@" > $O/out.txt 2> $O/err.txt
find $O -name "*kernel_stats.csv" -exec cp {} $O/stats.csv \;
