#!/bin/bash
# kernel statistics of one tridiagonalisation at N = $1 (launches one by one: SDPSR_FLAG_NO_GRAPH)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/look; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export SDPSR_TOOL_FLAGS=${2:-256}
rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -o s -- python3 $R/tools/sytrd_time.py ${1:-4096} > $O/log.txt 2>&1
tail -3 $O/log.txt
python3 - <<PY
import csv,glob
f=glob.glob("$O/st/**/s_kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print(r['Name'][:70].ljust(70), r['Calls'].rjust(7), "%8.1f us avg" % (float(r['AverageNs'])/1e3), "%8.1f ms" % (float(r['TotalDurationNs'])/1e6), "min %.1f max %.1f" % (float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
