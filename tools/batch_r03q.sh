#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03q; mkdir -p $O; cd $R
SDPSR_DEBUG=1 python tools/eig_only.py 4096 0 random 2>&1 | grep -E "done|syev n" | tail -8
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/eig4096 -o eig -- python3 $R/tools/eig_only.py 4096 0 random > $O/eig4096.log 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('/root/repo/gpurun_out/r03q/eig4096/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:22]: print("%-72s calls %5s tot/3 %9.1f us avg %8.1f" % (r['Name'][:72], r['Calls'], int(r['TotalDurationNs'])/3e3, float(r['AverageNs'])/1e3))
PY
