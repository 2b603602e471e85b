# rocprofv3 kernel statistics of the refinement probes (tools/refine_probe.py); $1 = tag, rest = probe arguments
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
tag=$1; shift
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o out -- python3 tools/refine_probe.py "$@" > gpurun_out/prof_${tag}.log 2>&1
f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/prof_${tag}_kernel_stats.csv
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:22]:
    print("%-70s calls %6s total %10.1f us avg %9.2f us  %5.1f%%" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"])/1e3, float(r["AverageNs"])/1e3, float(r["Percentage"])))
PY
