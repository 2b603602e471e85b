# rocprofv3 kernel statistics of the bench command for one instance: $1 = workload, $2 = tag
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
W=${1:-closed_scheme}; TAG=${2:-$W}; shift; shift
rm -rf /tmp/bs_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bs_$TAG -o bench -- python3 bench.py --steps 30 --warmup 5 --skip-roofline --workload $W "$@" > gpurun_out/bench_${TAG}_under_rocprof.json 2> /dev/null
cp $(find /tmp/bs_$TAG -name "*kernel_stats.csv" | head -1) gpurun_out/bench_${TAG}_kernel_stats.csv
python3 - gpurun_out/bench_${TAG}_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:26]:
    print("%-74s calls %5s avg %8.2f us  %5.1f%%" % (r["Name"][:74], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
