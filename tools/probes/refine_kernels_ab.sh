cd $GRAFT_REPO_ROOT
for w in closed_scheme theta_c32xk128 theta_er7xk72; do python3 bench.py --steps 20 --warmup 3 --skip-roofline --cpu-n 0 --workload $w 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('phase_ms_per_step'))"; done
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/abins -o b -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --skip-roofline --cpu-n 0 --workload theta_c32xk128 > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$GRAFT_REPO_ROOT/gpurun_out/abins/**/b_kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f))):
    if any(k in r['Name'] for k in ('insert','class_sums','uconst_check','refine_label_kernel')): print(r['Name'][:90], r['Calls'], "%.1f us"%(float(r['AverageNs'])/1e3))
PY
