cd $GRAFT_REPO_ROOT
for w in 2 3 4 5 6; do for wl in closed_scheme theta_c32xk128; do python3 bench.py --steps 20 --warmup 3 --skip-roofline --cpu-n 0 --workload $wl --insert-wgs $w 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w $wl', d['value'], d['ms_per_step'], d['phase_ms_per_step']['refine'])"; done; done
