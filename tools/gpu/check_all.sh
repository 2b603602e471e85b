# the full GPU test suite, then the refinement probes and the three instances (quick look after a change)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -6
python tools/refine_probe.py warm 4096 34 3000 30000 8388608 2>&1 | tail -5
for W in closed_scheme theta_c32xk128 theta_er7xk72; do
  python bench.py --steps 30 --warmup 5 --skip-roofline --workload $W | python -c "import json,sys; j=json.load(sys.stdin); print(j['config']['workload'][:70], j['value'], j['ms_per_step'], j['phase_ms_per_step'])"
done
