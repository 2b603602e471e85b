cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -8
python bench.py > gpurun_out/bench_a.json 2> gpurun_out/bench_a.err; tail -c 600 gpurun_out/bench_a.err
python - <<'PY'
import json
js = json.load(open("gpurun_out/bench_a.json"))
print("value", js["value"], "ms/step", js["ms_per_step"], "setup", js["setup_ms"])
for k, v in js["workloads"].items(): print(k, v["value"], v["phase_ms_per_step"])
for k, v in js["variants"].items(): print(k, v.get("value"), v.get("ms_per_call"), v.get("h2d_MB_per_call"), v.get("error"))
print("roofline", js["roofline"]["frac"], js["roofline"]["ms_per_launch"], js["roofline"].get("shader_clock_mhz"))
for k, v in js["kernels"].items():
    if "refine" in k: print(k, v["ms"], v["frac"])
print("cpu", js["cpu_baseline"]["value"], js["cpu_baseline"].get("like_for_like"))
PY
