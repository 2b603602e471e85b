#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03i; mkdir -p $O; cd $R
SDPSR_DEBUG=1 timeout 600 python tools/config2_bd_phases.py > $O/config2_bd_phases.txt 2>&1
timeout 900 python bench.py --steps 30 --warmup 5 --cpu-n 0 > $O/bench_default.json 2> $O/bench_default.err
tail -30 $O/config2_bd_phases.txt; python -c "
import json; d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step']); print(json.dumps(d.get('variants'))[:600]); print(json.dumps(d.get('workloads'))[:800]); print(json.dumps(d.get('kernels',{}).get('sytrd_total')))"
