cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "refine or bucket or partition or ctor or overflow" 2>&1 | tail -5
bash tools/gpu/prof_refine.sh warm8m warm 4096 8388608
python tools/refine_probe.py both 4096 8388608 2>&1 | tail -3
