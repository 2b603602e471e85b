set -x
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "refine or bucket or partition or ctor or overflow" 2>&1 | tail -15
python tools/refine_probe.py both 4096 2>&1 | tail -20
