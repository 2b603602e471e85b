cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python tools/refine_probe.py warm 4096 262144 8388608 2>&1 | tail -3
python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "generic_theta or config1" -s 2>&1 | tail -8
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "refine or bucket or partition" 2>&1 | tail -3
