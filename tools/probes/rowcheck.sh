cd $GRAFT_REPO_ROOT
SDPSR_TOOL_FLAGS=0 python3 tools/sytrd_time.py 512 1024 2048 3072 4096 2>&1 | tail -5
for n in 1024 2048 4096; do python3 tools/eig_only.py $n 0 random 2>&1 | grep -E "syev n=|resid |eigval" | tail -3; done
python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "syev or tridiagonal or eigen" 2>&1 | tail -2
