cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python tools/refine_probe.py both 4096 2>&1 | tail -12
bash tools/gpu/prof_refine.sh warm3k warm 4096 3000 | head -12
bash tools/gpu/prof_refine.sh warm30k warm 4096 30000 | head -12
python -m pytest tests -x -q -m gpu 2>&1 | tail -8
