"""Does torch still find the GPU when the library made the process's first HIP call?  (tools/config_times.py once failed
with "No HIP GPUs are available" at its first .cuda() after the library had run.)"""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package
pkg = load_package()
with pkg.Context(seed=1) as ctx:
    M = np.random.default_rng(0).integers(0, 5, (64, 64)).astype(np.float64)
    P = pkg.Partition.from_matrix(M, ctx=ctx) if hasattr(pkg.Partition, "from_matrix") else None
    print("library ran first:", None if P is None else P.nparts)
    import torch
    try:
        print("torch after library:", torch.zeros(4).cuda().sum().item(), torch.cuda.device_count())
    except Exception as e:
        print("torch after library FAILED:", type(e).__name__, e)
