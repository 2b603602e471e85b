import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
from __graft_entry__ import load_package
pkg = load_package(); pr = pkg.problems
g = np.load("tests/golden/golden_partitions.npz")
L, d = pr.kron_with_complete(g["er7_P"].astype(np.int64), 72, seed=5)
P = pkg.Partition(d, L.astype(np.uint32))
with pkg.Context(seed=3) as ctx:
    pkg.blockDiagonalize(P, ctx=ctx)
    os.environ["SDPSR_DEBUG"] = "1"
    bd = pkg.blockDiagonalize(P, ctx=ctx)
    print(sorted(bd.blkSizes))
