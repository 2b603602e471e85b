"""Failure rate of the randomized blockDiagonalize on ER(7) (x) K_k (blocks [2,2,2,2,3] twice):
the CPU oracle (reference-literal restatement) and the device path on the SAME partition.
usage: bd_failure_compare.py oracle|device [k] [runs] [eig_driver] [flags]"""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from __graft_entry__ import load_package
pkg = load_package()
who = sys.argv[1]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 8
runs = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
gold = np.load(os.path.join(ROOT, "tests", "golden", "golden_partitions.npz"))["er7_P"].astype(np.int64)
L, d = pkg.problems.kron_with_complete(gold, k, seed=5)
expect = sorted([2, 2, 2, 2, 3] * 2)
fails = {}
t0 = time.time()
if who == "oracle":
    import sdpsr_oracle as O
    P = O.Partition(d, L)
    for s in range(runs):
        try:
            sizes, _, _ = O.block_diagonalize(P, rng=np.random.default_rng(1000 + s))
            if sorted(sizes) != expect:
                fails["wrong_sizes"] = fails.get("wrong_sizes", 0) + 1
        except Exception as e:  # noqa: BLE001
            fails[type(e).__name__] = fails.get(type(e).__name__, 0) + 1
else:
    drv = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    flags = int(sys.argv[5]) if len(sys.argv) > 5 else 0  # sdpsr_opts.flags, e.g. 64 = SDPSR_FLAG_SINGLE_COUPLING_ELEMENT
    P = pkg.Partition(d, L.astype(np.uint32))
    with pkg.Context(seed=77, eig_driver=drv, flags=flags) as ctx:
        for s in range(runs):
            try:
                bd = pkg.blockDiagonalize(P, ctx=ctx)
                if sorted(bd.blkSizes) != expect:
                    fails["wrong_sizes"] = fails.get("wrong_sizes", 0) + 1
                    print("run", s, "sizes", sorted(bd.blkSizes), flush=True)
            except pkg.SdpsrError as e:
                fails[type(e).__name__] = fails.get(type(e).__name__, 0) + 1
print(who, " ".join(sys.argv[2:]), "N", L.shape[0], "dim", d, "runs", runs, "failures", fails, "%.1f s" % (time.time() - t0))
