"""Failure rate of the randomized blockDiagonalize on ER(7) (x) K_k (blocks [2,2,2,2,3] twice):
the CPU oracle (reference-literal restatement) and the device path on the SAME partition.
usage: bd_failure_compare.py oracle|device [k] [runs] [eig_driver] [flags]
       bd_failure_compare.py generic-oracle|generic-device n runs [eig_driver] [flags]
generic-*: the partition WITHOUT symmetry of a seeded G(n, 1/2) theta' problem (BASELINE configs[1]: every unordered pair
its own class, dim = n (n + 1) / 2) through eigen_decomposition (src/eigen_decomposition.jl:236-273) -- the step whose
NumericalInconsistency the reference answers with "try again" (:264-270); n eigenspaces of dimension 1, one block."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from __graft_entry__ import load_package
pkg = load_package()
who = sys.argv[1]
if who.startswith("generic"):
    n = int(sys.argv[2]); runs = int(sys.argv[3]) if len(sys.argv) > 3 else 300
    rng = np.random.default_rng(12345 + n)
    iu = np.triu_indices(n)
    L = np.zeros((n, n), dtype=np.int64)
    L[iu] = np.arange(1, len(iu[0]) + 1)          # every unordered pair {i, j} its own class ...
    L = np.maximum(L, L.T)
    L, _ = pkg.problems.canonical_labels(L)          # ... numbered canonically (first occurrence, column-major)
    d = int(L.max())
    fails = {}
    t0 = time.time()
    if who == "generic-oracle":
        import sdpsr_oracle as O
        P = O.Partition(d, L)
        for s in range(runs):
            try:
                O.eigen_decomposition(P, rng=np.random.default_rng(1000 + s))
            except Exception as e:  # noqa: BLE001
                fails[type(e).__name__] = fails.get(type(e).__name__, 0) + 1
    else:
        drv = int(sys.argv[4]) if len(sys.argv) > 4 else 0
        flags = int(sys.argv[5]) if len(sys.argv) > 5 else 0
        P = pkg.Partition(d, L.astype(np.uint32))
        with pkg.Context(seed=77, eig_driver=drv, flags=flags) as ctx:
            for s in range(runs):
                try:
                    pkg.eigen_decomposition(P, ctx=ctx)
                except pkg.SdpsrError as e:
                    fails[type(e).__name__] = fails.get(type(e).__name__, 0) + 1
    nf = sum(fails.values())
    print(who, " ".join(sys.argv[2:]), "N", n, "dim", d, "runs", runs, "failures", fails, "rate %.3f" % (nf / runs), "%.1f s" % (time.time() - t0), flush=True)
    sys.exit(0)
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
