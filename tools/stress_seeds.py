"""Seed sweep on the golden problems: admissible_subspace must give the golden matrix for every seed
and square mode; blockDiagonalize must give the pinned block sizes (failures = the reference's own
randomised failure modes: NumericalInconsistency / DimensionMismatch) -- counts are printed."""
import sys, os, collections, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
pkg = load_package(); pr = pkg.problems
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
g = np.load(os.path.join(root, "tests", "golden", "golden_partitions.npz"))
nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
flags = int(sys.argv[2]) if len(sys.argv) > 2 else 0  # sdpsr_opts.flags for A/B sweeps
channels = int(sys.argv[3]) if len(sys.argv) > 3 else 0
def problem(name):
    if name == "petersen":
        return pr.theta_prime_problem(pr.petersen_adjacency())
    if name.startswith("er"):
        return pr.theta_prime_problem(pr.er_graph_adjacency(int(name[2:])))
    fa, fb = pr.read_qapdata(os.path.join(root, "tests", "golden", "esc16j.dat"))
    return pr.qap_problem(fa, fb)
for name in ("petersen", "er3", "er5", "er7", "esc16j"):
    Cv, A, b = problem(name)
    setup = pkg.admissible_setup(Cv, A, b)
    bad = collections.Counter(); iters = collections.Counter()
    for mode, mname in ((pkg.SQUARE_I8, "i8"), (pkg.SQUARE_F32, "f32"), (pkg.SQUARE_F64, "f64")):
        for seed in range(nseeds):
            with pkg.Context(seed=1000 + seed, square_mode=mode, flags=flags, channels=channels) as ctx:
                P = pkg.admissible_subspace(Cv, A, b, ctx=ctx, setup=setup)
                iters[(mname, P.iterations)] += 1
                if not np.array_equal(P.matrix, g[f"{name}_P"]):
                    bad[mname] += 1
    print(f"[flags {flags} channels {channels}] admissible {name}: mismatches {dict(bad)} of {nseeds} seeds per mode; iterations {dict(sorted(iters.items()))}", flush=True)
for name, eps in (("petersen", None), ("er3", None), ("er5", None), ("er7", None), ("esc16j", None), ("numerical_issues", 1e-7), ("circ64", None), ("circ256", None)):
    L = g[f"{name}_P"]; P = pkg.Partition(int(L.max()), L.copy())
    out = collections.Counter()
    for seed in range(nseeds):
        with pkg.Context(seed=5000 + seed, flags=flags) as ctx:
            try:
                bd = pkg.blockDiagonalize(P, ctx=ctx, **({} if eps is None else {"epsilon": eps}))
                out["ok" if sorted(bd.blkSizes) == list(g[f"{name}_blk"]) else "wrong sizes %s" % sorted(bd.blkSizes)] += 1
            except Exception as e:
                out[type(e).__name__] += 1
    print(f"blockDiagonalize {name}: {dict(out)}", flush=True)
