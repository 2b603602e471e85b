import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from __graft_entry__ import load_package
pkg = load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
Cv, A, b = pkg.problems.theta_prime_problem(pkg.problems.gnp_adjacency(n, 0.5, seed=11))
with pkg.Context(seed=21) as ctx:
    P = pkg.admissible_subspace(Cv, A, b, ctx=ctx)
    print("dim", P.nparts, flush=True)
    print(pkg.eigen_decomposition(P, atol=1.4901161193847656e-8, ctx=ctx), flush=True)
    print(pkg.diagonalize(P, atol=1.4901161193847656e-8, ctx=ctx))
