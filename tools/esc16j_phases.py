import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
pkg = load_package(); pr = pkg.problems
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
fa, fb = pr.read_qapdata(os.path.join(root, "tests", "golden", "esc16j.dat"))
Cv, A, b = pr.qap_problem(fa, fb)
import torch
with pkg.Context(seed=1) as ctx:
    for rep in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        P = pkg.admissible_subspace(Cv, A, b, ctx=ctx)
        t1 = time.perf_counter()
        setup = pkg.admissible_setup(Cv, A, b)
        t2 = time.perf_counter()
        P2 = pkg.admissible_subspace(Cv, A, b, ctx=ctx, setup=setup)
        t3 = time.perf_counter()
        bd = pkg.blockDiagonalize(P, ctx=ctx)
        t4 = time.perf_counter()
        print("device-setup admissible %.2f ms (iters %d, phases %s) | host setup %.2f ms + loop %.2f ms | blockDiagonalize %.2f ms" % (
            (t1 - t) * 1e3, P.iterations, ["%.2f" % x for x in P.phase_ms[:4]], (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3))
