"""configs[1] G(1024, 1/2) theta': admissible_subspace alone, three calls (run under rocprofv3 --kernel-trace for the
kernel sequence of the last one)."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
pkg = load_package(); pr = pkg.problems
with pkg.Context(seed=1, flags=int(os.environ.get("SDPSR_TOOL_FLAGS", "0"))) as ctx:
    Cv, A, b = pr.theta_prime_problem(pr.gnp_adjacency(1024, 0.5, seed=11))
    setup = pkg.admissible_setup(Cv, A, b)
    for rep in range(3):
        t = time.perf_counter()
        P = pkg.admissible_subspace(Cv, A, b, ctx=ctx, setup=setup)
        print("admissible %.2f ms dim %d iters %d phases %s" % ((time.perf_counter() - t) * 1e3, P.nparts, P.iterations, ["%.2f" % x for x in P.phase_ms[:4]]), flush=True)
        print("MARK", time.time_ns(), flush=True)
