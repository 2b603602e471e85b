"""Host waits (ctx_sync_stream calls) per reduction of the three bench instances at N = 4096 (sdpsr_profile_host_waits)."""
import sys, os, ctypes as C, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from __graft_entry__ import load_package
pkg = load_package(); pr = pkg.problems; L = pkg._lib
prof = L.load_prof_library()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
gold = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "golden_partitions.npz"))["er7_P"].astype(np.int64)
def build(name):
    if name == "closed_scheme":
        Ls, d = pr.synthetic_jordan_partition(n, seed=1); return pr.partition_as_sdp(Ls, seed=1)
    if name == "theta_c32xk128":
        return pr.theta_prime_product_problem(pr.cycle_adjacency(32), pr.symmetric_circulant_labels(32), n // 32, seed=1)[:3]
    return pr.theta_prime_product_problem(pr.er_graph_adjacency(7), gold, max(1, round(n / 57)), seed=1)[:3]
with pkg.Context(seed=11) as ctx:
    for name in ("closed_scheme", "theta_c32xk128", "theta_er7xk72"):
        Cv, A, b = build(name)
        setup = pkg.admissible_setup(Cv, A, b)
        nn, CL, X0L, U = setup
        tCL, tX0 = torch.from_numpy(CL).cuda(), torch.from_numpy(X0L).cuda()
        tU = torch.from_numpy(np.ascontiguousarray(U.T)).cuda()
        tP = torch.empty(nn * nn, dtype=torch.int32, device="cuda")
        blk = torch.empty(64 * 200, dtype=torch.float64, device="cuda")
        dd, it, nb, ssq, ss = C.c_int64(0), C.c_int32(0), C.c_int32(0), C.c_int64(0), C.c_int64(0)
        def call():
            if getattr(setup, "hint", 0): ctx._lib.sdpsr_hint_symmetric_basis(ctx._h, setup.hint)
            return ctx._lib.sdpsr_jordan_reduce(ctx._h, nn, C.c_void_p(tCL.data_ptr()), C.c_void_p(tX0.data_ptr()), C.c_void_p(tU.data_ptr()), U.shape[1], 1.5e-8, 1.5e-8,
                                                C.c_void_p(tP.data_ptr()), C.byref(dd), C.byref(it), C.byref(nb), C.byref(ssq), C.byref(ss), C.c_void_p(blk.data_ptr()), blk.numel(), None, 0, None, 1)
        for _ in range(3): call()
        w0 = C.c_uint64(0); w1 = C.c_uint64(0)
        prof.sdpsr_profile_host_waits(ctx._h, C.byref(w0))
        reps = 20
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(reps): call()
        torch.cuda.synchronize(); el = (time.perf_counter() - t) / reps
        prof.sdpsr_profile_host_waits(ctx._h, C.byref(w1))
        print(f"{name:16s} N={nn} dim {dd.value} iters {it.value}: {(w1.value - w0.value) / reps:5.2f} host waits per reduction, {el * 1e3:.3f} ms = {1 / el:.0f} reductions/s", flush=True)
