"""The tridiagonal divide and conquer (csrc/kernels_stedc.hip) on hard tridiagonal matrices, through sdpsr_syev_f64
(the tridiagonalisation of a tridiagonal matrix is the identity), against LAPACK; eig_driver 5 = rocSOLVER's stedc."""
import sys, os, time, numpy as np, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
pkg = load_package()
import scipy.linalg as sl
# the LAPACK comparison between the calls must not leave a 128-thread BLAS pool spinning on the box's CPU share while
# the next call's host side runs (profiles/r03_stedc_check.txt: 75-95 ms "host to host" for 11 of 48 calls)
from threadpoolctl import threadpool_limits
threadpool_limits(8)
def cases(n, rng):
    yield "random", rng.standard_normal(n), rng.standard_normal(n - 1)
    yield "1-2-1", 2 * np.ones(n), -np.ones(n - 1)
    yield "wilkinson", np.abs(np.arange(n) - n // 2).astype(float), np.ones(n - 1)
    yield "graded", 10.0 ** (-np.arange(n) * 12.0 / n), 10.0 ** (-np.arange(n - 1) * 12.0 / n)
    yield "zero offdiag blocks", rng.standard_normal(n), rng.standard_normal(n - 1) * (rng.random(n - 1) < 0.5)
    Q0, _ = np.linalg.qr(rng.standard_normal((n, n)))
    Dg = np.repeat(rng.standard_normal(6) * 3, n // 6 + 1)[:n]
    A = (Q0 * Dg) @ Q0.T; A = (A + A.T) / 2
    H = sl.hessenberg(A)
    yield "clustered (6 eigenvalues)", np.diag(H).copy(), np.diag(H, -1).copy()
    yield "glued wilkinson", np.tile(np.abs(np.arange(21) - 10.0), n // 21 + 1)[:n], np.where((np.arange(n - 1) + 1) % 21 == 0, 1e-8, 1.0)
    yield "identity", np.ones(n), np.zeros(n - 1)
ns = [int(x) for x in sys.argv[1:]] or [200, 777, 1024]
for drv in (0, 5):
    with pkg.Context(seed=1, eig_driver=drv) as ctx:
        lib = ctx._lib
        prof = pkg._lib.load_prof_library()
        for n in ns:
            rng = np.random.default_rng(0)
            # ONE set of host buffers per order, touched before the first call: a fresh pageable NumPy array of a few MB per
            # call is pinned by the HIP runtime on first use (80-95 ms "host to host" in profiles/r03_stedc_check.txt)
            T = np.zeros((n, n), order="F"); w = np.zeros(n); V = np.zeros((n, n), order="F")
            for name, d, e in cases(n, rng):
                T[:] = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
                t0 = time.perf_counter()
                st = lib.sdpsr_syev_f64(ctx._h, n, C.c_void_p(T.ctypes.data), C.c_void_p(w.ctypes.data), C.c_void_p(V.ctypes.data), 0)
                dt = (time.perf_counter() - t0) * 1e3
                if st:
                    print("driver %d %-26s n=%4d  STATUS %d %s" % (drv, name, n, st, lib.sdpsr_last_error(ctx._h))); continue
                wl = np.linalg.eigvalsh(T); sc = max(1e-300, np.abs(wl).max())
                print("driver %d %-26s n=%4d  |w-wl|/|w| %.2e  resid %.2e  orth %.2e  sorted %s  (%.1f ms host to host)" % (
                    drv, name, n, np.abs(w - wl).max() / sc, np.abs(T @ V - V * w).max() / sc, np.abs(V.T @ V - np.eye(n)).max(), bool(np.all(np.diff(w) >= 0)), dt), flush=True)
        gs = (C.c_double * 3)()
        prof.sdpsr_profile_sytrd_graphs(ctx._h, gs)
        print("driver %d: tridiagonalisation graph cache: %d replays, %d builds, %.1f ms building" % (drv, gs[0], gs[1], gs[2]), flush=True)
