"""Stage 2 of a two-stage tridiagonalisation, measured: random symmetric band matrices of bandwidth b are
reduced to tridiagonal form by the bulge-chasing kernel of libsdpsr_prof.so; the eigenvalues of the result are compared
with those of the band matrix (scipy.linalg.eigvals_banded).  Usage: band_chase.py [n ...]"""
import sys, os, ctypes as C, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.linalg as sl
from __graft_entry__ import load_package
pkg = load_package()
prof = pkg._lib.load_prof_library()
ns = [int(x) for x in sys.argv[1:]] or [256, 1024, 2048, 4096]
rng = np.random.default_rng(3)
with pkg.Context(seed=1) as ctx:
    for n in ns:
        for b in (16, 32, 64):
            ab = np.zeros((b + 1, n))
            for k in range(b + 1):
                ab[k, :n - k] = rng.standard_normal(n - k)
            A = np.zeros((n, n), order="F")
            for k in range(b + 1):
                idx = np.arange(n - k)
                A[idx + k, idx] = ab[k, :n - k]
                A[idx, idx + k] = ab[k, :n - k]
            d = np.zeros(n); e = np.zeros(n - 1); out = (C.c_double * 2)()
            best = 1e9
            for rep in range(2):
                ctx.check(prof.sdpsr_profile_band_chase(ctx._h, n, b, A.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p), e.ctypes.data_as(C.c_void_p), out))
                best = min(best, out[0])
            ref = sl.eigvals_banded(ab, lower=True)
            got = sl.eigvalsh_tridiagonal(d, e)
            err = np.abs(np.sort(got) - np.sort(ref)).max() / max(1.0, np.abs(ref).max())
            print(f"band chase n={n:5d} b={b:2d}: {best:9.3f} ms  ({best * 1e3 / (3 * n):6.2f} us per dependent step of ~3n)  gave_up={int(out[1])}  max eigenvalue error {err:.2e}", flush=True)
