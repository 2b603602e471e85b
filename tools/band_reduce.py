"""Both stages of a two-stage tridiagonalisation, measured: stage 1 dense -> band (sdpsr_profile_band_reduce:
rocSOLVER panel QR + rocBLAS level-3 updates) followed by stage 2 band -> tridiagonal (sdpsr_profile_band_chase, own bulge
chasing) on the band it produced; eigenvalues of the band and of the tridiagonal matrix against numpy's of the dense input.
Usage: band_reduce.py [n ...]"""
import sys, os, ctypes as C, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from __graft_entry__ import load_package
from scipy.linalg import eigvals_banded, eigvalsh_tridiagonal
pkg = load_package()
prof = pkg._lib.load_prof_library()
sizes = [int(x) for x in sys.argv[1:]] or [1024, 2048, 4096]
out = (C.c_double * 2)()
with pkg.Context(seed=1) as ctx:
    for n in sizes:
        rng = np.random.default_rng(n)
        G = rng.standard_normal((n, n))
        A0 = np.asfortranarray(G + G.T)
        ref = np.linalg.eigvalsh(A0)
        for b in (16, 32, 64):
            A = A0.copy(order="F")
            ctx.check(prof.sdpsr_profile_band_reduce(ctx._h, n, b, A.ctypes.data_as(C.c_void_p), out))
            ms1, ms_panels = out[0], out[1]
            L = np.tril(A)
            beyond = np.abs(np.tril(L, -b - 1)).max()
            ab = np.zeros((b + 1, n))
            for k in range(b + 1):
                ab[k, :n - k] = np.diagonal(L, -k)
            err1 = np.abs(eigvals_banded(ab, lower=True) - ref).max() / np.abs(ref).max()
            Bd = np.asfortranarray(np.tril(L) + np.tril(L, -1).T)
            d = np.zeros(n); e = np.zeros(n - 1)
            ctx.check(prof.sdpsr_profile_band_chase(ctx._h, n, b, Bd.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p), e.ctypes.data_as(C.c_void_p), out))
            ms2 = out[0]
            err2 = np.abs(eigvalsh_tridiagonal(d, e) - ref).max() / np.abs(ref).max()
            print(f"two-stage n={n:5d} b={b:2d}: stage 1 {ms1:8.2f} ms (panel QR {ms_panels:7.2f}) + stage 2 {ms2:8.2f} ms = {ms1 + ms2:8.2f} ms;  "
                  f"beyond the band {beyond:.1e}, eigenvalue error after stage 1 {err1:.1e}, after stage 2 {err2:.1e}", flush=True)
