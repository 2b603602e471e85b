"""Every form of the tridiagonalisation against LAPACK through sdpsr_syev_f64, and the time of the whole tridiagonalisation:
row form (n <= 2048), panel form at every order (SDPSR_FLAG_SYTRD_PANELS), the default hybrid beyond 2048, the hybrid with the
one-launch panel columns (SDPSR_FLAG_SYTRD_ONE_LAUNCH)."""
import sys, os, time, numpy as np, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
pkg = load_package()
import torch
torch.cuda.init()
from threadpoolctl import threadpool_limits
threadpool_limits(8)
L = pkg._lib
rng = np.random.default_rng(7)
lib = pkg.load_library()
cases = [(777, L.FLAG_SYTRD_PANELS), (1536, L.FLAG_SYTRD_PANELS), (2500, 0), (2500, L.FLAG_SYTRD_ONE_LAUNCH), (4104, 0), (4104, L.FLAG_SYTRD_ONE_LAUNCH), (4200, L.FLAG_SYTRD_ONE_LAUNCH)]
if len(sys.argv) > 1 and sys.argv[1] == "quick": cases = cases[:3]
mats = {}
for n, flags in cases:
    if n not in mats:
        A = rng.standard_normal((n, n)); A = np.asfortranarray((A + A.T) / 2)
        A[np.triu_indices(n, 1)] = 1e300
        Asym = np.tril(A) + np.tril(A, -1).T
        mats[n] = (A, Asym, np.linalg.eigvalsh(Asym))
    A, Asym, wl = mats[n]
    w = np.zeros(n); V = np.zeros((n, n), order="F")
    with pkg.Context(seed=1, flags=flags) as ctx:
        ctx.check(lib.sdpsr_syev_f64(ctx._h, n, C.c_void_p(A.ctypes.data), C.c_void_p(w.ctypes.data), C.c_void_p(V.ctypes.data), 0))
    print("n=%d flags=%d  |w-wl| %.2e  orth %.2e  resid %.2e" % (n, flags, np.abs(w - wl).max(), np.abs(V.T @ V - np.eye(n)).max(),
                                                                np.abs(Asym @ V - V * w).max()), flush=True)
prof = L.load_prof_library()
for n, flags in ((1024, L.FLAG_SYTRD_PANELS), (2048, L.FLAG_SYTRD_PANELS), (3072, 0), (4096, 0), (4096, L.FLAG_SYTRD_ONE_LAUNCH), (4096, L.FLAG_SYTRD_PANELS), (8192, 0)):
    with pkg.Context(seed=1, flags=flags) as ctx:
        v = C.c_double(0)
        ctx.check(prof.sdpsr_profile_kernel(ctx._h, 6, n, 0, 3, C.byref(v)))
        print("sytrd n=%d flags=%d: %.2f ms" % (n, flags, v.value), flush=True)
