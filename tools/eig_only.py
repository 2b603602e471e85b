import sys, time, numpy as np, ctypes as C
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
pkg = load_package()
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
drv = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(0)
A = rng.standard_normal((n, n)); A = (A + A.T) / 2
dev = torch.device('cuda:0')
tA = torch.from_numpy(A).to(dev); tV = torch.empty_like(tA); tw = torch.empty(n, dtype=torch.float64, device=dev)
with pkg.Context(seed=1, eig_driver=drv) as ctx:
    lib = ctx._lib
    for rep in range(3):
        torch.cuda.synchronize(); t = time.time()
        ctx.check(lib.sdpsr_syev_f64(ctx._h, n, C.c_void_p(tA.data_ptr()), C.c_void_p(tw.data_ptr()), C.c_void_p(tV.data_ptr()), 1))
        torch.cuda.synchronize(); print("syev n=%d driver=%d: %.2f ms" % (n, drv, (time.time() - t) * 1e3))
w = tw.cpu().numpy(); V = tV.cpu().numpy().T  # row-major torch -> column-major matrix
print("resid", np.abs(A @ V - V * w).max(), "orth", np.abs(V.T @ V - np.eye(n)).max())
