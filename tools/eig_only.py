import sys, time, numpy as np, ctypes as C
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
pkg = load_package()
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
drv = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(0)
kind = sys.argv[3] if len(sys.argv) > 3 else "random"
if kind == "random":
    A = rng.standard_normal((n, n)); A = (A + A.T) / 2
elif kind == "degenerate":
    Q0, _ = np.linalg.qr(rng.standard_normal((n, n)))
    D = np.repeat(rng.standard_normal(10) * 10, n // 10 + 1)[:n]
    A = (Q0 * D) @ Q0.T; A = (A + A.T) / 2
else:
    Ls, d = pkg.problems.synthetic_jordan_partition(n, seed=1)
    A = np.ascontiguousarray(np.concatenate([[0.0], rng.random(d)])[Ls])  # C order: empty_like(tA) below must be row-major
dev = torch.device('cuda:0')
tA = torch.from_numpy(A).to(dev); tV = torch.empty_like(tA); tw = torch.empty(n, dtype=torch.float64, device=dev)
flags = int(os.environ.get("SDPSR_TOOL_FLAGS", "0"))
with pkg.Context(seed=1, eig_driver=drv, flags=flags) as ctx:
    lib = ctx._lib
    for rep in range(3):
        torch.cuda.synchronize(); t = time.time()
        ctx.check(lib.sdpsr_syev_f64(ctx._h, n, C.c_void_p(tA.data_ptr()), C.c_void_p(tw.data_ptr()), C.c_void_p(tV.data_ptr()), 1))
        torch.cuda.synchronize(); print("syev n=%d driver=%d: %.2f ms" % (n, drv, (time.time() - t) * 1e3))
Vt = tV.t()  # column-major buffer viewed by torch as its transpose
R = tA @ Vt - Vt * tw[None, :]
print("resid", float(R.abs().max()), "orth", float((Vt.t() @ Vt - torch.eye(n, dtype=torch.float64, device=dev)).abs().max()),
      "sym", float((tA - tA.t()).abs().max()), "wmin/max", float(tw.min()), float(tw.max()))
rq = ((tA @ Vt) * Vt).sum(0)
print("ascending:", bool((tw[1:] >= tw[:-1]).all()), "max|rayleigh - w|", float((rq - tw).abs().max()),
      "max|sorted rayleigh - w|", float((rq.sort().values - tw).abs().max()))
R2 = tA @ Vt - Vt * rq[None, :]
print("resid with rayleigh quotients", float(R2.abs().max()))
wl = np.linalg.eigvalsh(A) if n <= 2048 else None
if wl is not None: print("eigval err", np.abs(np.sort(tw.cpu().numpy()) - wl).max())

if len(sys.argv) > 4:
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    np.save(os.path.join(out, "eig_A.npy"), A); np.save(os.path.join(out, "eig_w.npy"), tw.cpu().numpy()); np.save(os.path.join(out, "eig_V.npy"), Vt.cpu().numpy())
