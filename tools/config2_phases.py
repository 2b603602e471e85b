"""Phase split of blockDiagonalize on the QAP-shaped config (N=900, dim 27828), device-resident output."""
import sys, os, time, numpy as np, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
pkg = load_package(); pr = pkg.problems; L = pkg._lib
import torch
flow, dist = pr.grid_qap_instance(5, 6, seed=4)
Cv, A, b = pr.qap_problem(flow, dist)
with pkg.Context(seed=1) as ctx:
    P = pkg.admissible_subspace(Cv, A, b, ctx=ctx)
    n = P.shape[0]; d = P.nparts
    lab = torch.from_numpy(np.ascontiguousarray(np.asarray(P.matrix).ravel(order="F")).astype(np.int32)).cuda()
    lib = ctx._lib
    for rep in range(2):
        nb = C.c_int32(0); ssq = C.c_int64(0); ss = C.c_int64(0)
        ms1 = (C.c_double * L.T_COUNT)(); ms2 = (C.c_double * L.T_COUNT)()
        torch.cuda.synchronize(); t = time.perf_counter()
        ctx.check(lib.sdpsr_block_diagonalize(ctx._h, n, C.c_void_p(lab.data_ptr()), d, 1.4901161193847656e-08, C.byref(nb), C.byref(ssq), C.byref(ss), C.cast(ms1, C.c_void_p), L.MEM_DEVICE))
        t1 = time.perf_counter()
        out = torch.empty(d * ssq.value, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize(); t2 = time.perf_counter()
        ctx.check(lib.sdpsr_block_images(ctx._h, C.c_void_p(out.data_ptr()), None, C.cast(ms2, C.c_void_p), L.MEM_DEVICE))
        torch.cuda.synchronize(); t3 = time.perf_counter()
        print("n %d dim %d blocks %d sum s^2 %d: diagonalize %.1f ms (phases %s), images %.1f ms (phase %s), output %.2f GB" % (
            n, d, nb.value, ssq.value, (t1 - t) * 1e3, ["%.1f" % x for x in list(ms1)[:8]], (t3 - t2) * 1e3, "%.1f" % list(ms2)[7], out.numel() * 8 / 1e9))
        del out
