"""configs[2] (QAP-type, N = 900, dim 27828, blocks up to 81): phases of blockDiagonalize with
the block images left on the device (the 12 GB result over PCIe is not the kernels' time)."""
import sys, os, time, ctypes as C, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from __graft_entry__ import load_package
pkg = load_package(); pr = pkg.problems; L = pkg._lib if hasattr(pkg, "_lib") else None
from importlib import import_module
Lm = import_module(pkg.__name__ + "._lib")
with pkg.Context(seed=1) as ctx:
    flow, dist = pr.grid_qap_instance(5, 6, seed=4)
    Cv, A, b = pr.qap_problem(flow, dist)
    setup = pkg.admissible_setup(Cv, A, b)
    P = pkg.admissible_subspace(Cv, A, b, ctx=ctx, setup=setup)
    n = P.shape[0]
    dev = torch.device("cuda:0")
    tP = torch.from_numpy(np.asfortranarray(P.matrix).ravel(order="F").view(np.int32).copy()).to(dev)
    lib = ctx._lib
    buf = None
    for rep in range(3):
        nb, ssq, ss = C.c_int32(0), C.c_int64(0), C.c_int64(0)
        ms1 = (C.c_double * Lm.T_COUNT)()
        t0 = time.perf_counter()
        st = lib.sdpsr_block_diagonalize(ctx._h, n, C.c_void_p(tP.data_ptr()), P.nparts, 1e-8, C.byref(nb), C.byref(ssq), C.byref(ss),
                                         C.cast(ms1, C.c_void_p), Lm.MEM_DEVICE)
        if st:
            print("status", st, lib.sdpsr_last_error(ctx._h)); continue
        if buf is None:
            buf = torch.empty(P.nparts * ssq.value, dtype=torch.float64, device=dev)
        ms2 = (C.c_double * Lm.T_COUNT)()
        ctx.check(lib.sdpsr_block_images(ctx._h, C.c_void_p(buf.data_ptr()), None, C.cast(ms2, C.c_void_p), Lm.MEM_DEVICE))
        dt = time.perf_counter() - t0
        print("N=%d dim %d sum s^2 = %d (%.1f GB of images): wall %.1f ms; diagonalize phases %s; images phases %s" % (
            n, P.nparts, ssq.value, P.nparts * ssq.value * 8 / 1e9, dt * 1e3, ["%.2f" % x for x in ms1], ["%.2f" % x for x in ms2]))
