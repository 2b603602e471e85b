import sys, os, time, ctypes as C, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
pkg = load_package(); pr = pkg.problems; L = pkg._lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
drv = int(sys.argv[2]) if len(sys.argv) > 2 else 0
Ls, d = pr.synthetic_jordan_partition(n, seed=1)
dev = torch.device("cuda:0")
tP = torch.from_numpy(np.ascontiguousarray(Ls.ravel(order="F")).astype(np.int32)).to(dev)
with pkg.Context(seed=5, eig_driver=drv) as ctx:
    lib = ctx._lib
    for rep in range(4):
        nb = C.c_int32(0); ssq = C.c_int64(0); ss = C.c_int64(0); ms1 = (C.c_double * L.T_COUNT)()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx.check(lib.sdpsr_block_diagonalize(ctx._h, n, C.c_void_p(tP.data_ptr()), d, 1.49e-8, C.byref(nb), C.byref(ssq), C.byref(ss), C.cast(ms1, C.c_void_p), 1))
        t1 = time.perf_counter()
        blk = torch.empty(d * ssq.value, dtype=torch.float64, device=dev); ms2 = (C.c_double * L.T_COUNT)()
        ctx.check(lib.sdpsr_block_images(ctx._h, C.c_void_p(blk.data_ptr()), None, C.cast(ms2, C.c_void_p), 1))
        t2 = time.perf_counter()
        print("rep", rep, "block_diagonalize wall %.2f ms (event total %.2f; phases %s)  block_images wall %.2f ms" % ((t1 - t0) * 1e3, ms1[0], ["%.2f" % x for x in list(ms1)[4:8]], (t2 - t1) * 1e3))
