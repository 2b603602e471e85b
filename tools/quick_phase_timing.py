import sys, time, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
pkg = load_package(); pr = pkg.problems
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
t=time.time(); Ls, d = pr.synthetic_jordan_partition(n, seed=1); C_, A, b = pr.partition_as_sdp(Ls, seed=1)
n_, CL, X0L, U = pkg.admissible_setup(C_, A, b); print("setup", time.time()-t, "d", d, "r", U.shape[1])
dev = torch.device('cuda:0')
tCL = torch.from_numpy(CL).to(dev); tX0 = torch.from_numpy(X0L).to(dev); tU = torch.from_numpy(np.ascontiguousarray(U.T)).to(dev)  # (r, n2) row-major == col-major n2 x r
for mode in (pkg.SQUARE_I8, pkg.SQUARE_F32, pkg.SQUARE_F64):
    with pkg.Context(seed=3, square_mode=mode) as ctx:
        class UU:  # shim with shape
            pass
        for rep in range(3):
            torch.cuda.synchronize(); t=time.time()
            P = pkg.admissible_subspace(None, None, None, ctx=ctx, setup=(n, tCL, tX0, _U:=type('U',(),{'shape':(n*n,U.shape[1]),'data_ptr':tU.data_ptr,'is_cuda':True})()))
            torch.cuda.synchronize(); t1=time.time()-t
        print("mode", mode, "adm", t1*1e3, "ms dim", P.nparts, "iters", P.iterations, ["%.3f"%x for x in P.phase_ms])
        ok = np.array_equal(P.matrix.cpu().numpy().astype(np.int64), Ls); print("match", ok)
        for rep in range(2):
            torch.cuda.synchronize(); t=time.time()
            bd = pkg.blockDiagonalize(P, ctx=ctx)
            torch.cuda.synchronize(); t2=time.time()-t
        print("blockdiag", t2*1e3, "ms", sorted(bd.blkSizes)[:5], len(bd.blkSizes), ["%.3f"%x for x in bd.phase_ms])
