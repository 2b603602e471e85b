"""Failure rate of blockDiagonalize on the synthetic N x N partition over many draws."""
import sys, os, collections, numpy as np, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
pkg = load_package(); pr = pkg.problems; L = pkg._lib
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
kind = sys.argv[3] if len(sys.argv) > 3 else "commutative"
if kind == "commutative":
    Ls, d = pr.synthetic_jordan_partition(n, seed=1)
else:  # non-commutative: ER(7) algebra (x) {I, J - I} on n // 57 points (blocks [2,2,2,2,3] twice)
    g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "golden_partitions.npz"))
    Ls, d = pr.kron_with_complete(g["er7_P"].astype(np.int64), max(2, n // 57), seed=5)
    n = Ls.shape[0]
lab = torch.from_numpy(np.ascontiguousarray(Ls.ravel(order="F")).astype(np.int32)).cuda()
out = collections.Counter()
msgs = []
with pkg.Context(seed=77) as ctx:
    lib = ctx._lib
    for rep in range(reps):
        nb = C.c_int32(0); ssq = C.c_int64(0); ss = C.c_int64(0)
        st = lib.sdpsr_block_diagonalize(ctx._h, n, C.c_void_p(lab.data_ptr()), d, 1.4901161193847656e-08, C.byref(nb), C.byref(ssq), C.byref(ss), None, L.MEM_DEVICE)
        out[(st, nb.value if st == 0 else -1)] += 1
        if st != 0 and len(msgs) < 6:
            msgs.append(lib.sdpsr_last_error(ctx._h).decode()[:160])
print("n", n, "dim", d, "results (status, nblocks):", dict(out))
for m in msgs:
    print("   ", m)
