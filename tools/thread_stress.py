"""Concurrency check of the dense eigensolver: T threads, one ctx each, R calls of sdpsr_syev_f64
(n = 200) per thread; reports the worst eigenvalue error per (thread, call).
usage: thread_stress.py [eig_driver] [threads] [calls] [prelude]"""
import ctypes as C, sys, threading
import numpy as np
sys.path.insert(0, ".")
from __graft_entry__ import load_package
pkg = load_package()
drv = int(sys.argv[1]) if len(sys.argv) > 1 else 0
T = int(sys.argv[2]) if len(sys.argv) > 2 else 4
R = int(sys.argv[3]) if len(sys.argv) > 3 else 10
prelude = int(sys.argv[4]) if len(sys.argv) > 4 else 0
n = 200
bad, worst = [], [0.0] * T


def work(t):
    rng = np.random.default_rng(100 + t)
    with pkg.Context(seed=t, eig_driver=drv) as ctx:
        if prelude == 2:  # blockDiagonalize of a small golden partition first, as the test does
            g = np.load("tests/golden/golden_partitions.npz")
            name = ["er5", "er7", "esc16j", "er3"][t % 4]
            L = g[f"{name}_P"]
            for _ in range(3):
                bd = pkg.blockDiagonalize(pkg.Partition(int(L.max()), L.copy()), ctx=ctx)
                assert sorted(bd.blkSizes) == list(g[f"{name}_blk"])
        elif prelude:  # something small first
            x = rng.random((57, 57)); x = np.asfortranarray(x + x.T)
            w = np.empty(57); v = np.empty(57 * 57)
            ctx.check(ctx._lib.sdpsr_syev_f64(ctx._h, 57, C.c_void_p(x.ctypes.data), C.c_void_p(w.ctypes.data), C.c_void_p(v.ctypes.data), pkg.MEM_HOST))
        for r in range(R):
            x = rng.random((n, n)); x = np.asfortranarray(x + x.T)
            w = np.empty(n); v = np.empty(n * n)
            ctx.check(ctx._lib.sdpsr_syev_f64(ctx._h, n, C.c_void_p(x.ctypes.data), C.c_void_p(w.ctypes.data), C.c_void_p(v.ctypes.data), pkg.MEM_HOST))
            err = float(np.abs(w - np.linalg.eigvalsh(x)).max())
            V = v.reshape(n, n, order="F")
            res = float(np.abs(x @ V - V * w).max())
            worst[t] = max(worst[t], err)
            if err > 1e-9 or res > 1e-8:
                bad.append((t, r, err, res))


ts = [threading.Thread(target=work, args=(t,)) for t in range(T)]
[t.start() for t in ts]
[t.join() for t in ts]
print(f"driver {drv} threads {T} calls {R} prelude {prelude}: bad {len(bad)} of {T * R}; worst per thread {['%.1e' % w for w in worst]}")
for b in bad[:8]:
    print("   thread %d call %d  eigenvalue error %.3e  residual %.3e" % b)
