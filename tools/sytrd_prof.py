"""One whole tridiagonalisation per call (profile hook kind 6): for rocprofv3 --kernel-trace --stats."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
pkg = load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
with pkg.Context(seed=1) as ctx:
    v = C.c_double(0)
    ctx.check(pkg._lib.load_prof_library().sdpsr_profile_kernel(ctx._h, 6, n, 0, reps, C.byref(v)))
    print("sytrd n=%d: %.3f ms" % (n, v.value))
