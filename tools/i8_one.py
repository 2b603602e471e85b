"""One int8 square launch configuration through sdpsr_profile_kernel: i8_one.py n aux reps"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
pkg = load_package()
n, aux, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
with pkg.Context(seed=1) as ctx:
    v = C.c_double(0)
    ctx.check(pkg._lib.load_prof_library().sdpsr_profile_kernel(ctx._h, 0, n, aux, reps, C.byref(v)))
    print(n, aux, "ms %.4f" % v.value)
