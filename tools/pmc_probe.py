"""Small driver for rocprofv3 --pmc passes: runs one hot kernel family through the profiling hook."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
pkg = load_package()
kind = int(sys.argv[1]); n = int(sys.argv[2]); aux = int(sys.argv[3]) if len(sys.argv) > 3 else 0
with pkg.Context(seed=1) as ctx:
    v = C.c_double(0)
    ctx.check(pkg._lib.load_prof_library().sdpsr_profile_kernel(ctx._h, kind, n, aux, 2, C.byref(v)))
    print("kind", kind, "n", n, "ms", v.value)
