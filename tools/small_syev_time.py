"""Time of the one-workgroup symmetric eigensolver (sdpsr_profile_kernel kind 8) per order."""
import ctypes as C, sys
sys.path.insert(0, ".")
from __graft_entry__ import load_package
pkg = load_package()
with pkg.Context(device=0, seed=1) as ctx:
    for n in (8, 16, 24, 34, 50, 64, 96, 128):
        v = C.c_double(0); sw = C.c_double(0)
        ctx.check(pkg._lib.load_prof_library().sdpsr_profile_kernel(ctx._h, 8, n, 0, 20, C.byref(v)))
        ctx.check(pkg._lib.load_prof_library().sdpsr_profile_kernel(ctx._h, 8, n, 1, 2, C.byref(sw)))
        print(f"small_syev n={n:4d}  {v.value*1e3:8.1f} us  sweeps {int(sw.value)}")
