"""int8 square as the product path launches it (4 channels, lower-triangle tiles): ms per launch."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
pkg = load_package()
with pkg.Context(seed=1) as ctx:
    for n in (4096, 8192):
        for aux in (104, 4):
            v = C.c_double(0)
            ctx.check(pkg._lib.load_prof_library().sdpsr_profile_kernel(ctx._h, 0, n, aux, 20, C.byref(v)))
            T = n // 128
            ex = (T + 1) / (2.0 * T) if aux >= 100 else 1.0
            print(n, "tri" if aux >= 100 else "full", "ms %.4f" % v.value, "executed POP/s %.3f" % (4 * 2.0 * n ** 3 * ex / v.value / 1e12 / 1e3 * 1e3 / 1e3))
