"""Shader clock while the hot kernels run (sdpsr_profile_clock): the int8 / fp32 / fp64 squares."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
pkg = load_package()
cases = [("int8 N=4096 2ch lower (product launch)", 0, 4096, 102, 40, 2 * 2 * 4096 ** 3 * (33 / 64.0)),
         ("int8 N=4096 4ch lower (round-2 launch)", 0, 4096, 104, 40, 4 * 2 * 4096 ** 3 * (33 / 64.0)),
         ("int8 N=8192 4ch full", 0, 8192, 4, 10, 4 * 2 * 8192 ** 3),
         ("fp32 N=8192", 1, 8192, 1, 10, 2 * 8192 ** 3),
         ("fp64 N=4096", 2, 4096, 1, 10, 2 * 4096 ** 3)]
with pkg.Context(seed=1) as ctx:
    for name, kind, n, aux, reps, ops in cases:
        out = (C.c_double * 3)()
        ctx.check(pkg._lib.load_prof_library().sdpsr_profile_clock(ctx._h, kind, n, aux, reps, out))
        print(f"{name:42s} {out[0]:8.4f} ms  {ops / out[0] / 1e9:9.1f} Tops/s  shader clock {out[1]:6.0f} MHz ({int(out[2])} intervals)")
