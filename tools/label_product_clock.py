"""Shader clock under the label product (sdpsr_profile_clock on kind 9) and, for comparison, under the fp64 GEMM."""
import ctypes as C, sys
sys.path.insert(0, ".")
from __graft_entry__ import load_package
pkg = load_package()
ctx = pkg.Context(device=0, seed=1)
out = (C.c_double * 3)()
for name, kind, n, aux, reps in [("label product w=34", 9, 4096, 34 | (1 << 8) | (34 << 12), 200),
                                 ("label product w=16", 9, 4096, 16 | (1 << 8) | (34 << 12), 200),
                                 ("label product w=64", 9, 4096, 64 | (1 << 8) | (34 << 12), 200),
                                 ("fp64 gemm", 2, 4096, 0, 10)]:
    ctx.check(pkg._lib.load_prof_library().sdpsr_profile_clock(ctx._h, kind, n, aux, reps, out))
    print(f"{name}: {out[0]*1e3:.1f} us per launch, shader clock {out[1]:.0f} MHz ({int(out[2])} intervals)")
