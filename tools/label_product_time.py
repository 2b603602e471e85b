"""Time of the label product Y = A(v) W (sdpsr_profile_kernel kind 9) per shape; set SDPSR_SPMM_VALU=1
in the environment for the VALU form (A/B)."""
import ctypes as C, os, sys
sys.path.insert(0, ".")
from __graft_entry__ import load_package
pkg = load_package()
ctx = pkg.Context(device=0, seed=1)
for n, w, G, d in [(4096, 34, 1, 34), (4096, 16, 1, 34), (4096, 32, 1, 34), (4096, 48, 1, 34), (4096, 64, 1, 34), (4104, 36, 1, 36),
                   (4104, 18, 2, 36), (4104, 12, 4, 36), (2048, 34, 1, 18), (1024, 40, 1, 2000)]:
    v = C.c_double(0)
    ctx.check(pkg._lib.load_prof_library().sdpsr_profile_kernel(ctx._h, 9, n, w | (G << 8) | (d << 12), 2, C.byref(v)))  # code objects, buffers
    ctx.check(pkg._lib.load_prof_library().sdpsr_profile_kernel(ctx._h, 9, n, w | (G << 8) | (d << 12), 20, C.byref(v)))
    flop = 2.0 * n * n * w * G
    print(f"n={n} w={w} G={G} d={d}: {v.value*1e3:8.1f} us  {flop / v.value / 1e9:8.2f} TFLOP/s useful "
          f"({'VALU' if os.environ.get('SDPSR_SPMM_VALU') else 'MFMA'})")
