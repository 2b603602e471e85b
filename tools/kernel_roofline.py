"""Per-kernel roofline numbers through sdpsr_profile_kernel (HIP events on the ctx stream)."""
import sys, os, json, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
pkg = load_package()
ns = [int(x) for x in sys.argv[1:]] or [4096, 8192]
with pkg.Context(seed=1) as ctx:
    lib = ctx._lib
    def prof(kind, n, aux=0, reps=10):
        v = C.c_double(0); ctx.check(pkg._lib.load_prof_library().sdpsr_profile_kernel(ctx._h, kind, n, aux, reps, C.byref(v))); return v.value
    for n in ns:
        fl = 2.0 * n ** 3
        out = {"n": n}
        for name, kind, peak in (("square_i8", 0, 5000.0), ("square_f32", 1, 157.3), ("gemm_f64", 2, 78.6)):
            ms = prof(kind, n, reps=5)
            out[name] = {"ms": round(ms, 3), "T(FL)OP/s": round(fl / ms / 1e9, 1), "frac_of_dtype_peak": round(fl / ms / 1e9 / peak, 3),
                         "frac_of_fp32_mfma_peak": round(fl / ms / 1e9 / 157.3, 3)}
        for d in (34, 100000, n * n // 2):
            ms = prof(3, n, aux=d, reps=3)
            out[f"refine_d{d}"] = {"ms": round(ms, 3), "GB/s_algorithmic_16B": round(16.0 * n * n / ms / 1e6, 1)}
        for r in (2, 33):
            ms = prof(4, n, aux=r, reps=3)
            out[f"project_r{r}"] = {"ms": round(ms, 3), "GB/s": round(((4 + 8.0 * r) * 2 + 8) * n * n / ms / 1e6, 1)}
        print(json.dumps(out))
