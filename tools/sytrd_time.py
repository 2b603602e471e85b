"""Timing of the dense eigen path pieces through sdpsr_profile_kernel (HIP events on ctx's stream):
kind 6 = one whole tridiagonalisation, kind 5 = its symv launches only (average per launch)."""
import sys, os, json, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
pkg = load_package()
ns = [int(x) for x in sys.argv[1:]] or [4096]
with pkg.Context(seed=1, flags=int(os.environ.get("SDPSR_TOOL_FLAGS", "0"))) as ctx:
    lib = ctx._lib
    def prof(kind, n, aux=0, reps=3):
        v = C.c_double(0); ctx.check(pkg._lib.load_prof_library().sdpsr_profile_kernel(ctx._h, kind, n, aux, reps, C.byref(v))); return v.value
    for n in ns:
        t6 = prof(6, n, reps=3)
        t5 = prof(5, n, reps=2)
        print(json.dumps({"n": n, "sytrd_total_ms": round(t6, 3), "symv_avg_us": round(t5 * 1e3, 3), "symv_sum_ms": round(t5 * (n - 1), 3),
                          "symv_alg_GBs_lower": round(8.0 * n * (2 * n - 1) / 12.0 / (t5 * 1e-3) / 1e9, 1)}))
