"""Refinement probes of sdpsr_profile_kernel at N = 4096 (len = 16.7 M): kind 3 (steady state: the class-count
prediction of the previous call stands) and kind 11 (every call as the FIRST refinement of an admissible_subspace call:
prediction reset, a many-classes input pays the overflowing first pass, the sample and the path it selects).
Usage: python tools/refine_probe.py [cold|warm|both] [n] [classes ...]"""
import sys, os, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from __graft_entry__ import load_package
pkg = load_package()
prof = pkg._lib.load_prof_library()
mode = sys.argv[1] if len(sys.argv) > 1 else "both"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
path = [a.split("=")[1] for a in sys.argv[3:] if a.startswith("path=")]  # path=no_mid, path=hash ... (Context(refine_path=))
path = path[0] if path else "auto"
classes = [int(x) for x in sys.argv[3:] if not x.startswith("path=")] or [34, 3000, 30000, 262144, n * n // 2]
ms = C.c_double(0)
for cls in classes:
    for kind, name in ((3, "warm"), (11, "cold")):
        if mode not in (name, "both"):
            continue
        with pkg.Context(seed=3, refine_path=path) as ctx:  # a fresh ctx per probe
            ctx.check(prof.sdpsr_profile_kernel(ctx._h, kind, n, cls, 10, C.byref(ms)))
            gbs = 16.0 * n * n / (ms.value * 1e-3) / 1e9
            print(f"refine n={n} classes={cls:9d} {name}{'' if path == 'auto' else ' ' + path}: {ms.value:8.4f} ms per call  {gbs:8.1f} GB/s of 16 B/entry = {gbs / 8000:.4f} of HBM", flush=True)
