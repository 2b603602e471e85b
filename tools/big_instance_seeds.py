"""Seed sweep on the two theta' bench instances (N = 4096 / 4104): admissible_subspace from device-resident
inputs must give the generator's closure for every seed; iteration counts are tallied.
usage: big_instance_seeds.py [seeds=40] [hint=1]"""
import sys, os, collections, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from __graft_entry__ import load_package
pkg = load_package(); pr = pkg.problems
from importlib import import_module
Lm = import_module(pkg.__name__ + "._lib")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
use_hint = int(sys.argv[2]) if len(sys.argv) > 2 else 1
flags = int(sys.argv[3]) if len(sys.argv) > 3 else 0  # sdpsr_opts.flags for A/B sweeps
channels = int(sys.argv[4]) if len(sys.argv) > 4 else 0
dev = torch.device("cuda:0")
gold = np.load(os.path.join(ROOT, "tests", "golden", "golden_partitions.npz"))["er7_P"].astype(np.int64)
for name, prob in (("theta_c32xk128", lambda: pr.theta_prime_product_problem(pr.cycle_adjacency(32), pr.symmetric_circulant_labels(32), 128, seed=1)),
                   ("theta_er7xk72", lambda: pr.theta_prime_product_problem(pr.er_graph_adjacency(7), gold, 72, seed=1))):
    Cv, A, b, Ls, d = prob()
    setup = pkg.admissible_setup(Cv, A, b)
    n, CL, X0L, U = setup
    r = U.shape[1]
    tCL, tX0 = torch.from_numpy(CL).to(dev), torch.from_numpy(X0L).to(dev)
    tU = torch.from_numpy(np.ascontiguousarray(U.T)).to(dev)
    tP = torch.empty(n * n, dtype=torch.int32, device=dev)
    golden = torch.from_numpy(np.ascontiguousarray(Ls.ravel(order="F")).astype(np.int32)).to(dev)
    bad, iters = 0, collections.Counter()
    for seed in range(nseeds):
        with pkg.Context(seed=7000 + seed, flags=flags, channels=channels) as ctx:
            dd, it = C.c_int64(0), C.c_int32(0)
            if use_hint and setup.hint:
                ctx._lib.sdpsr_hint_symmetric_basis(ctx._h, setup.hint)
            ctx.check(ctx._lib.sdpsr_admissible_subspace(ctx._h, n, C.c_void_p(tCL.data_ptr()), C.c_void_p(tX0.data_ptr()),
                                                         C.c_void_p(tU.data_ptr()), r, 1.5e-8, C.c_void_p(tP.data_ptr()),
                                                         C.byref(dd), C.byref(it), None, Lm.MEM_DEVICE))
            iters[it.value] += 1
            if dd.value != d or not bool((tP == golden).all()):
                bad += 1
    print(f"[flags {flags} channels {channels}] {name}: N={n} dim {d}: mismatches {bad} of {nseeds} seeds (hint {use_hint and setup.hint}); iterations {dict(iters)}", flush=True)
