"""CPU check of the algorithm behind csrc/kernels_stedc.hip: tools/probes/dc_prototype.py restates the tridiagonal divide
and conquer (tearing into leaves of 32, deflation, secular roots by bisection on the bit pattern of the shift,
Gu / Eisenstat vectors, merged order) step by step in NumPy.  The GPU tests compare the kernels with LAPACK; this one keeps
the restatement itself pinned to LAPACK on the matrices that break careless implementations."""
import importlib.util
import pathlib

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parents[1]


def _load():
    spec = importlib.util.spec_from_file_location("dc_prototype", ROOT / "tools" / "probes" / "dc_prototype.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_divide_and_conquer_prototype_against_lapack():
    dc = _load()
    rng = np.random.default_rng(5)
    n = 150
    cases = {
        "random": (rng.standard_normal(n), rng.standard_normal(n - 1)),
        "wilkinson": (np.abs(np.arange(n) - n // 2).astype(float), np.ones(n - 1)),
        "glued wilkinson": (np.tile(np.abs(np.arange(21) - 10.0), n // 21 + 1)[:n], np.where((np.arange(n - 1) + 1) % 21 == 0, 1e-8, 1.0)),
        "graded": (10.0 ** (-np.arange(n) * 12.0 / n), 10.0 ** (-np.arange(n - 1) * 12.0 / n)),
        "decoupled": (rng.standard_normal(n), rng.standard_normal(n - 1) * (rng.random(n - 1) < 0.5)),
        "identity": (np.ones(n), np.zeros(n - 1)),
    }
    for name, (d, e) in cases.items():
        T = np.diag(d) + np.diag(e, 1) + np.diag(e, -1)
        w, Z = dc.stedc(d, e)
        wl = np.linalg.eigvalsh(T)
        sc = np.abs(wl).max()
        assert np.all(np.diff(w) >= 0), name
        assert np.abs(w - wl).max() <= 1e-13 * sc, name
        assert np.abs(T @ Z - Z * w).max() <= 1e-13 * sc, name
        assert np.abs(Z.T @ Z - np.eye(n)).max() < 1e-13, name
