import importlib.util
import os
import pathlib
import sys

import numpy as np
import pytest

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")


def load_package():
    """The package directory is literally ``sdpsymmetryreduction.jl_amd`` (not an
    importable dotted name), so it is loaded by path under the name ``sdpsr_amd``."""
    from __graft_entry__ import load_package as _lp
    return _lp()


@pytest.fixture(scope="session")
def pkg():
    # torch initialises its HIP side lazily, and on this image it reports "no ROCm-capable device" when it does so AFTER
    # another library of the process has created streams on the device: let torch look at the GPU first, whatever subset
    # of the tests runs (harmless on a CPU-only box: is_available() is False there)
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:  # noqa: BLE001
        pass
    return load_package()


@pytest.fixture(scope="session")
def problems(pkg):
    return pkg.problems


@pytest.fixture(scope="session")
def oracle():
    import sdpsr_oracle
    return sdpsr_oracle


@pytest.fixture(scope="session")
def golden():
    return np.load(ROOT / "tests" / "golden" / "golden_partitions.npz")


@pytest.fixture(scope="session")
def gpu_ctx(pkg):
    """One device context for the whole GPU session (one process on the card)."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    ctx = pkg.Context(device=0, seed=1234)
    yield ctx
    ctx.close()
