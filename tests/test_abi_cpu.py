"""CPU-side checks: the C-ABI library loads and exports every symbol the header declares;
host logic of the Python mirror agrees with the oracle.  No compute calls (no GPU here)."""
import ctypes as C
import re

import numpy as np


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg.load_library()
    names = pkg._lib.declared_symbols()
    assert len(names) >= 25
    for s in names:
        assert hasattr(lib, s), s
    assert lib.sdpsr_version() == 1
    assert lib.sdpsr_status_string(3) == b"DIMENSION_MISMATCH"


def test_header_cites_reference_lines(pkg):
    txt = open(pkg._lib.HEADER_PATH).read()
    assert len(re.findall(r"src/[a-z_]+\.jl:\d+", txt)) >= 20


def test_host_setup_matches_oracle(pkg, problems, oracle):
    for q in (3, 5):
        Cv, A, b = problems.theta_prime_problem(problems.er_graph_adjacency(q))
        n, CL, X0L, U = pkg.admissible_setup(Cv, A, b)
        n2, U2, CL2, X02 = oracle.admissible_setup(Cv, A, b)
        assert n == n2
        assert np.allclose(CL, CL2.ravel(order="F"), atol=1e-12)
        assert np.allclose(X0L, X02.ravel(order="F"), atol=1e-12)
        assert np.allclose(U @ U.T, U2 @ U2.T, atol=1e-12)  # same projector


def test_hash_header_compiles_for_host_and_matches_python():
    """sdpsr_hash.h is plain integer arithmetic; restate fmix64 here and compare through a
    tiny C program compiled with gcc (host side of the shared header)."""
    import os
    import subprocess
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = r'''
    #include <stdio.h>
    #include "sdpsr_hash.h"
    int main(){ unsigned long long k = sdpsr_stream_key(42, 3);
      printf("%llu %llu %.17g %d %.17g\n", k, sdpsr_class_bits(k, 7), sdpsr_class_uniform(k, 7),
             sdpsr_class_i8(sdpsr_class_bits(k, 7), 2), sdpsr_clamp_round(0.0625*(1-2e-16), 1.4901161193847656e-8, 1e7)); return 0; }
    '''
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.cpp"), "w").write(src)
        subprocess.check_call(["g++", "-O1", "-I", os.path.join(root, "sdpsymmetryreduction.jl_amd", "csrc"),
                               os.path.join(d, "t.cpp"), "-o", os.path.join(d, "t")])
        out = subprocess.check_output([os.path.join(d, "t")]).decode().split()
    M = (1 << 64) - 1

    def fmix(z):
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
        return z ^ (z >> 31)

    k = fmix((42 + 0x9E3779B97F4A7C15 * 4) & M)
    bits = fmix((k + 0x9E3779B97F4A7C15 * 7) & M)
    assert int(out[0]) == k and int(out[1]) == bits
    assert float(out[2]) == (bits >> 11) / 2.0 ** 53
    b2 = (bits >> 16) & 0xFF
    assert int(out[3]) == (b2 - 256 if b2 >= 128 else b2)
    assert float(out[4]) == 0.0625
