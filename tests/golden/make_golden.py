"""Generates tests/golden/golden_partitions.npz with the CPU oracle
(oracle/sdpsr_oracle.py), cross-checked against the reference's own pinned
answers before anything is written:

  test/lovasz.jl:6,8,22,24,38,40    ER(3/5/7): dim 12/15/18, blocks [2,2,3] / [2,2,2,3] / [2,2,2,2,3]
  test/qap.jl:20,23                 esc16j: dim 150, blocks ten 1's + five 7's
  test/numerical_issues.jl:1-66     64x64 partition with 1312 classes (already canonical)
  test/runtests.jl:22-25,40         3x3 refine / desymmetrize triples

The canonical label matrix does not depend on the random numbers drawn (it is the
coarsest Jordan-closed partition, labelled by first occurrence), so these are
valid expected outputs for any correct implementation.  Run from the repo root:
    python tests/golden/make_golden.py
"""
import importlib.util
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "oracle"))
import sdpsr_oracle as O  # noqa: E402

spec = importlib.util.spec_from_file_location("problems", ROOT / "tests" / "problems.py")
pr = importlib.util.module_from_spec(spec)
spec.loader.exec_module(pr)

PINS = {  # name -> (dim, sorted blkSizes)
    "petersen": (3, [1, 1, 1]),
    "er3": (12, [2, 2, 3]),
    "er5": (15, [2, 2, 2, 3]),
    "er7": (18, [2, 2, 2, 2, 3]),
    "esc16j": (150, [1] * 10 + [7] * 5),
    "numerical_issues": (1312, [16, 48]),
}


def problems():
    yield "petersen", pr.theta_prime_problem(pr.petersen_adjacency())
    for q in (3, 5, 7):
        yield f"er{q}", pr.theta_prime_problem(pr.er_graph_adjacency(q))
    fa, fb = pr.read_qapdata(ROOT / "tests" / "golden" / "esc16j.dat")
    yield "esc16j", pr.qap_problem(fa, fb)


def main():
    out = {}
    for name, (C, A, b) in problems():
        mats = []
        for seed in (1, 2, 3):  # RNG independence of the canonical matrix
            P = O.admissible_subspace(C, A, b, rng=np.random.default_rng(seed))
            mats.append(P.matrix)
        assert all(np.array_equal(mats[0], m) for m in mats), name
        sizes, _, _ = O.block_diagonalize(P, rng=np.random.default_rng(11))
        assert (P.nparts, sorted(sizes)) == PINS[name], (name, P.nparts, sorted(sizes))
        out[f"{name}_P"] = mats[0].astype(np.uint32)
        out[f"{name}_blk"] = np.array(sorted(sizes), dtype=np.int32)
        print(name, P.nparts, sorted(sizes))
    L = np.loadtxt(ROOT / "tests" / "golden" / "numerical_issues_P64.txt", dtype=np.int64)
    P = O.partition_from_labels(L)
    assert np.array_equal(P.matrix, L) and P.nparts == 1312
    sizes, _, _ = O.block_diagonalize(P, epsilon=1e-7, rng=np.random.default_rng(11))
    assert sorted(sizes) == PINS["numerical_issues"][1]
    out["numerical_issues_P"] = L.astype(np.uint32)
    out["numerical_issues_blk"] = np.array(sorted(sizes), dtype=np.int32)
    # synthetic closures used at bench sizes, small instances
    for n in (64, 256):
        Ls, d = pr.synthetic_jordan_partition(n, seed=n)
        C, A, b = pr.partition_as_sdp(Ls, seed=1)
        P = O.admissible_subspace(C, A, b, rng=np.random.default_rng(4))
        assert np.array_equal(P.matrix, Ls), n  # the scheme is already Jordan-closed
        sizes, _, _ = O.block_diagonalize(P, rng=np.random.default_rng(11))
        out[f"circ{n}_P"] = Ls.astype(np.uint32)
        out[f"circ{n}_blk"] = np.array(sorted(sizes), dtype=np.int32)
        print("circ", n, d, sorted(sizes))
    np.savez_compressed(ROOT / "tests" / "golden" / "golden_partitions.npz", **out)


if __name__ == "__main__":
    main()
