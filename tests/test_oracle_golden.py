"""Pins the CPU oracle against the reference's own known answers and
against the committed golden fixtures.  CPU only."""
import pathlib

import numpy as np
import pytest

ROOT = pathlib.Path(__file__).resolve().parents[1]


def test_round_to_zero(oracle):
    # test/runtests.jl:11
    assert oracle.clamptol(1e-10) == 0


def test_partition_counts(oracle):
    # test/runtests.jl:13-20
    rng = np.random.default_rng(0)
    M = rng.integers(1, 11, size=(10, 10))
    M[0, 0] = 0
    assert oracle.partition_from_labels(M).nparts == len(np.unique(M)) - 1
    assert oracle.partition_from_values(M.astype(float)).nparts == len(np.unique(M)) - 1
    M = rng.integers(1, 11, size=(10, 10))
    assert oracle.partition_from_labels(M).nparts == len(np.unique(M))
    assert oracle.partition_from_values(M.astype(float)).nparts == len(np.unique(M))


def test_vectorised_scan_equals_literal_scan(oracle):
    rng = np.random.default_rng(1)
    M = rng.integers(0, 7, size=(9, 9)).astype(float) * 0.25
    M[3, 3] = -0.0  # isequal: -0.0 is its own key
    a = oracle.partition_from_values(M)
    b = oracle.partition_from_values_scan(M)
    assert a == b


def test_refine_triple(oracle):
    # test/runtests.jl:22-25
    P1 = oracle.partition_from_labels(np.array([[1, 2, 2], [2, 3, 3], [2, 3, 3]]))
    P2 = oracle.partition_from_labels(np.array([[1, 1, 2], [1, 1, 2], [1, 1, 3]]))
    P3 = oracle.partition_from_labels(np.array([[1, 2, 4], [2, 3, 5], [2, 3, 6]]))
    assert oracle.refine(P1, P2) == P3
    # test/runtests.jl:27
    assert oracle.partition_from_values(oracle.randomize(P1, np.random.default_rng(3))) == P1


def test_desymmetrize_triple(oracle):
    # test/runtests.jl:40
    P1 = oracle.partition_from_labels(np.array([[1, 2, 2], [2, 3, 3], [2, 3, 3]]))
    out = oracle.desymmetrize(P1)
    assert out.nparts == 4
    assert np.array_equal(out.matrix, np.array([[1, 3, 3], [2, 4, 4], [2, 4, 4]]))


def test_cyclic_c3_is_not_real_diagonalizable(oracle):
    # test/runtests.jl:50-56
    C3 = np.array([[1, 3, 2], [2, 1, 3], [3, 2, 1]])
    with pytest.raises(oracle.InvalidDecompositionField):
        oracle.block_diagonalize(oracle.partition_from_labels(C3))


def test_label_overflow_like_uint16(oracle):
    # src/partitions.jl:63,84: InexactError once the pair code exceeds typemax(UInt16)
    P1 = oracle.partition_from_labels(np.arange(1, 301 * 301 + 1).reshape(301, 301) % 300 + 1)
    P2 = oracle.partition_from_labels((np.arange(301 * 301).reshape(301, 301) // 7) % 300 + 1)
    with pytest.raises(oracle.LabelOverflow):
        oracle.refine(P1, P2, label_bits=16)


@pytest.mark.parametrize("name,dim,blocks", [
    ("petersen", 3, [1, 1, 1]),
    ("er3", 12, [2, 2, 3]),            # test/lovasz.jl:6,8
    ("er5", 15, [2, 2, 2, 3]),         # test/lovasz.jl:22,24
    ("er7", 18, [2, 2, 2, 2, 3]),      # test/lovasz.jl:38,40
])
def test_theta_prime_pins(oracle, problems, golden, name, dim, blocks):
    adj = problems.petersen_adjacency() if name == "petersen" else problems.er_graph_adjacency(int(name[2:]))
    C, A, b = problems.theta_prime_problem(adj)
    P = oracle.admissible_subspace(C, A, b, rng=np.random.default_rng(99))
    assert P.nparts == dim
    assert np.array_equal(P.matrix, golden[f"{name}_P"])
    sizes, blks, _ = oracle.block_diagonalize(P, rng=np.random.default_rng(5))
    assert sorted(sizes) == blocks == list(golden[f"{name}_blk"])
    # spectrum invariant
    x = np.random.default_rng(8).random(P.nparts)
    full, blk = oracle.spectrum_invariant(P, blks, x)
    assert len(full) == len(blk)
    assert np.allclose(full, blk, rtol=1e-6, atol=1e-9)


def test_esc16j_pin(oracle, problems, golden):
    # test/qap.jl:13-23
    fa, fb = problems.read_qapdata(ROOT / "tests" / "golden" / "esc16j.dat")
    C, A, b = problems.qap_problem(fa, fb)
    assert A.shape == (33, 65536)
    P = oracle.admissible_subspace(C, A, b, rng=np.random.default_rng(17))
    assert P.nparts == 150
    assert np.array_equal(P.matrix, golden["esc16j_P"])
    sizes, _, _ = oracle.block_diagonalize(P, rng=np.random.default_rng(5))
    assert sorted(sizes) == [1] * 10 + [7] * 5


def test_numerical_issues_partition(oracle, golden):
    # test/numerical_issues.jl:1-66,91-94 (10 000 runs there; 200 here keeps CPU CI short)
    L = golden["numerical_issues_P"].astype(np.int64)
    P = oracle.partition_from_labels(L)
    assert P.nparts == 1312 and np.array_equal(P.matrix, L)
    rng = np.random.default_rng(2)
    for _ in range(200):
        oracle.eigen_decomposition(P, atol=1e-7, rng=rng)
    sizes, _, _ = oracle.block_diagonalize(P, epsilon=1e-7, rng=rng)
    assert sorted(sizes) == [16, 48]


def test_trunc_rounding_splits_esc16j(oracle, problems):
    """Documents SURVEY.md fact 4: the literal truncation of src/utils.jl:49-53 puts
    0.0625 on a bucket edge and splits its class; nearest rounding does not."""
    v = np.array([0.0625, 0.0625 - 1e-17, 0.0625 * (1 - 2e-16)])
    assert len(np.unique(oracle.clamp_round(v, round_mode="nearest"))) == 1
    assert len(np.unique(oracle.clamp_round(v, round_mode="trunc"))) == 2


def test_synthetic_scheme_is_fixed_point(oracle, problems, golden):
    for n in (64, 256):
        Ls, d = problems.synthetic_jordan_partition(n, seed=n)
        assert np.array_equal(Ls, golden[f"circ{n}_P"])
        C, A, b = problems.partition_as_sdp(Ls, seed=1)
        P = oracle.admissible_subspace(C, A, b, rng=np.random.default_rng(4))
        assert np.array_equal(P.matrix, Ls)


def test_basis_image_fast_equals_literal(oracle, problems, golden):
    P = oracle.partition_from_labels(golden["er5_P"].astype(np.int64))
    Q = oracle.diagonalize(P, atol=oracle.RTOL_DEFAULT, rng=np.random.default_rng(1))
    a = oracle.basis_image(Q, P)
    b = oracle.basis_image_fast(Q, P)
    for i in range(P.nparts):
        for k in range(len(Q)):
            assert np.allclose(a[i][k], b[i][k], atol=1e-12)


def test_complex_path_pins(oracle):
    """test/runtests.jl:43-57: blockDiagonalize(P; complex=true).blkSizes == [1,1,1] for the 4 x 4
    circulant-type partition and for the cyclic group C3 (whose real request must throw
    InvalidDecompositionField, pinned above)."""
    P4 = oracle.partition_from_labels(np.array([[1, 2, 3, 2], [2, 1, 2, 3], [3, 2, 1, 2], [2, 3, 2, 1]]))
    C3 = oracle.partition_from_labels(np.array([[1, 3, 2], [2, 1, 3], [3, 2, 1]]))
    for P in (P4, C3):
        for seed in range(3):
            sizes, blks, Q, Pd = oracle.block_diagonalize_complex(P, rng=np.random.default_rng(seed))
            assert sizes == [1, 1, 1]
            full, blk = oracle.spectrum_invariant_complex(Pd, blks, np.random.default_rng(9).random(Pd.nparts))
            assert len(full) == len(blk) and np.allclose(full, blk, atol=1e-8)
