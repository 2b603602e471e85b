"""Problem generators for the Jordan-reduction path: the reference's own test
problems plus the synthetic configs of BASELINE.json.  Pure NumPy/SciPy; no
dependence on the HIP library, so the oracle tests can use them as well.

All vectorisations are column-major (Julia ``vec``).
"""
from __future__ import annotations

import itertools
import math

import numpy as np
import scipy.sparse as sp


# ---------------------------------------------------------------------------
# Lovasz theta' problems (test/sd_problems.jl:16-27 in the reference)
# ---------------------------------------------------------------------------
def theta_prime_problem(adj):
    """C = ones(N^2), A = [vec(Adj)'; vec(I)'], b = [0, 1]
    (same form as test/sd_problems.jl:22-26)."""
    adj = np.asarray(adj, dtype=np.float64)
    n = adj.shape[0]
    C = np.ones(n * n)
    A = np.vstack([adj.ravel(order="F"), np.eye(n).ravel(order="F")])
    b = np.array([0.0, 1.0])
    return C, A, b


def er_graph_adjacency(q):
    """Erdos-Renyi polarity graph on PG(2,q) -- test/sd_problems.jl:16-21."""
    pts = [[0, 0, 1]] + [[0, 1, b] for b in range(q)] + [[1, a, b] for a in range(q) for b in range(q)]
    pts = np.array(pts, dtype=np.int64)
    dots = (pts @ pts.T) % q
    adj = (dots == 0)
    np.fill_diagonal(adj, False)  # x != y
    return adj.astype(np.float64)


def petersen_adjacency():
    """Petersen graph = Kneser graph K(5,2) (BASELINE.json configs[0])."""
    verts = list(itertools.combinations(range(5), 2))
    n = len(verts)
    adj = np.zeros((n, n))
    for i, a in enumerate(verts):
        for j, b in enumerate(verts):
            if not set(a) & set(b):
                adj[i, j] = 1.0
    return adj


def gnp_adjacency(n, p=0.5, seed=0):
    """Seeded symmetric Bernoulli(p) adjacency, G(n,p) (BASELINE.json configs[1])."""
    rng = np.random.default_rng(seed)
    upper = np.triu(rng.random((n, n)) < p, 1)
    adj = upper | upper.T
    return adj.astype(np.float64)


# ---------------------------------------------------------------------------
# QAP relaxation (test/sd_problems.jl:63-105, test/qap.jl:3-11)
# ---------------------------------------------------------------------------
def read_qapdata(path):
    """QAPLIB .dat reader -- test/qap.jl:3-11.  Returns (A, B) n x n."""
    toks = open(path).read().split()
    n = int(toks[0])
    vals = np.array(toks[1:1 + 2 * n * n], dtype=np.float64)
    A = vals[: n * n].reshape(n, n)
    B = vals[n * n:].reshape(n, n)
    return A, B


def qap_constraints(n):
    """``__qap_Ab`` -- test/sd_problems.jl:63-92.  A is (2n+1) x n^4, CSR."""
    In = sp.identity(n, format="csr")
    Jn = np.ones((n, n))
    rows = []
    b = []

    def vec_kron(X, Y):
        K = sp.kron(X, Y, format="coo")
        # column-major linear index of (r, c) in an N x N matrix
        N = K.shape[0]
        lin = K.row + K.col * N
        return sp.csr_matrix((K.data, (np.zeros_like(lin), lin)), shape=(1, N * N))

    for j in range(n):
        Ejj = sp.csr_matrix(([1.0], ([j], [j])), shape=(n, n))
        rows.append(vec_kron(In, Ejj))
        b.append(1.0)
        if j < n - 1:
            rows.append(vec_kron(Ejj, In))
            b.append(1.0)
    JmI = sp.csr_matrix(Jn - np.eye(n))
    rows.append(vec_kron(In, JmI) + vec_kron(JmI, In))
    b.append(0.0)
    rows.append(sp.csr_matrix(np.ones((1, n ** 4))))
    b.append(float(n * n))
    return sp.vstack(rows, format="csr"), np.array(b)


def qap_problem(flowA, flowB):
    """``QuadraticAssignment(flowA, flowB)`` -- test/sd_problems.jl:94-105."""
    flowA = np.asarray(flowA, dtype=np.float64)
    flowB = np.asarray(flowB, dtype=np.float64)
    n = flowA.shape[0]
    assert flowA.shape == flowB.shape == (n, n)
    A, b = qap_constraints(n)
    C = np.kron(flowA, flowB)
    if not np.array_equal(C, C.T):
        C = (C + C.T) / 2
    return C.ravel(order="F"), A, b


# ---------------------------------------------------------------------------
# synthetic Jordan algebras given directly as partitions (BASELINE.json configs[3,4])
# ---------------------------------------------------------------------------
def symmetric_circulant_labels(m):
    """Symmetric circulant scheme on Z_m: label(i,j) = 1 + min(|i-j|, m-|i-j|).
    m//2 + 1 classes, commutative (all blocks of size 1)."""
    i = np.arange(m)
    d = np.abs(i[:, None] - i[None, :])
    return 1 + np.minimum(d, m - d)


def kron_labels(L1, L2):
    """Labels of the Kronecker product algebra: (a, b) -> (a-1)*d2 + b, first index
    slow.  dims multiply."""
    d2 = int(L2.max())
    n1, n2 = L1.shape[0], L2.shape[0]
    out = (L1[:, None, :, None] - 1) * d2 + L2[None, :, None, :]
    return out.reshape(n1 * n2, n1 * n2)


def ones_labels(m):
    return np.ones((m, m), dtype=np.int64)


def permute_labels(L, seed):
    """Conjugate by a seeded permutation so the structure is not index-aligned."""
    rng = np.random.default_rng(seed)
    p = rng.permutation(L.shape[0])
    return L[np.ix_(p, p)]


def canonical_labels(L):
    """First-occurrence (column-major) relabel, 0 preserved -- the canonical form
    the reference's ``Partition`` constructor produces (src/partitions.jl:37-60)."""
    flat = np.asarray(L).ravel(order="F")
    uniq, first, inv = np.unique(flat, return_index=True, return_inverse=True)
    order = np.argsort(first, kind="stable")
    rank = np.zeros(len(uniq), dtype=np.int64)
    k = 0
    for u in order:
        if uniq[u] == 0:
            continue
        k += 1
        rank[u] = k
    return rank[inv.reshape(-1)].reshape(L.shape, order="F"), k


def complete_scheme_labels(k):
    """Trivial association scheme on k points: {I, J - I}."""
    return 2 - np.eye(k, dtype=np.int64)


def synthetic_jordan_partition(n, kind="circulant", seed=0):
    """Seeded partition with known closure.

    kind="circulant":  symmetric circulant scheme on Z_m  (x)  {I, J-I} on k points,
                       m*k = n, m ~ sqrt(n)/2 (m = 32, k = 128 at n = 4096: 34 classes).
                       A commutative association scheme: Jordan-closed, contains I,
                       so ``admissible_subspace`` must return it unchanged and every
                       block has size 1.
    kind="er7":        ER(7) coherent algebra (N=57, dim 18, blocks [2,2,2,2,3])
                       (x) {I, J-I} on k = n/57 points: non-commutative, dim 36,
                       blocks [2,2,2,2,3] twice.  ``base`` must be the 57x57 label
                       matrix (tests pass the golden one).
    Returns (labels int64 n x n canonical, dim).
    """
    if kind == "circulant":
        m = 1
        while (2 * m) * (2 * m) * 4 <= n:
            m *= 2
        m = max(m, 2)
        while n % m:
            m -= 1
        L = kron_labels(symmetric_circulant_labels(m), complete_scheme_labels(n // m))
    else:
        raise ValueError(kind)
    L = permute_labels(L, seed)
    return canonical_labels(L)


def kron_with_complete(base_labels, k, seed=0):
    """base (x) {I, J-I}_k, permuted and canonicalised."""
    L = kron_labels(np.asarray(base_labels, dtype=np.int64), complete_scheme_labels(k))
    return canonical_labels(permute_labels(L, seed))


def partition_as_sdp(L, seed=0):
    """Wrap a partition as an SDP (C = sum_i c_i 1[P==i], A = one trace row, b=[1])
    so that ``admissible_subspace`` must recover (the closure of) it."""
    rng = np.random.default_rng(seed)
    d = int(L.max())
    c = np.concatenate([[0.0], np.round(rng.random(d) * 1000 + 1)])
    C = c[L].ravel(order="F")
    n = L.shape[0]
    A = np.eye(n).ravel(order="F")[None, :]
    b = np.array([1.0])
    return C, A, b


def grid_qap_instance(rows=5, cols=6, seed=0, symmetric_flow=True):
    """Synthetic "nug30-shaped" QAP (BASELINE.json configs[2]; the real nug30 data is not in the
    reference's test/qapdata): n = rows*cols facilities, distance = Manhattan metric of the
    rows x cols grid, flow = seeded sparse symmetric integer matrix.  With
    ``symmetric_flow`` the flow is made invariant under the grid's reflections so that the
    relaxation has a non-trivial symmetry group (otherwise dim(P) = (N^2+N)/2)."""
    n = rows * cols
    pts = np.array([(i, j) for i in range(rows) for j in range(cols)])
    dist = np.abs(pts[:, None, :] - pts[None, :, :]).sum(-1).astype(np.float64)
    rng = np.random.default_rng(seed)
    flow = np.zeros((n, n))
    mask = np.triu(rng.random((n, n)) < 0.25, 1)
    vals = rng.integers(1, 6, size=(n, n))
    flow[mask] = vals[mask]
    flow = flow + flow.T
    if symmetric_flow:
        idx = np.arange(n).reshape(rows, cols)
        perms = [idx.ravel(), idx[::-1, :].ravel(), idx[:, ::-1].ravel(), idx[::-1, ::-1].ravel()]
        flow = sum(flow[np.ix_(p, p)] for p in perms)
    return flow, dist


# ---------------------------------------------------------------------------
# theta'-type SDPs whose Jordan closure is a known product scheme (bench workloads that need
# several refinement rounds: the loop starts from {diagonal, edges, non-edges})
# ---------------------------------------------------------------------------
def cartesian_with_complete_adjacency(base_adj, k):
    """Cartesian product G [] K_k: (a,u) ~ (b,v) iff (a ~ b and u = v) or (a = b and u != v).
    Vertex (a, u) has index a*k + u (first index slow, as in ``kron_labels``)."""
    base_adj = np.asarray(base_adj, dtype=np.float64)
    m = base_adj.shape[0]
    return np.kron(base_adj, np.eye(k)) + np.kron(np.eye(m), np.ones((k, k)) - np.eye(k))


def cycle_adjacency(m):
    i = np.arange(m)
    d = np.abs(i[:, None] - i[None, :])
    return (np.minimum(d, m - d) == 1).astype(np.float64)


def theta_prime_product_problem(base_adj, base_labels, k, seed=0):
    """theta' SDP (test/sd_problems.jl:22-26 form) of the Cartesian product ``base [] K_k`` under a
    seeded vertex permutation, together with the canonical labels of the product scheme
    ``base_labels (x) {I, J-I}_k`` under the same permutation -- the partition the Jordan
    reduction has to arrive at when ``base_labels`` is the closure of the base problem.
    Returns (C, A, b, labels, dim)."""
    adj = cartesian_with_complete_adjacency(base_adj, k)
    n = adj.shape[0]
    p = np.random.default_rng(seed).permutation(n)
    adj = adj[np.ix_(p, p)]
    L = kron_labels(np.asarray(base_labels, dtype=np.int64), complete_scheme_labels(k))[np.ix_(p, p)]
    L, d = canonical_labels(L)
    C, A, b = theta_prime_problem(adj)
    return C, A, b, L, d
