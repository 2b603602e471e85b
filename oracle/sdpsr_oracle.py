"""CPU ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

NumPy/SciPy restatement of the Jordan-reduction hot path of
DanielBrosch/SDPSymmetryReduction.jl (reference v0.2.1, pure Julia).  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this module, and only as the checker / the timed CPU baseline.  The
product path (``sdpsymmetryreduction.jl_amd``) never imports it.

Pinning: the reference is Julia and cannot run here (no ``julia`` binary in the
image).  The oracle is pinned by the reference's own known answers
(``tests/test_oracle_golden.py``): ``test/runtests.jl:13-27,40,56``,
``test/lovasz.jl:6,8,22,24,38,40``, ``test/qap.jl:20,23``,
``test/numerical_issues.jl:1-66,91-94``.

Every function cites the reference lines it restates (paths relative to
``/root/reference``).  Matrices are NumPy 2-D arrays; every scan that the
reference does with ``eachindex``/``zip(M, ...)`` is column-major, restated here
with ``ravel(order="F")``.

Documented deviation: ``unsafe_round`` (``src/utils.jl:49-53``)
truncates the mantissa; values sitting exactly on a bucket edge (0.0625 = 0.5*2^-3)
are then split by last-bit noise.  ``round_mode="nearest"`` (default) rounds the
scaled mantissa to nearest instead; ``round_mode="trunc"`` is the literal rule.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np
import scipy.linalg as sla

RTOL_DEFAULT = math.sqrt(np.finfo(np.float64).eps)  # Base.rtoldefault(Float64)


# --------------------------------------------------------------------------
# exceptions (src/eigen_decomposition.jl:140-161, src/diagonalize.jl:1-23)
# --------------------------------------------------------------------------
class InvalidDecompositionField(Exception):
    """src/eigen_decomposition.jl:140-150"""


class NumericalInconsistency(Exception):
    """src/eigen_decomposition.jl:152-161"""


class DimensionMismatch(Exception):
    """src/diagonalize.jl:4-9"""


class LabelOverflow(Exception):
    """Julia's InexactError when labels do not fit T (src/partitions.jl:63)."""


# --------------------------------------------------------------------------
# utils.jl
# --------------------------------------------------------------------------
def clamptol(a, atol=RTOL_DEFAULT):
    """src/utils.jl:14-26 -- |x| < atol -> 0."""
    a = np.asarray(a)
    return np.where(np.abs(a) < atol, 0.0, a)


def clamp_round(a, atol=RTOL_DEFAULT, sigdigits=None, round_mode="nearest"):
    """src/utils.jl:34-53 (_clamp_round! + unsafe_round).

    |a| < atol -> 0; otherwise mantissa x in [0.5,1) (frexp) is reduced to
    ``sigdigits = floor(-log10(atol))`` decimal digits and re-scaled with ldexp.
    """
    a = np.asarray(a, dtype=np.float64)
    if sigdigits is None:
        sigdigits = int(math.floor(-math.log10(atol)))
    scale = float(10 ** sigdigits)
    x, e = np.frexp(a)
    if round_mode == "trunc":
        y = np.trunc(scale * x) / scale  # unsafe_trunc(Int, scale*x)/scale
    elif round_mode == "nearest":
        y = np.rint(scale * x) / scale
    else:
        raise ValueError(round_mode)
    out = np.ldexp(y, e)
    return np.where(np.abs(a) < atol, 0.0, out)


def symmetrize(v, n):
    """src/utils.jl:71-81 -- (M + M')/2."""
    M = np.asarray(v, dtype=np.float64).reshape(n, n, order="F")
    return ((M + M.T) / 2).ravel(order="F")


def rowspace_basis(A, rtol=1e-12):
    """Orthonormal basis U (n^2 x r) of rowspace(A) = colspace(A').

    Stands in for ``qr(A')`` (src/partitions.jl:124): the reference only ever uses
    the factorisation through ``project_colspace!`` (src/utils.jl:62-66), i.e. as
    the orthogonal projector ``A' (qr(A') \\ v) = U U' v``.
    Column-pivoted QR so that a rank-deficient A (SPQR handles it in the
    reference's sparse path) gives the same projector.
    """
    At = np.asarray(A.todense() if hasattr(A, "todense") else A, dtype=np.float64).T
    Q, R, _ = sla.qr(At, mode="economic", pivoting=True)
    d = np.abs(np.diag(R))
    r = int(np.sum(d > rtol * d.max())) if d.size and d.max() > 0 else 0
    return np.ascontiguousarray(Q[:, :r])


def project_colspace(v, U):
    """src/utils.jl:62-66 with Afact folded into the orthonormal basis U."""
    return U @ (U.T @ v)


def min_norm_solution(A, b):
    """``Krylov.craig(A, b)`` (src/partitions.jl:137): min-norm solution of Ax=b."""
    Ad = np.asarray(A.todense() if hasattr(A, "todense") else A, dtype=np.float64)
    x, *_ = np.linalg.lstsq(Ad, np.asarray(b, dtype=np.float64), rcond=None)
    return x


# --------------------------------------------------------------------------
# Partition (src/partitions.jl:1-75)
# --------------------------------------------------------------------------
@dataclass
class Partition:
    """src/partitions.jl:6-17.  ``matrix`` holds labels 0..nparts (int64)."""

    nparts: int
    matrix: np.ndarray

    def __eq__(self, other):  # src/partitions.jl:16-17
        return self.nparts == other.nparts and np.array_equal(self.matrix, other.matrix)

    @property
    def shape(self):
        return self.matrix.shape


def dim(P: Partition) -> int:
    return P.nparts


def _relabel_fast(flat_keys, zero_key_mask):
    """Canonical relabel of a column-major flat key vector.

    Classes = distinct keys, numbered 1.. in order of first occurrence; entries
    under ``zero_key_mask`` keep label 0.  Vectorised equivalent of the
    ``Dict``/``unique`` scans at src/partitions.jl:24-35 and :44-60.
    """
    uniq, first_idx, inv = np.unique(flat_keys, return_index=True, return_inverse=True)
    inv = inv.reshape(-1)
    is_zero_u = np.zeros(len(uniq), dtype=bool)
    if zero_key_mask.any():
        is_zero_u[inv[zero_key_mask]] = True
    order = np.argsort(first_idx, kind="stable")
    nz = ~is_zero_u[order]
    rank = np.zeros(len(uniq), dtype=np.int64)
    rank[order[nz]] = np.arange(1, int(nz.sum()) + 1)
    return rank[inv], int(nz.sum())


def partition_from_values(M) -> Partition:
    """``Partition{T}(M::AbstractMatrix)`` -- src/partitions.jl:24-35.

    Column-major scan; ``Dict`` seeded with ``0 => 0``; a new value gets ``l+1``.
    ``Dict`` keys compare with ``isequal`` so ``-0.0`` is NOT the zero key.
    """
    M = np.asarray(M)
    flat = M.ravel(order="F")
    if np.issubdtype(flat.dtype, np.floating):
        keys = np.ascontiguousarray(flat, dtype=np.float64).view(np.int64)  # isequal on bits
        zero_mask = keys == 0  # +0.0 only
    else:
        keys = flat.astype(np.int64)
        zero_mask = keys == 0
    labels, k = _relabel_fast(keys, zero_mask)
    return Partition(k, labels.reshape(M.shape, order="F"))


def partition_from_values_scan(M) -> Partition:
    """Literal loop form of src/partitions.jl:24-35 (small inputs; checks the
    vectorised form)."""
    M = np.asarray(M)
    flat = M.ravel(order="F")
    l = 0
    d = {}
    isfloat = np.issubdtype(flat.dtype, np.floating)
    zero_key = (np.float64(0.0).tobytes() if isfloat else 0)
    d[zero_key] = 0
    res = np.zeros(flat.shape, dtype=np.int64)
    for idx, v in enumerate(flat):
        key = np.float64(v).tobytes() if isfloat else int(v)
        if key not in d:
            d[key] = l + 1
        k = d[key]
        if k == l + 1:
            l = k
        res[idx] = k
    return Partition(l, res.reshape(M.shape, order="F"))


def sort_unique(P: Partition) -> Partition:
    """``__sort_unique!`` -- src/partitions.jl:44-60."""
    flat = P.matrix.ravel(order="F").astype(np.int64)
    assert flat.min() >= 0
    labels, k = _relabel_fast(flat, flat == 0)
    return Partition(k, labels.reshape(P.matrix.shape, order="F"))


def partition_from_labels(M) -> Partition:
    """Integer ctor -- src/partitions.jl:37-42."""
    return sort_unique(Partition(0, np.asarray(M, dtype=np.int64)))


def refine(P1: Partition, P2: Partition, label_bits: Optional[int] = None) -> Partition:
    """``refine!`` -- src/partitions.jl:62-66: P1 + P2*(dim(P1)+1), then relabel.

    ``label_bits`` emulates Julia's checked integer width (UInt16 default in
    ``admissible_subspace``, src/partitions.jl:84) and raises ``LabelOverflow``
    where the reference throws ``InexactError``.
    """
    combined = P1.matrix.astype(np.int64) + P2.matrix.astype(np.int64) * (P1.nparts + 1)
    if label_bits is not None and combined.max() >= (1 << label_bits):
        raise LabelOverflow(f"label {combined.max()} does not fit in {label_bits} bits")
    return sort_unique(Partition(0, combined))


def fill(P: Partition, values) -> np.ndarray:
    """``fill!(M, P; values)`` -- src/partitions.jl:68-75."""
    values = np.asarray(values, dtype=np.float64)
    assert len(values) == P.nparts
    table = np.concatenate([[0.0], values])
    return table[P.matrix]


def randomize(P: Partition, rng) -> np.ndarray:
    """``randomize!`` -- src/abstract_part.jl:107-110: uniform [0,1) per class."""
    return fill(P, rng.random(P.nparts))


def constraints(P: Partition) -> List[np.ndarray]:
    """``_constraints`` -- src/diagonalize.jl:42-50 (0-based linear indices)."""
    flat = P.matrix.ravel(order="F")
    order = np.argsort(flat, kind="stable")
    counts = np.bincount(flat, minlength=P.nparts + 1)
    ends = np.cumsum(counts)
    return [order[ends[i - 1]:ends[i]] for i in range(1, P.nparts + 1)]


# --------------------------------------------------------------------------
# admissible_subspace (src/partitions.jl:77-190)
# --------------------------------------------------------------------------
def admissible_setup(C, A, b, atol=RTOL_DEFAULT, round_mode="nearest"):
    """Setup stage -- src/partitions.jl:117-146.  Returns (n, U, CL, X0L)."""
    C = np.asarray(C.todense() if hasattr(C, "todense") else C, dtype=np.float64).reshape(-1)
    n = math.isqrt(len(C))
    assert n * n == len(C)  # :118
    U = rowspace_basis(A)  # :124
    # CL  (:129-134)
    c = C - project_colspace(C, U)
    c = clamp_round(c, atol, round_mode=round_mode)
    c = symmetrize(c, n)
    CL = c.reshape(n, n, order="F")
    # X0L^perp (:137-142)
    x = min_norm_solution(A, b)
    x = symmetrize(x, n)
    x = project_colspace(x, U)
    x = clamp_round(x, atol, round_mode=round_mode)
    X0L = x.reshape(n, n, order="F")
    return n, U, CL, X0L


def admissible_subspace(C, A, b, atol=RTOL_DEFAULT, rng=None, round_mode="nearest",
                        label_bits=None, trace=None, setup=None):
    """``admissible_subspace`` -- src/partitions.jl:109-190."""
    rng = np.random.default_rng(0) if rng is None else rng
    n, U, CL, X0L = setup if setup is not None else admissible_setup(C, A, b, atol, round_mode)
    S = partition_from_values(CL)  # :145
    S = refine(S, partition_from_values(X0L), label_bits)  # :146
    maximal = (n * n + n) // 2  # :148
    current = S.nparts
    it = 0
    while current < maximal:  # :154
        it += 1
        X = randomize(S, rng)  # :159
        x = X.ravel(order="F")
        x = x - project_colspace(x, U)  # :161
        x = clamp_round(x, atol, round_mode=round_mode)  # :162
        S = refine(S, partition_from_values(x.reshape(n, n, order="F")), label_bits)  # :164
        if current != S.nparts:  # :166-168
            X = randomize(S, rng)
        else:
            X = x.reshape(n, n, order="F")
        X2 = X @ X  # :172
        X2 = clamp_round(X2, atol, round_mode=round_mode)  # :173
        S = refine(S, partition_from_values(X2), label_bits)  # :174
        if trace is not None:
            trace.append(S.nparts)
        if current == S.nparts:  # :180-182
            break
        current = S.nparts  # :184
    S.iterations = it
    return S


def desymmetrize(P: Partition, atol=RTOL_DEFAULT, rng=None, round_mode="nearest") -> Partition:
    """``desymmetrize`` -- src/partitions.jl:197-223 (WL step; out of v1 GPU scope,
    restated for the exact test vector test/runtests.jl:40)."""
    rng = np.random.default_rng(0) if rng is None else rng
    P = Partition(P.nparts, P.matrix.copy())
    current = P.nparts
    while True:
        X = randomize(P, rng)
        Y = randomize(P, rng)
        XY = clamp_round(X @ Y, atol, round_mode=round_mode)
        P = refine(P, partition_from_values(XY))
        if current == P.nparts:
            break
        current = P.nparts
    return P


# --------------------------------------------------------------------------
# eigen_decomposition.jl
# --------------------------------------------------------------------------
@dataclass
class EigenDecomposition:
    """src/eigen_decomposition.jl:14-41.  ``ptrs`` are 0-based half-open bounds."""

    values: np.ndarray
    vectors: np.ndarray
    ptrs: List[int] = field(default_factory=list)
    warned_no_gap: bool = False

    def __len__(self):
        return len(self.ptrs) - 1

    def rng(self, i):
        return slice(self.ptrs[i], self.ptrs[i + 1])

    def dim(self, i):
        return self.ptrs[i + 1] - self.ptrs[i]


def eigenspace_ptrs(values, atol):
    """Boundary rule of src/eigen_decomposition.jl:24-38:
    new eigenspace where ``!isapprox(v[i+1], v[i]; atol)`` = |dv| > atol (rtol=0)."""
    ptrs = [0]
    n = len(values)
    warned = False
    for i in range(n):
        if i == n - 1:
            ptrs.append(n)
            break
        if not (abs(values[i + 1] - values[i]) <= atol):
            ptrs.append(i + 1)
            m = max(abs(values[i]), abs(values[i + 1]))
            if abs(values[i + 1] - values[i]) < np.spacing(m):
                warned = True
    return ptrs, warned


def make_eigen_decomposition(values, vectors, atol=None) -> EigenDecomposition:
    if atol is None:
        atol = 1e-12 * len(values)
    ptrs, warned = eigenspace_ptrs(values, atol)
    return EigenDecomposition(np.asarray(values), np.asarray(vectors), ptrs, warned)


def log_histogram(X, num_bins, atol):
    """src/eigen_decomposition.jl:83-98."""
    ax = np.abs(np.asarray(X, dtype=np.float64)).ravel(order="F")
    min_val, max_val = ax.min(), ax.max()
    if min_val < atol:
        min_val = atol
    assert min_val > 0
    edges = np.exp(np.linspace(math.log(min_val), math.log(max_val), num_bins + 1))
    counts = np.zeros(num_bins, dtype=np.int64)
    # note: the reference bins the raw x (not |x|), :92-95; X here is a norm matrix (>=0)
    for x in np.asarray(X, dtype=np.float64).ravel(order="F"):
        gt = np.nonzero(edges > x)[0]
        k = (gt[0] + 1 if len(gt) else num_bins + 1) - 1  # 1-based findfirst, minus 1
        b = min(max(k, 1), num_bins)
        counts[b - 1] += 1
    return counts, edges


def otsu_threshold(X, atol):
    """src/eigen_decomposition.jl:112-139."""
    n_bins = max(int(math.ceil(-math.log10(np.finfo(np.float64).eps))), 4)  # 16
    counts, edges = log_histogram(X, n_bins, atol)
    pdf = counts / counts.sum()
    w = np.cumsum(pdf)
    mu0 = np.cumsum(np.log(edges[:-1]) * pdf)
    muT = mu0[-1]
    with np.errstate(divide="ignore", invalid="ignore"):
        s2 = (muT * w - mu0) ** 2 / (w * (1 - w))
    # Julia argmax: NaN wins (first NaN), else first maximum -- :129
    cand = s2[:-1]
    nan = np.isnan(cand)
    k = int(np.nonzero(nan)[0][0]) if nan.any() else int(np.argmax(cand))
    return edges[k + 1]


def block_norms_inf(QAQ, ed: EigenDecomposition):
    """``block_norms(Q'AQ, eigdec, Inf)`` -- src/eigen_decomposition.jl:177-193."""
    ne = len(ed)
    out = np.zeros((ne, ne))
    aq = np.abs(QAQ)
    for i in range(ne):
        for j in range(i, ne):
            if ed.dim(i) != ed.dim(j):
                v = 0.0
            else:
                v = aq[ed.rng(i), ed.rng(j)].max()
            out[i, j] = out[j, i] = v
    return out


class IntDisjointSets:
    """DataStructures.jl 0.18 ``IntDisjointSets`` (union by rank + path
    compression), as used at src/eigen_decomposition.jl:164,208,214,301.
    0-based here."""

    def __init__(self, n):
        self.parents = list(range(n))
        self.ranks = [0] * n

    def __len__(self):
        return len(self.parents)

    def find_root(self, x):
        p = self.parents
        r = x
        while p[r] != r:
            r = p[r]
        while p[x] != r:  # path compression
            p[x], x = r, p[x]
        return r

    def union(self, x, y):
        x = self.find_root(x)
        y = self.find_root(y)
        if x == y:
            return x
        if self.ranks[x] < self.ranks[y]:
            x, y = y, x
        elif self.ranks[x] == self.ranks[y]:
            self.ranks[x] += 1
        self.parents[y] = x
        return x


def is_consistent(K: IntDisjointSets):
    """``__isconsistent`` -- src/eigen_decomposition.jl:163-167."""
    kp = [K.find_root(i) for i in range(len(K))]
    seen = {}
    for i, r in enumerate(kp):
        seen.setdefault(r, i)
    return all(r == first for r, first in seen.items())


def isomorphism_partition(ed: EigenDecomposition, A, atol):
    """src/eigen_decomposition.jl:201-219."""
    Q = ed.vectors
    QAQ = Q.T @ A @ Q
    norms = block_norms_inf(QAQ, ed)
    thr = otsu_threshold(norms, atol)
    ne = len(ed)
    K = IntDisjointSets(ne)
    for i in range(ne):
        for j in range(i + 1, ne):
            if norms[i, j] >= thr:
                K.union(i, j)
    return K


def _eigen(A):
    """Julia ``eigen(A)`` (src/eigen_decomposition.jl:246): symmetric -> dsyevr
    ascending; otherwise general solver, complex spectrum ->
    InvalidDecompositionField (:247-253)."""
    if np.array_equal(A, A.T):
        vals, Q = sla.eigh(A, driver="evr")
        return vals, Q
    vals, Q = sla.eig(A)
    if np.any(np.abs(vals.imag) > 0) or np.iscomplexobj(Q) and np.any(np.abs(Q.imag) > 0):
        raise InvalidDecompositionField("Float64 requested, ComplexF64 found")
    order = np.argsort(vals.real, kind="stable")
    return vals.real[order], Q.real[:, order]


def eigen_decomposition(P: Partition, atol=None, rng=None):
    """src/eigen_decomposition.jl:236-273."""
    rng = np.random.default_rng(0) if rng is None else rng
    n = P.matrix.shape[0]
    atol = 1e-12 * n if atol is None else atol
    A = randomize(P, rng)  # :242
    vals, Q = _eigen(A)  # :246
    ed = make_eigen_decomposition(vals, Q, atol)  # :254
    A = randomize(P, rng)  # :259
    K = isomorphism_partition(ed, A, atol)  # :262
    if not is_consistent(K):  # :264-270
        raise NumericalInconsistency("the K-partition seems inconsistent with eigenspaces")
    return ed, K


def irreducible_decomposition(ed: EigenDecomposition, K: IntDisjointSets, P: Partition, rng=None):
    """src/eigen_decomposition.jl:295-348."""
    rng = np.random.default_rng(1) if rng is None else rng
    kp = [K.find_root(i) for i in range(len(K))]  # :301
    roots = list(dict.fromkeys(kp))  # unique, first-occurrence order :303
    A = randomize(P, rng)  # :306
    P_hat = []
    Q = ed.vectors
    for i in roots:
        Ki = [j for j, r in enumerate(kp) if r == i]
        assert Ki[0] == i  # :310
        if len(Ki) == 1:
            P_hat.append(Q[:, ed.ptrs[i]:ed.ptrs[i] + 1].copy())  # :312
            continue
        QKi = np.hstack([Q[:, ed.rng(j)] for j in Ki])  # :316
        m = ed.dim(i)
        Pi = np.zeros((QKi.shape[1], QKi.shape[1]))
        Pi[:m, :m] = np.eye(m)  # :326
        Qi = Q[:, ed.rng(i)]
        for nn, j in enumerate(Ki[1:], start=1):
            Qj = Q[:, ed.rng(j)]
            blk = (Qi.T @ A @ Qj).T  # :333, block(A,Ei,Ej)'
            blk = blk / np.linalg.norm(blk[0, :])  # :335
            Pi[nn * m:(nn + 1) * m, nn * m:(nn + 1) * m] = blk
        if m == 1:
            P_hat.append(QKi @ Pi)  # :339
        else:
            P_hat.append(QKi @ Pi[:, 0:m * len(Ki):m])  # :342-343
    return P_hat


def check_block_sizes(Q_hat, P: Partition):
    """src/diagonalize.jl:1-11 (real case)."""
    sizes = [q.shape[1] for q in Q_hat]
    final_dim = sum(s * (s + 1) // 2 for s in sizes)
    if final_dim != P.nparts:
        raise DimensionMismatch(f"final_dim={final_dim} block_sizes={sizes} expected={P.nparts}")


def diagonalize(P: Partition, atol=None, rng=None):
    """``diagonalize(Float64, P)`` -- src/diagonalize.jl:25-40."""
    rng = np.random.default_rng(0) if rng is None else rng
    n = P.matrix.shape[0]
    atol = 1e-12 * n if atol is None else atol
    ed, K = eigen_decomposition(P, atol, rng)
    Q_hat = irreducible_decomposition(ed, K, P, rng)
    return [clamptol(q, atol) for q in Q_hat]  # :39


def basis_image(Q_hat, P: Partition, atol=None):
    """src/diagonalize.jl:64-89: blks[i][k] = Q_k' 1[P==i] Q_k, clamped at 1e-12*n."""
    n = P.matrix.shape[0]
    atol = 1e-12 * n if atol is None else atol
    out = []
    for i in range(1, P.nparts + 1):
        Mi = (P.matrix == i).astype(np.float64)
        out.append([clamptol(q.T @ (Mi @ q), atol) for q in Q_hat])
    return out


def basis_image_fast(Q_hat, P: Partition, atol=None):
    """Same values as ``basis_image`` via one pass over the entries (used by the
    CPU baseline where dim(P) * n^2 dense products would be wasteful)."""
    n = P.matrix.shape[0]
    atol = 1e-12 * n if atol is None else atol
    out = [[None] * len(Q_hat) for _ in range(P.nparts)]
    L = P.matrix
    rows = np.arange(n)
    for k, q in enumerate(Q_hat):
        s = q.shape[1]
        # T[i, r, :] = (1[P==i] q)[r, :] = sum_c [L[r,c]==i] q[c,:]
        T = np.zeros((P.nparts + 1, n, s))
        for c in range(n):
            T[L[:, c], rows, :] += q[c, :]  # (label,row) pairs are distinct within a column
        for i in range(1, P.nparts + 1):
            out[i - 1][k] = clamptol(q.T @ T[i], atol)
    return out


def block_diagonalize(P: Partition, epsilon=RTOL_DEFAULT, rng=None):
    """``blockDiagonalize(Float64, P)`` -- src/compat.jl:46-68.
    Returns (blkSizes, blks)."""
    rng = np.random.default_rng(0) if rng is None else rng
    Pc = Partition(P.nparts, P.matrix.copy())
    Q_hat = diagonalize(Pc, atol=epsilon, rng=rng)  # :53
    check_block_sizes(Q_hat, P)  # :60
    blks = basis_image(Q_hat, P)  # :63
    return [q.shape[1] for q in Q_hat], blks, Q_hat


# --------------------------------------------------------------------------
# complex path: blockDiagonalize(P; complex=true) -- src/compat.jl:26-32,46-68 with T = ComplexF64,
# src/diagonalize.jl:13-28.  Restated literally: a generic element with COMPLEX coefficients
# (rand(ComplexF64, dim), src/abstract_part.jl:108), the general eigen() of LinearAlgebra
# (eigenvalues sorted by (real, imag), Julia's default eigsortby), adjoints where the source
# writes '.  Pinned by test/runtests.jl:43-57.
# --------------------------------------------------------------------------
def randomize_complex(P: Partition, rng) -> np.ndarray:
    vals = rng.random(P.nparts) + 1j * rng.random(P.nparts)
    table = np.concatenate([[0.0 + 0.0j], vals])
    return table[P.matrix]


def _eigen_complex(A):
    vals, Q = sla.eig(A)
    order = np.lexsort((vals.imag, vals.real))  # sortby = λ -> (real(λ), imag(λ))
    return vals[order], Q[:, order]


def eigen_decomposition_complex(P: Partition, atol, rng):
    """src/eigen_decomposition.jl:236-273 with T = ComplexF64."""
    A = randomize_complex(P, rng)
    vals, Q = _eigen_complex(A)
    ed = make_eigen_decomposition(vals, Q, atol)
    A = randomize_complex(P, rng)
    QAQ = Q.conj().T @ A @ Q
    norms = block_norms_inf(QAQ, ed)
    thr = otsu_threshold(norms, atol)
    ne = len(ed)
    K = IntDisjointSets(ne)
    for i in range(ne):
        for j in range(i + 1, ne):
            if norms[i, j] >= thr:
                K.union(i, j)
    if not is_consistent(K):
        raise NumericalInconsistency("the K-partition seems inconsistent with eigenspaces")
    return ed, K


def irreducible_decomposition_complex(ed, K, P, rng):
    """src/eigen_decomposition.jl:295-348 with complex matrices (' = adjoint)."""
    kp = [K.find_root(i) for i in range(len(K))]
    roots = list(dict.fromkeys(kp))
    A = randomize_complex(P, rng)
    Q = ed.vectors
    P_hat = []
    for i in roots:
        Ki = [j for j, r in enumerate(kp) if r == i]
        assert Ki[0] == i
        if len(Ki) == 1:
            P_hat.append(Q[:, ed.ptrs[i]:ed.ptrs[i] + 1].copy())
            continue
        QKi = np.hstack([Q[:, ed.rng(j)] for j in Ki])
        m = ed.dim(i)
        Pi = np.zeros((QKi.shape[1], QKi.shape[1]), dtype=complex)
        Pi[:m, :m] = np.eye(m)
        Qi = Q[:, ed.rng(i)]
        for nn, j in enumerate(Ki[1:], start=1):
            Qj = Q[:, ed.rng(j)]
            blk = (Qi.conj().T @ A @ Qj).conj().T  # block(A, Ei, Ej)'
            blk = blk / np.linalg.norm(blk[0, :])
            Pi[nn * m:(nn + 1) * m, nn * m:(nn + 1) * m] = blk
        P_hat.append(QKi @ Pi if m == 1 else QKi @ Pi[:, 0:m * len(Ki):m])
    return P_hat


def block_diagonalize_complex(P: Partition, epsilon=RTOL_DEFAULT, rng=None):
    """``blockDiagonalize(ComplexF64, P)`` -- src/compat.jl:46-68.  Returns
    (blkSizes, blks, Q_hat, desymmetrized partition)."""
    rng = np.random.default_rng(0) if rng is None else rng
    Pd = desymmetrize(Partition(P.nparts, P.matrix.copy()), rng=rng)  # src/diagonalize.jl:26-28
    ed, K = eigen_decomposition_complex(Pd, epsilon, rng)
    Q_hat = [np.where(np.abs(q) < epsilon, 0, q) for q in irreducible_decomposition_complex(ed, K, Pd, rng)]
    Pd2 = desymmetrize(Partition(P.nparts, P.matrix.copy()), atol=epsilon, rng=rng)  # src/compat.jl:54-57
    sizes = [q.shape[1] for q in Q_hat]
    if sum(s * s for s in sizes) != Pd2.nparts:  # src/diagonalize.jl:13-23
        raise DimensionMismatch(f"final_dim={sum(s * s for s in sizes)} block_sizes={sizes} expected={Pd2.nparts}")
    n = P.matrix.shape[0]
    blks = []
    for i in range(1, Pd2.nparts + 1):
        Mi = (Pd2.matrix == i).astype(np.float64)
        row = []
        for q in Q_hat:
            b = q.conj().T @ (Mi @ q)
            row.append(np.where(np.abs(b) < 1e-12 * n, 0, b))
        blks.append(row)
    return sizes, blks, Q_hat, Pd2


def spectrum_invariant_complex(P: Partition, blks, x, tol=1e-7):
    """Complex analogue of ``spectrum_invariant``: distinct eigenvalues (complex) of
    sum_i x_i 1[P==i] (P need not be symmetric) vs the union of the block spectra; both sorted
    by (real, imag)."""
    A = fill(P, x)
    full = np.linalg.eigvals(A)
    vals = np.concatenate([np.linalg.eigvals(sum(x[i] * blks[i][k] for i in range(P.nparts))) for k in range(len(blks[0]))])

    def distinct(v):
        scale = max(1.0, np.abs(v).max())
        keep = []
        for z in v[np.lexsort((np.round(v.imag, 9), np.round(v.real, 9)))]:
            if all(abs(z - k) > tol * scale for k in keep):
                keep.append(z)
        keep = np.array(keep)
        return keep[np.lexsort((np.round(keep.imag, 6), np.round(keep.real, 6)))]

    return distinct(full), distinct(vals)


# --------------------------------------------------------------------------
# spectrum invariant: defines "block eigenvalues within 1e-6 rel"
# --------------------------------------------------------------------------
def spectrum_invariant(P: Partition, blks, x, tol=1e-7):
    """Distinct eigenvalues of sum_i x_i 1[P==i] vs the union of the block spectra.

    Returns (full_distinct, block_union_distinct) as sorted arrays.
    """
    A = fill(P, x)
    full = np.linalg.eigvalsh((A + A.T) / 2)
    nb = len(blks[0])
    blk_vals = []
    for k in range(nb):
        B = sum(x[i] * blks[i][k] for i in range(P.nparts))
        blk_vals.append(np.linalg.eigvalsh((B + B.T) / 2))
    blk_vals = np.sort(np.concatenate(blk_vals))

    def distinct(v):
        v = np.sort(v)
        scale = max(1.0, np.abs(v).max())
        keep = [v[0]]
        for t in v[1:]:
            if abs(t - keep[-1]) > tol * scale:
                keep.append(t)
        return np.array(keep)

    return distinct(full), distinct(blk_vals)
