"""Host-side mirror of the reference's interface for the Jordan-reduction path.

Same names, argument meaning and error behaviour as the Julia package
(``Partition``, ``dim``, ``refine``, ``fill``, ``randomize``, ``admissible_subspace``,
``diagonalize``, ``blockDiagonalize``); all computation happens in libsdpsr_hip.so
through the C ABI.  Arrays may be NumPy (host, copied by the library) or torch CUDA
tensors (device-resident, used in place).

Reference lines mirrored: src/partitions.jl:6-17,24-75,109-190; src/compat.jl:26-68;
src/diagonalize.jl:25-40,64-89; src/abstract_part.jl:7-16,107-110.
"""
from __future__ import annotations

import ctypes as C
import math
from collections import namedtuple

import numpy as np

from . import _lib as L

RTOL_DEFAULT = math.sqrt(np.finfo(np.float64).eps)  # Base.rtoldefault(Float64)


class SdpsrError(RuntimeError):
    status = None


class InvalidDecompositionField(SdpsrError):
    """src/eigen_decomposition.jl:140-150"""


class NumericalInconsistency(SdpsrError):
    """src/eigen_decomposition.jl:152-161"""


class DimensionMismatch(SdpsrError):
    """src/diagonalize.jl:4-9"""


class LabelOverflow(SdpsrError):
    """InexactError of src/partitions.jl:63"""


class NotConverged(SdpsrError):
    pass


_EXC = {1: InvalidDecompositionField, 2: NumericalInconsistency, 3: DimensionMismatch,
        4: LabelOverflow, 9: NotConverged}


def _is_torch(x):
    return hasattr(x, "data_ptr") and hasattr(x, "is_cuda")


def _ptr(x):
    if x is None:
        return None
    if _is_torch(x):
        return C.c_void_p(x.data_ptr())
    return C.c_void_p(x.ctypes.data)


class Context:
    """One device context = one HIP stream + workspace (sdpsr_create)."""

    def __init__(self, device=0, seed=0, square_mode=L.SQUARE_AUTO, channels=0, max_iters=0,
                 confirm_rounds=0, eig_driver=0, flags=0, round_mode="nearest", basis_image_kernel="auto",
                 refine_path="auto", label_bits=0, insert_wgs_per_cu=0, square_kernel=0):
        """``flags``: OR of ``_lib.FLAG_*``; ``round_mode``: "nearest" (default) or "trunc" (the
        reference's ``unsafe_round``, src/utils.jl:49-53); ``label_bits``: 0, or the width of the
        reference's label type ``T`` in ``Partition{T}`` to get ``LabelOverflow`` where it would throw."""
        self._lib = L.load_library()
        o = L.Opts()
        o.struct_size = C.sizeof(L.Opts)
        o.square_mode = int(square_mode)
        o.channels = int(channels)
        o.max_iters = int(max_iters)
        o.confirm_rounds = int(confirm_rounds)
        o.eig_driver = int(eig_driver)
        o.flags = int(flags)
        o.round_mode = {"nearest": L.ROUND_NEAREST, "trunc": L.ROUND_TRUNC}[round_mode] if isinstance(round_mode, str) else int(round_mode)
        o.basis_image_kernel = L.BASIS_IMAGE_KERNELS[basis_image_kernel] if isinstance(basis_image_kernel, str) else int(basis_image_kernel)
        o.refine_path = L.REFINE_PATHS[refine_path] if isinstance(refine_path, str) else int(refine_path)
        o.label_bits = int(label_bits)
        o.insert_wgs_per_cu = int(insert_wgs_per_cu)
        o.square_kernel = int(square_kernel)
        h = C.c_void_p()
        st = self._lib.sdpsr_create(int(device), C.c_uint64(seed & (2 ** 64 - 1)), C.byref(o), C.byref(h))
        if st != 0:
            raise SdpsrError(f"sdpsr_create failed: {L.STATUS.get(st, st)} (is a GPU visible?)")
        self._h = h
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            self._lib.sdpsr_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, st):
        if st == 0:
            return
        msg = self._lib.sdpsr_last_error(self._h).decode()
        exc = _EXC.get(st, SdpsrError)(msg or L.STATUS.get(st, str(st)))
        exc.status = st
        raise exc

    def set_seed(self, seed):
        self.check(self._lib.sdpsr_set_seed(self._h, C.c_uint64(seed & (2 ** 64 - 1))))

    def set_stream(self, stream_ptr):
        self.check(self._lib.sdpsr_set_stream(self._h, C.c_void_p(stream_ptr)))

    def synchronize(self):
        self.check(self._lib.sdpsr_synchronize(self._h))

    def dimension_trajectory(self):
        """dim(S) after the initial refinement and after every iteration of the last
        ``admissible_subspace`` on this context (what the reference logs under ``verbose``,
        src/partitions.jl:150,156,187-188)."""
        cnt = C.c_int32(0)
        self.check(self._lib.sdpsr_dimension_trajectory(self._h, None, 0, C.byref(cnt)))
        dims = np.zeros(max(cnt.value, 1), dtype=np.int64)
        self.check(self._lib.sdpsr_dimension_trajectory(self._h, dims.ctypes.data_as(C.c_void_p), cnt.value, C.byref(cnt)))
        return [int(x) for x in dims[:cnt.value]]

    def transfer_bytes(self):
        """(host -> device, device -> host) bytes this context has moved since its creation (``sdpsr_transfer_bytes``)."""
        a, b = C.c_uint64(0), C.c_uint64(0)
        self.check(self._lib.sdpsr_transfer_bytes(self._h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def wait_for(self, *arrays):
        """Order ctx's stream behind torch's current stream when any argument is a CUDA tensor
        (it may have been produced by a kernel that is still running): sdpsr_wait_stream."""
        for a in arrays:
            if _is_torch(a) and a.is_cuda:
                import torch
                sp = torch.cuda.current_stream(a.device).cuda_stream
                self.check(self._lib.sdpsr_wait_stream(self._h, C.c_void_p(sp)))
                return


_default_ctx = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context()
    return _default_ctx


def _ctx(ctx):
    return ctx if ctx is not None else default_context()


def _f(a, dtype):
    """Column-major flat host view of a matrix (the only order the reference scans)."""
    a = np.asarray(a, dtype=dtype)
    return np.ascontiguousarray(a.ravel(order="F"))


class Partition:
    """``Partition`` (src/partitions.jl:6-17): ``matrix`` holds labels 0..nparts."""

    def __init__(self, nparts, matrix):
        self.nparts = int(nparts)
        self.matrix = matrix

    # --- constructors: Partition(M) float ctor (:24-35) / integer ctor (:37-42) ---
    @classmethod
    def from_matrix(cls, M, ctx=None):
        ctx = _ctx(ctx)
        M = np.asarray(M)
        shape = M.shape
        n = C.c_int64(0)
        out = np.empty(M.size, dtype=np.uint32)
        if np.issubdtype(M.dtype, np.floating):
            flat = _f(M, np.float64)
            ctx.check(ctx._lib.sdpsr_partition_from_f64(ctx._h, flat.size, _ptr(flat), _ptr(out), C.byref(n), L.MEM_HOST))
        else:
            if M.size and M.min() < 0:  # @assert 0 <= first(M_vals), src/partitions.jl:46
                raise ValueError("labels must be non-negative")
            if M.size and M.max() >= 2 ** 32:
                flat = _f(M, np.uint64)
                ctx.check(ctx._lib.sdpsr_partition_from_u64(ctx._h, flat.size, _ptr(flat), _ptr(out), C.byref(n), L.MEM_HOST))
            else:
                flat = _f(M, np.uint32)
                ctx.check(ctx._lib.sdpsr_partition_from_u32(ctx._h, flat.size, _ptr(flat), _ptr(out), C.byref(n), L.MEM_HOST))
        return cls(n.value, out.reshape(shape, order="F"))

    def __eq__(self, other):  # :16-17
        return self.nparts == other.nparts and np.array_equal(np.asarray(self.matrix), np.asarray(other.matrix))

    @property
    def shape(self):
        return tuple(self.matrix.shape)

    def size(self, i=None):
        return self.shape if i is None else self.shape[i]

    def __repr__(self):
        return f"Partition(dim={self.nparts}, size={self.shape})"


def dim(P):
    return P.nparts


def refine(P1, P2, ctx=None):
    """``refine!(P1, P2)`` (src/partitions.jl:62-66); P1 is updated and returned."""
    ctx = _ctx(ctx)
    assert P1.shape == P2.shape
    a = _f(P1.matrix, np.uint32).copy()
    b = _f(P2.matrix, np.uint32)
    d1 = C.c_int64(P1.nparts)
    ctx.check(ctx._lib.sdpsr_refine(ctx._h, a.size, _ptr(a), C.byref(d1), _ptr(b), P2.nparts, L.MEM_HOST))
    P1.matrix = a.reshape(P1.shape, order="F")
    P1.nparts = d1.value
    return P1


def partition_checksum(P, ctx=None):
    """128-bit checksum of the canonical label matrix: a probabilistic ``==`` of two partitions
    (src/partitions.jl:16-17) that avoids moving n^2 labels between GPUs.
    ``P`` is a Partition or a flat torch/NumPy array of column-major labels."""
    ctx = _ctx(ctx)
    if isinstance(P, Partition):
        lab, mem = _labels_arg(P)
    elif _is_torch(P):
        lab, mem = P.contiguous().view(-1), (L.MEM_DEVICE if P.is_cuda else L.MEM_HOST)
    else:
        lab, mem = np.ascontiguousarray(P, dtype=np.uint32).ravel(), L.MEM_HOST
    n_entries = lab.numel() if _is_torch(lab) else lab.size
    ctx.wait_for(lab)
    out = (C.c_uint64 * 2)()
    ctx.check(ctx._lib.sdpsr_partition_checksum(ctx._h, n_entries, _ptr(lab), C.cast(out, C.c_void_p), mem))
    return int(out[0]), int(out[1])


def relabel_keys(keys, ctx=None):
    """Canonical relabel (first-occurrence order, 0 stays 0) of a flat torch CUDA int64 tensor of
    arbitrary 64-bit keys, on the device: ``sdpsr_partition_from_u64``.  Returns (labels int32 CUDA
    tensor, nparts) -- the ``relabel`` callback of ``parallel.agree_partition`` on a GPU."""
    import torch
    ctx = _ctx(ctx)
    keys = keys.contiguous().view(-1)
    assert keys.is_cuda and keys.dtype == torch.int64
    out = torch.empty(keys.numel(), dtype=torch.int32, device=keys.device)
    n = C.c_int64(0)
    ctx.wait_for(keys)
    ctx.check(ctx._lib.sdpsr_partition_from_u64(ctx._h, keys.numel(), _ptr(keys), _ptr(out), C.byref(n), L.MEM_DEVICE))
    return out, n.value


def fill(P, values, ctx=None):
    """``fill!(M, P; values)`` (src/partitions.jl:68-75)."""
    ctx = _ctx(ctx)
    values = np.ascontiguousarray(values, dtype=np.float64)
    if len(values) != P.nparts:  # @assert length(values) == dim(P), :69
        raise ValueError("length(values) != dim(P)")
    lab = _f(P.matrix, np.uint32)
    out = np.empty(lab.size, dtype=np.float64)
    ctx.check(ctx._lib.sdpsr_fill(ctx._h, lab.size, _ptr(lab), _ptr(values), P.nparts, _ptr(out), L.MEM_HOST))
    return out.reshape(P.shape, order="F")


def randomize(P, ctx=None):
    """``randomize(P)`` (src/abstract_part.jl:97-110)."""
    ctx = _ctx(ctx)
    lab = _f(P.matrix, np.uint32)
    out = np.empty(lab.size, dtype=np.float64)
    ctx.check(ctx._lib.sdpsr_randomize(ctx._h, lab.size, _ptr(lab), _ptr(out), L.MEM_HOST))
    return out.reshape(P.shape, order="F")


# ---------------------------------------------------------------------------
# admissible_subspace
# ---------------------------------------------------------------------------
def _dense(A):
    return np.asarray(A.todense()) if hasattr(A, "todense") else np.asarray(A)


def clamp_round_host(a, atol):
    """_clamp_round! (src/utils.jl:34-53), nearest rounding; host copy used only by the
    setup stage below (O(n^2), once)."""
    sig = math.floor(-math.log10(atol))
    scale = float(10 ** sig)
    x, e = np.frexp(a)
    out = np.ldexp(np.rint(scale * x) / scale, e)
    return np.where(np.abs(a) < atol, 0.0, out)


def admissible_setup(C_, A, b, atol=RTOL_DEFAULT):
    """Setup stage of ``admissible_subspace`` (src/partitions.jl:117-142), on the host:
    qr(A') -> orthonormal basis U of rowspace(A); C_L; min-norm x0 projected.
    Returns (n, CL, X0L, U) with CL/X0L flat column-major, U (n^2 x r) Fortran-ordered."""
    import scipy.linalg as sla
    c = np.asarray(C_.todense()).reshape(-1) if hasattr(C_, "todense") else np.asarray(C_, dtype=np.float64).reshape(-1)
    n = math.isqrt(len(c))
    if n * n != len(c):  # @assert n^2 == length(C), :118
        raise ValueError("length(C) is not a perfect square")
    Ad = np.asarray(_dense(A), dtype=np.float64)
    if Ad.shape[0] > 0:
        Q, R, _piv = sla.qr(Ad.T, mode="economic", pivoting=True)
        dg = np.abs(np.diag(R))
        r = int(np.sum(dg > 1e-12 * dg.max())) if dg.size and dg.max() > 0 else 0
        U = np.asfortranarray(Q[:, :r])
        # min-norm solution of A x = b (Krylov.craig, :137) from the factorisation at hand:
        # A[piv, :] = R' Q'  =>  x0 = Q_r y with R_rr' y = b[piv][:r]  (a LAPACK gelsd call on an
        # m x n^2 matrix with n^2 = 16.7M has been seen to crash inside the library)
        bb = np.asarray(b, dtype=np.float64)[_piv]
        y = sla.solve_triangular(R[:r, :r], bb[:r], trans="T", lower=False) if r else np.zeros(0)
        x0 = U @ y
    else:
        U = np.zeros((n * n, 0), order="F")
        x0 = np.zeros(n * n)

    def proj(v):
        return U @ (U.T @ v)

    def symm(v):
        M = v.reshape(n, n, order="F")
        return ((M + M.T) / 2).ravel(order="F")

    CL = symm(clamp_round_host(c - proj(c), atol))
    X0L = clamp_round_host(proj(symm(x0)), atol)
    CL, X0L = np.ascontiguousarray(CL), np.ascontiguousarray(X0L)
    m1, m2 = CL.reshape(n, n, order="F"), X0L.reshape(n, n, order="F")
    return Setup((n, CL, X0L, U), basis_is_symmetric(n, U), bool(np.array_equal(m1, m1.T) and np.array_equal(m2, m2.T)))


class Setup(tuple):
    """(n, CL, X0L, U) of ``admissible_setup`` plus what the host knows about symmetry:
    ``basis_symmetric`` (every column of U is a symmetric n x n matrix) and ``inputs_symmetric``
    (CL and X0L are exactly symmetric, as the reference's setup makes them).  ``hint`` is the bit
    mask for ``sdpsr_hint_symmetric_basis``."""

    def __new__(cls, items, basis_symmetric=False, inputs_symmetric=False):
        t = super().__new__(cls, items)
        t.basis_symmetric = bool(basis_symmetric)
        t.inputs_symmetric = bool(inputs_symmetric)
        t.hint = (1 if basis_symmetric else 0) | (2 if inputs_symmetric else 0)
        return t


def basis_is_symmetric(n, U, tol=1e-12):
    """True if every column of U (n^2 x r), reshaped column-major to n x n, is symmetric."""
    U = np.asarray(U)
    for k in range(U.shape[1]):
        M = U[:, k].reshape(n, n, order="F")
        if not np.allclose(M, M.T, rtol=0.0, atol=tol):
            return False
    return True


def _admissible_subspace_device_setup(C_, A, b, atol, ctx, verbose):
    """Whole ``admissible_subspace`` through ``sdpsr_admissible_subspace_dense``: the setup stage
    (qr(A'), C_L, min-norm x0; src/partitions.jl:117-142) runs on the device as well."""
    c = np.asarray(C_.todense()).reshape(-1) if hasattr(C_, "todense") else np.asarray(C_, dtype=np.float64).reshape(-1)
    c = np.ascontiguousarray(c, dtype=np.float64)
    n = math.isqrt(len(c))
    if n * n != len(c):
        raise ValueError("length(C) is not a perfect square")
    Ad = np.asfortranarray(_dense(A), dtype=np.float64)
    m = Ad.shape[0]
    bb = np.ascontiguousarray(b, dtype=np.float64)
    P = np.empty(n * n, dtype=np.uint32)
    d = C.c_int64(0)
    it = C.c_int32(0)
    ms = (C.c_double * L.T_COUNT)()
    ctx.check(ctx._lib.sdpsr_admissible_subspace_dense(ctx._h, n, m, _ptr(c), _ptr(Ad), _ptr(bb), float(atol), _ptr(P),
                                                       C.byref(d), C.byref(it), C.cast(ms, C.c_void_p), L.MEM_HOST))
    if verbose:
        print(f"[sdpsr] admissible subspace (device setup): dim {d.value} after {it.value} iterations, loop {ms[L.T_TOTAL]:.3f} ms")
    out = Partition(d.value, P.reshape(n, n, order="F"))
    out.iterations = it.value
    out.phase_ms = list(ms)
    out.dims = ctx.dimension_trajectory()
    return out


def admissible_subspace(C_, A, b, atol=RTOL_DEFAULT, ctx=None, verbose=False, setup=None,
                        return_info=False, host_setup=False):
    """``admissible_subspace(C, A, b; verbose, atol)`` (src/partitions.jl:77-190).

    By default the setup stage runs on the device too (dense copy of ``A``, at most 4 GiB);
    ``host_setup=True`` (or a precomputed ``setup``) uses the NumPy/SciPy setup, which is also the
    path for very large sparse ``A``."""
    ctx = _ctx(ctx)
    if setup is None and not host_setup and A is not None:
        m_rows = A.shape[0]
        if m_rows * int(np.prod(np.shape(C_))) * 8 <= (4 << 30):
            return _admissible_subspace_device_setup(C_, A, b, atol, ctx, verbose)
    if setup is None:
        setup = admissible_setup(C_, A, b, atol)
    n, CL, X0L, U = setup
    if getattr(setup, "hint", 0):
        ctx._lib.sdpsr_hint_symmetric_basis(ctx._h, int(setup.hint))  # applies to the call below only
    on_dev = _is_torch(CL)
    r = U.shape[1]
    if on_dev:
        import torch
        P = torch.empty(n * n, dtype=torch.int32, device=CL.device)  # uint32 bit pattern
    else:
        U = np.asfortranarray(U, dtype=np.float64)
        P = np.empty(n * n, dtype=np.uint32)
    d = C.c_int64(0)
    it = C.c_int32(0)
    ms = (C.c_double * L.T_COUNT)()
    ctx.wait_for(CL, X0L, U)
    st = ctx._lib.sdpsr_admissible_subspace(
        ctx._h, n, _ptr(CL), _ptr(X0L), _ptr(U) if r > 0 else None, r, float(atol), _ptr(P),
        C.byref(d), C.byref(it), C.cast(ms, C.c_void_p), L.MEM_DEVICE if on_dev else L.MEM_HOST)
    ctx.check(st)
    out_dims = ctx.dimension_trajectory()
    if verbose:
        print(f"[sdpsr] Starting the reduction. Dimensions: maximal = {(n * n + n) // 2}, initial = {out_dims[0] if out_dims else '?'}")
        for k, dk in enumerate(out_dims[:-1]):
            print(f"[sdpsr] Iteration {k + 1}, Current dimension: {dk}")
        print(f"[sdpsr] admissible subspace: dim {d.value} after {it.value} iterations, "
              f"{ms[L.T_TOTAL]:.3f} ms (project {ms[L.T_PROJECT]:.3f}, square {ms[L.T_SQUARE]:.3f}, "
              f"refine {ms[L.T_REFINE]:.3f})")
    mat = P.view(n, n).t() if on_dev else P.reshape(n, n, order="F")
    out = Partition(d.value, mat)
    out.iterations = it.value
    out.phase_ms = list(ms)
    out.dims = out_dims
    return out


class Problem:
    """The loop's inputs (C_L, X0_L, U of ``admissible_setup``) made device-resident ONCE (``sdpsr_problem_create``); any
    number of ``reduce`` / ``reduce_batch`` calls then read them there.  The reference's seam hands host arrays to
    ``admissible_subspace`` on every call (src/partitions.jl:109-116); restarting the randomized reduction of one problem
    ("try again", src/eigen_decomposition.jl:264-270) through it pays the upload -- 3 x 134 MB at N = 4096 -- every time."""

    def __init__(self, C_=None, A=None, b=None, atol=RTOL_DEFAULT, ctx=None, setup=None):
        self.ctx = _ctx(ctx)
        setup = setup if setup is not None else admissible_setup(C_, A, b, atol)
        self.n, CL, X0L, U = setup
        self.r = U.shape[1]
        self.atol = atol
        Uf = np.asfortranarray(U) if self.r else None
        h = C.c_void_p()
        self.ctx.check(self.ctx._lib.sdpsr_problem_create(self.ctx._h, self.n, _ptr(CL), _ptr(X0L), _ptr(Uf) if self.r else None, self.r,
                                                          int(getattr(setup, "hint", 0)), L.MEM_HOST, C.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self.ctx._lib.sdpsr_problem_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def reduce(self, seed=None, epsilon=RTOL_DEFAULT, atol=None):
        """One reduction (``admissible_subspace`` + ``blockDiagonalize``): the dict of a one-restart batch; raises the
        reference's exception for a status that is not 0."""
        res = self.reduce_batch(1, seeds=None if seed is None else [seed], epsilon=epsilon, atol=atol)[0]
        if res["status"]:
            self.ctx.check(res["status"])
        return res

    def reduce_batch(self, restarts=2, seeds=None, epsilon=RTOL_DEFAULT, atol=None):
        """``restarts`` independent random restarts in one call (``sdpsr_problem_reduce_batch``); a list of dicts as
        ``jordan_reduce_batch`` returns them."""
        ctx, lib, n = self.ctx, self.ctx._lib, self.n
        R = int(restarts)
        atol = self.atol if atol is None else atol
        if seeds is None:  # explicit seeds: the sizes call and the images call must run the same restarts
            seeds = [int(x) for x in np.random.SeedSequence().generate_state(R, dtype=np.uint64)]
        if len(seeds) != R:
            raise ValueError(f"{len(seeds)} seeds for {R} restarts")
        sd = (C.c_uint64 * R)(*[int(x) & (2 ** 64 - 1) for x in seeds])

        def call(blk_arrays):
            Ps = [np.zeros(n * n, dtype=np.uint32) for _ in range(R)]
            pP = (C.c_void_p * R)(*[a.ctypes.data for a in Ps])
            dd, it, nb = (C.c_int64 * R)(), (C.c_int32 * R)(), (C.c_int32 * R)()
            ssq, ss = (C.c_int64 * R)(), (C.c_int64 * R)()
            st = (C.c_int32 * R)(*([-1] * R))  # a sentinel no status code has
            pb = caps = None
            if blk_arrays is not None:
                pb = (C.c_void_p * R)(*[a.ctypes.data for a in blk_arrays])
                caps = (C.c_int64 * R)(*[a.size for a in blk_arrays])
            rc = lib.sdpsr_problem_reduce_batch(ctx._h, self._h, R, C.cast(sd, C.c_void_p), atol, epsilon, C.cast(pP, C.c_void_p), dd, it, nb, ssq, ss,
                                                C.cast(pb, C.c_void_p) if pb is not None else None, caps, st, L.MEM_HOST)
            # the call's own status counts unless some restart reports one: a failure before the restarts start (bad
            # arguments, memory) must not read as R clean restarts of dimension 0
            if rc != 0 and all(x in (0, -1) for x in st):
                ctx.check(rc)
            return Ps, dd, it, nb, ssq, ss, st

        Ps, dd, it, nb, ssq, ss, st = call(None)  # sizes first, then the images into buffers of the right size
        bl = [np.zeros(max(1, dd[i] * ssq[i])) for i in range(R)]
        Ps, dd, it, nb, ssq, ss, st = call(bl)
        out = []
        for i in range(R):
            out.append({"status": int(st[i]), "P": Partition(int(dd[i]), Ps[i].reshape(n, n, order="F")), "iterations": int(it[i]),
                        "nblocks": int(nb[i]), "sum_sq": int(ssq[i]), "sum_s": int(ss[i]),
                        "blks": bl[i][:dd[i] * ssq[i]].reshape(dd[i], ssq[i]) if st[i] == 0 else None})
        return out


def jordan_reduce_batch(C_, A, b, restarts=2, seeds=None, atol=RTOL_DEFAULT, epsilon=RTOL_DEFAULT, ctx=None, setup=None):
    """``restarts`` independent random restarts of ``admissible_subspace`` + ``blockDiagonalize`` of one problem in ONE
    call on one host thread: restart i with its own random streams, HIP stream and workspace; while one restart's host
    side waits for a verdict the others' work is submitted.  The reference's answer to the randomized failures of
    ``blockDiagonalize`` is "try again" (src/eigen_decomposition.jl:264-270, src/diagonalize.jl:4-9): here the tries run
    side by side and the caller takes the first whose status is 0.  The problem is uploaded once (``Problem``) and both
    passes -- sizes, then images -- read it on the device; ``seeds=None`` draws explicit seeds, so that the two passes run
    the same restarts.  Returns a list of dicts: status, P (Partition), iterations, nblocks, sum_sq, sum_s, blks
    (d x sum s_k^2 array, class-major; None for a restart whose status is not 0)."""
    with Problem(C_, A, b, atol=atol, ctx=ctx, setup=setup) as prob:
        return prob.reduce_batch(restarts, seeds=seeds, epsilon=epsilon)


def reduce_constraints(P, A, ctx=None):
    """``A * PMat`` with ``PMat = hcat([vec(P.matrix .== i) for i = 1:dim(P)]...)``
    (README.md:57-60, test/sd_problems.jl:32-37); ``A`` dense m x n^2 (or a vector: C' * PMat)."""
    ctx = _ctx(ctx)
    A = np.asarray(_dense(A), dtype=np.float64)
    vec = A.ndim == 1
    A2 = A.reshape(1, -1) if vec else A
    m, ln = A2.shape
    lab = _f(P.matrix, np.uint32)
    assert ln == lab.size
    Af = np.asfortranarray(A2)
    out = np.zeros((m, P.nparts), order="F")
    ctx.check(ctx._lib.sdpsr_reduce_constraints(ctx._h, ln, _ptr(lab), P.nparts, m, _ptr(Af), _ptr(out), L.MEM_HOST))
    return out[0] if vec else out


def desymmetrize(P, ctx=None):
    """``desymmetrize(P)`` (src/partitions.jl:197-223); returns a new Partition."""
    ctx = _ctx(ctx)
    n = P.shape[0]
    lab = _f(P.matrix, np.uint32).copy()
    d = C.c_int64(P.nparts)
    it = C.c_int32(0)
    ctx.check(ctx._lib.sdpsr_desymmetrize(ctx._h, n, _ptr(lab), C.byref(d), C.byref(it), L.MEM_HOST))
    out = Partition(d.value, lab.reshape(P.shape, order="F"))
    out.iterations = it.value
    return out


unSymmetrize = desymmetrize  # src/compat.jl:70


# ---------------------------------------------------------------------------
# blockDiagonalize
# ---------------------------------------------------------------------------
BlockDiagonalization = namedtuple("BlockDiagonalization", ["blkSizes", "blks", "Q_hat", "phase_ms"])
ComplexBlockDiagonalization = namedtuple("ComplexBlockDiagonalization", ["blkSizes", "blks", "Q_hat", "partition"])


def _labels_arg(P):
    m = P.matrix
    if _is_torch(m):
        # stored as the transposed view of a flat column-major buffer (see admissible_subspace)
        flat = m.t().contiguous().view(-1)
        return flat, L.MEM_DEVICE
    return _f(m, np.uint32), L.MEM_HOST


def blockDiagonalize(P, verbose=False, epsilon=RTOL_DEFAULT, complex=False, ctx=None, retries=0):
    """``blockDiagonalize(P, verbose; epsilon, complex)`` (src/compat.jl:26-68), real path.

    ``retries`` > 0 re-runs the randomized decomposition with fresh generic elements when it ends
    in ``NumericalInconsistency`` / ``DimensionMismatch`` -- what the reference's error texts ask
    the caller to do ("try again"); the default 0 is the reference's behaviour."""
    ctx = _ctx(ctx)
    if complex:
        return _block_diagonalize_complex(P, verbose, epsilon, ctx, retries)
    for attempt in range(int(retries)):
        try:
            return blockDiagonalize(P, verbose=verbose, epsilon=epsilon, ctx=ctx, retries=0)
        except (NumericalInconsistency, DimensionMismatch) as e:
            if verbose:
                print(f"[sdpsr] blockDiagonalize attempt {attempt + 1} failed ({type(e).__name__}); retrying")
    n = P.shape[0]
    lab, mem = _labels_arg(P)
    ctx.wait_for(lab)
    nb = C.c_int32(0)
    ssq = C.c_int64(0)
    ss = C.c_int64(0)
    ms1 = (C.c_double * L.T_COUNT)()
    ctx.check(ctx._lib.sdpsr_block_diagonalize(ctx._h, n, _ptr(lab), P.nparts, float(epsilon), C.byref(nb),
                                               C.byref(ssq), C.byref(ss), C.cast(ms1, C.c_void_p), mem))
    sizes = np.zeros(nb.value, dtype=np.int32)
    ctx.check(ctx._lib.sdpsr_block_sizes(ctx._h, _ptr(sizes)))
    blks = np.empty(P.nparts * ssq.value, dtype=np.float64)
    qh = np.empty(n * ss.value, dtype=np.float64)
    ms2 = (C.c_double * L.T_COUNT)()
    ctx.check(ctx._lib.sdpsr_block_images(ctx._h, _ptr(blks), _ptr(qh), C.cast(ms2, C.c_void_p), L.MEM_HOST))
    out = []
    blks = blks.reshape(P.nparts, ssq.value) if ssq.value else blks.reshape(P.nparts, 0)
    for i in range(P.nparts):
        row, off = [], 0
        for s in sizes:
            row.append(blks[i, off:off + s * s].reshape(s, s, order="F"))
            off += s * s
        out.append(row)
    Q, off = [], 0
    qh = qh.reshape(n, ss.value, order="F")
    for s in sizes:
        Q.append(qh[:, off:off + s])
        off += s
    ms = [a + b for a, b in zip(ms1, ms2)]
    if verbose:
        print(f"[sdpsr] blockDiagonalize: blocks {list(sizes)}; eigen {ms[L.T_EIGEN]:.3f} ms, iso {ms[L.T_ISO]:.3f}, "
              f"irreducible {ms[L.T_IRRED]:.3f}, image {ms[L.T_IMAGE]:.3f}")
    return BlockDiagonalization([int(s) for s in sizes], out, Q, ms)


def _block_diagonalize_complex(P, verbose, epsilon, ctx, retries):
    """``blockDiagonalize(ComplexF64, P)`` (src/compat.jl:46-68, src/diagonalize.jl:13-28): the
    block images are indexed by the classes of the DESYMMETRIZED partition (src/compat.jl:54-57),
    returned as ``.partition`` of the result."""
    for attempt in range(int(retries)):
        try:
            return _block_diagonalize_complex(P, verbose, epsilon, ctx, 0)
        except (NumericalInconsistency, DimensionMismatch):
            pass
    n = P.shape[0]
    lab = _f(np.asarray(P.matrix.cpu() if _is_torch(P.matrix) else P.matrix), np.uint32)
    Pd = np.empty(n * n, dtype=np.uint32)
    dd = C.c_int64(0)
    nb = C.c_int32(0)
    ssq = C.c_int64(0)
    ss = C.c_int64(0)
    ctx.check(ctx._lib.sdpsr_block_diagonalize_complex(ctx._h, n, _ptr(lab), P.nparts, float(epsilon), _ptr(Pd), C.byref(dd),
                                                       C.byref(nb), C.byref(ssq), C.byref(ss), L.MEM_HOST))
    sizes = np.zeros(nb.value, dtype=np.int32)
    ctx.check(ctx._lib.sdpsr_block_sizes_complex(ctx._h, _ptr(sizes)))
    d = dd.value
    blks = np.empty(2 * d * ssq.value, dtype=np.float64)
    qh = np.empty(2 * n * ss.value, dtype=np.float64)
    ctx.check(ctx._lib.sdpsr_block_images_complex(ctx._h, _ptr(blks), _ptr(qh), L.MEM_HOST))
    blks = blks.view(np.complex128).reshape(d, ssq.value)
    qh = qh.view(np.complex128).reshape(n, ss.value, order="F")
    out = []
    for i in range(d):
        row, off = [], 0
        for s in sizes:
            row.append(blks[i, off:off + s * s].reshape(s, s, order="F"))
            off += s * s
        out.append(row)
    Q, off = [], 0
    for s in sizes:
        Q.append(qh[:, off:off + s])
        off += s
    res = ComplexBlockDiagonalization([int(s) for s in sizes], out, Q, Partition(d, Pd.reshape(n, n, order="F")))
    if verbose:
        print(f"[sdpsr] blockDiagonalize over ComplexF64: blocks {res.blkSizes}, dim(desymmetrized P) = {d}")
    return res


def diagonalize(P, atol=None, ctx=None):
    """``diagonalize(Float64, P; atol)`` (src/diagonalize.jl:25-40): ``Q_hat``, a list of
    n x s_k matrices (one column per merged eigenspace of an isomorphism class)."""
    n = P.shape[0]
    atol = 1e-12 * n if atol is None else atol
    ctx = _ctx(ctx)
    lab, mem = _labels_arg(P)
    ctx.wait_for(lab)
    nb = C.c_int32(0)
    ssq = C.c_int64(0)
    ss = C.c_int64(0)
    st = ctx._lib.sdpsr_block_diagonalize(ctx._h, n, _ptr(lab), P.nparts, float(atol), C.byref(nb),
                                          C.byref(ssq), C.byref(ss), None, mem)
    if st != 3:  # diagonalize itself does not run check_block_sizes (src/compat.jl:60 does)
        ctx.check(st)
    sizes = np.zeros(nb.value, dtype=np.int32)
    ctx.check(ctx._lib.sdpsr_block_sizes(ctx._h, _ptr(sizes)))
    qh = np.empty(n * ss.value, dtype=np.float64)
    ctx.check(ctx._lib.sdpsr_q_hat(ctx._h, _ptr(qh), L.MEM_HOST))
    qh = qh.reshape(n, ss.value, order="F")
    Q, off = [], 0
    for s in sizes:
        Q.append(qh[:, off:off + s])
        off += s
    return Q


def eigen_decomposition(P, atol=None, ctx=None):
    """``eigen_decomposition(P, A; atol)`` (src/eigen_decomposition.jl:236-273): returns
    (number of eigenspaces, number of isomorphism classes) or raises."""
    n = P.shape[0]
    atol = 1e-12 * n if atol is None else atol
    ctx = _ctx(ctx)
    lab, mem = _labels_arg(P)
    ctx.wait_for(lab)
    ne = C.c_int32(0)
    nc = C.c_int32(0)
    ctx.check(ctx._lib.sdpsr_eigen_decomposition(ctx._h, n, _ptr(lab), P.nparts, float(atol), C.byref(ne), C.byref(nc), mem))
    return ne.value, nc.value


def eigen_decomposition_batched(P, count, atol=None, values=None, ctx=None, raise_on_failure=True):
    """``count`` independent runs of ``eigen_decomposition(P, A; atol)`` on one partition
    (test/numerical_issues.jl:85-94 runs 10 000 of them): one workgroup per run on the device.
    ``values``: optional array (count, 2, dim(P)) of class values for the two generic elements of
    every run.  Returns (status, neig, nclasses) int32 arrays; raises the first failure like the
    reference would unless ``raise_on_failure`` is False."""
    n = P.shape[0]
    atol = 1e-12 * n if atol is None else atol
    ctx = _ctx(ctx)
    lab, mem = _labels_arg(P)
    ctx.wait_for(lab)
    vals = None
    if values is not None:
        vals = np.ascontiguousarray(values, dtype=np.float64).reshape(count, 2, P.nparts)
        if mem != L.MEM_HOST:
            raise ValueError("explicit values need host-resident labels")
    st = np.zeros(count, dtype=np.int32)
    ne = np.zeros(count, dtype=np.int32)
    nc = np.zeros(count, dtype=np.int32)
    rc = ctx._lib.sdpsr_eigen_decomposition_batched(ctx._h, n, _ptr(lab), P.nparts, float(atol), int(count), _ptr(vals),
                                                    _ptr(st), _ptr(ne), _ptr(nc), mem)
    if rc != 0 and (raise_on_failure or rc not in (2, 9)):
        ctx.check(rc)
    return st, ne, nc
