# SDPSymmetryReductionHIP.jl -- Julia-side binding of libsdpsr_hip.so (include/sdpsr.h).
#
# WRITTEN BLIND: there is no Julia toolchain in the build image, so this file has never been
# executed.  It shows the binding a maintainer of SDPSymmetryReduction.jl would add: a new
# `AbstractPartition` backend whose whole-function specialisations are one `ccall` each
# (the generic functions call `mul!`/`eigen` on plain matrices, so the backend specialises
# `admissible_subspace(::Type{HIPPartition}, ...)`, `diagonalize(::Type{Float64}, ::HIPPartition)`
# and `basis_image(Q, ::HIPPartition)`, exactly where test/partitions_set.jl plugs in its
# `Partition{BitSet}`).
module SDPSymmetryReductionHIP

import SDPSymmetryReduction as SR
using LinearAlgebra, SparseArrays

const libsdpsr = get(ENV, "SDPSR_HIP_LIB", "libsdpsr_hip.so")
const MEM_HOST = Cint(0)

struct StatusError <: Exception
    code::Cint
    msg::String
end

# sdpsr_opts (include/sdpsr.h, ABI 0.3): 16 32-bit words; everything zero = the library's defaults
struct Opts
    struct_size::UInt32
    square_mode::Int32
    channels::Int32
    max_iters::Int32
    confirm_rounds::Int32
    eig_driver::Int32
    flags::UInt32            # SDPSR_FLAG_*
    round_mode::Int32        # 0 nearest (default), 1 trunc = unsafe_round as written (utils.jl:49-53)
    basis_image_kernel::Int32
    refine_path::Int32
    label_bits::Int32        # 8 * sizeof(T) of Partition{T}: InexactError where the reference throws it; 0 = never
    insert_wgs_per_cu::Int32 # measurement knob, 0 = default
    square_kernel::Int32     # 0 = by size, 1 = 128 x 128 tiles, 64 = persistent 256 x 256 launch forced
    reserved::NTuple{3,Int32}
end
Opts(; flags=0, round_mode=0, label_bits=0, square_mode=0, channels=0) =
    Opts(UInt32(64), square_mode, channels, 0, 0, 0, UInt32(flags), round_mode, 0, 0, label_bits, 0, 0, (Int32(0), Int32(0), Int32(0)))

mutable struct Context
    handle::Ptr{Cvoid}
    function Context(; device::Integer=0, seed::Integer=rand(UInt64), opts::Opts=Opts())
        h = Ref{Ptr{Cvoid}}(C_NULL)
        st = ccall((:sdpsr_create, libsdpsr), Cint, (Cint, UInt64, Ref{Opts}, Ref{Ptr{Cvoid}}),
                   device, seed % UInt64, Ref(opts), h)
        st == 0 || throw(StatusError(st, "sdpsr_create"))
        ctx = new(h[])
        finalizer(c -> ccall((:sdpsr_destroy, libsdpsr), Cvoid, (Ptr{Cvoid},), c.handle), ctx)
        return ctx
    end
end

const DEFAULT_CTX = Ref{Union{Nothing,Context}}(nothing)
ctx() = (DEFAULT_CTX[] === nothing && (DEFAULT_CTX[] = Context()); DEFAULT_CTX[])

function check(c::Context, st::Cint)
    st == 0 && return
    msg = unsafe_string(ccall((:sdpsr_last_error, libsdpsr), Cstring, (Ptr{Cvoid},), c.handle))
    st == 1 && throw(SR.InvalidDecompositionField(Float64, ComplexF64))   # eigen_decomposition.jl:140
    st == 2 && throw(SR.NumericalInconsistency("eigen_decomposition", msg)) # :152
    st == 3 && throw(DimensionMismatch(msg))                               # diagonalize.jl:6
    st == 4 && throw(InexactError(:refine!, UInt32, 0))                    # partitions.jl:63
    throw(StatusError(st, msg))
end

# ---- the partition backend (AbstractPartition contract, abstract_part.jl:7-16) ----------
mutable struct HIPPartition <: SR.AbstractPartition
    nparts::Int
    matrix::Matrix{UInt32}
end
SR.dim(p::HIPPartition) = p.nparts
Base.size(p::HIPPartition, args...) = size(p.matrix, args...)
Base.:(==)(p::HIPPartition, q::HIPPartition) = p.nparts == q.nparts && p.matrix == q.matrix

function HIPPartition(M::AbstractMatrix{<:AbstractFloat})               # partitions.jl:24-35
    Md = Matrix{Float64}(M); out = Matrix{UInt32}(undef, size(M)); n = Ref{Int64}(0)
    c = ctx()
    check(c, ccall((:sdpsr_partition_from_f64, libsdpsr), Cint,
                   (Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{UInt32}, Ref{Int64}, Cint),
                   c.handle, length(Md), Md, out, n, MEM_HOST))
    return HIPPartition(n[], out)
end
function HIPPartition(M::AbstractMatrix{<:Integer})                      # partitions.jl:37-60
    Mi = Matrix{UInt32}(M); out = similar(Mi); n = Ref{Int64}(0)
    c = ctx()
    check(c, ccall((:sdpsr_partition_from_u32, libsdpsr), Cint,
                   (Ptr{Cvoid}, Int64, Ptr{UInt32}, Ptr{UInt32}, Ref{Int64}, Cint),
                   c.handle, length(Mi), Mi, out, n, MEM_HOST))
    return HIPPartition(n[], out)
end
function SR.refine!(p::HIPPartition, q::HIPPartition)                   # partitions.jl:62-66
    d = Ref{Int64}(p.nparts); c = ctx()
    check(c, ccall((:sdpsr_refine, libsdpsr), Cint,
                   (Ptr{Cvoid}, Int64, Ptr{UInt32}, Ref{Int64}, Ptr{UInt32}, Int64, Cint),
                   c.handle, length(p.matrix), p.matrix, d, q.matrix, q.nparts, MEM_HOST))
    p.nparts = d[]
    return p
end
function Base.fill!(M::AbstractMatrix{Float64}, p::HIPPartition; values::AbstractVector) # :68-75
    @assert length(values) == SR.dim(p)
    v = Vector{Float64}(values); c = ctx()
    check(c, ccall((:sdpsr_fill, libsdpsr), Cint,
                   (Ptr{Cvoid}, Int64, Ptr{UInt32}, Ptr{Float64}, Int64, Ptr{Float64}, Cint),
                   c.handle, length(p.matrix), p.matrix, v, length(v), M, MEM_HOST))
    return M
end
SR._constraints(p::HIPPartition) = SR._constraints(SR.Partition{UInt32}(p.nparts, p.matrix))

# 128-bit checksum of the canonical labels: a probabilistic `==` (partitions.jl:16-17) that lets
# independent restarts on several GPUs agree without exchanging the n x n label matrices
function checksum(p::HIPPartition)
    out = zeros(UInt64, 2); c = ctx()
    check(c, ccall((:sdpsr_partition_checksum, libsdpsr), Cint,
                   (Ptr{Cvoid}, Int64, Ptr{UInt32}, Ptr{UInt64}, Cint),
                   c.handle, length(p.matrix), p.matrix, out, MEM_HOST))
    return (out[1], out[2])
end

# ---- admissible_subspace: setup on the host (partitions.jl:117-142), loop on the device ----
function SR.admissible_subspace(::Type{HIPPartition}, C::AbstractVector{T}, A::AbstractMatrix{T},
                                b::AbstractVector{T}; verbose::Bool=false,
                                atol=Base.rtoldefault(real(T))) where {T<:AbstractFloat}
    n = isqrt(length(C)); @assert n^2 == length(C)
    A′ = A'; F = qr(A′)
    U = Matrix(F.Q)[:, 1:rank(A)]                      # orthonormal basis of rowspace(A)
    proj(v) = U * (U' * v)
    c = Vector(C); c .-= proj(c); SR._clamp_round!(c, atol=atol); SR._symmetrize!(c, n)
    x0, _ = SR.Krylov.craig(A, b); SR._symmetrize!(x0, n); x0 = proj(x0); SR._clamp_round!(x0, atol=atol)
    P = Matrix{UInt32}(undef, n, n); d = Ref{Int64}(0); it = Ref{Int32}(0); cx = ctx()
    # symmetric basis matrices (the usual case): the projection step may work on the lower triangle
    # (bit 1: c and x0 were symmetrised above)
    hint = 2 | (all(k -> (M = reshape(view(U, :, k), n, n); isapprox(M, M'; atol=1e-12, rtol=0)), 1:size(U, 2)) ? 1 : 0)
    ccall((:sdpsr_hint_symmetric_basis, libsdpsr), Cint, (Ptr{Cvoid}, Cint), cx.handle, hint)
    check(cx, ccall((:sdpsr_admissible_subspace, libsdpsr), Cint,
                    (Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Float64,
                     Ptr{UInt32}, Ref{Int64}, Ref{Int32}, Ptr{Float64}, Cint),
                    cx.handle, n, c, x0, U, size(U, 2), atol, P, d, it, C_NULL, MEM_HOST))
    if verbose  # the reference's log lines (partitions.jl:150,156,187-188) from the dimension trajectory
        cnt = Ref{Int32}(0)
        ccall((:sdpsr_dimension_trajectory, libsdpsr), Cint, (Ptr{Cvoid}, Ptr{Int64}, Int32, Ref{Int32}), cx.handle, C_NULL, 0, cnt)
        dims = Vector{Int64}(undef, cnt[])
        ccall((:sdpsr_dimension_trajectory, libsdpsr), Cint, (Ptr{Cvoid}, Ptr{Int64}, Int32, Ref{Int32}), cx.handle, dims, cnt[], cnt)
        @info "Starting the reduction. Dimensions:" maximal = (n^2 + n) ÷ 2 initial = dims[1]
        for k in 1:length(dims)-1
            @debug "Iteration $k, Current dimension: $(dims[k])"
        end
        @info "Minimal admissible subspace converged in $(it[]) iterations at dimension:" final = d[]
    end
    return HIPPartition(d[], P)
end

# ---- the whole reduction in ONE call (sdpsr_jordan_reduce): admissible_subspace + blockDiagonalize with the
# partition staying on the device; the images are fetched with sdpsr_block_images once their size is known ----
function jordan_reduce(C::AbstractVector{Float64}, A::AbstractMatrix{Float64}, b::AbstractVector{Float64};
                       atol=Base.rtoldefault(Float64), epsilon=Base.rtoldefault(Float64))
    n = isqrt(length(C)); @assert n^2 == length(C)
    F = qr(A'); U = Matrix(F.Q)[:, 1:rank(A)]; proj(v) = U * (U' * v)
    c = Vector(C); c .-= proj(c); SR._clamp_round!(c, atol=atol); SR._symmetrize!(c, n)
    x0, _ = SR.Krylov.craig(A, b); SR._symmetrize!(x0, n); x0 = proj(x0); SR._clamp_round!(x0, atol=atol)
    P = Matrix{UInt32}(undef, n, n); d = Ref{Int64}(0); it = Ref{Int32}(0); cx = ctx()
    nb = Ref{Int32}(0); ssq = Ref{Int64}(0); ss = Ref{Int64}(0)
    check(cx, ccall((:sdpsr_jordan_reduce, libsdpsr), Cint,
                    (Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Float64, Float64, Ptr{UInt32}, Ref{Int64},
                     Ref{Int32}, Ref{Int32}, Ref{Int64}, Ref{Int64}, Ptr{Float64}, Int64, Ptr{Float64}, Int64, Ptr{Float64}, Cint),
                    cx.handle, n, c, x0, U, size(U, 2), atol, epsilon, P, d, it, nb, ssq, ss, C_NULL, 0, C_NULL, 0, C_NULL, MEM_HOST))
    sizes = Vector{Int32}(undef, nb[])
    check(cx, ccall((:sdpsr_block_sizes, libsdpsr), Cint, (Ptr{Cvoid}, Ptr{Int32}), cx.handle, sizes))
    flat = Vector{Float64}(undef, d[] * ssq[])
    check(cx, ccall((:sdpsr_block_images, libsdpsr), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Cint),
                    cx.handle, flat, C_NULL, C_NULL, MEM_HOST))
    return HIPPartition(d[], P), Int.(sizes), reshape(flat, Int(ssq[]), Int(d[]))   # column i = the blocks of class i, concatenated
end

# ---- blockDiagonalize(Float64, P) (compat.jl:46-68) -------------------------------------
function SR.blockDiagonalize(::Type{Float64}, P::HIPPartition, verbose=true;
                             epsilon=Base.rtoldefault(Float64))
    n = size(P, 1); cx = ctx()
    nb = Ref{Int32}(0); ssq = Ref{Int64}(0); ss = Ref{Int64}(0)
    check(cx, ccall((:sdpsr_block_diagonalize, libsdpsr), Cint,
                    (Ptr{Cvoid}, Int64, Ptr{UInt32}, Int64, Float64, Ref{Int32}, Ref{Int64}, Ref{Int64},
                     Ptr{Float64}, Cint),
                    cx.handle, n, P.matrix, P.nparts, epsilon, nb, ssq, ss, C_NULL, MEM_HOST))
    sizes = Vector{Int32}(undef, nb[])
    check(cx, ccall((:sdpsr_block_sizes, libsdpsr), Cint, (Ptr{Cvoid}, Ptr{Int32}), cx.handle, sizes))
    flat = Vector{Float64}(undef, P.nparts * ssq[])
    check(cx, ccall((:sdpsr_block_images, libsdpsr), Cint,
                    (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Cint),
                    cx.handle, flat, C_NULL, C_NULL, MEM_HOST))
    blks = Vector{Vector{Matrix{Float64}}}(undef, P.nparts)
    for i in 1:P.nparts
        off = (i - 1) * ssq[]; blks[i] = Matrix{Float64}[]
        for s in sizes
            push!(blks[i], reshape(flat[off+1:off+s*s], Int(s), Int(s))); off += s * s
        end
    end
    return (blkSizes=Int.(sizes), blks=blks)
end

# ---- diagonalize(Float64, P) (diagonalize.jl:25-40): Q_hat itself ---------------------------
function SR.diagonalize(::Type{Float64}, P::HIPPartition; verbose=false, atol=1e-12 * size(P, 1))
    n = size(P, 1); cx = ctx()
    nb = Ref{Int32}(0); ssq = Ref{Int64}(0); ss = Ref{Int64}(0)
    st = ccall((:sdpsr_block_diagonalize, libsdpsr), Cint,
               (Ptr{Cvoid}, Int64, Ptr{UInt32}, Int64, Float64, Ref{Int32}, Ref{Int64}, Ref{Int64}, Ptr{Float64}, Cint),
               cx.handle, n, P.matrix, P.nparts, atol, nb, ssq, ss, C_NULL, MEM_HOST)
    st == 3 || check(cx, st)   # check_block_sizes belongs to blockDiagonalize (compat.jl:60), not to diagonalize
    sizes = Vector{Int32}(undef, nb[])
    check(cx, ccall((:sdpsr_block_sizes, libsdpsr), Cint, (Ptr{Cvoid}, Ptr{Int32}), cx.handle, sizes))
    Q = Matrix{Float64}(undef, n, ss[])
    check(cx, ccall((:sdpsr_q_hat, libsdpsr), Cint, (Ptr{Cvoid}, Ptr{Float64}, Cint), cx.handle, Q, MEM_HOST))
    offs = cumsum(vcat(0, Int.(sizes)))
    return [Q[:, offs[k]+1:offs[k+1]] for k in eachindex(sizes)]
end

# ---- blockDiagonalize(ComplexF64, P) (compat.jl:26-32,54-57; n <= 3072 in this library version) ---
function SR.blockDiagonalize(::Type{ComplexF64}, P::HIPPartition, verbose=true;
                             epsilon=Base.rtoldefault(Float64))
    n = size(P, 1); cx = ctx()
    Pd = Matrix{UInt32}(undef, n, n); dd = Ref{Int64}(0)
    nb = Ref{Int32}(0); ssq = Ref{Int64}(0); ss = Ref{Int64}(0)
    check(cx, ccall((:sdpsr_block_diagonalize_complex, libsdpsr), Cint,
                    (Ptr{Cvoid}, Int64, Ptr{UInt32}, Int64, Float64, Ptr{UInt32}, Ref{Int64}, Ref{Int32}, Ref{Int64},
                     Ref{Int64}, Cint),
                    cx.handle, n, P.matrix, P.nparts, epsilon, Pd, dd, nb, ssq, ss, MEM_HOST))
    sizes = Vector{Int32}(undef, nb[])
    check(cx, ccall((:sdpsr_block_sizes_complex, libsdpsr), Cint, (Ptr{Cvoid}, Ptr{Int32}), cx.handle, sizes))
    flat = Vector{ComplexF64}(undef, dd[] * ssq[])      # (re, im) pairs = ComplexF64 layout
    check(cx, ccall((:sdpsr_block_images_complex, libsdpsr), Cint, (Ptr{Cvoid}, Ptr{ComplexF64}, Ptr{ComplexF64}, Cint),
                    cx.handle, flat, C_NULL, MEM_HOST))
    blks = Vector{Vector{Matrix{ComplexF64}}}(undef, dd[])
    for i in 1:dd[]
        off = (i - 1) * ssq[]; blks[i] = Matrix{ComplexF64}[]
        for s in sizes
            push!(blks[i], reshape(flat[off+1:off+s*s], Int(s), Int(s))); off += s * s
        end
    end
    return (blkSizes=Int.(sizes), blks=blks)
end

# ---- R independent random restarts of the reduction in ONE call on this task's thread (sdpsr_jordan_reduce_batch):
# the reference's "try again" after NumericalInconsistency / DimensionMismatch (eigen_decomposition.jl:264-270,
# diagonalize.jl:4-9) run side by side; returns the restarts' partitions, statuses and block-size sums, first the sizes
# (images: call jordan_reduce / blockDiagonalize on the partition of the first restart whose status is 0) ----
function jordan_reduce_batch(C::AbstractVector{Float64}, A::AbstractMatrix{Float64}, b::AbstractVector{Float64}, R::Integer;
                             seeds::Union{Nothing,Vector{UInt64}}=nothing, atol=Base.rtoldefault(Float64),
                             epsilon=Base.rtoldefault(Float64))
    n = isqrt(length(C)); @assert n^2 == length(C)
    F = qr(A'); U = Matrix(F.Q)[:, 1:rank(A)]; proj(v) = U * (U' * v)
    c = Vector(C); c .-= proj(c); SR._clamp_round!(c, atol=atol); SR._symmetrize!(c, n)
    x0, _ = SR.Krylov.craig(A, b); SR._symmetrize!(x0, n); x0 = proj(x0); SR._clamp_round!(x0, atol=atol)
    cx = ctx()
    Ps = [Matrix{UInt32}(undef, n, n) for _ in 1:R]
    pP = [pointer(P) for P in Ps]
    d = zeros(Int64, R); it = zeros(Int32, R); nb = zeros(Int32, R); ssq = zeros(Int64, R); ss = zeros(Int64, R); st = zeros(Int32, R)
    hint = 2 | (all(k -> (M = reshape(view(U, :, k), n, n); isapprox(M, M'; atol=1e-12, rtol=0)), 1:size(U, 2)) ? 1 : 0)
    ccall((:sdpsr_hint_symmetric_basis, libsdpsr), Cint, (Ptr{Cvoid}, Cint), cx.handle, hint)
    GC.@preserve Ps begin
        ccall((:sdpsr_jordan_reduce_batch, libsdpsr), Cint,
              (Ptr{Cvoid}, Int32, Ptr{UInt64}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Float64, Float64,
               Ptr{Ptr{UInt32}}, Ptr{Int64}, Ptr{Int32}, Ptr{Int32}, Ptr{Int64}, Ptr{Int64}, Ptr{Ptr{Float64}}, Ptr{Int64},
               Ptr{Int32}, Cint),
              cx.handle, R, seeds === nothing ? C_NULL : seeds, n, c, x0, U, size(U, 2), atol, epsilon, pP, d, it, nb, ssq, ss,
              C_NULL, C_NULL, st, MEM_HOST)
    end
    return [(status=Int(st[i]), P=HIPPartition(Int(d[i]), Ps[i]), iterations=Int(it[i]), nblocks=Int(nb[i]),
             sum_sq=Int(ssq[i]), sum_s=Int(ss[i])) for i in 1:R]
end
# (the library uploads C_L, X0_L, U once for the R restarts of this call; a caller who makes SEVERAL such calls on one
# problem keeps a `Problem` and calls `reduce_batch(problem, R)`: nothing is uploaded again)

# ---- upload once, restart many (sdpsr_problem_create / sdpsr_problem_reduce_batch): C_L, X0_L, U travel to the device
# ONCE; every later reduce / reduce_batch call names the handle and moves only its results.  `mem` = MEM_DEVICE takes
# device pointers instead (an AMDGPU.jl ROCArray: pass `pointer(a)` of the ROCArray{Float64} -- they are copied on the
# device, the caller's arrays are free on return). ----
const MEM_DEVICE = Cint(1)
mutable struct Problem
    handle::Ptr{Cvoid}
    n::Int
    ctx::Context
    function Problem(C::AbstractVector{Float64}, A::AbstractMatrix{Float64}, b::AbstractVector{Float64};
                     atol=Base.rtoldefault(Float64), cx::Context=ctx())
        n = isqrt(length(C)); @assert n^2 == length(C)
        F = qr(A'); U = Matrix(F.Q)[:, 1:rank(A)]; proj(v) = U * (U' * v)
        c = Vector(C); c .-= proj(c); SR._clamp_round!(c, atol=atol); SR._symmetrize!(c, n)
        x0, _ = SR.Krylov.craig(A, b); SR._symmetrize!(x0, n); x0 = proj(x0); SR._clamp_round!(x0, atol=atol)
        hint = 2 | (all(k -> (M = reshape(view(U, :, k), n, n); isapprox(M, M'; atol=1e-12, rtol=0)), 1:size(U, 2)) ? 1 : 0)
        h = Ref{Ptr{Cvoid}}(C_NULL)
        check(cx, ccall((:sdpsr_problem_create, libsdpsr), Cint,
                        (Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Int64, Cint, Cint, Ref{Ptr{Cvoid}}),
                        cx.handle, n, c, x0, U, size(U, 2), hint, MEM_HOST, h))
        p = new(h[], n, cx)
        finalizer(q -> (q.handle != C_NULL && ccall((:sdpsr_problem_destroy, libsdpsr), Cint, (Ptr{Cvoid},), q.handle); q.handle = C_NULL), p)
        return p
    end
end

function reduce_batch(p::Problem, R::Integer; seeds::Union{Nothing,Vector{UInt64}}=nothing, atol=Base.rtoldefault(Float64),
                      epsilon=Base.rtoldefault(Float64))
    n = p.n; cx = p.ctx
    Ps = [Matrix{UInt32}(undef, n, n) for _ in 1:R]
    pP = [pointer(P) for P in Ps]
    d = zeros(Int64, R); it = zeros(Int32, R); nb = zeros(Int32, R); ssq = zeros(Int64, R); ss = zeros(Int64, R)
    st = fill(Int32(-1), R)
    rc = GC.@preserve Ps ccall((:sdpsr_problem_reduce_batch, libsdpsr), Cint,
              (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Ptr{UInt64}, Float64, Float64, Ptr{Ptr{UInt32}}, Ptr{Int64}, Ptr{Int32}, Ptr{Int32},
               Ptr{Int64}, Ptr{Int64}, Ptr{Ptr{Float64}}, Ptr{Int64}, Ptr{Int32}, Cint),
              cx.handle, p.handle, R, seeds === nothing ? C_NULL : seeds, atol, epsilon, pP, d, it, nb, ssq, ss, C_NULL, C_NULL, st, MEM_HOST)
    (rc != 0 && all(x -> x <= 0, st)) && check(cx, rc)   # a failure before the restarts started
    return [(status=Int(st[i]), P=HIPPartition(Int(d[i]), Ps[i]), iterations=Int(it[i]), nblocks=Int(nb[i]),
             sum_sq=Int(ssq[i]), sum_s=Int(ss[i])) for i in 1:R]
end

# ---- test/numerical_issues.jl:85-94 in one call: `count` runs of eigen_decomposition on all CUs ----
function eigen_decomposition_batched(P::HIPPartition, count::Integer; atol=1e-12 * size(P, 1))
    n = size(P, 1); cx = ctx()
    st = Vector{Int32}(undef, count); ne = similar(st); nc = similar(st)
    check(cx, ccall((:sdpsr_eigen_decomposition_batched, libsdpsr), Cint,
                    (Ptr{Cvoid}, Int64, Ptr{UInt32}, Int64, Float64, Int64, Ptr{Float64}, Ptr{Int32}, Ptr{Int32},
                     Ptr{Int32}, Cint),
                    cx.handle, n, P.matrix, P.nparts, atol, count, C_NULL, st, ne, nc, MEM_HOST))
    return (status=st, neig=ne, nclasses=nc)
end

end # module
