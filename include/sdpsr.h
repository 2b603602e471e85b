/*
 * sdpsr.h -- C ABI of libsdpsr_hip.so: the MI355X (gfx950) implementation of the
 * Jordan-reduction hot path of SDPSymmetryReduction.jl.
 *
 * The reference is pure Julia and has no FFI; its seam is dispatch on the
 * partition type (AbstractPartition contract, src/abstract_part.jl:7-16, exercised
 * by the alternate backend in test/partitions_set.jl:6-92,102-105).  A Julia
 * backend type `HIPPartition` binds these entry points with `ccall`
 * (INTEGRATION.md shows the stub).  Every entry point cites the reference lines it
 * replaces (paths relative to the reference repo root).
 *
 * Conventions
 *  - all matrices column-major, n x n unless stated; "len" = number of entries
 *    scanned in column-major linear order (the only order the reference uses);
 *  - labels are uint32: 0 = structurally-zero class (not counted), 1..dim;
 *  - every array argument lives in the memory space named by `mem`
 *    (SDPSR_MEM_HOST: caller-owned host memory, copied by the library;
 *     SDPSR_MEM_DEVICE: device pointers on ctx's device, used in place);
 *  - scalar outputs (int64_t* etc.) are always host pointers;
 *  - calls on one ctx are serialised on ctx's HIP stream; distinct ctxs (on the same or on
 *    different devices) may be used from distinct host threads; no global state; no callbacks;
 *  - ordering of SDPSR_MEM_DEVICE arguments: inputs must be complete when the call is made, or
 *    be produced on a stream that ctx has been told about (sdpsr_set_stream, or
 *    sdpsr_wait_stream right before the call); on return from EVERY entry point the outputs
 *    are complete (the call synchronises ctx's stream before it returns);
 *  - return value: sdpsr_status; details via sdpsr_last_error(ctx).
 */
#ifndef SDPSR_H
#define SDPSR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SDPSR_VERSION_MAJOR 0
#define SDPSR_VERSION_MINOR 2

typedef struct sdpsr_ctx sdpsr_ctx;

typedef enum sdpsr_status {
    SDPSR_OK = 0,
    /* InvalidDecompositionField, src/eigen_decomposition.jl:140-150,247-253 */
    SDPSR_INVALID_DECOMPOSITION_FIELD = 1,
    /* NumericalInconsistency, src/eigen_decomposition.jl:152-161,264-270 */
    SDPSR_NUMERICAL_INCONSISTENCY = 2,
    /* DimensionMismatch from check_block_sizes, src/diagonalize.jl:1-11 */
    SDPSR_DIMENSION_MISMATCH = 3,
    /* Julia's InexactError on label overflow, src/partitions.jl:63 */
    SDPSR_LABEL_OVERFLOW = 4,
    /* @assert n^2 == length(C) (src/partitions.jl:118), length(values)==dim(P) (:69) ... */
    SDPSR_BAD_ARGUMENT = 5,
    SDPSR_HIP_ERROR = 6,
    SDPSR_SOLVER_ERROR = 7,
    SDPSR_OUT_OF_MEMORY = 8,
    SDPSR_NOT_CONVERGED = 9,
    SDPSR_BAD_STATE = 10
} sdpsr_status;

typedef enum sdpsr_mem { SDPSR_MEM_HOST = 0, SDPSR_MEM_DEVICE = 1 } sdpsr_mem;

/* How the random square of src/partitions.jl:172 is evaluated. */
typedef enum sdpsr_square_mode {
    SDPSR_SQUARE_AUTO = 0, /* = I8 */
    /* exact integers: `channels` independent draws of int8 class values, int8 MFMA,
       int32 accumulate; classes compared by exact equality (no rounding involved) */
    SDPSR_SQUARE_I8 = 1,
    /* exact integers on the fp32 MFMA (|v| <= floor(sqrt(2^24/n)) so every partial sum
       is an exactly representable integer) */
    SDPSR_SQUARE_F32 = 2,
    /* reference-literal: uniform [0,1) doubles, fp64 MFMA, 7-digit rounding
       (src/utils.jl:34-53) */
    SDPSR_SQUARE_F64 = 3
} sdpsr_square_mode;

typedef struct sdpsr_opts {
    uint32_t struct_size;   /* = sizeof(sdpsr_opts) */
    int32_t square_mode;    /* sdpsr_square_mode */
    int32_t channels;       /* independent draws per square step (1..8), 0 = default 4 */
    int32_t max_iters;      /* 0 = default (10000) */
    int32_t confirm_rounds; /* extra no-change square rounds demanded before stopping */
    int32_t eig_driver;     /* 0 = default */
    int32_t reserved[10];
} sdpsr_opts;

/* phase_ms slots filled by sdpsr_admissible_subspace / sdpsr_block_diagonalize
   (the reference only prints @timed per phase: src/diagonalize.jl:32-37,
   src/compat.jl:63-65) */
enum {
    SDPSR_T_TOTAL = 0,
    SDPSR_T_PROJECT = 1,   /* randomize + projection + signature */
    SDPSR_T_SQUARE = 2,    /* randomize + N x N square(s) */
    SDPSR_T_REFINE = 3,    /* partition refinement + canonical relabel */
    SDPSR_T_EIGEN = 4,     /* symmetric eigendecomposition */
    SDPSR_T_ISO = 5,       /* Q'AQ, block norms, isomorphism classes */
    SDPSR_T_IRRED = 6,     /* irreducible_decomposition */
    SDPSR_T_IMAGE = 7,     /* basis_image */
    SDPSR_T_COUNT = 8
};

/* ---- lifecycle ------------------------------------------------------------ */
/* opts may be NULL (defaults).  seed drives every random draw of the ctx. */
int sdpsr_create(int device_id, uint64_t seed, const sdpsr_opts* opts, sdpsr_ctx** out);
void sdpsr_destroy(sdpsr_ctx* ctx);
const char* sdpsr_last_error(const sdpsr_ctx* ctx);
const char* sdpsr_status_string(int status);
int sdpsr_version(void);
/* Use an existing HIP stream (hipStream_t) for all work of ctx; NULL = ctx's own. */
int sdpsr_set_stream(sdpsr_ctx* ctx, void* hip_stream);
int sdpsr_synchronize(sdpsr_ctx* ctx);
/* Make all later work of ctx wait for what has been submitted to `hip_stream` (hipStream_t,
   NULL = the legacy default stream) so far: event record + hipStreamWaitEvent, no host wait.
   For device-resident arguments produced by the caller's own kernels. */
int sdpsr_wait_stream(sdpsr_ctx* ctx, void* hip_stream);
/* Hints for the NEXT sdpsr_admissible_subspace call on this ctx (bit mask `yes`):
   bit 0: the columns of U, read as n x n matrices, are symmetric (true whenever the constraint
          matrices A_i are; a host-side setup knows).  With symmetric labels the projection step then
          works on the lower triangle only (half the bytes).  Without it the first iteration's
          dot-product pass carries a randomized symmetry probe (<U_k, W - W'> for a pseudo-random W)
          and later iterations use its verdict.
   bit 1: CL and X0L are symmetric (the reference symmetrises both, src/partitions.jl:128-141): the
          initial partition is formed from the lower triangle.
   A wrong hint is the caller's error (the result is then the partition of the mirrored lower triangle). */
int sdpsr_hint_symmetric_basis(sdpsr_ctx* ctx, int yes);
/* Reseed (tests; independent restarts use distinct seeds per rank). */
int sdpsr_set_seed(sdpsr_ctx* ctx, uint64_t seed);

/* ---- AbstractPartition contract (primitives) ------------------------------- */
/* Partition{T}(M::AbstractMatrix) float ctor, src/partitions.jl:24-35: classes of
   bit-equal values, labelled by first occurrence in column-major order, +0.0 -> 0. */
int sdpsr_partition_from_f64(sdpsr_ctx* ctx, int64_t len, const double* M,
                             uint32_t* labels, int64_t* nparts, int mem);
/* Integer ctor + __sort_unique!, src/partitions.jl:37-60. in == out allowed. */
int sdpsr_partition_from_u32(sdpsr_ctx* ctx, int64_t len, const uint32_t* in,
                             uint32_t* labels, int64_t* nparts, int mem);
/* refine!(P1, P2), src/partitions.jl:62-66: p1 <- canonical relabel of the pairs
   (p1, p2); label 0 only where both are 0.  *d1 is updated.
   Classes are told apart by a 64-bit mixing hash of the pair (the reference's exact pair code
   l1 + l2 * (d1 + 1) overflows its label type, src/partitions.jl:63): two distinct pairs collide
   -- and are merged -- with probability ~ d^2 / 2^65 per call (2e-6 at 8M classes, 3e-17 at 34). */
int sdpsr_refine(sdpsr_ctx* ctx, int64_t len, uint32_t* p1, int64_t* d1,
                 const uint32_t* p2, int64_t d2, int mem);
/* Base.:(==)(p::Partition, q::Partition), src/partitions.jl:16-17 (same matrix), as a 128-bit
   position-weighted checksum of the canonical label matrix: equal partitions have equal
   checksums, different ones collide with probability ~2^-64 per word.  Used to agree the result
   of independent restarts across GPUs without moving the n x n labels.
   out[0..1] is host memory; `mem` says where `labels` lives. */
int sdpsr_partition_checksum(sdpsr_ctx* ctx, int64_t len, const uint32_t* labels, uint64_t* out, int mem);
/* fill!(M, P; values), src/partitions.jl:68-75: M[idx] = values[label-1], 0 -> 0.0.
   `values` has d entries. */
int sdpsr_fill(sdpsr_ctx* ctx, int64_t len, const uint32_t* labels, const double* values,
               int64_t d, double* M, int mem);
/* randomize!(M, P), src/abstract_part.jl:107-110: one uniform [0,1) draw per class from
   the ctx's counter-based generator (a fresh stream per call). */
int sdpsr_randomize(sdpsr_ctx* ctx, int64_t len, const uint32_t* labels, double* M, int mem);
/* _clamp_round!, src/utils.jl:34-53; in place.  BEHAVIOURAL DIFFERENCE a Julia caller will
   see: the reference TRUNCATES the 7-digit decimal mantissa (unsafe_trunc, src/utils.jl:49-53),
   this library rounds it TO NEAREST.  Values that sit on a truncation edge (0.0625 = 0.5 * 2^-3)
   are split into two classes by last-bit noise under truncation (esc16j: 10712 classes instead
   of the pinned 150 with NumPy's projection arithmetic); nearest rounding keeps them together.
   Results differ from Julia's only for entries within 1 ulp of such an edge. */
int sdpsr_clamp_round(sdpsr_ctx* ctx, int64_t len, double* a, double atol, int mem);
/* x .-= projL(x), src/partitions.jl:161 + src/utils.jl:62-66, with qr(A') folded into an
   orthonormal basis U (len x r, column-major) of rowspace(A). */
int sdpsr_project_out(sdpsr_ctx* ctx, int64_t len, double* x, const double* U, int64_t r,
                      int mem);

/* ---- the N x N square, mul!(X2, X, X) src/partitions.jl:172 ----------------- */
/* X symmetric, column-major, leading dimension n. */
int sdpsr_square_f64(sdpsr_ctx* ctx, int64_t n, const double* X, double* X2, int mem);
int sdpsr_square_f32(sdpsr_ctx* ctx, int64_t n, const float* X, float* X2, int mem);
int sdpsr_square_i8(sdpsr_ctx* ctx, int64_t n, const int8_t* X, int32_t* X2, int mem);
/* C = A' * B, all column-major fp64: A is k x m (lda), B is k x n (ldb), C m x n (ldc).
   The Q'AQ products of src/eigen_decomposition.jl:203 and the block products of :70. */
int sdpsr_gemm_tn_f64(sdpsr_ctx* ctx, int64_t m, int64_t n, int64_t k, const double* A,
                      int64_t lda, const double* B, int64_t ldb, double* C, int64_t ldc,
                      int mem);

/* ---- admissible_subspace, src/partitions.jl:109-190 ------------------------- */
/* Loop :145-185 on the device.  The setup stage (:117-142: qr(A'), C_L, CRAIG x0) stays
   with the caller, which passes
     CL   n x n  reshape(_symmetrize!(_clamp_round!(c - projL(c))), n, n)      (:129-134)
     X0L  n x n  reshape(_clamp_round!(projL(_symmetrize!(craig(A,b)))), n, n) (:137-142)
     U    n^2 x r orthonormal basis of rowspace(A)                              (:124)
   Outputs: P_out (n x n labels), *dim_out, *iters_out, phase_ms[SDPSR_T_COUNT] (may be
   NULL).
   BEHAVIOURAL DIFFERENCE (int8 mode, symmetric labels and basis, <= 4 basis matrices): an
   iteration refines the partition ONCE, by the projected random element and the square of a second
   random element of the same partition together; the reference refines after the projection and
   draws the squared element from the refined partition (:159-174).  Both loops end at the same
   partition (the smallest partition subspace containing C_L and X0 that is closed under the
   projection and under squaring; P_out is canonical, so it is the same matrix); *iters_out counts
   joint steps (equal to the reference-structured count on every test problem).
   SDPSR_SEPARATE_REFINEMENTS=1 in the environment restores two refinements per iteration. */
int sdpsr_admissible_subspace(sdpsr_ctx* ctx, int64_t n, const double* CL, const double* X0L,
                              const double* U, int64_t r, double atol, uint32_t* P_out,
                              int64_t* dim_out, int32_t* iters_out, double* phase_ms, int mem);
/* Convenience for dense problems: the setup stage (:117-142) runs on the DEVICE as well
   (pivoted modified Gram-Schmidt with re-orthogonalisation on the rows of A in place of qr(A'),
   C_L, min-norm x0 = U R^-T b), then the loop.  C: n^2, A: m x n^2 column-major, b: m;
   host pointers (copied by the library).  P_out is in `mem_out`. */
int sdpsr_admissible_subspace_dense(sdpsr_ctx* ctx, int64_t n, int64_t m, const double* C,
                                    const double* A, const double* b, double atol,
                                    uint32_t* P_out, int64_t* dim_out, int32_t* iters_out,
                                    double* phase_ms, int mem_out);

/* ---- desymmetrize(P) / unSymmetrize, src/partitions.jl:197-223, src/compat.jl:70 -------------- */
/* WL-type refinement with products X*Y of two independent random elements until the dimension
   stalls: P (n x n labels) is refined in place, *dim updated, *iters (may be NULL) = rounds.
   The products are evaluated exactly (int8 channels), like the squares of admissible_subspace. */
int sdpsr_desymmetrize(sdpsr_ctx* ctx, int64_t n, uint32_t* P, int64_t* dim, int32_t* iters, int mem);

/* ---- reduced-SDP assembly, README.md:57-60, test/sd_problems.jl:32-37 ------------------------- */
/* out = A * PMat with PMat = hcat(vec(P.matrix .== i) for i = 1:d): the columns of A summed per
   class.  A: m x len column-major (dense), labels: len, out: m x d column-major.  C' * PMat is the
   m = 1 case.  Sparse A stays with the caller (one sparse-times-indicator product). */
int sdpsr_reduce_constraints(sdpsr_ctx* ctx, int64_t len, const uint32_t* labels, int64_t d, int64_t m,
                             const double* A, double* out, int mem);

/* ---- blockDiagonalize, src/compat.jl:46-68 ---------------------------------- */
/* Phase 1 = diagonalize(Float64, P; atol=epsilon) (src/diagonalize.jl:25-40) +
   check_block_sizes (:1-11).  Keeps Q_hat on the device inside ctx.
   Outputs: *nblocks, *sum_sq = sum_k s_k^2, *sum_s = sum_k s_k. */
int sdpsr_block_diagonalize(sdpsr_ctx* ctx, int64_t n, const uint32_t* P, int64_t d,
                            double epsilon, int32_t* nblocks, int64_t* sum_sq, int64_t* sum_s,
                            double* phase_ms, int mem);
/* blkSizes of the last phase 1 that got as far as check_block_sizes (host array of nblocks ints). */
int sdpsr_block_sizes(sdpsr_ctx* ctx, int32_t* blk_sizes);
/* Q_hat = diagonalize(Float64, P; atol) itself (src/diagonalize.jl:25-40), n x sum_s column-major,
   blocks side by side: available after sdpsr_block_diagonalize returned OK or
   SDPSR_DIMENSION_MISMATCH (diagonalize does not run check_block_sizes, src/compat.jl:60 does). */
int sdpsr_q_hat(sdpsr_ctx* ctx, double* Q_hat, int mem);
/* Phase 2 = basis_image(Q_hat, P) (src/diagonalize.jl:64-89).
   blks: d * sum_sq doubles, class-major, then block, each block column-major s_k x s_k.
   Q_hat (optional, may be NULL): n x sum_s column-major, blocks side by side. */
int sdpsr_block_images(sdpsr_ctx* ctx, double* blks, double* Q_hat, double* phase_ms, int mem);
/* ---- blockDiagonalize(P; complex = true), src/compat.jl:26-32,46-68 with T = ComplexF64 ---------
   diagonalize(ComplexF64, P) = desymmetrize (src/diagonalize.jl:26-28) + Murota's decomposition
   over C + check_block_sizes with sum s_k^2 == dim(P) (:13-23); the partition handed to
   basis_image is the desymmetrized one (src/compat.jl:54-57).
   n <= 64: every step in single-workgroup kernels (the reference's own complex tests are 3 x 3
   and 4 x 4, test/runtests.jl:43-57).  64 < n <= 3072: the Hermitian elements go through their real
   symmetric embedding (2n x 2n: the real dense eigensolver and the fp64 MFMA GEMMs; eigenspaces
   are extracted per eigenvalue cluster, SDPSR_NUMERICAL_INCONSISTENCY if a cluster does not
   split evenly).  Larger n: SDPSR_BAD_ARGUMENT.
   BEHAVIOURAL DIFFERENCE: the generic elements are Hermitian (A + A^H with complex class
   coefficients) and the eigensolver is a Hermitian Jacobi iteration, where the reference hands a
   general complex element to eigen(); block sizes and block spectra are the same.
   P_desym (optional, n x n labels in `mem`) / *d_desym: the desymmetrized partition whose classes
   index the block images. */
int sdpsr_block_diagonalize_complex(sdpsr_ctx* ctx, int64_t n, const uint32_t* P, int64_t d, double epsilon,
                                    uint32_t* P_desym, int64_t* d_desym, int32_t* nblocks, int64_t* sum_sq,
                                    int64_t* sum_s, int mem);
int sdpsr_block_sizes_complex(sdpsr_ctx* ctx, int32_t* blk_sizes);
/* blks: d_desym * sum_sq complex numbers as (re, im) pairs, class-major, then block, each block
   column-major s_k x s_k.  Q_hat (optional): n x sum_s complex, (re, im) pairs, column-major. */
int sdpsr_block_images_complex(sdpsr_ctx* ctx, double* blks, double* Q_hat, int mem);

/* eigen_decomposition(P, A; atol), src/eigen_decomposition.jl:236-273: status only
   (test/numerical_issues.jl:91-94), *neig = number of eigenspaces, *nclasses = number of
   isomorphism classes. */
int sdpsr_eigen_decomposition(sdpsr_ctx* ctx, int64_t n, const uint32_t* P, int64_t d,
                              double atol, int32_t* neig, int32_t* nclasses, int mem);
/* Batched small-N mode: `count` independent runs of eigen_decomposition(P, A; atol) on ONE
   partition, each with its own pair of generic elements -- the shape of the reference's
   robustness pin (test/numerical_issues.jl:85-94: 10 000 runs on a 64 x 64 partition, none may
   throw).  For n <= 64 one workgroup handles one run (Jacobi eigensolver, Q'AQ, block norms,
   Otsu threshold, union-find and __isconsistent all inside the workgroup) and all CUs work in
   parallel; larger n goes through the single-problem path run by run.
   values: NULL (fresh draws from ctx's generator) or 2*count*d doubles in `mem` -- run r uses
           values[(2r)*d ...] as the class values of element #1 and values[(2r+1)*d ...] of #2
           (n <= 64 only; lets a caller replay given elements);
   status/neig/nclasses: host arrays of `count` ints, each may be NULL.
   Returns SDPSR_OK when every run is OK, else the status of the first failing run. */
int sdpsr_eigen_decomposition_batched(sdpsr_ctx* ctx, int64_t n, const uint32_t* P, int64_t d, double atol,
                                      int64_t count, const double* values, int32_t* status, int32_t* neig,
                                      int32_t* nclasses, int mem);
/* Symmetric eigendecomposition used by the path (eigen(A), :246): ascending values,
   orthonormal vectors (column-major n x n, overwrites nothing of A). */
int sdpsr_syev_f64(sdpsr_ctx* ctx, int64_t n, const double* A, double* values, double* vectors,
                   int mem);

/* ---- measurement hook (bench.py roofline leg) --------------------------------- */
/* Times `reps` back-to-back launches of one hot kernel on ctx's stream with HIP events, on
   resident synthetic data of order n (n is rounded up to the kernel's tile).  kind:
   0 = square int8 MFMA (one channel), 1 = square fp32 MFMA, 2 = square / Q'AQ fp64 MFMA,
   3 = partition refine of n*n signatures with `aux` distinct classes,
   4 = fused gather+projection+signature pass with r = aux basis vectors,
   5 = the tridiagonalisation's symmetric-product (symv) kernel, one launch per column j = 0..n-2
       (average over the n-1 launches), 6 = one whole tridiagonalisation of order n,
   8 = the one-workgroup Jacobi eigensolver on a random symmetric matrix of order n <= 128,
   9 = the label product Y = A(v) W of the module-compression driver (n x n labels with d classes,
       W n x w, G elements per pass): aux = w | G << 8 | d << 12; with bit 30 of aux set the call
       returns in ms_per_launch[0] the largest absolute deviation of sampled rows of Y from a host
       evaluation in extended precision instead of the time.
   ms_per_launch[0] = average milliseconds per launch. */
int sdpsr_profile_kernel(sdpsr_ctx* ctx, int kind, int64_t n, int64_t aux, int reps,
                         double* ms_per_launch);
/* The same measurement with the shader clock sampled meanwhile by a one-wave kernel on a side
   stream (clock64 against the 100 MHz wall_clock64, ~20 us intervals): out[0] = ms per launch,
   out[1] = median shader clock in MHz while the timed launches ran, out[2] = intervals used.
   The int8 squares run power-limited (the clock drops under the kernel); the roofline of
   bench.py reports the fraction of the peak both at the nominal and at this measured clock. */
int sdpsr_profile_clock(sdpsr_ctx* ctx, int kind, int64_t n, int64_t aux, int reps, double* out);

#ifdef __cplusplus
}
#endif
#endif /* SDPSR_H */
