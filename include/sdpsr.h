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
#define SDPSR_VERSION_MINOR 5

typedef struct sdpsr_ctx sdpsr_ctx;
typedef struct sdpsr_problem sdpsr_problem; /* device-resident inputs of the loop, sdpsr_problem_create */

typedef enum sdpsr_status {
    SDPSR_OK = 0,
    /* InvalidDecompositionField, src/eigen_decomposition.jl:140-150,247-253 */
    SDPSR_INVALID_DECOMPOSITION_FIELD = 1,
    /* NumericalInconsistency, src/eigen_decomposition.jl:152-161,264-270 */
    SDPSR_NUMERICAL_INCONSISTENCY = 2,
    /* DimensionMismatch from check_block_sizes, src/diagonalize.jl:1-11 */
    SDPSR_DIMENSION_MISMATCH = 3,
    /* Julia's InexactError on label overflow (src/partitions.jl:29,63).  Only with sdpsr_opts.label_bits = 8 / 16 / 32
       (the width of the reference's label type T): sdpsr_partition_from_* when the class count exceeds typemax(T);
       sdpsr_refine when the largest pair code l1 + l2 (dim(P1) + 1) does (exactly the reference's condition);
       sdpsr_admissible_subspace when dim(S) after a refinement does -- a NECESSARY condition of the reference's
       error there (its intermediate pair code may overflow although the refined dimension fits: the loop never
       forms Part(X) on its own, so that case is not reproduced).  label_bits = 0 (default): never returned,
       labels are uint32 with 64-bit signatures. */
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

/* How _clamp_round! / unsafe_round (src/utils.jl:34-53) treats the 7-digit decimal mantissa. */
typedef enum sdpsr_round_mode {
    SDPSR_ROUND_NEAREST = 0, /* default: round to nearest (see sdpsr_clamp_round) */
    SDPSR_ROUND_TRUNC = 1    /* reference-literal: unsafe_trunc(Int, scale * x) / scale */
} sdpsr_round_mode;

/* Behaviour switches (sdpsr_opts.flags).  They replace the environment variables of ABI 0.2: the
   library reads no environment variable that changes results (SDPSR_DEBUG only adds stderr traces). */
enum {
    /* admissible_subspace: two refinements per iteration, as src/partitions.jl:159-174 is written,
       instead of the joint one (see sdpsr_admissible_subspace) */
    SDPSR_FLAG_SEPARATE_REFINEMENTS = 1u << 0,
    /* irreducible_decomposition draws its own generic element (src/eigen_decomposition.jl:306)
       instead of reusing the products T = A2 Q of the isomorphism step */
    SDPSR_FLAG_FRESH_IRREDUCIBLE_ELEMENT = 1u << 1,
    /* module-compression driver: always run the second orthonormalisation step of an absorb */
    SDPSR_FLAG_ALWAYS_REORTHOGONALIZE = 1u << 2,
    /* refinement: signatures always through an array in HBM (no fusion into the insert pass) */
    SDPSR_FLAG_REFINE_NO_FUSE = 1u << 3,
    /* int8 loop: unpack the lower-triangle labels to the full matrix after every refinement */
    SDPSR_FLAG_UNPACK_EVERY_STEP = 1u << 4,
    /* module growth: one label product per generic element instead of one pass for all of a round */
    SDPSR_FLAG_SPMM_ONE_BY_ONE = 1u << 5,
    /* dense driver: never raise the coupling matrix by extra generic elements (reference-literal
       single element, src/eigen_decomposition.jl:259-262) */
    SDPSR_FLAG_SINGLE_COUPLING_ELEMENT = 1u << 6,
    /* compressed problems (order <= 64): eigensolver and Murota's steps in one-workgroup kernels on
       the device instead of on the host inside the read-back the driver makes anyway */
    SDPSR_FLAG_SMALL_EIGEN_ON_DEVICE = 1u << 7,
    /* dense driver: launch the kernels of the tridiagonalisation one by one instead of replaying
       their hipGraph (per-kernel profiles) */
    SDPSR_FLAG_NO_GRAPH = 1u << 8,
    /* admissible_subspace: every round runs the full refinement (insert / rank / label passes); by
       default a round that is expected not to refine -- a confirm round, the first iteration -- first
       compares every entry with the representative of its class and skips the relabel when none differs */
    SDPSR_FLAG_NO_VERIFY_SHORTCUT = 1u << 9,
    /* basis_image: always the projection formula Q_k' 1[P==i] Q_k (src/diagonalize.jl:64-89); by default, when every
       block is 1 x 1, the images are read off as eigenvalues, blks[i][k] = q_k'(1[P==i] x) with x = sum_k q_k, under
       a randomized self-check that falls back to the projection formula (see sdpsr_block_images) */
    SDPSR_FLAG_FULL_BASIS_IMAGE = 1u << 10,
    /* admissible_subspace: run the projection half of every iteration (src/partitions.jl:159-164).  By default, once every
       basis matrix U_k is found constant on the classes of S (checked on the device after a refinement), that half is
       skipped for the rest of the call: x - U U'x of a class-constant x is then class-constant for every x, it cannot
       refine S any more (see sdpsr_admissible_subspace) */
    SDPSR_FLAG_ALWAYS_PROJECT = 1u << 11,
    /* dense driver: the panel form of the tridiagonalisation (two launches per column, rank-64 trailing updates on
       the matrix cores) at every order; by default orders <= 2048 take the one-launch-per-column row form */
    SDPSR_FLAG_SYTRD_PANELS = 1u << 12,
    /* eigen_decomposition: the coupling matrix of the eigenspaces (block norms, src/eigen_decomposition.jl:177-217) is
       always read back and thresholded on the host; by default, from 256 eigenspaces on, its extrema, the histogram
       counts of the Otsu threshold and one bit per pair are formed on the device (same classes) */
    SDPSR_FLAG_COUPLING_ON_HOST = 1u << 13,
    /* dense driver, orders in (2048, 8192]: the panel columns of the tridiagonalisation with ONE launch per column -- the
       product is taken with the unnormalised column, the reflector is finished by the next launch, redundantly in every
       tile (csrc/kernels_sytrd_look.hip).  Same results to rounding; measured slower than the two-launch form on MI355X
       (DESIGN.md 4.1), kept for comparison */
    SDPSR_FLAG_SYTRD_ONE_LAUNCH = 1u << 14,
    /* A/B: every host wait of the loop is a wait for the stream and every verdict is read where it arises (rounds 1-4) --
       no return on the label pass's report, no verdicts deferred to the reduction's later waits (round 5).  Same
       partition, iterations and dimension trajectory either way. */
    SDPSR_FLAG_WAIT_FOR_EVERY_VERDICT = 1u << 15
};

typedef struct sdpsr_opts {
    uint32_t struct_size;   /* = sizeof(sdpsr_opts) */
    int32_t square_mode;    /* sdpsr_square_mode */
    int32_t channels;       /* independent int8 / f32 draws per square step (1..8).  0 = default: 2 channels AND
                               confirm_rounds >= 1 (a false stop needs confirm_rounds + 1 consecutive squares that
                               miss every needed split, each w.p. <= (2/256)^channels: the (2/256)^4 of four
                               channels, at 2 (I + 1) instead of 4 I channel squares for I iterations) */
    int32_t max_iters;      /* 0 = default (10000) */
    int32_t confirm_rounds; /* extra no-change square rounds demanded before stopping (default with channels = 0: 1) */
    int32_t eig_driver;     /* driver of diagonalize: 0 = default (module compression when dim(P) is small against n, else
                               the dense eigensolver: own tridiagonalisation, own tridiagonal divide and conquer, own
                               back-transformation); 4 = dense forced; 5 = dense with rocSOLVER's stedc for the tridiagonal
                               problem; 6 = module compression forced; 1, 2, 3 = rocSOLVER syevd / sytrd + stedc + ormtr /
                               sytrd + steqr + ormtr (comparison only) */
    /* ---- ABI 0.3 (reserved, zero, in 0.2) ---- */
    uint32_t flags;             /* SDPSR_FLAG_* */
    int32_t round_mode;         /* sdpsr_round_mode */
    int32_t basis_image_kernel; /* 0 = by shape, 1 = two-stage (class sums per row), 2 = outer products per class,
                                   3 = sorted chunks with partial sums */
    int32_t refine_path;        /* 0 = by class count (hash tables; beyond 2^18 classes the bucketed grouping), 1 = hash tables
                                   only, 2 = hipCUB radix-sort relabel forced (comparison), 3 = bucketed grouping forced,
                                   4 = as 0 without the one-workgroup-per-CU insert kernel of array sources (comparison), 6 = that kernel
                                   without the workgroups that go first (comparison) */
    int32_t label_bits;         /* 0 = no emulation; 8 / 16 / 32: width of the reference's label type T in
                                   Partition{T} (admissible_subspace defaults to UInt16, src/partitions.jl:84):
                                   SDPSR_LABEL_OVERFLOW where the reference throws InexactError (see below) */
    int32_t insert_wgs_per_cu;  /* measurement knob: resident workgroups per CU of the refinement's insert pass (0 = default) */
    int32_t square_kernel;      /* int8 square of symmetric labels (src/partitions.jl:172): 0 = default (one persistent launch
                                   of 256 x 256 macro-tiles once they fill the chip, 128 x 128 tiles below that), 1 = always
                                   128 x 128 tiles (the launch of ABI 0.3), 64 = always the persistent launch (64-byte K
                                   stages in a ring of four).  Same integers every way. */
    int32_t reserved[3];
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
/* Dimension trajectory of the last sdpsr_admissible_subspace call on ctx -- what the reference logs under
   verbose (src/partitions.jl:150,156,187-188): dims[0] = dim(S) after S = refine!(Part(CL), Part(X0L)),
   dims[k] = dim(S) at the end of iteration k; *count = iterations + 1 (also when capacity is smaller; at most
   `capacity` entries are written).  Host arrays. */
int sdpsr_dimension_trajectory(sdpsr_ctx* ctx, int64_t* dims, int32_t capacity, int32_t* count);

/* ---- AbstractPartition contract (primitives) ------------------------------- */
/* Partition{T}(M::AbstractMatrix) float ctor, src/partitions.jl:24-35: classes of
   bit-equal values, labelled by first occurrence in column-major order, +0.0 -> 0. */
int sdpsr_partition_from_f64(sdpsr_ctx* ctx, int64_t len, const double* M,
                             uint32_t* labels, int64_t* nparts, int mem);
/* Integer ctor + __sort_unique!, src/partitions.jl:37-60. in == out allowed. */
int sdpsr_partition_from_u32(sdpsr_ctx* ctx, int64_t len, const uint32_t* in,
                             uint32_t* labels, int64_t* nparts, int mem);
/* The same for 64-bit integer entries (Partition{T}(M::AbstractMatrix{<:Integer}) is generic in the
   entry type; also the canonical relabel of the hash-combined labels of several restarts):
   0 stays 0, every other key is a class of its own value (told apart by a 64-bit mixing hash,
   collision bound as for sdpsr_refine). */
int sdpsr_partition_from_u64(sdpsr_ctx* ctx, int64_t len, const uint64_t* in,
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
/* _clamp_round!, src/utils.jl:34-53; in place.  BEHAVIOURAL DIFFERENCE a Julia caller will see with
   the default round_mode: the reference TRUNCATES the 7-digit decimal mantissa (unsafe_trunc,
   src/utils.jl:49-53), SDPSR_ROUND_NEAREST rounds it to nearest -- the 7th digit differs on about every
   second input.  What the path outputs are the CLASSES of equal rounded values, and those only differ
   for values that sit on an edge of the rule in use: a truncation edge such as 0.0625 = 0.5 * 2^-3 is
   split into two classes by last-bit noise (esc16j: 10712 classes instead of the pinned 150 with
   NumPy's projection arithmetic); nearest rounding keeps them together, hence the default.
   sdpsr_opts.round_mode = SDPSR_ROUND_TRUNC is the reference's rule bit for bit (this primitive, the
   projection step and the fp64 square of the loop, the setup stage of sdpsr_admissible_subspace_dense). */
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
/* The square step as the int8 loop runs it: `batch` (1..8) SYMMETRIC n x n int8 matrices (the channel matrices of one
   draw, batch-major; symmetry is the caller's word, as the loop has it from the labels), squared exactly in one launch
   that computes the lower-triangle tiles only (sdpsr_opts.square_kernel picks the kernel); X2 receives the full
   matrices, the upper triangles mirrored.  mul!(X2, X, X), src/partitions.jl:172, for symmetric X. */
int sdpsr_square_i8_symmetric(sdpsr_ctx* ctx, int64_t n, int64_t batch, const int8_t* X, int32_t* X2, int mem);
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
   sdpsr_opts.flags & SDPSR_FLAG_SEPARATE_REFINEMENTS restores two refinements per iteration.
   In that joint form the projection half of an iteration is SKIPPED once it can no longer refine anything: when every
   basis matrix U_k is constant on every class of the current partition S (compared on the device through the rounding of
   _clamp_round!, :159-164), x - U U'x of any x that is constant on the classes of S is constant on them too -- U_k in
   span(S) implies U U'x in span(S) -- and every finer partition inherits that; generically it holds right after the first
   projection refinement.  TOLERANCE CAVEAT: "constant" is decided on the per-entry rounded codes of U_k (atol, 7 digits: an
   entry |U_k| < atol counts as 0), while the reference rounds x - U U'x AFTER multiplying those differences by the
   coefficients U_k'x (of the order of |x|): differences of U_k inside a class that are below the rounding resolution can
   still split it there.  So the partition returned is the same up to that rounding -- exactly the same whenever the U_k
   take exactly representable values on the classes (constraint matrices with integer entries: every test problem and
   BASELINE config) --, and SDPSR_FLAG_ALWAYS_PROJECT keeps the projection in every iteration.  The check is made at most
   three times per call. */
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
   BEHAVIOURAL NOTE (commutative algebras, every block 1 x 1): Q_hat from Murota's decomposition spans invariant
   subspaces, 1[P==i] q_k = blks[i][k] q_k, so the images are obtained from the class sums of ONE vector x = sum_k q_k
   (n^2 label reads) instead of all sum_k s_k columns; a second vector with random signs goes through the same pass and
   the two answers must agree to 2e-10, else -- and always with SDPSR_FLAG_FULL_BASIS_IMAGE -- the projection formula
   Q_k' 1[P==i] Q_k is evaluated as the reference does.  Both give Q_k' 1[P==i] Q_k up to rounding (tests: 1e-9).
   blks: d * sum_sq doubles, class-major, then block, each block column-major s_k x s_k.
   Q_hat (optional, may be NULL): n x sum_s column-major, blocks side by side. */
int sdpsr_block_images(sdpsr_ctx* ctx, double* blks, double* Q_hat, double* phase_ms, int mem);
/* ---- one reduction in one call ---------------------------------------------------------------------
   sdpsr_admissible_subspace followed by sdpsr_block_diagonalize on its result and -- when the images fit the
   caller's buffer -- sdpsr_block_images, with the partition staying on the device and no host synchronisation
   between the stages (the separate entry points each return with their outputs complete).  Same arguments and
   meaning as those three; `mem` names the memory space of CL / X0L / U / P_out / blks / Q_hat.
     P_out           n x n labels, may be NULL (the partition is then only kept inside ctx for sdpsr_block_images);
     blks            d * sum_sq doubles if that is <= blks_capacity (in doubles), else untouched: the caller reads
                     *dim_out * *sum_sq, allocates and calls sdpsr_block_images; blks_capacity = 0: sizes only;
     Q_hat           n * sum_s doubles under the same rule with qhat_capacity; may be NULL.
   With device-resident P_out the partition is formed and read in the caller's buffer (no copy); once this call has
   delivered the images itself, a further sdpsr_block_images needs a new sdpsr_block_diagonalize (sizes and Q_hat stay
   available through sdpsr_block_sizes / sdpsr_q_hat).
   Status: that of the first stage that fails.  After SDPSR_NUMERICAL_INCONSISTENCY / SDPSR_DIMENSION_MISMATCH
   (the randomized failures of blockDiagonalize, "try again" in the reference) the partition outputs are valid and
   the caller retries with sdpsr_block_diagonalize on them; SDPSR_NOT_CONVERGED is reported after the other stages ran. */
int sdpsr_jordan_reduce(sdpsr_ctx* ctx, int64_t n, const double* CL, const double* X0L, const double* U, int64_t r,
                        double atol, double epsilon, uint32_t* P_out, int64_t* dim_out, int32_t* iters_out,
                        int32_t* nblocks, int64_t* sum_sq, int64_t* sum_s, double* blks, int64_t blks_capacity,
                        double* Q_hat, int64_t qhat_capacity, double* phase_ms, int mem);

/* R independent random restarts of the same reduction in ONE call on ONE host thread (1 <= R <= 64): restart i runs
   sdpsr_jordan_reduce with its own random streams (seeds[i]; seeds == NULL: the ctx's own stream for restart 0, derived
   seeds for the others), its own HIP stream and its own workspace inside ctx.  While one restart's host side waits for a
   verdict from the device (the reference's loop has one per refinement, src/partitions.jl:154-185), the calling thread
   submits the other restarts' work: on one MI355X two restarts in flight finish 1.4-1.5 x the reductions per second of
   one at a time.  These are the "independent random restarts" of the multi-GPU north-star on the per-GPU side, and the
   reference's own answer to the randomized failures of blockDiagonalize ("try again", src/eigen_decomposition.jl:264-270,
   src/diagonalize.jl:4-9): the caller takes the first restart whose status[i] is SDPSR_OK.
     CL, X0L, U      shared by all restarts (same problem), in memory space `mem`;
     P_out, blks     arrays of R pointers (each as in sdpsr_jordan_reduce; P_out or blks or single entries may be NULL),
                     blks_capacity[R] in doubles; Q_hat is not delivered here (sdpsr_q_hat of a single-restart call);
     dim_out .. sum_s  arrays of R entries (iters_out, nblocks, sum_sq, sum_s may be NULL);
     status[R]       per restart, the status sdpsr_jordan_reduce would have returned.
   Returns SDPSR_OK if every restart did, else the first restart's failure.  A hint given with
   sdpsr_hint_symmetric_basis applies to all R restarts, and so does the ordering of ctx's stream (sdpsr_wait_stream,
   sdpsr_set_stream): every restart's stream starts behind what ctx's stream has been ordered behind.  No threads are created; do not call it from two host threads
   on one ctx. */
int sdpsr_jordan_reduce_batch(sdpsr_ctx* ctx, int32_t R, const uint64_t* seeds, int64_t n, const double* CL, const double* X0L,
                              const double* U, int64_t r, double atol, double epsilon, uint32_t* const* P_out, int64_t* dim_out,
                              int32_t* iters_out, int32_t* nblocks, int64_t* sum_sq, int64_t* sum_s, double* const* blks,
                              const int64_t* blks_capacity, int32_t* status, int mem);
/* (With host arrays, mem = SDPSR_MEM_HOST, C_L / X0_L / U are uploaded ONCE for all R restarts -- into ctx's own input
   buffers -- not once per restart.) */
/* blkSizes (nblocks[restart] ints) of restart `restart` of the last batch call on ctx whose status[restart] was SDPSR_OK or
   SDPSR_DIMENSION_MISMATCH -- sdpsr_block_sizes for a restart. */
int sdpsr_batch_block_sizes(sdpsr_ctx* ctx, int32_t restart, int32_t* blk_sizes);

/* ---- upload once, reduce many times ----------------------------------------------------------------
   The reference's seam hands host arrays to admissible_subspace on every call (src/partitions.jl:109-116); a caller
   who restarts the randomized reduction of ONE problem many times (the reference's "try again",
   src/eigen_decomposition.jl:264-270, the multi-GPU restarts) pays the upload of C_L, X0_L and U -- 3 x 134 MB at
   N = 4096, ~9 ms against a 0.8 ms reduction -- on each of them.  A problem handle holds the three arrays on ctx's
   device: created once (mem names where CL / X0L / U live; device-resident inputs are COPIED, the caller's buffers are
   free on return), used by any number of sdpsr_problem_reduce / sdpsr_problem_reduce_batch calls of ctxs on that
   device, read-only, destroyed by the caller (after the last call that names it has returned).
     hint    bits of sdpsr_hint_symmetric_basis that hold for these inputs (they then apply to every reduction);
     mem_out where P_out / blks / Q_hat live; everything else as in sdpsr_jordan_reduce / sdpsr_jordan_reduce_batch.
   A Julia caller holding AMDGPU.jl arrays passes their device pointers with SDPSR_MEM_DEVICE. */
int sdpsr_problem_create(sdpsr_ctx* ctx, int64_t n, const double* CL, const double* X0L, const double* U, int64_t r, int hint,
                         int mem, sdpsr_problem** out);
int sdpsr_problem_destroy(sdpsr_problem* problem);
int sdpsr_problem_reduce(sdpsr_ctx* ctx, const sdpsr_problem* problem, double atol, double epsilon, uint32_t* P_out,
                         int64_t* dim_out, int32_t* iters_out, int32_t* nblocks, int64_t* sum_sq, int64_t* sum_s, double* blks,
                         int64_t blks_capacity, double* Q_hat, int64_t qhat_capacity, double* phase_ms, int mem_out);
int sdpsr_problem_reduce_batch(sdpsr_ctx* ctx, const sdpsr_problem* problem, int32_t R, const uint64_t* seeds, double atol,
                               double epsilon, uint32_t* const* P_out, int64_t* dim_out, int32_t* iters_out, int32_t* nblocks,
                               int64_t* sum_sq, int64_t* sum_s, double* const* blks, const int64_t* blks_capacity,
                               int32_t* status, int mem_out);
/* Bytes ctx (and the restarts' ctxs inside it) has moved host -> device / device -> host since its creation: what a
   caller of the host-array interface pays on PCIe (tests: a 4-restart batch uploads what one call uploads). */
int sdpsr_transfer_bytes(sdpsr_ctx* ctx, uint64_t* h2d, uint64_t* d2h);

/* ---- blockDiagonalize(P; complex = true), src/compat.jl:26-32,46-68 with T = ComplexF64 ---------
   diagonalize(ComplexF64, P) = desymmetrize (src/diagonalize.jl:26-28) + Murota's decomposition
   over C + check_block_sizes with sum s_k^2 == dim(P) (:13-23); the partition handed to
   basis_image is the desymmetrized one (src/compat.jl:54-57).
   n <= 64: every step in single-workgroup kernels (the reference's own complex tests are 3 x 3
   and 4 x 4, test/runtests.jl:43-57).  64 < n <= 4096: the Hermitian elements go through their real
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

#ifdef __cplusplus
}
#endif
#endif /* SDPSR_H */
