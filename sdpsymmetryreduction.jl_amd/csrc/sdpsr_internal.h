// Internal declarations shared by the host orchestration (api.cpp) and the kernel
// translation units.  Nothing here is part of the ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include <map>
#include <functional>
#include <string>
#include <string_view>
#include <vector>

#include "../../include/sdpsr.h"
#include "sdpsr_hash.h"

namespace sdpsr { struct SytrdGraphCache; }  // kernels_sytrd.hip

// ---------------------------------------------------------------------------
// grow-only named device buffers: no hipMalloc inside steady-state loops
// ---------------------------------------------------------------------------
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

struct sdpsr_ctx {
    int device = 0;
    int num_cus = 256;
    uint64_t seed = 0;
    uint64_t stream_counter = 0;  // fresh RNG stream per randomize call
    sdpsr_opts opts{};
    hipStream_t stream = nullptr;
    hipStream_t side_stream = nullptr;  // lazily created: work that overlaps a one-workgroup kernel
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_wait = nullptr;
    hipEvent_t ev_bt_fork = nullptr, ev_bt_join = nullptr;  // eigen.cpp: first pass of the back-transformation beside the tridiagonal solver
    hipStream_t main_shadow = nullptr;  // the main stream while `stream` temporarily points at side_stream
    bool own_stream = false;
    std::string err;
    double dbg_last_ms = 0;  // SDPSR_DEBUG traces: time of the previous mark
    std::map<std::string, DevBuf, std::less<>> bufs;  // (transparent comparator: looked up by string_view, no allocation per ctx_buf call)
    void* pinned = nullptr;  // small pinned host scratch for scalar read-backs
    size_t pinned_bytes = 0;
    uint32_t* pinned_small = nullptr;  // 256 B pinned: flags read back without their own synchronisation
    void* h2d_ring = nullptr;  // pinned ring for small stream-ordered uploads (no sync per upload)
    int h2d_ring_next = 0;
    void* rocblas = nullptr;  // rocblas_handle, created lazily
    sdpsr::SytrdGraphCache* sytrd_graphs = nullptr;  // captured launch sequences of the tridiagonalisation
    // --- block-diagonalisation state kept between phase 1 and phase 2 ---
    int64_t bd_n = 0, bd_d = 0;
    std::vector<int32_t> bd_sizes;
    int64_t bd_sum_s = 0, bd_sum_sq = 0;
    bool bd_valid = false;
    // complex path (sdpsr_block_diagonalize_complex): its own state, n <= 64
    int64_t bdc_n = 0, bdc_d = 0, bdc_sum_s = 0, bdc_sum_sq = 0;
    std::vector<int32_t> bdc_sizes;
    bool bdc_valid = false;
    uint32_t bd_sym_epoch = 0;            // != 0: "bd_symflag"[0] == epoch <=> bd_sym_labels are NOT symmetric (copy + check pass of blockDiagonalize)
    const uint32_t* bd_sym_labels = nullptr;
    const uint32_t* bd_labels_ext = nullptr;  // sdpsr_jordan_reduce: the labels of the pending phase 2 live in the caller's device buffer, not in "bd_labels"
    const uint32_t* bd_trusted_symmetric = nullptr;  // labels the library made itself and knows to be symmetric (sdpsr_jordan_reduce)
    uint32_t epoch_counter = 0;
    // "ref_first"[l - 1] = first-occurrence index of class l in the label array `first_idx_labels` (written by the
    // ranking pass of the refinement that made those labels); nullptr = not available (sort path, other arrays)
    const uint32_t* first_idx_labels = nullptr;
    int hint_symmetric_basis = 0;         // sdpsr_hint_symmetric_basis: applies to the next admissible_subspace call
    std::vector<int64_t> adm_dims;        // dimension trajectory of the last admissible_subspace call (sdpsr_dimension_trajectory)
    bool bd_q_valid = false;  // "bd_qhat" holds Q_hat of the last diagonalize (even when check_block_sizes failed)
    bool bd_labels_owned = false;
    // hash table capacity hint (log2) for the next refine
    int table_log2_hint = 12;
    hipEvent_t ev[2 * SDPSR_T_COUNT] = {};
    // sdpsr_jordan_reduce_batch: this ctx runs one restart on a fiber of the calling thread; a host wait for one of its
    // streams polls and hands the thread to the other restarts (ctx_sync_stream).  nullptr outside a batch call.
    void (*yield_fn)(void*) = nullptr;
    void* yield_arg = nullptr;
    std::vector<sdpsr_ctx*> batch_children;  // the ctxs of restarts 1 .. R - 1 (created on demand, destroyed with this ctx)
    uint64_t h2d_bytes = 0, d2h_bytes = 0;   // bytes this ctx has moved over PCIe (sdpsr_transfer_bytes)
    // the loop's branch predictor: the previous admissible_subspace call on this ctx (order predict_n) found its input closed in
    // the first iteration -- the next call of that order runs the confirm round speculatively behind the first verify pass
    bool predict_closed = false;
    int64_t predict_n = 0;
    uint64_t host_waits = 0;                 // host waits for one of this ctx's streams (ctx_sync_stream); sdpsr_profile_host_waits
    uint32_t report_seq = 0;                 // stamps of the label passes' reports to pinned memory (ctx_wait_word)
    // sdpsr_jordan_reduce: the loop may leave the verdicts of its LAST verify passes (first iteration of an input predicted
    // closed: the verify pass and the speculative confirm round) unread and return "converged"; the reduction goes on in stream
    // order and reads them behind its next host waits (reduce.cpp) -- one host wait less per reduction.  Violated verdicts
    // discard everything and the reduction is repeated without the guess.
    bool allow_deferred_verdict = false;
    const volatile uint32_t* deferred_verdict = nullptr;  // words [0] and [16] must be 0
};

// sdpsr_problem_create: the loop's inputs, device-resident, shared (read-only) by every reduction / restart that names them
struct sdpsr_problem {
    int device = 0;
    int64_t n = 0, r = 0;
    double *CL = nullptr, *X0 = nullptr, *U = nullptr;  // hipMalloc'ed, owned
    int hint = 0;  // sdpsr_hint_symmetric_basis bits that hold for these inputs
};

// Host wait for a stream of ctx c (every wait of the library goes through here).  Inside a batch call the wait polls and
// hands the thread to the other restarts' fibers.  hipStreamQuery records its "not ready" as the thread's last error, and
// all fibers share that slot: exactly that value is dropped on every way out (a fiber whose first query succeeds must not
// inherit another fiber's "not ready" either), anything else -- a failed launch of this restart, recorded before the
// wait -- is handed to the caller as the wait's result instead of being cleared with it.
inline hipError_t ctx_sync_stream(sdpsr_ctx* c, hipStream_t s) {
    if (c) ++c->host_waits;
    if (!c || !c->yield_fn) return hipStreamSynchronize(s);
    const hipError_t before = hipPeekAtLastError();
    for (;;) {
        const hipError_t e = hipStreamQuery(s);
        if (e != hipErrorNotReady) {
            if (hipPeekAtLastError() == hipErrorNotReady) (void)hipGetLastError();
            if (e == hipSuccess && before != hipSuccess && before != hipErrorNotReady) return before;
            return e;
        }
        c->yield_fn(c->yield_arg);
    }
}

// Wait for a REPORT instead of for the stream: a kernel has been told to store its result words into pinned host memory and
// then the stamp `seq` into *w (system-scope fence in between).  The label pass of a refinement knows its counters when it
// STARTS -- the ranking kernels before it produced them -- so the host has them ~15 us before the pass ends and enqueues the
// next kernels meanwhile (they follow in stream order).  Only for callers whose own return is covered by a later wait for
// the stream (the loop of admissible_subspace): the stream is NOT idle when this returns.  The stream is still asked now
// and then: a failed launch or a kernel that never reports must not hang the host.
inline hipError_t ctx_wait_word(sdpsr_ctx* c, hipStream_t s, const volatile uint32_t* w, uint32_t seq) {
    ++c->host_waits;
    const hipError_t before = hipPeekAtLastError();
    bool queried = false;
    hipError_t res = hipSuccess;
    for (unsigned spins = 1;; ++spins) {
        if (*w == seq) break;
        if (c->yield_fn || (spins & 2047u) == 0) {
            queried = true;
            const hipError_t e = hipStreamQuery(s);
            if (e != hipErrorNotReady) {  // drained (or failed): the report is there, or will never be
                res = e != hipSuccess ? e : (*w == seq ? hipSuccess : hipErrorUnknown);
                break;
            }
            if (c->yield_fn) c->yield_fn(c->yield_arg);
        } else {
            __builtin_ia32_pause();
        }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    if (queried && hipPeekAtLastError() == hipErrorNotReady) (void)hipGetLastError();
    if (res == hipSuccess && before != hipSuccess && before != hipErrorNotReady) return before;
    return res;
}

void* ctx_buf(sdpsr_ctx* c, const char* name, size_t bytes);  // throws std::bad_alloc-like via status
int ctx_fail(sdpsr_ctx* c, int status, const std::string& msg);

#define HIP_TRY(c, expr)                                                              \
    do {                                                                              \
        hipError_t _e = (expr);                                                       \
        if (_e != hipSuccess)                                                         \
            return ctx_fail((c), SDPSR_HIP_ERROR,                                     \
                            std::string(#expr) + ": " + hipGetErrorString(_e));       \
    } while (0)

// ---------------------------------------------------------------------------
// kernels_partition.hip
// ---------------------------------------------------------------------------
namespace sdpsr {

// M[e] = values[L[e]-1] (0 -> 0.0)
void launch_fill_f64(hipStream_t s, int64_t len, const uint32_t* L, const double* values, int64_t d,
                     double* M, uint32_t* bad_flag);
// M[e] = uniform(key, L[e])
void launch_randomize_f64(hipStream_t s, int64_t len, const uint32_t* L, uint64_t key, double* M);
// in-place clamp+round
void launch_clamp_round(hipStream_t s, int64_t len, double* a, double atol, double scale);
// int8 / f32 channel matrices, padded to ld (>= n, multiple of 16), zero in the padding.
// X[t] is at X + t*ld*ld.
void launch_gather_i8(hipStream_t s, int64_t n, int64_t ld, int T, const uint32_t* L,
                      uint64_t key, int8_t* X, int64_t dmax = 0);
void launch_gather_f32(hipStream_t s, int64_t n, int64_t ld, int T, int vmax, const uint32_t* L,
                       uint64_t key, float* X);
void launch_gather_f64_padded(hipStream_t s, int64_t n, int64_t ld, const uint32_t* L,
                              uint64_t key, double* X);
// dense copy with padding: dst[ld x ld] <- src[n x n], zero padding (templated by bytes)
void launch_unpad_mirror_lower_i32(hipStream_t s, int64_t n, int64_t ld, const int32_t* src, int32_t* dst);
void launch_pad_copy(hipStream_t s, int64_t n, int64_t ld, const void* src, void* dst,
                     int elem_bytes);
void launch_unpad_copy(hipStream_t s, int64_t n, int64_t ld, const void* src, void* dst,
                       int elem_bytes);

// projection: coef[k] = sum_e U[e,k] * x[e] with x[e] = uniform(key, L[e]) (Lx == nullptr)
// or x[e] = xin[e].  partial: r * nblk doubles scratch.
void launch_proj_coef(hipStream_t s, int64_t len, int64_t r, const double* U, const uint32_t* L,
                      uint64_t key, const double* xin, double* partial, int nblk, double* coef);
// y[e] = x[e] - sum_k U[e,k] coef[k]; optional outputs: yout (rounded value, fp64),
// sig (signature chained on L).  x as above.
void launch_gather_i8_sym_packed(hipStream_t s, int64_t n, int64_t ld, int T, const uint32_t* Lp, uint64_t key, int8_t* X,
                                 int64_t dmax);
void launch_proj_coef_lower(hipStream_t s, int64_t n, int64_t r, const double* U, const uint32_t* L, int lab_packed,
                            uint64_t key, double* partial, int nblk, double* coef);
void launch_proj_coef_probe(hipStream_t s, int64_t len, int64_t n, int64_t r, const double* U, const uint32_t* L, uint64_t key,
                            double* partial, int nblk, double* coef);
void launch_proj_apply_lower(hipStream_t s, int64_t n, int64_t r, const double* U, const uint32_t* L, int lab_packed, uint64_t key,
                             const double* coef, double atol, double scale, uint64_t* sig);
void launch_proj_apply(hipStream_t s, int64_t len, int64_t r, const double* U, const uint32_t* L,
                       uint64_t key, const double* xin, const double* coef, double atol,
                       double scale, int do_round, double* yout, uint64_t* sig);

// signatures.  sig = 0 <=> (L == 0 and key == 0).  L may be nullptr (all zero labels).
void launch_sig_f64(hipStream_t s, int64_t len, const uint32_t* L, const double* v, uint64_t* sig);
void launch_sig_f64_pair(hipStream_t s, int64_t len, const double* a, const double* b, uint64_t* sig);
void launch_sig_f64_rounded(hipStream_t s, int64_t n, int64_t ld, const uint32_t* L,
                            const double* v, double atol, double scale, uint64_t* sig);
void launch_sig_u32(hipStream_t s, int64_t len, const uint32_t* L, const uint32_t* k,
                    uint64_t* sig);
void launch_sig_u64(hipStream_t s, int64_t len, const uint64_t* k, uint64_t* sig);
// T channels of int32 / f32 squares, padded ld, C[t] at C + t*ld*ld
void launch_sig_i32(hipStream_t s, int64_t n, int64_t ld, int T, const uint32_t* L,
                    const int32_t* C, uint64_t* sig, const uint32_t* nonsym_flag = nullptr, int packed = 0, int lab_packed = 0);
void launch_sig_f32(hipStream_t s, int64_t n, int64_t ld, int T, const uint32_t* L,
                    const float* C, uint64_t* sig, const uint32_t* nonsym_flag = nullptr, int packed = 0, int lab_packed = 0);

// canonical relabel of signatures (hash table + first-occurrence ranking).
// Workspace layout is owned by the caller (see refine_workspace_bytes).
// slot of the refinement's global table: signature and first index side by side -- ONE 16-byte gather per look-up (a gather that
// misses the L1 costs ~5 clocks per lane on this part: signature and minimum in separate arrays were two of them per entry)
struct RefSlot {
    unsigned long long sig;
    uint32_t min;
    uint32_t pad;
};
struct RefineWs {
    RefSlot* tab;        // cap
    uint32_t* tab_lab;   // cap
    uint32_t* blk_cnt;   // nblk + 1
    int insert_wgs_per_cu = 0;  // sdpsr_opts.insert_wgs_per_cu (0 = default)
    uint32_t* host_counters = nullptr;  // pinned host memory: counters[0..2] are also stored there by the label pass (plain label pass only)
    uint32_t host_seq = 0;              // ... and then this stamp into word 3: the host may read the three words once it sees the stamp
    int mid = 0;           // array source, ~1000 .. 6000 classes predicted: the one-workgroup-per-CU LDS table (refine_insert_mid_kernel)
    int expect_small = 0;  // host prediction: <= refine_small_k() classes (see launch_refine)
    uint32_t* first_idx = nullptr;  // optional: first-occurrence index of class l at [l - 1], l <= refine_first_cap()
    void* rank_ws = nullptr;        // refine_rank_slots_workspace_bytes(len): the ranking of more than refine_small_k() classes
    size_t rank_ws_bytes = 0;
    uint32_t* counters;  // [0] = inserted, [1] = overflow flag, [2] = nparts, [16..] slot list (refine_counters_bytes())
    int log2cap;
    int nblk;
};
size_t refine_block_entries();
size_t refine_counters_bytes();
uint32_t refine_small_k();
// Where the insert pass takes its signatures from: an array (SIG_ARRAY), or computed on the fly
// from the data the matching signature kernel reads (the signatures never travel through HBM).
// `sig` is the array (SIG_ARRAY) or len entries of scratch for the paths that need one (sort).
enum { SIG_ARRAY = 0, SIG_PAIR = 1, SIG_PROJ = 2, SIG_CHAN_I32 = 3, SIG_CHAN_F32 = 4, SIG_JOINT_I32 = 5 };  // JOINT: SIG_PROJ and SIG_CHAN_I32 fields together (packed)
struct SigSource {
    int kind = SIG_ARRAY;
    uint64_t* sig = nullptr;
    const double *a = nullptr, *b = nullptr;                      // SIG_PAIR (launch_sig_f64_pair)
    const double *U = nullptr, *coef = nullptr;                   // SIG_PROJ (launch_proj_apply, rounded, sig only)
    const uint32_t* L = nullptr;                                  // old labels (SIG_PROJ, SIG_CHAN_*)
    int r = 0;
    uint64_t key = 0;
    double atol = 0, scale = 1;
    int64_t n = 0, ld = 0;                                        // SIG_CHAN_* (launch_sig_i32 / launch_sig_f32)
    int T = 0;
    const void* C = nullptr;
    int packed = 0;                                               // lower triangle only, densely packed (symmetric labels; SIG_PROJ: and symmetric basis, needs n)
    int lab_packed = 0;                                           // (with packed) L is the packed lower triangle itself: label of packed entry e = L[e]
    const uint32_t* zero_flag = nullptr;                          // device constant 0 when packed (the kernels' "lower" flag)
};
bool sig_source_fusable(const SigSource& q);
uint32_t refine_first_cap();
size_t verify_ref_bytes(int64_t d);
size_t uconst_ref_bytes(int64_t d, int64_t r);
bool launch_basis_constant_on_classes(hipStream_t s, int64_t n, int64_t r, const double* U, const uint32_t* Lp, int64_t d,
                                      const uint32_t* first_idx, double atol, double scale, void* ref, uint32_t* flag);
bool launch_verify_no_split(hipStream_t s, const SigSource& q, int64_t d, const uint32_t* first_idx, void* ref, uint32_t* flag);
void launch_sig_materialize(hipStream_t s, int64_t len, const SigSource& q, uint64_t* sig);
// slot: len entries of scratch; labels_out may alias q.L (it is written only by a pass that succeeded)
// sym_n > 0 (and len == sym_n^2): the label pass also checks the new labels for symmetry, counters[3] = 1 if NOT symmetric
bool refine_mid_supports(const SigSource& q);  // RefineWs::mid is honoured for this source
void launch_refine(hipStream_t s, int64_t len, const SigSource& q, uint32_t* slot, uint32_t* labels_out,
                   const RefineWs& ws, int64_t sym_n = 0);

// kernels_refine_sort.hip: radix-sort relabel for the many-classes regime
size_t refine_bucketed_workspace_bytes(int64_t len);
bool refine_bucket_set_device_attributes();
bool launch_refine_bucketed(hipStream_t s, int64_t len, const uint64_t* sig, uint32_t* labels_out, void* ws, size_t ws_bytes,
                            uint32_t* counters, uint32_t* first_idx, uint32_t first_cap, uint32_t* host_counters = nullptr, uint32_t host_seq = 0);
size_t refine_rank_slots_workspace_bytes(int64_t len);
bool launch_rank_slots(hipStream_t s, int64_t len, int64_t cap, const RefSlot* tab, uint32_t* tab_lab,
                       uint32_t* counters, uint32_t small_k, uint32_t* first_idx, uint32_t first_cap, void* ws, size_t ws_bytes);
// distinct-signature estimate of a signature array from <= 65536 sampled entries; host_out (pinned, 4 words): non-zero
// entries sampled, distinct signatures among them, signatures seen once, seen twice.  Returns the sample size (0: failed)
size_t refine_sample_workspace_bytes();
int64_t launch_refine_sample(hipStream_t s, int64_t len, const uint64_t* sig, void* ws, uint32_t* host_out);
size_t refine_sorted_workspace_bytes(int64_t len);
bool launch_refine_sorted(hipStream_t s, int64_t len, const uint64_t* sig, uint32_t* labels_out, void* ws, size_t ws_bytes,
                          uint32_t* counters);

void launch_max_pair_code(hipStream_t s, int64_t len, const uint32_t* p1, const uint32_t* p2, uint64_t d1, uint64_t* out);
void launch_transpose_labels(hipStream_t s, int64_t n, const uint32_t* L, uint32_t* Lt);
void launch_labels_checksum(hipStream_t s, int64_t len, const uint32_t* L, uint64_t* partial, uint64_t* out);
int64_t reduce_columns_chunk(int64_t len, int64_t m, int64_t d);
bool launch_reduce_columns(hipStream_t s, int64_t len, int64_t m, int64_t d, const uint32_t* L, const double* A,
                           double* partial, double* out);
// kernels_module.hip / setup-stage helpers used across api.cpp
void launch_symmetrize(hipStream_t s, int64_t m, int64_t ld, double* B);
void launch_tall_times_small(hipStream_t s, int64_t n, int64_t ldi, const double* In, int kk, const double* S,
                             int lds_, int ncols, double alpha, double beta, double* out, int64_t ldo);
void launch_transpose_rows(hipStream_t s, int64_t len, int64_t m, const double* A, double* R);
void launch_col_norms2(hipStream_t s, int64_t len, int64_t k, const double* V, double* partial, int nblk, double* out);
void launch_scale_copy(hipStream_t s, int64_t len, const double* v, double alpha, double* out);
void launch_rank1_update(hipStream_t s, int64_t len, int64_t m, double* R, const double* u, const double* dots);
void launch_sub_round(hipStream_t s, int64_t len, const double* a, const double* b, double atol, double scale, double* out);

// Kernels with more than 64 KiB of dynamic LDS carry a per-DEVICE attribute: sdpsr_create() sets
// them all with the ctx's device current (no process-global "already set" flags) and fails when one
// attribute call does (false: a later launch would otherwise fail with an opaque error).
bool gemm_set_device_attributes();
bool blockdiag_set_device_attributes();
bool module_set_device_attributes();
bool partition_set_device_attributes();
bool sytrd_set_device_attributes();
bool small_syev_set_device_attributes();
bool stedc_set_device_attributes();
bool batched_set_device_attributes();
bool backtransform_set_device_attributes();
bool complex_set_device_attributes();
// kernels_complex.hip (complex path of blockDiagonalize, n <= 64; planes re / im, ld = n)
void launch_cx_embed(hipStream_t s, int64_t n, const double* Hr, const double* Hi, int64_t ld2, double* M);
void launch_cx_rot(hipStream_t s, int64_t n, int64_t cols, int64_t ld2, const double* E, double* R);
void launch_cx_zero_pad(hipStream_t s, int64_t m, int64_t mc, int64_t ld, int64_t ldc, double* A);
void launch_cx_combine(hipStream_t s, int64_t n, int64_t ld2, const double* E, const int32_t* desc, const double* coef, double* Vr,
                       double* Vi);
void launch_cx_stack(hipStream_t s, int64_t n, int64_t cols, const double* Vr, const double* Vi, int64_t ldv, int64_t ld2, double* E);
void launch_cx_block_norms_general(hipStream_t s, int64_t n, int64_t ldn, const double* Gr, const double* Gi, const int32_t* space_of,
                                   int neig, unsigned long long* norms);
void launch_cx_irreducible_general(hipStream_t s, int64_t n, const double* Hr, const double* Hi, const double* Vr, const double* Vi,
                                   const int32_t* desc, int ncols, double atol, double* Qhat);
void launch_cx_basis_image_sorted(hipStream_t s, int64_t n, int64_t d, int64_t S, const uint32_t* ent, const int64_t* cls_ptr,
                                  const double* Qhat, const int32_t* descA, const int32_t* descB, double atol, double* out);
void launch_cx_gather_herm(hipStream_t s, int64_t n, const uint32_t* L, uint64_t key, double* Hr, double* Hi);
void launch_cx_heev(hipStream_t s, int64_t n, const double* Hr, const double* Hi, double* w, double* Vr, double* Vi, int* info);
void launch_cx_block_norms(hipStream_t s, int64_t n, const double* Hr, const double* Hi, const double* Vr, const double* Vi,
                           const int32_t* space_of, int neig, unsigned long long* norms);
void launch_cx_irreducible(hipStream_t s, int64_t n, const double* Hr, const double* Hi, const double* Vr, const double* Vi,
                           const int32_t* desc, int ncols, double atol, double* Qhat);
void launch_cx_basis_image(hipStream_t s, int64_t n, int64_t d, int64_t S, const uint32_t* L, const double* Qhat,
                           const int32_t* descA, const int32_t* descB, double atol, double* out);
// kernels_batched.hip: `count` runs of eigen_decomposition on one partition of order n <= 64
void launch_eigdec_batched64(hipStream_t s, int64_t n, int64_t d, int64_t count, const uint32_t* L, const double* values,
                             uint64_t seed, uint64_t stream_base, double atol, int32_t* status, int32_t* neig,
                             int32_t* nclasses, int num_cus);
// kernels_sytrd_look.hip: the panel form with one launch per column (the launches are emitted by kernels_sytrd.hip)
struct SytrdLookArgs {
    double* A;  // n x n, leading dimension ld (multiple of 128, <= 8192), lower triangle referenced
    int64_t ld;
    int n;
    double* PT;        // ld x 64 row-major panel [V | W] (the layout of the trailing update's MFMA kernel)
    double* PTc;       // 64 x ld: the same panel column-major (the layout a row's owner thread reads)
    double* P;         // [2][ld / 128][ld]: partial products of the unnormalised column, one slot per partner block
    double* acol;      // [2][ld]: the updated column of the launch
    double* rec_norm;  // [2][ld / 128]: partial |a|^2 of the rows of a block
    double* rec_dots;  // [2][ld / 128][64]: partial [V'a | W'a]
    double* rec_vaz;   // [2][vaz_cap]: partial a'(A0 a), two per tile
    int64_t vaz_cap;
    double *d, *e, *tau;
};
const void* sytrd_look_kernel_fn(int mode, int64_t ld);
void sytrd_graph_cache_destroy(SytrdGraphCache* g);
void sytrd_graph_cache_stats(const SytrdGraphCache* g, uint64_t* hits, uint64_t* misses, double* instantiate_ms);

// symmetric-labels check: flag[0] = 1 if some L[i,j] != L[j,i]
void launch_check_symmetric(hipStream_t s, int64_t n, const uint32_t* L, uint32_t* flag);
void launch_copy_check_symmetric(hipStream_t s, int64_t n, const uint32_t* src, uint32_t* dst, uint32_t* flag, uint32_t epoch);

// ---------------------------------------------------------------------------
// kernels_gemm.hip:  C = A' * B (column-major), MFMA tiles staged through LDS.
// Operands must be padded: k multiple of KT, m and n multiples of 128, pointers 16-B
// aligned, lda/ldb multiples of 16 bytes.
// ---------------------------------------------------------------------------
// num_cus > 0: the persistent 256 x 256 launch (kernels_gemm_sym.hip) when n is a multiple of 256; variant =
// sdpsr_opts.square_kernel (0 = by size, 1 = 128 x 128 tiles of kernels_gemm.hip, 64 = persistent forced)
void launch_gemm_tn_i8_sym(hipStream_t s, int64_t n, int64_t k, const int8_t* X, int64_t ldx, int32_t* C, int64_t ldc,
                           int batch, int64_t strideX, int64_t strideC, const uint32_t* nonsym_flag, int num_cus = 0,
                           int variant = 0);
bool launch_i8_symsquare(hipStream_t s, int64_t n, int64_t k, const int8_t* X, int64_t ldx, int32_t* C, int64_t ldc, int batch,
                         int64_t strideX, int64_t strideC, const uint32_t* nonsym_flag, int num_cus, int variant);
bool i8_symsquare_pays(int64_t n, int T, int num_cus);
bool gemm_sym_set_device_attributes();
void launch_gemm_tn_f32_sym(hipStream_t s, int64_t n, int64_t k, const float* X, int64_t ldx, float* C, int64_t ldc,
                            int batch, int64_t strideX, int64_t strideC, const uint32_t* nonsym_flag);
void launch_unpack_symmetric_labels(hipStream_t s, int64_t n, const uint32_t* Lp, uint32_t* L);
void launch_gemm_tn_i8(hipStream_t s, int64_t m, int64_t n, int64_t k, const int8_t* A,
                       int64_t lda, const int8_t* B, int64_t ldb, int32_t* C, int64_t ldc,
                       int batch, int64_t strideA, int64_t strideB, int64_t strideC);
void launch_gemm_tn_f32(hipStream_t s, int64_t m, int64_t n, int64_t k, const float* A,
                        int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc,
                        int batch, int64_t strideA, int64_t strideB, int64_t strideC);
void launch_gemm_tn_f64(hipStream_t s, int64_t m, int64_t n, int64_t k, const double* A,
                        int64_t lda, const double* B, int64_t ldb, double* C, int64_t ldc,
                        int batch, int64_t strideA, int64_t strideB, int64_t strideC);

// ---------------------------------------------------------------------------
// kernels_blockdiag.hip
// ---------------------------------------------------------------------------
// norms[bi*neig+bj] = max |M[i,j]| over the (bi,bj) eigenspace block; space_of[n] maps an
// index to its eigenspace.  norms must be zeroed (as uint64 bit patterns of doubles).
void launch_small_qtaq_block_norms(hipStream_t s, int64_t n, int64_t ld, const double* A, const double* Q,
                                   const int32_t* space_of, int neig, unsigned long long* norms, double* Mout, double* Tout = nullptr);
// the same with the eigenvalue clustering on the device (|dv| > atol starts a new eigenspace) and everything the
// host needs next in one packed device buffer: [status, sweeps, -, -, eigenspaces | 64 B][space_of n | padded to
// 64 B][values n][norms neig * neig]  (einfo: the eigensolver's "eig_info")
size_t small_cluster_pack_bytes(int64_t n);
void launch_small_cluster_qtaq_block_norms(hipStream_t s, int64_t n, int64_t ld, const double* A, const double* Q,
                                           const double* evals, double atol, const int* einfo, char* pack, double* Tout);
void launch_coupling_symmetrize_minmax(hipStream_t s, int neig, unsigned long long* norms, const int32_t* dims, unsigned long long* stat);
void launch_coupling_count(hipStream_t s, int neig, const unsigned long long* norms, const double* edges17, unsigned long long* stat);
void launch_coupling_bits(hipStream_t s, int neig, const unsigned long long* norms, double thr, unsigned long long* bits);
void launch_block_norms(hipStream_t s, int64_t n, int64_t ld, const double* M,
                        const int32_t* space_of, int neig, unsigned long long* norms);
// y = A * x for symmetric A (n x n, ld) and nv vectors (columns of X, ldx): Y[:,v]
void launch_symm_multi_gemv(hipStream_t s, int64_t n, int64_t ld, const double* A,
                            const double* X, int64_t ldx, int nv, double* Y, int64_t ldy);
// out[j] = sum_i Q[i, col0 + j] * a[i], j < m   (Q' a over a column range)
void launch_gemv_t(hipStream_t s, int64_t n, int64_t ld, const double* Q, int64_t col0,
                   int64_t m, const double* a, double* out);
// dst[i] (+)= alpha_dev[0] * sum_j Q[i, col0+j] * w[j]
void launch_gemv_n_scaled(hipStream_t s, int64_t n, int64_t ld, const double* Q, int64_t col0,
                          int64_t m, const double* w, const double* inv_norm, double* dst);
// copy column / clamp
void launch_copy_col(hipStream_t s, int64_t n, const double* src, double* dst);
void launch_irreducible_pairs(hipStream_t s, int64_t n, int64_t ld, const double* Q, const double* Bf, int npairs,
                              int max_m2, const int32_t* desc, double* Qhat);
void launch_copy_cols(hipStream_t s, int64_t n, int64_t count, const int32_t* src_cols, const int32_t* dst_cols,
                      const double* src, int64_t ld_src, double* dst, int64_t ld_dst);
void launch_clamptol(hipStream_t s, int64_t len, double* a, double atol);
// norm2 of a vector -> out[0] = 1/||v||
void launch_inv_norm(hipStream_t s, int64_t m, const double* v, double* out);

// basis image.  Qrm: row-major n x S1 (Q_hat rows contiguous).  entries sorted by class:
// ent[e] = linear index, class_ptr[d+1].  desc: per output (colA, colB) pairs (S of them).
void launch_basis_image(hipStream_t s, int64_t n, int64_t d, int64_t S1, int64_t S,
                        const double* Qrm, const uint32_t* ent, const int64_t* class_ptr,
                        const int32_t* descA, const int32_t* descB, const int64_t* chunk_ptr,
                        int64_t nchunks_total, const int32_t* chunk_class,
                        const int64_t* chunk_begin, const int64_t* chunk_end, double* partial,
                        double* out, double atol);
bool basis_image_two_stage_fits(int64_t n, int64_t d, int64_t S1);
void launch_basis_image_outer(hipStream_t s, int64_t n, int64_t d, int64_t S1, int64_t S, int nblocks, int max_s,
                              const double* Qrm, const uint32_t* ent, const int64_t* cls_ptr,
                              const int32_t* blk_col, const int32_t* blk_size, const int64_t* blk_off, double atol,
                              double* out);
// out[r + i*n] = sum over c with L[c + r*n] == i+1 of x[c]   (class sums of a vector, i < d)
bool class_sums_supports(int64_t n, int64_t d, int64_t ldo);
void launch_class_sums(hipStream_t s, int64_t n, int64_t d, const uint32_t* L, const double* x, double* out, int64_t ldo);
void launch_basis_image_two_stage(hipStream_t s, int64_t n, int64_t d, int64_t S1, int64_t S,
                                  const uint32_t* L, const double* Qrm, double* T, const int32_t* colA,
                                  const int32_t* colB, double atol, double* out);
size_t basis_image_commutative_workspace_doubles(int64_t n, int64_t d);
size_t basis_image_blocks_workspace_doubles(int64_t n, int64_t d);
bool launch_basis_image_blocks(hipStream_t s, int64_t n, int64_t d, int64_t S1, int64_t S, int nblocks, const int32_t* blk_col, const int32_t* blk_size,
                               const int64_t* blk_off, const uint32_t* L, const double* Qrm, uint64_t key, int only, double atol, double tol, double* ws,
                               double* out, uint32_t* flag);
bool launch_basis_image_fix_pair(hipStream_t s, int64_t n, int64_t d, int64_t S1, const uint32_t* L, const double* Qrm, int k1, int k2,
                                 double atol, double* ws, double* out);
bool launch_basis_image_commutative(hipStream_t s, int64_t n, int64_t d, int64_t S1, const uint32_t* L, const double* Qrm, uint64_t key,
                                    double atol, double tol, double* ws, double* out, uint32_t* flag);
void launch_transpose_to_rowmajor(hipStream_t s, int64_t n, int64_t S1, const double* Qcm,
                                  double* Qrm);
// stable sort of entries by label (label 0 dropped): ent sorted, hist[d+1]
int sort_entries_by_label(sdpsr_ctx* c, int64_t len, int64_t d, const uint32_t* L,
                          uint32_t** ent_out, int64_t** class_ptr_host);

// ---------------------------------------------------------------------------
// eigen.cpp (rocSOLVER)
// ---------------------------------------------------------------------------
// A (n x n, leading dimension lda) is overwritten with the eigenvectors; w[n] ascending.
// host_w (optional): the eigenvalues are also delivered to the host, riding on the status read-back
// after_launch: called once the eigensolver's kernels are enqueued, before the read-back synchronises
// defer_readback: return right after the launches; the status stays in ctx buffer "eig_info" for the caller's next read-back
int syev_device(sdpsr_ctx* c, int64_t n, double* A, int64_t lda, double* w, double* host_w = nullptr,
                const std::function<void()>* after_launch = nullptr, bool defer_readback = false);
void* ctx_pinned(sdpsr_ctx* c, size_t bytes);
void destroy_handle(sdpsr_ctx* c);

}  // namespace sdpsr
