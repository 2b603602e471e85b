// Symmetric eigendecomposition of the generic algebra element, eigen(A) at
// src/eigen_decomposition.jl:246 (LAPACK dsyevr in the reference: all eigenpairs, ascending).
//
// Round-1 driver: rocSOLVER in three explicit phases so that each can be timed and the
// tridiagonalisation can be replaced by the hand-written HIP panel kernel without touching
// the callers:   sytrd (A = Q T Q')  ->  stedc (T = Z D Z')  ->  ormtr (V = Q Z).
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>

#include "host_internal.h"

namespace sdpsr {

size_t sytrd_workspace_doubles(int64_t n, int64_t ld);
void launch_sytrd(sdpsr_ctx* c, int64_t n, double* A, int64_t ld, double* d, double* e, double* tau, double* ws);
bool launch_small_syev(hipStream_t s, int64_t n, double* A, int64_t lda, double* w, double* Vtmp, int* info);
void launch_splitk_reduce(hipStream_t s, int64_t len, int Z, int64_t stride, const double* P, double* C);
void launch_bt_extract_panel(hipStream_t s, int64_t n, int64_t ld, const double* A, int64_t j0, int64_t r0, double* Vp,
                             double* VpT);
void launch_bt_larft(hipStream_t s, const double* G, const double* tau, int64_t nblk, int64_t n, double* T);
void launch_gemm_tn_f64_sub(hipStream_t s, int64_t m, int64_t n, int64_t k, const double* A, int64_t lda, const double* B,
                            int64_t ldb, double* C, int64_t ldc);
// kernels_stedc.hip: divide and conquer for the tridiagonal problem
size_t stedc_workspace_bytes(int64_t ld);
size_t stedc_descriptors(int64_t ld, std::vector<int>& out);
void* stedc_descriptor_slot(void* ws, int64_t ld);
bool launch_stedc(hipStream_t s, int64_t n, int64_t ld, const double* d, const double* e, double* w, double* Z, double* W1, double* W2,
                  void* ws);
void launch_stedc_check(hipStream_t s, int64_t n, const double* a, const double* b, int pre, int* info);

// C (m x n, dense: ldc == m) = A' B with the K range split over workgroups when the output alone
// would leave most CUs idle; partial tiles are summed in fixed order.
static int bt_gemm_splitk(sdpsr_ctx* c, int64_t m, int64_t n, int64_t k, const double* A, int64_t lda, const double* B,
                          int64_t ldb, double* C) {
    const int64_t tiles = (m / 128) * (n / 128);
    // K is split over Z workgroups per output tile: the largest divisor of the K-tile count that keeps
    // >= 128 of K per workgroup and the launch within ~one workgroup per CU (any divisor, not only
    // powers of two: ld = 4224 = 33 * 128 at N = 4104 has 264 = 8 * 33 K-tiles)
    int Z = 1;
    {
        const int64_t kt = k / 16;
        for (int64_t z = 1; z <= kt && tiles * z <= 256; ++z)
            if (kt % z == 0 && k / z >= 128) Z = (int)z;
    }
    if (Z == 1) {
        launch_gemm_tn_f64(c->stream, m, n, k, A, lda, B, ldb, C, m, 1, 0, 0, 0);
        return SDPSR_OK;
    }
    double* P = (double*)ctx_buf(c, "bt_partials", (size_t)Z * m * n * 8);
    if (!P) return SDPSR_OUT_OF_MEMORY;
    const int64_t kz = k / Z;
    launch_gemm_tn_f64(c->stream, m, n, kz, A, lda, B, ldb, P, m, Z, kz, kz, m * n);
    launch_splitk_reduce(c->stream, m * n, Z, m * n, P, C);
    return SDPSR_OK;
}

// Z <- Q Z with Q = H_0 ... H_{n-2} from the tridiagonalisation (reflectors below the subdiagonal of
// A, tau): compact-WY blocks of 128 reflectors, last block first, every product on the fp64
// matrix cores (kernels_backtransform.hip has the plan).  Z: ld x ld, zero padded.
// Two halves: what depends on the reflectors only -- the panels V_b, their Gram matrices, every T_b, X_b = T_b' V_b' --
// is launched by backtransform_prepare and runs on the side stream BESIDE the tridiagonal solver (which touches neither
// A nor these buffers); backtransform_apply then needs two products per block, W = V_b' Z and Z -= X_b' W.
static int backtransform_prepare(sdpsr_ctx* c, int64_t n, const double* A, int64_t ld, const double* tau) {
    hipStream_t s = c->stream;
    const int64_t nblk = (n - 1 + 127) / 128;
    const size_t per = (size_t)ld * 128 * 8;
    double* VpAll = (double*)ctx_buf(c, "bt_vp_all", per * std::max<int64_t>(nblk, 1));
    double* VpTAll = (double*)ctx_buf(c, "bt_vpt_all", per * std::max<int64_t>(nblk, 1));
    double* XAll = (double*)ctx_buf(c, "bt_x_all", per * std::max<int64_t>(nblk, 1));
    double* G = (double*)ctx_buf(c, "bt_g", (size_t)std::max<int64_t>(nblk, 1) * 128 * 128 * 8);
    double* T = (double*)ctx_buf(c, "bt_t", (size_t)std::max<int64_t>(nblk, 1) * 128 * 128 * 8);
    if (!VpAll || !VpTAll || !XAll || !G || !T) return SDPSR_OUT_OF_MEMORY;
    // the Gram matrices of all panels, then every T factor in ONE launch (one workgroup per block)
    for (int64_t b = 0; b < nblk; ++b) {
        const int64_t j0 = 128 * b, r0 = j0, m = ld - r0;
        double* Vp = VpAll + b * ld * 128;
        launch_bt_extract_panel(s, n, ld, A, j0, r0, Vp, VpTAll + b * ld * 128);
        int st = bt_gemm_splitk(c, 128, 128, m, Vp + r0, ld, Vp + r0, ld, G + b * 128 * 128);  // G_b = V'V
        if (st) return st;
    }
    launch_bt_larft(s, G, tau, nblk, n, T);
    for (int64_t b = 0; b < nblk; ++b) {
        const int64_t r0 = 128 * b, m = ld - r0;
        // X[:, r] = (V T)[r, :]':  X = T' V' as 128 x m (rows r0..)
        launch_gemm_tn_f64(s, 128, m, 128, T + b * 128 * 128, 128, VpTAll + b * ld * 128 + r0 * 128, 128, XAll + b * ld * 128 + r0 * 128, 128, 1, 0, 0, 0);
    }
    if (hipGetLastError() != hipSuccess) return ctx_fail(c, SDPSR_HIP_ERROR, "back-transformation launch failed");
    return SDPSR_OK;
}
static int backtransform_apply(sdpsr_ctx* c, int64_t n, int64_t ld, double* Z) {
    hipStream_t s = c->stream;
    const int64_t nblk = (n - 1 + 127) / 128;
    double* VpAll = (double*)ctx_buf(c, "bt_vp_all", (size_t)ld * 128 * 8 * std::max<int64_t>(nblk, 1));
    double* XAll = (double*)ctx_buf(c, "bt_x_all", (size_t)ld * 128 * 8 * std::max<int64_t>(nblk, 1));
    double* W = (double*)ctx_buf(c, "bt_w", (size_t)ld * 128 * 8);
    if (!VpAll || !XAll || !W) return SDPSR_OUT_OF_MEMORY;
    for (int64_t b = nblk - 1; b >= 0; --b) {  // the blocks applied last first
        const int64_t r0 = 128 * b, m = ld - r0;
        int st = bt_gemm_splitk(c, 128, ld, m, VpAll + b * ld * 128 + r0, ld, Z + r0, ld, W);  // W = V' Z
        if (st) return st;
        launch_gemm_tn_f64_sub(s, m, ld, 128, XAll + b * ld * 128 + r0 * 128, 128, W, 128, Z + r0, ld);  // Z -= (V T) W
    }
    if (hipGetLastError() != hipSuccess) return ctx_fail(c, SDPSR_HIP_ERROR, "back-transformation launch failed");
    return SDPSR_OK;
}

static int ensure_handle(sdpsr_ctx* c) {
    if (!c->rocblas) {
        rocblas_handle h = nullptr;
        if (rocblas_create_handle(&h) != rocblas_status_success)
            return ctx_fail(c, SDPSR_SOLVER_ERROR, "rocblas_create_handle failed");
        c->rocblas = h;
    }
    if (rocblas_set_stream((rocblas_handle)c->rocblas, c->stream) != rocblas_status_success)
        return ctx_fail(c, SDPSR_SOLVER_ERROR, "rocblas_set_stream failed");
    return SDPSR_OK;
}

void destroy_handle(sdpsr_ctx* c) {
    if (c->rocblas) rocblas_destroy_handle((rocblas_handle)c->rocblas);
    c->rocblas = nullptr;
}

// A: n x n column-major with leading dimension lda, lower triangle referenced; on exit the
// columns of A are the orthonormal eigenvectors, w ascending.
int syev_device(sdpsr_ctx* c, int64_t n, double* A, int64_t lda, double* w, double* host_w,
                const std::function<void()>* after_launch, bool defer_readback) {
    // rocBLAS / rocSOLVER only behind the comparison drivers (eig_driver 1 / 2 / 3 / 5) and for shapes the own path does
    // not take: the default path calls no library routine, so it pays neither the handle's start-up cost nor the
    // hipBLASLt initialisation inside rocblas_create_handle (ABI contract points)
    const int drv = c->opts.eig_driver;
    const bool small_path = n <= 128 && (drv == 0 || drv >= 4);
    const bool own_path = (drv == 0 || drv >= 4) && (lda % 128) == 0;
    const bool own_dc_path = own_path && drv != 5 && lda <= 8192;
    rocblas_handle h = nullptr;
    if (!small_path && (drv == 1 || !own_path || !own_dc_path)) {
        const int st = ensure_handle(c);
        if (st) return st;
        h = (rocblas_handle)c->rocblas;
    }
    double* E = (double*)ctx_buf(c, "eig_E", (size_t)n * sizeof(double));
    double* tau = (double*)ctx_buf(c, "eig_tau", (size_t)n * sizeof(double));
    rocblas_int* info = (rocblas_int*)ctx_buf(c, "eig_info", 64);
    if (!E || !tau || !info) return SDPSR_OUT_OF_MEMORY;
    rocblas_status rs;
    if (n <= 128 && (c->opts.eig_driver == 0 || c->opts.eig_driver >= 4)) {
        // one-workgroup Jacobi (kernels_sytrd.hip): the blocked path is pure launch latency here
        double* Vt = (double*)ctx_buf(c, "eig_Z", (size_t)n * n * sizeof(double));
        if (!Vt) return SDPSR_OUT_OF_MEMORY;
        launch_small_syev(c->stream, n, A, lda, w, Vt, (int*)info);
    } else if (c->opts.eig_driver == 1) {
        rs = rocsolver_dsyevd(h, rocblas_evect_original, rocblas_fill_lower, (rocblas_int)n, A,
                              (rocblas_int)lda, w, E, info);
        if (rs != rocblas_status_success)
            return ctx_fail(c, SDPSR_SOLVER_ERROR, "rocsolver_dsyevd status " + std::to_string(rs));
    } else {
        // own path (eig_driver 0 / 4): hand-written tridiagonalisation, rocSOLVER stedc, own
        // compact-WY back-transformation.  eig_driver 2 / 3: rocSOLVER sytrd / steqr + ormtr (comparison).
        const bool own = (c->opts.eig_driver == 0 || c->opts.eig_driver >= 4) && (lda % 128) == 0;
        const int64_t ldz = own ? lda : n;
        double* Z = (double*)ctx_buf(c, "eig_Z", (size_t)ldz * ldz * sizeof(double));
        if (!Z) return SDPSR_OUT_OF_MEMORY;
        bool bt_forked = false;
        // an error return between the fork and the join below must not leave the side stream reading A (the caller's
        // buffer in sdpsr_syev_f64) behind
        struct SideGuard {
            sdpsr_ctx* c;
            bool armed;
            ~SideGuard() {
                if (armed && c->side_stream) ctx_sync_stream(c, c->side_stream);
            }
        } side_guard{c, false};
        if (!own) {
            rs = rocsolver_dsytrd(h, rocblas_fill_lower, (rocblas_int)n, A, (rocblas_int)lda, w, E, tau);
            if (rs != rocblas_status_success)
                return ctx_fail(c, SDPSR_SOLVER_ERROR, "rocsolver_dsytrd status " + std::to_string(rs));
        } else {
            // hand-written tridiagonalisation (kernels_sytrd.hip); LAPACK-compatible output
            double* ws = (double*)ctx_buf(c, "eig_sytrd_ws", sytrd_workspace_doubles(n, lda) * sizeof(double));
            if (!ws) return SDPSR_OUT_OF_MEMORY;
            launch_sytrd(c, n, A, lda, w, E, tau, ws);
            if (dbg_on()) { ctx_sync_stream(c, c->stream); dbg_mark(c, "syev: tridiagonalisation done"); }
            if (hipGetLastError() != hipSuccess) return ctx_fail(c, SDPSR_HIP_ERROR, "sytrd launch failed");
            // first half of the back-transformation on the side stream, beside the tridiagonal solver
            const bool side_ok = ctx_ensure_side(c);
            hipStream_t main_stream = c->stream;
            if (side_ok && c->side_stream != main_stream && hipEventRecord(c->ev_bt_fork, main_stream) == hipSuccess &&
                hipStreamWaitEvent(c->side_stream, c->ev_bt_fork, 0) == hipSuccess) {
                c->stream = c->side_stream;  // every helper launches on c->stream
                c->main_shadow = main_stream;
                const int pst = backtransform_prepare(c, n, A, lda, tau);
                const bool rec = hipEventRecord(c->ev_bt_join, c->side_stream) == hipSuccess;
                c->stream = main_stream;
                c->main_shadow = nullptr;
                if (pst || !rec) {
                    ctx_sync_stream(c, c->side_stream);
                    return pst ? pst : ctx_fail(c, SDPSR_HIP_ERROR, "back-transformation: event record failed");
                }
                bt_forked = true;
                side_guard.armed = true;
            } else {
                (void)hipGetLastError();
            }
            if (hipMemsetAsync(Z, 0, (size_t)ldz * ldz * sizeof(double), c->stream) != hipSuccess)
                return ctx_fail(c, SDPSR_HIP_ERROR, "memset of the eigenvector buffer failed");
        }
        // the tridiagonal problem: own divide and conquer (kernels_stedc.hip); eig_driver 5 keeps rocSOLVER's stedc
        const bool own_dc = own && c->opts.eig_driver != 5 && ldz <= 8192;
        if (own_dc) {
            void* dws = ctx_buf(c, "eig_dc_ws", stedc_workspace_bytes(ldz));
            double* W1 = (double*)ctx_buf(c, "eig_dc_w1", (size_t)ldz * ldz * sizeof(double));
            double* W2 = (double*)ctx_buf(c, "eig_dc_w2", (size_t)ldz * ldz * sizeof(double));
            double* wtmp = (double*)ctx_buf(c, "eig_dc_w", (size_t)n * sizeof(double));
            if (!dws || !W1 || !W2 || !wtmp) return SDPSR_OUT_OF_MEMORY;
            std::vector<int> desc;
            const size_t db = stedc_descriptors(ldz, desc);
            int hst = h2d_sync(c, stedc_descriptor_slot(dws, ldz), desc.data(), db);
            if (hst) return hst;
            // (d = w and the output eigenvalues share the caller's array: the solver reads d, e in its first launch only)
            launch_stedc_check(c->stream, n, w, E, 1, (int*)info);  // info = 1 on NaN / Inf in the tridiagonal matrix
            if (!launch_stedc(c->stream, n, ldz, w, E, wtmp, Z, W1, W2, dws))
                return ctx_fail(c, SDPSR_HIP_ERROR, "tridiagonal divide and conquer: launch failed");
            if (hipMemcpyAsync(w, wtmp, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, c->stream) != hipSuccess)
                return ctx_fail(c, SDPSR_HIP_ERROR, "tridiagonal divide and conquer: copy failed");
            launch_stedc_check(c->stream, n, wtmp, nullptr, 0, (int*)info);  // ... or non-finite / unordered eigenvalues
            rs = rocblas_status_success;
        } else if (c->opts.eig_driver == 3)
            rs = rocsolver_dsteqr(h, rocblas_evect_tridiagonal, (rocblas_int)n, w, E, Z, (rocblas_int)ldz, info);
        else
            rs = rocsolver_dstedc(h, rocblas_evect_tridiagonal, (rocblas_int)n, w, E, Z, (rocblas_int)ldz, info);
        if (rs != rocblas_status_success)
            return ctx_fail(c, SDPSR_SOLVER_ERROR, "rocsolver tridiagonal solver status " + std::to_string(rs));
        if (dbg_on()) { ctx_sync_stream(c, c->stream); dbg_mark(c, "syev: tridiagonalisation + tridiagonal solver done"); }
        if (own) {
            if (bt_forked) {
                if (hipStreamWaitEvent(c->stream, c->ev_bt_join, 0) != hipSuccess) return ctx_fail(c, SDPSR_HIP_ERROR, "back-transformation: join failed");
                side_guard.armed = false;  // the main stream now waits for the side stream's work itself
            } else {
                const int pst = backtransform_prepare(c, n, A, lda, tau);
                if (pst) return pst;
            }
            const int bst = backtransform_apply(c, n, lda, Z);
            if (bst) return bst;
        } else {
            rs = rocsolver_dormtr(h, rocblas_side_left, rocblas_fill_lower, rocblas_operation_none,
                                  (rocblas_int)n, (rocblas_int)n, A, (rocblas_int)lda, tau, Z,
                                  (rocblas_int)ldz);
            if (rs != rocblas_status_success)
                return ctx_fail(c, SDPSR_SOLVER_ERROR, "rocsolver_dormtr status " + std::to_string(rs));
        }
        if (hipMemcpy2DAsync(A, lda * sizeof(double), Z, ldz * sizeof(double), n * sizeof(double), n,
                             hipMemcpyDeviceToDevice, c->stream) != hipSuccess)
            return ctx_fail(c, SDPSR_HIP_ERROR, "copy of eigenvectors failed");
    }
    if (dbg_on()) { ctx_sync_stream(c, c->stream); dbg_mark(c, "syev: back-transformation done"); }
    if (after_launch) (*after_launch)();  // everything of the eigensolver is enqueued; the caller overlaps its own launches here
    if (defer_readback) return SDPSR_OK;  // the caller reads "eig_info" (status, sweeps) and w with its own next read-back
    // read-back through the pinned scratch of the ctx (a pageable 4-byte copy costs tens of us)
    rocblas_int* hpin = (rocblas_int*)ctx_pinned(c, 64 + (host_w ? (size_t)n * sizeof(double) : 0));
    if (!hpin || hipMemcpyAsync(hpin, info, 2 * sizeof(rocblas_int), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
        (host_w && hipMemcpyAsync((char*)hpin + 64, w, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream) != hipSuccess) ||
        ctx_sync_stream(c, c->stream) != hipSuccess)
        return ctx_fail(c, SDPSR_HIP_ERROR, "eigensolver info read-back failed");
    if (host_w) memcpy(host_w, (char*)hpin + 64, (size_t)n * sizeof(double));
    const rocblas_int hinfo = hpin[0];
    if (dbg_on() && n <= 128) fprintf(stderr, "[sdpsr] small syev n=%lld: %d sweeps\n", (long long)n, (int)hpin[1]);
    if (hinfo != 0)
        return ctx_fail(c, SDPSR_SOLVER_ERROR, "eigensolver did not converge, info=" + std::to_string(hinfo));
    return SDPSR_OK;
}

}  // namespace sdpsr
