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

#include "sdpsr_internal.h"

namespace sdpsr {

size_t sytrd_workspace_doubles(int64_t n, int64_t ld);
void launch_sytrd(sdpsr_ctx* c, int64_t n, double* A, int64_t ld, double* d, double* e, double* tau, double* ws);
bool launch_small_syev(hipStream_t s, int64_t n, double* A, int64_t lda, double* w, double* Vtmp, int* info);

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
int syev_device(sdpsr_ctx* c, int64_t n, double* A, int64_t lda, double* w, double* host_w) {
    int st = ensure_handle(c);
    if (st) return st;
    rocblas_handle h = (rocblas_handle)c->rocblas;
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
        double* Z = (double*)ctx_buf(c, "eig_Z", (size_t)n * n * sizeof(double));
        if (!Z) return SDPSR_OUT_OF_MEMORY;
        if (c->opts.eig_driver == 2 || (lda & 1)) {
            rs = rocsolver_dsytrd(h, rocblas_fill_lower, (rocblas_int)n, A, (rocblas_int)lda, w, E, tau);
            if (rs != rocblas_status_success)
                return ctx_fail(c, SDPSR_SOLVER_ERROR, "rocsolver_dsytrd status " + std::to_string(rs));
        } else {
            // hand-written tridiagonalisation (kernels_sytrd.hip); LAPACK-compatible output
            double* ws = (double*)ctx_buf(c, "eig_sytrd_ws", sytrd_workspace_doubles(n, lda) * sizeof(double));
            if (!ws) return SDPSR_OUT_OF_MEMORY;
            launch_sytrd(c, n, A, lda, w, E, tau, ws);
            if (hipGetLastError() != hipSuccess) return ctx_fail(c, SDPSR_HIP_ERROR, "sytrd launch failed");
        }
        if (c->opts.eig_driver == 3)
            rs = rocsolver_dsteqr(h, rocblas_evect_tridiagonal, (rocblas_int)n, w, E, Z, (rocblas_int)n, info);
        else
            rs = rocsolver_dstedc(h, rocblas_evect_tridiagonal, (rocblas_int)n, w, E, Z, (rocblas_int)n, info);
        if (rs != rocblas_status_success)
            return ctx_fail(c, SDPSR_SOLVER_ERROR, "rocsolver tridiagonal solver status " + std::to_string(rs));
        rs = rocsolver_dormtr(h, rocblas_side_left, rocblas_fill_lower, rocblas_operation_none,
                              (rocblas_int)n, (rocblas_int)n, A, (rocblas_int)lda, tau, Z,
                              (rocblas_int)n);
        if (rs != rocblas_status_success)
            return ctx_fail(c, SDPSR_SOLVER_ERROR, "rocsolver_dormtr status " + std::to_string(rs));
        if (hipMemcpy2DAsync(A, lda * sizeof(double), Z, n * sizeof(double), n * sizeof(double), n,
                             hipMemcpyDeviceToDevice, c->stream) != hipSuccess)
            return ctx_fail(c, SDPSR_HIP_ERROR, "copy of eigenvectors failed");
    }
    // read-back through the pinned scratch of the ctx (a pageable 4-byte copy costs tens of us)
    rocblas_int* hpin = (rocblas_int*)ctx_pinned(c, 64 + (host_w ? (size_t)n * sizeof(double) : 0));
    if (!hpin || hipMemcpyAsync(hpin, info, 2 * sizeof(rocblas_int), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
        (host_w && hipMemcpyAsync((char*)hpin + 64, w, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream) != hipSuccess) ||
        hipStreamSynchronize(c->stream) != hipSuccess)
        return ctx_fail(c, SDPSR_HIP_ERROR, "eigensolver info read-back failed");
    if (host_w) memcpy(host_w, (char*)hpin + 64, (size_t)n * sizeof(double));
    const rocblas_int hinfo = hpin[0];
    if (getenv("SDPSR_DEBUG") && n <= 128) fprintf(stderr, "[sdpsr] small syev n=%lld: %d sweeps\n", (long long)n, (int)hpin[1]);
    if (hinfo != 0)
        return ctx_fail(c, SDPSR_SOLVER_ERROR, "eigensolver did not converge, info=" + std::to_string(hinfo));
    return SDPSR_OK;
}

}  // namespace sdpsr
