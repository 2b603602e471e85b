// One reduction in one call: admissible_subspace (src/partitions.jl:109-190) followed by
// blockDiagonalize (src/compat.jl:46-68) on its result, the partition never leaving the device and
// the host never waiting between the stages (the three separate entry points each return with their
// outputs complete, include/sdpsr.h: two host synchronisations and one label copy + symmetry pass per
// reduction that a caller who wants both results does not need).
#include <algorithm>
#include <cstring>
#include <new>

#include "host_internal.h"

using namespace sdpsr;

namespace sdpsr {
// mem_in: where CL / X0L / U live; mem_out: where P_out / blks / Q_hat live
int jordan_reduce_impl(sdpsr_ctx* c, int64_t n, const double* CL, const double* X0L, const double* U, int64_t r, double atol, double epsilon,
                       uint32_t* P_out, int64_t* dim_out, int32_t* iters_out, int32_t* nblocks, int64_t* sum_sq, int64_t* sum_s, double* blks,
                       int64_t blks_capacity, double* Q_hat, int64_t qhat_capacity, double* phase_ms, int mem_in, int mem_out) {
    const int mem = mem_out;
    CHECK_CTX(c);
    if (!dim_out || n < 1 || !(epsilon > 0) || blks_capacity < 0 || qhat_capacity < 0 || (blks_capacity > 0 && !blks) ||
        (qhat_capacity > 0 && !Q_hat))
        return ctx_fail(c, SDPSR_BAD_ARGUMENT, "bad arguments");
    const int64_t len = n * n;
    int st = check_len(c, len);
    if (st) return st;
    hipStream_t s = c->stream;
    // the partition is formed where blockDiagonalize reads its labels: in the caller's device buffer P_out when there
    // is one (no copy at all), else in the ctx buffer "bd_labels"
    const bool in_place = mem == SDPSR_MEM_DEVICE && P_out != nullptr;
    uint32_t* L = in_place ? P_out : (uint32_t*)ctx_buf(c, "bd_labels", (size_t)len * 4);
    if (!L) return SDPSR_OUT_OF_MEMORY;
    c->bd_valid = false;
    c->bd_q_valid = false;
    c->bd_labels_ext = nullptr;
    double pm_a[SDPSR_T_COUNT] = {}, pm_b[SDPSR_T_COUNT] = {}, pm_i[SDPSR_T_COUNT] = {};
    int labels_sym = 0;
    c->deferred_verdict = nullptr;
    c->allow_deferred_verdict = !(c->opts.flags & SDPSR_FLAG_WAIT_FOR_EVERY_VERDICT);  // (the loop's last verdicts may ride on this reduction's later host waits, see sdpsr_internal.h)
    st = admissible_subspace_impl(c, n, CL, X0L, U, r, atol, L, dim_out, iters_out, phase_ms ? pm_a : nullptr, mem_in, SDPSR_MEM_DEVICE,
                                  /*final_sync=*/false, &labels_sym);
    c->allow_deferred_verdict = false;
    const int st_loop = st;
    if (st && st != SDPSR_NOT_CONVERGED) return st;
    if (P_out && !in_place) {  // stream-ordered; complete when the call returns (it ends with a synchronisation on every path)
        HIP_TRY(c, hipMemcpyAsync(P_out, L, (size_t)len * 4, mem == SDPSR_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s));
        if (mem != SDPSR_MEM_DEVICE) c->d2h_bytes += (size_t)len * 4;
    }
    const int64_t d = *dim_out;
    int32_t nb = 0;
    int64_t S = 0, S1 = 0;
    st = block_diagonalize_impl(c, n, L, d, epsilon, &nb, &S, &S1, phase_ms ? pm_b : nullptr, SDPSR_MEM_DEVICE, labels_sym != 0,
                                /*final_sync=*/false, in_place);
    if (nblocks) *nblocks = nb;
    if (sum_sq) *sum_sq = S;
    if (sum_s) *sum_s = S1;
    bool images_done = false;
    if (st == SDPSR_OK && blks && d * S <= blks_capacity && (!Q_hat || n * S1 <= qhat_capacity)) {
        st = sdpsr_block_images(c, blks, (Q_hat && n * S1 <= qhat_capacity) ? Q_hat : nullptr, phase_ms ? pm_i : nullptr, mem);  // ends synchronised
        images_done = st == SDPSR_OK;
    }
    if (in_place) {
        // the caller's buffer stops serving phase 2 when this call returns: a later sdpsr_block_images (images not
        // delivered here) finds the labels in the ctx buffer.  On every path out of this block the ctx forgets the
        // caller's pointer; a failed hand-over also drops the pending phase 2 (it would read a buffer the caller owns).
        struct Forget {
            sdpsr_ctx* c;
            bool ok = false;
            ~Forget() {
                c->bd_labels_ext = nullptr;
                c->bd_trusted_symmetric = nullptr;
                if (!ok) c->bd_valid = false;
            }
        } forget{c};
        if (!images_done && c->bd_valid) {
            uint32_t* keep = (uint32_t*)ctx_buf(c, "bd_labels", (size_t)len * 4);
            if (!keep) return SDPSR_OUT_OF_MEMORY;
            HIP_TRY(c, hipMemcpyAsync(keep, P_out, (size_t)len * 4, hipMemcpyDeviceToDevice, s));
        } else if (images_done) {
            c->bd_valid = false;  // nothing left for a second sdpsr_block_images to work on (sizes and Q_hat stay available)
        }
        forget.ok = true;
    }
    if (!images_done) {  // (sdpsr_block_images ends synchronised, and nothing has been enqueued since)
        const hipError_t e = ctx_sync_stream(c, s);
        if (e != hipSuccess && st == SDPSR_OK) st = ctx_fail(c, SDPSR_HIP_ERROR, std::string("hipStreamSynchronize: ") + hipGetErrorString(e));
    }
    if (c->deferred_verdict) {
        // The stream has been waited for: the verdicts the loop left unread are in.  A violated one means the loop stopped on a
        // partition that one more step would have refined -- whatever blockDiagonalize made of it (a failure included) is void;
        // the reduction is done again, this time reading the verdicts where they arise.
        const volatile uint32_t* dv = c->deferred_verdict;
        c->deferred_verdict = nullptr;
        if (dv[0] != 0 || dv[16] != 0) {
            c->predict_closed = false;
            if (dbg_on()) fprintf(stderr, "[sdpsr] jordan_reduce: the input was not closed after all (deferred verdicts %u %u): reduction repeated\n", dv[0], dv[16]);
            return jordan_reduce_impl(c, n, CL, X0L, U, r, atol, epsilon, P_out, dim_out, iters_out, nblocks, sum_sq, sum_s, blks, blks_capacity, Q_hat,
                                      qhat_capacity, phase_ms, mem_in, mem_out);
        }
    }
    if (phase_ms) {
        for (int i = 0; i < SDPSR_T_COUNT; ++i) phase_ms[i] = pm_a[i] + pm_b[i] + pm_i[i];
    }
    return st ? st : st_loop;
}
}  // namespace sdpsr

extern "C" int sdpsr_jordan_reduce(sdpsr_ctx* c, int64_t n, const double* CL, const double* X0L, const double* U, int64_t r,
                                   double atol, double epsilon, uint32_t* P_out, int64_t* dim_out, int32_t* iters_out,
                                   int32_t* nblocks, int64_t* sum_sq, int64_t* sum_s, double* blks, int64_t blks_capacity,
                                   double* Q_hat, int64_t qhat_capacity, double* phase_ms, int mem) {
    return jordan_reduce_impl(c, n, CL, X0L, U, r, atol, epsilon, P_out, dim_out, iters_out, nblocks, sum_sq, sum_s, blks, blks_capacity, Q_hat,
                              qhat_capacity, phase_ms, mem, mem);
}

// ---------------------------------------------------------------------------
// Upload once, reduce many times: the problem handle (include/sdpsr.h)
// ---------------------------------------------------------------------------
extern "C" int sdpsr_problem_create(sdpsr_ctx* c, int64_t n, const double* CL, const double* X0L, const double* U, int64_t r, int hint, int mem,
                                    sdpsr_problem** out) {
    CHECK_CTX(c);
    if (!out || !CL || !X0L || n < 1 || r < 0 || (r > 0 && !U)) return ctx_fail(c, SDPSR_BAD_ARGUMENT, "bad arguments");
    *out = nullptr;
    const int64_t len = n * n;
    int st = check_len(c, len);
    if (st) return st;
    DeviceGuard dg(c->device);
    sdpsr_problem* p = new (std::nothrow) sdpsr_problem();
    if (!p) return SDPSR_OUT_OF_MEMORY;
    p->device = c->device;
    p->n = n;
    p->r = r;
    p->hint = hint & 3;
    const size_t vb = (size_t)len * 8;
    auto fail = [&](int code, const char* what) {
        sdpsr_problem_destroy(p);
        return ctx_fail(c, code, what);
    };
    if (hipMalloc((void**)&p->CL, vb) != hipSuccess || hipMalloc((void**)&p->X0, vb) != hipSuccess ||
        (r > 0 && hipMalloc((void**)&p->U, vb * (size_t)r) != hipSuccess))
        return fail(SDPSR_OUT_OF_MEMORY, "sdpsr_problem_create: device memory");
    const hipMemcpyKind kind = mem == SDPSR_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    if (hipMemcpyAsync(p->CL, CL, vb, kind, c->stream) != hipSuccess || hipMemcpyAsync(p->X0, X0L, vb, kind, c->stream) != hipSuccess ||
        (r > 0 && hipMemcpyAsync(p->U, U, vb * (size_t)r, kind, c->stream) != hipSuccess))
        return fail(SDPSR_HIP_ERROR, "sdpsr_problem_create: copy");
    if (mem != SDPSR_MEM_DEVICE) c->h2d_bytes += vb * (size_t)(2 + r);
    if (ctx_sync_stream(c, c->stream) != hipSuccess) return fail(SDPSR_HIP_ERROR, "sdpsr_problem_create: synchronisation");
    *out = p;
    return SDPSR_OK;
}

extern "C" int sdpsr_problem_destroy(sdpsr_problem* p) {
    if (!p) return SDPSR_OK;
    DeviceGuard dg(p->device);
    if (p->CL) hipFree(p->CL);
    if (p->X0) hipFree(p->X0);
    if (p->U) hipFree(p->U);
    delete p;
    return SDPSR_OK;
}

extern "C" int sdpsr_problem_reduce(sdpsr_ctx* c, const sdpsr_problem* p, double atol, double epsilon, uint32_t* P_out, int64_t* dim_out,
                                    int32_t* iters_out, int32_t* nblocks, int64_t* sum_sq, int64_t* sum_s, double* blks, int64_t blks_capacity,
                                    double* Q_hat, int64_t qhat_capacity, double* phase_ms, int mem_out) {
    CHECK_CTX(c);
    if (!p || p->device != c->device) return ctx_fail(c, SDPSR_BAD_ARGUMENT, "sdpsr_problem_reduce: no problem, or one of another device");
    c->hint_symmetric_basis = p->hint;
    return jordan_reduce_impl(c, p->n, p->CL, p->X0, p->U, p->r, atol, epsilon, P_out, dim_out, iters_out, nblocks, sum_sq, sum_s, blks,
                              blks_capacity, Q_hat, qhat_capacity, phase_ms, SDPSR_MEM_DEVICE, mem_out);
}

extern "C" int sdpsr_problem_reduce_batch(sdpsr_ctx* c, const sdpsr_problem* p, int32_t R, const uint64_t* seeds, double atol, double epsilon,
                                          uint32_t* const* P_out, int64_t* dim_out, int32_t* iters_out, int32_t* nblocks, int64_t* sum_sq,
                                          int64_t* sum_s, double* const* blks, const int64_t* blks_capacity, int32_t* status, int mem_out) {
    CHECK_CTX(c);
    if (!p || p->device != c->device) return ctx_fail(c, SDPSR_BAD_ARGUMENT, "sdpsr_problem_reduce_batch: no problem, or one of another device");
    return jordan_reduce_batch_impl(c, R, seeds, p->n, p->CL, p->X0, p->U, p->r, p->hint, atol, epsilon, P_out, dim_out, iters_out, nblocks, sum_sq,
                                    sum_s, blks, blks_capacity, status, SDPSR_MEM_DEVICE, mem_out);
}
