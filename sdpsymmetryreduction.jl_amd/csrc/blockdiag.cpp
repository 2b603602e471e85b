// blockDiagonalize (src/compat.jl:46-68): diagonalize + check_block_sizes (src/diagonalize.jl:1-40),
// then basis_image (:42-89).  Entry points sdpsr_block_diagonalize / _block_sizes / _q_hat / _block_images.
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <functional>
#include <numeric>

#include "host_internal.h"

using namespace sdpsr;

namespace sdpsr {
int block_diagonalize_impl(sdpsr_ctx* c, int64_t n, const uint32_t* P, int64_t d, double epsilon, int32_t* nblocks, int64_t* sum_sq,
                           int64_t* sum_s, double* phase_ms, int mem, bool trusted_symmetric, bool final_sync, bool in_place) {
    CHECK_CTX(c);
    if (!P || n < 1 || d < 0 || !(epsilon > 0)) return ctx_fail(c, SDPSR_BAD_ARGUMENT, "bad arguments");
    const int64_t len = n * n;
    int st = check_len(c, len);
    if (st) return st;
    hipStream_t s = c->stream;
    c->bd_valid = false;
    c->bd_q_valid = false;
    PhaseTimer tm(c, phase_ms != nullptr);
    TotalEvents ev_total(phase_ms != nullptr, s);
    // keep a device copy of the labels for phase 2 (or, in_place, the caller's device buffer itself)
    in_place = in_place && mem == SDPSR_MEM_DEVICE;
    uint32_t* L = in_place ? const_cast<uint32_t*>(P) : (uint32_t*)ctx_buf(c, "bd_labels", len * 4);
    if (!L) return SDPSR_OUT_OF_MEMORY;
    c->bd_labels_ext = in_place ? P : nullptr;
    c->bd_sym_labels = nullptr;
    c->bd_sym_epoch = 0;
    c->bd_trusted_symmetric = (trusted_symmetric && mem == SDPSR_MEM_DEVICE && P == L) ? L : nullptr;
    if (mem == SDPSR_MEM_DEVICE) {
        if (P != L) {
            // copy and symmetry check of the same tiles in one pass; the verdict ("bd_symflag"[0] ==
            // epoch <=> not symmetric) is read back by the driver with its first synchronisation
            const bool fresh = c->bufs.find("bd_symflag") == c->bufs.end();
            uint32_t* sf = (uint32_t*)ctx_buf(c, "bd_symflag", 64);
            if (!sf) return SDPSR_OUT_OF_MEMORY;
            if (fresh) HIP_TRY(c, hipMemsetAsync(sf, 0, 64, s));
            if (++c->epoch_counter == 0) ++c->epoch_counter;
            launch_copy_check_symmetric(s, n, P, L, sf, c->epoch_counter);
            c->bd_sym_epoch = c->epoch_counter;
            c->bd_sym_labels = L;
        }
    } else {
        HIP_TRY(c, hipMemcpyAsync(L, P, len * 4, hipMemcpyHostToDevice, s));
    }
    dbg_mark(c, "block_diagonalize: entered, labels copied");
    const double atol = epsilon;  // diagonalize(T, P; atol=epsilon), src/compat.jl:53
    EigInfo info;
    std::vector<int32_t> sizes;
    int64_t S1 = 0, S = 0;
    st = DRIVER_FALLBACK;
    if (compression_eligible(c, n, d)) st = compressed_diagonalize(c, n, L, d, atol, info, sizes, S1, S, tm);
    if (st == DRIVER_FALLBACK && c->opts.eig_driver == 6)
        return ctx_fail(c, SDPSR_SOLVER_ERROR, "requested driver not applicable to this partition (" + c->err + ")");
    if (st != SDPSR_OK && st != DRIVER_FALLBACK) return st;
    if (st == DRIVER_FALLBACK) {
        st = dense_diagonalize(c, n, L, nullptr, atol, info, sizes, S1, S, tm, d);  // d: extra coupling elements on demand
        if (st) return st;
    }

    dbg_mark(c, "block_diagonalize: diagonalize done");
    // check_block_sizes (src/diagonalize.jl:1-11)
    int64_t final_dim = 0;
    for (int32_t sz : sizes) final_dim += (int64_t)sz * (sz + 1) / 2;
    c->bd_n = n;
    c->bd_d = d;
    c->bd_sizes = sizes;
    c->bd_sum_s = S1;
    c->bd_sum_sq = S;
    c->bd_q_valid = true;
    if (nblocks) *nblocks = (int32_t)sizes.size();
    if (sum_sq) *sum_sq = S;
    if (sum_s) *sum_s = S1;
    if (phase_ms) {
        const float ms = ev_total.stop(s);
        tm.collect();
        for (int i = 0; i < SDPSR_T_COUNT; ++i) phase_ms[i] = tm.acc[i];
        phase_ms[SDPSR_T_TOTAL] = ms;
    } else if (final_sync) {
        HIP_TRY(c, ctx_sync_stream(c, s));
    }
    if (final_dim != d) {
        std::string szs;
        for (int32_t sz : sizes) szs += std::to_string(sz) + " ";
        return ctx_fail(c, SDPSR_DIMENSION_MISMATCH,
                        "final_dim=" + std::to_string(final_dim) + " block_sizes=[" + szs + "] expected dim(P)=" +
                            std::to_string(d) + " (rounding error: try another epsilon or try again; or the algebra is not block-diagonalizable over the reals)");
    }
    c->bd_valid = true;
    return SDPSR_OK;
}
}  // namespace sdpsr

extern "C" {

int sdpsr_block_diagonalize(sdpsr_ctx* c, int64_t n, const uint32_t* P, int64_t d, double epsilon,
                            int32_t* nblocks, int64_t* sum_sq, int64_t* sum_s, double* phase_ms,
                            int mem) {
    return block_diagonalize_impl(c, n, P, d, epsilon, nblocks, sum_sq, sum_s, phase_ms, mem, false, true);
}

}  // extern "C"

extern "C" {

int sdpsr_block_sizes(sdpsr_ctx* c, int32_t* blk_sizes) {
    if (!c || !blk_sizes) return SDPSR_BAD_ARGUMENT;
    if (c->bd_sizes.empty()) return ctx_fail(c, SDPSR_BAD_STATE, "no block diagonalisation available");
    memcpy(blk_sizes, c->bd_sizes.data(), c->bd_sizes.size() * sizeof(int32_t));
    return SDPSR_OK;
}

int sdpsr_q_hat(sdpsr_ctx* c, double* Q_hat, int mem) {
    CHECK_CTX(c);
    if (!c->bd_q_valid) return ctx_fail(c, SDPSR_BAD_STATE, "no diagonalisation available on this ctx");
    if (!Q_hat) return ctx_fail(c, SDPSR_BAD_ARGUMENT, "null pointer");
    const size_t cnt = (size_t)c->bd_n * c->bd_sum_s;
    double* Qhat = (double*)ctx_buf(c, "bd_qhat", cnt * 8);
    if (!Qhat) return SDPSR_OUT_OF_MEMORY;
    HIP_TRY(c, hipMemcpyAsync(Q_hat, Qhat, cnt * 8, mem == SDPSR_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost,
                              c->stream));
    HIP_TRY(c, ctx_sync_stream(c, c->stream));
    return SDPSR_OK;
}

int sdpsr_block_images(sdpsr_ctx* c, double* blks, double* Q_hat, double* phase_ms, int mem) {
    CHECK_CTX(c);
    if (!c->bd_valid) return ctx_fail(c, SDPSR_BAD_STATE, "sdpsr_block_diagonalize has not succeeded on this ctx");
    if (!blks) return ctx_fail(c, SDPSR_BAD_ARGUMENT, "null pointer");
    hipStream_t s = c->stream;
    const int64_t n = c->bd_n, d = c->bd_d, S1 = c->bd_sum_s, S = c->bd_sum_sq, len = n * n;
    TotalEvents ev_total(phase_ms != nullptr, s);
    int st = SDPSR_OK;
    uint32_t* L = c->bd_labels_ext ? const_cast<uint32_t*>(c->bd_labels_ext) : (uint32_t*)ctx_buf(c, "bd_labels", len * 4);
    double* Qhat = (double*)ctx_buf(c, "bd_qhat", (size_t)n * S1 * 8);
    double* Qrm = (double*)ctx_buf(c, "bd_qrm", (size_t)n * S1 * 8);
    double* out = out_dev(c, "bd_blks", blks, (size_t)d * S, mem, &st);
    if (st || !L || !Qhat || !Qrm) return st ? st : SDPSR_OUT_OF_MEMORY;
    launch_transpose_to_rowmajor(s, n, S1, Qhat, Qrm);
    const double atol = 1e-12 * (double)n;  // basis_image default atol (src/diagonalize.jl:67)
    // opts.basis_image_kernel = 1 two_stage | 2 outer | 3 chunk forces one of the three kernels (tests: the
    // automatic choice reaches `outer` / `chunk` only for shapes far beyond the test sizes)
    const int force = c->opts.basis_image_kernel;
    const bool f_two = force == 1, f_outer = force == 2, f_chunk = force == 3;
    // commutative case (every block 1 x 1): the images are eigenvalues, lambda_ik = q_k'(1[P==i] x) for x = sum_k q_k
    // -- one vector's class sums -- with a randomized self-check; the projection formula below runs if the check
    // fails (kernels_blockdiag.hip, launch_basis_image_commutative) and always under SDPSR_FLAG_FULL_BASIS_IMAGE
    bool done = false;
    bool done_synced = false;  // the stream was synchronised by the commutative path's verdict and nothing was enqueued since
    if (S == S1 && force == 0 && !(c->opts.flags & SDPSR_FLAG_FULL_BASIS_IMAGE) && d >= 1 && n >= 64) {
        double* ws = (double*)ctx_buf(c, "bi_comm_ws", basis_image_commutative_workspace_doubles(n, d) * 8);
        // per-column verdicts, stored by the check kernel straight into pinned host memory: its own words, beside
        // (not inside) the refinement's counters at the start of the buffer
        uint32_t* hv = (uint32_t*)ctx_pinned(c, 256 + (size_t)(S1 + 1) * 4);
        if (!ws || !hv) return SDPSR_OUT_OF_MEMORY;
        hv += 64;
        if (launch_basis_image_commutative(s, n, d, S1, L, Qrm, next_key(c), atol, 2e-10, ws, out, hv)) {
            HIP_TRY(c, ctx_sync_stream(c, s));
            HIP_TRY(c, hipGetLastError());
            const uint32_t nbad = hv[0];
            done = nbad == 0;
            done_synced = done;
            if (!done && nbad <= 8) {
                // a few columns failed (the eigenvectors of a pair of close eigenvalues): the projection formula for
                // those columns only, two per extra class-sum pass
                std::vector<int> badk;
                for (int64_t k2 = 0; k2 < S1; ++k2)
                    if (hv[1 + k2]) badk.push_back((int)k2);
                done = true;
                for (size_t q = 0; q < badk.size() && done; q += 2)
                    done = launch_basis_image_fix_pair(s, n, d, S1, L, Qrm, badk[q], badk[q + 1 < badk.size() ? q + 1 : q], atol, ws, out);
                if (dbg_on()) fprintf(stderr, "[sdpsr] basis_image: invariance check failed for %u column(s), projection formula for those\n", nbad);
            } else if (!done && dbg_on()) {
                fprintf(stderr, "[sdpsr] basis_image: invariance check failed for %u columns, projection formula instead\n", nbad);
            }
        }
    }
    // blocks up to 3 x 3 (non-commutative algebras with small blocks: ER(q) (x) K_k): the images from FOUR vectors' class
    // sums with a per-block check (kernels_blockdiag.hip, launch_basis_image_blocks); blocks that fail get the projection
    // formula, one block per extra pair of passes; many failures: the two-stage kernels below for everything
    if (!done && S != S1 && force == 0 && !(c->opts.flags & SDPSR_FLAG_FULL_BASIS_IMAGE) && d >= 1 && n >= 64 && c->bd_sizes.size() <= 4096) {
        int max_s = 0;
        for (int32_t sz : c->bd_sizes) max_s = std::max(max_s, (int)sz);
        if (max_s <= 3) {
            const int nb = (int)c->bd_sizes.size();
            std::vector<int32_t> hcs(2 * (size_t)nb);
            std::vector<int64_t> hoff(nb);
            int64_t colbase = 0, off = 0;
            for (int k2 = 0; k2 < nb; ++k2) {
                hcs[k2] = (int32_t)colbase;
                hcs[nb + k2] = c->bd_sizes[k2];
                hoff[k2] = off;
                colbase += c->bd_sizes[k2];
                off += (int64_t)c->bd_sizes[k2] * c->bd_sizes[k2];
            }
            int32_t* d_cs = (int32_t*)ctx_buf(c, "bi_colsz", (size_t)2 * nb * 4);
            int64_t* d_off = (int64_t*)ctx_buf(c, "bi_off", (size_t)nb * 8);
            double* ws = (double*)ctx_buf(c, "bi_comm_ws", basis_image_blocks_workspace_doubles(n, d) * 8);
            uint32_t* hv = (uint32_t*)ctx_pinned(c, 256 + (size_t)(nb + 1) * 4);
            if (!d_cs || !d_off || !ws || !hv) return SDPSR_OUT_OF_MEMORY;
            hv += 64;
            st = h2d_sync(c, d_cs, hcs.data(), (size_t)2 * nb * 4);
            if (!st) st = h2d_sync(c, d_off, hoff.data(), (size_t)nb * 8);
            if (st) return st;
            if (launch_basis_image_blocks(s, n, d, S1, S, nb, d_cs, d_cs + nb, d_off, L, Qrm, next_key(c), -1, atol, 2e-10, ws, out, hv)) {
                HIP_TRY(c, ctx_sync_stream(c, s));
                HIP_TRY(c, hipGetLastError());
                const uint32_t nbad = hv[0];
                done = nbad == 0;
                done_synced = done;
                if (!done && nbad <= 4) {
                    done = true;
                    for (int k2 = 0; k2 < nb && done; ++k2)
                        if (hv[1 + k2]) done = launch_basis_image_blocks(s, n, d, S1, S, nb, d_cs, d_cs + nb, d_off, L, Qrm, 0, k2, atol, 2e-10, ws, out, nullptr);
                }
                if (!done_synced && dbg_on()) fprintf(stderr, "[sdpsr] basis_image: invariance check failed for %u block(s)%s\n", nbad, done ? ", projection formula for those" : ", two-stage kernels instead");
            }
        }
    }
    if (done) {
        // (out is complete)
    } else if (basis_image_two_stage_fits(n, d, S1) && !f_outer && !f_chunk) {
        // two-stage form (class sums per row, then the s_k x s_k dots): descriptor = the two
        // columns of Q_hat every output multiplies, blocks side by side, column-major inside
        std::vector<int32_t> hdesc(2 * (size_t)S);
        {
            int64_t o = 0, colbase = 0;
            for (int32_t sz : c->bd_sizes) {
                for (int b2 = 0; b2 < sz; ++b2)
                    for (int a2 = 0; a2 < sz; ++a2) {
                        hdesc[o] = (int32_t)(colbase + a2);
                        hdesc[S + o] = (int32_t)(colbase + b2);
                        ++o;
                    }
                colbase += sz;
            }
        }
        int32_t* d_desc = (int32_t*)ctx_buf(c, "bi_desc", (size_t)2 * S * 4);
        double* Tb = (double*)ctx_buf(c, "bi_T", (size_t)d * n * S1 * 8);
        if (!d_desc || !Tb) return SDPSR_OUT_OF_MEMORY;
        st = h2d_sync(c, d_desc, hdesc.data(), (size_t)2 * S * 4);
        if (st) return st;
        launch_basis_image_two_stage(s, n, d, S1, S, L, Qrm, Tb, d_desc, d_desc + S, atol, out);
    } else {
    // _constraints(P): entries grouped by class (src/diagonalize.jl:42-50)
    uint32_t* ent = nullptr;
    int64_t* class_ptr = nullptr;  // host, size d+2: class_ptr[l]..class_ptr[l+1] = label l
    st = sort_entries_by_label(c, len, d, L, &ent, &class_ptr);
    if (st) return st;
    int max_s = 0;
    for (int32_t sz : c->bd_sizes) max_s = std::max(max_s, (int)sz);
    // many small classes (average class below 4096 entries) and blocks up to 256: outer-product
    // kernel, one workgroup per (class, block), every output written once, no partial sums
    (void)f_two;
    if (max_s <= 256 && d > 0 && (len / d < 4096 || f_outer) && !f_chunk && d <= 0x7FFFFFFF && c->bd_sizes.size() <= 65535) {
        const int nb = (int)c->bd_sizes.size();
        std::vector<int32_t> hcol(nb), hsz(nb);
        std::vector<int64_t> hoff(nb);
        int64_t colbase = 0, off = 0;
        for (int k2 = 0; k2 < nb; ++k2) {
            hcol[k2] = (int32_t)colbase;
            hsz[k2] = c->bd_sizes[k2];
            hoff[k2] = off;
            colbase += hsz[k2];
            off += (int64_t)hsz[k2] * hsz[k2];
        }
        int32_t* d_col = (int32_t*)ctx_buf(c, "bi_col", (size_t)nb * 4);
        int32_t* d_sz = (int32_t*)ctx_buf(c, "bi_sz", (size_t)nb * 4);
        int64_t* d_off = (int64_t*)ctx_buf(c, "bi_off", (size_t)nb * 8);
        int64_t* d_cls = (int64_t*)ctx_buf(c, "bi_cls_ptr", (size_t)(d + 2) * 8);
        if (!d_col || !d_sz || !d_off || !d_cls) {
            free(class_ptr);
            return SDPSR_OUT_OF_MEMORY;
        }
        st = h2d_sync(c, d_col, hcol.data(), (size_t)nb * 4);
        if (!st) st = h2d_sync(c, d_sz, hsz.data(), (size_t)nb * 4);
        if (!st) st = h2d_sync(c, d_off, hoff.data(), (size_t)nb * 8);
        if (!st) st = h2d_sync(c, d_cls, class_ptr, (size_t)(d + 2) * 8);
        free(class_ptr);
        if (st) return st;
        launch_basis_image_outer(s, n, d, S1, S, nb, max_s, Qrm, ent, d_cls, d_col, d_sz, d_off, atol, out);
    } else {
    // chunks + output descriptors
    const int64_t CH = 4096;
    std::vector<int64_t> chunk_ptr(d + 1, 0), cb, ce;
    for (int64_t i = 1; i <= d; ++i) {
        chunk_ptr[i - 1] = (int64_t)cb.size();
        for (int64_t p = class_ptr[i]; p < class_ptr[i + 1]; p += CH) {
            cb.push_back(p);
            ce.push_back(std::min(p + CH, class_ptr[i + 1]));
        }
    }
    chunk_ptr[d] = (int64_t)cb.size();
    free(class_ptr);
    std::vector<int32_t> dA(S), dB(S);
    {
        int64_t o = 0, colbase = 0;
        for (int32_t sz : c->bd_sizes) {
            for (int b = 0; b < sz; ++b)
                for (int a = 0; a < sz; ++a) {
                    dA[o] = (int32_t)(colbase + a);
                    dB[o] = (int32_t)(colbase + b);
                    ++o;
                }
            colbase += sz;
        }
    }
    const int64_t nch = (int64_t)cb.size();
    int64_t* d_chunk_ptr = (int64_t*)ctx_buf(c, "bi_chunk_ptr", (d + 1) * 8);
    int64_t* d_cb = (int64_t*)ctx_buf(c, "bi_cb", std::max<int64_t>(nch, 1) * 8);
    int64_t* d_ce = (int64_t*)ctx_buf(c, "bi_ce", std::max<int64_t>(nch, 1) * 8);
    int32_t* d_dA = (int32_t*)ctx_buf(c, "bi_da", std::max<int64_t>(S, 1) * 4);
    int32_t* d_dB = (int32_t*)ctx_buf(c, "bi_db", std::max<int64_t>(S, 1) * 4);
    double* partial = (double*)ctx_buf(c, "bi_partial", (size_t)std::max<int64_t>(nch * S, 1) * 8);
    if (!d_chunk_ptr || !d_cb || !d_ce || !d_dA || !d_dB || !partial) return SDPSR_OUT_OF_MEMORY;
    HIP_TRY(c, hipMemcpyAsync(d_chunk_ptr, chunk_ptr.data(), (d + 1) * 8, hipMemcpyHostToDevice, s));
    if (nch) {
        HIP_TRY(c, hipMemcpyAsync(d_cb, cb.data(), nch * 8, hipMemcpyHostToDevice, s));
        HIP_TRY(c, hipMemcpyAsync(d_ce, ce.data(), nch * 8, hipMemcpyHostToDevice, s));
    }
    if (S) {
        HIP_TRY(c, hipMemcpyAsync(d_dA, dA.data(), S * 4, hipMemcpyHostToDevice, s));
        HIP_TRY(c, hipMemcpyAsync(d_dB, dB.data(), S * 4, hipMemcpyHostToDevice, s));
    }
    launch_basis_image(s, n, d, S1, S, Qrm, ent, nullptr, d_dA, d_dB, d_chunk_ptr, nch, nullptr, d_cb, d_ce,
                       partial, out, atol);
    }
    }
    HIP_TRY(c, hipGetLastError());
    // ONE host wait for everything (the images, Q_hat, and the host vectors above, which must outlive their copies): the
    // copies are enqueued first.  (Three waits before: the second and third found an idle stream, ~3 us each.)
    if (mem != SDPSR_MEM_DEVICE) {
        HIP_TRY(c, hipMemcpyAsync(blks, out, (size_t)d * S * 8, hipMemcpyDeviceToHost, s));
        c->d2h_bytes += (size_t)d * S * 8;
    }
    if (Q_hat) {
        HIP_TRY(c, hipMemcpyAsync(Q_hat, Qhat, (size_t)n * S1 * 8, mem == SDPSR_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s));
        if (mem != SDPSR_MEM_DEVICE) c->d2h_bytes += (size_t)n * S1 * 8;
    }
    if (!(done_synced && mem == SDPSR_MEM_DEVICE && !Q_hat)) HIP_TRY(c, ctx_sync_stream(c, s));
    if (phase_ms) {
        const float ms = ev_total.stop(s);
        for (int i = 0; i < SDPSR_T_COUNT; ++i) phase_ms[i] = 0;
        phase_ms[SDPSR_T_IMAGE] = ms;
        phase_ms[SDPSR_T_TOTAL] = ms;
    }
    return SDPSR_OK;
}

}  // extern "C"
