// Module-compression driver of diagonalize (src/diagonalize.jl:25-40 on the restriction of the
// partition algebra to a cyclic module): see the block comment below and DESIGN.md.
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

// ===========================================================================
// Module-compression driver of diagonalize (DESIGN.md "module compression").
//
// For a random x the cyclic module M = <S> x (S = the partition subspace, <S> the associative
// algebra it generates) is invariant under every element of S and contains every irreducible
// constituent of the action with multiplicity min(s_k, m_k) >= 1, so Murota's algorithm run on
// the restriction S|_M (dimension w = sum_k s_k min(s_k, m_k) <= sum_k s_k^2 < 2 dim(P))
// finds the same blocks, and Q_hat = W * Q_hat_small (W an orthonormal basis of M) is a valid
// Q_hat of the full problem.  M is grown one vector at a time: y = A z for a fresh generic
// element A and a random z in the current span; y is appended if it leaves the span (CGS2);
// three consecutive misses end the growth.  Only first powers of well-scaled matrices are
// involved, so the rank decisions are sharp (eps vs O(1)).
// Cost: w passes over an n x n element + a dense w x w diagonalisation, against 4/3 n^3.
// ===========================================================================
int compressed_diagonalize(sdpsr_ctx* c, int64_t n, const uint32_t* L, int64_t d, double atol, EigInfo& info,
                           std::vector<int32_t>& sizes, int64_t& S1, int64_t& S, PhaseTimer& tm) {
    hipStream_t s = c->stream;
    const int64_t ld = round_up(n, 128);
    const int wmax = (int)std::min<int64_t>(std::min<int64_t>(n / 2, 500), 2 * d + 8);
    if (wmax < 2) return driver_fallback(c, "module too small to compress");
    const int64_t wcap = round_up(wmax + 2, 128);
    const int64_t ycap = 2 * wcap;                  // candidate columns of one round
    const int64_t wtot = wcap + ycap + 128;         // basis | candidates | padding of the last tile
    uint32_t* flag = (uint32_t*)ctx_buf(c, "bd_flag", 64);
    double* W = (double*)ctx_buf(c, "cm_w", (size_t)ld * wtot * 8);
    double* T = (double*)ctx_buf(c, "cm_t", (size_t)ld * wcap * 8);
    double* zy = (double*)ctx_buf(c, "cm_zy", (size_t)ld * 2 * 8);
    double* dout = (double*)ctx_buf(c, "cm_out", 64);
    if (!flag || !W || !T || !zy || !dout) return SDPSR_OUT_OF_MEMORY;
    // symmetric check: the verdict is copied back without a synchronisation of its own and is
    // looked at after the first read-back of the module growth (the kernels in between are
    // memory-safe for any labels, their results are simply discarded)
    if (!c->pinned_small) return SDPSR_OUT_OF_MEMORY;
    const bool sym_pre = c->bd_sym_epoch != 0 && c->bd_sym_labels == L;  // checked by the copy pass of blockDiagonalize
    const bool sym_trusted = c->bd_trusted_symmetric == L;               // labels of the library's own symmetric loop
    c->pinned_small[0] = 0;
    if (!sym_trusted) {
        if (!sym_pre) launch_check_symmetric(s, n, L, flag);
        HIP_TRY(c, hipMemcpyAsync(c->pinned_small, sym_pre ? (const uint32_t*)ctx_buf(c, "bd_symflag", 64) : flag, 4,
                                  hipMemcpyDeviceToHost, s));
    }
    bool sym_checked = sym_trusted;
    // Y <- A W for a fresh generic element A: fused label product when the shape allows it,
    // gather + split-K MFMA GEMM otherwise.  Columns >= wcols of dst keep their old content.
    auto apply_generic = [&](int wcols, double* dst) -> int {
        const uint64_t key = next_key(c);
        if (wcols <= 64 && d <= 4000) {
            double* part = (double*)ctx_buf(c, "cm_part", label_spmm_partial_doubles(n, 64) * 8);
            if (!part) return SDPSR_OUT_OF_MEMORY;
            if (launch_label_spmm(s, n, L, key, d, W, ld, wcols, part, dst, ld)) return SDPSR_OK;
        }
        double* Afull = (double*)ctx_buf(c, "cm_a", (size_t)ld * ld * 8);
        const int64_t wcp = round_up(wcols, 128);
        double* tmpo = (double*)ctx_buf(c, "cm_tmpo", (size_t)ld * wcp * 8);
        if (!Afull || !tmpo) return SDPSR_OUT_OF_MEMORY;
        launch_gather_f64_padded(s, n, ld, L, key, Afull);
        int e3 = gemm_tn_splitk(c, ld, wcp, ld, Afull, ld, W, ld, tmpo, ld);
        if (e3) return e3;
        HIP_TRY(c, hipMemcpyAsync(dst, tmpo, (size_t)ld * wcols * 8, hipMemcpyDeviceToDevice, s));
        return SDPSR_OK;
    };
    dbg_mark(c, "compressed: buffers + symmetric check done");
    tm.begin(SDPSR_T_EIGEN);
    HIP_TRY(c, hipMemsetAsync(W, 0, (size_t)ld * wtot * 8, s));
    double* y = zy;
    launch_random_vector(s, n, next_key(c), y);
    launch_normalize_columns(s, n, ld, W, 0, y, ld, 1, dout);  // W[:,0] = x / |x|
    int w = 1;
    // Block growth.  The candidates of a round (class sums of x, then A_g W for G fresh generic
    // elements) are written right behind the basis, Y = W[:, w : w+m), so that ONE split-K MFMA
    // product [W Y]' Y delivers both C = W'Y and the Gram matrix Y'Y.  On the host the Gram matrix
    // of the projected candidates is G - C'C (its cancellation error ~eps |Y|^2 sits four orders
    // below the rank threshold), a pivoted Cholesky factorisation picks the new directions
    // (rank gap: O(1) against eps^2, sharp because every candidate is a first power of a
    // well-scaled matrix) and Q1 = [W Y] [-C X; X] forms them in one pass.  A second product
    // [W Q1]' Q1 with the same structure re-orthonormalises (CholQR2).  A round that adds nothing
    // (the module is complete) therefore costs one product.
    double* Cc = (double*)ctx_buf(c, "cm_c", (size_t)(wcap + ycap + 128) * ycap * 8);
    double* dSm = (double*)ctx_buf(c, "cm_sm", (size_t)(wcap + ycap) * ycap * 8);
    double* Q1 = (double*)ctx_buf(c, "cm_q1", (size_t)ld * ycap * 8);
    if (!Cc || !dSm || !Q1) return SDPSR_OUT_OF_MEMORY;
    HIP_TRY(c, hipMemsetAsync(Q1, 0, (size_t)ld * ycap * 8, s));  // rows >= n stay zero for good
    std::vector<double> hG;
    // pivoted Cholesky of the m x m Gram matrix (leading dimension ldg): returns rank r, the
    // pivot order and X = R11^-1 scattered into an m x r coefficient matrix (column-major, ld m)
    double piv_max = 0, piv_min = 0;  // first / last accepted pivot of the last gram_select (diagonal pivoting: decreasing)
    auto gram_select = [&](const std::vector<double>& G, int64_t ldg, int m, double tol_abs, std::vector<double>& coef) -> int {
        std::vector<double> Gm((size_t)m * m);
        for (int j = 0; j < m; ++j)
            for (int i = 0; i < m; ++i) Gm[(size_t)i + (size_t)j * m] = 0.5 * (G[(size_t)i + (size_t)j * ldg] + G[(size_t)j + (size_t)i * ldg]);
        std::vector<int> perm(m);
        std::iota(perm.begin(), perm.end(), 0);
        std::vector<double> R((size_t)m * m, 0.0);
        int r = 0;
        for (int kk2 = 0; kk2 < m; ++kk2) {
            int p = kk2;
            for (int i = kk2 + 1; i < m; ++i)
                if (Gm[(size_t)i + (size_t)i * m] > Gm[(size_t)p + (size_t)p * m]) p = i;
            if (!(Gm[(size_t)p + (size_t)p * m] > tol_abs)) break;
            if (p != kk2) {
                for (int i = 0; i < m; ++i) std::swap(Gm[(size_t)i + (size_t)kk2 * m], Gm[(size_t)i + (size_t)p * m]);
                for (int j = 0; j < m; ++j) std::swap(Gm[(size_t)kk2 + (size_t)j * m], Gm[(size_t)p + (size_t)j * m]);
                for (int i = 0; i < kk2; ++i) std::swap(R[(size_t)i + (size_t)kk2 * m], R[(size_t)i + (size_t)p * m]);
                std::swap(perm[kk2], perm[p]);
            }
            const double rkk = std::sqrt(Gm[(size_t)kk2 + (size_t)kk2 * m]);
            if (kk2 == 0) piv_max = rkk * rkk;
            piv_min = rkk * rkk;
            R[(size_t)kk2 + (size_t)kk2 * m] = rkk;
            for (int j = kk2 + 1; j < m; ++j) R[(size_t)kk2 + (size_t)j * m] = Gm[(size_t)kk2 + (size_t)j * m] / rkk;
            for (int j = kk2 + 1; j < m; ++j) {
                const double rj = R[(size_t)kk2 + (size_t)j * m];
                for (int i = kk2 + 1; i <= j; ++i) {
                    Gm[(size_t)i + (size_t)j * m] -= R[(size_t)kk2 + (size_t)i * m] * rj;
                    Gm[(size_t)j + (size_t)i * m] = Gm[(size_t)i + (size_t)j * m];
                }
            }
            ++r;
        }
        // X = R11^-1 (upper triangular r x r), column by column
        std::vector<double> X((size_t)r * r, 0.0);
        for (int cc = 0; cc < r; ++cc) {
            for (int i = cc; i >= 0; --i) {
                double sum = (i == cc) ? 1.0 : 0.0;
                for (int t = i + 1; t <= cc; ++t) sum -= R[(size_t)i + (size_t)t * m] * X[(size_t)t + (size_t)cc * r];
                X[(size_t)i + (size_t)cc * r] = sum / R[(size_t)i + (size_t)i * m];
            }
        }
        coef.assign((size_t)m * std::max(r, 1), 0.0);
        for (int cc = 0; cc < r; ++cc)
            for (int i = 0; i <= cc; ++i) coef[(size_t)perm[i] + (size_t)cc * m] = X[(size_t)i + (size_t)cc * r];
        return r;
    };
    double ref = 0;  // squared scale of the current round's candidate columns before projection
    int abs_err = SDPSR_OK;
    std::function<void()> before_wait;  // enqueued between an orthogonalisation step's product and its host wait
    bool spec_b2 = false;               // the second compressed element was formed behind the final round's product ...
    size_t spec_off = 0;                // ... and sits at this byte offset of the pinned buffer
    // One orthonormalisation step on the mc columns behind the basis, V = W[:, w : w+mc):
    // product [W V]' V, projected Gram matrix on the host, selection X (mc x r); returns r and
    // the stacked coefficients S = [-C X; X] ((w+mc) x r) with V_new = [W V] S.  r < 0: error.
    auto ortho_step = [&](int mc, double tol_abs, bool take_ref, std::vector<double>& stacked) -> int {
        const int64_t ap = round_up(w + mc, 128), mp = round_up(mc, 128);
        // the exact-shape Gram kernel stores the product into pinned host memory as well: no copy launch before the read-back
        double* hpin = (double*)ctx_pinned(c, (size_t)ap * mp * 8);
        bool host_filled = false;
        abs_err = hpin ? gram_tn(c, w + mc, mc, ld, W, ld, W + (size_t)w * ld, ld, Cc, ap, mp, hpin, &host_filled) : SDPSR_OUT_OF_MEMORY;
        if (abs_err) return -1;
        if (before_wait) {  // work that rides on this step's host wait (one-shot)
            const std::function<void()> f = std::move(before_wait);
            before_wait = nullptr;
            f();
        }
        hG.resize((size_t)ap * mp);
        if (host_filled) {
            if (ctx_sync_stream(c, s) != hipSuccess) {
                abs_err = ctx_fail(c, SDPSR_HIP_ERROR, "hipStreamSynchronize (module growth)");
                return -1;
            }
            memcpy(hG.data(), hpin, (size_t)ap * mc * 8);
        } else {
            abs_err = d2h_sync(c, hG.data(), Cc, (size_t)ap * mc * 8);  // the mc columns the host looks at
            if (abs_err) return -1;
        }
        if (!sym_checked) {  // the stream has been synchronised: the verdict of the symmetric check is in
            sym_checked = true;
            if (sym_pre ? c->pinned_small[0] == c->bd_sym_epoch : c->pinned_small[0] != 0) {
                abs_err = ctx_fail(c, SDPSR_INVALID_DECOMPOSITION_FIELD,
                                   "partition is not symmetric: decomposition over Float64 requested but the generic element has a complex spectrum");
                return -1;
            }
        }
        // scale reference: THIS round's candidates BEFORE projection (after it, a complete module
        // leaves only rounding noise and a relative test would compare noise with noise).  The
        // cancellation error of G - C'C and the projection error are both ~eps |candidate|^2 of the
        // round at hand: class sums have |.|^2 ~ class valency, growth candidates A_g W up to ~n^2/4.
        if (take_ref) {
            ref = 0;
            for (int i = 0; i < mc; ++i) ref = std::max(ref, hG[(size_t)(w + i) + (size_t)i * ap]);
        }
        // rank 0 is decided by the largest diagonal entry of the projected Gram matrix alone (diagonal
        // pivoting: it is the first pivot): the invariance round of a complete module stops here, without
        // the mc^2 w products of the full matrix
        {
            double dmax = 0;
            for (int i = 0; i < mc; ++i) {
                double v = hG[(size_t)(w + i) + (size_t)i * ap];
                for (int t = 0; t < w; ++t) v -= hG[(size_t)t + (size_t)i * ap] * hG[(size_t)t + (size_t)i * ap];
                dmax = std::max(dmax, v);
            }
            if (!(dmax > (take_ref ? tol_abs * ref : tol_abs))) return 0;
        }
        std::vector<double> G1((size_t)mc * mc);
        for (int j = 0; j < mc; ++j)
            for (int i = 0; i < mc; ++i) {
                double v = hG[(size_t)(w + i) + (size_t)j * ap];
                for (int t = 0; t < w; ++t) v -= hG[(size_t)t + (size_t)i * ap] * hG[(size_t)t + (size_t)j * ap];
                G1[(size_t)i + (size_t)j * mc] = v;
            }
        std::vector<double> X;
        const int r = gram_select(G1, mc, mc, take_ref ? tol_abs * ref : tol_abs, X);
        if (r <= 0) return 0;
        stacked.assign((size_t)(w + mc) * r, 0.0);
        for (int cc = 0; cc < r; ++cc) {
            double* col = stacked.data() + (size_t)cc * (w + mc);
            for (int i = 0; i < mc; ++i) {
                const double xi = X[(size_t)i + (size_t)cc * mc];
                col[w + i] = xi;
                if (xi != 0.0)
                    for (int t = 0; t < w; ++t) col[t] -= hG[(size_t)t + (size_t)i * ap] * xi;
            }
        }
        return r;
    };
    // V_new = [W V] S into Q1, then back behind the basis (the old V is dead by then)
    auto apply_stacked = [&](int mc, int r, const std::vector<double>& stacked) -> int {
        int e2 = h2d_sync(c, dSm, stacked.data(), (size_t)(w + mc) * r * 8);
        if (e2) return e2;
        launch_tall_times_small(s, n, ld, W, w + mc, dSm, w + mc, r, 1.0, 0.0, Q1, ld);
        HIP_TRY(c, hipMemcpyAsync(W + (size_t)w * ld, Q1, (size_t)ld * r * 8, hipMemcpyDeviceToDevice, s));
        return SDPSR_OK;
    };
    // absorb m candidate columns W[:, w : w+m) into the basis; returns the number of new basis
    // vectors (0 = nothing left the span), < 0 on error (status in `abs_err`)
    bool first_product_intact = false;
    auto absorb = [&](int m) -> int {
        std::vector<double> st1, st2;
        // threshold 1e-10 |Y|^2: the projected Gram matrix comes from the cancellation G - C'C,
        // whose error is ~ w sqrt(n) eps |Y|^2 (~1e-13 at n = 4096: with 1e-12 about one
        // invariance round in ten let noise-level candidates through to the second step); a
        // genuine new direction of a generic element has an O(1) relative component
        const int r_new = ortho_step(m, 1e-10, true, st1);
        if (dbg_on() && r_new > 0) fprintf(stderr, "[sdpsr] absorb(%d): rank %d, pivots %.3e .. %.3e (ratio %.1e)\n", m, r_new, piv_max, piv_min, piv_max / piv_min);
        first_product_intact = (r_new == 0);  // Cc still holds [W Y]'Y (no second product ran)
        if (r_new <= 0) return r_new;
        if (w + r_new >= wmax) {
            abs_err = driver_fallback(c, "module dimension exceeds " + std::to_string(wmax));
            return -1;
        }
        abs_err = apply_stacked(m, r_new, st1);
        if (abs_err) return -1;
        // One Cholesky-based step leaves an orthogonality error of ~eps * cond(projected Gram) (and
        // the projection against W one of ~eps * |candidate|^2 / smallest pivot): with both ratios
        // below 1e3 that is < 1e-12 and the second step (another product, another host round trip,
        // ~80 us at N = 4096) adds nothing.  Measured ratios: 2e2-5e2 for the class sums of a
        // commutative scheme, 1e4-1e5 for the non-commutative growth rounds (those keep the second step).
        const double worst = std::max(piv_max, ref) / piv_min;
        if (worst <= 1e3 && !(c->opts.flags & SDPSR_FLAG_ALWAYS_REORTHOGONALIZE)) {
            w += r_new;
            return r_new;
        }
        const int r2 = ortho_step(r_new, 1e-6, false, st2);  // Q1 columns have unit scale
        if (r2 <= 0) return r2;
        abs_err = apply_stacked(r_new, r2, st2);
        if (abs_err) return -1;
        w += r2;
        return r2;
    };
    // level 1: S x = span{P_i x}: all d class sums of x in ONE pass over the labels (the
    // row-sum kernel of basis_image with a single column).  For a commutative algebra this
    // already is the whole module.
    if (class_sums_supports(n, d, ld) && basis_image_two_stage_fits(n, d, 1) && d <= ycap && d + 1 < wmax) {
        launch_class_sums(s, n, d, L, W, W + (size_t)w * ld, ld);  // W[:, w + i] = P_{i+1} x
        const int got = absorb((int)d);
        if (got < 0) {
            tm.end();
            tm.collect();
            return abs_err;
        }
    }
    bool have_saved = false;
    bool module_complete = false;  // an invariance round added nothing
    int64_t saved_ld = 0;
    for (int round = 0; round < 40; ++round) {
        const bool fused = (w <= 64 && d <= 4000);
        int G = (fused && w <= 16) ? 4 : 2;  // generic elements per round
        while (G > 2 && (int64_t)G * w > ycap) --G;
        // dim <S> x <= dim S = d: once w has reached d the round is (almost surely) only the
        // invariance check, for which ONE generic element suffices (the elements that map the
        // module into itself form a subspace of S; it contains a generic point iff it is S)
        if ((int64_t)G * w > ycap || w >= d) G = 1;
        const int m = G * w;
        // candidates: W[:, w + g*w + (0:w)] = A_g W for fresh generic elements A_g (rows >= n zero)
        double* Y = W + (size_t)w * ld;
        HIP_TRY(c, hipMemsetAsync(Y, 0, (size_t)ld * round_up(m, 128) * 8, s));
        bool batched = false;
        if (fused && G > 1 && G != 3 && G * w <= 64 && !(c->opts.flags & SDPSR_FLAG_SPMM_ONE_BY_ONE)) {
            // the G generic elements of the round in one pass over the labels
            double* part = (double*)ctx_buf(c, "cm_part", label_spmm_partial_doubles(n, 64) * 8);
            if (!part) return SDPSR_OUT_OF_MEMORY;
            uint64_t keys[4];
            const uint64_t save = c->stream_counter;
            for (int gidx = 0; gidx < G; ++gidx) keys[gidx] = next_key(c);
            batched = launch_label_spmm_multi(s, n, L, keys, G, d, W, ld, w, part, Y, ld);
            if (!batched) c->stream_counter = save;
        }
        for (int gidx = 0; gidx < G && !batched; ++gidx) {
            int e2 = apply_generic(w, Y + (size_t)gidx * w * ld);
            if (e2) return e2;
        }
        // A round with ONE element is the invariance check of a module that is (almost surely) complete, and the host path of
        // the small problem will then want the second compressed element B2 = W'A2 W right away: it is formed behind this
        // round's product and comes back with the same host wait (its pinned words sit behind the round's Gram matrix).
        // If the round does add a direction, the element is dropped.
        spec_b2 = false;
        if (G == 1 && w <= 64 && !(c->opts.flags & SDPSR_FLAG_SMALL_EIGEN_ON_DEVICE) && (c->opts.eig_driver == 0 || c->opts.eig_driver >= 4)) {
            const int ws = w;
            spec_off = ((size_t)round_up(ws + m, 128) * (size_t)round_up(m, 128) * 8 + 255) & ~size_t(255);
            // everything sized BEFORE the round's own requests: no buffer moves between the product and the wait
            char* pinb = (char*)ctx_pinned(c, spec_off + (size_t)ws * ws * 8);
            double* dBs = (double*)ctx_buf(c, "cm_bsmall", (size_t)64 * 64 * 8);
            // (the largest request the round's own product can make: the pointer handed out here stays valid)
            const size_t gpb = gram_small_partial_doubles(n, 128, 128) * 8;
            double* gps = (double*)ctx_buf(c, "gram_partials", gpb);
            if (pinb && dBs && gps)
                before_wait = [&, ws, dBs, gps]() {
                    if (apply_generic(ws, T) != SDPSR_OK) return;  // T[:, 0:ws) = A2 W
                    launch_gram_small(s, n, ws, ws, W, ld, T, ld, gps, dBs, ws, ws, ws, (double*)((char*)c->pinned + spec_off));
                    spec_b2 = true;
                };
        }
        const int got = absorb(m);
        before_wait = nullptr;
        if (got < 0) {
            tm.end();
            tm.collect();
            return abs_err;
        }
        if (got != 0) spec_b2 = false;
        if (got == 0) {
            module_complete = true;
            // the module is complete: the top block of this round's product, C = W' (A W), IS the
            // compressed generic element W' A W of the round's first element -- keep it for the
            // eigen stage instead of forming another one (one label product + one GEMM saved)
            saved_ld = round_up(w + m, 128);
            have_saved = first_product_intact;  // not if noise-level candidates went through the second step
            break;
        }
    }
    if (!module_complete) {  // never diagonalise a module that no round has confirmed invariant
        tm.end();
        tm.collect();
        return driver_fallback(c, "module growth did not close within 40 rounds");
    }
    // ---- small modules (w <= 64): Murota's steps on the host (small_eigen_host.cpp) ----
    // The first compressed generic element B1 = W'A1 W is the top block of the final invariance
    // round's product, which the host already holds (hG); the device forms B2 = W'A2 W while the
    // host diagonalises B1; after one w x w read-back everything up to Q_hat_small (w x S1) is host
    // arithmetic on a few KiB, and one upload + one tall product lift it: Q_hat = W Q_hat_small.
    if (w <= 64 && !(c->opts.flags & SDPSR_FLAG_SMALL_EIGEN_ON_DEVICE) && (c->opts.eig_driver == 0 || c->opts.eig_driver >= 4)) {
        tm.end();
        tm.collect();
        tm.begin(SDPSR_T_ISO);
        const size_t bbytes = (size_t)w * w * 8;
        double* dB = (double*)ctx_buf(c, "cm_bsmall", bbytes);
        double* gp = (double*)ctx_buf(c, "gram_partials", gram_small_partial_doubles(n, w, w) * 8);
        double* pin = (double*)ctx_pinned(c, bbytes);
        if (!dB || !gp || !pin) return SDPSR_OUT_OF_MEMORY;
        // B = W'(A W) for a fresh generic element, compact w x w, copied towards the pinned buffer; no host wait
        auto enqueue_element = [&]() -> int {
            const int e2 = apply_generic(w, T);  // T[:, 0:w) = A W (rows < n)
            if (e2) return e2;
            launch_gram_small(s, n, w, w, W, ld, T, ld, gp, dB, w, w, w, pin);  // product stored into the pinned buffer itself
            return SDPSR_OK;
        };
        auto fetch = [&](double* dst) -> int {  // waits for the enqueued element, symmetrised like _symmetrize!
            HIP_TRY(c, ctx_sync_stream(c, s));
            for (int j = 0; j < w; ++j)
                for (int i = 0; i < w; ++i) dst[(size_t)i + (size_t)j * w] = 0.5 * (pin[(size_t)i + (size_t)j * w] + pin[(size_t)j + (size_t)i * w]);
            return SDPSR_OK;
        };
        std::vector<double> B1((size_t)w * w);
        if (have_saved) {
            for (int j = 0; j < w; ++j)
                for (int i = 0; i < w; ++i)
                    B1[(size_t)i + (size_t)j * w] = 0.5 * (hG[(size_t)i + (size_t)j * saved_ld] + hG[(size_t)j + (size_t)i * saved_ld]);
        } else {
            int e2 = enqueue_element();
            if (!e2) e2 = fetch(B1.data());
            if (e2) return e2;
        }
        bool spec_avail = spec_b2 && have_saved;  // B2 came back with the final round's product: no launch, no wait
        bool pending = spec_avail ? false : enqueue_element() == SDPSR_OK;  // else B2 is formed while the host diagonalises B1
        const std::function<int(double*)> next_element = [&](double* dst) -> int {
            if (spec_avail) {
                spec_avail = false;
                const double* ps = (const double*)((const char*)c->pinned + spec_off);
                for (int j = 0; j < w; ++j)
                    for (int i = 0; i < w; ++i) dst[(size_t)i + (size_t)j * w] = 0.5 * (ps[(size_t)i + (size_t)j * w] + ps[(size_t)j + (size_t)i * w]);
                return SDPSR_OK;
            }
            if (!pending) {
                const int e2 = enqueue_element();
                if (e2) return e2;
            }
            pending = false;
            return fetch(dst);
        };
        std::vector<double> Qs;
        int st = murota_small_host(c, w, B1.data(), next_element, atol, d, sizes, S1, S, Qs);
        if (pending) ctx_sync_stream(c, s);  // never leave a copy into the pinned buffer in flight
        tm.end();
        if (st) {
            tm.collect();
            return st;
        }
        tm.begin(SDPSR_T_IRRED);
        double* qs = (double*)ctx_buf(c, "cm_qsmall", (size_t)w * S1 * 8);
        double* Qhat = (double*)ctx_buf(c, "bd_qhat", (size_t)n * S1 * 8);
        if (!qs || !Qhat) return SDPSR_OUT_OF_MEMORY;
        st = h2d_sync(c, qs, Qs.data(), (size_t)w * S1 * 8);
        if (st) return st;
        launch_tall_times_small(s, n, ld, W, w, qs, w, (int)S1, 1.0, 0.0, Qhat, n);
        launch_clamptol(s, n * S1, Qhat, atol);  // src/diagonalize.jl:39
        tm.end();
        HIP_TRY(c, hipGetLastError());  // no host wait here: the caller synchronises (or keeps enqueueing: sdpsr_jordan_reduce)
        dbg_mark(c, "compressed: small problem solved on the host, lift enqueued");
        return SDPSR_OK;
    }
    // columns >= w must be zero for the padded products below
    const int64_t wp = round_up(w, 128);
    HIP_TRY(c, hipMemsetAsync(W + (size_t)w * ld, 0, (size_t)ld * (wtot - w) * 8, s));
    tm.end();
    tm.collect();
    if (dbg_on()) fprintf(stderr, "[sdpsr] module compression: n=%lld dim(P)=%lld -> w=%d\n", (long long)n, (long long)d, w);

    dbg_mark(c, "compressed: module grown");
    ElemGen gen;
    gen.make = [&](double* dst) -> int {
        if (have_saved) {  // first element: the product of the final invariance round (see above)
            have_saved = false;
            launch_extract_symmetric(s, w, wp, Cc, saved_ld, dst);  // zero padding + copy + symmetrize in one launch
            return SDPSR_OK;
        }
        HIP_TRY(c, hipMemsetAsync(T, 0, (size_t)ld * wp * 8, s));
        { int e2 = apply_generic(w, T); if (e2) return e2; }   // T = A W
        gram_tn(c, w, w, ld, W, ld, T, ld, dst, wp, wp);  // B = W' T  (w x w in wp x wp)
        launch_symmetrize(s, w, wp, dst);
        return SDPSR_OK;
    };
    bool forked = false;
    auto ensure_side = [&]() -> int {
        return ctx_ensure_side(c) ? SDPSR_OK : SDPSR_HIP_ERROR;
    };
    gen.fork = [&]() -> int {
        forked = false;
        if (have_saved) return SDPSR_BAD_STATE;  // the next element is the saved one: nothing to overlap
        if (ensure_side()) return SDPSR_HIP_ERROR;
        if (hipEventRecord(c->ev_fork, c->stream) != hipSuccess) return SDPSR_HIP_ERROR;
        forked = true;
        return SDPSR_OK;
    };
    gen.prefetch = [&](double* dst) -> int {
        if (ensure_side()) return SDPSR_HIP_ERROR;
        if (have_saved) return SDPSR_BAD_STATE;  // the next element is the saved one: nothing to overlap
        hipStream_t main_stream = c->stream;
        if (!forked && hipEventRecord(c->ev_fork, main_stream) != hipSuccess) return SDPSR_HIP_ERROR;
        forked = false;
        if (hipStreamWaitEvent(c->side_stream, c->ev_fork, 0) != hipSuccess) return SDPSR_HIP_ERROR;
        c->stream = c->side_stream;  // every helper launches on c->stream / s
        c->main_shadow = main_stream;
        s = c->side_stream;
        const int e2 = gen.make(dst);
        const bool rec = hipEventRecord(c->ev_join, c->side_stream) == hipSuccess;
        c->stream = main_stream;
        c->main_shadow = nullptr;
        s = main_stream;
        if (e2 || !rec) {
            ctx_sync_stream(c, c->side_stream);
            return e2 ? e2 : SDPSR_HIP_ERROR;
        }
        return SDPSR_OK;
    };
    gen.join = [&]() -> int {
        HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));
        return SDPSR_OK;
    };
    if (!ctx_buf(c, "bd_qhat", (size_t)n * wmax * 8)) return SDPSR_OUT_OF_MEMORY;  // final size now: no move later
    int st = dense_diagonalize(c, w, nullptr, &gen, atol, info, sizes, S1, S, tm, d);
    if (st) return st;
    dbg_mark(c, "compressed: small dense diagonalize done");
    // lift: Q_hat = W * Q_hat_small
    tm.begin(SDPSR_T_IRRED);
    // "bd_qhat" was sized for n x wmax before the small problem ran (S1 <= w < wmax), so the
    // small Q_hat sits at its start and the buffer does not move here: stream order suffices
    double* qs = (double*)ctx_buf(c, "cm_qsmall", (size_t)w * S1 * 8);
    double* Qhat = (double*)ctx_buf(c, "bd_qhat", (size_t)n * S1 * 8);
    if (!qs || !Qhat) return SDPSR_OUT_OF_MEMORY;
    HIP_TRY(c, hipMemcpyAsync(qs, Qhat, (size_t)w * S1 * 8, hipMemcpyDeviceToDevice, s));
    launch_tall_times_small(s, n, ld, W, w, qs, w, (int)S1, 1.0, 0.0, Qhat, n);
    launch_clamptol(s, n * S1, Qhat, atol);
    tm.end();
    HIP_TRY(c, ctx_sync_stream(c, s));
    HIP_TRY(c, hipGetLastError());
    dbg_mark(c, "compressed: lifted");
    return SDPSR_OK;
}

// eig_driver: 0 auto (module compression when dim(P) is small against n, dense otherwise),
// 4 dense forced, 6 module compression forced, 1-3 rocSOLVER variants (comparison only).
bool compression_eligible(const sdpsr_ctx* c, int64_t n, int64_t d) {
    if (c->opts.eig_driver == 6) return true;
    if (c->opts.eig_driver != 0) return false;
    return n >= 512 && 2 * d + 8 <= std::min<int64_t>(n / 2, 500);
}

}  // namespace sdpsr
