// blockDiagonalize(P; complex = true) (src/compat.jl:26-32,46-68 with T = ComplexF64;
// src/diagonalize.jl:13-28): Hermitian generic elements, orders n > 64 through the real embedding.
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

// ---- complex path, orders n > 64: Hermitian eigendecomposition through the real embedding ----------
// M(H) (2n x 2n, symmetric) goes through the real dense eigensolver; every eigenvalue of H shows
// up twice and a cluster of 2m real eigenvectors (x; y) spans, read as z = x + iy, the m-dimensional
// complex eigenspace.  The m orthonormal complex vectors are picked by a pivoted Cholesky
// factorisation of the cluster's Gram matrix  G = Z^H Z = I + i (X'Y - Y'X)  (eigenvalues 0 and 2:
// rank m, perfectly conditioned) on the host; the combinations run on the device.
static int cx_heev_general(sdpsr_ctx* c, int64_t n, const double* Hr, const double* Hi, double atol, double* Vr, double* Vi,
                           std::vector<double>& vals) {
    hipStream_t s = c->stream;
    const int64_t n2 = 2 * n, ld2 = round_up(n2, 128);
    double* M = (double*)ctx_buf(c, "bdc_m", (size_t)ld2 * ld2 * 8);
    double* R = (double*)ctx_buf(c, "bdc_r", (size_t)ld2 * ld2 * 8);
    double* K = (double*)ctx_buf(c, "bdc_k", (size_t)ld2 * ld2 * 8);
    double* w2 = (double*)ctx_buf(c, "bdc_w2", (size_t)n2 * 8);
    if (!M || !R || !K || !w2) return SDPSR_OUT_OF_MEMORY;
    launch_cx_embed(s, n, Hr, Hi, ld2, M);
    std::vector<double> hw((size_t)n2);
    int st = syev_device(c, n2, M, ld2, w2, hw.data());
    if (st) return st;
    launch_cx_zero_pad(s, n2, n2, ld2, ld2, M);  // whatever the solver left in the padding
    HIP_TRY(c, hipMemsetAsync(R, 0, (size_t)ld2 * ld2 * 8, s));
    launch_cx_rot(s, n, n2, ld2, M, R);
    launch_gemm_tn_f64(s, ld2, ld2, ld2, M, ld2, R, ld2, K, ld2, 1, 0, 0, 0);  // K = X'Y - Y'X
    HIP_TRY(c, hipGetLastError());
    std::vector<double> hK((size_t)ld2 * ld2);
    st = d2h_sync(c, hK.data(), K, hK.size() * 8);
    if (st) return st;
    typedef std::complex<double> cd;
    std::vector<int32_t> desc;
    std::vector<double> coef;
    vals.clear();
    int64_t o = 0;
    while (o < n2) {
        int64_t e = o + 1;
        while (e < n2 && std::fabs(hw[e] - hw[e - 1]) <= atol) ++e;
        const int sz = (int)(e - o);
        if (sz % 2 != 0)
            return ctx_fail(c, SDPSR_NUMERICAL_INCONSISTENCY,
                            "complex path: an eigenvalue cluster of the embedded element has odd size (decrease `atol`, or try again)");
        const int m = sz / 2;
        std::vector<cd> G((size_t)sz * sz), Rf((size_t)sz * sz, cd(0, 0));
        for (int b = 0; b < sz; ++b)
            for (int a = 0; a < sz; ++a) G[(size_t)a + (size_t)b * sz] = cd(a == b ? 1.0 : 0.0, hK[(size_t)(o + a) + (size_t)(o + b) * ld2]);
        std::vector<int> perm(sz);
        for (int i = 0; i < sz; ++i) perm[i] = i;
        int r = 0;
        for (int k = 0; k < sz; ++k) {
            int p = k;
            double best = G[(size_t)k + (size_t)k * sz].real();
            for (int j = k + 1; j < sz; ++j)
                if (G[(size_t)j + (size_t)j * sz].real() > best) best = G[(size_t)j + (size_t)j * sz].real(), p = j;
            if (!(best > 1e-8)) break;
            if (p != k) {
                for (int i = 0; i < sz; ++i) std::swap(G[(size_t)i + (size_t)k * sz], G[(size_t)i + (size_t)p * sz]);
                for (int j = 0; j < sz; ++j) std::swap(G[(size_t)k + (size_t)j * sz], G[(size_t)p + (size_t)j * sz]);
                for (int i = 0; i < k; ++i) std::swap(Rf[(size_t)i + (size_t)k * sz], Rf[(size_t)i + (size_t)p * sz]);
                std::swap(perm[k], perm[p]);
            }
            const double rkk = std::sqrt(G[(size_t)k + (size_t)k * sz].real());
            Rf[(size_t)k + (size_t)k * sz] = rkk;
            for (int j = k + 1; j < sz; ++j) Rf[(size_t)k + (size_t)j * sz] = G[(size_t)k + (size_t)j * sz] / rkk;
            for (int j = k + 1; j < sz; ++j)
                for (int i = k + 1; i < sz; ++i)  // G22 -= r' conj(r): G = R^H R
                    G[(size_t)i + (size_t)j * sz] -= std::conj(Rf[(size_t)k + (size_t)i * sz]) * Rf[(size_t)k + (size_t)j * sz];
            ++r;
        }
        if (r != m)
            return ctx_fail(c, SDPSR_NUMERICAL_INCONSISTENCY,
                            "complex path: eigenspace extraction found rank " + std::to_string(r) + " in a cluster of " +
                                std::to_string(sz) + " embedded eigenvectors (decrease `atol`, or try again)");
        // X = R11^{-1} (upper triangular m x m); column a of the coefficients = P[:, :m] X[:, a]
        std::vector<cd> X((size_t)m * m, cd(0, 0));
        for (int cc = 0; cc < m; ++cc)
            for (int i = cc; i >= 0; --i) {
                cd sum = (i == cc) ? cd(1, 0) : cd(0, 0);
                for (int t = i + 1; t <= cc; ++t) sum -= Rf[(size_t)i + (size_t)t * sz] * X[(size_t)t + (size_t)cc * m];
                X[(size_t)i + (size_t)cc * m] = sum / Rf[(size_t)i + (size_t)i * sz];
            }
        for (int a = 0; a < m; ++a) {
            const int32_t cof = (int32_t)(coef.size() / 2);
            std::vector<cd> col(sz, cd(0, 0));
            for (int i = 0; i <= a; ++i) col[perm[i]] = X[(size_t)i + (size_t)a * m];
            for (int b = 0; b < sz; ++b) {
                coef.push_back(col[b].real());
                coef.push_back(col[b].imag());
            }
            const int32_t dsc[3] = {(int32_t)o, (int32_t)sz, cof};
            desc.insert(desc.end(), dsc, dsc + 3);
            vals.push_back(hw[(size_t)o + 2 * (size_t)a]);
        }
        o = e;
    }
    if ((int64_t)vals.size() != n) return ctx_fail(c, SDPSR_NUMERICAL_INCONSISTENCY, "complex path: eigenvector count mismatch");
    int32_t* ddesc = (int32_t*)ctx_buf(c, "bdc_cdesc", desc.size() * 4);
    double* dcoef = (double*)ctx_buf(c, "bdc_coef", coef.size() * 8);
    if (!ddesc || !dcoef) return SDPSR_OUT_OF_MEMORY;
    st = h2d_sync(c, ddesc, desc.data(), desc.size() * 4);
    if (!st) st = h2d_sync(c, dcoef, coef.data(), coef.size() * 8);
    if (st) return st;
    launch_cx_combine(s, n, ld2, M, ddesc, dcoef, Vr, Vi);
    HIP_TRY(c, hipGetLastError());
    return SDPSR_OK;
}

// max |(V^H H V)[a, b]| over the pairs of eigenspaces, n > 64: three real MFMA GEMMs on the embedding
static int cx_block_norms_general(sdpsr_ctx* c, int64_t n, const double* Hr, const double* Hi, const double* Vr, const double* Vi,
                                  const int32_t* dspace, int neig, unsigned long long* dnorms) {
    hipStream_t s = c->stream;
    const int64_t ld2 = round_up(2 * n, 128), ldn = round_up(n, 128);
    double* M = (double*)ctx_buf(c, "bdc_m", (size_t)ld2 * ld2 * 8);
    double* E = (double*)ctx_buf(c, "bdc_e", (size_t)ld2 * ldn * 8);
    double* T = (double*)ctx_buf(c, "bdc_t", (size_t)ld2 * ldn * 8);
    double* T2 = (double*)ctx_buf(c, "bdc_t2", (size_t)ld2 * ldn * 8);
    double* Gr = (double*)ctx_buf(c, "bdc_gr", (size_t)ldn * ldn * 8);
    double* Gi = (double*)ctx_buf(c, "bdc_gi", (size_t)ldn * ldn * 8);
    if (!M || !E || !T || !T2 || !Gr || !Gi) return SDPSR_OUT_OF_MEMORY;
    launch_cx_embed(s, n, Hr, Hi, ld2, M);
    HIP_TRY(c, hipMemsetAsync(E, 0, (size_t)ld2 * ldn * 8, s));
    launch_cx_stack(s, n, n, Vr, Vi, n, ld2, E);
    launch_gemm_tn_f64(s, ld2, ldn, ld2, M, ld2, E, ld2, T, ld2, 1, 0, 0, 0);  // T = M(H)' E = M(H) E: [Re(HV); Im(HV)]
    HIP_TRY(c, hipMemsetAsync(T2, 0, (size_t)ld2 * ldn * 8, s));
    launch_cx_rot(s, n, n, ld2, T, T2);                                          // [Im; -Re]
    launch_gemm_tn_f64(s, ldn, ldn, ld2, E, ld2, T, ld2, Gr, ldn, 1, 0, 0, 0);   // Re(V^H H V)
    launch_gemm_tn_f64(s, ldn, ldn, ld2, E, ld2, T2, ld2, Gi, ldn, 1, 0, 0, 0);  // Im(V^H H V)
    launch_cx_block_norms_general(s, n, ldn, Gr, Gi, dspace, neig, dnorms);
    HIP_TRY(c, hipGetLastError());
    return SDPSR_OK;
}

// ---- blockDiagonalize over C (src/compat.jl:26-32,46-68 with T = ComplexF64) ----------------
extern "C" {

int sdpsr_block_diagonalize_complex(sdpsr_ctx* c, int64_t n, const uint32_t* P, int64_t d, double epsilon,
                                    uint32_t* P_desym, int64_t* d_desym, int32_t* nblocks, int64_t* sum_sq,
                                    int64_t* sum_s, int mem) {
    CHECK_CTX(c);
    if (!P || n < 1 || d < 0 || !(epsilon > 0)) return ctx_fail(c, SDPSR_BAD_ARGUMENT, "bad arguments");
    if (n > 4096) return ctx_fail(c, SDPSR_BAD_ARGUMENT, "complex path: this version covers n <= 4096 (see sdpsr.h)");
    const bool small = n <= 64;  // one-workgroup kernels; larger orders go through the real embedding
    const int64_t len = n * n;
    hipStream_t s = c->stream;
    c->bdc_valid = false;
    // diagonalize(ComplexF64, P) desymmetrizes first (src/diagonalize.jl:26-28)
    uint32_t* L = (uint32_t*)ctx_buf(c, "bdc_labels", len * 4);
    if (!L) return SDPSR_OUT_OF_MEMORY;
    HIP_TRY(c, hipMemcpyAsync(L, P, len * 4, mem == SDPSR_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
    int64_t dd = d;
    int st = sdpsr_desymmetrize(c, n, L, &dd, nullptr, SDPSR_MEM_DEVICE);
    if (st) return st;
    const double atol = epsilon;
    double* Hr = (double*)ctx_buf(c, "bdc_hr", len * 8);
    double* Hi = (double*)ctx_buf(c, "bdc_hi", len * 8);
    double* Vr = (double*)ctx_buf(c, "bdc_vr", len * 8);
    double* Vi = (double*)ctx_buf(c, "bdc_vi", len * 8);
    double* w = (double*)ctx_buf(c, "bdc_w", n * 8);
    int* info = (int*)ctx_buf(c, "eig_info", 64);
    if (!Hr || !Hi || !Vr || !Vi || !w || !info) return SDPSR_OUT_OF_MEMORY;
    // Step 1-2: Hermitian generic element and its eigendecomposition (src/eigen_decomposition.jl:242-254)
    launch_cx_gather_herm(s, n, L, next_key(c), Hr, Hi);
    EigInfo ei;
    if (small) {
        launch_cx_heev(s, n, Hr, Hi, w, Vr, Vi, info);
        ei.vals.resize(n);
        int hinfo[2] = {0, 0};
        st = d2h_sync(c, ei.vals.data(), w, n * 8);
        if (!st) st = d2h_sync(c, hinfo, info, 8);
        if (st) return st;
        if (hinfo[0]) return ctx_fail(c, SDPSR_SOLVER_ERROR, "Hermitian Jacobi eigensolver did not converge");
    } else {
        st = cx_heev_general(c, n, Hr, Hi, atol, Vr, Vi, ei.vals);
        if (st) return st;
    }
    ei.ptrs.assign(1, 0);
    for (int64_t i = 0; i < n; ++i) {
        if (i == n - 1) {
            ei.ptrs.push_back((int)n);
            break;
        }
        if (!(std::fabs(ei.vals[i + 1] - ei.vals[i]) <= atol)) ei.ptrs.push_back((int)i + 1);
    }
    const int neig = (int)ei.ptrs.size() - 1;
    std::vector<int32_t> space_of(n);
    for (int b = 0; b < neig; ++b)
        for (int i = ei.ptrs[b]; i < ei.ptrs[b + 1]; ++i) space_of[i] = b;
    // Step 3: second generic element, Q'AQ, block norms, isomorphism classes (:259-262, :201-217)
    int32_t* dspace = (int32_t*)ctx_buf(c, "bd_space", (size_t)n * 4);
    unsigned long long* dnorms = (unsigned long long*)ctx_buf(c, "bd_norms", (size_t)neig * neig * 8);
    if (!dspace || !dnorms) return SDPSR_OUT_OF_MEMORY;
    st = h2d_sync(c, dspace, space_of.data(), n * 4);
    if (st) return st;
    HIP_TRY(c, hipMemsetAsync(dnorms, 0, (size_t)neig * neig * 8, s));
    launch_cx_gather_herm(s, n, L, next_key(c), Hr, Hi);
    if (small) launch_cx_block_norms(s, n, Hr, Hi, Vr, Vi, dspace, neig, dnorms);
    else {
        st = cx_block_norms_general(c, n, Hr, Hi, Vr, Vi, dspace, neig, dnorms);
        if (st) return st;
    }
    std::vector<double> norms((size_t)neig * neig);
    st = d2h_sync(c, norms.data(), dnorms, (size_t)neig * neig * 8);
    if (st) return st;
    auto dimof = [&](int b) { return ei.ptrs[b + 1] - ei.ptrs[b]; };
    for (int i = 0; i < neig; ++i)
        for (int j = i; j < neig; ++j) {
            const double v = (dimof(i) != dimof(j)) ? 0.0 : norms[(size_t)i * neig + j];
            norms[(size_t)i * neig + j] = norms[(size_t)j * neig + i] = v;
        }
    st = isomorphism_classes(c, norms, neig, atol, ei.kpart);
    if (st) return st;
    // irreducible_decomposition (:295-348)
    std::vector<int> roots;
    std::vector<std::vector<int>> members;
    class_structure(ei.kpart, roots, members);
    std::vector<int32_t> sizes(roots.size());
    int64_t S1 = 0, S = 0;
    std::vector<int32_t> desc;
    for (size_t p = 0; p < roots.size(); ++p) {
        sizes[p] = (int32_t)members[p].size();
        const int i = roots[p];
        for (size_t q = 0; q < members[p].size(); ++q) {
            const int j = members[p][q];
            const int32_t dsc[6] = {q == 0 ? 0 : 1, (int32_t)ei.ptrs[i], (int32_t)dimof(i), (int32_t)ei.ptrs[j], (int32_t)dimof(j),
                                    (int32_t)(S1 + (int64_t)q)};
            desc.insert(desc.end(), dsc, dsc + 6);
        }
        S1 += sizes[p];
        S += (int64_t)sizes[p] * sizes[p];
    }
    double* Qhat = (double*)ctx_buf(c, "bdc_qhat", (size_t)2 * n * S1 * 8);
    int32_t* ddesc = (int32_t*)ctx_buf(c, "bdc_desc", desc.size() * 4);
    if (!Qhat || !ddesc) return SDPSR_OUT_OF_MEMORY;
    st = h2d_sync(c, ddesc, desc.data(), desc.size() * 4);
    if (st) return st;
    launch_cx_gather_herm(s, n, L, next_key(c), Hr, Hi);  // generic element #3 (:306)
    if (small) launch_cx_irreducible(s, n, Hr, Hi, Vr, Vi, ddesc, (int)S1, atol, Qhat);
    else launch_cx_irreducible_general(s, n, Hr, Hi, Vr, Vi, ddesc, (int)S1, atol, Qhat);
    HIP_TRY(c, hipGetLastError());
    if (P_desym) HIP_TRY(c, hipMemcpyAsync(P_desym, L, len * 4, mem == SDPSR_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s));
    HIP_TRY(c, ctx_sync_stream(c, s));
    c->bdc_n = n;
    c->bdc_d = dd;
    c->bdc_sizes = sizes;
    c->bdc_sum_s = S1;
    c->bdc_sum_sq = S;
    if (d_desym) *d_desym = dd;
    if (nblocks) *nblocks = (int32_t)sizes.size();
    if (sum_sq) *sum_sq = S;
    if (sum_s) *sum_s = S1;
    // check_block_sizes over C: sum s^2 == dim(P) (src/diagonalize.jl:13-23)
    if (S != dd) {
        std::string szs;
        for (int32_t sz : sizes) szs += std::to_string(sz) + " ";
        return ctx_fail(c, SDPSR_DIMENSION_MISMATCH, "final_dim=" + std::to_string(S) + " block_sizes=[" + szs + "] expected dim(P)=" +
                                                         std::to_string(dd) + " over ComplexF64 (rounding error: try another epsilon or try again)");
    }
    c->bdc_valid = true;
    return SDPSR_OK;
}

int sdpsr_block_sizes_complex(sdpsr_ctx* c, int32_t* blk_sizes) {
    if (!c || !blk_sizes) return SDPSR_BAD_ARGUMENT;
    if (c->bdc_sizes.empty()) return ctx_fail(c, SDPSR_BAD_STATE, "no complex block diagonalisation available");
    memcpy(blk_sizes, c->bdc_sizes.data(), c->bdc_sizes.size() * sizeof(int32_t));
    return SDPSR_OK;
}

int sdpsr_block_images_complex(sdpsr_ctx* c, double* blks, double* Q_hat, int mem) {
    CHECK_CTX(c);
    if (!c->bdc_valid) return ctx_fail(c, SDPSR_BAD_STATE, "sdpsr_block_diagonalize_complex has not succeeded on this ctx");
    if (!blks) return ctx_fail(c, SDPSR_BAD_ARGUMENT, "null pointer");
    hipStream_t s = c->stream;
    const int64_t n = c->bdc_n, d = c->bdc_d, S1 = c->bdc_sum_s, S = c->bdc_sum_sq;
    uint32_t* L = (uint32_t*)ctx_buf(c, "bdc_labels", n * n * 4);
    double* Qhat = (double*)ctx_buf(c, "bdc_qhat", (size_t)2 * n * S1 * 8);
    int st = SDPSR_OK;
    double* out = out_dev(c, "bdc_blks", blks, (size_t)2 * d * S, mem, &st);
    int32_t* ddesc = (int32_t*)ctx_buf(c, "bdc_desc2", (size_t)2 * S * 4);
    if (st || !L || !Qhat || !ddesc) return st ? st : SDPSR_OUT_OF_MEMORY;
    std::vector<int32_t> hdesc(2 * (size_t)S);
    {
        int64_t o = 0, colbase = 0;
        for (int32_t sz : c->bdc_sizes) {
            for (int b2 = 0; b2 < sz; ++b2)
                for (int a2 = 0; a2 < sz; ++a2) {
                    hdesc[o] = (int32_t)(colbase + a2);
                    hdesc[S + o] = (int32_t)(colbase + b2);
                    ++o;
                }
            colbase += sz;
        }
    }
    st = h2d_sync(c, ddesc, hdesc.data(), hdesc.size() * 4);
    if (st) return st;
    if (n <= 64) {
        launch_cx_basis_image(s, n, d, S, L, Qhat, ddesc, ddesc + S, 1e-12 * (double)n, out);
    } else {
        // entries grouped by class (_constraints(P), src/diagonalize.jl:42-50): a class workgroup walks its own entries only
        uint32_t* ent = nullptr;
        int64_t* class_ptr = nullptr;
        st = sort_entries_by_label(c, n * n, d, L, &ent, &class_ptr);
        if (st) return st;
        int64_t* d_cls = (int64_t*)ctx_buf(c, "bi_cls_ptr", (size_t)(d + 2) * 8);
        if (!d_cls) {
            free(class_ptr);
            return SDPSR_OUT_OF_MEMORY;
        }
        st = h2d_sync(c, d_cls, class_ptr, (size_t)(d + 2) * 8);
        free(class_ptr);
        if (st) return st;
        launch_cx_basis_image_sorted(s, n, d, S, ent, d_cls, Qhat, ddesc, ddesc + S, 1e-12 * (double)n, out);
    }
    HIP_TRY(c, hipGetLastError());
    st = out_finish(c, blks, out, (size_t)2 * d * S, mem);
    if (st) return st;
    if (Q_hat) {
        HIP_TRY(c, hipMemcpyAsync(Q_hat, Qhat, (size_t)2 * n * S1 * 8, mem == SDPSR_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, s));
        HIP_TRY(c, ctx_sync_stream(c, s));
    }
    return SDPSR_OK;
}

}  // extern "C"
