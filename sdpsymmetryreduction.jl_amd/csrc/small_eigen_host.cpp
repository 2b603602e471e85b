// Murota's decomposition of the COMPRESSED problem on the host (module-compression driver, order
// w <= 64): eigen(A) (src/eigen_decomposition.jl:246), the EigenDecomposition clustering (:19-40),
// Q'AQ and the block norms (:177-205), Otsu threshold + union-find + __isconsistent (:83-139,163-167,
// 205-217) and irreducible_decomposition (:295-348) on w x w matrices.
//
// Why on the host: the w x w compressed elements B = W'A W are a few KiB and the driver reads back
// the product that contains the first of them anyway (module growth: the Gram matrix of the final
// invariance round).  A 34 x 34 symmetric eigenproblem is ~0.3 Mflop -- tens of microseconds on one
// host core -- while the one-workgroup Jacobi kernel it replaces used 1 of 256 CUs for 235 us (409 us
// at w = 50) and sat on the critical path of every reduction.  The host solves it while the device
// forms the second generic element.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>
#include <numeric>
#include <vector>

#include "host_internal.h"

namespace sdpsr {

// eigen_decomposition + irreducible_decomposition of the compressed problem, entirely on the host.
//   B1            first compressed generic element (w x w, column-major, ld w), symmetric
//   next_element  delivers a further independent compressed generic element (w x w) into dst: called
//                 once for the coupling element (:259-262) AFTER B1 has been diagonalised -- the device
//                 forms it meanwhile --, again for every extra coupling element (raised on demand when
//                 the classes do not add up to expect_dim, see eigen_decomposition_device) and for the
//                 irreducible step when a fresh element is required (:306)
//   Qs            out: Q_hat of the compressed problem, w x S1 column-major
int murota_small_host(sdpsr_ctx* c, int w, const double* B1, const std::function<int(double*)>& next_element, double atol,
                      int64_t expect_dim, std::vector<int32_t>& sizes, int64_t& S1, int64_t& S, std::vector<double>& Qs) {
    std::vector<double> vals(w), Q((size_t)w * w);
    const int info = host_syev(w, B1, w, vals.data(), Q.data(), w);
    if (info != 0) return ctx_fail(c, SDPSR_SOLVER_ERROR, "host eigensolver did not converge, info=" + std::to_string(info));
    // EigenDecomposition ctor (:19-40): a new eigenspace where |dv| > atol
    std::vector<int> ptrs(1, 0);
    for (int i = 0; i + 1 < w; ++i)
        if (!(std::fabs(vals[i + 1] - vals[i]) <= atol)) ptrs.push_back(i + 1);
    ptrs.push_back(w);
    const int neig = (int)ptrs.size() - 1;
    std::vector<int> space_of(w);
    for (int b = 0; b < neig; ++b)
        for (int i = ptrs[b]; i < ptrs[b + 1]; ++i) space_of[i] = b;
    auto dimof = [&](int b) { return ptrs[b + 1] - ptrs[b]; };
    // second generic element: T = A2 Q, M = Q' T, block maxima (:201-205)
    std::vector<double> A2((size_t)w * w), T((size_t)w * w), M((size_t)w * w), norms((size_t)neig * neig, 0.0);
    std::vector<int> kpart;
    bool t_valid = true;
    auto couple = [&](std::vector<double>& Tout) -> int {
        const int st = next_element(A2.data());
        if (st) return st;
        host_gemm_tn(w, w, w, A2.data(), w, Q.data(), w, Tout.data(), w);  // A2 symmetric: A2' Q = A2 Q
        host_gemm_tn(w, w, w, Q.data(), w, Tout.data(), w, M.data(), w);
        for (int j = 0; j < w; ++j)
            for (int i = 0; i < w; ++i) {
                double& nv = norms[(size_t)space_of[i] * neig + space_of[j]];
                nv = std::max(nv, std::fabs(M[(size_t)i + (size_t)j * w]));
            }
        return SDPSR_OK;
    };
    int st = couple(T);
    if (st) return st;
    std::vector<double> sym((size_t)neig * neig);
    for (int extra = 0;; ++extra) {
        // blocks between eigenspaces of different dimension count as zero (:185-186)
        for (int i = 0; i < neig; ++i)
            for (int j = i; j < neig; ++j) {
                const double v = (dimof(i) != dimof(j)) ? 0.0 : norms[(size_t)i * neig + j];
                sym[(size_t)i * neig + j] = sym[(size_t)j * neig + i] = v;
            }
        st = isomorphism_classes(c, sym, neig, atol, kpart);
        if (st != SDPSR_OK && st != SDPSR_NUMERICAL_INCONSISTENCY) return st;
        bool done = extra >= 2 || expect_dim < 0 || (c->opts.flags & SDPSR_FLAG_SINGLE_COUPLING_ELEMENT);
        if (!done && st == SDPSR_OK) {
            std::vector<int> cnt(neig, 0);
            for (int i = 0; i < neig; ++i) ++cnt[kpart[i]];
            int64_t fd = 0;
            for (int i = 0; i < neig; ++i) fd += (int64_t)cnt[i] * (cnt[i] + 1) / 2;
            done = fd == expect_dim;
        }
        if (done) {
            if (st) return st;
            c->err.clear();
            break;
        }
        // the coupling of an isomorphic pair of 1- or 2-dimensional eigenspaces under ONE element is a
        // single random number: raise the coupling matrix by another independent element (maxima only grow)
        t_valid = false;
        std::vector<double> Tx((size_t)w * w);
        st = couple(Tx);
        if (st) return st;
    }
    // irreducible_decomposition (:295-348)
    std::vector<int> roots;
    std::vector<std::vector<int>> members;
    class_structure(kpart, roots, members);
    sizes.assign(roots.size(), 0);
    S1 = 0;
    S = 0;
    bool merged = false;
    for (size_t p = 0; p < roots.size(); ++p) {
        sizes[p] = (int32_t)members[p].size();
        S1 += sizes[p];
        S += (int64_t)sizes[p] * sizes[p];
        merged = merged || members[p].size() > 1;
    }
    if (merged && (!t_valid || (c->opts.flags & SDPSR_FLAG_FRESH_IRREDUCIBLE_ELEMENT))) {
        st = next_element(A2.data());  // generic element #3 (:306)
        if (st) return st;
        host_gemm_tn(w, w, w, A2.data(), w, Q.data(), w, T.data(), w);
    }
    Qs.assign((size_t)w * S1, 0.0);
    int64_t col = 0;
    std::vector<double> wv(w), cv(w);
    for (size_t p = 0; p < roots.size(); ++p) {
        const int i = roots[p];
        const int ci = ptrs[i], mi = dimof(i);
        memcpy(&Qs[(size_t)col * w], &Q[(size_t)ci * w], (size_t)w * sizeof(double));  // P1 = I (:311-313, :326)
        ++col;
        for (size_t q = 1; q < members[p].size(); ++q) {
            const int j = members[p][q];
            const int cj = ptrs[j], mj = dimof(j);
            // first column of P_blk = block(A, Ei, Ej)' is Qj'(A q_i1) (:333), normalised by
            // |Qi'(A q_j1)| (:335); the column of P_hat is Qj times it (:338-344)
            const double* bi = &T[(size_t)ci * w];
            const double* bj = &T[(size_t)cj * w];
            for (int t = 0; t < mj; ++t) {
                double s = 0;
                for (int r = 0; r < w; ++r) s += Q[(size_t)r + (size_t)(cj + t) * w] * bi[r];
                wv[t] = s;
            }
            double nn = 0;
            for (int t = 0; t < mi; ++t) {
                double s = 0;
                for (int r = 0; r < w; ++r) s += Q[(size_t)r + (size_t)(ci + t) * w] * bj[r];
                cv[t] = s;
                nn += s * s;
            }
            const double inv = 1.0 / std::sqrt(nn);
            double* dst = &Qs[(size_t)col * w];
            for (int t = 0; t < mj; ++t) {
                const double f = wv[t] * inv;
                const double* qc = &Q[(size_t)(cj + t) * w];
                for (int r = 0; r < w; ++r) dst[r] += qc[r] * f;
            }
            ++col;
        }
    }
    return SDPSR_OK;
}

}  // namespace sdpsr
