// Dense symmetric eigensolver and small products for the HOST side of the module-compression driver
// (small_eigen_host.cpp): plain C++, no HIP -- compiled by the host compiler with clones of the hot
// loops for AVX-512 / AVX2 hosts (resolved at load time; the baseline clone runs anywhere).
//
// eigen(A) of src/eigen_decomposition.jl:246 for the w x w compressed elements (w <= 64):
// Householder tridiagonalisation with the reflectors accumulated, implicit-shift QL on the
// tridiagonal matrix, eigenvalues ascending -- the textbook dense method (what LAPACK's dsteqr path
// does), ~10 w^3 flop: 0.4 Mflop at w = 34.
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstring>
#include <numeric>
#include <vector>

#if defined(__x86_64__) && defined(__GNUC__) && !defined(__clang__)
#define SDPSR_HOST_CLONES __attribute__((target_clones("avx512f", "avx2", "default")))
#else
#define SDPSR_HOST_CLONES
#endif

namespace sdpsr {

namespace {

// y += a * x
SDPSR_HOST_CLONES void axpy(int n, double a, const double* __restrict__ x, double* __restrict__ y) {
    for (int i = 0; i < n; ++i) y[i] += a * x[i];
}
SDPSR_HOST_CLONES double dot(int n, const double* __restrict__ x, const double* __restrict__ y) {
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    int i = 0;
    for (; i + 3 < n; i += 4) {
        s0 += x[i] * y[i];
        s1 += x[i + 1] * y[i + 1];
        s2 += x[i + 2] * y[i + 2];
        s3 += x[i + 3] * y[i + 3];
    }
    for (; i < n; ++i) s0 += x[i] * y[i];
    return (s0 + s1) + (s2 + s3);
}
// b -= v * qj + q * vj   (one column of the symmetric rank-2 update)
SDPSR_HOST_CLONES void rank2_col(int n, double* __restrict__ b, const double* __restrict__ v, const double* __restrict__ q, double vj,
                                 double qj) {
    for (int i = 0; i < n; ++i) b[i] -= v[i] * qj + q[i] * vj;
}
// plane rotation of two columns: (z0, z1) <- (c z0 - s z1, s z0 + c z1)
SDPSR_HOST_CLONES void rot_cols(int n, double* __restrict__ z0, double* __restrict__ z1, double c, double s) {
    for (int k = 0; k < n; ++k) {
        const double f = z1[k], g = z0[k];
        z1[k] = s * g + c * f;
        z0[k] = c * g - s * f;
    }
}
inline double pythag(double a, double b) {  // sqrt(a^2 + b^2) without spurious over/underflow
    a = std::fabs(a);
    b = std::fabs(b);
    const double hi = a > b ? a : b, lo = a > b ? b : a;
    if (hi > 1e-140 && hi < 1e140) return std::sqrt(hi * hi + lo * lo);  // the common case: one square root
    if (hi == 0.0) return 0.0;
    const double t = lo / hi;
    return hi * std::sqrt(1.0 + t * t);
}

}  // namespace

// A: n x n column-major (lda), both triangles valid.  Z (n x n, ldz): orthonormal eigenvectors in
// columns, w ascending.  Returns 0, or k > 0 if the QL iteration for eigenvalue k did not converge.
int host_syev(int n, const double* A, int lda, double* w, double* Z, int ldz) {
    if (n <= 0) return 0;
    std::vector<double> a((size_t)n * n), e(n, 0.0), tau(n, 0.0), p(n), v(n);
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) a[(size_t)i + (size_t)j * n] = 0.5 * (A[(size_t)i + (size_t)j * lda] + A[(size_t)j + (size_t)i * lda]);
    auto at = [&](int i, int j) -> double& { return a[(size_t)i + (size_t)j * n]; };
    // --- tridiagonalisation: H_k = I - tau_k v_k v_k', v_k = (1, x) kept below the subdiagonal of column k
    for (int k = 0; k + 2 < n; ++k) {
        const int m = n - k - 1;  // order of the trailing block
        double* x = &at(k + 1, k);
        const double xn2 = dot(m - 1, x + 1, x + 1);
        if (xn2 == 0.0) {
            e[k] = x[0];
            tau[k] = 0.0;
            continue;
        }
        const double alpha = x[0];
        const double beta = -std::copysign(std::sqrt(alpha * alpha + xn2), alpha);
        tau[k] = (beta - alpha) / beta;
        const double sc = 1.0 / (alpha - beta);
        v[0] = 1.0;
        for (int i = 1; i < m; ++i) v[i] = x[i] * sc;
        e[k] = beta;
        // p = tau * B v  (B = trailing block, symmetric, both triangles kept current)
        for (int j = 0; j < m; ++j) p[j] = tau[k] * dot(m, &at(k + 1, k + 1 + j), v.data());
        const double half = 0.5 * tau[k] * dot(m, p.data(), v.data());
        axpy(m, -half, v.data(), p.data());  // q = p - (tau/2)(p'v) v
        for (int j = 0; j < m; ++j) rank2_col(m, &at(k + 1, k + 1 + j), v.data(), p.data(), v[j], p[j]);  // B -= v q' + q v'
        for (int i = 1; i < m; ++i) x[i] = v[i];
    }
    std::vector<double> d(n);
    for (int i = 0; i < n; ++i) d[i] = at(i, i);
    if (n >= 2) e[n - 2] = at(n - 1, n - 2);
    // --- Q = H_0 H_1 ... H_{n-3}, accumulated backwards into Z
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) Z[(size_t)i + (size_t)j * ldz] = (i == j) ? 1.0 : 0.0;
    for (int k = n - 3; k >= 0; --k) {
        if (tau[k] == 0.0) continue;
        const int m = n - k - 1;
        v[0] = 1.0;
        for (int i = 1; i < m; ++i) v[i] = at(k + 1 + i, k);
        for (int j = k + 1; j < n; ++j) {  // Z[k+1:, j] -= tau v (v' Z[k+1:, j])
            double* zj = Z + (size_t)(k + 1) + (size_t)j * ldz;
            axpy(m, -tau[k] * dot(m, v.data(), zj), v.data(), zj);
        }
    }
    // --- implicit QL with Wilkinson shifts on (d, e); rotations applied to the columns of Z
    // The scalar recurrence that generates a sweep's rotations is a dependent chain (square root,
    // division, four multiply-adds: ~60 clocks per rotation) with almost no instructions in it; the
    // rotation of two columns of Z is ~n/2 independent vector operations.  Rotations are therefore
    // applied ONE STEP LATE (generation order is kept): the out-of-order core runs the vector work of
    // rotation k in the shadow of the chain of rotation k + 1.
    const double eps = 2.220446049250313e-16;
    int held_i = -1;  // the one rotation generated but not yet applied
    double held_c = 1, held_s = 0;
    auto apply_held = [&]() {
        if (held_i >= 0) rot_cols(n, Z + (size_t)held_i * ldz, Z + (size_t)(held_i + 1) * ldz, held_c, held_s);
        held_i = -1;
    };
    for (int l = 0; l < n; ++l) {
        int iter = 0;
        for (;;) {
            int m = l;
            for (; m + 1 < n; ++m)
                if (std::fabs(e[m]) <= eps * (std::fabs(d[m]) + std::fabs(d[m + 1]))) break;
            if (m == l) break;
            if (++iter > 60) return l + 1;
            double g = (d[l + 1] - d[l]) / (2.0 * e[l]);
            double r = pythag(g, 1.0);
            g = d[m] - d[l] + e[l] / (g + std::copysign(r, g));
            double s = 1.0, c = 1.0, pp = 0.0;
            int i = m - 1;
            for (; i >= l; --i) {
                const double f = s * e[i], b = c * e[i];
                r = pythag(f, g);
                e[i + 1] = r;
                if (r == 0.0) {
                    d[i + 1] -= pp;
                    e[m] = 0.0;
                    break;
                }
                const double rinv = 1.0 / r;
                s = f * rinv;
                c = g * rinv;
                g = d[i + 1] - pp;
                r = (d[i] - g) * s + 2.0 * c * b;
                pp = s * r;
                d[i + 1] = g + pp;
                g = c * r - b;
                apply_held();  // the PREVIOUS rotation: independent of this step's chain
                held_i = i;
                held_c = c;
                held_s = s;
            }
            if (r == 0.0 && i >= l) continue;
            d[l] -= pp;
            e[l] = g;
            e[m] = 0.0;
        }
    }
    apply_held();
    // --- ascending order
    std::vector<int> ord(n);
    std::iota(ord.begin(), ord.end(), 0);
    std::stable_sort(ord.begin(), ord.end(), [&](int x, int y) { return d[x] < d[y]; });
    std::vector<double> zs((size_t)n * n);
    for (int j = 0; j < n; ++j) {
        w[j] = d[ord[j]];
        memcpy(&zs[(size_t)j * n], Z + (size_t)ord[j] * ldz, (size_t)n * sizeof(double));
    }
    for (int j = 0; j < n; ++j) memcpy(Z + (size_t)j * ldz, &zs[(size_t)j * n], (size_t)n * sizeof(double));
    return 0;
}

// hist[c] = number of x with exactly c of the 17 ascending edges <= x, c = 0..17 (a NaN counts as 17: it compares
// false with every "edge > x").  The log-histogram of otsu_threshold (src/eigen_decomposition.jl:83-110) over the
// neig^2 block norms -- a million values at neig = 1024 -- eight values per step, no branches.
SDPSR_HOST_CLONES void host_count_edges17(const double* __restrict__ x, size_t n, const double* __restrict__ ed, int64_t* __restrict__ hist) {
    int64_t h[8][20] = {};
    size_t e = 0;
    for (; e + 8 <= n; e += 8) {
        int64_t le[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 17; ++i) {
            const double t = ed[i];
            for (int q = 0; q < 8; ++q) le[q] += (t > x[e + q]) ? 0 : 1;
        }
        for (int q = 0; q < 8; ++q) ++h[q][le[q]];
    }
    for (; e < n; ++e) {
        int le = 0;
        for (int i = 0; i < 17; ++i) le += (ed[i] > x[e]) ? 0 : 1;
        ++h[0][le];
    }
    for (int f = 0; f <= 17; ++f) {
        hist[f] = 0;
        for (int q = 0; q < 8; ++q) hist[f] += h[q][f];
    }
}

// C (m x n) = A' B with A: k x m (lda), B: k x n (ldb), column-major, small orders
void host_gemm_tn(int m, int n, int k, const double* A, int lda, const double* B, int ldb, double* C, int ldc) {
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < m; ++i) C[(size_t)i + (size_t)j * ldc] = dot(k, A + (size_t)i * lda, B + (size_t)j * ldb);
}

}  // namespace sdpsr
