// Complex path of blockDiagonalize (src/compat.jl:26-32,54-57; src/diagonalize.jl:13-28):
// desymmetrize, then Murota's decomposition over C.  Orders n <= 64 run every step in
// single-workgroup kernels on LDS-/L2-resident data (the complex path is the reference's answer to
// small algebras that do not split over the reals: test/runtests.jl:43-57 runs it on 3 x 3 and 4 x 4
// partitions); larger orders go through the real symmetric embedding (second half of this file).
//
// DEVIATION: the reference draws a generic element with complex coefficients and
// calls the general (non-Hermitian) eigen(); here the generic elements are HERMITIAN,
// H = A + A^H with A = sum_i c_i 1[P==i], c_i complex -- an element of the same *-closed algebra
// whose eigenspaces split C^n the same way (Murota et al. work with self-adjoint generic
// elements) -- so the eigensolver is a Hermitian Jacobi iteration with orthonormal vectors and
// real eigenvalues, and Q'AQ is a unitary congruence.  Block sizes and the spectra of the block
// images are those of the reference (pinned by its tests and by the spectrum invariant).
//
// Complex matrices are kept as separate real / imaginary planes (column-major, ld = n).
#include "sdpsr_internal.h"
#include "jacobi64.h"

namespace sdpsr {

struct cxd {
    double re, im;
};
__device__ __forceinline__ cxd cx_mul(cxd a, cxd b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ cxd cx_mulc(cxd a, cxd b) { return {a.re * b.re + a.im * b.im, a.im * b.re - a.re * b.im}; }  // a * conj(b)
__device__ __forceinline__ cxd cx_cmul(cxd a, cxd b) { return {a.re * b.re + a.im * b.im, a.re * b.im - a.im * b.re}; }  // conj(a) * b

// complex class value: real and imaginary part uniform in [0,1) (rand(ComplexF64), src/abstract_part.jl:108)
__device__ __forceinline__ cxd cx_class_value(uint64_t key, uint32_t l) {
    if (l == 0u) return {0.0, 0.0};
    return {sdpsr_class_uniform(key, l), sdpsr_class_uniform(key ^ 0xA5A5A5A55A5A5A5Aull, l)};
}

// H = A + A^H, A[r,c] = value(L[r,c])
__global__ void cx_gather_herm_kernel(int n, const uint32_t* __restrict__ L, uint64_t key, double* __restrict__ Hr,
                                      double* __restrict__ Hi) {
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n * n; e += gridDim.x * blockDim.x) {
        const int c = e / n, r = e - c * n;
        const cxd a = cx_class_value(key, L[r + c * n]), b = cx_class_value(key, L[c + r * n]);
        Hr[e] = a.re + b.re;
        Hi[e] = a.im - b.im;
    }
}

// ---------------------------------------------------------------------------
// Hermitian eigensolver, n <= 64, one workgroup: two-sided Jacobi with the round-robin ordering of
// jacobi64.h.  Pair (p,q): a_pq = b e^{i phi}; U = diag(1, e^{-i phi}) * R(theta) with the REAL
// rotation of (a_pp, a_qq, b), so that U^H A U is diagonal on the pair.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(1024)
cx_heev_jacobi64_kernel(int n, const double* __restrict__ Hr, const double* __restrict__ Hi, double* __restrict__ wout,
                        double* __restrict__ Vr, double* __restrict__ Vi, int* __restrict__ info) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    __shared__ double s_red[1024 / 64 + 2];
    __shared__ int s_rank[64];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int m = (n + 1) & ~1, half = m >> 1, ldl = m | 1;
    double* sAr = sm;
    double* sAi = sAr + (size_t)ldl * m;
    double* sVr = sAi + (size_t)ldl * m;
    double* sVi = sVr + (size_t)ldl * m;
    int* s_pq = reinterpret_cast<int*>(sVi + (size_t)ldl * m);
    for (int e = tid; e < m * m; e += nthr) {
        const int j = e / m, i = e - j * m;
        const bool in = i < n && j < n;
        sAr[i + j * ldl] = in ? Hr[i + j * n] : 0.0;
        sAi[i + j * ldl] = in ? Hi[i + j * n] : 0.0;
        sVr[i + j * ldl] = (i == j) ? 1.0 : 0.0;
        sVi[i + j * ldl] = 0.0;
    }
    jacobi64_fill_pairs(m, s_pq);
    __syncthreads();
    const int nw = nthr >> 6;
    const bool active = tid < half * half;
    const int k1 = active ? tid / half : 0, k2 = active ? tid - k1 * half : 0;
    const int i0 = 2 * k2;
    int sweep = 0;
    for (; sweep < 60; ++sweep) {
        double off = 0, dg = 0;
        for (int e = tid; e < m * m; e += nthr) {
            const int j = e / m, i = e - j * m;
            const double vr = sAr[i + j * ldl], vi = sAi[i + j * ldl];
            if (i == j) dg = fma(vr, vr, dg);
            else off = fma(vr, vr, fma(vi, vi, off));
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            off += __shfl_down(off, o, 64);
            dg += __shfl_down(dg, o, 64);
        }
        __syncthreads();
        if ((tid & 63) == 0) s_red[tid >> 6] = off;
        __syncthreads();
        if (tid == 0) {
            double t = 0;
            for (int k = 0; k < nw; ++k) t += s_red[k];
            s_red[nw] = t;
        }
        __syncthreads();
        if ((tid & 63) == 0) s_red[tid >> 6] = dg;
        __syncthreads();
        if (tid == 0) {
            double t = 0;
            for (int k = 0; k < nw; ++k) t += s_red[k];
            s_red[nw + 1] = t;
        }
        __syncthreads();
        const double s_off = s_red[nw], s_diag = s_red[nw + 1];
        const double tolr = (double)n * 2.220446049250313e-16;
        if (s_off <= tolr * tolr * (s_diag + s_off) || s_off == 0.0) break;
        const int* tab = s_pq;
        for (int step = 0; step < m - 1; ++step, tab += half) {
            double cr = 1, sr = 0, cc = 1, sc = 0;
            cxd er = {1, 0}, ec = {1, 0};
            cxd x00 = {0, 0}, x01 = {0, 0}, x10 = {0, 0}, x11 = {0, 0}, va0 = {0, 0}, va1 = {0, 0}, vb0 = {0, 0}, vb1 = {0, 0};
            int a00 = 0, a01 = 0, a10 = 0, a11 = 0, v0a = 0, v0b = 0;
            if (active) {
                const int pq1 = tab[k1], pq2 = tab[k2];
                const int r0 = pq1 & 0xFFFF, r1 = pq1 >> 16, c0 = pq2 & 0xFFFF, c1 = pq2 >> 16;
                a00 = r0 + c0 * ldl;
                a01 = r0 + c1 * ldl;
                a10 = r1 + c0 * ldl;
                a11 = r1 + c1 * ldl;
                v0a = i0 + r0 * ldl;
                v0b = i0 + r1 * ldl;
                auto angle = [&](int p, int q, double& c, double& s, cxd& e) {
                    const double app = sAr[p + p * ldl], aqq = sAr[q + q * ldl];
                    const double pr = sAr[p + q * ldl], pi = sAi[p + q * ldl];
                    const double b2 = fma(pr, pr, pi * pi);
                    if (b2 > 0.0) {
                        const double b = sqrt(b2);
                        e = {pr / b, pi / b};
                        jacobi_angle(app, aqq, b, c, s);
                    } else {
                        e = {1.0, 0.0};
                        c = 1.0;
                        s = 0.0;
                    }
                };
                angle(r0, r1, cr, sr, er);
                angle(c0, c1, cc, sc, ec);
                x00 = {sAr[a00], sAi[a00]};
                x01 = {sAr[a01], sAi[a01]};
                x10 = {sAr[a10], sAi[a10]};
                x11 = {sAr[a11], sAi[a11]};
                va0 = {sVr[v0a], sVi[v0a]};
                vb0 = {sVr[v0b], sVi[v0b]};
                va1 = {sVr[v0a + 1], sVi[v0a + 1]};
                vb1 = {sVr[v0b + 1], sVi[v0b + 1]};
            }
            __syncthreads();  // every read of this step is done before any write
            if (active) {
                // rows: U^H = [[c, -s e], [s, c e]]
                const cxd e1_10 = cx_mul(er, x10), e1_11 = cx_mul(er, x11);
                const cxd y00 = {cr * x00.re - sr * e1_10.re, cr * x00.im - sr * e1_10.im};
                const cxd y01 = {cr * x01.re - sr * e1_11.re, cr * x01.im - sr * e1_11.im};
                const cxd y10 = {sr * x00.re + cr * e1_10.re, sr * x00.im + cr * e1_10.im};
                const cxd y11 = {sr * x01.re + cr * e1_11.re, sr * x01.im + cr * e1_11.im};
                // columns: U = [[c, s], [-s conj(e), c conj(e)]]
                const cxd f01 = cx_mulc(y01, ec), f11 = cx_mulc(y11, ec);
                sAr[a00] = cc * y00.re - sc * f01.re;
                sAi[a00] = cc * y00.im - sc * f01.im;
                sAr[a01] = sc * y00.re + cc * f01.re;
                sAi[a01] = sc * y00.im + cc * f01.im;
                sAr[a10] = cc * y10.re - sc * f11.re;
                sAi[a10] = cc * y10.im - sc * f11.im;
                sAr[a11] = sc * y10.re + cc * f11.re;
                sAi[a11] = sc * y10.im + cc * f11.im;
                // V <- V U on the columns of the row pair, rows i0, i0+1
                const cxd g0 = cx_mulc(vb0, er), g1 = cx_mulc(vb1, er);
                sVr[v0a] = cr * va0.re - sr * g0.re;
                sVi[v0a] = cr * va0.im - sr * g0.im;
                sVr[v0b] = sr * va0.re + cr * g0.re;
                sVi[v0b] = sr * va0.im + cr * g0.im;
                sVr[v0a + 1] = cr * va1.re - sr * g1.re;
                sVi[v0a + 1] = cr * va1.im - sr * g1.im;
                sVr[v0b + 1] = sr * va1.re + cr * g1.re;
                sVi[v0b + 1] = sr * va1.im + cr * g1.im;
            }
            __syncthreads();
        }
    }
    for (int i = tid; i < n; i += nthr) {
        const double li = sAr[i + i * ldl];
        int rk = 0;
        for (int j = 0; j < n; ++j) {
            const double lj = sAr[j + j * ldl];
            rk += (lj < li) || (lj == li && j < i);
        }
        s_rank[i] = rk;
        wout[rk] = li;
    }
    __syncthreads();
    for (int e = tid; e < n * n; e += nthr) {
        const int j = e / n, i = e - j * n;
        Vr[i + s_rank[j] * n] = sVr[i + j * ldl];
        Vi[i + s_rank[j] * n] = sVi[i + j * ldl];
    }
    if (tid == 0) {
        info[0] = (sweep >= 60) ? 1 : 0;
        info[1] = sweep;
    }
}

// norms[sa * neig + sb] = max |(V^H H V)[a, b]| over a in E_sa, b in E_sb (one workgroup; T = H V in LDS)
__global__ void __launch_bounds__(1024)
cx_block_norms_kernel(int n, const double* __restrict__ Hr, const double* __restrict__ Hi, const double* __restrict__ Vr,
                      const double* __restrict__ Vi, const int32_t* __restrict__ space_of, int neig,
                      unsigned long long* __restrict__ norms) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double* Tr = sm;
    double* Ti = Tr + n * n;
    const int tid = threadIdx.x, nthr = blockDim.x;
    for (int e = tid; e < n * n; e += nthr) {
        const int j = e / n, i = e - j * n;  // T[i,j] = sum_k H[i,k] V[k,j]
        cxd acc = {0, 0};
        for (int k = 0; k < n; ++k) {
            const cxd h = {Hr[i + k * n], Hi[i + k * n]}, v = {Vr[k + j * n], Vi[k + j * n]};
            const cxd p = cx_mul(h, v);
            acc.re += p.re;
            acc.im += p.im;
        }
        Tr[e] = acc.re;
        Ti[e] = acc.im;
    }
    __syncthreads();
    for (int e = tid; e < n * n; e += nthr) {
        const int b = e / n, a = e - b * n;  // M[a,b] = sum_k conj(V[k,a]) T[k,b]
        cxd acc = {0, 0};
        for (int k = 0; k < n; ++k) {
            const cxd v = {Vr[k + a * n], Vi[k + a * n]}, t = {Tr[k + b * n], Ti[k + b * n]};
            const cxd p = cx_cmul(v, t);
            acc.re += p.re;
            acc.im += p.im;
        }
        const double mag = sqrt(acc.re * acc.re + acc.im * acc.im);
        atomicMax(&norms[space_of[a] * neig + space_of[b]], (unsigned long long)__double_as_longlong(mag));
    }
}

// irreducible_decomposition (src/eigen_decomposition.jl:295-348) over C with a Hermitian generic
// element H3.  desc per output column: {kind, i0, mi, j0, mj, col}: kind 0 = copy eigenvector i0
// (first member of a class), kind 1 = member j of the class rooted at i:
//     column = Q_j (Q_j^H H3 q_i1) / || q_j1^H H3 Q_i ||
// One workgroup per column.  Qhat: n x S1, interleaved (re, im).
__global__ void __launch_bounds__(256)
cx_irreducible_kernel(int n, const double* __restrict__ Hr, const double* __restrict__ Hi, const double* __restrict__ Vr,
                      const double* __restrict__ Vi, const int32_t* __restrict__ desc, double atol, double* __restrict__ Qhat) {
    __shared__ double tr[64], ti[64], ur[64], ui[64], cr[64], ci[64];
    __shared__ double s_nrm;
    const int32_t* dsc = desc + 6 * blockIdx.x;
    const int kind = dsc[0], i0 = dsc[1], mi = dsc[2], j0 = dsc[3], mj = dsc[4], col = dsc[5];
    const int tid = threadIdx.x;
    double* out = Qhat + (size_t)2 * n * col;
    if (kind == 0) {
        for (int r = tid; r < n; r += blockDim.x) {
            cxd v = {Vr[r + i0 * n], Vi[r + i0 * n]};
            if (sqrt(v.re * v.re + v.im * v.im) < atol) v = {0, 0};
            out[2 * r] = v.re;
            out[2 * r + 1] = v.im;
        }
        return;
    }
    if (tid < n) {  // t = H3 q_i1, u = H3 q_j1
        cxd t = {0, 0}, u = {0, 0};
        for (int k = 0; k < n; ++k) {
            const cxd h = {Hr[tid + k * n], Hi[tid + k * n]};
            const cxd p = cx_mul(h, {Vr[k + i0 * n], Vi[k + i0 * n]}), q = cx_mul(h, {Vr[k + j0 * n], Vi[k + j0 * n]});
            t.re += p.re;
            t.im += p.im;
            u.re += q.re;
            u.im += q.im;
        }
        tr[tid] = t.re;
        ti[tid] = t.im;
        ur[tid] = u.re;
        ui[tid] = u.im;
    }
    __syncthreads();
    if (tid < mj) {  // c = Q_j^H t
        cxd acc = {0, 0};
        for (int k = 0; k < n; ++k) {
            const cxd p = cx_cmul({Vr[k + (j0 + tid) * n], Vi[k + (j0 + tid) * n]}, {tr[k], ti[k]});
            acc.re += p.re;
            acc.im += p.im;
        }
        cr[tid] = acc.re;
        ci[tid] = acc.im;
    }
    if (tid == 64) {  // || Q_i^H u ||
        double s2 = 0;
        for (int a = 0; a < mi; ++a) {
            cxd acc = {0, 0};
            for (int k = 0; k < n; ++k) {
                const cxd p = cx_cmul({Vr[k + (i0 + a) * n], Vi[k + (i0 + a) * n]}, {ur[k], ui[k]});
                acc.re += p.re;
                acc.im += p.im;
            }
            s2 += acc.re * acc.re + acc.im * acc.im;
        }
        s_nrm = sqrt(s2);
    }
    __syncthreads();
    const double inv = s_nrm > 0 ? 1.0 / s_nrm : 0.0;
    for (int r = tid; r < n; r += blockDim.x) {
        cxd acc = {0, 0};
        for (int a = 0; a < mj; ++a) {
            const cxd p = cx_mul({Vr[r + (j0 + a) * n], Vi[r + (j0 + a) * n]}, {cr[a], ci[a]});
            acc.re += p.re;
            acc.im += p.im;
        }
        acc.re *= inv;
        acc.im *= inv;
        if (sqrt(acc.re * acc.re + acc.im * acc.im) < atol) acc = {0, 0};
        out[2 * r] = acc.re;
        out[2 * r + 1] = acc.im;
    }
}

// basis_image over C (src/diagonalize.jl:64-89): blks[i][k] = Q_k^H 1[P==i] Q_k.  One workgroup per
// class i; out: d x S complex (interleaved), class-major, blocks side by side, column-major inside.
// descA/descB: the two columns of Qhat of every output.
__global__ void __launch_bounds__(256)
cx_basis_image_kernel(int n, int S, const uint32_t* __restrict__ L, const double* __restrict__ Qhat,
                      const int32_t* __restrict__ descA, const int32_t* __restrict__ descB, double atol,
                      double* __restrict__ out) {
    const uint32_t cls = blockIdx.x + 1;
    const int tid = threadIdx.x;
    for (int o0 = 0; o0 < S; o0 += 256) {
        const int o = o0 + tid;
        const bool ok = o < S;
        const int ca = ok ? descA[o] : 0, cb = ok ? descB[o] : 0;
        cxd acc = {0, 0};
        for (int e = 0; e < n * n; ++e) {
            if (L[e] != cls) continue;  // uniform over the workgroup
            const int c = e / n, r = e - c * n;
            const cxd qa = {Qhat[2 * (r + (size_t)ca * n)], Qhat[2 * (r + (size_t)ca * n) + 1]};
            const cxd qb = {Qhat[2 * (c + (size_t)cb * n)], Qhat[2 * (c + (size_t)cb * n) + 1]};
            const cxd p = cx_cmul(qa, qb);
            acc.re += p.re;
            acc.im += p.im;
        }
        if (ok) {
            if (sqrt(acc.re * acc.re + acc.im * acc.im) < atol) acc = {0, 0};
            out[2 * ((size_t)blockIdx.x * S + o)] = acc.re;
            out[2 * ((size_t)blockIdx.x * S + o) + 1] = acc.im;
        }
    }
}

// ---------------------------------------------------------------------------
// Orders n > 64: the same steps on matrices that live in HBM.  A Hermitian H is handled through
// its real symmetric embedding  M(H) = [[Re H, -Im H], [Im H, Re H]]  (2n x 2n; H -> M(H) is a
// *-isomorphism onto the real matrices that commute with J = [[0, -I], [I, 0]]): the real dense
// eigensolver and the fp64 MFMA GEMMs of the real path do the O(n^3) work, small kernels move
// between the two pictures.  A complex n x k matrix Z = X + iY is the real (2n) x k matrix [X; Y].
// ---------------------------------------------------------------------------
// M (ld2 x ld2, zeroed by the caller beyond 2n) = M(H)
__global__ void cx_embed_kernel(int n, const double* __restrict__ Hr, const double* __restrict__ Hi, int64_t ld2,
                                double* __restrict__ M) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < (int64_t)n * n; e += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(e / n), r = (int)(e - (int64_t)c * n);
        const double re = Hr[e], im = Hi[e];
        M[r + (int64_t)c * ld2] = re;
        M[(r + n) + (int64_t)(c + n) * ld2] = re;
        M[(r + n) + (int64_t)c * ld2] = im;
        M[r + (int64_t)(c + n) * ld2] = -im;
    }
}
// R = J' E columnwise: R[0:n, j] = E[n:2n, j], R[n:2n, j] = -E[0:n, j]   (so that E' R = X'Y - Y'X = Im(Z^H Z))
__global__ void cx_rot_kernel(int n, int64_t cols, int64_t ld2, const double* __restrict__ E, double* __restrict__ R) {
    const int64_t total = (int64_t)n * cols;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t j = e / n;
        const int r = (int)(e - j * n);
        R[r + j * ld2] = E[(r + n) + j * ld2];
        R[(r + n) + j * ld2] = -E[r + j * ld2];
    }
}
// rows / columns >= m of an ld x ldc buffer back to zero
__global__ void cx_zero_pad_kernel(int64_t m, int64_t mc, int64_t ld, int64_t ldc, double* __restrict__ A) {
    const int64_t total = ld * ldc;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t j = e / ld, i = e - j * ld;
        if (i >= m || j >= mc) A[e] = 0.0;
    }
}
// eigenvectors of H from those of M(H): output column j (planes, ld = n) is the combination
//   sum_b Z[:, off_j + b] * C_j[b],  Z = X + iY the real eigenvectors read as complex vectors,
// C_j (complex, m2_j entries at coef + 2 * cof_j) from the host's rank-revealing Cholesky of the
// cluster's Gram matrix.  desc[j] = {off, m2, cof}
__global__ void cx_combine_kernel(int n, int64_t ld2, const double* __restrict__ E, const int32_t* __restrict__ desc,
                                  const double* __restrict__ coef, double* __restrict__ Vr, double* __restrict__ Vi) {
    const int j = blockIdx.y;
    const int off = desc[3 * j], m2 = desc[3 * j + 1], cof = desc[3 * j + 2];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    cxd acc = {0, 0};
    for (int b = 0; b < m2; ++b) {
        const cxd z = {E[i + (int64_t)(off + b) * ld2], E[(i + n) + (int64_t)(off + b) * ld2]};
        const cxd cf = {coef[2 * (cof + b)], coef[2 * (cof + b) + 1]};
        const cxd p = cx_mul(z, cf);
        acc.re += p.re;
        acc.im += p.im;
    }
    Vr[i + (int64_t)j * n] = acc.re;
    Vi[i + (int64_t)j * n] = acc.im;
}
// E (ld2 x ldc, zero beyond) = [Vr; Vi]
__global__ void cx_stack_kernel(int n, int64_t cols, const double* __restrict__ Vr, const double* __restrict__ Vi, int64_t ldv,
                                int64_t ld2, double* __restrict__ E) {
    const int64_t total = (int64_t)n * cols;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t j = e / n;
        const int r = (int)(e - j * n);
        E[r + j * ld2] = Vr[r + j * ldv];
        E[(r + n) + j * ld2] = Vi[r + j * ldv];
    }
}
// norms[sa * neig + sb] = max |Gr + i Gi| over the block (Gr, Gi: n x n in ldn x ldn buffers)
__global__ void cx_block_norms_general_kernel(int n, int64_t ldn, const double* __restrict__ Gr, const double* __restrict__ Gi,
                                              const int32_t* __restrict__ space_of, int neig, unsigned long long* __restrict__ norms) {
    const int64_t total = (int64_t)n * n;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(e / n), a = (int)(e - (int64_t)b * n);
        const double re = Gr[a + (int64_t)b * ldn], im = Gi[a + (int64_t)b * ldn];
        const double mag = sqrt(re * re + im * im);
        atomicMax(&norms[(int64_t)space_of[a] * neig + space_of[b]], (unsigned long long)__double_as_longlong(mag));
    }
}

// cx_irreducible_kernel for any n: the vectors t = H3 q_i1, u = H3 q_j1 and the coefficients live
// in dynamic LDS (6 n doubles), every loop strides over the workgroup.
__global__ void __launch_bounds__(256)
cx_irreducible_general_kernel(int n, const double* __restrict__ Hr, const double* __restrict__ Hi, const double* __restrict__ Vr,
                              const double* __restrict__ Vi, const int32_t* __restrict__ desc, double atol,
                              double* __restrict__ Qhat) {
    extern __shared__ __attribute__((aligned(16))) double sg[];
    __shared__ double s_part[256];
    double *tr = sg, *ti = tr + n, *ur = ti + n, *ui = ur + n, *cr = ui + n, *ci = cr + n;
    const int32_t* dsc = desc + 6 * blockIdx.x;
    const int kind = dsc[0], i0 = dsc[1], mi = dsc[2], j0 = dsc[3], mj = dsc[4], col = dsc[5];
    const int tid = threadIdx.x, nthr = blockDim.x;
    double* out = Qhat + (size_t)2 * n * col;
    if (kind == 0) {
        for (int r = tid; r < n; r += nthr) {
            cxd v = {Vr[r + (int64_t)i0 * n], Vi[r + (int64_t)i0 * n]};
            if (sqrt(v.re * v.re + v.im * v.im) < atol) v = {0, 0};
            out[2 * r] = v.re;
            out[2 * r + 1] = v.im;
        }
        return;
    }
    for (int r = tid; r < n; r += nthr) {  // t = H3 q_i1, u = H3 q_j1
        cxd t = {0, 0}, u = {0, 0};
        for (int k = 0; k < n; ++k) {
            const cxd h = {Hr[r + (int64_t)k * n], Hi[r + (int64_t)k * n]};
            const cxd p = cx_mul(h, {Vr[k + (int64_t)i0 * n], Vi[k + (int64_t)i0 * n]});
            const cxd q = cx_mul(h, {Vr[k + (int64_t)j0 * n], Vi[k + (int64_t)j0 * n]});
            t.re += p.re;
            t.im += p.im;
            u.re += q.re;
            u.im += q.im;
        }
        tr[r] = t.re;
        ti[r] = t.im;
        ur[r] = u.re;
        ui[r] = u.im;
    }
    __syncthreads();
    for (int a = tid; a < mj; a += nthr) {  // c = Q_j^H t
        cxd acc = {0, 0};
        for (int k = 0; k < n; ++k) {
            const cxd p = cx_cmul({Vr[k + (int64_t)(j0 + a) * n], Vi[k + (int64_t)(j0 + a) * n]}, {tr[k], ti[k]});
            acc.re += p.re;
            acc.im += p.im;
        }
        cr[a] = acc.re;
        ci[a] = acc.im;
    }
    double s2 = 0;  // || Q_i^H u ||^2, members of the root eigenspace strided over the threads
    for (int a = tid; a < mi; a += nthr) {
        cxd acc = {0, 0};
        for (int k = 0; k < n; ++k) {
            const cxd p = cx_cmul({Vr[k + (int64_t)(i0 + a) * n], Vi[k + (int64_t)(i0 + a) * n]}, {ur[k], ui[k]});
            acc.re += p.re;
            acc.im += p.im;
        }
        s2 += acc.re * acc.re + acc.im * acc.im;
    }
    s_part[tid] = s2;
    __syncthreads();
    double tot = 0;
    for (int k = 0; k < nthr; ++k) tot += s_part[k];  // fixed order
    const double nrm = sqrt(tot);
    const double inv = nrm > 0 ? 1.0 / nrm : 0.0;
    for (int r = tid; r < n; r += nthr) {
        cxd acc = {0, 0};
        for (int a = 0; a < mj; ++a) {
            const cxd p = cx_mul({Vr[r + (int64_t)(j0 + a) * n], Vi[r + (int64_t)(j0 + a) * n]}, {cr[a], ci[a]});
            acc.re += p.re;
            acc.im += p.im;
        }
        acc.re *= inv;
        acc.im *= inv;
        if (sqrt(acc.re * acc.re + acc.im * acc.im) < atol) acc = {0, 0};
        out[2 * r] = acc.re;
        out[2 * r + 1] = acc.im;
    }
}

static inline unsigned cx_grid(int64_t work) {
    int64_t g = (work + 255) / 256;
    if (g < 1) g = 1;
    if (g > 256 * 8) g = 256 * 8;
    return (unsigned)g;
}
void launch_cx_embed(hipStream_t s, int64_t n, const double* Hr, const double* Hi, int64_t ld2, double* M) {
    hipMemsetAsync(M, 0, (size_t)ld2 * ld2 * 8, s);
    cx_embed_kernel<<<cx_grid(n * n), 256, 0, s>>>((int)n, Hr, Hi, ld2, M);
}
void launch_cx_rot(hipStream_t s, int64_t n, int64_t cols, int64_t ld2, const double* E, double* R) {
    cx_rot_kernel<<<cx_grid(n * cols), 256, 0, s>>>((int)n, cols, ld2, E, R);
}
void launch_cx_zero_pad(hipStream_t s, int64_t m, int64_t mc, int64_t ld, int64_t ldc, double* A) {
    cx_zero_pad_kernel<<<cx_grid(ld * ldc), 256, 0, s>>>(m, mc, ld, ldc, A);
}
void launch_cx_combine(hipStream_t s, int64_t n, int64_t ld2, const double* E, const int32_t* desc, const double* coef, double* Vr,
                       double* Vi) {
    dim3 g((unsigned)((n + 255) / 256), (unsigned)n);
    cx_combine_kernel<<<g, 256, 0, s>>>((int)n, ld2, E, desc, coef, Vr, Vi);
}
void launch_cx_stack(hipStream_t s, int64_t n, int64_t cols, const double* Vr, const double* Vi, int64_t ldv, int64_t ld2, double* E) {
    cx_stack_kernel<<<cx_grid(n * cols), 256, 0, s>>>((int)n, cols, Vr, Vi, ldv, ld2, E);
}
void launch_cx_block_norms_general(hipStream_t s, int64_t n, int64_t ldn, const double* Gr, const double* Gi, const int32_t* space_of,
                                   int neig, unsigned long long* norms) {
    cx_block_norms_general_kernel<<<cx_grid(n * n), 256, 0, s>>>((int)n, ldn, Gr, Gi, space_of, neig, norms);
}
void launch_cx_irreducible_general(hipStream_t s, int64_t n, const double* Hr, const double* Hi, const double* Vr, const double* Vi,
                                   const int32_t* desc, int ncols, double atol, double* Qhat) {
    cx_irreducible_general_kernel<<<ncols, 256, (size_t)6 * n * sizeof(double), s>>>((int)n, Hr, Hi, Vr, Vi, desc, atol, Qhat);
}

// cx_basis_image_kernel over the entries of the class only (ent: linear indices grouped by label,
// cls_ptr[l] .. cls_ptr[l + 1] = label l): O(|class| * S) per class instead of a scan of all labels.
__global__ void __launch_bounds__(256)
cx_basis_image_sorted_kernel(int n, int S, const uint32_t* __restrict__ ent, const int64_t* __restrict__ cls_ptr,
                             const double* __restrict__ Qhat, const int32_t* __restrict__ descA,
                             const int32_t* __restrict__ descB, double atol, double* __restrict__ out) {
    const int64_t p0 = cls_ptr[blockIdx.x + 1], p1 = cls_ptr[blockIdx.x + 2];
    const int tid = threadIdx.x;
    for (int o0 = 0; o0 < S; o0 += 256) {
        const int o = o0 + tid;
        const bool ok = o < S;
        const int ca = ok ? descA[o] : 0, cb = ok ? descB[o] : 0;
        cxd acc = {0, 0};
        for (int64_t p = p0; p < p1; ++p) {  // entry order: fixed, reproducible
            const uint32_t lin = ent[p];     // uniform over the workgroup
            const uint32_t c = lin / (uint32_t)n, r = lin - c * (uint32_t)n;
            const cxd qa = {Qhat[2 * (r + (size_t)ca * n)], Qhat[2 * (r + (size_t)ca * n) + 1]};
            const cxd qb = {Qhat[2 * (c + (size_t)cb * n)], Qhat[2 * (c + (size_t)cb * n) + 1]};
            const cxd pr = cx_cmul(qa, qb);
            acc.re += pr.re;
            acc.im += pr.im;
        }
        if (ok) {
            if (sqrt(acc.re * acc.re + acc.im * acc.im) < atol) acc = {0, 0};
            out[2 * ((size_t)blockIdx.x * S + o)] = acc.re;
            out[2 * ((size_t)blockIdx.x * S + o) + 1] = acc.im;
        }
    }
}
void launch_cx_basis_image_sorted(hipStream_t s, int64_t n, int64_t d, int64_t S, const uint32_t* ent, const int64_t* cls_ptr,
                                  const double* Qhat, const int32_t* descA, const int32_t* descB, double atol, double* out) {
    cx_basis_image_sorted_kernel<<<(unsigned)d, 256, 0, s>>>((int)n, (int)S, ent, cls_ptr, Qhat, descA, descB, atol, out);
}

bool complex_set_device_attributes() {
    bool ok = true;
    ok &= hipSuccess == hipFuncSetAttribute(reinterpret_cast<const void*>(&cx_heev_jacobi64_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                        150 * 1024);
    ok &= hipSuccess == hipFuncSetAttribute(reinterpret_cast<const void*>(&cx_block_norms_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                        64 * 1024);
    ok &= hipSuccess == hipFuncSetAttribute(reinterpret_cast<const void*>(&cx_irreducible_general_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                        144 * 1024);
    return ok;
}

void launch_cx_gather_herm(hipStream_t s, int64_t n, const uint32_t* L, uint64_t key, double* Hr, double* Hi) {
    cx_gather_herm_kernel<<<(unsigned)((n * n + 255) / 256), 256, 0, s>>>((int)n, L, key, Hr, Hi);
}
void launch_cx_heev(hipStream_t s, int64_t n, const double* Hr, const double* Hi, double* w, double* Vr, double* Vi, int* info) {
    const int m = (int)((n + 1) & ~int64_t(1)), half = m / 2, ldl = m | 1;
    int threads = (half * half + 63) / 64 * 64;
    if (threads < 64) threads = 64;
    const size_t lds = (size_t)4 * ldl * m * 8 + (size_t)(m - 1) * half * 4 + 64;
    cx_heev_jacobi64_kernel<<<1, threads, lds, s>>>((int)n, Hr, Hi, w, Vr, Vi, info);
}
void launch_cx_block_norms(hipStream_t s, int64_t n, const double* Hr, const double* Hi, const double* Vr, const double* Vi,
                           const int32_t* space_of, int neig, unsigned long long* norms) {
    cx_block_norms_kernel<<<1, 1024, (size_t)2 * n * n * 8, s>>>((int)n, Hr, Hi, Vr, Vi, space_of, neig, norms);
}
void launch_cx_irreducible(hipStream_t s, int64_t n, const double* Hr, const double* Hi, const double* Vr, const double* Vi,
                           const int32_t* desc, int ncols, double atol, double* Qhat) {
    cx_irreducible_kernel<<<ncols, 256, 0, s>>>((int)n, Hr, Hi, Vr, Vi, desc, atol, Qhat);
}
void launch_cx_basis_image(hipStream_t s, int64_t n, int64_t d, int64_t S, const uint32_t* L, const double* Qhat,
                           const int32_t* descA, const int32_t* descB, double atol, double* out) {
    cx_basis_image_kernel<<<(unsigned)d, 256, 0, s>>>((int)n, (int)S, L, Qhat, descA, descB, atol, out);
}

}  // namespace sdpsr
