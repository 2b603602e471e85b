// Eigendecomposition T = Z diag(w) Z' of the symmetric tridiagonal matrix the tridiagonalisation leaves behind
// (second phase of eigen(A), src/eigen_decomposition.jl:246): Cuppen's divide and conquer with the eigenvectors of
// Gu and Eisenstat -- the published algorithm of LAPACK's dstedc / dlaed0-4 -- laid out for gfx950:
//   * T (order n, padded to ld = a multiple of 128 by decoupled dummy entries that sort last) is torn into leaves of
//     32: one workgroup per leaf solves it by the parallel Jacobi of jacobi64.h; the tree above is binary over the
//     leaves inside a block of 128 and over the blocks of 128 beyond, so every product from the second level on is
//     a multiple-of-128 problem for the fp64 MFMA GEMM of kernels_gemm.hip;
//   * the eigenvector matrix is kept TRANSPOSED (row j = eigenvector j, block diagonal over the nodes): the vector z
//     of a merge -- last row of Q1, first row of Q2 -- is two contiguous columns, and the update is the GEMM form
//     this library has, QT_new = U' QT_old (the root writes Z = QT_old' U, untransposed);
//   * a merge = D + rho z z' of the two sorted halves: merge by binary search; deflation (negligible z_i; close
//     poles rotated into each other) in parallel over the RUNS of close poles, sequential inside a run; one WAVE per
//     secular root: the root is bracketed next to its closer pole d_K and the shift mu = lambda - d_K is found by
//     bisection on its BIT PATTERN (63 evaluations of the secular function, the exact nearest double, no iteration
//     that could stall -- dlaed4's rational interpolation needs ~6 evaluations but pages of safeguards); the
//     differences d_i - lambda_j are always formed as (d_i - d_K) - mu; z is recomputed from the roots (Loewner's
//     formula) so that the vectors zhat_i / (d_i - lambda_j) are orthogonal to working precision; the rotations of
//     the deflation are folded into the rows of U, so QT_old is only ever read by the GEMM.
// tools/probes/dc_prototype.py is the same algorithm in NumPy (validated there on random, graded, Wilkinson and
// clustered spectra before this port).  rocSOLVER's stedc (sdpsr_opts.eig_driver = 5 keeps it) spends 3.3 of its
// 4.3 ms at n = 1024 in six one-workgroup merge kernels.
#include <cstdint>
#include <vector>
#include "sdpsr_internal.h"
#include "jacobi64.h"

namespace sdpsr {

constexpr int DC_LEAF = 32;
constexpr int DC_THREADS = 256;

struct DcMerge {
    int a, n1, n2, pad;  // node [a, a + n1) joins [a + n1, a + n1 + n2)
};

struct DcArgs {
    int n, ld;
    double* D;       // ld: eigenvalues of the nodes of the level being merged
    double* Dn;      // ld: of the next level
    double* E;       // ld: scaled off-diagonals
    double* scale;   // [0]: norm of T
    double* QTo;     // ld x ld: transposed eigenvectors of the nodes (block diagonal)
    double* U;       // ld x ld: U of every merge of the level (block diagonal)
    // per merge, stored from index a of the merge's range
    double *dl, *wz, *dfval, *lam, *mu, *zhat, *rho;
    int *ndorig, *dforig, *Kidx, *meta;  // meta[4 a + 0..3] = k, number of rotations, number of runs, -
    double* rotc;    // 2 ld: (c, s) of rotation i
    int* rotab;      // 2 ld: (deflated row, partner row)
    int* runoff;     // 2 ld: rotation index where run t of merge q (of the level) starts: runoff[a + q + t]; N + 1 entries per merge
    const DcMerge* desc;
    int nmerge;
};

__device__ __forceinline__ double dc_wave_sum(double x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}

// ---------------------------------------------------------------------------
// scaling, tearing, padding
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(DC_THREADS)
dc_prepare_kernel(int n, int ld, const double* __restrict__ d, const double* __restrict__ e, double* __restrict__ D,
                  double* __restrict__ E, double* __restrict__ scale) {
    __shared__ double s_red[DC_THREADS / 64];
    __shared__ double s_scale;
    double mx = 0.0;
    for (int i = threadIdx.x; i < n; i += DC_THREADS) {
        mx = fmax(mx, fabs(d[i]));
        if (i < n - 1) mx = fmax(mx, fabs(e[i]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        double m = fmax(fmax(s_red[0], s_red[1]), fmax(s_red[2], s_red[3]));
        if (!(m > 0.0)) m = 1.0;  // T = 0 (or NaN): nothing to scale
        s_scale = m;
        scale[0] = m;
    }
    __syncthreads();
    const double inv = 1.0 / s_scale;
    for (int i = threadIdx.x; i < ld; i += DC_THREADS) {
        const double ei = (i < n - 1) ? e[i] * inv : 0.0;
        E[i] = ei;
        // T = diag(T1, T2) + |e| v v' at every leaf boundary b: d[b-1] -= |e[b-1]|, d[b] -= |e[b-1]|
        // dummies: decoupled, distinct, above every eigenvalue of the scaled T (<= 3) but of its size -- the deflation
        // tolerance is relative to the largest value in a merge
        double di = (i < n) ? d[i] * inv : 3.25 + 0.5 * (double)(i - n + 1) / 129.0;
        if (i < n) {
            if ((i + 1) % DC_LEAF == 0 && i < n - 1) di -= fabs(ei);
            if (i % DC_LEAF == 0 && i > 0) di -= fabs(e[i - 1] * inv);
        }
        D[i] = di;
    }
}

// ---------------------------------------------------------------------------
// leaves: Jacobi on the 32 x 32 tridiagonal, eigenvalues ascending, QT rows = eigenvectors
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(DC_THREADS)
dc_leaf_kernel(int ld, double* __restrict__ D, const double* __restrict__ E, double* __restrict__ QT) {
    constexpr int m = DC_LEAF, ldl = DC_LEAF | 1;
    __shared__ __attribute__((aligned(16))) double sA[4 * ldl * m];
    __shared__ double s_red[2 * DC_THREADS / 64];
    __shared__ int s_rank[m];
    __shared__ double s_w[m];
    const int a = DC_LEAF * blockIdx.x, tid = threadIdx.x;
    double* sV = sA + ldl * m;
    for (int t = tid; t < m * m; t += DC_THREADS) {
        const int j = t / m, i = t - j * m;
        double v = 0.0;
        if (i == j) v = D[a + i];
        else if (i == j + 1) v = E[a + j];
        else if (j == i + 1) v = E[a + i];
        sA[i + j * ldl] = v;
        sV[i + j * ldl] = (i == j) ? 1.0 : 0.0;
    }
    __syncthreads();
    jacobi64_sweeps_pp(m, m, ldl, sA, sV, sV + ldl * m, sV + 2 * ldl * m, s_red);
    __syncthreads();
    if (tid < m) {
        const double li = sA[tid + tid * ldl];
        int rk = 0;
        for (int j = 0; j < m; ++j) {
            const double lj = sA[j + j * ldl];
            rk += (lj < li) || (lj == li && j < tid);
        }
        s_rank[tid] = rk;
        s_w[tid] = li;
    }
    __syncthreads();
    if (tid < m) D[a + s_rank[tid]] = s_w[tid];
    for (int t = tid; t < m * m; t += DC_THREADS) {
        const int j = t / m, i = t - j * m;  // eigenvector j, component i
        QT[(int64_t)(a + s_rank[j]) + (int64_t)(a + i) * ld] = sV[i + j * ldl];
    }
}

// ---------------------------------------------------------------------------
// merge, step 1: z, merged order, deflation
// ---------------------------------------------------------------------------
// exclusive scan of flags f[0..N) (global ints) into pos[0..N); returns the total.  One workgroup.
__device__ int dc_block_scan(int N, const int* __restrict__ f, int* __restrict__ pos, int* s_part) {
    int base = 0;
    for (int c0 = 0; c0 < N; c0 += DC_THREADS) {
        const int i = c0 + threadIdx.x;
        const int v = (i < N) ? f[i] : 0;
        int x = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int y = __shfl_up(x, o, 64);
            if ((int)(threadIdx.x & 63) >= o) x += y;
        }
        if ((threadIdx.x & 63) == 63) s_part[threadIdx.x >> 6] = x;
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) woff += s_part[w];
        const int tot = s_part[0] + s_part[1] + s_part[2] + s_part[3];
        if (i < N) pos[i] = base + woff + x - v;
        base += tot;
        __syncthreads();
    }
    return base;
}

// scratch of the setup kernel, per merge range [a, a + N): sorted values, sorted z, sorted original index, flags
struct DcScratch {
    double *sd, *sz;
    int *so, *f0, *f1, *p0;
};

__global__ void __launch_bounds__(DC_THREADS)
dc_setup_kernel(DcArgs g, DcScratch w) {
    __shared__ int s_part[4];
    __shared__ double s_redd[4];
    __shared__ double s_tol;
    __shared__ int s_all;
    const DcMerge m = g.desc[blockIdx.x];
    const int a = m.a, n1 = m.n1, N = m.n1 + m.n2, tid = threadIdx.x;
    const int64_t ld = g.ld;
    const double* __restrict__ D = g.D + a;
    double* sd = w.sd + a;
    double* sz = w.sz + a;
    int* so = w.so + a;
    int* f0 = w.f0 + a;
    int* f1 = w.f1 + a;
    int* p0 = w.p0 + a;
    const double e = g.E[a + n1 - 1];
    const double sgn = e >= 0 ? 1.0 : -1.0;
    const double rho = 2.0 * fabs(e);
    // merged ascending order of the two sorted halves (ties: first half first); z = [last column of QT1; sgn first column of QT2] / sqrt 2
    double dmax = 0.0, zmax = 0.0;
    for (int i = tid; i < N; i += DC_THREADS) {
        const double di = D[i];
        int lo, hi;
        if (i < n1) {  // count of second-half values < di
            lo = n1;
            hi = N;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (D[mid] < di) lo = mid + 1;
                else hi = mid;
            }
            lo = i + (lo - n1);
        } else {  // count of first-half values <= di
            lo = 0;
            hi = n1;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (D[mid] <= di) lo = mid + 1;
                else hi = mid;
            }
            lo = (i - n1) + lo;
        }
        const double zi = (i < n1 ? g.QTo[(int64_t)(a + i) + (int64_t)(a + n1 - 1) * ld] : sgn * g.QTo[(int64_t)(a + i) + (int64_t)(a + n1) * ld]) *
                          0.70710678118654752440;
        sd[lo] = di;
        sz[lo] = zi;
        so[lo] = i;
        dmax = fmax(dmax, fabs(di));
        zmax = fmax(zmax, fabs(zi));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        dmax = fmax(dmax, __shfl_xor(dmax, o, 64));
        zmax = fmax(zmax, __shfl_xor(zmax, o, 64));
    }
    if ((tid & 63) == 0) {
        s_redd[tid >> 6] = dmax;
    }
    __syncthreads();
    const double dmx = fmax(fmax(s_redd[0], s_redd[1]), fmax(s_redd[2], s_redd[3]));
    __syncthreads();
    if ((tid & 63) == 0) s_redd[tid >> 6] = zmax;
    __syncthreads();
    const double zmx = fmax(fmax(s_redd[0], s_redd[1]), fmax(s_redd[2], s_redd[3]));
    if (tid == 0) {
        s_tol = 8.0 * 2.220446049250313e-16 * fmax(dmx, zmx);
        s_all = (rho * zmx <= 8.0 * 2.220446049250313e-16 * fmax(dmx, zmx)) ? 1 : 0;  // nothing couples: only the order changes
        g.rho[a] = rho;
    }
    __syncthreads();
    const double tol = s_tol;
    const bool all_defl = s_all != 0;
    // survivors of the z test, compacted in sorted order: p0[i] = position among the survivors
    for (int i = tid; i < N; i += DC_THREADS) f0[i] = (!all_defl && rho * fabs(sz[i]) > tol) ? 1 : 0;
    __syncthreads();
    const int ns = dc_block_scan(N, f0, p0, s_part);
    // survivor list in f1[0..ns) (sorted positions)
    for (int i = tid; i < N; i += DC_THREADS)
        if (f0[i]) f1[p0[i]] = i;
    __syncthreads();
    // close pairs of consecutive survivors (original values): runs of them are walked sequentially, one lane per run.
    // status, per sorted position: 0 = deflated by the z test, 1 = not deflated, 2 = deflated by a rotation (partner and
    // angle stored at the position).  The walk updates sd / sz of the positions it touches.
    int* stat = f0;  // f0 is rewritten below: 1 for every survivor first
    int* runflag = p0;  // run start marks over the survivor list
    for (int t = tid; t < ns; t += DC_THREADS) {
        bool start = true;
        if (t > 0) {
            const int ip = f1[t - 1], ic = f1[t];
            double s = sz[ip], c = sz[ic];
            const double tau = hypot(c, s);
            c /= tau;
            s = -s / tau;
            start = !(fabs((sd[ic] - sd[ip]) * c * s) <= tol);
        }
        runflag[t] = start ? 1 : 0;
    }
    __syncthreads();
    double* rotc = g.rotc + 2 * a;
    int* rotab = g.rotab + 2 * a;
    for (int t = tid; t < ns; t += DC_THREADS) {
        if (!runflag[t]) continue;
        int pj = f1[t];
        double zp = sz[pj], dp = sd[pj];
        for (int u = t + 1; u < ns && !runflag[u]; ++u) {
            const int ic = f1[u];
            double s = zp, c = sz[ic];
            const double tau = hypot(c, s);
            const double tt = sd[ic] - dp;
            c /= tau;
            s = -s / tau;
            if (fabs(tt * c * s) <= tol) {  // rotate pole pj into ic: pj deflates
                const double dnew_p = dp * c * c + sd[ic] * s * s;
                const double dnew_c = dp * s * s + sd[ic] * c * c;
                sd[pj] = dnew_p;
                sz[pj] = 0.0;
                stat[pj] = 2;
                rotc[2 * pj] = c;  // (stored at the deflated position; compacted below)
                rotc[2 * pj + 1] = s;
                rotab[2 * pj] = so[pj];
                rotab[2 * pj + 1] = so[ic];
                zp = tau;
                dp = dnew_c;
            } else {  // pj stays
                sd[pj] = dp;
                sz[pj] = zp;
                zp = sz[ic];
                dp = sd[ic];
            }
            pj = ic;
        }
        sd[pj] = dp;
        sz[pj] = zp;
    }
    __syncthreads();
    // compaction.  Rotations keep the order of their (deflated) sorted positions = the order a sequential scan makes
    // them in; run t owns the rotations runoff[t] .. runoff[t + 1] - 1.
    for (int t = tid; t < ns; t += DC_THREADS)
        if (runflag[t]) stat[f1[t]] |= 4;  // run starts, marked on sorted positions (f1 / p0 are free after this)
    __syncthreads();
    for (int i = tid; i < N; i += DC_THREADS) f1[i] = ((stat[i] & 3) == 2) ? 1 : 0;
    __syncthreads();
    const int nrot = dc_block_scan(N, f1, p0, s_part);
    int* rix = g.Kidx + a;  // (free until the secular kernel runs): rotation index of a sorted position
    for (int i = tid; i < N; i += DC_THREADS) {
        rix[i] = p0[i];
        f1[i] = (stat[i] & 4) ? 1 : 0;
    }
    __syncthreads();
    const int nrun = dc_block_scan(N, f1, p0, s_part);
    int* runoff = g.runoff + a + blockIdx.x;  // N + 1 entries per merge
    for (int i = tid; i < N; i += DC_THREADS)
        if (stat[i] & 4) runoff[p0[i]] = rix[i];
    if (tid == 0) runoff[nrun] = nrot;
    // rotations moved from their sorted position i to their index rix[i] <= i: a chunk is read, then written
    for (int c0 = 0; c0 < N; c0 += DC_THREADS) {
        const int i = c0 + tid;
        const bool isrot = i < N && (stat[i] & 3) == 2;
        double rc0 = 0, rc1 = 0;
        int ra = 0, rb = 0;
        if (isrot) {
            rc0 = rotc[2 * i];
            rc1 = rotc[2 * i + 1];
            ra = rotab[2 * i];
            rb = rotab[2 * i + 1];
        }
        __syncthreads();
        if (isrot) {
            const int q = rix[i];
            rotc[2 * q] = rc0;
            rotc[2 * q + 1] = rc1;
            rotab[2 * q] = ra;
            rotab[2 * q + 1] = rb;
        }
        __syncthreads();
    }
    // not deflated -> (dl, wz, ndorig), ascending; deflated -> (dfval, dforig)
    for (int i = tid; i < N; i += DC_THREADS) f1[i] = ((stat[i] & 3) == 1) ? 1 : 0;
    __syncthreads();
    const int k = dc_block_scan(N, f1, p0, s_part);
    for (int i = tid; i < N; i += DC_THREADS) {
        if ((stat[i] & 3) == 1) {
            const int q = p0[i];
            g.dl[a + q] = sd[i];
            g.wz[a + q] = sz[i];
            g.ndorig[a + q] = so[i];
        } else {
            const int q = i - p0[i];  // deflated entries in front of position i
            g.dfval[a + q] = sd[i];
            g.dforig[a + q] = so[i];
        }
    }
    if (tid == 0) {
        g.meta[4 * a + 0] = k;
        g.meta[4 * a + 1] = nrot;
        g.meta[4 * a + 2] = nrun;
    }
}

// ---------------------------------------------------------------------------
// merge, step 2: one wave per secular root.  TPL > 0: the root's terms (w_i^2, d_i - d_K) sit in TPL registers per
// lane for the 63 evaluations (k <= 64 TPL); TPL = 0: read from memory every time (k beyond 1024).
// ---------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ double dc_dpp(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double dc_readlane(double x, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l), __builtin_amdgcn_readlane(__double2loint(x), l));
}
// all 64 lanes active; uniform result (quad permutes, row rotations, four read-lanes)
__device__ __forceinline__ double dc_wave_sum_dpp(double x) {
    x += dc_dpp<0xB1>(x);
    x += dc_dpp<0x4E>(x);
    x += dc_dpp<0x124>(x);
    x += dc_dpp<0x128>(x);
    return (dc_readlane(x, 0) + dc_readlane(x, 16)) + (dc_readlane(x, 32) + dc_readlane(x, 48));
}
// 1 / x from the hardware seed and two Newton steps (the secular function is evaluated ~64 k times per root; the
// IEEE division sequence is five times the instructions).  Out of the seed's range: the division.
__device__ __forceinline__ double dc_recip(double x) {
    const double ax = fabs(x);
    if (ax > 1e-290 && ax < 1e290) {
        double r = __builtin_amdgcn_rcp(x);
        r = fma(fma(-x, r, 1.0), r, r);
        r = fma(fma(-x, r, 1.0), r, r);
        return r;
    }
    return 1.0 / x;
}

template <int TPL>
__global__ void __launch_bounds__(DC_THREADS)
dc_secular_kernel(DcArgs g) {
    const DcMerge m = g.desc[blockIdx.y];
    const int a = m.a;
    const int k = g.meta[4 * a + 0];
    const int j = blockIdx.x * (DC_THREADS / 64) + (threadIdx.x >> 6);
    if (j >= k) return;  // wave-uniform
    const int lane = threadIdx.x & 63;
    const double* __restrict__ dl = g.dl + a;
    const double* __restrict__ wz = g.wz + a;
    const double rho = g.rho[a];
    constexpr int NR = TPL > 0 ? TPL : 1;
    double w2[NR], dd[NR];  // this lane's terms: w_i^2 (0 beyond k) and d_i - d_K
    if (TPL > 0) {
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            const int i = lane + 64 * q;
            const double w = i < k ? wz[i] : 0.0;
            w2[q] = w * w;
            dd[q] = i < k ? dl[i] : 1e300;  // (terms beyond k: 0 over a denominator that never vanishes)
        }
    }
    // 1 + rho sum_i w_i^2 / ((d_i - d_K) - mu); dK = 0 after the terms have been shifted
    auto gfun = [&](double dK, double mu) -> double {
        double s = 0.0;
        if (TPL > 0) {
#pragma unroll
            for (int q = 0; q < NR; ++q) s = fma(w2[q], dc_recip((dd[q] - dK) - mu), s);
        } else {
            for (int i = lane; i < k; i += 64) {
                const double w = wz[i];
                s = fma(w * w, dc_recip((dl[i] - dK) - mu), s);
            }
        }
        return fma(rho, dc_wave_sum_dpp(s), 1.0);
    };
    int K;
    double hi, sign;
    if (j < k - 1) {
        const double dj = dl[j];
        const double half = 0.5 * (dl[j + 1] - dj);
        const double fm = gfun(dj, half);
        if (fm >= 0) {
            K = j;
            sign = 1.0;
        } else {
            K = j + 1;
            sign = -1.0;
        }
        hi = half;
    } else {
        double s = 0.0;
        if (TPL > 0) {
#pragma unroll
            for (int q = 0; q < NR; ++q) s += w2[q];
        } else {
            for (int i = lane; i < k; i += 64) s += wz[i] * wz[i];
        }
        K = k - 1;
        sign = 1.0;
        hi = rho * dc_wave_sum_dpp(s);
    }
    const double dK = dl[K];
    if (TPL > 0) {
#pragma unroll
        for (int q = 0; q < NR; ++q) dd[q] = (lane + 64 * q < k) ? dd[q] - dK : 1e300;
    }
    // bisection on the bit pattern of t = |mu| in (0, hi]:  sign = +1: g(t) < 0 below the root;  sign = -1 (mu = -t):
    // g(-t) > 0 below the root.  (A NaN -- a difference of 0 at the far end of the bracket -- reads as "below".)
    // (Measured and not kept: Newton steps after ten halvings, galloping from the end Newton arrives at -- 20 evaluations
    // per root on average instead of 63, but up to 45 for one root in nine, each 1.5 x the work (derivative, second
    // reduction, division): with one wave per SIMD the launch takes as long as its slowest root, 89 us against 70.)
    long long lb = 0, hb = __double_as_longlong(hi);
    while (hb - lb > 1) {
        const long long mb = lb + ((hb - lb) >> 1);
        const double t = __longlong_as_double(mb);
        const double val = gfun(TPL > 0 ? 0.0 : dK, sign * t);
        const bool upper = sign > 0 ? (val >= 0) : (val <= 0);
        if (upper) hb = mb;
        else lb = mb;
    }
    if (lane == 0) {
        const double mu = sign * __longlong_as_double(hb);
        g.Kidx[a + j] = K;
        g.mu[a + j] = mu;
        g.lam[a + j] = dK + mu;
    }
}

// ---------------------------------------------------------------------------
// merge, step 3: zhat_i = sign(z_i) sqrt | delta(i,i) prod_{j != i} delta(i,j) / (d_i - d_j) |,  delta(i,j) = d_i - lambda_j
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(DC_THREADS)
dc_zhat_kernel(DcArgs g) {
    const DcMerge m = g.desc[blockIdx.y];
    const int a = m.a;
    const int k = g.meta[4 * a + 0];
    const int i = blockIdx.x * (DC_THREADS / 64) + (threadIdx.x >> 6);  // one wave per i: partial products per lane
    if (i >= k) return;
    const int lane = threadIdx.x & 63;
    const double* __restrict__ dl = g.dl + a;
    const double di = dl[i];
    double prod = 1.0;
    for (int j = lane; j < k; j += 64) {
        const double delta = (di - dl[g.Kidx[a + j]]) - g.mu[a + j];
        prod *= (j == i) ? delta : delta / (di - dl[j]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) prod *= __shfl_xor(prod, o, 64);
    if (lane == 0) g.zhat[a + i] = copysign(sqrt(fabs(prod)), g.wz[a + i]);
}

// ---------------------------------------------------------------------------
// merge, step 4: the columns of U (new ascending order), rotations of the deflation folded into its rows.
// One workgroup per column; the column lives in LDS (N <= DC_COLMAX) while thread t walks the rotations of run t in
// reverse.  U(:, c) is written once, contiguously.
// ---------------------------------------------------------------------------
constexpr int DC_COLMAX = 8192;

__global__ void __launch_bounds__(DC_THREADS)
dc_build_u_kernel(DcArgs g) {
    extern __shared__ double s_col[];  // N
    __shared__ double s_redd[4];
    __shared__ int s_redi[4];
    const DcMerge m = g.desc[blockIdx.y];
    const int a = m.a, N = m.n1 + m.n2, tid = threadIdx.x;
    const int c = blockIdx.x;
    if (c >= N) return;
    const int k = g.meta[4 * a + 0], nrot = g.meta[4 * a + 1], nrun = g.meta[4 * a + 2];
    const int nd = N - k;
    const double* __restrict__ lam = g.lam + a;
    const double* __restrict__ dfval = g.dfval + a;
    for (int i = tid; i < N; i += DC_THREADS) s_col[i] = 0.0;
    // column c of the work list: the first k are the secular vectors, the rest the deflated unit vectors; its place in
    // the merged ascending order: rank = own position in its list + number of entries of the other list in front
    int cnt = 0;
    double val;
    if (c < k) {
        val = lam[c];
        for (int p = tid; p < nd; p += DC_THREADS) cnt += (dfval[p] <= val) ? 1 : 0;
    } else {
        const int p0 = c - k;
        val = dfval[p0];
        for (int p = tid; p < nd; p += DC_THREADS) cnt += (dfval[p] < val || (dfval[p] == val && p < p0)) ? 1 : 0;
        for (int j = tid; j < k; j += DC_THREADS) cnt += (lam[j] < val) ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    if ((tid & 63) == 0) s_redi[tid >> 6] = cnt;
    __syncthreads();
    const int rank = (c < k ? c : 0) + s_redi[0] + s_redi[1] + s_redi[2] + s_redi[3];
    if (c < k) {
        const double* __restrict__ dl = g.dl + a;
        const double dK = dl[g.Kidx[a + c]], mu = g.mu[a + c];
        double nrm = 0.0;
        for (int i = tid; i < k; i += DC_THREADS) {
            const double u = g.zhat[a + i] / ((dl[i] - dK) - mu);
            nrm = fma(u, u, nrm);
        }
        nrm = dc_wave_sum(nrm);
        if ((tid & 63) == 0) s_redd[tid >> 6] = nrm;
        __syncthreads();
        const double inv = 1.0 / sqrt((s_redd[0] + s_redd[1]) + (s_redd[2] + s_redd[3]));
        for (int i = tid; i < k; i += DC_THREADS) s_col[g.ndorig[a + i]] = g.zhat[a + i] / ((dl[i] - dK) - mu) * inv;
    } else {
        if (tid == 0) s_col[g.dforig[a + (c - k)]] = 1.0;
    }
    __syncthreads();
    // Q_old G_1 ... G_r Utilde: the rotations applied to the rows of the column last first; the runs touch disjoint rows
    if (nrot > 0) {
        const int* __restrict__ runoff = g.runoff + a + blockIdx.y;
        const double* __restrict__ rotc = g.rotc + 2 * a;
        const int* __restrict__ rotab = g.rotab + 2 * a;
        for (int t = tid; t < nrun; t += DC_THREADS) {
            const int r0 = runoff[t], r1 = runoff[t + 1];
            for (int r = r1 - 1; r >= r0; --r) {
                const int ra = rotab[2 * r], rb = rotab[2 * r + 1];
                const double cc = rotc[2 * r], ss = rotc[2 * r + 1];
                const double xa = s_col[ra], xb = s_col[rb];
                s_col[ra] = cc * xa - ss * xb;
                s_col[rb] = ss * xa + cc * xb;
            }
        }
        __syncthreads();
    }
    double* __restrict__ Uc = g.U + (int64_t)a + (int64_t)(a + rank) * g.ld;
    for (int i = tid; i < N; i += DC_THREADS) Uc[i] = s_col[i];
    if (tid == 0) g.Dn[a + rank] = val;
}

// nodes that have no partner at this level: eigenvalues and vectors carried over unchanged
__global__ void __launch_bounds__(DC_THREADS)
dc_carry_kernel(int a, int N, int64_t ld, const double* __restrict__ D, double* __restrict__ Dn, const double* __restrict__ QTo,
                double* __restrict__ QTn) {
    const int c = blockIdx.x;
    for (int i = threadIdx.x; i < N; i += DC_THREADS) QTn[(int64_t)(a + i) + (int64_t)(a + c) * ld] = QTo[(int64_t)(a + i) + (int64_t)(a + c) * ld];
    if (threadIdx.x == 0) Dn[a + c] = D[a + c];
}

// C (64 x 64) = A' B per merge of the 64-level (A = U block, B = QT block): below the MFMA kernel's 128 tiles
__global__ void __launch_bounds__(DC_THREADS)
dc_gemm64_kernel(int64_t ld, const double* __restrict__ A, const double* __restrict__ B, double* __restrict__ C) {
    __shared__ double sA[64 * 65], sB[64 * 65];
    const int64_t off = (int64_t)64 * blockIdx.x * (ld + 1);
    const int tid = threadIdx.x;
    for (int t = tid; t < 64 * 64; t += DC_THREADS) {
        const int j = t >> 6, i = t & 63;
        sA[i + j * 65] = A[off + i + (int64_t)j * ld];
        sB[i + j * 65] = B[off + i + (int64_t)j * ld];
    }
    __syncthreads();
    const int r0 = (tid & 15) * 4, c0 = (tid >> 4) * 4;  // 4 x 4 outputs per thread: C[r, c] = sum_k A[k, r] B[k, c]
    double acc[4][4] = {};
    for (int kk = 0; kk < 64; ++kk) {
        double av[4], bv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            av[q] = sA[kk + (r0 + q) * 65];
            bv[q] = sB[kk + (c0 + q) * 65];
        }
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[p][q] = fma(av[p], bv[q], acc[p][q]);
    }
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) C[off + (r0 + p) + (int64_t)(c0 + q) * ld] = acc[p][q];
}

// eigenvalues back to the scale of T; the dummy part of Z cleared (rows / columns n .. ld-1)
__global__ void __launch_bounds__(DC_THREADS)
dc_finish_kernel(int n, int ld, const double* __restrict__ D, const double* __restrict__ scale, double* __restrict__ w,
                 double* __restrict__ Z) {
    const double sc = scale[0];
    const int64_t tot = (int64_t)ld * ld;
    for (int64_t t = (int64_t)blockIdx.x * DC_THREADS + threadIdx.x; t < tot; t += (int64_t)gridDim.x * DC_THREADS) {
        const int j = (int)(t / ld), i = (int)(t - (int64_t)j * ld);
        if (i >= n || j >= n) Z[t] = 0.0;
    }
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < n; i += DC_THREADS) w[i] = D[i] * sc;
}

void launch_gemm_tn_f64(hipStream_t s, int64_t m, int64_t n, int64_t k, const double* A, int64_t lda, const double* B, int64_t ldb, double* C,
                        int64_t ldc, int batch, int64_t strideA, int64_t strideB, int64_t strideC);

size_t stedc_workspace_bytes(int64_t ld) {
    // doubles: D, Dn, E, scale(8), dl, wz, dfval, lam, mu, zhat, rho, sd, sz (13 ld) + rotc (2 ld); ints: ndorig, dforig, Kidx, so, f0, f1, p0,
    // runoff (2 ld) (9 ld) + meta (4 ld) + rotab (2 ld); descriptors
    return (size_t)(16 * ld + 64) * 8 + (size_t)(16 * ld + 64) * 4 + (size_t)(ld / DC_LEAF + 8) * 2 * sizeof(DcMerge);
}

struct DcLevel {
    int first, count;
    std::vector<std::pair<int, int>> carry;  // (a, n) of a node without partner
};
// the tree: nodes = leaves of 32; every level pairs neighbours (32 -> 64 -> 128 inside the blocks of 128, then the blocks)
static void stedc_plan(int64_t ld, std::vector<DcMerge>& desc, std::vector<DcLevel>& levels) {
    std::vector<std::pair<int, int>> nodes;
    for (int a0 = 0; a0 < ld; a0 += DC_LEAF) nodes.push_back({a0, DC_LEAF});
    desc.clear();
    levels.clear();
    while (nodes.size() > 1) {
        DcLevel L;
        L.first = (int)desc.size();
        std::vector<std::pair<int, int>> nxt;
        size_t i = 0;
        for (; i + 1 < nodes.size(); i += 2) {
            desc.push_back({nodes[i].first, nodes[i].second, nodes[i + 1].second, 0});
            nxt.push_back({nodes[i].first, nodes[i].second + nodes[i + 1].second});
        }
        if (i < nodes.size()) {
            L.carry.push_back(nodes[i]);
            nxt.push_back(nodes[i]);
        }
        L.count = (int)desc.size() - L.first;
        levels.push_back(L);
        nodes.swap(nxt);
    }
}
// the merge descriptors of order ld, for the caller to upload to stedc_descriptor_slot(ws, ld) before launch_stedc
size_t stedc_descriptors(int64_t ld, std::vector<int>& out) {
    std::vector<DcMerge> desc;
    std::vector<DcLevel> levels;
    stedc_plan(ld, desc, levels);
    out.resize(desc.size() * 4);
    for (size_t i = 0; i < desc.size(); ++i) {
        out[4 * i] = desc[i].a;
        out[4 * i + 1] = desc[i].n1;
        out[4 * i + 2] = desc[i].n2;
        out[4 * i + 3] = 0;
    }
    return desc.size() * sizeof(DcMerge);
}
void* stedc_descriptor_slot(void* ws, int64_t ld) { return (char*)ws + (size_t)(16 * ld + 64) * 8 + (size_t)(16 * ld + 64) * 4; }
bool stedc_set_device_attributes() {
    bool ok = true;
    ok &= hipSuccess == hipFuncSetAttribute(reinterpret_cast<const void*>(&dc_build_u_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, DC_COLMAX * 8);
    return ok;
}

// d, e: device (n, n-1); w: device n (ascending); Z: ld x ld (eigenvectors in the leading n x n, rest zero); W1, W2:
// two more ld x ld buffers; ws: stedc_workspace_bytes(ld) with the descriptors of stedc_descriptors(ld) uploaded to
// stedc_descriptor_slot(ws, ld).
bool launch_stedc(hipStream_t s, int64_t n, int64_t ld, const double* d, const double* e, double* w, double* Z, double* W1, double* W2,
                  void* ws) {
    if (n < 1 || ld % 128 != 0 || ld < n || ld > DC_COLMAX) return false;
    char* p = (char*)ws;
    auto takeD = [&](size_t cnt) {
        double* r = (double*)p;
        p += cnt * 8;
        return r;
    };
    auto takeI = [&](size_t cnt) {
        int* r = (int*)p;
        p += cnt * 4;
        return r;
    };
    DcArgs g;
    g.n = (int)n;
    g.ld = (int)ld;
    double* D0 = takeD(ld);
    double* D1 = takeD(ld);
    g.E = takeD(ld);
    g.scale = takeD(8);
    g.dl = takeD(ld);
    g.wz = takeD(ld);
    g.dfval = takeD(ld);
    g.lam = takeD(ld);
    g.mu = takeD(ld);
    g.zhat = takeD(ld);
    g.rho = takeD(ld);
    DcScratch sc;
    sc.sd = takeD(ld);
    sc.sz = takeD(ld);
    g.rotc = takeD(2 * ld);
    g.ndorig = takeI(ld);
    g.dforig = takeI(ld);
    g.Kidx = takeI(ld);
    sc.so = takeI(ld);
    sc.f0 = takeI(ld);
    sc.f1 = takeI(ld);
    sc.p0 = takeI(ld);
    g.runoff = takeI(2 * ld + 8);
    g.meta = takeI(4 * ld);
    g.rotab = takeI(2 * ld + 8);
    DcMerge* ddesc = (DcMerge*)stedc_descriptor_slot(ws, ld);
    std::vector<DcMerge> hdesc;
    std::vector<DcLevel> levels;
    stedc_plan(ld, hdesc, levels);
    // QT ping-pong ends in Z: the root writes Z = QT_old' U, so QT_old of the root level is any buffer but Z
    const int nlev = (int)levels.size();
    double* bufs[2] = {W1, W2};
    dc_prepare_kernel<<<1, DC_THREADS, 0, s>>>((int)n, (int)ld, d, e, D0, g.E, g.scale);
    double* QTcur = bufs[0];
    if (hipMemsetAsync(QTcur, 0, (size_t)ld * ld * 8, s) != hipSuccess) return false;
    dc_leaf_kernel<<<(unsigned)(ld / DC_LEAF), DC_THREADS, 0, s>>>((int)ld, D0, g.E, QTcur);
    double* Dcur = D0;
    double* Dnxt = D1;
    if (nlev == 0) return false;  // ld >= 128: at least two levels
    // U needs its own ld x ld buffer: Z serves as U until the root, whose U must not be the output...  the root's
    // output is Z, so the root's U sits in the QT buffer that is free at that point
    for (int lv = 0; lv < nlev; ++lv) {
        const DcLevel& L = levels[lv];
        const bool root = lv == nlev - 1;
        double* QTnext = (QTcur == bufs[0]) ? bufs[1] : bufs[0];
        double* Ubuf = root ? QTnext : Z;
        g.D = Dcur;
        g.Dn = Dnxt;
        g.QTo = QTcur;
        g.U = Ubuf;
        g.desc = ddesc + L.first;
        g.nmerge = L.count;
        int Nmax = 0;
        for (int q = 0; q < L.count; ++q) Nmax = std::max(Nmax, hdesc[L.first + q].n1 + hdesc[L.first + q].n2);
        dc_setup_kernel<<<(unsigned)L.count, DC_THREADS, 0, s>>>(g, sc);
        const unsigned wg4 = (unsigned)((Nmax + 3) / 4);
        if (Nmax <= 64) dc_secular_kernel<1><<<dim3(wg4, (unsigned)L.count), DC_THREADS, 0, s>>>(g);
        else if (Nmax <= 256) dc_secular_kernel<4><<<dim3(wg4, (unsigned)L.count), DC_THREADS, 0, s>>>(g);
        else if (Nmax <= 1024) dc_secular_kernel<16><<<dim3(wg4, (unsigned)L.count), DC_THREADS, 0, s>>>(g);
        else if (Nmax <= 2048) dc_secular_kernel<32><<<dim3(wg4, (unsigned)L.count), DC_THREADS, 0, s>>>(g);
        else if (Nmax <= 4096) dc_secular_kernel<64><<<dim3(wg4, (unsigned)L.count), DC_THREADS, 0, s>>>(g);
        else dc_secular_kernel<0><<<dim3(wg4, (unsigned)L.count), DC_THREADS, 0, s>>>(g);
        dc_zhat_kernel<<<dim3(wg4, (unsigned)L.count), DC_THREADS, 0, s>>>(g);
        dc_build_u_kernel<<<dim3((unsigned)Nmax, (unsigned)L.count), DC_THREADS, (size_t)Nmax * 8, s>>>(g);
        double* out = root ? Z : QTnext;
        // the next level reads blocks of twice the size: their off-diagonal quarters must be zero
        if (!root && hipMemsetAsync(out, 0, (size_t)ld * ld * 8, s) != hipSuccess) return false;
        // products: QT_new = U' QT_old (root: Z = QT_old' U)
        const int N0 = hdesc[L.first].n1 + hdesc[L.first].n2;
        if (N0 == 64) {
            dc_gemm64_kernel<<<(unsigned)L.count, DC_THREADS, 0, s>>>(ld, Ubuf, QTcur, out);
        } else {
            // merges of equal size at a constant stride go in one batched launch
            int q = 0;
            while (q < L.count) {
                const DcMerge& m0 = hdesc[L.first + q];
                const int Nq = m0.n1 + m0.n2;
                int cntq = 1;
                while (q + cntq < L.count) {
                    const DcMerge& mq = hdesc[L.first + q + cntq];
                    if (mq.n1 + mq.n2 != Nq || mq.a != m0.a + cntq * Nq) break;
                    ++cntq;
                }
                const int64_t off = (int64_t)m0.a * (ld + 1), stride = (int64_t)Nq * (ld + 1);
                const double* Aop = root ? QTcur + off : Ubuf + off;
                const double* Bop = root ? Ubuf + off : QTcur + off;
                launch_gemm_tn_f64(s, Nq, Nq, Nq, Aop, ld, Bop, ld, out + off, ld, cntq, stride, stride, stride);
                q += cntq;
            }
        }
        for (const auto& cn : L.carry) dc_carry_kernel<<<(unsigned)cn.second, DC_THREADS, 0, s>>>(cn.first, cn.second, ld, Dcur, Dnxt, QTcur, out);
        QTcur = out;
        std::swap(Dcur, Dnxt);
    }
    dc_finish_kernel<<<256, DC_THREADS, 0, s>>>((int)n, (int)ld, Dcur, g.scale, w, Z);
    return hipGetLastError() == hipSuccess;
}

// Status of the own tridiagonal solver.  NaN / Inf in the tridiagonal input pass through the Jacobi leaves and the secular
// bisection ("below") and can come out as finite garbage eigenpairs with a clean status; the library solvers behind
// eig_driver 1 / 2 / 3 / 5 report SOLVER_ERROR there.  pre = 1 (before the solve): info[0] = 1 if d or e holds a
// non-finite value, else 0; info[1] = 0.  pre = 0 (after it): info[0] |= 1 if an eigenvalue is not finite or the
// values are not ascending.
__global__ void __launch_bounds__(256) dc_check_kernel(int n, const double* __restrict__ a, const double* __restrict__ b, int pre, int* __restrict__ info) {
    __shared__ int bad;
    if (threadIdx.x == 0) bad = 0;
    __syncthreads();
    int f = 0;
    for (int i = threadIdx.x; i < n; i += 256) {
        const double v = a[i];
        if (!(fabs(v) <= 1.79769313486231570e308)) f = 1;                 // NaN or Inf
        if (pre) {
            if (i + 1 < n && !(fabs(b[i]) <= 1.79769313486231570e308)) f = 1;
        } else if (i + 1 < n && !(v <= a[i + 1])) {
            f = 1;                                                         // not ascending (or NaN)
        }
    }
    if (f) bad = 1;
    __syncthreads();
    if (threadIdx.x == 0) {
        info[0] = pre ? bad : (info[0] | bad);
        info[1] = 0;
    }
}
void launch_stedc_check(hipStream_t s, int64_t n, const double* a, const double* b, int pre, int* info) {
    dc_check_kernel<<<1, 256, 0, s>>>((int)n, a, b, pre, info);
}

}  // namespace sdpsr
