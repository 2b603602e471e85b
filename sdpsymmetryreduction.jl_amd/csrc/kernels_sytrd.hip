// Householder tridiagonalisation A = Q T Q' of the generic algebra element for gfx950
// (first phase of eigen(A), src/eigen_decomposition.jl:246; LAPACK dsytrd/dlatrd, lower).
//
// Design (MI355X-first, not a translation of dlatrd's BLAS-2 call sequence):
//  * the full symmetric trailing matrix is kept (both triangles), so the big product
//    p = A v of every column is a set of contiguous column dots: one wave per column,
//    16-byte loads, v staged once per workgroup in LDS, no atomics, no cross-workgroup
//    reduction for p (bitwise reproducible);
//  * two launches per column, both spread over the whole chip:
//      form_kernel (j):   finish W(:, j-1) from the column dots (corrections with the panel
//                         V, W; the v'Av term comes from per-workgroup partials), then form
//                         the updated column a_j = A(:,j) - V W(j,:)' - W V(j,:)' and its
//                         partial squared norms;
//      symv_kernel (j):   every workgroup redundantly finishes the reflector scalars
//                         (beta, tau, scale) from the partial norms, builds v in LDS and
//                         computes its share of p = A v, W'v, V'v and v'p;
//  * after NB columns the trailing matrix gets the rank-2NB update A -= V W' + W V' (both
//    triangles) in one tiled kernel.
// Output is LAPACK-compatible (d, e, tau, reflectors below the subdiagonal of A), so the
// tridiagonal solve (rocSOLVER stedc) and the back-transformation (ormtr) plug in unchanged.
#include <cstdlib>
#include "sdpsr_internal.h"
#include "jacobi64.h"

namespace sdpsr {

constexpr int SY_NB = 32;       // panel width
constexpr int SY_THREADS = 256;

struct SytrdArgs {
    double* A;        // n x n, leading dimension ld (even), full symmetric on entry
    int64_t ld;
    int n;
    double* Vp;       // ld x NB panel of reflectors (explicit, v[j+1] = 1)
    double* Wp;       // ld x NB panel W
    double* p0;       // n: A v
    double* g1;       // NB: W' v
    double* g2;       // NB: V' v
    double* part_norm;  // per form-workgroup partial sum of a[r]^2, r >= j+2
    double* part_vav;   // per symv-workgroup partial of v' (A v)
    double* d;
    double* e;
    double* tau;
};

// ---------------------------------------------------------------------------
// form_kernel: rows r in [j, n); a workgroup owns SY_FROWS = 64 rows, its 4 waves split the
// panel columns (c = wave, wave + 4, ...) so that the 2 x cf strided panel reads of a row are
// spread over four waves and issued in batches; partial sums meet in LDS.
//   do_finish: column jf = j-1 (panel column cf) gets its W column.
//   do_form:   column j is updated with the finished panel columns and its norm partials
//              are produced.  n_vav = number of part_vav entries written by symv(jf).
// ---------------------------------------------------------------------------
constexpr int SY_FROWS = 64;

__global__ void __launch_bounds__(SY_THREADS)
sytrd_form_kernel(SytrdArgs a, int j, int cf, int do_finish, int do_form, int n_vav) {
    __shared__ double s_g1[SY_NB], s_g2[SY_NB], s_wrow[SY_NB + 1], s_vrow[SY_NB + 1];
    __shared__ double s_part[4][2][SY_FROWS];
    __shared__ double s_scal[2];  // tau, alpha2
    const int tid = threadIdx.x;
    const int lane = tid & 63, q = tid >> 6;
    const int64_t ld = a.ld;
    const int n = a.n;
    const int jf = j - 1;
    const int r = j + blockIdx.x * SY_FROWS + lane;

    // independent loads first (they overlap the reductions below)
    double pre_v = 0, pre_p0 = 0, pre_x = 0;
    if (q == 0 && r < n) {
        if (do_finish) {
            pre_v = a.Vp[r + (int64_t)cf * ld];
            pre_p0 = a.p0[r];
        }
        if (do_form) pre_x = a.A[r + (int64_t)j * ld];
    }
    double pv[SY_NB / 4], pw[SY_NB / 4];
#pragma unroll
    for (int u = 0; u < SY_NB / 4; ++u) {
        const int c = q + 4 * u;
        const bool ok = (r < n) && (c < cf);
        pv[u] = ok ? a.Vp[r + (int64_t)c * ld] : 0.0;
        pw[u] = ok ? a.Wp[r + (int64_t)c * ld] : 0.0;
    }

    if (do_finish) {
        if (tid < cf) {
            s_g1[tid] = a.g1[tid];
            s_g2[tid] = a.g2[tid];
        }
        if (q == 1) {  // wave 1: fixed-shape tree reductions (bitwise reproducible)
            double vav = 0;
            for (int b = lane; b < n_vav; b += 64) vav += a.part_vav[b];
            double gg = (lane < cf) ? a.g1[lane] * a.g2[lane] : 0.0;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                vav += __shfl_down(vav, o, 64);
                gg += __shfl_down(gg, o, 64);
            }
            if (lane == 0) {
                const double tau = a.tau[jf];
                const double dot = tau * (vav - 2.0 * gg);  // p'v with p = tau (A v - V g1 - W g2)
                s_scal[0] = tau;
                s_scal[1] = -0.5 * tau * dot;
            }
        }
    }
    if (do_form && tid < cf) {  // rows of the finished panel columns (before this launch)
        s_wrow[tid] = a.Wp[j + (int64_t)tid * ld];
        s_vrow[tid] = a.Vp[j + (int64_t)tid * ld];
    }
    __syncthreads();
    if (do_finish && do_form && tid == 0) {
        // row j of the W column that this launch finishes: needed by every row of the form part
        double s = 0;
        for (int c = 0; c < cf; ++c) s += s_vrow[c] * s_g1[c] + s_wrow[c] * s_g2[c];
        const double vj = a.Vp[j + (int64_t)cf * ld];
        s_wrow[cf] = s_scal[0] * (a.p0[j] - s) + s_scal[1] * vj;
        s_vrow[cf] = vj;
    }

    double sf = 0, sm = 0;
#pragma unroll
    for (int u = 0; u < SY_NB / 4; ++u) {
        const int c = q + 4 * u;
        if (c < cf) {
            if (do_finish) sf += pv[u] * s_g1[c] + pw[u] * s_g2[c];
            if (do_form) sm += pv[u] * s_wrow[c] + pw[u] * s_vrow[c];
        }
    }
    s_part[q][0][lane] = sf;
    s_part[q][1][lane] = sm;
    __syncthreads();
    if (q != 0) return;
    double sq = 0;
    if (r < n) {
        sf = s_part[0][0][lane] + s_part[1][0][lane] + s_part[2][0][lane] + s_part[3][0][lane];
        sm = s_part[0][1][lane] + s_part[1][1][lane] + s_part[2][1][lane] + s_part[3][1][lane];
        double wnew = 0, v = 0;
        if (do_finish) {
            v = pre_v;
            wnew = s_scal[0] * (pre_p0 - sf) + s_scal[1] * v;
            a.Wp[r + (int64_t)cf * ld] = wnew;
            // LAPACK storage of the finished reflector: v below the subdiagonal of column jf
            if (r >= jf + 2) a.A[r + (int64_t)jf * ld] = v;
        }
        if (do_form) {
            double x = pre_x;
            if (do_finish) sm += v * s_wrow[cf] + wnew * s_vrow[cf];
            x -= sm;
            a.A[r + (int64_t)j * ld] = x;
            if (r == j) a.d[j] = x;
            if (r >= j + 2) sq = x * x;
        }
    }
    if (do_form) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sq += __shfl_down(sq, o, 64);
        if (lane == 0) a.part_norm[blockIdx.x] = sq;
    }
}

// ---------------------------------------------------------------------------
// symv_kernel(j): reflector of column j (panel column cf), then column dots.
//   work items: trailing columns k in [j+1, n) -> p0[k];  cf columns of Wp -> g1;  cf columns
//   of Vp -> g2.  n_norm = number of part_norm entries written by form(j).
// ---------------------------------------------------------------------------
template <int SY_MAXV>  // rows of column j held in registers per thread: n <= SY_MAXV * 256
__global__ void __launch_bounds__(SY_THREADS)
sytrd_symv_kernel(SytrdArgs a, int j, int cf, int n_norm) {
    extern __shared__ __attribute__((aligned(16))) double s_v[];  // rows r0 .. n (r0 even)
    __shared__ double s_bcast[3];
    __shared__ double s_red[SY_THREADS / 64];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int64_t ld = a.ld;
    const int n = a.n;
    const int r0 = (j + 1) & ~1;  // even start so that 16-byte loads are aligned
    const int len = n - r0;       // entries of s_v
    // raw column j (unscaled) first, so that these loads overlap the norm reduction
    double raw[SY_MAXV];
    const int nper = (len + SY_THREADS - 1) / SY_THREADS;
#pragma unroll
    for (int u = 0; u < SY_MAXV; ++u) {
        const int t = tid + u * SY_THREADS;
        raw[u] = (u < nper && t < len && r0 + t >= j + 2) ? a.A[(r0 + t) + (int64_t)j * ld] : 0.0;
    }
    if (wave == 0) {
        double xn2 = 0;
        for (int b = lane; b < n_norm; b += 64) xn2 += a.part_norm[b];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) xn2 += __shfl_down(xn2, o, 64);
      if (lane == 0) {
        const double alpha = a.A[(j + 1) + (int64_t)j * ld];
        double beta, tau, scale;
        if (xn2 == 0.0) {  // dlarfg: H = I
            tau = 0.0;
            beta = alpha;
            scale = 0.0;
        } else {
            beta = -copysign(sqrt(alpha * alpha + xn2), alpha);
            tau = (beta - alpha) / beta;
            scale = 1.0 / (alpha - beta);
        }
        s_bcast[0] = beta;
        s_bcast[1] = tau;
        s_bcast[2] = scale;
        if (blockIdx.x == 0) {
            a.e[j] = beta;
            a.tau[j] = tau;
        }
      }
    }
    __syncthreads();
    const double scale = s_bcast[2];
#pragma unroll
    for (int u = 0; u < SY_MAXV; ++u) {
        const int t = tid + u * SY_THREADS;
        if (u < nper && t < len) {
            const int r = r0 + t;
            double v;
            if (r <= j) v = 0.0;
            else if (r == j + 1) v = 1.0;
            else v = raw[u] * scale;
            s_v[t] = v;
            if (blockIdx.x == 0 && r > j) a.Vp[r + (int64_t)cf * ld] = v;
        }
    }
    if ((len & 1) && tid == 0) s_v[len] = 0.0;  // pad for the double2 reads
    __syncthreads();

    const int ncols = n - (j + 1);
    const int nwork = ncols + 2 * cf;
    const int wpb = SY_THREADS / 64;
    double vav = 0;
    const int len2 = (len + 1) >> 1;  // double2 elements
    const double2* sv2 = reinterpret_cast<const double2*>(s_v);
    auto col_ptr = [&](int w) -> const double2* {
        const double* col;
        if (w < ncols) col = a.A + (int64_t)(j + 1 + w) * ld;
        else if (w < ncols + cf) col = a.Wp + (int64_t)(w - ncols) * ld;
        else col = a.Vp + (int64_t)(w - ncols - cf) * ld;
        return reinterpret_cast<const double2*>(col + r0);
    };
    auto emit = [&](int w, double acc) {
        if (w < ncols) {
            a.p0[j + 1 + w] = acc;
            vav += acc * s_v[(j + 1 + w) - r0];
        } else if (w < ncols + cf) {
            a.g1[w - ncols] = acc;
        } else {
            a.g2[w - ncols - cf] = acc;
        }
    };
    // a wave walks two columns at a time: four 16-byte loads in flight per lane, and every v
    // value read from LDS serves both columns.  (The tail double2 of an odd-length column
    // reads one element past row n-1: ld is even and > n there, and the matching s_v pad is 0.)
    const int nwaves = gridDim.x * wpb;
    const int npairs = (nwork + 1) >> 1;
    for (int pr = blockIdx.x * wpb + wave; pr < npairs; pr += nwaves) {
        const int w0 = 2 * pr, w1 = (2 * pr + 1 < nwork) ? 2 * pr + 1 : 2 * pr;
        const double2* c0 = col_ptr(w0);
        const double2* c1 = col_ptr(w1);
        double a00 = 0, a01 = 0, a10 = 0, a11 = 0;
        int t = lane;
        for (; t + 64 < len2; t += 128) {
            const double2 x0 = c0[t], x1 = c0[t + 64];
            const double2 y0 = c1[t], y1 = c1[t + 64];
            const double2 v0 = sv2[t], v1 = sv2[t + 64];
            a00 = fma(x0.x, v0.x, a00);
            a00 = fma(x0.y, v0.y, a00);
            a01 = fma(x1.x, v1.x, a01);
            a01 = fma(x1.y, v1.y, a01);
            a10 = fma(y0.x, v0.x, a10);
            a10 = fma(y0.y, v0.y, a10);
            a11 = fma(y1.x, v1.x, a11);
            a11 = fma(y1.y, v1.y, a11);
        }
        for (; t < len2; t += 64) {
            const double2 x0 = c0[t];
            const double2 y0 = c1[t];
            const double2 v0 = sv2[t];
            a00 = fma(x0.x, v0.x, a00);
            a00 = fma(x0.y, v0.y, a00);
            a10 = fma(y0.x, v0.x, a10);
            a10 = fma(y0.y, v0.y, a10);
        }
        double acc0 = a00 + a01, acc1 = a10 + a11;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            acc0 += __shfl_down(acc0, o, 64);
            acc1 += __shfl_down(acc1, o, 64);
        }
        if (lane == 0) {
            emit(w0, acc0);
            if (w1 != w0) emit(w1, acc1);
        }
    }
    if (lane == 0) s_red[wave] = vav;
    __syncthreads();
    if (tid == 0) a.part_vav[blockIdx.x] = s_red[0] + s_red[1] + s_red[2] + s_red[3];
}

// ---------------------------------------------------------------------------
// rank-2NB update of the trailing matrix:
//   A[r, s] -= sum_c V[r,c] W[s,c] + W[r,c] V[s,c],   r, s >= j1
// Only tiles with r-tile >= s-tile are computed; every value is written to (r,s) AND (s,r), so
// the trailing matrix stays bitwise symmetric.  (That matters: the column-dot symv reads
// A(:,k) where the algebra means row k; with two independently rounded triangles the
// mismatch, of the size of eps * |A| at the deflation panel, is re-injected at every later
// column and the tridiagonalisation of a highly degenerate matrix loses ~5 digits.)
// 64 x 64 tile per workgroup, 4 x 4 outputs per thread, panels staged in LDS.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(SY_THREADS)
sytrd_syr2k_kernel(SytrdArgs a, int j1, int cnt, int ntile) {
    constexpr int CH = 16;  // panel columns staged per pass (4 x 16 x 64 doubles = 32 KiB)
    __shared__ double sVr[CH][64], sWr[CH][64], sVs[CH][64], sWs[CH][64];
    // linear tile index -> (bx >= by) of the lower triangle of tiles
    int by = 0, bx = blockIdx.x;
    {
        // row-wise enumeration: tiles (bx, by) with by <= bx, index = bx*(bx+1)/2 + by
        int t = blockIdx.x;
        int x = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
        while ((x + 1) * (x + 2) / 2 <= t) ++x;
        while (x * (x + 1) / 2 > t) --x;
        bx = x;
        by = t - x * (x + 1) / 2;
    }
    (void)ntile;
    const int tid = threadIdx.x;
    const int64_t ld = a.ld;
    const int n = a.n;
    const int rb = j1 + bx * 64, sb = j1 + by * 64;
    const int tr = (tid & 15) * 4, ts = (tid >> 4) * 4;
    double acc[4][4];
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) acc[x][y] = 0.0;
    for (int c0 = 0; c0 < cnt; c0 += CH) {
        const int cc = (cnt - c0 < CH) ? cnt - c0 : CH;
        __syncthreads();
        for (int t = tid; t < 64 * cc; t += SY_THREADS) {
            const int c = t / 64, q = t % 64;
            const int rr = rb + q, ss = sb + q;
            sVr[c][q] = (rr < n) ? a.Vp[rr + (int64_t)(c0 + c) * ld] : 0.0;
            sWr[c][q] = (rr < n) ? a.Wp[rr + (int64_t)(c0 + c) * ld] : 0.0;
            sVs[c][q] = (ss < n) ? a.Vp[ss + (int64_t)(c0 + c) * ld] : 0.0;
            sWs[c][q] = (ss < n) ? a.Wp[ss + (int64_t)(c0 + c) * ld] : 0.0;
        }
        __syncthreads();
        for (int c = 0; c < cc; ++c) {
            double vr[4], wr[4], vs[4], ws[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                vr[x] = sVr[c][tr + x];
                wr[x] = sWr[c][tr + x];
                vs[x] = sVs[c][ts + x];
                ws[x] = sWs[c][ts + x];
            }
#pragma unroll
            for (int y = 0; y < 4; ++y)
#pragma unroll
                for (int x = 0; x < 4; ++x) acc[x][y] = fma(vr[x], ws[y], fma(wr[x], vs[y], acc[x][y]));
        }
    }
#pragma unroll
    for (int y = 0; y < 4; ++y) {
        const int s = sb + ts + y;
        if (s >= n) continue;
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const int r = rb + tr + x;
            if (r >= n || r < s) continue;  // diagonal tiles: lower part only
            const double val = a.A[r + (int64_t)s * ld] - acc[x][y];
            a.A[r + (int64_t)s * ld] = val;
            if (r != s) a.A[s + (int64_t)r * ld] = val;
        }
    }
}

// ---------------------------------------------------------------------------
// host driver.  A: n x n, ld even and >= n + (n odd ? 1 : 0) so that the 16-byte column
// reads of the symv kernel stay inside the allocation (callers pad ld to a multiple of 128).
// ws: Vp, Wp (ld*NB each), p0 (n), g1/g2 (NB), part_norm/part_vav (<= 4096 each).
// ---------------------------------------------------------------------------
size_t sytrd_workspace_doubles(int64_t n, int64_t ld) {
    return (size_t)2 * ld * SY_NB + (size_t)n + 2 * SY_NB + 2 * 4096 + 64;
}

// per-device kernel attributes, set by sdpsr_create() with the ctx's device current
void sytrd_set_device_attributes() {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&sytrd_symv_kernel<8>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&sytrd_symv_kernel<16>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&sytrd_symv_kernel<32>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&sytrd_symv_kernel<64>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096);
}

static void launch_sytrd_direct(hipStream_t s, int64_t n64, double* A, int64_t ld, double* d, double* e, double* tau,
                                double* ws) {
    const int n = (int)n64;
    SytrdArgs a;
    a.A = A;
    a.ld = ld;
    a.n = n;
    a.Vp = ws;
    a.Wp = a.Vp + ld * SY_NB;
    a.p0 = a.Wp + ld * SY_NB;
    a.g1 = a.p0 + n;
    a.g2 = a.g1 + SY_NB;
    a.part_norm = a.g2 + SY_NB;
    a.part_vav = a.part_norm + 4096;
    a.d = d;
    a.e = e;
    a.tau = tau;
    hipMemsetAsync(ws, 0, sytrd_workspace_doubles(n, ld) * sizeof(double), s);
    if (n == 1) {
        hipMemcpyAsync(d, A, sizeof(double), hipMemcpyDeviceToDevice, s);
        return;
    }
    int n_vav = 0;
    int cf = 0;  // panel column of the column currently being formed
    const size_t lds_v = ((size_t)n + 4) * sizeof(double);
    // column 0: plain form (no panel yet)
    {
        const int nb_form = (n + SY_FROWS - 1) / SY_FROWS;
        sytrd_form_kernel<<<nb_form, SY_THREADS, 0, s>>>(a, 0, 0, 0, 1, 0);
    }
    for (int j = 0; j <= n - 2; ++j) {
        // reflector of column j + column dots
        const int n_norm = (n - j + SY_FROWS - 1) / SY_FROWS;
        const int nwork = (n - j - 1) + 2 * cf;
        int nblk = (nwork + 7) / 8;  // one column pair per wave
        if (nblk > 512) nblk = 512;
        if (nblk < 1) nblk = 1;
        const int rows_left = n - j;
        if (rows_left <= 8 * SY_THREADS)
            sytrd_symv_kernel<8><<<nblk, SY_THREADS, lds_v, s>>>(a, j, cf, n_norm);
        else if (rows_left <= 16 * SY_THREADS)
            sytrd_symv_kernel<16><<<nblk, SY_THREADS, lds_v, s>>>(a, j, cf, n_norm);
        else if (rows_left <= 32 * SY_THREADS)
            sytrd_symv_kernel<32><<<nblk, SY_THREADS, lds_v, s>>>(a, j, cf, n_norm);
        else
            sytrd_symv_kernel<64><<<nblk, SY_THREADS, lds_v, s>>>(a, j, cf, n_norm);
        n_vav = nblk;
        const int jn = j + 1;
        const int rows = n - jn;
        const int nb_form = (rows + SY_FROWS - 1) / SY_FROWS;
        if (cf + 1 < SY_NB && jn <= n - 1) {
            // finish W(:, cf) and form column j+1 inside the same panel
            sytrd_form_kernel<<<nb_form, SY_THREADS, 0, s>>>(a, jn, cf, 1, 1, n_vav);
            ++cf;
        } else {
            // panel complete: finish W, update the trailing matrix, start a new panel
            sytrd_form_kernel<<<nb_form, SY_THREADS, 0, s>>>(a, jn, cf, 1, 0, n_vav);
            const int cnt = cf + 1;
            const int tr = n - jn;
            if (tr > 0) {
                const int nt = (tr + 63) / 64;
                sytrd_syr2k_kernel<<<nt * (nt + 1) / 2, SY_THREADS, 0, s>>>(a, jn, cnt, nt);
                sytrd_form_kernel<<<nb_form, SY_THREADS, 0, s>>>(a, jn, 0, 0, 1, 0);
            }
            cf = 0;
        }
    }
}

// measurement hook: only the symv launches of a full tridiagonalisation (same grid shapes and
// memory traffic, numerically meaningless because form/syr2k are skipped)
void launch_sytrd_symv_sweep(hipStream_t s, int64_t n64, double* A, int64_t ld, double* d, double* e,
                             double* tau, double* ws) {
    const int n = (int)n64;
    SytrdArgs a;
    a.A = A;
    a.ld = ld;
    a.n = n;
    a.Vp = ws;
    a.Wp = a.Vp + ld * SY_NB;
    a.p0 = a.Wp + ld * SY_NB;
    a.g1 = a.p0 + n;
    a.g2 = a.g1 + SY_NB;
    a.part_norm = a.g2 + SY_NB;
    a.part_vav = a.part_norm + 4096;
    a.d = d;
    a.e = e;
    a.tau = tau;
    hipMemsetAsync(ws, 0, sytrd_workspace_doubles(n, ld) * sizeof(double), s);
    const size_t lds_v = ((size_t)n + 4) * sizeof(double);
    for (int j = 0; j <= n - 2; ++j) {
        const int cf = j % SY_NB;
        const int n_norm = (n - j + SY_FROWS - 1) / SY_FROWS;
        const int nwork = (n - j - 1) + 2 * cf;
        int nblk = (nwork + 7) / 8;
        if (nblk > 512) nblk = 512;
        if (nblk < 1) nblk = 1;
        const int rows_left = n - j;
        if (rows_left <= 8 * SY_THREADS)
            sytrd_symv_kernel<8><<<nblk, SY_THREADS, lds_v, s>>>(a, j, cf, n_norm);
        else if (rows_left <= 16 * SY_THREADS)
            sytrd_symv_kernel<16><<<nblk, SY_THREADS, lds_v, s>>>(a, j, cf, n_norm);
        else if (rows_left <= 32 * SY_THREADS)
            sytrd_symv_kernel<32><<<nblk, SY_THREADS, lds_v, s>>>(a, j, cf, n_norm);
        else
            sytrd_symv_kernel<64><<<nblk, SY_THREADS, lds_v, s>>>(a, j, cf, n_norm);
    }
}


// For n up to a few thousand the ~2n + n/32 launches of the tridiagonalisation are shorter than
// the cost of launching them (2-3 us of kernel against 4-5 us of launch): the launch sequence of
// one problem shape is captured once into a hipGraph and replayed.  Key = every value baked into
// the nodes (order, leading dimension, all pointers); the ctx's buffers are grow-only, so the key
// is stable across calls.  A few graphs are kept (generic elements alternate between two or
// three buffers).
// The cache belongs to the ctx (sdpsr_ctx::sytrd_graphs): no process-global state, and calls on
// one ctx are serialised by contract, so no lock is needed.
struct SytrdGraph {
    int64_t n = 0, ld = 0;
    const void *A = nullptr, *d = nullptr, *e = nullptr, *tau = nullptr, *ws = nullptr;
    hipGraphExec_t exec = nullptr;
    uint64_t last_use = 0;
};
struct SytrdGraphCache {
    SytrdGraph slots[6];
    uint64_t clock = 0;
};
void sytrd_graph_cache_destroy(SytrdGraphCache* g) {
    if (!g) return;
    for (auto& sl : g->slots)
        if (sl.exec) hipGraphExecDestroy(sl.exec);
    delete g;
}

void launch_sytrd(sdpsr_ctx* c, int64_t n64, double* A, int64_t ld, double* d, double* e, double* tau, double* ws) {
    hipStream_t s = c->stream;
    const bool no_graph = getenv("SDPSR_NO_GRAPH") != nullptr;
    if (no_graph || n64 < 64 || n64 > 3072) {  // large orders: the kernels outlast their launches
        launch_sytrd_direct(s, n64, A, ld, d, e, tau, ws);
        return;
    }
    if (!c->sytrd_graphs) c->sytrd_graphs = new SytrdGraphCache();
    SytrdGraphCache& gc = *c->sytrd_graphs;
    ++gc.clock;
    SytrdGraph* slot = nullptr;
    for (auto& g : gc.slots)
        if (g.exec && g.n == n64 && g.ld == ld && g.A == A && g.d == d && g.e == e && g.tau == tau && g.ws == ws) slot = &g;
    if (!slot) {
        SytrdGraph* victim = &gc.slots[0];
        for (auto& g : gc.slots)
            if (!g.exec) {
                victim = &g;
                break;
            } else if (g.last_use < victim->last_use) {
                victim = &g;
            }
        if (victim->exec) hipGraphExecDestroy(victim->exec);
        victim->exec = nullptr;
        hipGraph_t graph = nullptr;
        bool ok = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess;
        if (ok) {
            launch_sytrd_direct(s, n64, A, ld, d, e, tau, ws);
            ok = hipStreamEndCapture(s, &graph) == hipSuccess && graph;
        }
        if (ok) ok = hipGraphInstantiate(&victim->exec, graph, nullptr, nullptr, 0) == hipSuccess;
        if (graph) hipGraphDestroy(graph);
        if (!ok) {  // capture not possible here (e.g. the stream is already being captured): plain launches
            victim->exec = nullptr;
            (void)hipGetLastError();
            launch_sytrd_direct(s, n64, A, ld, d, e, tau, ws);
            return;
        }
        victim->n = n64;
        victim->ld = ld;
        victim->A = A;
        victim->d = d;
        victim->e = e;
        victim->tau = tau;
        victim->ws = ws;
        slot = victim;
    }
    slot->last_use = gc.clock;
    if (hipGraphLaunch(slot->exec, s) != hipSuccess) {
        (void)hipGetLastError();
        launch_sytrd_direct(s, n64, A, ld, d, e, tau, ws);
    }
}

}  // namespace sdpsr

// ---------------------------------------------------------------------------
// Small symmetric eigenproblems (n <= 128) in ONE workgroup: parallel cyclic Jacobi with the
// matrix resident in LDS.  The compressed problems of the module-compression driver (w x w,
// w < 2 dim P) and the reference's small test algebras are this size; the blocked
// tridiagonalisation path costs ~5 launches per column there, i.e. pure launch latency.
// Round-robin ordering: n/2 disjoint rotations per step, n-1 steps per sweep; every step is
// (angles) -> barrier -> (two-sided 2 x 2 block updates of A, column rotations of V) -> barrier.
// ---------------------------------------------------------------------------
namespace sdpsr {

constexpr int JAC_MAXTHREADS = 1024;
constexpr int JAC_MAXN = 128;

__global__ void __launch_bounds__(JAC_MAXTHREADS)
small_syev_jacobi_kernel(int n, double* __restrict__ Ag, int64_t lda, double* __restrict__ wout,
                         double* __restrict__ Vglob, int* __restrict__ info, int v_in_lds) {
    extern __shared__ __attribute__((aligned(16))) double sA[];  // n x n, leading dimension ldl
    __shared__ double s_c[JAC_MAXN / 2], s_s[JAC_MAXN / 2];
    __shared__ int s_p[JAC_MAXN / 2], s_q[JAC_MAXN / 2];
    __shared__ double s_red[JAC_MAXTHREADS / 64];
    __shared__ double s_off, s_diag;
    __shared__ int s_rank[JAC_MAXN];
    const int tid = threadIdx.x;
    const int JAC_THREADS = blockDim.x;
    const int ldl = n | 1;  // odd leading dimension: column walks hit distinct banks
    const int m = (n + 1) & ~1;  // players of the tournament (a dummy one when n is odd)
    const int half = m >> 1;
    // the eigenvector accumulator lives in LDS next to A when both fit (n <= 96), else in global
    double* __restrict__ Vtmp = v_in_lds ? (sA + (size_t)ldl * n + 8) : Vglob;
    const int ldv = v_in_lds ? ldl : n;
    // load A (symmetric part from the lower triangle, like LAPACK with uplo = 'L') and V = I
    for (int e = tid; e < n * n; e += JAC_THREADS) {
        const int j = e / n, i = e - j * n;
        const int ii = i > j ? i : j, jj = i > j ? j : i;
        sA[i + j * ldl] = Ag[ii + (int64_t)jj * lda];
        Vtmp[i + j * ldv] = (i == j) ? 1.0 : 0.0;
    }
    __syncthreads();
    int sweep = 0;
    for (; sweep < 40; ++sweep) {
        // off-diagonal and diagonal norms
        double off = 0, dg = 0;
        for (int e = tid; e < n * n; e += JAC_THREADS) {
            const int j = e / n, i = e - j * n;
            const double v = sA[i + j * ldl];
            if (i == j) dg = fma(v, v, dg);
            else off = fma(v, v, off);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            off += __shfl_down(off, o, 64);
            dg += __shfl_down(dg, o, 64);
        }
        if ((tid & 63) == 0) s_red[tid >> 6] = off;
        __syncthreads();
        if (tid == 0) {
            double t = 0;
            for (int k = 0; k < JAC_THREADS / 64; ++k) t += s_red[k];
            s_off = t;
        }
        __syncthreads();
        if ((tid & 63) == 0) s_red[tid >> 6] = dg;
        __syncthreads();
        if (tid == 0) {
            double t = 0;
            for (int k = 0; k < JAC_THREADS / 64; ++k) t += s_red[k];
            s_diag = t;
        }
        __syncthreads();
        // stop at the backward-error level of a LAPACK solver: ||off(A)||_F <= n eps ||A||_F (the
        // rounding floor of the sweeps themselves is ~ sqrt(n) eps, a tighter bound only spins)
        const double tolr = (double)n * 2.220446049250313e-16;
        if (s_off <= tolr * tolr * (s_diag + s_off) || s_off == 0.0) break;
        for (int step = 0; step < m - 1; ++step) {
            if (tid < half) {
                int p, q;
                if (tid == 0) {
                    p = m - 1;
                    q = step;
                } else {
                    p = (step + tid) % (m - 1);
                    q = (step - tid + (m - 1)) % (m - 1);
                }
                if (p > q) {
                    const int t = p;
                    p = q;
                    q = t;
                }
                double c = 1.0, s = 0.0;
                if (q < n) {
                    const double apq = sA[p + q * ldl];
                    if (apq != 0.0) {
                        // t = tan of the rotation angle (smaller root), c = 1/sqrt(1+t^2), s = t c.
                        // Hardware reciprocal / reciprocal-square-root seeds plus Newton steps in
                        // FMAs instead of the IEEE division and sqrt sequences (which dominated the
                        // step): only c needs full precision (c^2 + s^2 = 1 keeps V orthogonal), an
                        // error in t merely leaves a residual a_pq for the next sweep.
                        const double dd = sA[q + q * ldl] - sA[p + p * ldl], bb = 2.0 * apq;
                        const double h2 = fma(dd, dd, bb * bb);
                        if (!(h2 > 1e-280 && h2 < 1e280)) {  // out of the seeds' range: IEEE sequences
                            const double theta = dd / bb;
                            const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                            c = 1.0 / sqrt(t * t + 1.0);
                            s = t * c;
                        } else {
                        double y = __builtin_amdgcn_rsq(h2);
                        y = y * fma(-0.5 * h2 * y, y, 1.5);
                        const double r = h2 * y;  // hypot(dd, bb)
                        const double den = fabs(dd) + r;
                        double ri = __builtin_amdgcn_rcp(den);
                        ri = ri * fma(-den, ri, 2.0);
                        ri = ri * fma(-den, ri, 2.0);
                        const double t = (dd >= 0 ? bb : -bb) * ri;
                        const double u = fma(t, t, 1.0);
                        double z = __builtin_amdgcn_rsq(u);
                        z = z * fma(-0.5 * u * z, z, 1.5);
                        z = z * fma(-0.5 * u * z, z, 1.5);
                        c = z;
                        s = t * c;
                        }
                    }
                }
                s_c[tid] = c;
                s_s[tid] = s;
                s_p[tid] = p;
                s_q[tid] = (q < n) ? q : -1;
            }
            __syncthreads();
            // A <- J' A J on the 2 x 2 blocks (row pair k1, column pair k2): one thread owns a block,
            // so the two-sided update needs no barrier in between; V <- V J alongside
            for (int e = tid; e < half * half; e += JAC_THREADS) {
                const int k1 = e / half, k2 = e - k1 * half;
                const int r0 = s_p[k1], r1 = s_q[k1], c0 = s_p[k2], c1 = s_q[k2];
                const double cr = s_c[k1], sr = s_s[k1], cc = s_c[k2], sc = s_s[k2];
                if (sr == 0.0 && sc == 0.0) continue;
                const bool vr = r1 >= 0, vc = c1 >= 0;
                const double x00 = sA[r0 + c0 * ldl];
                const double x01 = vc ? sA[r0 + c1 * ldl] : 0.0;
                const double x10 = vr ? sA[r1 + c0 * ldl] : 0.0;
                const double x11 = (vr && vc) ? sA[r1 + c1 * ldl] : 0.0;
                const double y00 = cr * x00 - sr * x10, y10 = sr * x00 + cr * x10;
                const double y01 = cr * x01 - sr * x11, y11 = sr * x01 + cr * x11;
                sA[r0 + c0 * ldl] = cc * y00 - sc * y01;
                if (vc) sA[r0 + c1 * ldl] = sc * y00 + cc * y01;
                if (vr) sA[r1 + c0 * ldl] = cc * y10 - sc * y11;
                if (vr && vc) sA[r1 + c1 * ldl] = sc * y10 + cc * y11;
            }
            for (int e = tid; e < half * n; e += JAC_THREADS) {
                const int k = e / n, i = e - k * n;
                const int q = s_q[k];
                if (q < 0) continue;
                const int p = s_p[k];
                const double c = s_c[k], s = s_s[k];
                if (s == 0.0) continue;
                const double vx = Vtmp[i + p * ldv], vy = Vtmp[i + q * ldv];
                Vtmp[i + p * ldv] = c * vx - s * vy;
                Vtmp[i + q * ldv] = s * vx + c * vy;
            }
            __syncthreads();
        }
    }
    // ascending order: rank of every diagonal entry (ties broken by index)
    for (int i = tid; i < n; i += JAC_THREADS) {
        const double li = sA[i + i * ldl];
        int rk = 0;
        for (int j = 0; j < n; ++j) {
            const double lj = sA[j + j * ldl];
            rk += (lj < li) || (lj == li && j < i);
        }
        s_rank[i] = rk;
        wout[rk] = li;
    }
    __syncthreads();
    for (int e = tid; e < n * n; e += JAC_THREADS) {
        const int j = e / n, i = e - j * n;
        Ag[i + (int64_t)s_rank[j] * lda] = Vtmp[i + j * ldv];
    }
    if (tid == 0) {
        info[0] = (sweep >= 40) ? 1 : 0;
        info[1] = sweep;
    }
}

// The matrix is padded to an even order m with a zero row/column (the dummy player of the
// tournament: a_pq = 0 gives the identity rotation), so the step has no validity branches; the
// pairs of all m-1 rounds sit in an LDS table.
//
// jacobi64_fill_pairs / jacobi64_sweeps are the workgroup-level core, shared by the single-problem
// kernel below and by the batched eigen_decomposition kernel (kernels_batched.hip includes this
// file's declarations through jacobi64.h).
__global__ void __launch_bounds__(JAC_MAXTHREADS)
small_syev_jacobi64_kernel(int n, double* __restrict__ Ag, int64_t lda, double* __restrict__ wout,
                           int* __restrict__ info) {
    extern __shared__ __attribute__((aligned(16))) double sA[];  // A (m x m, ld ldl), V (same), pair table
    __shared__ double s_red[JAC_MAXTHREADS / 64 + 2];
    __shared__ int s_rank[64];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int m = (n + 1) & ~1, half = m >> 1;
    const int ldl = m | 1;
    double* __restrict__ sV = sA + (size_t)ldl * m;
    int* __restrict__ s_pq = reinterpret_cast<int*>(sV + (size_t)ldl * m);  // [(m-1) * half]: p | q << 16, p < q
    for (int e = tid; e < m * m; e += nthr) {
        const int j = e / m, i = e - j * m;
        const int ii = i > j ? i : j, jj = i > j ? j : i;
        sA[i + j * ldl] = (ii < n) ? Ag[ii + (int64_t)jj * lda] : 0.0;
        sV[i + j * ldl] = (i == j) ? 1.0 : 0.0;
    }
    jacobi64_fill_pairs(m, s_pq);
    __syncthreads();
    const int sweep = jacobi64_sweeps(n, m, ldl, sA, sV, s_pq, s_red);
    for (int i = tid; i < n; i += nthr) {
        const double li = sA[i + i * ldl];
        int rk = 0;
        for (int j = 0; j < n; ++j) {
            const double lj = sA[j + j * ldl];
            rk += (lj < li) || (lj == li && j < i);
        }
        s_rank[i] = rk;
        wout[rk] = li;
    }
    __syncthreads();
    for (int e = tid; e < n * n; e += nthr) {
        const int j = e / n, i = e - j * n;
        Ag[i + (int64_t)s_rank[j] * lda] = sV[i + j * ldl];
    }
    if (tid == 0) {
        info[0] = (sweep >= 40) ? 1 : 0;
        info[1] = sweep;
    }
}

void small_syev_set_device_attributes() {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&small_syev_jacobi64_kernel),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&small_syev_jacobi_kernel),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
}

bool launch_small_syev(hipStream_t s, int64_t n, double* A, int64_t lda, double* w, double* Vtmp, int* info) {
    if (n < 1 || n > JAC_MAXN) return false;
    size_t lds = (size_t)((n | 1) * n + 8) * sizeof(double);
    if (n <= 64) {
        const int half = (int)((n + 1) / 2), mm = 2 * half;
        int threads = (half * half + 63) / 64 * 64;
        if (threads < 64) threads = 64;
        const size_t lds64 = 2 * (size_t)(mm | 1) * mm * sizeof(double) + (size_t)(mm - 1) * half * sizeof(int) + 64;
        small_syev_jacobi64_kernel<<<1, threads, lds64, s>>>((int)n, A, lda, w, info);
        return true;
    }
    const int v_in_lds = (2 * lds <= 150 * 1024) ? 1 : 0;
    if (v_in_lds) lds *= 2;
    // one thread per element of the n/2 rotated column pairs, whole waves, at most 1024
    int threads = (int)(((n + 1) / 2) * n + 63) / 64 * 64;
    if (threads > JAC_MAXTHREADS) threads = JAC_MAXTHREADS;
    if (threads < 64) threads = 64;
    small_syev_jacobi_kernel<<<1, threads, lds, s>>>((int)n, A, lda, w, Vtmp, info, v_in_lds);
    return true;
}

}  // namespace sdpsr
