// Workgroup-level parallel cyclic Jacobi for symmetric matrices of order n <= 64 resident in LDS
// (eigen(A) of src/eigen_decomposition.jl:246 for the small algebras and for the compressed
// problems of the module-compression driver).  Device code only; included by kernels_sytrd.hip
// (one problem per launch) and kernels_batched.hip (one problem per workgroup, all CUs).
//
// One thread per 2 x 2 block (row pair k1, column pair k2) of a tournament step.  Every thread
// derives the two rotation angles it needs from the diagonal blocks itself (redundantly, in
// parallel), so a step is  read -> barrier -> rotate + write -> barrier  with no separate angle
// phase; the thread also rotates two rows of V for its row pair's columns.
#pragma once
#include <hip/hip_runtime.h>

namespace sdpsr {

__device__ __forceinline__ void jacobi_angle(double app, double aqq, double apq, double& c, double& s) {
    // t = tan of the rotation angle (smaller root), c = 1/sqrt(1+t^2), s = t c.  Hardware
    // reciprocal / reciprocal-square-root seeds instead of the IEEE division and sqrt sequences;
    // t comes from the raw seeds (it only has to make the rotated a_pq small against a_pq), c gets
    // two Newton steps (c^2 + s^2 = 1 keeps V orthogonal).  The dependent chain of this function
    // is what a tournament step costs.
    const double dd = aqq - app, bb = 2.0 * apq;
    const double h2 = fma(dd, dd, bb * bb);
    if (h2 > 1e-280 && h2 < 1e280) {
        const double y = __builtin_amdgcn_rsq(h2);
        const double den = fabs(dd) + h2 * y;  // |dd| + hypot(dd, bb)
        const double t = (dd >= 0 ? bb : -bb) * __builtin_amdgcn_rcp(den);
        const double u = fma(t, t, 1.0);
        double z = __builtin_amdgcn_rsq(u);
        z = z * fma(-0.5 * u * z, z, 1.5);
        z = z * fma(-0.5 * u * z, z, 1.5);
        c = z;
        s = t * z;
    } else if (apq != 0.0) {  // out of the seeds' range: IEEE sequences
        const double theta = dd / bb;
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        c = 1.0 / sqrt(t * t + 1.0);
        s = t * c;
    } else {
        c = 1.0;
        s = 0.0;
    }
}

// round-robin tournament of m players (m even, player m-1 fixed): s_pq[(m-1) * m/2] = p | q << 16
__device__ __forceinline__ void jacobi64_fill_pairs(int m, int* __restrict__ s_pq) {
    const int half = m >> 1;
    for (int e = threadIdx.x; e < (m - 1) * half; e += blockDim.x) {
        const int step = e / half, kk = e - step * half;
        int p, q;
        if (kk == 0) {
            p = m - 1;
            q = step;
        } else {
            p = (step + kk) % (m - 1);
            q = (step - kk + (m - 1)) % (m - 1);
        }
        if (p > q) {
            const int t = p;
            p = q;
            q = t;
        }
        s_pq[e] = p | (q << 16);
    }
}

// Sweeps until ||off(A)||_F <= n eps ||A||_F (the backward-error level of a LAPACK solver) or 40
// sweeps.  sA: m x m symmetric (leading dimension ldl, odd), sV: accumulates the rotations
// (identity on entry), s_red: blockDim/64 + 2 doubles of scratch.  All threads of the workgroup
// must call it (blockDim >= (m/2)^2); returns the number of sweeps done (40 = not converged).
// On return the eigenvalues are the diagonal of sA (unsorted), the eigenvectors the columns of sV.
__device__ __forceinline__ int jacobi64_sweeps(int n, int m, int ldl, double* __restrict__ sA, double* __restrict__ sV,
                                               const int* __restrict__ s_pq, double* __restrict__ s_red) {
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int half = m >> 1, nw = nthr >> 6;
    const bool active = tid < half * half;
    const int k1 = active ? tid / half : 0, k2 = active ? tid - k1 * half : 0;
    const int i0 = 2 * k2;  // rows of V this thread rotates (columns of pair k1)
    int sweep = 0;
    for (; sweep < 40; ++sweep) {
        double off = 0, dg = 0;
        for (int e = tid; e < m * m; e += nthr) {
            const int j = e / m, i = e - j * m;
            const double v = sA[i + j * ldl];
            if (i == j) dg = fma(v, v, dg);
            else off = fma(v, v, off);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            off += __shfl_down(off, o, 64);
            dg += __shfl_down(dg, o, 64);
        }
        __syncthreads();  // s_red of the previous round has been read by everyone
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
            double x00 = 0, x01 = 0, x10 = 0, x11 = 0, va0 = 0, va1 = 0, vb0 = 0, vb1 = 0;
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
                const double app = sA[r0 + r0 * ldl], aqq = sA[r1 + r1 * ldl], apq = sA[r0 + r1 * ldl];
                const double bpp = sA[c0 + c0 * ldl], bqq = sA[c1 + c1 * ldl], bpq = sA[c0 + c1 * ldl];
                x00 = sA[a00];
                x01 = sA[a01];
                x10 = sA[a10];
                x11 = sA[a11];
                va0 = sV[v0a];
                vb0 = sV[v0b];
                va1 = sV[v0a + 1];
                vb1 = sV[v0b + 1];
                jacobi_angle(app, aqq, apq, cr, sr);
                jacobi_angle(bpp, bqq, bpq, cc, sc);
            }
            __syncthreads();  // every read of this step is done before any write
            if (active) {
                const double y00 = cr * x00 - sr * x10, y10 = sr * x00 + cr * x10;
                const double y01 = cr * x01 - sr * x11, y11 = sr * x01 + cr * x11;
                sA[a00] = cc * y00 - sc * y01;
                sA[a01] = sc * y00 + cc * y01;
                sA[a10] = cc * y10 - sc * y11;
                sA[a11] = sc * y10 + cc * y11;
                sV[v0a] = cr * va0 - sr * vb0;
                sV[v0b] = sr * va0 + cr * vb0;
                sV[v0a + 1] = cr * va1 - sr * vb1;
                sV[v0b + 1] = sr * va1 + cr * vb1;
            }
            __syncthreads();
        }
    }
    return sweep;
}

// pair kk of round `step` of the round-robin tournament (the same schedule as jacobi64_fill_pairs,
// from registers: no table load in front of the matrix reads)
__device__ __forceinline__ void jacobi64_pair(int m1, int step, int kk, int& p, int& q) {
    int a = step + kk, b = step - kk;
    if (a >= m1) a -= m1;
    if (b < 0) b += m1;
    if (kk == 0) b = m1;
    p = a < b ? a : b;
    q = a < b ? b : a;
}

// Ping-pong variant of jacobi64_sweeps: a tournament step reads (cA, cV) and writes (aA, aV) --
// every entry of A and of V belongs to exactly one thread per step -- so a step costs ONE barrier
// instead of two, and nothing but register arithmetic sits in front of its LDS reads.  sA2/sV2:
// two more m x m buffers (contents irrelevant); s_red: 2 * blockDim/64 doubles.  Results land in sA / sV like jacobi64_sweeps.
__device__ __forceinline__ int jacobi64_sweeps_pp(int n, int m, int ldl, double* sA, double* sV, double* sA2, double* sV2,
                                                  double* s_red) {
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int half = m >> 1, nw = nthr >> 6, m1 = m - 1;
    const bool active = tid < half * half;
    const int k1 = active ? tid / half : 0, k2 = active ? tid - k1 * half : 0;
    const int i0 = 2 * k2;  // rows of V this thread rotates (columns of pair k1)
    double *cA = sA, *cV = sV, *aA = sA2, *aV = sV2;
    int sweep = 0;
    for (; sweep < 40; ++sweep) {
        double off = 0, dg = 0;
        for (int e = tid; e < m * m; e += nthr) {
            const int j = e / m, i = e - j * m;
            const double v = cA[i + j * ldl];
            if (i == j) dg = fma(v, v, dg);
            else off = fma(v, v, off);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            off += __shfl_down(off, o, 64);
            dg += __shfl_down(dg, o, 64);
        }
        if ((tid & 63) == 0) {
            s_red[tid >> 6] = off;
            s_red[nw + (tid >> 6)] = dg;
        }
        __syncthreads();
        double s_off = 0, s_diag = 0;
        for (int k = 0; k < nw; ++k) {
            s_off += s_red[k];
            s_diag += s_red[nw + k];
        }
        __syncthreads();  // s_red is free again
        const double tolr = (double)n * 2.220446049250313e-16;
        if (s_off <= tolr * tolr * (s_diag + s_off) || s_off == 0.0) break;
        for (int step = 0; step < m1; ++step) {
            if (active) {
                int r0, r1, c0, c1;
                jacobi64_pair(m1, step, k1, r0, r1);
                jacobi64_pair(m1, step, k2, c0, c1);
                const int a00 = r0 + c0 * ldl, a01 = r0 + c1 * ldl, a10 = r1 + c0 * ldl, a11 = r1 + c1 * ldl;
                const int v0a = i0 + r0 * ldl, v0b = i0 + r1 * ldl;
                const double app = cA[r0 + r0 * ldl], aqq = cA[r1 + r1 * ldl], apq = cA[r0 + r1 * ldl];
                const double bpp = cA[c0 + c0 * ldl], bqq = cA[c1 + c1 * ldl], bpq = cA[c0 + c1 * ldl];
                const double x00 = cA[a00], x01 = cA[a01], x10 = cA[a10], x11 = cA[a11];
                const double va0 = cV[v0a], vb0 = cV[v0b], va1 = cV[v0a + 1], vb1 = cV[v0b + 1];
                double cr, sr, cc, sc;
                jacobi_angle(app, aqq, apq, cr, sr);
                jacobi_angle(bpp, bqq, bpq, cc, sc);
                const double y00 = cr * x00 - sr * x10, y10 = sr * x00 + cr * x10;
                const double y01 = cr * x01 - sr * x11, y11 = sr * x01 + cr * x11;
                aA[a00] = cc * y00 - sc * y01;
                aA[a01] = sc * y00 + cc * y01;
                aA[a10] = cc * y10 - sc * y11;
                aA[a11] = sc * y10 + cc * y11;
                aV[v0a] = cr * va0 - sr * vb0;
                aV[v0b] = sr * va0 + cr * vb0;
                aV[v0a + 1] = cr * va1 - sr * vb1;
                aV[v0b + 1] = sr * va1 + cr * vb1;
            }
            __syncthreads();
            double* t = cA;
            cA = aA;
            aA = t;
            t = cV;
            cV = aV;
            aV = t;
        }
    }
    if (cA != sA) {  // uniform
        for (int e = tid; e < m * m; e += nthr) {
            const int j = e / m, i = e - j * m;
            sA[i + j * ldl] = cA[i + j * ldl];
            sV[i + j * ldl] = cV[i + j * ldl];
        }
    }
    __syncthreads();
    return sweep;
}

}  // namespace sdpsr
