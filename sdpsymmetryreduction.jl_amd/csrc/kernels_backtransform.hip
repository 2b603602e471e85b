// Back-transformation of the symmetric eigensolver: eigenvectors of A = Q T Q' are Q Z with Z the
// eigenvectors of the tridiagonal T (third phase of eigen(A), src/eigen_decomposition.jl:246;
// LAPACK dormtr).  Q = H_0 H_1 ... H_{n-2} is applied in blocks of 128 reflectors in compact-WY
// form, Q_b = I - V_b T_b V_b', last block first, and every O(n^3) step is a product on the fp64
// matrix cores (gemm_tn_dma_kernel<f64>, kernels_gemm.hip):
//     G   = V_b' V_b                  Gram matrix of the panel            (split-K)
//     T_b = larft(G, tau)             one workgroup, LDS
//     X   = (V_b T_b)' as rows        128 x m, one small product
//     W   = V_b' Z                    128 x n                             (split-K)
//     Z  -= X' W                      m x n, K = 128, read-modify-write epilogue
// The kernels here are the glue: panel extraction in the two operand layouts and the T factor.
#include "sdpsr_internal.h"

namespace sdpsr {

constexpr int BT_KB = 128;  // reflectors per block

// Vp (ld x 128 column-major) and VpT (ld x 128 row-major) <- block of reflectors j0 .. j0+127
// stored LAPACK-style below the subdiagonal of A: v_j = [0 .. 0, 1 (row j+1), A[j+2.., j]].
// Reflectors past n-2 do not exist: zero columns.  Rows r0 .. ld-1 are written (r0 multiple of 128).
__global__ void __launch_bounds__(256)
bt_extract_panel_kernel(int n, int64_t ld, const double* __restrict__ A, int j0, int r0, double* __restrict__ Vp,
                        double* __restrict__ VpT) {
    __shared__ double tile[64][65];
    // tile of 64 rows x 64 panel columns
    const int rb = r0 + blockIdx.x * 64, cb = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int cc = ty; cc < 64; cc += 4) {
        const int r = rb + tx, c = cb + cc, j = j0 + c;
        double v = 0.0;
        if (r < n && j <= n - 2) {
            if (r == j + 1) v = 1.0;
            else if (r > j + 1) v = A[r + (int64_t)j * ld];
        }
        Vp[r + (int64_t)c * ld] = v;
        tile[cc][tx] = v;
    }
    __syncthreads();
    for (int rr = ty; rr < 64; rr += 4) VpT[(int64_t)(rb + rr) * BT_KB + cb + tx] = tile[tx][rr];
}

// T (128 x 128, column-major, upper triangular) from the Gram matrix G = V'V and tau (dlarft,
// forward / columnwise):  T[j,j] = tau_j,  T[0:j, j] = -tau_j * T[0:j, 0:j] * G[0:j, j].
// One workgroup per BLOCK of reflectors (blockIdx.x = b: reflectors 128 b ..): T_b depends on its own panel only, so
// the T factors of all blocks are formed in one launch before the blocks are applied one after the other (the
// recurrence is a 167-us dependent chain: 32 of them in sequence were 5.3 ms of the back-transformation at N = 4096).
// T lives in LDS (thread i owns row i, so the recurrence needs no barriers), G passes through LDS 16 columns at a
// time, prefetched into registers a chunk ahead.
__global__ void __launch_bounds__(128)
bt_larft_kernel(const double* __restrict__ Gall, const double* __restrict__ tau, int n, double* __restrict__ Tall) {
    const int j0 = BT_KB * (int)blockIdx.x;
    const double* __restrict__ G = Gall + (int64_t)blockIdx.x * BT_KB * BT_KB;
    double* __restrict__ T = Tall + (int64_t)blockIdx.x * BT_KB * BT_KB;
    extern __shared__ double sm[];
    double* sT = sm;                    // 128 x 128, ld 129 (row i read by thread i: conflict-free)
    double* sg = sT + 128 * 129;        // 16 columns of G at a time: sg[c * 128 + i]
    __shared__ double s_tau[128];
    const int i = threadIdx.x;
    for (int e = i; e < 128 * 129; e += 128) sT[e] = 0.0;
    {
        const int j = j0 + i;
        s_tau[i] = (j <= n - 2) ? tau[j] : 0.0;
    }
    // G is read 16 columns ahead into registers (16 independent loads per thread), so the
    // sequential recurrence never waits on global memory
    double gnext[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) gnext[c] = G[i + (int64_t)c * 128];
    for (int jb = 0; jb < 128; jb += 16) {
        __syncthreads();  // the previous chunk of sg has been consumed
#pragma unroll
        for (int c = 0; c < 16; ++c) sg[c * 128 + i] = gnext[c];
        if (jb + 16 < 128) {
#pragma unroll
            for (int c = 0; c < 16; ++c) gnext[c] = G[i + (int64_t)(jb + 16 + c) * 128];
        }
        __syncthreads();
        for (int jc = 0; jc < 16; ++jc) {
            const int j = jb + jc;
            const double tj = s_tau[j];
            const double* gcol = sg + jc * 128;
            if (i < j) {
                // (T[0:j,0:j] * g)[i] = sum_{k >= i} T[i,k] g[k]  (T upper triangular); two chains
                // eight independent LDS load pairs in flight per trip (two waves cannot hide the LDS
                // latency of a one-load-at-a-time chain), four accumulation chains
                double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
                const double* trow = sT + i * 129;
                int k = i;
                for (; k + 7 < j; k += 8) {
                    double tv[8], gv[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        tv[u] = trow[k + u];
                        gv[u] = gcol[k + u];
                    }
                    a0 = fma(tv[0], gv[0], a0);
                    a1 = fma(tv[1], gv[1], a1);
                    a2 = fma(tv[2], gv[2], a2);
                    a3 = fma(tv[3], gv[3], a3);
                    a0 = fma(tv[4], gv[4], a0);
                    a1 = fma(tv[5], gv[5], a1);
                    a2 = fma(tv[6], gv[6], a2);
                    a3 = fma(tv[7], gv[7], a3);
                }
                for (; k < j; ++k) a0 = fma(trow[k], gcol[k], a0);
                sT[i * 129 + j] = -tj * ((a0 + a1) + (a2 + a3));  // column j of T is only read by later steps
            }
            if (i == j) sT[i * 129 + j] = tj;
            // no barrier: thread i reads and writes row i of T only; the shared column of G is read-only
        }
    }
    for (int e = i; e < 128 * 128; e += 128) {
        const int c = e >> 7, r = e & 127;
        T[r + (int64_t)c * 128] = sT[r * 129 + c];
    }
}

bool backtransform_set_device_attributes() {
    bool ok = true;
    ok &= hipSuccess == hipFuncSetAttribute(reinterpret_cast<const void*>(&bt_larft_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                        (128 * 129 + 16 * 128) * 8);
    return ok;
}

void launch_bt_extract_panel(hipStream_t s, int64_t n, int64_t ld, const double* A, int64_t j0, int64_t r0, double* Vp,
                             double* VpT) {
    dim3 grid((unsigned)((ld - r0) / 64), 2);
    bt_extract_panel_kernel<<<grid, 256, 0, s>>>((int)n, ld, A, (int)j0, (int)r0, Vp, VpT);
}
// G, T: nblk consecutive 128 x 128 matrices (block b at offset b * 128 * 128)
void launch_bt_larft(hipStream_t s, const double* G, const double* tau, int64_t nblk, int64_t n, double* T) {
    bt_larft_kernel<<<(unsigned)nblk, 128, (128 * 129 + 16 * 128) * 8, s>>>(G, tau, (int)n, T);
}

}  // namespace sdpsr
