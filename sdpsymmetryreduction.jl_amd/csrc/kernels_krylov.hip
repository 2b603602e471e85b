// Kernels of the Krylov driver of blockDiagonalize (DESIGN.md "Krylov driver"): the generic
// element of a symmetric algebra has only k = sum_k s_k distinct eigenvalues (k <= dim(P)), so a
// Lanczos process with full re-orthogonalisation started from a random vector breaks down after
// exactly k steps and its Ritz pairs are the k eigenvalues and one generic unit vector of every
// eigenspace -- everything src/eigen_decomposition.jl:236-348 uses of eigen(A).  All passes
// are HBM-bound reads of the n x n element (sym_gemv) or small tall-skinny updates.
#include <algorithm>
#include "sdpsr_internal.h"

namespace sdpsr {

// y = A x for symmetric A (column dots: one wave per column, 16-byte loads).  ld even.
__global__ void __launch_bounds__(256)
sym_gemv_kernel(int n, int64_t ld, const double* __restrict__ A, const double* __restrict__ x,
                double* __restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) double s_x[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n2 = (n + 1) >> 1;
    for (int t = tid; t < 2 * n2; t += 256) s_x[t] = (t < n) ? x[t] : 0.0;
    __syncthreads();
    const double2* sx2 = reinterpret_cast<const double2*>(s_x);
    const int nwaves = gridDim.x * 4;
    for (int k = blockIdx.x * 4 + wave; k < n; k += nwaves) {
        const double2* c2 = reinterpret_cast<const double2*>(A + (int64_t)k * ld);
        double a0 = 0, a1 = 0;
        int t = lane;
        for (; t + 64 < n2; t += 128) {
            const double2 u0 = c2[t], u1 = c2[t + 64];
            const double2 v0 = sx2[t], v1 = sx2[t + 64];
            a0 = fma(u0.x, v0.x, a0);
            a0 = fma(u0.y, v0.y, a0);
            a1 = fma(u1.x, v1.x, a1);
            a1 = fma(u1.y, v1.y, a1);
        }
        for (; t < n2; t += 64) {
            const double2 u0 = c2[t];
            const double2 v0 = sx2[t];
            a0 = fma(u0.x, v0.x, a0);
            a0 = fma(u0.y, v0.y, a0);
        }
        double acc = a0 + a1;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
        if (lane == 0) y[k] = acc;
    }
}
void launch_sym_gemv(hipStream_t s, int64_t n, int64_t ld, const double* A, const double* x, double* y) {
    int nblk = (int)((n + 7) / 8);
    if (nblk > 1024) nblk = 1024;
    if (nblk < 1) nblk = 1;
    const size_t lds = ((size_t)n + 4) * sizeof(double);
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&sym_gemv_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096);
        attr_set = true;
    }
    sym_gemv_kernel<<<nblk, 256, lds, s>>>((int)n, ld, A, x, y);
}

// ---------------------------------------------------------------------------
// One Lanczos step of run r = blockIdx.x (classical Gram-Schmidt, twice):
//   w = W[:, r]; c = H' w; w -= H c (over the t+1 stored vectors), repeated;
//   alpha = (c1 + c2)[t]; beta = ||w||; H[:, t+1] = w / beta.
// H_r = H + r * hstride (n x cap, leading dimension ld).  1024 threads; w lives in registers
// (n <= 16 * 1024).  active[r] == 0 -> nothing to do.
// ---------------------------------------------------------------------------
constexpr int LZ_THREADS = 1024;

__global__ void __launch_bounds__(LZ_THREADS)
lanczos_orth_kernel(int n, int64_t ld, double* __restrict__ H, int64_t hstride, const int* __restrict__ tcur,
                    const double* __restrict__ W, int64_t ldw, const int* __restrict__ active,
                    double* __restrict__ alpha_out, double* __restrict__ beta_out, double* __restrict__ nin_out) {
    const int r = blockIdx.x;
    if (active && !active[r]) return;
    extern __shared__ __attribute__((aligned(16))) double s_w[];  // n doubles: the vector being orthogonalised
    __shared__ double s_c[512];        // coefficients of the current pass (t + 1 <= 512)
    __shared__ double s_alpha;
    __shared__ double s_red[LZ_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t = tcur[r];
    double* Hr = H + (int64_t)r * hstride;
    double sq0 = 0;
    for (int i = tid; i < n; i += LZ_THREADS) {
        const double v = W[i + (int64_t)r * ldw];
        s_w[i] = v;
        sq0 = fma(v, v, sq0);
    }
    if (tid == 0) s_alpha = 0.0;
    if (nin_out) {  // norm of the incoming vector (rank decisions of the module-compression driver)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sq0 += __shfl_down(sq0, o, 64);
        if (lane == 0) s_red[wave] = sq0;
        __syncthreads();
        if (tid == 0) {
            double tot = 0;
            for (int q = 0; q < LZ_THREADS / 64; ++q) tot += s_red[q];
            nin_out[r] = sqrt(tot);
        }
    }
    __syncthreads();
    for (int pass = 0; pass < 2; ++pass) {
        // c[h] = H[:,h] . w : one wave per stored vector
        for (int h = wave; h <= t; h += LZ_THREADS / 64) {
            const double* hc = Hr + (int64_t)h * ld;
            double acc = 0;
            for (int i = lane; i < n; i += 64) acc = fma(hc[i], s_w[i], acc);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
            if (lane == 0) s_c[h] = acc;
        }
        __syncthreads();
        if (tid == 0) s_alpha += s_c[t];
        // w -= H c : one thread per row, all stored vectors
        for (int i = tid; i < n; i += LZ_THREADS) {
            double wv = s_w[i];
            for (int h = 0; h <= t; ++h) wv = fma(-s_c[h], Hr[(int64_t)h * ld + i], wv);
            s_w[i] = wv;
        }
        __syncthreads();
    }
    double sq = 0;
    for (int i = tid; i < n; i += LZ_THREADS) sq = fma(s_w[i], s_w[i], sq);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_down(sq, o, 64);
    if (lane == 0) s_red[wave] = sq;
    __syncthreads();
    if (tid == 0) {
        double tot = 0;
        for (int k = 0; k < LZ_THREADS / 64; ++k) tot += s_red[k];
        s_red[0] = sqrt(tot);
        alpha_out[r] = s_alpha;
        beta_out[r] = s_red[0];
    }
    __syncthreads();
    const double beta = s_red[0];
    const double inv = (beta > 0) ? 1.0 / beta : 0.0;
    double* hn = Hr + (int64_t)(t + 1) * ld;
    for (int i = tid; i < n; i += LZ_THREADS) hn[i] = s_w[i] * inv;
}
void launch_lanczos_orth(hipStream_t s, int64_t n, int64_t ld, double* H, int64_t hstride, const int* tcur,
                         double* W, int64_t ldw, const int* active, int nruns, double* alpha_out, double* beta_out,
                         double* nin_out) {
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&lanczos_orth_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 8192);
        attr_set = true;
    }
    lanczos_orth_kernel<<<nruns, LZ_THREADS, (size_t)n * sizeof(double), s>>>((int)n, ld, H, hstride, tcur, W, ldw, active,
                                                                            alpha_out, beta_out, nin_out);
}

// H_r[:, 0] = X[:, r] / ||X[:, r]||, norm0[r] = ||X[:, r]||   (one block per run)
__global__ void __launch_bounds__(LZ_THREADS)
lanczos_init_kernel(int n, int64_t ld, double* __restrict__ H, int64_t hstride, const double* __restrict__ X,
                    int64_t ldx, double* __restrict__ norm0) {
    __shared__ double s_red[LZ_THREADS / 64];
    const int r = blockIdx.x, tid = threadIdx.x;
    double sq = 0;
    for (int i = tid; i < n; i += LZ_THREADS) {
        const double v = X[i + (int64_t)r * ldx];
        sq = fma(v, v, sq);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_down(sq, o, 64);
    if ((tid & 63) == 0) s_red[tid >> 6] = sq;
    __syncthreads();
    if (tid == 0) {
        double tot = 0;
        for (int k = 0; k < LZ_THREADS / 64; ++k) tot += s_red[k];
        s_red[0] = sqrt(tot);
        norm0[r] = s_red[0];
    }
    __syncthreads();
    const double inv = s_red[0] > 0 ? 1.0 / s_red[0] : 0.0;
    double* h0 = H + (int64_t)r * hstride;
    for (int i = tid; i < n; i += LZ_THREADS) h0[i] = X[i + (int64_t)r * ldx] * inv;
}
void launch_lanczos_init(hipStream_t s, int64_t n, int64_t ld, double* H, int64_t hstride, const double* X,
                         int64_t ldx, int nruns, double* norm0) {
    (void)ld;
    lanczos_init_kernel<<<nruns, LZ_THREADS, 0, s>>>((int)n, ld, H, hstride, X, ldx, norm0);
}

// gather the current Lanczos vectors into a dense block: V[:, r] = H_r[:, tcur[r]] (0 if inactive)
__global__ void lanczos_pack_kernel(int n, int64_t ld, const double* __restrict__ H, int64_t hstride,
                                    const int* __restrict__ tcur, const int* __restrict__ active,
                                    double* __restrict__ V, int64_t ldv) {
    const int r = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    V[i + (int64_t)r * ldv] = active[r] ? H[(int64_t)r * hstride + (int64_t)tcur[r] * ld + i] : 0.0;
}
void launch_lanczos_pack(hipStream_t s, int64_t n, int64_t ld, const double* H, int64_t hstride, const int* tcur,
                         const int* active, int nruns, double* V, int64_t ldv) {
    dim3 g((unsigned)((n + 255) / 256), (unsigned)nruns);
    lanczos_pack_kernel<<<g, 256, 0, s>>>((int)n, ld, H, hstride, tcur, active, V, ldv);
}

// out[:, c] = sum_t Hsrc_c[:, t] * S[t + c * lds_],  t < kk[c];  Hsrc_c = H + run[c] * hstride
__global__ void ritz_combine_kernel(int n, int64_t ld, const double* __restrict__ H, int64_t hstride,
                                    const int* __restrict__ run, const int* __restrict__ kk,
                                    const double* __restrict__ S, int lds_, double* __restrict__ out, int64_t ldo) {
    const int c = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double* Hr = H + (int64_t)run[c] * hstride;
    const double* sc = S + (int64_t)c * lds_;
    double acc = 0;
    const int k = kk[c];
    for (int t = 0; t < k; ++t) acc = fma(Hr[(int64_t)t * ld + i], sc[t], acc);
    out[i + (int64_t)c * ldo] = acc;
}
void launch_ritz_combine(hipStream_t s, int64_t n, int64_t ld, const double* H, int64_t hstride, const int* run,
                         const int* kk, const double* S, int lds_, int ncols, double* out, int64_t ldo) {
    if (ncols <= 0) return;
    dim3 g((unsigned)((n + 255) / 256), (unsigned)ncols);
    ritz_combine_kernel<<<g, 256, 0, s>>>((int)n, ld, H, hstride, run, kk, S, lds_, out, ldo);
}

// deterministic pseudo-random start vector in (-1, 1)
__global__ void random_vector_kernel(int n, uint64_t key, double* __restrict__ x) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = 2.0 * sdpsr_class_uniform(key, (uint32_t)(i + 1)) - 1.0;
}
void launch_random_vector(hipStream_t s, int64_t n, uint64_t key, double* x) {
    random_vector_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>((int)n, key, x);
}

// B <- (B + B') / 2 on the leading m x m part (ld), in place
__global__ void symmetrize_kernel(int m, int64_t ld, double* __restrict__ B) {
    const int j = blockIdx.y * blockDim.y + threadIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m && j < m && i > j) {
        const double v = 0.5 * (B[i + (int64_t)j * ld] + B[j + (int64_t)i * ld]);
        B[i + (int64_t)j * ld] = v;
        B[j + (int64_t)i * ld] = v;
    }
}
void launch_symmetrize(hipStream_t s, int64_t m, int64_t ld, double* B) {
    dim3 b(32, 8), g((unsigned)((m + 31) / 32), (unsigned)((m + 7) / 8));
    symmetrize_kernel<<<g, b, 0, s>>>((int)m, ld, B);
}

// out[:, c] = beta * out[:, c] + alpha * sum_{t < kk} In[:, t] * S[t + c * lds]   (tall-skinny times small)
__global__ void tall_times_small_kernel(int n, int64_t ldi, const double* __restrict__ In, int kk,
                                        const double* __restrict__ S, int lds_, double alpha, double beta,
                                        double* __restrict__ out, int64_t ldo) {
    const int c = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double* sc = S + (int64_t)c * lds_;
    double acc = 0;
    for (int t = 0; t < kk; ++t) acc = fma(In[(int64_t)t * ldi + i], sc[t], acc);
    double* o = out + i + (int64_t)c * ldo;
    *o = (beta == 0.0 ? 0.0 : beta * *o) + alpha * acc;
}
void launch_tall_times_small(hipStream_t s, int64_t n, int64_t ldi, const double* In, int kk, const double* S,
                             int lds_, int ncols, double alpha, double beta, double* out, int64_t ldo) {
    if (ncols <= 0) return;
    dim3 g((unsigned)((n + 255) / 256), (unsigned)ncols);
    tall_times_small_kernel<<<g, 256, 0, s>>>((int)n, ldi, In, kk, S, lds_, alpha, beta, out, ldo);
}

// C[e] = sum_z P[z * stride + e]  (split-K partial sums, fixed order)
__global__ void splitk_reduce_kernel(int64_t len, int Z, int64_t stride, const double* __restrict__ P,
                                     double* __restrict__ C) {
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < len; e += step) {
        double acc = 0;
        for (int z = 0; z < Z; ++z) acc += P[(int64_t)z * stride + e];
        C[e] = acc;
    }
}
void launch_splitk_reduce(hipStream_t s, int64_t len, int Z, int64_t stride, const double* P, double* C) {
    int64_t g = (len + 255) / 256;
    if (g > 2048) g = 2048;
    splitk_reduce_kernel<<<(unsigned)g, 256, 0, s>>>(len, Z, stride, P, C);
}

// ---------------------------------------------------------------------------
// Fused randomize! + product:  Y = A W  with  A[r,c] = value(L[r,c])  never materialised
// (src/abstract_part.jl:107-110 fused into the products of the module-compression driver).
// Reads the 4-byte labels coalesced along r, the per-class values come from a d+1 entry table
// built in LDS from the counter-based generator.  Workgroup = 64 rows; grid.y splits the column
// range, partial sums are reduced in fixed order.
// ---------------------------------------------------------------------------
// W is read through the SCALAR data path.  (A first version staged W through LDS and broadcast it
// to the lanes: one 1 KiB LDS read per two FMAs per wave, bound by the LDS pipe at 128 B/clk/CU
// instead of the FP64 rate: 102 us at N = 4096, w = 34 against 57 us for this one.)  The four waves of a workgroup split the w output columns (JW = WMAX/4 each)
// and every wave walks ALL matrix columns c of its 64 rows: the column index is wave-uniform, so
// W'[wave][c][0..JW) (a transposed, zero-padded copy) arrives by s_load in SGPRs and is fed to
// the FMAs as a scalar operand; JW accumulators per lane keep the occupancy high, which hides the
// scalar-load latency.  Labels are read once per wave (4x through L1, coalesced).
__global__ void transpose_pad_w_kernel(int n, int npad, int w, int jw, const double* __restrict__ W, int64_t ldw,
                                       double* __restrict__ Wt) {
    const int64_t per_wave = (int64_t)npad * jw;
    const int64_t total = 4 * per_wave;
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += step) {
        const int wv = (int)(e / per_wave);
        const int64_t rem = e - (int64_t)wv * per_wave;
        const int c = (int)(rem / jw), j = (int)(rem - (int64_t)c * jw);
        const int jj = wv * jw + j;
        Wt[e] = (jj < w && c < n) ? W[c + (int64_t)jj * ldw] : 0.0;
    }
}

template <int WMAX>
__global__ void __launch_bounds__(256)
label_spmm_sload_kernel(int n, int npad, const uint32_t* __restrict__ L, uint64_t key, int d,
                        const double* __restrict__ Wt, int w, int cols_per_block, double* __restrict__ P) {
    constexpr int JW = WMAX / 4;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* sV = smem;  // [d + 1] class values
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = blockIdx.x * 64 + lane;
    const int c_begin = blockIdx.y * cols_per_block;
    int c_end = c_begin + cols_per_block;
    if (c_end > n) c_end = n;
    for (int i = tid; i <= d; i += 256) sV[i] = i ? sdpsr_class_uniform(key, (uint32_t)i) : 0.0;
    __syncthreads();
    double acc[JW];
#pragma unroll
    for (int j = 0; j < JW; ++j) acc[j] = 0.0;
    const double* __restrict__ Ww = Wt + (int64_t)wave * npad * JW;
    const bool row_ok = r < n;
#pragma unroll 1
    for (int cb = c_begin; cb < c_end; cb += 8) {
        uint32_t lab[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = cb + u;
            lab[u] = (row_ok && c < c_end) ? L[r + (int64_t)c * n] : 0u;  // label 0 -> value 0
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const double v = sV[lab[u]];
            const double* __restrict__ wr = Ww + (int64_t)(cb + u) * JW;  // uniform address (rows >= n are zero)
#pragma unroll
            for (int j = 0; j < JW; ++j) acc[j] = fma(v, wr[j], acc[j]);
        }
    }
    if (row_ok) {
        double* p = P + (int64_t)blockIdx.y * n * w;
#pragma unroll
        for (int j = 0; j < JW; ++j) {
            const int jj = wave * JW + j;
            if (jj < w) p[r + (int64_t)jj * n] = acc[j];
        }
    }
}

// Y[r + j*ldy] = sum_z P[z][r + j*n]
__global__ void label_spmm_reduce_kernel(int n, int w, int Z, const double* __restrict__ P,
                                         double* __restrict__ Y, int64_t ldy) {
    const int64_t total = (int64_t)n * w;
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += step) {
        const int64_t j = e / n, r = e - j * n;
        double acc = 0;
        for (int z = 0; z < Z; ++z) acc += P[(int64_t)z * total + e];
        Y[r + j * ldy] = acc;
    }
}

size_t label_spmm_partial_doubles(int64_t n, int w) {
    int zg = (int)(2048 / ((n + 63) / 64));
    if (zg < 1) zg = 1;
    if (zg > 32) zg = 32;
    (void)w;
    return (size_t)zg * n * 64 + (size_t)(n + 8) * 64;  // partial sums + transposed copy of W
}

// returns false when the shape is not supported (w > 64 or the class table does not fit in LDS)
bool launch_label_spmm(hipStream_t s, int64_t n, const uint32_t* L, uint64_t key, int64_t d, const double* W,
                       int64_t ldw, int w, double* partials, double* Y, int64_t ldy) {
    if (w < 1 || w > 64 || d > 4000) return false;
    const int rb = (int)((n + 63) / 64);
    int zg = 2048 / rb;
    if (zg < 1) zg = 1;
    if (zg > 32) zg = 32;
    const int zg_cap = zg;
    int cpb = (int)((n + zg - 1) / zg);
    cpb = (cpb + 127) / 128 * 128;
    zg = (int)((n + cpb - 1) / cpb);
    dim3 g((unsigned)rb, (unsigned)zg);
    // transposed zero-padded copy of W behind the partial sums (label_spmm_partial_doubles leaves room)
    double* Wt = partials + (size_t)zg_cap * n * 64;
    const int npad = (int)n + 8;
    auto go = [&](auto kern, int wmax) {
        transpose_pad_w_kernel<<<(unsigned)std::min<int64_t>(((int64_t)npad * wmax + 255) / 256, 2048), 256, 0, s>>>(
            (int)n, npad, w, wmax / 4, W, ldw, Wt);
        const size_t lds = ((size_t)d + 2) * sizeof(double);
        kern<<<g, 256, lds, s>>>((int)n, npad, L, key, (int)d, Wt, w, cpb, partials);
    };
    if (w <= 8) go(label_spmm_sload_kernel<8>, 8);
    else if (w <= 16) go(label_spmm_sload_kernel<16>, 16);
    else if (w <= 32) go(label_spmm_sload_kernel<32>, 32);
    else if (w <= 48) go(label_spmm_sload_kernel<48>, 48);
    else go(label_spmm_sload_kernel<64>, 64);
    int64_t gr = ((int64_t)n * w + 255) / 256;
    if (gr > 2048) gr = 2048;
    label_spmm_reduce_kernel<<<(unsigned)gr, 256, 0, s>>>((int)n, w, zg, partials, Y, ldy);
    return true;
}

}  // namespace sdpsr
