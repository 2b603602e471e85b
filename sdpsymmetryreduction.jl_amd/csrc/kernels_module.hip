// Kernels of the module-compression driver of blockDiagonalize (DESIGN.md "module compression"):
// Y = A(v) W straight from the labels (label_spmm), tall-times-small products, split-K
// reductions and the small glue around them; plus helpers shared with the setup stage.
#include <algorithm>
#include "sdpsr_internal.h"

namespace sdpsr {

constexpr int LZ_THREADS = 1024;

// H_r[:, 0] = X[:, r] / ||X[:, r]||, norm0[r] = ||X[:, r]||   (one block per run)
__global__ void __launch_bounds__(LZ_THREADS)
normalize_columns_kernel(int n, int64_t ld, double* __restrict__ H, int64_t hstride, const double* __restrict__ X,
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
void launch_normalize_columns(hipStream_t s, int64_t n, int64_t ld, double* H, int64_t hstride, const double* X,
                         int64_t ldx, int nruns, double* norm0) {
    (void)ld;
    normalize_columns_kernel<<<nruns, LZ_THREADS, 0, s>>>((int)n, ld, H, hstride, X, ldx, norm0);
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

// G generic elements A_g (keys k[g]) applied to the same W in ONE pass over the labels:
// Y[:, g w + j] = (A_g W)[:, j].  The growth rounds of the module-compression driver draw 2-4
// elements per round; one launch per element read the label matrix that many times.
struct SpmmKeys {
    uint64_t k[4];
};
template <int WMAX, int G>
__global__ void __launch_bounds__(256)
label_spmm_sload_kernel(int n, int npad, const uint32_t* __restrict__ L, SpmmKeys keys, int d,
                        const double* __restrict__ Wt, int w, int cols_per_block, double* __restrict__ P) {
    constexpr int JW = WMAX / 4;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* sV = smem;  // [G][d + 1] class values
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = blockIdx.x * 64 + lane;
    const int c_begin = blockIdx.y * cols_per_block;
    int c_end = c_begin + cols_per_block;
    if (c_end > n) c_end = n;
    const int dv = d + 1;
#pragma unroll
    for (int g = 0; g < G; ++g)
        for (int i = tid; i <= d; i += 256) sV[g * dv + i] = i ? sdpsr_class_uniform(keys.k[g], (uint32_t)i) : 0.0;
    __syncthreads();
    double acc[G][JW];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int j = 0; j < JW; ++j) acc[g][j] = 0.0;
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
            const double* __restrict__ wr = Ww + (int64_t)(cb + u) * JW;  // uniform address (rows >= n are zero)
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const double v = sV[g * dv + lab[u]];
#pragma unroll
                for (int j = 0; j < JW; ++j) acc[g][j] = fma(v, wr[j], acc[g][j]);
            }
        }
    }
    if (row_ok) {
        double* p = P + (int64_t)blockIdx.y * n * (G * w);
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int j = 0; j < JW; ++j) {
                const int jj = wave * JW + j;
                if (jj < w) p[r + (int64_t)(g * w + jj) * n] = acc[g][j];
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

// returns false when the shape is not supported (w > 64 or the class table does not fit in LDS).
// G keys: Y gets G blocks of w columns (G w <= 64).
bool launch_label_spmm_multi(hipStream_t s, int64_t n, const uint32_t* L, const uint64_t* keys, int G, int64_t d, const double* W,
                             int64_t ldw, int w, double* partials, double* Y, int64_t ldy) {
    if (w < 1 || G < 1 || G > 4 || G == 3 || G * w > 64 || d > 4000 || (size_t)G * (d + 2) * 8 > 64 * 1024) return false;
    const int rb = (int)((n + 63) / 64);
    int zg = 2048 / rb;
    if (zg < 1) zg = 1;
    if (zg > 32) zg = 32;
    const int zg_cap = zg;
    int cpb = (int)((n + zg - 1) / zg);
    cpb = (cpb + 7) / 8 * 8;  // the kernel walks its columns in steps of 8 (rounding to 128 halved the grid at n = 4104)
    zg = (int)((n + cpb - 1) / cpb);
    dim3 g((unsigned)rb, (unsigned)zg);
    // transposed zero-padded copy of W behind the partial sums (label_spmm_partial_doubles leaves room)
    double* Wt = partials + (size_t)zg_cap * n * 64;
    const int npad = (int)n + 8;
    SpmmKeys kk;
    for (int i = 0; i < 4; ++i) kk.k[i] = keys[i < G ? i : 0];
    auto go = [&](auto kern, int wmax) {
        transpose_pad_w_kernel<<<(unsigned)std::min<int64_t>(((int64_t)npad * wmax + 255) / 256, 2048), 256, 0, s>>>(
            (int)n, npad, w, wmax / 4, W, ldw, Wt);
        const size_t lds = (size_t)G * ((size_t)d + 2) * sizeof(double);
        kern<<<g, 256, lds, s>>>((int)n, npad, L, kk, (int)d, Wt, w, cpb, partials);
    };
    if (G == 1) {
        if (w <= 8) go(label_spmm_sload_kernel<8, 1>, 8);
        else if (w <= 16) go(label_spmm_sload_kernel<16, 1>, 16);
        else if (w <= 32) go(label_spmm_sload_kernel<32, 1>, 32);
        else if (w <= 48) go(label_spmm_sload_kernel<48, 1>, 48);
        else go(label_spmm_sload_kernel<64, 1>, 64);
    } else if (G == 2) {
        if (w <= 8) go(label_spmm_sload_kernel<8, 2>, 8);
        else if (w <= 16) go(label_spmm_sload_kernel<16, 2>, 16);
        else go(label_spmm_sload_kernel<32, 2>, 32);
    } else {
        if (w <= 8) go(label_spmm_sload_kernel<8, 4>, 8);
        else go(label_spmm_sload_kernel<16, 4>, 16);
    }
    const int wt = G * w;
    int64_t gr = ((int64_t)n * wt + 255) / 256;
    if (gr > 2048) gr = 2048;
    label_spmm_reduce_kernel<<<(unsigned)gr, 256, 0, s>>>((int)n, wt, zg, partials, Y, ldy);
    return true;
}
bool launch_label_spmm(hipStream_t s, int64_t n, const uint32_t* L, uint64_t key, int64_t d, const double* W,
                       int64_t ldw, int w, double* partials, double* Y, int64_t ldy) {
    if (w > 64) return false;
    return launch_label_spmm_multi(s, n, L, &key, 1, d, W, ldw, w, partials, Y, ldy);
}

}  // namespace sdpsr
