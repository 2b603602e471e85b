// Kernels of the module-compression driver of blockDiagonalize (DESIGN.md "module compression"):
// Y = A(v) W straight from the labels (label_spmm), tall-times-small products, split-K
// reductions and the small glue around them; plus helpers shared with the setup stage.
#include <algorithm>
#include <type_traits>
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

// dst (mp x mp, dense) <- symmetric part of the leading m x m block of src (ld lds_), zero elsewhere:
// memset + 2-D copy + symmetrize of the first compressed element in one launch
__global__ void extract_symmetric_kernel(int m, int mp, const double* __restrict__ src, int64_t lds_, double* __restrict__ dst) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= mp * mp) return;
    const int j = e / mp, i = e - j * mp;
    dst[e] = (i < m && j < m) ? 0.5 * (src[i + (int64_t)j * lds_] + src[j + (int64_t)i * lds_]) : 0.0;
}
void launch_extract_symmetric(hipStream_t s, int64_t m, int64_t mp, const double* src, int64_t lds_, double* dst) {
    extract_symmetric_kernel<<<(unsigned)((mp * mp + 255) / 256), 256, 0, s>>>((int)m, (int)mp, src, lds_, dst);
}

// out[:, c] = beta * out[:, c] + alpha * sum_{t < kk} In[:, t] * S[t + c * lds]   (tall-skinny times small)
// One wave = 64 rows x TS_COLS output columns: a row's In values are read once per TS_COLS outputs (they
// were read once per output: kk x ncols passes over In through L2), the coefficients are wave-uniform
// (scalar loads), eight In loads are in flight per lane.
constexpr int TS_COLS = 8;
__global__ void __launch_bounds__(64)
tall_times_small_kernel(int n, int64_t ldi, const double* __restrict__ In, int kk,
                        const double* __restrict__ S, int lds_, int ncols, double alpha, double beta,
                        double* __restrict__ out, int64_t ldo) {
    const int c0 = blockIdx.y * TS_COLS;
    const int i = blockIdx.x * 64 + threadIdx.x;
    const int ic = i < n ? i : n - 1;
    const double* in = In + ic;
    double acc[TS_COLS];
#pragma unroll
    for (int q = 0; q < TS_COLS; ++q) acc[q] = 0.0;
    // coefficient columns past ncols: clamped (their sums are not stored)
    const double* sc[TS_COLS];
#pragma unroll
    for (int q = 0; q < TS_COLS; ++q) sc[q] = S + (int64_t)(c0 + q < ncols ? c0 + q : ncols - 1) * lds_;
    int t = 0;
    for (; t + 8 <= kk; t += 8) {
        double x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = in[(int64_t)(t + u) * ldi];
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int q = 0; q < TS_COLS; ++q) acc[q] = fma(x[u], sc[q][t + u], acc[q]);
    }
    for (; t < kk; ++t) {
        const double x = in[(int64_t)t * ldi];
#pragma unroll
        for (int q = 0; q < TS_COLS; ++q) acc[q] = fma(x, sc[q][t], acc[q]);
    }
    if (i < n) {
#pragma unroll
        for (int q = 0; q < TS_COLS; ++q)
            if (c0 + q < ncols) {
                double* o = out + i + (int64_t)(c0 + q) * ldo;
                *o = (beta == 0.0 ? 0.0 : beta * *o) + alpha * acc[q];
            }
    }
}
void launch_tall_times_small(hipStream_t s, int64_t n, int64_t ldi, const double* In, int kk, const double* S,
                             int lds_, int ncols, double alpha, double beta, double* out, int64_t ldo) {
    if (ncols <= 0) return;
    dim3 g((unsigned)((n + 63) / 64), (unsigned)((ncols + TS_COLS - 1) / TS_COLS));
    tall_times_small_kernel<<<g, 64, 0, s>>>((int)n, ldi, In, kk, S, lds_, ncols, alpha, beta, out, ldo);
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
// Skinny Gram products of the module-compression driver:  C = A' B  with A (k x ma), B (k x nb),
// ma, nb <= 128 (the basis and candidate columns, a few dozen each) and k = n rows.  Padded to the
// 128 x 128 tile of the MFMA GEMM with K split over the batch this was one tile per workgroup on
// 32-64 CUs, 7x the bytes and flop of the exact shape (22 + 9 us at n = 4096).  Here a workgroup
// takes GS_ROWS rows of both matrices through LDS (transposed, pitch = 1 mod 16 doubles: the
// column-major reads are coalesced, the transposed writes and the row reads conflict-free), every
// thread accumulates a 4 x 4 register tile per 64 x 64 output block, the per-workgroup partial
// sums (exact shape) are added in fixed order by gram_reduce_kernel, which also writes the zero
// padding of the mp x np result the callers index.
// ---------------------------------------------------------------------------
constexpr int GS_ROWS = 32;

__global__ void __launch_bounds__(256)
gram_partial_kernel(int k, int ma, int nb, const double* __restrict__ A, int64_t lda, const double* __restrict__ B, int64_t ldb,
                    int pa, int pb, double* __restrict__ P) {
    extern __shared__ __attribute__((aligned(16))) double gs_smem[];
    double* sA = gs_smem;                 // [GS_ROWS][pa]
    double* sB = gs_smem + GS_ROWS * pa;  // [GS_ROWS][pb]
    const int tid = threadIdx.x;
    const int row0 = blockIdx.x * GS_ROWS;
    // batches of four loads per thread, all issued before the first LDS store; an index past the end is
    // clamped to the last element (the same value goes to the same place twice: harmless) so that no store
    // is guarded -- a guarded store pulls its load into the branch and the loads serialise
    auto stage = [&](const double* __restrict__ M, int64_t ldm, int cols, int pitch, double* dst) {
        const int total = GS_ROWS * cols;
        for (int e0 = tid; e0 < total; e0 += 256 * 4) {
            double v[4];
            int at[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                int e = e0 + 256 * q;
                e = e < total ? e : total - 1;
                const int r = e & (GS_ROWS - 1), i = e / GS_ROWS;
                const int rr = row0 + r < k ? row0 + r : k - 1;
                const double x = M[rr + (int64_t)i * ldm];
                v[q] = row0 + r < k ? x : 0.0;
                at[q] = r * pitch + i;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) dst[at[q]] = v[q];
        }
    };
    stage(A, lda, ma, pa, sA);
    stage(B, ldb, nb, pb, sB);
    __syncthreads();
    const int ti = tid & 15, tj = tid >> 4;
    double* p = P + (int64_t)blockIdx.x * ma * nb;
    for (int j0 = 0; j0 < nb; j0 += 64)
        for (int i0 = 0; i0 < ma; i0 += 64) {
            // outputs (i0 + ti + 16 a, j0 + tj + 16 b): consecutive lanes read consecutive doubles
            int ia[4], jb[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int i = i0 + ti + 16 * a, j = j0 + tj + 16 * a;
                ia[a] = i < ma ? i : ma - 1;
                jb[a] = j < nb ? j : nb - 1;
            }
            double acc[4][4];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
#pragma unroll 4
            for (int r = 0; r < GS_ROWS; ++r) {
                double xa[4], yb[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) xa[a] = sA[r * pa + ia[a]];
#pragma unroll
                for (int b = 0; b < 4; ++b) yb[b] = sB[r * pb + jb[b]];
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc[a][b] = fma(xa[a], yb[b], acc[a][b]);
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int j = j0 + tj + 16 * b;
                if (j < nb) {
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        const int i = i0 + ti + 16 * a;
                        if (i < ma) p[i + (int64_t)j * ma] = acc[a][b];
                    }
                }
            }
        }
}

// C[i + j ldc] = sum_z P[z][i + j ma]  (i < ma, j < nb), 0 on the rest of the mp x np result.
// Workgroup = 16 consecutive outputs x 16 groups of z: thread (o, g) adds P[z] for z = g, g + 16, ...
// (independent loads, one round trip), the 16 group sums of an output are added in group order.
__global__ void __launch_bounds__(256)
gram_reduce_kernel(int ma, int nb, int mp, int np, int Z, const double* __restrict__ P, double* __restrict__ C, int64_t ldc,
                   double* __restrict__ host_C) {
    __shared__ double part[16][17];
    const int o = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int e = blockIdx.x * 16 + o;  // index in the padded result
    const int j = e / mp, i = e - j * mp;
    const bool in = e < mp * np && i < ma && j < nb;
    double acc = 0;
    if (in) {
        const double* p = P + i + (int64_t)j * ma;
        const int64_t stride = (int64_t)ma * nb;
        double a0 = 0, a1 = 0;
        int z = g;
        for (; z + 16 < Z; z += 32) {
            a0 += p[(int64_t)z * stride];
            a1 += p[(int64_t)(z + 16) * stride];
        }
        if (z < Z) a0 += p[(int64_t)z * stride];
        acc = a0 + a1;
    }
    part[g][o] = acc;
    __syncthreads();
    if (g == 0 && e < mp * np) {
        double t = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += part[q][o];
        C[i + (int64_t)j * ldc] = t;
        // optional second copy straight into pinned host memory (same layout): the host reads the product after its
        // next synchronisation without a copy launch in between
        if (host_C) host_C[i + (int64_t)j * ldc] = t;
    }
}

size_t gram_small_partial_doubles(int64_t k, int ma, int nb) { return (size_t)((k + GS_ROWS - 1) / GS_ROWS) * ma * nb; }
// ma, nb <= 128; C gets the mp x np (padded) result, ldc >= mp
void launch_gram_small(hipStream_t s, int64_t k, int ma, int nb, const double* A, int64_t lda, const double* B, int64_t ldb,
                       double* partials, double* C, int64_t ldc, int mp, int np, double* host_C) {
    const int Z = (int)((k + GS_ROWS - 1) / GS_ROWS);
    const int pa = ((ma + 15) / 16) * 16 + 1, pb = ((nb + 15) / 16) * 16 + 1;
    const size_t lds = (size_t)GS_ROWS * (pa + pb) * sizeof(double);
    gram_partial_kernel<<<Z, 256, lds, s>>>((int)k, ma, nb, A, lda, B, ldb, pa, pb, partials);
    gram_reduce_kernel<<<(mp * np + 15) / 16, 256, 0, s>>>(ma, nb, mp, np, Z, partials, C, ldc, host_C);
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

// ---------------------------------------------------------------------------
// The same product on the fp64 matrix cores.  Y = A W is a GEMM whose left operand is formed on the
// fly: v_mfma_f64_16x16x4_f64 wants A as one double per lane (row m = lane & 15, k = lane >> 4), which
// is exactly one label look-up per lane per four matrix columns -- shared by all JT 16-column tiles
// of W, where the VALU form above spends one look-up per wave and column for WMAX/4 outputs only and
// waits on a scalar load per column.  Workgroup = 64 rows (wave = 16 rows) x a column range, walked
// in slabs of 64 columns: the slab of W' (row-major [c][STRIDE], STRIDE = 16 mod 32 doubles: the
// 4 x 16 operand read of a half-wave touches every LDS bank once) sits in one of two LDS buffers,
// the next slab travels global -> registers -> the other buffer behind the MFMAs of the current one
// (one barrier per slab), the labels of the next slab (16 steps of four columns) are prefetched
// meanwhile.  Per step and wave: 1 label load, G look-ups, JT operand reads, G*JT MFMAs.  The column
// split is as small as keeps two or three workgroups on every CU: the partial sums (z n w doubles,
// written here and read by the reduction) are the second HBM stream next to the 4 n^2 label bytes.
// Bound: the fp64 MFMA rate, 2 n^2 (16 JT) flop per element.  Fixed summation order: reproducible.
// ---------------------------------------------------------------------------
typedef double spmm_v4d __attribute__((ext_vector_type(4)));
constexpr int spmm_stride(int jt) { return (jt * 16) % 32 == 16 ? jt * 16 : jt * 16 + 16; }
constexpr int SPMM_SLAB = 64;  // columns per slab = 16 steps

// Wt[c][STRIDE] = W[c][0..w) zero-padded, rows c >= n zero.  32 x 32 tiles through LDS: reads walk a
// column of W, writes a row of W'.  grid (ceil(rows / 32), ceil(stride / 32)), block (32, 8).
__global__ void __launch_bounds__(256)
transpose_w_rowmajor_kernel(int n, int64_t rows, int w, int stride, const double* __restrict__ W, int64_t ldw,
                            double* __restrict__ Wt) {
    __shared__ double tile[32][33];
    const int c0 = blockIdx.x * 32, j0 = blockIdx.y * 32;
    const int tx = threadIdx.x, ty = threadIdx.y;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int j = j0 + ty + 8 * k, c = c0 + tx;
        tile[ty + 8 * k][tx] = (j < w && c < n) ? W[c + (int64_t)j * ldw] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + 8 * k, j = j0 + tx;
        if (c < rows && j < stride) Wt[(int64_t)c * stride + j] = tile[tx][ty + 8 * k];
    }
}

template <int JT, int G>
__global__ void __launch_bounds__(256)
label_spmm_mfma_kernel(int n, const uint32_t* __restrict__ L, SpmmKeys keys, int d, const double* __restrict__ Wt,
                       int w, int cols_per_block, double* __restrict__ P) {
    constexpr int STRIDE = spmm_stride(JT);
    constexpr int U = SPMM_SLAB / 4;             // steps of four columns per slab
    constexpr int SP = SPMM_SLAB * STRIDE / 2;   // double2 per slab: a multiple of 256 for every STRIDE
    constexpr int K = SP / 256;
    static_assert(SP % 256 == 0, "slab copy without guards");
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int dv = (d + 2) & ~1;  // class values per element, even: the slabs stay 16-byte aligned
    double* sV = smem;            // [G][dv]
    double* sW = smem + G * dv;   // [2][SPMM_SLAB][STRIDE]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15, kq = lane >> 4;
    const int c_begin = blockIdx.y * cols_per_block;
    const int nsteps = cols_per_block >> 2;
    const int nslabs = (nsteps + U - 1) / U;
    // label of (row, c_begin + 4 step + kq).  No masking: the address is clamped into the matrix and what a
    // clamped load brings is harmless -- rows >= n are never stored (an output row depends on its own
    // operand row only), columns >= n meet the zero rows of W', steps >= nsteps are never used.
    const int row = blockIdx.x * 64 + wave * 16 + m;
    const uint32_t* Lr = L + (row < n ? row : n - 1);
    auto fetch = [&](int step) -> uint32_t {
        int c = c_begin + 4 * step + kq;
        c = c < n ? c : n - 1;
        return Lr[(int64_t)c * n];
    };
    // slab sl of W' -> registers (W' has SPMM_SLAB zero or foreign rows behind the last range: never multiplied)
    typedef double spmm_v2d __attribute__((ext_vector_type(2)));
    spmm_v2d t[K];
    auto slab_load = [&](int sl) {
        const spmm_v2d* src = reinterpret_cast<const spmm_v2d*>(Wt + ((int64_t)c_begin + (int64_t)sl * SPMM_SLAB) * STRIDE);
#pragma unroll
        for (int k = 0; k < K; ++k) t[k] = src[tid + 256 * k];
    };
    auto slab_store = [&](int buf) {
        spmm_v2d* dst = reinterpret_cast<spmm_v2d*>(sW + buf * (SPMM_SLAB * STRIDE));
#pragma unroll
        for (int k = 0; k < K; ++k) dst[tid + 256 * k] = t[k];
    };
    uint32_t cur[U], nxt[U];
#pragma unroll
    for (int u = 0; u < U; ++u) cur[u] = fetch(u);  // in flight while the tables are filled
    slab_load(0);
#pragma unroll
    for (int g = 0; g < G; ++g)
        for (int i = tid; i <= d; i += 256) sV[g * dv + i] = i ? sdpsr_class_uniform(keys.k[g], (uint32_t)i) : 0.0;
    slab_store(0);
    __syncthreads();
    spmm_v4d acc[G][JT];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) acc[g][jt] = spmm_v4d{0.0, 0.0, 0.0, 0.0};
    // operands of step st (LDS -> registers) and the MFMAs on them: the reads of step st + 1 are issued
    // before the MFMAs of step st (pinned with sched_barrier).  Measured: the same time as with the reads
    // behind the MFMAs -- with loads, LDS reads and barriers all removed the loop runs within 12 % of
    // the full kernel; what separates it from the 64 clocks per MFMA of tools/probes/mfma_f64_probe.hip
    // is the prologue / epilogue of a 30 us launch, not the steady state.
    auto rd = [&](const double* wl, int st, uint32_t lab, double (&a)[G], double (&b)[JT]) {
        const double* wr = wl + st * (4 * STRIDE);
#pragma unroll
        for (int g = 0; g < G; ++g) a[g] = sV[g * dv + lab];
#pragma unroll
        for (int jt = 0; jt < JT; ++jt) b[jt] = wr[jt * 16];
    };
    auto mm = [&](const double (&a)[G], const double (&b)[JT]) {
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int jt = 0; jt < JT; ++jt)
                acc[g][jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(b[jt], a[g], acc[g][jt], 0, 0, 0);
    };
#pragma unroll 1
    for (int sl = 0; sl < nslabs; ++sl) {
        const bool more = sl + 1 < nslabs;  // uniform
        if (more) slab_load(sl + 1);
#pragma unroll
        for (int u = 0; u < U; ++u) nxt[u] = fetch((sl + 1) * U + u);
        const double* wl = sW + (sl & 1) * (SPMM_SLAB * STRIDE) + kq * STRIDE + m;
        const int ns = nsteps - sl * U;
        if (ns >= U) {  // whole slab: no branches
            double a0[G], b0[JT], a1[G], b1[JT];
            rd(wl, 0, cur[0], a0, b0);
#pragma unroll
            for (int u = 0; u < U; u += 2) {
                rd(wl, u + 1, cur[u + 1], a1, b1);
                __builtin_amdgcn_sched_barrier(0);
                mm(a0, b0);
                __builtin_amdgcn_sched_barrier(0);
                if (u + 2 < U) rd(wl, u + 2, cur[u + 2], a0, b0);
                __builtin_amdgcn_sched_barrier(0);
                mm(a1, b1);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (u < ns) {  // uniform
                    double a0[G], b0[JT];
                    rd(wl, u, cur[u], a0, b0);
                    mm(a0, b0);
                }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) cur[u] = nxt[u];
        if (more) slab_store((sl + 1) & 1);  // the buffer read two slabs ago: everybody is past the barrier since
        __syncthreads();
    }
    // D[j][r]: r = lane & 15 (matrix row), j = (lane >> 4) + 4 * reg (column of the tile)
    if (row < n) {
        double* p = P + (int64_t)blockIdx.y * n * (G * w);
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int jt = 0; jt < JT; ++jt)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int jj = jt * 16 + kq + 4 * v;
                    if (jj < w) p[row + (int64_t)(g * w + jj) * n] = acc[g][jt][v];
                }
    }
}

// per-device kernel attributes, set by sdpsr_create() (see gemm_set_device_attributes): two slabs of the
// four-tile form are 80 KiB
bool module_set_device_attributes() {
    bool ok = true;
    ok &= hipSuccess == hipFuncSetAttribute(reinterpret_cast<const void*>(&label_spmm_mfma_kernel<4, 1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                        150 * 1024);
    return ok;
}

size_t label_spmm_partial_doubles(int64_t n, int w) {
    int zg = (int)(2048 / ((n + 63) / 64));
    if (zg < 1) zg = 1;
    if (zg > 32) zg = 32;
    (void)w;
    // partial sums + transposed copy of W (VALU form: (n + 8) x 64; MFMA form: up to (n + 4 zg) x 80)
    return (size_t)zg * n * 64 + (size_t)(n + 8 + 4 * 128) * 80;
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
    if (n >= 64) {
        // matrix-core form.  Column split z: R = 2 or 3 workgroups on every CU, all resident at once (LDS:
        // class values + two slabs; registers: >= 4 waves per SIMD for every instantiation), the smallest
        // estimated time of  R x (columns per workgroup) MFMA steps  +  the partial sums through HBM.
        const int JT = (w + 15) / 16;
        const int stride = spmm_stride(JT);
        const size_t lds = (size_t)G * ((d + 2) & ~(int64_t)1) * 8 + (size_t)2 * SPMM_SLAB * stride * 8;
        int best_zg = 0, best_cpb = 0;
        double best_cost = 0;
        const int64_t lds_fit = (int64_t)(158 * 1024 / lds);
        for (int R = 1; R <= 3 && R <= lds_fit; ++R) {
            int64_t z0 = 256 * R / rb;
            if (z0 < 1) z0 = 1;
            if (z0 * 32 > n) z0 = std::max<int64_t>(1, n / 32);
            if (z0 * G * w > (int64_t)zg_cap * 64) z0 = (int64_t)zg_cap * 64 / (G * w);
            int cp = (int)((n + z0 - 1) / z0);
            cp = (cp + 3) / 4 * 4;
            const int z = (int)((n + cp - 1) / cp);
            const int64_t per_cu = ((int64_t)rb * z + 255) / 256;
            // microseconds: 64 clocks per MFMA at 2.4 GHz; partial sums written and read at ~4 TB/s
            const double t_mfma = (double)per_cu * (cp / 4) * (JT * G) * 64.0 / 2400.0 * (per_cu == 1 ? 1.5 : 1.0);
            const double t_part = (double)z * n * G * w * 16.0 / 4.0e6;
            const double cost = t_mfma + t_part;
            if (!best_zg || cost < best_cost) {
                best_zg = z;
                best_cpb = cp;
                best_cost = cost;
            }
        }
        if (best_zg && lds <= (JT == 4 ? 150 : 64) * 1024) {
            double* Wt = partials + (size_t)zg_cap * n * 64;
            const int64_t rows = (int64_t)best_zg * best_cpb + SPMM_SLAB;
            transpose_w_rowmajor_kernel<<<dim3((unsigned)((rows + 31) / 32), (unsigned)((stride + 31) / 32)), dim3(32, 8), 0, s>>>(
                (int)n, rows, w, stride, W, ldw, Wt);
            SpmmKeys kk;
            for (int i = 0; i < 4; ++i) kk.k[i] = keys[i < G ? i : 0];
            dim3 g((unsigned)rb, (unsigned)best_zg);
            auto go = [&](auto kern) { kern<<<g, 256, lds, s>>>((int)n, L, kk, (int)d, Wt, w, best_cpb, partials); };
            if (G == 1) {
                if (JT == 1) go(label_spmm_mfma_kernel<1, 1>);
                else if (JT == 2) go(label_spmm_mfma_kernel<2, 1>);
                else if (JT == 3) go(label_spmm_mfma_kernel<3, 1>);
                else go(label_spmm_mfma_kernel<4, 1>);
            } else if (G == 2) {
                if (JT == 1) go(label_spmm_mfma_kernel<1, 2>);
                else go(label_spmm_mfma_kernel<2, 2>);
            } else {
                go(label_spmm_mfma_kernel<1, 4>);
            }
            const int wt = G * w;
            int64_t gr = ((int64_t)n * wt + 255) / 256;
            if (gr > 2048) gr = 2048;
            label_spmm_reduce_kernel<<<(unsigned)gr, 256, 0, s>>>((int)n, wt, best_zg, partials, Y, ldy);
            return true;
        }
    }
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
