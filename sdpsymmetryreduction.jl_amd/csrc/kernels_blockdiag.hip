// Kernels of the block-diagonalisation stage (src/eigen_decomposition.jl:177-219,295-348,
// src/diagonalize.jl:42-89): eigenspace block norms, the small products of
// irreducible_decomposition, and basis_image as a segmented outer-product reduction.
#include <cstdlib>
#include <type_traits>
#include "host_internal.h"

namespace sdpsr {

typedef double v4d __attribute__((ext_vector_type(4)));

static inline int grid_for(int64_t work_items, int block, int max_blocks = 256 * 8) {
    int64_t g = (work_items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > max_blocks) g = max_blocks;
    return (int)g;
}

// ---------------------------------------------------------------------------
// block_norms(Q'AQ, eigdec, Inf), src/eigen_decomposition.jl:177-193: max |M[i,j]| per
// (eigenspace, eigenspace) block.  |x| >= 0 so the bit pattern orders like the value and
// atomicMax on uint64 is exact.  A wave first reduces lanes that fall in the same block.
// ---------------------------------------------------------------------------
__global__ void block_norms_kernel(int64_t n, int64_t ld, const double* __restrict__ M,
                                   const int32_t* __restrict__ space_of, int neig,
                                   unsigned long long* __restrict__ norms) {
    const int64_t len = n * n;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e0 = (int64_t)blockIdx.x * blockDim.x; e0 < len; e0 += stride) {
        const int64_t e = e0 + threadIdx.x;
        bool valid = e < len;
        int64_t j = valid ? e / n : 0, i = valid ? e - j * n : 0;
        int bi = space_of[i], bj = space_of[j];
        unsigned long long v = valid ? (unsigned long long)__double_as_longlong(fabs(M[i + j * ld])) : 0ull;
        int key = bi * neig + bj;
        // lanes of a wave are consecutive i in one column (mostly one block): reduce runs
        int prev_key = __shfl_up(key, 1, 64);
        bool head = ((threadIdx.x & 63) == 0) || prev_key != key;
        // segmented max via log-step scan over equal keys
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            unsigned long long ov = __shfl_down(v, o, 64);
            int ok = __shfl_down(key, o, 64);
            if ((int)(threadIdx.x & 63) + o < 64 && ok == key && ov > v) v = ov;
        }
        if (valid && head && v) atomicMax(&norms[key], v);
    }
}
// Q'AQ and its block norms for a small problem (n <= 64: the compressed problems of the
// module-compression driver) in ONE workgroup: A, Q in LDS, T = A Q, M = Q'T entry by entry into
// the block maxima.  Replaces two padded 128 x 128 MFMA launches (21 us each, pure latency) and the
// block-norm launch.  Also leaves M in Mout (ld-strided) for the callers that read it.
__global__ void __launch_bounds__(1024)
small_qtaq_block_norms_kernel(int n, int64_t ld, const double* __restrict__ A, const double* __restrict__ Q,
                              const int32_t* __restrict__ space_of, int neig, unsigned long long* __restrict__ norms,
                              double* __restrict__ Mout, double* __restrict__ Tout) {
    extern __shared__ __attribute__((aligned(16))) double sq[];
    const int ldl = n | 1;
    double* sA = sq;
    double* sQ = sA + (size_t)ldl * n;
    double* sT = sQ + (size_t)ldl * n;
    const int tid = threadIdx.x, nthr = blockDim.x;
    for (int e = tid; e < n * n; e += nthr) {
        const int j = e / n, i = e - j * n;
        sA[i + j * ldl] = A[i + (int64_t)j * ld];
        sQ[i + j * ldl] = Q[i + (int64_t)j * ld];
    }
    __syncthreads();
    for (int e = tid; e < n * n; e += nthr) {  // T = A Q (A symmetric: row i = column i)
        const int j = e / n, i = e - j * n;
        double acc = 0.0;
        for (int k = 0; k < n; ++k) acc = fma(sA[k + i * ldl], sQ[k + j * ldl], acc);
        sT[i + j * ldl] = acc;
        if (Tout) Tout[i + (int64_t)j * ld] = acc;
    }
    __syncthreads();
    for (int e = tid; e < n * n; e += nthr) {  // M[a, b] = Q[:, a]' T[:, b]
        const int b = e / n, a = e - b * n;
        double acc = 0.0;
        for (int k = 0; k < n; ++k) acc = fma(sQ[k + a * ldl], sT[k + b * ldl], acc);
        if (Mout) Mout[a + (int64_t)b * ld] = acc;
        const unsigned long long v = (unsigned long long)__double_as_longlong(fabs(acc));
        if (v) atomicMax(&norms[space_of[a] * neig + space_of[b]], v);
    }
}
void launch_small_qtaq_block_norms(hipStream_t s, int64_t n, int64_t ld, const double* A, const double* Q,
                                   const int32_t* space_of, int neig, unsigned long long* norms, double* Mout, double* Tout) {
    const size_t lds = (size_t)3 * (n | 1) * n * 8;
    small_qtaq_block_norms_kernel<<<1, 1024, lds, s>>>((int)n, ld, A, Q, space_of, neig, norms, Mout, Tout);
}
// The same with the clustering of the (ascending) eigenvalues done here, as the EigenDecomposition
// constructor does it (src/eigen_decomposition.jl:19-40: a new eigenspace where |dv| > atol), so that the
// host needs ONE read-back (status and values of the eigensolver, eigenspaces, block norms) instead of one
// after the eigensolver and one after the norms.
__global__ void __launch_bounds__(1024)
small_cluster_qtaq_block_norms_kernel(int n, int64_t ld, const double* __restrict__ A, const double* __restrict__ Q,
                                      const double* __restrict__ evals, double atol, int32_t* __restrict__ space_of,
                                      int32_t* __restrict__ meta, unsigned long long* __restrict__ norms,
                                      double* __restrict__ Tout, const int* __restrict__ einfo, double* __restrict__ vals_out) {
    extern __shared__ __attribute__((aligned(16))) double sq[];
    __shared__ int s_space[64];
    __shared__ int s_neig;
    const int ldl = n | 1;
    double* sA = sq;
    double* sQ = sA + (size_t)ldl * n;
    double* sT = sQ + (size_t)ldl * n;
    const int tid = threadIdx.x, nthr = blockDim.x;
    if (tid == 0) {
        int b = 0;
        s_space[0] = 0;
        for (int i = 1; i < n; ++i) {
            if (!(fabs(evals[i] - evals[i - 1]) <= atol)) ++b;
            s_space[i] = b;
        }
        s_neig = b + 1;
        meta[0] = einfo[0];  // status and sweep count of the eigensolver ride along
        meta[1] = einfo[1];
        meta[4] = b + 1;
    }
    if (tid < n) vals_out[tid] = evals[tid];
    for (int e = tid; e < n * n; e += nthr) {
        const int j = e / n, i = e - j * n;
        sA[i + j * ldl] = A[i + (int64_t)j * ld];
        sQ[i + j * ldl] = Q[i + (int64_t)j * ld];
    }
    __syncthreads();
    const int neig = s_neig;
    if (tid < n) space_of[tid] = s_space[tid];
    for (int e = tid; e < neig * neig; e += nthr) norms[e] = 0ull;
    for (int e = tid; e < n * n; e += nthr) {  // T = A Q (A symmetric: row i = column i)
        const int j = e / n, i = e - j * n;
        double acc = 0.0;
        for (int k = 0; k < n; ++k) acc = fma(sA[k + i * ldl], sQ[k + j * ldl], acc);
        sT[i + j * ldl] = acc;
        if (Tout) Tout[i + (int64_t)j * ld] = acc;
    }
    __syncthreads();  // T complete; the zeroes of norms are visible to the atomics of this workgroup
    for (int e = tid; e < n * n; e += nthr) {  // M[a, b] = Q[:, a]' T[:, b]
        const int b = e / n, a = e - b * n;
        double acc = 0.0;
        for (int k = 0; k < n; ++k) acc = fma(sQ[k + a * ldl], sT[k + b * ldl], acc);
        const unsigned long long v = (unsigned long long)__double_as_longlong(fabs(acc));
        if (v) atomicMax(&norms[s_space[a] * neig + s_space[b]], v);
    }
}
// pack (device, one contiguous read-back): [status, sweeps, -, -, eigenspaces | pad to 64 B][space_of n, padded to
// 64 B][values n][norms neig * neig]
size_t small_cluster_pack_bytes(int64_t n) { return 64 + (((size_t)n * 4 + 63) / 64) * 64 + (size_t)n * 8 + (size_t)n * n * 8; }
void launch_small_cluster_qtaq_block_norms(hipStream_t s, int64_t n, int64_t ld, const double* A, const double* Q,
                                           const double* evals, double atol, const int* einfo, char* pack, double* Tout) {
    const size_t lds = (size_t)3 * (n | 1) * n * 8;
    const size_t o_space = 64, o_vals = o_space + (((size_t)n * 4 + 63) / 64) * 64, o_norms = o_vals + (size_t)n * 8;
    small_cluster_qtaq_block_norms_kernel<<<1, 1024, lds, s>>>((int)n, ld, A, Q, evals, atol, (int32_t*)(pack + o_space),
                                                                (int32_t*)pack, (unsigned long long*)(pack + o_norms), Tout, einfo,
                                                                (double*)(pack + o_vals));
}
// ---------------------------------------------------------------------------
// Isomorphism classes of MANY eigenspaces (neig >= 256, e.g. the 1024 one-dimensional eigenspaces of a partition
// without symmetry): the neig x neig coupling matrix stays on the device.  What the host needs from it
// (src/eigen_decomposition.jl:83-139, 205-217) is: min / max of the entries (-> the 17 edges of the log-histogram), how
// many entries have c = 0..17 edges <= them (-> the Otsu threshold), and which pairs i < j reach the threshold (one bit
// each, for the union-find).  Three small kernels with a 16- / 144-byte / neig^2 / 8-byte read-back each instead of
// 8 neig^2 bytes through PCIe and four host passes over them.  Every quantity is an exact function of the entries
// (extrema, integer counts, comparisons): the classes are the ones the host code finds.
//   stat[0] = max of ~bits(x) over the entries (= ~bits(min); 0 when there is no entry), stat[1] = max of bits(x),
//   stat[2 + c] = number of entries with exactly c edges <= x.   (x >= 0: the bit pattern orders like the value; a NaN
//   never replaces an extremum and counts as 17, as in the host loops.)
// ---------------------------------------------------------------------------
// blocks between eigenspaces of different dimension count as zero (:185-186); norms[i, j] (i <= j: row space i) is
// mirrored like end_norm[i, j] = end_norm[j, i]
__global__ void __launch_bounds__(256)
coupling_symmetrize_minmax_kernel(int neig, unsigned long long* __restrict__ norms, const int32_t* __restrict__ dims,
                                  unsigned long long* __restrict__ stat) {
    const int64_t len = (int64_t)neig * neig;
    double mn = INFINITY, mx = 0.0;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < len; e += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(e / neig), j = (int)(e - (int64_t)i * neig);
        if (i > j) continue;
        const unsigned long long v = (dims[i] != dims[j]) ? 0ull : norms[e];
        norms[e] = v;
        norms[(int64_t)j * neig + i] = v;
        const double a = fabs(__longlong_as_double((long long)v));
        if (a < mn) mn = a;
        if (mx < a) mx = a;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double on = __shfl_xor(mn, o, 64), ox = __shfl_xor(mx, o, 64);
        if (on < mn) mn = on;
        if (mx < ox) mx = ox;
    }
    if ((threadIdx.x & 63) == 0) {
        if (mn < INFINITY) atomicMax(&stat[0], ~(unsigned long long)__double_as_longlong(mn));
        if (mx > 0.0) atomicMax(&stat[1], (unsigned long long)__double_as_longlong(mx));
    }
}
struct CouplingEdges {
    double e[17];
};
__global__ void __launch_bounds__(256)
coupling_count_kernel(int64_t len, const unsigned long long* __restrict__ norms, CouplingEdges ed,
                      unsigned long long* __restrict__ stat) {
    __shared__ unsigned int h[18];
    if (threadIdx.x < 18) h[threadIdx.x] = 0u;
    __syncthreads();
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < len; e += (int64_t)gridDim.x * blockDim.x) {
        const double x = __longlong_as_double((long long)norms[e]);
        int c = 0;
#pragma unroll
        for (int i = 0; i < 17; ++i) c += (ed.e[i] > x) ? 0 : 1;
        atomicAdd(&h[c], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 18 && h[threadIdx.x]) atomicAdd(&stat[2 + threadIdx.x], (unsigned long long)h[threadIdx.x]);
}
// bits[i * W + w], bit b: pair (i, j = 64 w + b), j > i, coupling >= thr
__global__ void __launch_bounds__(256)
coupling_bits_kernel(int neig, int W, const unsigned long long* __restrict__ norms, double thr,
                     unsigned long long* __restrict__ bits) {
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (wave >= (int64_t)neig * W) return;  // wave-uniform
    const int i = (int)(wave / W), w = (int)(wave - (int64_t)i * W);
    const int j = 64 * w + (threadIdx.x & 63);
    const bool on = j > i && j < neig && __longlong_as_double((long long)norms[(int64_t)i * neig + j]) >= thr;
    const unsigned long long m = __ballot(on);
    if ((threadIdx.x & 63) == 0) bits[wave] = m;
}
void launch_coupling_symmetrize_minmax(hipStream_t s, int neig, unsigned long long* norms, const int32_t* dims, unsigned long long* stat) {
    coupling_symmetrize_minmax_kernel<<<grid_for((int64_t)neig * neig, 256), 256, 0, s>>>(neig, norms, dims, stat);
}
void launch_coupling_count(hipStream_t s, int neig, const unsigned long long* norms, const double* edges17, unsigned long long* stat) {
    CouplingEdges ed;
    for (int i = 0; i < 17; ++i) ed.e[i] = edges17[i];
    coupling_count_kernel<<<grid_for((int64_t)neig * neig, 256), 256, 0, s>>>((int64_t)neig * neig, norms, ed, stat);
}
void launch_coupling_bits(hipStream_t s, int neig, const unsigned long long* norms, double thr, unsigned long long* bits) {
    const int W = (neig + 63) / 64;
    const int64_t threads = (int64_t)neig * W * 64;
    coupling_bits_kernel<<<(unsigned)((threads + 255) / 256), 256, 0, s>>>(neig, W, norms, thr, bits);
}

void launch_block_norms(hipStream_t s, int64_t n, int64_t ld, const double* M,
                        const int32_t* space_of, int neig, unsigned long long* norms) {
    block_norms_kernel<<<grid_for(n * n, 256), 256, 0, s>>>(n, ld, M, space_of, neig, norms);
}

// ---------------------------------------------------------------------------
// small dense helpers for irreducible_decomposition (src/eigen_decomposition.jl:329-344)
// ---------------------------------------------------------------------------
// out[j] = sum_i Q[i, col0 + j] * a[i]: one wave per column j
__global__ void gemv_t_kernel(int64_t n, int64_t ld, const double* __restrict__ Q, int64_t col0,
                              int64_t m, const double* __restrict__ a, double* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t j = wave; j < m; j += nw) {
        const double* q = Q + (col0 + j) * ld;
        double acc = 0;
        for (int64_t i = lane; i < n; i += 64) acc = fma(q[i], a[i], acc);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
        if (lane == 0) out[j] = acc;
    }
}
void launch_gemv_t(hipStream_t s, int64_t n, int64_t ld, const double* Q, int64_t col0,
                   int64_t m, const double* a, double* out) {
    gemv_t_kernel<<<grid_for(m * 64, 256), 256, 0, s>>>(n, ld, Q, col0, m, a, out);
}

// irreducible_decomposition, all (class, member) pairs in ONE launch
// (src/eigen_decomposition.jl:326-344).  Pair descriptor: 7 ints
//   {first column of E_i, dim E_i, first column of E_j, dim E_j, column of B for i, column of B for j, output column}.
// One workgroup per pair:  w = Q_j' b_i,  c = Q_i' b_j,  column = Q_j w / ||c||   (b_x = A q_x1).
__global__ void __launch_bounds__(256)
irreducible_pairs_kernel(int64_t n, int64_t ld, const double* __restrict__ Q, const double* __restrict__ Bf,
                         const int32_t* __restrict__ desc, double* __restrict__ Qhat) {
    extern __shared__ __attribute__((aligned(16))) double ip_smem[];  // w[mj], c[mi], 1 scalar
    const int32_t* dsc = desc + (int64_t)blockIdx.x * 7;
    const int ci = dsc[0], mi = dsc[1], cj = dsc[2], mj = dsc[3], fi = dsc[4], fj = dsc[5], oc = dsc[6];
    double* wv = ip_smem;
    double* cv = ip_smem + mj;
    double* sc = cv + mi;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double* bi = Bf + (int64_t)fi * ld;
    const double* bj = Bf + (int64_t)fj * ld;
    for (int t = wave; t < mj + mi; t += 4) {  // one wave per dot product
        const bool first = t < mj;
        const double* q = Q + (int64_t)(first ? cj + t : ci + (t - mj)) * ld;
        const double* b = first ? bi : bj;
        double acc = 0;
        for (int64_t r = lane; r < n; r += 64) acc = fma(q[r], b[r], acc);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
        if (lane == 0) (first ? wv[t] : cv[t - mj]) = acc;
    }
    __syncthreads();
    if (wave == 0) {
        double acc = 0;
        for (int t = lane; t < mi; t += 64) acc = fma(cv[t], cv[t], acc);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
        if (lane == 0) sc[0] = 1.0 / sqrt(acc);
    }
    __syncthreads();
    const double inv = sc[0];
    double* dst = Qhat + (int64_t)oc * n;
    for (int64_t r = threadIdx.x; r < n; r += 256) {
        double acc = 0;
        for (int t = 0; t < mj; ++t) acc = fma(Q[r + (int64_t)(cj + t) * ld], wv[t], acc);
        dst[r] = acc * inv;
    }
}
void launch_irreducible_pairs(hipStream_t s, int64_t n, int64_t ld, const double* Q, const double* Bf, int npairs,
                              int max_m2, const int32_t* desc, double* Qhat) {
    if (npairs <= 0) return;
    const size_t lds = (size_t)(max_m2 + 2) * sizeof(double);  // max over pairs of mi + mj
    irreducible_pairs_kernel<<<npairs, 256, lds, s>>>(n, ld, Q, Bf, desc, Qhat);
}

// dst[i] = inv_norm[0] * sum_j Q[i, col0 + j] * w[j]
__global__ void gemv_n_scaled_kernel(int64_t n, int64_t ld, const double* __restrict__ Q,
                                     int64_t col0, int64_t m, const double* __restrict__ w,
                                     const double* __restrict__ inv_norm,
                                     double* __restrict__ dst) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const double sc = inv_norm[0];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        double acc = 0;
        for (int64_t j = 0; j < m; ++j) acc = fma(Q[i + (col0 + j) * ld], w[j], acc);
        dst[i] = acc * sc;
    }
}
void launch_gemv_n_scaled(hipStream_t s, int64_t n, int64_t ld, const double* Q, int64_t col0,
                          int64_t m, const double* w, const double* inv_norm, double* dst) {
    gemv_n_scaled_kernel<<<grid_for(n, 256), 256, 0, s>>>(n, ld, Q, col0, m, w, inv_norm, dst);
}

__global__ void copy_col_kernel(int64_t n, const double* __restrict__ src,
                                double* __restrict__ dst) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        dst[i] = src[i];
}
void launch_copy_col(hipStream_t s, int64_t n, const double* src, double* dst) {
    copy_col_kernel<<<grid_for(n, 256), 256, 0, s>>>(n, src, dst);
}

// many column copies in one launch: the (source, destination) column pairs travel by value in
// the kernel arguments, so no index upload and no per-column launch
struct ColPairs {
    int32_t src[128];
    int32_t dst[128];
};
__global__ void copy_cols_kernel(int64_t n, ColPairs cp, const double* __restrict__ src, int64_t lds_,
                                 double* __restrict__ dst, int64_t ldd) {
    const int k = blockIdx.y;
    const double* a = src + (int64_t)cp.src[k] * lds_;
    double* b = dst + (int64_t)cp.dst[k] * ldd;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) b[i] = a[i];
}
void launch_copy_cols(hipStream_t s, int64_t n, int64_t count, const int32_t* src_cols, const int32_t* dst_cols,
                      const double* src, int64_t ld_src, double* dst, int64_t ld_dst) {
    for (int64_t k0 = 0; k0 < count; k0 += 128) {
        ColPairs cp;
        const int m = (int)((count - k0 < 128) ? (count - k0) : 128);
        for (int k = 0; k < m; ++k) {
            cp.src[k] = src_cols[k0 + k];
            cp.dst[k] = dst_cols[k0 + k];
        }
        dim3 g((unsigned)grid_for(n, 256), (unsigned)m);
        copy_cols_kernel<<<g, 256, 0, s>>>(n, cp, src, ld_src, dst, ld_dst);
    }
}

__global__ void clamptol_kernel(int64_t len, double* __restrict__ a, double atol) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < len; e += stride) {
        double v = a[e];
        a[e] = (fabs(v) < atol) ? 0.0 : v;
    }
}
void launch_clamptol(hipStream_t s, int64_t len, double* a, double atol) {
    clamptol_kernel<<<grid_for(len, 256), 256, 0, s>>>(len, a, atol);
}

// out[0] = 1 / ||v||_2   (single block)
__global__ void inv_norm_kernel(int64_t m, const double* __restrict__ v, double* out) {
    __shared__ double sh[4];
    double acc = 0;
    for (int64_t i = threadIdx.x; i < m; i += blockDim.x) acc = fma(v[i], v[i], acc);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = 1.0 / sqrt(sh[0] + sh[1] + sh[2] + sh[3]);
}
void launch_inv_norm(hipStream_t s, int64_t m, const double* v, double* out) {
    inv_norm_kernel<<<1, 256, 0, s>>>(m, v, out);
}

// ---------------------------------------------------------------------------
// basis_image, src/diagonalize.jl:64-89:
//   blks[i][k][a,b] = sum over entries (r,c) of class i of Q_k[r,a] * Q_k[c,b]
// Entries are grouped by class (stable radix sort by label), every class is cut into
// chunks of BI_CHUNK entries; a workgroup = (chunk, tile of BI_OT outputs): threads walk
// the entries, keep BI_OT accumulators in registers and reduce them across the workgroup.
// Chunk partials are summed in chunk order by a second kernel (bitwise reproducible).
// ---------------------------------------------------------------------------
constexpr int BI_THREADS = 256;
constexpr int BI_CHUNK = 4096;
constexpr int BI_OT = 16;

__global__ void __launch_bounds__(BI_THREADS)
basis_image_kernel(int64_t n, int64_t S1, int64_t S, const double* __restrict__ Qrm,
                   const uint32_t* __restrict__ ent, const int64_t* __restrict__ chunk_begin,
                   const int64_t* __restrict__ chunk_end, const int32_t* __restrict__ descA,
                   const int32_t* __restrict__ descB, double* __restrict__ partial) {
    __shared__ int sA[BI_OT], sB[BI_OT];
    __shared__ double red[BI_THREADS / 64][BI_OT];
    const int64_t chunk = blockIdx.x;
    const int64_t o0 = (int64_t)blockIdx.y * BI_OT;
    if (threadIdx.x < BI_OT) {
        int64_t o = o0 + threadIdx.x;
        sA[threadIdx.x] = (o < S) ? descA[o] : 0;
        sB[threadIdx.x] = (o < S) ? descB[o] : 0;
    }
    __syncthreads();
    double acc[BI_OT];
#pragma unroll
    for (int o = 0; o < BI_OT; ++o) acc[o] = 0.0;
    const int64_t b = chunk_begin[chunk], e_end = chunk_end[chunk];
    for (int64_t p = b + threadIdx.x; p < e_end; p += BI_THREADS) {
        const uint32_t lin = ent[p];
        const int64_t c = lin / n, r = lin - c * n;
        const double* qr = Qrm + r * S1;
        const double* qc = Qrm + c * S1;
#pragma unroll
        for (int o = 0; o < BI_OT; ++o) acc[o] = fma(qr[sA[o]], qc[sB[o]], acc[o]);
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int o = 0; o < BI_OT; ++o) {
        double v = acc[o];
#pragma unroll
        for (int sft = 32; sft > 0; sft >>= 1) v += __shfl_down(v, sft, 64);
        if (lane == 0) red[w][o] = v;
    }
    __syncthreads();
    if (threadIdx.x < BI_OT) {
        int64_t o = o0 + threadIdx.x;
        if (o < S) {
            double v = 0;
            for (int k = 0; k < BI_THREADS / 64; ++k) v += red[k][threadIdx.x];
            partial[chunk * S + o] = v;
        }
    }
}

__global__ void basis_image_reduce_kernel(int64_t d, int64_t S, const int64_t* __restrict__ chunk_ptr,
                                          const double* __restrict__ partial, double atol,
                                          double* __restrict__ out) {
    const int64_t total = d * S;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const int64_t cls = t / S, o = t - cls * S;
        double v = 0;
        for (int64_t ch = chunk_ptr[cls]; ch < chunk_ptr[cls + 1]; ++ch) v += partial[ch * S + o];
        out[t] = (fabs(v) < atol) ? 0.0 : v;
    }
}

void launch_basis_image(hipStream_t s, int64_t n, int64_t d, int64_t S1, int64_t S,
                        const double* Qrm, const uint32_t* ent, const int64_t* class_ptr,
                        const int32_t* descA, const int32_t* descB, const int64_t* chunk_ptr,
                        int64_t nchunks_total, const int32_t* chunk_class,
                        const int64_t* chunk_begin, const int64_t* chunk_end, double* partial,
                        double* out, double atol) {
    (void)class_ptr;
    (void)chunk_class;
    if (nchunks_total > 0 && S > 0) {
        dim3 g((unsigned)nchunks_total, (unsigned)((S + BI_OT - 1) / BI_OT));
        basis_image_kernel<<<g, BI_THREADS, 0, s>>>(n, S1, S, Qrm, ent, chunk_begin, chunk_end,
                                                    descA, descB, partial);
    }
    basis_image_reduce_kernel<<<grid_for(d * S, 256), 256, 0, s>>>(d, S, chunk_ptr, partial, atol,
                                                                   out);
}

// ---------------------------------------------------------------------------
// basis_image for MANY SMALL classes (d ~ n^2 / few dozen, e.g. QAP relaxations): one workgroup
// per (class i, block k) accumulates the rank-1 terms Q_k[r,:]' Q_k[c,:] of the class's entries
// as outer products: the two row segments of a batch of entries are staged in LDS, a thread owns
// one row index a of the s_k x s_k output and every G-th column b (accumulators in registers), so
// there is one LDS read per FMA, no partial sums and every output is written exactly once.
// ---------------------------------------------------------------------------
constexpr int BO_THREADS = 256;
constexpr int BO_EB = 8;       // entries staged per batch
constexpr int BO_MAXACC = 32;  // accumulators per thread and pass

__global__ void __launch_bounds__(BO_THREADS)
basis_image_outer_kernel(int64_t n, int64_t S1, int64_t S, const double* __restrict__ Qrm,
                         const uint32_t* __restrict__ ent, const int64_t* __restrict__ cls_ptr,
                         const int32_t* __restrict__ blk_col, const int32_t* __restrict__ blk_size,
                         const int64_t* __restrict__ blk_off, double atol, double* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) double bo_smem[];  // qr[EB][s], qc[EB][s]
    const int i = blockIdx.x, k = blockIdx.y;
    const int s = blk_size[k], cb = blk_col[k];
    const int64_t p_begin = cls_ptr[i + 1], p_end = cls_ptr[i + 2];  // label i + 1
    double* qr = bo_smem;
    double* qc = bo_smem + BO_EB * s;
    const int G = BO_THREADS / s;  // >= 1 (s <= 256)
    const int tid = threadIdx.x;
    const bool active = tid < G * s;
    const int a = active ? tid % s : 0, g = active ? tid / s : 0;
    double* o = out + (int64_t)i * S + blk_off[k];
    for (int b0 = 0; b0 < s; b0 += G * BO_MAXACC) {  // passes over the columns b (one pass if s <= G * 32)
        double acc[BO_MAXACC];
#pragma unroll
        for (int j = 0; j < BO_MAXACC; ++j) acc[j] = 0.0;
        for (int64_t p0 = p_begin; p0 < p_end; p0 += BO_EB) {
            const int ne = (int)((p_end - p0 < BO_EB) ? (p_end - p0) : BO_EB);
            __syncthreads();
            for (int t = tid; t < ne * s; t += BO_THREADS) {
                const int e = t / s, j = t - e * s;
                const uint32_t lin = ent[p0 + e];
                const int64_t c = lin / n, r = lin - c * n;
                qr[e * s + j] = Qrm[r * S1 + cb + j];
                qc[e * s + j] = Qrm[c * S1 + cb + j];
            }
            __syncthreads();
            if (active)
                for (int e = 0; e < ne; ++e) {
                    const double x = qr[e * s + a];
                    const double* qce = qc + e * s + b0 + g;
#pragma unroll
                    for (int j = 0; j < BO_MAXACC; ++j)
                        if (b0 + g + j * G < s) acc[j] = fma(x, qce[j * G], acc[j]);
                }
        }
        if (active)
#pragma unroll
            for (int j = 0; j < BO_MAXACC; ++j) {
                const int b = b0 + g + j * G;
                if (b < s) {
                    const double v = acc[j];
                    o[a + (int64_t)b * s] = (fabs(v) < atol) ? 0.0 : v;
                }
            }
    }
}

// ---------------------------------------------------------------------------
// The same (class i, block k) workgroups on the fp64 matrix cores: the block image is the small
// product  B = X Y'  with  X[a][e] = Q_k[r_e][a],  Y[b][e] = Q_k[c_e][b]  over the entries e of the
// class (K = class size, M = N = s_k).  The gathered row segments of a batch of 32 entries are
// staged in LDS (zero-padded to 16-element tiles and to the batch), a wave owns the tile rows
// ti = wave, wave + 4 and all tile columns, operands are 8-byte LDS reads of consecutive doubles,
// v_mfma_f64_16x16x4_f64 accumulates 16 x 16 tiles; every output is written exactly once, in
// entry order (fixed summation order: reproducible).  Blocks up to 128 (8 tiles per dimension).
// ---------------------------------------------------------------------------
constexpr int BM_EB = 32;  // entries per staged batch (8 MFMA k-steps)

template <int T>
__device__ __forceinline__ void bo_mfma_block(int s, int sp, int64_t p_begin, int64_t p_end, int64_t n, int64_t S1, int cb,
                                              const double* __restrict__ Qrm, const uint32_t* __restrict__ ent, double atol,
                                              double* __restrict__ o, double* qr, double* qc, int* s_rc) {
    constexpr int TA = (T + 3) / 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m = lane & 15, g = lane >> 4;
    v4d acc[TA][T];
#pragma unroll
    for (int u = 0; u < TA; ++u)
#pragma unroll
        for (int tj = 0; tj < T; ++tj) acc[u][tj] = v4d{0.0, 0.0, 0.0, 0.0};
    for (int64_t p0 = p_begin; p0 < p_end; p0 += BM_EB) {
        const int ne = (int)((p_end - p0 < BM_EB) ? (p_end - p0) : BM_EB);
        __syncthreads();  // the previous batch has been consumed
        if (tid < BM_EB) {
            uint32_t r = 0, c = 0;
            if (tid < ne) {
                const uint32_t lin = ent[p0 + tid];
                c = lin / (uint32_t)n;
                r = lin - c * (uint32_t)n;
            }
            s_rc[2 * tid] = (int)r;
            s_rc[2 * tid + 1] = (int)c;
        }
        __syncthreads();
        // wave w stages entries w, w + 4, ...; lanes walk the (padded) row segment
        for (int e = wave; e < BM_EB; e += 4) {
            const bool ev = e < ne;
            const double* rr = Qrm + (int64_t)s_rc[2 * e] * S1 + cb;
            const double* rc = Qrm + (int64_t)s_rc[2 * e + 1] * S1 + cb;
            for (int j = lane; j < sp; j += 64) {
                const bool in = ev && j < s;
                qr[e * sp + j] = in ? rr[j] : 0.0;
                qc[e * sp + j] = in ? rc[j] : 0.0;
            }
        }
        __syncthreads();
        const int ksteps = (ne + 3) >> 2;
        for (int ks = 0; ks < ksteps; ++ks) {
            const int e = 4 * ks + g;
            double xa[TA], yb[T];
#pragma unroll
            for (int u = 0; u < TA; ++u) {
                const int ti = wave + 4 * u;
                xa[u] = (ti < T) ? qr[e * sp + ti * 16 + m] : 0.0;
            }
#pragma unroll
            for (int tj = 0; tj < T; ++tj) yb[tj] = qc[e * sp + tj * 16 + m];
#pragma unroll
            for (int u = 0; u < TA; ++u)
#pragma unroll
                for (int tj = 0; tj < T; ++tj)
                    acc[u][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(yb[tj], xa[u], acc[u][tj], 0, 0, 0);
        }
    }
    // D[jj][ii]: ii = lane & 15 (row a), jj = (lane >> 4) + 4 * reg (column b).  (Round 3 tried sending the block through
    // LDS so that every wave instruction writes 1 KiB of consecutive addresses, 16 bytes per lane, instead of four 128-byte
    // pieces at stride 8 s: images of configs[2] 6.8 -> 8.0 ms.  The stores as they are were not the limit; not kept.)
#pragma unroll
    for (int u = 0; u < TA; ++u) {
        const int ti = wave + 4 * u;
        const int a = ti * 16 + m;
        if (ti < T && a < s) {
#pragma unroll
            for (int tj = 0; tj < T; ++tj)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int b = tj * 16 + g + 4 * r;
                    if (b < s) {
                        const double v = acc[u][tj][r];
                        __builtin_nontemporal_store((fabs(v) < atol) ? 0.0 : v, &o[a + (int64_t)b * s]);  // written once, never re-read here
                    }
                }
        }
    }
}

__global__ void __launch_bounds__(256, 2)
basis_image_outer_mfma_kernel(int64_t n, int64_t S1, int64_t S, const double* __restrict__ Qrm,
                              const uint32_t* __restrict__ ent, const int64_t* __restrict__ cls_ptr,
                              const int32_t* __restrict__ blk_col, const int32_t* __restrict__ blk_size,
                              const int64_t* __restrict__ blk_off, double atol, double* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) double bm_smem[];  // qr[EB][sp], qc[EB][sp], (r, c)[EB]
    const int i = blockIdx.x, k = blockIdx.y;
    const int s = blk_size[k], cb = blk_col[k];
    const int T = (s + 15) >> 4;
    const int sp = T * 16;
    const int64_t p_begin = cls_ptr[i + 1], p_end = cls_ptr[i + 2];  // label i + 1
    double* qr = bm_smem;
    double* qc = qr + BM_EB * sp;
    int* s_rc = reinterpret_cast<int*>(qc + BM_EB * sp);
    double* o = out + (int64_t)i * S + blk_off[k];
    switch (T) {  // uniform
        case 1: bo_mfma_block<1>(s, sp, p_begin, p_end, n, S1, cb, Qrm, ent, atol, o, qr, qc, s_rc); break;
        case 2: bo_mfma_block<2>(s, sp, p_begin, p_end, n, S1, cb, Qrm, ent, atol, o, qr, qc, s_rc); break;
        case 3: bo_mfma_block<3>(s, sp, p_begin, p_end, n, S1, cb, Qrm, ent, atol, o, qr, qc, s_rc); break;
        case 4: bo_mfma_block<4>(s, sp, p_begin, p_end, n, S1, cb, Qrm, ent, atol, o, qr, qc, s_rc); break;
        case 5: bo_mfma_block<5>(s, sp, p_begin, p_end, n, S1, cb, Qrm, ent, atol, o, qr, qc, s_rc); break;
        case 6: bo_mfma_block<6>(s, sp, p_begin, p_end, n, S1, cb, Qrm, ent, atol, o, qr, qc, s_rc); break;
        case 7: bo_mfma_block<7>(s, sp, p_begin, p_end, n, S1, cb, Qrm, ent, atol, o, qr, qc, s_rc); break;
        default: bo_mfma_block<8>(s, sp, p_begin, p_end, n, S1, cb, Qrm, ent, atol, o, qr, qc, s_rc); break;
    }
}

void launch_basis_image_outer(hipStream_t s, int64_t n, int64_t d, int64_t S1, int64_t S, int nblocks, int max_s,
                              const double* Qrm, const uint32_t* ent, const int64_t* cls_ptr,
                              const int32_t* blk_col, const int32_t* blk_size, const int64_t* blk_off, double atol,
                              double* out) {
    dim3 g((unsigned)d, (unsigned)nblocks);
    if (max_s >= 16 && max_s <= 128 && n < 65536) {
        // matrix-core form: worth it once a block spans at least one full MFMA tile
        const int spmax = ((max_s + 15) / 16) * 16;
        const size_t lds_m = (size_t)2 * BM_EB * spmax * sizeof(double) + (size_t)2 * BM_EB * sizeof(int);
        basis_image_outer_mfma_kernel<<<g, 256, lds_m, s>>>(n, S1, S, Qrm, ent, cls_ptr, blk_col, blk_size, blk_off, atol, out);
        return;
    }
    const size_t lds = (size_t)2 * BO_EB * max_s * sizeof(double);
    basis_image_outer_kernel<<<g, BO_THREADS, lds, s>>>(n, S1, S, Qrm, ent, cls_ptr, blk_col, blk_size, blk_off, atol, out);
}

// ---------------------------------------------------------------------------
// basis_image, two-stage form for symmetric partitions (the only ones diagonalize accepts):
//   stage 1   T[i][r][:] = sum over c with L[c,r] == i of Qhat[c,:]      ( = (1[P==i] Qhat)[r,:] )
//             one wave per row r: the wave walks column r of L (contiguous), lanes own the S1
//             columns of Qhat, per-class accumulators live in LDS (d x S1 doubles, single
//             wave => plain read-modify-write, fixed order, bitwise reproducible);
//   stage 2   blks[i][k] = Q_k' T_i[:, cols_k]   (s_k x s_k dots over n).
// Work n^2 S1 + d n sum s_k^2 instead of n^2 sum s_k^2 gathers.
// ---------------------------------------------------------------------------
template <int VEC>
__global__ void __launch_bounds__(64)
basis_image_rows_kernel(int n, int d, int S1, const uint32_t* __restrict__ L, const double* __restrict__ Qrm,
                        double* __restrict__ T, int lower) {
    // One wave per row r.  The n labels of column r of L (== row r, symmetric partition) are
    // first sorted by class with a STABLE counting sort in LDS (ranks inside a 64-entry chunk
    // come from ballots over the distinct labels of the chunk), then every class segment is
    // summed in index order with independent loads: no read-modify-write chains, fixed order.
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    int* s_off = reinterpret_cast<int*>(smem_raw);                       // [d + 2] running offsets
    int* s_start = s_off + (d + 2);                                      // [d + 2] segment starts
    unsigned short* s_pos = reinterpret_cast<unsigned short*>(s_start + (d + 2));  // [n] sorted entries
    const int lane = threadIdx.x;
    // lower != 0: only the strictly lower part of the row (entries c < r); the caller adds the
    // mirrored half and the diagonal (the partition and every 1[P==i] are symmetric).  Rows are
    // walked from the longest to the shortest so that the tail of the grid is short work.
    const int r = lower ? (n - 1 - (int)blockIdx.x) : (int)blockIdx.x;
    const uint32_t* col = L + (int64_t)r * n;
    const int nn = n;
    n = lower ? r : n;  // entries considered
    for (int t = lane; t < d + 2; t += 64) s_off[t] = 0;
    __syncthreads();
    // peers(l) = lanes of the chunk that hold the same label, from one ballot per label bit (no
    // loop over the distinct labels); the lowest peer is the leader that updates the counters.
    const int nbits = 32 - __clz(d);  // labels 0..d
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    // (four chunks per round: their label loads are issued together, the ballots follow)
    for (int c0 = 0; c0 < n; c0 += 256) {  // histogram, shifted by one
        uint32_t lq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = c0 + q * 64 + lane;
            lq[q] = (c < n) ? col[c] : 0u;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool valid = (c0 + q * 64 + lane) < n;
            const uint32_t l = lq[q];
            unsigned long long m = __ballot(valid);
            for (int b = 0; b < nbits; ++b) {
                const bool bit = (l >> b) & 1u;
                const unsigned long long bal = __ballot(bit);
                m &= bit ? bal : ~bal;
            }
            if (valid && (m & lt_mask) == 0ull) s_off[l + 1] += __popcll(m);
        }
    }
    __syncthreads();
    if (lane == 0) {  // exclusive scan (d + 1 classes incl. the zero class)
        int run = 0;
        for (int i = 0; i <= d; ++i) {
            const int cnt = s_off[i + 1];
            s_start[i] = run;
            run += cnt;
        }
        s_start[d + 1] = run;
    }
    __syncthreads();
    for (int t = lane; t < d + 2; t += 64) s_off[t] = s_start[t];
    __syncthreads();
    for (int c0 = 0; c0 < n; c0 += 256) {  // stable scatter of the entry indices
        uint32_t lq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = c0 + q * 64 + lane;
            lq[q] = (c < n) ? col[c] : 0u;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = c0 + q * 64 + lane;
            const bool valid = c < n;
            const uint32_t l = lq[q];
            unsigned long long m = __ballot(valid);
            for (int b = 0; b < nbits; ++b) {
                const bool bit = (l >> b) & 1u;
                const unsigned long long bal = __ballot(bit);
                m &= bit ? bal : ~bal;
            }
            if (valid) {
                const int base = s_off[l];  // read by every peer before the leader bumps it (in-order LDS)
                const int rank = __popcll(m & lt_mask);
                s_pos[base + rank] = (unsigned short)c;
                if (rank == 0) s_off[l] = base + __popcll(m);
            }
        }
    }
    __syncthreads();
    // Segment sums.  A lane owns VEC adjacent columns of Qhat (16-byte loads when VEC == 2); when
    // fewer than 64 lanes are needed for the S1 columns the wave splits into `groups` lane groups
    // that stride the segment, and the groups' sums are added in group order: fixed order, reproducible.
    typedef typename std::conditional<VEC == 2, double2, double>::type vec_t;
    const int cols = S1 / VEC;
    // groups of exactly `cols` lanes (not the next power of two: 17 vector columns are three entries per
    // load instruction instead of two, 51 busy lanes instead of 34)
    const int sub = cols < 64 ? cols : 64;
    const int groups = 64 / sub;
    const int jl = lane % sub, g = lane / sub;  // g == groups: idle lanes behind the last whole group
    const vec_t* __restrict__ Qv = reinterpret_cast<const vec_t*>(Qrm);
    const uint32_t diag = lower ? col[r] : 0u;
    for (int j0 = 0; j0 < cols; j0 += sub) {
        const int j = j0 + jl;
        const bool act = j < cols && g < groups;
        for (int i = 1; i <= d; ++i) {
            const int p0 = s_start[i], p1 = s_start[i + 1];
            double a[8][VEC];
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int v = 0; v < VEC; ++v) a[u][v] = 0.0;
            if (act) {
                int p = p0 + g;
                for (; p + 7 * groups < p1; p += 8 * groups) {
                    vec_t q[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) q[u] = Qv[(int64_t)s_pos[p + u * groups] * cols + j];
#pragma unroll
                    for (int u = 0; u < 8; ++u)
#pragma unroll
                        for (int v = 0; v < VEC; ++v) a[u][v] += reinterpret_cast<const double*>(&q[u])[v];
                }
                for (; p < p1; p += groups) {
                    const vec_t q = Qv[(int64_t)s_pos[p] * cols + j];
#pragma unroll
                    for (int v = 0; v < VEC; ++v) a[0][v] += reinterpret_cast<const double*>(&q)[v];
                }
            }
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                double t = ((a[0][v] + a[1][v]) + (a[2][v] + a[3][v])) + ((a[4][v] + a[5][v]) + (a[6][v] + a[7][v]));
                {  // the groups' sums in group order (lane jl of every group)
                    double tot = __shfl(t, jl, 64);
                    for (int g2 = 1; g2 < groups; ++g2) tot += __shfl(t, jl + g2 * sub, 64);
                    t = tot;
                }
                if (act && g == 0) {
                    // the diagonal entry enters with weight 1/2: the caller forms U + U' (below)
                    if (lower && diag == (uint32_t)i) t = fma(0.5, Qrm[(int64_t)r * S1 + (int64_t)j * VEC + v], t);
                    T[((int64_t)(i - 1) * nn + r) * S1 + (int64_t)j * VEC + v] = t;
                }
            }
        }
    }
}

// out[i][o] = sum_r Q[r][ca_o] T_i[r][cb_o] + Q[r][cb_o] T_i[r][ca_o]   (= U + U', see the rows kernel)
// for the S = sum s_k^2 outputs o of every class i; (ca_o, cb_o) come from a descriptor table.
// grid (d, ceil(S / 4)): a workgroup owns 4 outputs x 64 row groups, so 1 x 1 blocks (commutative
// algebras) use the whole workgroup just like large blocks do.
__global__ void __launch_bounds__(256)
basis_image_blocks_kernel(int n, int S1, int64_t S, const double* __restrict__ Qrm, const double* __restrict__ T,
                          const int32_t* __restrict__ colA, const int32_t* __restrict__ colB, double atol,
                          double* __restrict__ out) {
    __shared__ double red[256];
    const int i = blockIdx.x;
    const int64_t o = (int64_t)blockIdx.y * 4 + (threadIdx.x & 3);
    const int g = threadIdx.x >> 2;  // 64 row groups
    const double* Ti = T + (int64_t)i * n * S1;
    double a0 = 0, a1 = 0;
    if (o < S) {
        const int ca = colA[o], cb = colB[o];
        int r = g;
        for (; r + 64 < n; r += 128) {
            const double qa0 = Qrm[(int64_t)r * S1 + ca], qb0 = Qrm[(int64_t)r * S1 + cb];
            const double ta0 = Ti[(int64_t)r * S1 + ca], tb0 = Ti[(int64_t)r * S1 + cb];
            const double qa1 = Qrm[(int64_t)(r + 64) * S1 + ca], qb1 = Qrm[(int64_t)(r + 64) * S1 + cb];
            const double ta1 = Ti[(int64_t)(r + 64) * S1 + ca], tb1 = Ti[(int64_t)(r + 64) * S1 + cb];
            a0 = fma(qa0, tb0, a0);
            a0 = fma(qb0, ta0, a0);
            a1 = fma(qa1, tb1, a1);
            a1 = fma(qb1, ta1, a1);
        }
        for (; r < n; r += 64) {
            a0 = fma(Qrm[(int64_t)r * S1 + ca], Ti[(int64_t)r * S1 + cb], a0);
            a0 = fma(Qrm[(int64_t)r * S1 + cb], Ti[(int64_t)r * S1 + ca], a0);
        }
    }
    red[threadIdx.x] = a0 + a1;
    __syncthreads();
    if (threadIdx.x < 4 && o < S) {
        double v = 0;
        for (int gg = 0; gg < 64; ++gg) v += red[gg * 4 + threadIdx.x];
        out[(int64_t)i * S + o] = (fabs(v) < atol) ? 0.0 : v;
    }
}

bool basis_image_two_stage_fits(int64_t n, int64_t d, int64_t S1) {
    return n <= 65535 && (2 * (d + 2) * 4 + n * 2) <= 60 * 1024 && d * n * S1 * 8 <= ((int64_t)4 << 30);
}

// Class sums of ONE vector, few classes: out[(i-1) n + r] = sum over c with L[c,r] == i of x[c].
// x sits in LDS, every lane adds into its private (d+1)-entry table in LDS (no sort, no atomics),
// then lane i adds the 64 tables of class i in lane order: fixed order, reproducible.
// Workgroup = 4 waves = 4 rows at a time, sharing x.
__global__ void __launch_bounds__(256)
class_sums_small_d_kernel(int n, int d, int tstride, const uint32_t* __restrict__ L, const double* __restrict__ x,
                          double* __restrict__ out, int64_t ldo) {
    extern __shared__ __attribute__((aligned(16))) double cs_smem[];
    double* sx = cs_smem;                                  // [n]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double* acc = cs_smem + n + (size_t)wave * 64 * tstride;  // [64][tstride], tstride odd >= d + 1
    double* mine = acc + (size_t)lane * tstride;
    for (int c = threadIdx.x; c < n; c += 256) sx[c] = x[c];
    __syncthreads();
    const int rows_per_round = gridDim.x * 4;
    // The work of a wave is a flat list of batches b = (row index rr, eight 64-column chunks bb): labels of
    // (row, columns 512 bb + 64 u + lane).  Three batches are in flight while one is added -- with one batch
    // per HBM round trip the latency was the whole time of this kernel.  The four buffers rotate by name
    // (a register move of a buffer still in flight would wait for it).  Pure loads from clamped addresses:
    // a row past the matrix is summed and not stored, a column past it adds 0.
    const int bpr = (n + 511) / 512;                                       // batches per row
    const int my_rows = (n - blockIdx.x * 4 + rows_per_round - 1) / rows_per_round;  // rounds of this workgroup (>= 1)
    const int total = my_rows * bpr;
    auto fetch = [&](int bq, uint32_t (&lab)[8]) {
        const int rr = bq / bpr, bb = bq - rr * bpr;
        const int r = blockIdx.x * 4 + rr * rows_per_round + wave;
        const uint32_t* col = L + (int64_t)((bq < total && r < n) ? r : 0) * n;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = bb * 512 + u * 64 + lane;
            lab[u] = col[c < n ? c : n - 1];
        }
    };
    auto process = [&](int bq, const uint32_t (&lab)[8]) {
        if (bq >= total) return;  // uniform over the workgroup
        const int rr = bq / bpr, bb = bq - rr * bpr;
        if (bb == 0)
            for (int i = 0; i <= d; ++i) mine[i] = 0.0;
        // LDS adds without a return value: the eight updates of a batch queue up behind each other in the LDS pipe (in order
        // per wave, so the sums are formed in the same order as by `mine[l] += v`) instead of eight read - add - write round
        // trips, each waiting for the one before (round 4: 32.9 -> 29.2 us at N = 4096, d = 34)
        double vv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = bb * 512 + u * 64 + lane;
            vv[u] = c < n ? sx[c] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) atomicAdd(&mine[lab[u]], vv[u]);
        if (bb == bpr - 1) {
            const int r = blockIdx.x * 4 + rr * rows_per_round + wave;
            __syncthreads();
            if (r < n)
                for (int i = 1 + lane; i <= d; i += 64) {
                    double t = 0;
                    for (int l2 = 0; l2 < 64; ++l2) t += acc[(size_t)l2 * tstride + i];
                    out[(int64_t)(i - 1) * ldo + r] = t;
                }
            __syncthreads();
        }
    };
    uint32_t qa[8], qb[8], qc[8], qd[8];
    fetch(0, qa);
    fetch(1, qb);
    fetch(2, qc);
#pragma unroll 1
    for (int bq = 0; bq < total; bq += 4) {
        fetch(bq + 3, qd);
        process(bq, qa);
        fetch(bq + 4, qa);
        process(bq + 1, qb);
        fetch(bq + 5, qb);
        process(bq + 2, qc);
        fetch(bq + 6, qc);
        process(bq + 3, qd);
    }
}

// ---------------------------------------------------------------------------
// basis_image through the block structure (commutative case: every block 1 x 1).
//
// Q_hat from Murota's decomposition spans invariant subspaces: A_i q_k = lambda_ik q_k for every class
// matrix A_i = 1[P==i] and every column q_k, and the columns are orthonormal.  Hence for
// x = sum_k q_k:   q_k'(A_i x) = lambda_ik = q_k' A_i q_k = blks[i][k]  -- ONE vector's class sums
// (n^2 label reads, no gathers of Q_hat rows) instead of the class sums of all S1 columns
// (n^2 S1 / 2 gathered 8-byte words, the L2-bound 190 us of basis_image_rows_kernel at N = 4096).
// The identity is only as good as the invariance, so it is CHECKED, not assumed: a second vector
// x' = sum_k g_k q_k with generic real weights g_k in [0.5, 1.5) goes through the same pass (double2 lanes), and
// q_k'(A_i x) must agree with q_k'(A_i x') / g_k -- a cross term q_k' A_i q_j, j != k, enters the two with the
// different factors 1 and g_j / g_k, so ANY non-zero cross term shows with probability 1 (round 3 used random signs:
// a coupling of two columns with equal signs went unseen, probability 1/2 per pair).  The verdict is per COLUMN k:
// the columns that fail (typically the two eigenvectors of a pair of close eigenvalues of the random generic element,
// accurate to eps |A| / gap only) get their images from the projection formula q_k' 1[P==i] q_k itself, two columns per
// extra class-sum pass (launch_basis_image_fix_pair); only when many columns fail does the caller run the projection
// formula for all of them (two-stage kernels above), which is also what SDPSR_FLAG_FULL_BASIS_IMAGE always does.
// ---------------------------------------------------------------------------
__device__ __forceinline__ double bi_weight(uint64_t key, int k) {
    return 0.5 + (double)(sdpsr_fmix64(key + 0x9E3779B97F4A7C15ULL * (uint64_t)(k + 1)) >> 11) * (1.0 / 9007199254740992.0);
}

// X[r] = (sum_k Q[r,k], sum_k g_k Q[r,k]);  Qrm row-major n x S1;  flag[0] = failing columns so far, flag[1 + k] = column k failed
__global__ void bi_signed_sums_kernel(int n, int S1, const double* __restrict__ Qrm, uint64_t key, double2* __restrict__ X, uint32_t* flag) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.x == 0)
        for (int k = threadIdx.x; k <= S1; k += blockDim.x) flag[k] = 0u;
    if (r >= n) return;
    double a = 0, b = 0;
    for (int k = 0; k < S1; ++k) {
        const double q = Qrm[(int64_t)r * S1 + k];
        a += q;
        b = fma(bi_weight(key, k), q, b);
    }
    X[r] = make_double2(a, b);
}

// X[r] = (Q[r, k1], Q[r, k2]): the pair of columns whose images are recomputed by the projection formula
__global__ void bi_pair_vector_kernel(int n, int S1, const double* __restrict__ Qrm, int k1, int k2, double2* __restrict__ X) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) X[r] = make_double2(Qrm[(int64_t)r * S1 + k1], Qrm[(int64_t)r * S1 + k2]);
}

// out[i * S1 + k1] = q_k1' (A_i q_k1), out[i * S1 + k2] = q_k2' (A_i q_k2) from the class sums Y of that pair; grid d
__global__ void __launch_bounds__(256)
bi_contract_pair_kernel(int n, int S1, const double* __restrict__ Qrm, const double2* __restrict__ Y, int k1, int k2, double atol,
                        double* __restrict__ out) {
    __shared__ double red[2][256];
    const int i = blockIdx.x;
    const double2* Yi = Y + (int64_t)i * n;
    double a0 = 0, a1 = 0;
    for (int r = threadIdx.x; r < n; r += 256) {
        const double2 y = Yi[r];
        a0 = fma(Qrm[(int64_t)r * S1 + k1], y.x, a0);
        a1 = fma(Qrm[(int64_t)r * S1 + k2], y.y, a1);
    }
    red[0][threadIdx.x] = a0;
    red[1][threadIdx.x] = a1;
    __syncthreads();
    if (threadIdx.x < 2) {
        double v = 0;
        for (int g = 0; g < 256; ++g) v += red[threadIdx.x][g];
        const int k = threadIdx.x == 0 ? k1 : k2;
        out[(int64_t)i * S1 + k] = (fabs(v) < atol) ? 0.0 : v;
    }
}

// Class sums of a PAIR of vectors: out[(i-1) n + r] = sum over c with L[c,r] == i of X[c]  (both halves).
// Same scheme as class_sums_small_d_kernel (private per-lane tables in LDS, fixed summation order), the
// tables hold double2 (one 16-byte read-modify-write per entry); X comes from global memory (64 KiB at
// N = 4096, L2-resident, coalesced along c) and is prefetched with the labels.  W waves per workgroup:
// W * 64 * tstride * 16 bytes of LDS.
template <int W>
__global__ void __launch_bounds__(64 * W)
class_sums2_kernel(int n, int d, int tstride, const uint32_t* __restrict__ L, const double2* __restrict__ X,
                   double2* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) double2 cs2_smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double2* acc = cs2_smem + (size_t)wave * 64 * tstride;  // [64][tstride], tstride odd >= d + 1
    double2* mine = acc + (size_t)lane * tstride;
    const int rows_per_round = gridDim.x * W;
    const int bpr = (n + 511) / 512;  // batches of 8 x 64 columns per row
    const int my_rows = (n - (int)blockIdx.x * W + rows_per_round - 1) / rows_per_round;
    const int total = my_rows * bpr;
    auto fetch = [&](int bq, uint32_t (&lab)[8], double2 (&xv)[8]) {
        const int rr = bq / bpr, bb = bq - rr * bpr;
        const int r = blockIdx.x * W + rr * rows_per_round + wave;
        const uint32_t* col = L + (int64_t)((bq < total && r < n) ? r : 0) * n;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = bb * 512 + u * 64 + lane;
            const int cc = c < n ? c : n - 1;
            lab[u] = col[cc];
            xv[u] = X[cc];
        }
    };
    auto process = [&](int bq, const uint32_t (&lab)[8], const double2 (&xv)[8]) {
        if (bq >= total) return;  // uniform over the workgroup
        const int rr = bq / bpr, bb = bq - rr * bpr;
        if (bb == 0)
            for (int i = 0; i <= d; ++i) mine[i] = make_double2(0.0, 0.0);
        // (LDS adds without a return value as in class_sums_small_d_kernel -- two per entry here -- measured no faster: 39.8 -> 41.3 us)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = bb * 512 + u * 64 + lane;
            if (c < n) {
                double2 t = mine[lab[u]];
                t.x += xv[u].x;
                t.y += xv[u].y;
                mine[lab[u]] = t;
            }
        }
        if (bb == bpr - 1) {
            const int r = blockIdx.x * W + rr * rows_per_round + wave;
            __syncthreads();
            if (r < n)
                for (int i = 1 + lane; i <= d; i += 64) {
                    double tx = 0, ty = 0;
                    for (int l2 = 0; l2 < 64; ++l2) {
                        const double2 v = acc[(size_t)l2 * tstride + i];
                        tx += v.x;
                        ty += v.y;
                    }
                    out[(int64_t)(i - 1) * n + r] = make_double2(tx, ty);
                }
            __syncthreads();
        }
    };
    uint32_t la[8], lb[8], lc[8];
    double2 xa[8], xb[8], xc[8];
    fetch(0, la, xa);
    fetch(1, lb, xb);
#pragma unroll 1
    for (int bq = 0; bq < total; bq += 3) {
        fetch(bq + 2, lc, xc);
        process(bq, la, xa);
        fetch(bq + 3, la, xa);
        process(bq + 1, lb, xb);
        fetch(bq + 4, lb, xb);
        process(bq + 2, lc, xc);
    }
}

// out[i * S1 + k] = q_k' Y_i (first halves), checked against q_k' Y_i (second halves) / g_k: where they differ by more
// than tol, column k is marked (flag[1 + k] = 1, counted once in flag[0]).  grid (d, ceil(S1 / 4)), 256 threads = 4
// outputs x 64 row groups.
__global__ void __launch_bounds__(256)
bi_contract_check_kernel(int n, int S1, const double* __restrict__ Qrm, const double2* __restrict__ Y, uint64_t key, double atol,
                         double tol, double* __restrict__ out, uint32_t* __restrict__ flag) {
    __shared__ double red[2][256];
    const int i = blockIdx.x;
    const int k = blockIdx.y * 4 + (threadIdx.x & 3);
    const int g = threadIdx.x >> 2;
    const double2* Yi = Y + (int64_t)i * n;
    double a0 = 0, a1 = 0;
    if (k < S1)
        for (int r0 = g; r0 < n; r0 += 64 * 8) {  // eight rows per trip, their loads in flight together
            double q[8];
            double2 y[8];
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                const int r = r0 + 64 * x;
                const int rc = r < n ? r : g;
                q[x] = r < n ? Qrm[(int64_t)rc * S1 + k] : 0.0;
                y[x] = Yi[rc];
            }
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                a0 = fma(q[x], y[x].x, a0);
                a1 = fma(q[x], y[x].y, a1);
            }
        }
    red[0][threadIdx.x] = a0;
    red[1][threadIdx.x] = a1;
    __syncthreads();
    if (threadIdx.x < 4 && k < S1) {
        double v0 = 0, v1 = 0;
        for (int gg = 0; gg < 64; ++gg) {
            v0 += red[0][gg * 4 + threadIdx.x];
            v1 += red[1][gg * 4 + threadIdx.x];
        }
        if (!(fabs(v0 - v1 / bi_weight(key, k)) <= tol)) {
            if (atomicExch(&flag[1 + k], 1u) == 0u) atomicAdd(&flag[0], 1u);
        }
        out[(int64_t)i * S1 + k] = (fabs(v0) < atol) ? 0.0 : v0;
    }
}

size_t basis_image_commutative_workspace_doubles(int64_t n, int64_t d) { return (size_t)2 * n * (d + 1); }
// All blocks 1 x 1 (S = S1).  ws: 2 n (d + 1) doubles.  Returns false when the shape has no instance (d > 148);
// after the launches flag[0] = number of columns whose check failed, flag[1 + k] = 1 for those columns (flag: 1 + S1
// words the device can write, e.g. pinned host memory): their entries of `out` must be recomputed by the projection
// formula (launch_basis_image_fix_pair, or everything by the caller).
bool launch_basis_image_commutative(hipStream_t s, int64_t n, int64_t d, int64_t S1, const uint32_t* L, const double* Qrm, uint64_t key,
                                    double atol, double tol, double* ws, double* out, uint32_t* flag) {
    const int tstride = (int)((d + 1) | 1);
    const size_t per_wave = (size_t)64 * tstride * sizeof(double2);
    int W = 4;
    while (W > 1 && per_wave * W > 150 * 1024) W >>= 1;
    if (per_wave * W > 150 * 1024 || n < 1 || n > 0x7FFFFFFF / 2) return false;
    double2* X = reinterpret_cast<double2*>(ws);
    double2* Y = X + n;
    bi_signed_sums_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>((int)n, (int)S1, Qrm, key, X, flag);
    int g = (int)((n + W - 1) / W);
    if (g > 256) g = 256;  // one resident workgroup per CU, rows in rounds
    const size_t lds = per_wave * W;
    if (W == 4) class_sums2_kernel<4><<<g, 256, lds, s>>>((int)n, (int)d, tstride, L, X, Y);
    else if (W == 2) class_sums2_kernel<2><<<g, 128, lds, s>>>((int)n, (int)d, tstride, L, X, Y);
    else class_sums2_kernel<1><<<g, 64, lds, s>>>((int)n, (int)d, tstride, L, X, Y);
    dim3 gc((unsigned)d, (unsigned)((S1 + 3) / 4));
    bi_contract_check_kernel<<<gc, 256, 0, s>>>((int)n, (int)S1, Qrm, Y, key, atol, tol, out, flag);
    return true;
}

// The projection formula for two columns k1, k2 of a commutative Q_hat (k2 = k1 allowed): one more class-sum pass of
// the pair (q_k1, q_k2) and its contraction, overwriting out[i * S1 + k1], out[i * S1 + k2] for every class i.
bool launch_basis_image_fix_pair(hipStream_t s, int64_t n, int64_t d, int64_t S1, const uint32_t* L, const double* Qrm, int k1, int k2,
                                 double atol, double* ws, double* out) {
    const int tstride = (int)((d + 1) | 1);
    const size_t per_wave = (size_t)64 * tstride * sizeof(double2);
    int W = 4;
    while (W > 1 && per_wave * W > 150 * 1024) W >>= 1;
    if (per_wave * W > 150 * 1024 || n < 1 || n > 0x7FFFFFFF / 2) return false;
    double2* X = reinterpret_cast<double2*>(ws);
    double2* Y = X + n;
    bi_pair_vector_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>((int)n, (int)S1, Qrm, k1, k2, X);
    int g = (int)((n + W - 1) / W);
    if (g > 256) g = 256;
    const size_t lds = per_wave * W;
    if (W == 4) class_sums2_kernel<4><<<g, 256, lds, s>>>((int)n, (int)d, tstride, L, X, Y);
    else if (W == 2) class_sums2_kernel<2><<<g, 128, lds, s>>>((int)n, (int)d, tstride, L, X, Y);
    else class_sums2_kernel<1><<<g, 64, lds, s>>>((int)n, (int)d, tstride, L, X, Y);
    bi_contract_pair_kernel<<<(unsigned)d, 256, 0, s>>>((int)n, (int)S1, Qrm, Y, k1, k2, atol, out);
    return true;
}

// ---------------------------------------------------------------------------
// basis_image through the block structure, blocks up to 3 x 3 (round 5; the 1 x 1 case above is its special case).
//
// Murota's columns of one block, Q_k = [q_k1 .. q_ks], span an INVARIANT subspace: the isotypic component is
// C^s (x) C^m, the eigenspaces of the generic element are e_j (x) C^m, and irreducible_decomposition takes q_kj = e_j (x) u
// with one and the same u for j = 1 .. s (src/eigen_decomposition.jl:326-344: q_k1 is an eigenvector, the others are its
// images under the coupling blocks) -- so 1[P==i] Q_k = Q_k B for the s x s image B = blks[i][k], and the subspaces of
// different blocks are orthogonal.  For x = sum_k Q_k g_k (one s_k-vector of weights per block):  Q_k'(1[P==i] x) = B g_k.
// FOUR vectors -- two passes of the pair class-sum kernel, instead of the class sums of all S1 columns (the two-stage
// kernels: 0.2 ms at ER(7) (x) K_72) -- give B G = Y with G = [g^(0) .. g^(s-1)] (identity + a generic perturbation:
// well conditioned) and at least one more generic vector to CHECK: B g^(s) must reproduce Y's column s.  The identity is
// only as good as the invariance, so a block whose check fails (close eigenvalues of the random generic element) gets its
// image from the projection formula itself -- the same kernels with the weights e_1 .. e_s on that block alone.
// ---------------------------------------------------------------------------
constexpr int BIB_T = 4;  // vectors per run (two double2 passes)
// weight of column j (of block k, size sz) in vector t
__device__ __forceinline__ double bib_weight(uint64_t key, int k, int j, int t, int sz) {
    const double u = (double)(sdpsr_fmix64(key + 0x9E3779B97F4A7C15ULL * (uint64_t)(1 + t + BIB_T * (j + 4 * k))) >> 11) * (1.0 / 9007199254740992.0);
    if (t < sz) return (t == j ? 1.0 : 0.0) + 0.25 * (u - 0.5);
    return 0.5 + u;  // check vectors: generic
}
// X0[r] = (x0, x1), X1[r] = (x2, x3);  only >= 0: weights e_t on block `only` alone (the projection formula for that block)
__global__ void bib_vectors_kernel(int n, int S1, int nblocks, const int32_t* __restrict__ blk_col, const int32_t* __restrict__ blk_size, uint64_t key, int only,
                                   const double* __restrict__ Qrm, double2* __restrict__ X0, double2* __restrict__ X1, uint32_t* flag,
                                   double* __restrict__ ginv) {
    extern __shared__ double bib_w[];  // weights of the S1 columns, BIB_T each (all blocks at once)
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (flag && blockIdx.x == 0)
        for (int k = threadIdx.x; k <= nblocks; k += blockDim.x) flag[k] = 0u;
    if (only < 0) {
        for (int k = threadIdx.x; k < nblocks; k += blockDim.x) {
            const int c0 = blk_col[k], sz = blk_size[k];
            for (int j = 0; j < sz; ++j)
                for (int t = 0; t < BIB_T; ++t) bib_w[(c0 + j) * BIB_T + t] = bib_weight(key, k, j, t, sz);
            if (blockIdx.x == 0) {
                // G^-1 of the block (G = weights of the first sz vectors: identity + a perturbation of at most 1/8 per entry,
                // strictly diagonally dominant), once per block instead of once per (class, block) in the contraction; closed
                // forms (adjugate / determinant): no indexed local array
                double g00 = 1, g01 = 0, g02 = 0, g10 = 0, g11 = 1, g12 = 0, g20 = 0, g21 = 0, g22 = 1;
                g00 = bib_weight(key, k, 0, 0, sz);
                if (sz >= 2) {
                    g01 = bib_weight(key, k, 0, 1, sz);
                    g10 = bib_weight(key, k, 1, 0, sz);
                    g11 = bib_weight(key, k, 1, 1, sz);
                }
                if (sz >= 3) {
                    g02 = bib_weight(key, k, 0, 2, sz);
                    g12 = bib_weight(key, k, 1, 2, sz);
                    g20 = bib_weight(key, k, 2, 0, sz);
                    g21 = bib_weight(key, k, 2, 1, sz);
                    g22 = bib_weight(key, k, 2, 2, sz);
                }
                const double c00 = g11 * g22 - g12 * g21, c01 = g12 * g20 - g10 * g22, c02 = g10 * g21 - g11 * g20;
                const double idet = 1.0 / (g00 * c00 + g01 * c01 + g02 * c02);
                double* gi = ginv + (int64_t)k * 9;  // row-major 3 x 3; rows / columns >= sz belong to the identity padding, never read
                gi[0] = c00 * idet;
                gi[1] = (g02 * g21 - g01 * g22) * idet;
                gi[2] = (g01 * g12 - g02 * g11) * idet;
                gi[3] = c01 * idet;
                gi[4] = (g00 * g22 - g02 * g20) * idet;
                gi[5] = (g02 * g10 - g00 * g12) * idet;
                gi[6] = c02 * idet;
                gi[7] = (g01 * g20 - g00 * g21) * idet;
                gi[8] = (g00 * g11 - g01 * g10) * idet;
            }
        }
        __syncthreads();
    }
    if (r >= n) return;
    double x[BIB_T] = {0, 0, 0, 0};
    const double* q = Qrm + (int64_t)r * S1;
    if (only >= 0) {
        const int c0 = blk_col[only], sz = blk_size[only];
        for (int j = 0; j < sz && j < BIB_T; ++j) x[j] = q[c0 + j];
    } else {
        for (int cc = 0; cc < S1; ++cc) {
            const double v = q[cc];
#pragma unroll
            for (int t = 0; t < BIB_T; ++t) x[t] = fma(bib_w[cc * BIB_T + t], v, x[t]);
        }
    }
    X0[r] = make_double2(x[0], x[1]);
    X1[r] = make_double2(x[2], x[3]);
}
// grid (d, blocks to do); 256 threads = 4 columns x 64 row groups.  y[j][t] = q_kj' Y^(t)_i, B = Y G^-1, check, store.
__global__ void __launch_bounds__(256)
bib_contract_kernel(int n, int S1, int64_t S, const int32_t* __restrict__ blk_col, const int32_t* __restrict__ blk_size, const int64_t* __restrict__ blk_off,
                    const double* __restrict__ Qrm, const double2* __restrict__ Y0, const double2* __restrict__ Y1, const double* __restrict__ ginv,
                    uint64_t key, int only, double atol, double tol, double* __restrict__ out, uint32_t* __restrict__ flag) {
    __shared__ double red[BIB_T][256];
    __shared__ double ys[4][BIB_T];
    const int i = blockIdx.x;
    const int k = only >= 0 ? only : (int)blockIdx.y;
    const int c0 = blk_col[k], sz = blk_size[k];
    const int j = threadIdx.x & 3, g = threadIdx.x >> 2;
    double a[BIB_T] = {0, 0, 0, 0};
    if (j < sz) {
        const double2* y0 = Y0 + (int64_t)i * n;
        const double2* y1 = Y1 + (int64_t)i * n;
        // eight rows per trip, their 24 loads in flight together (one row per trip was a dependent L2 round trip each: 50 us)
        for (int r0 = g; r0 < n; r0 += 64 * 8) {
            double q[8];
            double2 u[8], w[8];
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                const int r = r0 + 64 * x;
                const int rc = r < n ? r : g;
                q[x] = r < n ? Qrm[(int64_t)rc * S1 + c0 + j] : 0.0;
                u[x] = y0[rc];
                w[x] = y1[rc];
            }
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                a[0] = fma(q[x], u[x].x, a[0]);
                a[1] = fma(q[x], u[x].y, a[1]);
                a[2] = fma(q[x], w[x].x, a[2]);
                a[3] = fma(q[x], w[x].y, a[3]);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < BIB_T; ++t) red[t][threadIdx.x] = a[t];
    __syncthreads();
    if (threadIdx.x < 16) {  // (column j, vector t): sum over the 64 row groups in a fixed order
        const int jj = threadIdx.x & 3, t = threadIdx.x >> 2;
        double v = 0;
        for (int gg = 0; gg < 64; ++gg) v += red[t][gg * 4 + jj];
        ys[jj][t] = v;
    }
    __syncthreads();
    __shared__ double Bs[3][3];
    __shared__ int s_bad;
    if (threadIdx.x == 0) s_bad = 0;
    if (threadIdx.x < 9) {
        const int a2 = threadIdx.x / 3, b2 = threadIdx.x % 3;
        double v = 0;
        if (a2 < sz && b2 < sz) {
            if (only >= 0) v = ys[a2][b2];  // Q_k'(1[P==i] q_kb): the projection formula
            else
                for (int t = 0; t < sz; ++t) v += ys[a2][t] * ginv[(int64_t)k * 9 + t * 3 + b2];  // B = Y G^-1
        }
        Bs[a2][b2] = v;
    }
    __syncthreads();
    if (only < 0 && threadIdx.x < 12) {  // the check vectors t = sz .. 3: B g^(t) must reproduce column t of Y
        const int a2 = threadIdx.x % 3, t = sz + threadIdx.x / 3;
        if (a2 < sz && t < BIB_T) {
            double v = 0;
            for (int b2 = 0; b2 < sz; ++b2) v += Bs[a2][b2] * bib_weight(key, k, b2, t, sz);
            if (!(fabs(v - ys[a2][t]) <= tol)) s_bad = 1;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && s_bad && atomicExch(&flag[1 + k], 1u) == 0u) atomicAdd(&flag[0], 1u);
    if (threadIdx.x < 9) {
        const int a2 = threadIdx.x / 3, b2 = threadIdx.x % 3;
        if (a2 < sz && b2 < sz) {
            const double v = 0.5 * (Bs[a2][b2] + Bs[b2][a2]);  // Q_k' 1[P==i] Q_k is symmetric
            out[(int64_t)i * S + blk_off[k] + a2 + b2 * sz] = (fabs(v) < atol) ? 0.0 : v;
        }
    }
}
size_t basis_image_blocks_workspace_doubles(int64_t n, int64_t d) { return (size_t)4 * n * (d + 1) + 9 * 65536; }
// Blocks up to 3 x 3.  ws: 4 n (d + 1) doubles.  only < 0: every block from four generic vectors, flag[0] = blocks whose
// check failed, flag[1 + k] = 1 for those; only = k: block k alone by the projection formula (no check, flag untouched).
// Returns false when the shape has no instance.
bool launch_basis_image_blocks(hipStream_t s, int64_t n, int64_t d, int64_t S1, int64_t S, int nblocks, const int32_t* blk_col, const int32_t* blk_size,
                               const int64_t* blk_off, const uint32_t* L, const double* Qrm, uint64_t key, int only, double atol, double tol, double* ws,
                               double* out, uint32_t* flag) {
    const int tstride = (int)((d + 1) | 1);
    const size_t per_wave = (size_t)64 * tstride * sizeof(double2);
    int W = 4;
    while (W > 1 && per_wave * W > 150 * 1024) W >>= 1;
    if (per_wave * W > 150 * 1024 || n < 1 || n > 0x7FFFFFFF / 2 || nblocks < 1 || nblocks > 65535) return false;
    double2* X0 = reinterpret_cast<double2*>(ws);
    double2* X1 = X0 + n;
    double2* Y0 = X1 + n;
    double2* Y1 = Y0 + (size_t)d * n;
    double* ginv = reinterpret_cast<double*>(Y1 + (size_t)d * n);  // 9 doubles per block
    if ((size_t)S1 * BIB_T * 8 > 48 * 1024) return false;
    bib_vectors_kernel<<<(unsigned)((n + 255) / 256), 256, (size_t)S1 * BIB_T * 8, s>>>((int)n, (int)S1, nblocks, blk_col, blk_size, key, only, Qrm, X0, X1, only < 0 ? flag : nullptr, ginv);
    int g = (int)((n + W - 1) / W);
    if (g > 256) g = 256;
    const size_t lds = per_wave * W;
    for (int pass = 0; pass < 2; ++pass) {
        const double2* X = pass ? X1 : X0;
        double2* Y = pass ? Y1 : Y0;
        if (W == 4) class_sums2_kernel<4><<<g, 256, lds, s>>>((int)n, (int)d, tstride, L, X, Y);
        else if (W == 2) class_sums2_kernel<2><<<g, 128, lds, s>>>((int)n, (int)d, tstride, L, X, Y);
        else class_sums2_kernel<1><<<g, 64, lds, s>>>((int)n, (int)d, tstride, L, X, Y);
    }
    dim3 gc((unsigned)d, only >= 0 ? 1u : (unsigned)nblocks);
    bib_contract_kernel<<<gc, 256, 0, s>>>((int)n, (int)S1, S, blk_col, blk_size, blk_off, Qrm, Y0, Y1, ginv, key, only, atol, tol, out, flag);
    return true;
}

// per-device kernel attributes, set by sdpsr_create() (see gemm_set_device_attributes)
bool blockdiag_set_device_attributes() {
    bool ok = true;
    ok &= hipSuccess == hipFuncSetAttribute(reinterpret_cast<const void*>(&class_sums2_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    ok &= hipSuccess == hipFuncSetAttribute(reinterpret_cast<const void*>(&class_sums2_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    ok &= hipSuccess == hipFuncSetAttribute(reinterpret_cast<const void*>(&class_sums2_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    ok &= hipSuccess == hipFuncSetAttribute(reinterpret_cast<const void*>(&class_sums_small_d_kernel),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    ok &= hipSuccess == hipFuncSetAttribute(reinterpret_cast<const void*>(&small_qtaq_block_norms_kernel),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 104 * 1024);
    ok &= hipSuccess == hipFuncSetAttribute(reinterpret_cast<const void*>(&small_cluster_qtaq_block_norms_kernel),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 104 * 1024);
    ok &= hipSuccess == hipFuncSetAttribute(reinterpret_cast<const void*>(&basis_image_outer_mfma_kernel),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024);
    ok &= hipSuccess == hipFuncSetAttribute(reinterpret_cast<const void*>(&basis_image_rows_kernel<1>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    ok &= hipSuccess == hipFuncSetAttribute(reinterpret_cast<const void*>(&basis_image_rows_kernel<2>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    return ok;
}

// ldo: distance between the output columns (>= n; rows >= n are not written)
bool class_sums_supports(int64_t n, int64_t d, int64_t ldo) {
    const int tstride = (int)((d + 1) | 1);
    return ldo == n || ((size_t)n + (size_t)4 * 64 * tstride) * sizeof(double) <= 150 * 1024;
}
void launch_class_sums(hipStream_t s, int64_t n, int64_t d, const uint32_t* L, const double* x, double* out, int64_t ldo) {
    const int tstride = (int)((d + 1) | 1);
    const size_t lds_small = ((size_t)n + (size_t)4 * 64 * tstride) * sizeof(double);
    if (lds_small <= 150 * 1024) {
        int g = (int)((n + 3) / 4);
        if (g > 256) g = 256;  // one resident workgroup per CU, rows in rounds
        class_sums_small_d_kernel<<<g, 256, lds_small, s>>>((int)n, (int)d, tstride, L, x, out, ldo);
        return;
    }
    const size_t lds = (size_t)2 * (d + 2) * 4 + (size_t)n * 2 + 16;
    basis_image_rows_kernel<1><<<(unsigned)n, 64, lds, s>>>((int)n, (int)d, 1, L, x, out, 0);
}

void launch_basis_image_two_stage(hipStream_t s, int64_t n, int64_t d, int64_t S1, int64_t S,
                                  const uint32_t* L, const double* Qrm, double* T, const int32_t* colA,
                                  const int32_t* colB, double atol, double* out) {
    const size_t lds = (size_t)2 * (d + 2) * 4 + (size_t)n * 2 + 16;
    if (S1 % 2 == 0)  // rows of Qrm are 16-byte aligned
        basis_image_rows_kernel<2><<<(unsigned)n, 64, lds, s>>>((int)n, (int)d, (int)S1, L, Qrm, T, 1);
    else
        basis_image_rows_kernel<1><<<(unsigned)n, 64, lds, s>>>((int)n, (int)d, (int)S1, L, Qrm, T, 1);
    dim3 g((unsigned)d, (unsigned)((S + 3) / 4));
    basis_image_blocks_kernel<<<g, 256, 0, s>>>((int)n, (int)S1, S, Qrm, T, colA, colB, atol, out);
}

// Qrm[r * S1 + k] = Qcm[r + k * n]
__global__ void transpose_to_rowmajor_kernel(int64_t n, int64_t S1, const double* __restrict__ Qcm,
                                             double* __restrict__ Qrm) {
    const int64_t total = n * S1;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const int64_t r = t / S1, k = t - r * S1;
        Qrm[t] = Qcm[r + k * n];
    }
}
void launch_transpose_to_rowmajor(hipStream_t s, int64_t n, int64_t S1, const double* Qcm,
                                  double* Qrm) {
    transpose_to_rowmajor_kernel<<<grid_for(n * S1, 256), 256, 0, s>>>(n, S1, Qcm, Qrm);
}

// ---------------------------------------------------------------------------
// entries grouped by class: _constraints(P), src/diagonalize.jl:42-50
// ---------------------------------------------------------------------------
// class_start[l] = first position of label l in the sorted key array (sorted_keys ascending)
__global__ void class_starts_kernel(int64_t len, const uint32_t* __restrict__ sorted_keys,
                                    int64_t* __restrict__ class_start) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < len; e += stride) {
        uint32_t k = sorted_keys[e];
        if (e == 0 || sorted_keys[e - 1] != k) class_start[k] = e;
    }
}

// Stable sort of (label, index) pairs by label: an LSD radix sort with 4-bit digits, written for this use (round 5; hipCUB's
// SortPairs before).  A workgroup owns 2048 CONSECUTIVE pairs, a thread 8 consecutive ones, so "earlier thread, then earlier
// register" is index order and the ranks below are stable.  Per pass: per-workgroup digit histograms (bin-major, so that one
// linear exclusive scan gives every workgroup its write offset per digit: no atomics anywhere, a deterministic order), then
// the scatter with ranks from a block-wide scan of the threads' packed digit counts.
constexpr int RX_T = 256, RX_E = 8, RX_CH = RX_T * RX_E;
__device__ __forceinline__ void rx_load(int64_t len, int64_t e0, const uint32_t* __restrict__ k, const uint32_t* __restrict__ v, uint32_t (&kk)[RX_E],
                                        uint32_t (&vv)[RX_E]) {
#pragma unroll
    for (int q = 0; q < RX_E; ++q) {
        const int64_t e = e0 + q;
        kk[q] = e < len ? k[e] : 0xFFFFFFFFu;
        vv[q] = e < len ? (v ? v[e] : (uint32_t)e) : 0u;
    }
}
__global__ void __launch_bounds__(RX_T)
rx_hist_kernel(int64_t len, const uint32_t* __restrict__ keys, int shift, uint32_t G, uint32_t* __restrict__ hist) {
    __shared__ uint32_t h[16];
    if (threadIdx.x < 16) h[threadIdx.x] = 0u;
    __syncthreads();
    const int64_t e0 = (int64_t)blockIdx.x * RX_CH + (int64_t)threadIdx.x * RX_E;
    uint32_t c[16] = {};
#pragma unroll
    for (int q = 0; q < RX_E; ++q)
        if (e0 + q < len) {
            const uint32_t dg = (keys[e0 + q] >> shift) & 15u;
#pragma unroll
            for (int b = 0; b < 16; ++b) c[b] += dg == (uint32_t)b;
        }
#pragma unroll
    for (int b = 0; b < 16; ++b)
        if (c[b]) atomicAdd(&h[b], c[b]);
    __syncthreads();
    if (threadIdx.x < 16) hist[(size_t)threadIdx.x * G + blockIdx.x] = h[threadIdx.x];
}
// exclusive scan of v[0 .. m) in place by one workgroup
__global__ void __launch_bounds__(1024)
rx_scan_kernel(int64_t m, uint32_t* __restrict__ v) {
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry_s;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0u;
    __syncthreads();
    for (int64_t base = 0; base < m; base += 8192) {
        const int64_t i0 = base + (int64_t)threadIdx.x * 8;
        uint32_t x[8], sum = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            x[q] = (i0 + q < m) ? v[i0 + q] : 0u;
            sum += x[q];
        }
        uint32_t incl = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(incl, o, 64);
            if (lane >= o) incl += y;
        }
        if (lane == 63) wsum[w] = incl;
        __syncthreads();
        uint32_t run = carry_s + incl - sum;
        for (int k = 0; k < w; ++k) run += wsum[k];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (i0 + q < m) v[i0 + q] = run;
            run += x[q];
        }
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = run;
        __syncthreads();
    }
}
__global__ void __launch_bounds__(RX_T)
rx_scatter_kernel(int64_t len, const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals, int shift, uint32_t G,
                  const uint32_t* __restrict__ offs, uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out) {
    __shared__ uint32_t wtot[RX_T / 64][8];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t e0 = (int64_t)blockIdx.x * RX_CH + (int64_t)threadIdx.x * RX_E;
    uint32_t kk[RX_E], vv[RX_E];
    rx_load(len, e0, keys, vals, kk, vv);
    // the thread's counts per digit, two 16-bit counters per word (<= 2048 per workgroup)
    uint32_t pk[8] = {};
#pragma unroll
    for (int q = 0; q < RX_E; ++q)
        if (e0 + q < len) {
            const uint32_t dg = (kk[q] >> shift) & 15u;
#pragma unroll
            for (int j = 0; j < 8; ++j) pk[j] += (dg >> 1) == (uint32_t)j ? (1u << (16 * (dg & 1u))) : 0u;
        }
    // exclusive scan over the threads, every word (both counters at once)
    uint32_t ex[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        uint32_t incl = pk[j];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(incl, o, 64);
            if (lane >= o) incl += y;
        }
        if (lane == 63) wtot[w][j] = incl;
        ex[j] = incl - pk[j];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j)
        for (int k = 0; k < w; ++k) ex[j] += wtot[k][j];
    // write: position = the workgroup's offset of the digit + earlier threads' records of that digit + earlier registers'
    uint32_t seen[8] = {};
#pragma unroll
    for (int q = 0; q < RX_E; ++q)
        if (e0 + q < len) {
            const uint32_t dg = (kk[q] >> shift) & 15u;
            uint32_t before = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if ((dg >> 1) == (uint32_t)j) {
                    before = ((ex[j] + seen[j]) >> (16 * (dg & 1u))) & 0xFFFFu;
                    seen[j] += 1u << (16 * (dg & 1u));
                }
            const uint32_t pos = offs[(size_t)dg * G + blockIdx.x] + before;
            keys_out[pos] = kk[q];
            vals_out[pos] = vv[q];
        }
}

int sort_entries_by_label(sdpsr_ctx* c, int64_t len, int64_t d, const uint32_t* L,
                          uint32_t** ent_out, int64_t** class_ptr_host) {
    hipStream_t s = c->stream;
    if (len >= (int64_t(1) << 32)) return ctx_fail(c, SDPSR_BAD_ARGUMENT, "sort_entries_by_label: len >= 2^32");
    const uint32_t G = (uint32_t)((len + RX_CH - 1) / RX_CH);
    uint32_t* kA = (uint32_t*)ctx_buf(c, "bi_key_a", len * sizeof(uint32_t));
    uint32_t* kB = (uint32_t*)ctx_buf(c, "bi_key_out", len * sizeof(uint32_t));
    uint32_t* vA = (uint32_t*)ctx_buf(c, "bi_idx_in", len * sizeof(uint32_t));
    uint32_t* vB = (uint32_t*)ctx_buf(c, "bi_idx_out", len * sizeof(uint32_t));
    uint32_t* hist = (uint32_t*)ctx_buf(c, "bi_sort_tmp", (size_t)16 * G * sizeof(uint32_t));
    int64_t* cstart = (int64_t*)ctx_buf(c, "bi_cstart", (d + 2) * sizeof(int64_t));
    if (!kA || !kB || !vA || !vB || !hist || !cstart) return SDPSR_OUT_OF_MEMORY;
    int bits = 1;
    while (((int64_t)1 << bits) <= d) ++bits;
    const int passes = (bits + 3) / 4;
    // pass p reads (kin, vin) and writes (kout, vout); the first pass reads the labels themselves with the index as the value
    const uint32_t* kin = L;
    const uint32_t* vin = nullptr;
    uint32_t* kout = (passes & 1) ? kB : kA;  // so that the last pass ends in (kB, vB)
    uint32_t* vout = (passes & 1) ? vB : vA;
    for (int p = 0; p < passes; ++p) {
        rx_hist_kernel<<<G, RX_T, 0, s>>>(len, kin, 4 * p, G, hist);
        rx_scan_kernel<<<1, 1024, 0, s>>>((int64_t)16 * G, hist);
        rx_scatter_kernel<<<G, RX_T, 0, s>>>(len, kin, vin, 4 * p, G, hist, kout, vout);
        kin = kout;
        vin = vout;
        kout = (kout == kA) ? kB : kA;
        vout = (vout == vA) ? vB : vA;
    }
    const uint32_t* key_sorted = kin;
    uint32_t* idx_sorted = const_cast<uint32_t*>(vin);
    HIP_TRY(c, hipMemsetAsync(cstart, 0xFF, (d + 2) * sizeof(int64_t), s));  // -1 = class absent
    class_starts_kernel<<<grid_for(len, 256), 256, 0, s>>>(len, key_sorted, cstart);
    int64_t* h = (int64_t*)malloc((d + 2) * sizeof(int64_t));  // handed to the caller, who frees it
    if (!h) return SDPSR_OUT_OF_MEMORY;
    const int st = d2h_sync(c, h, cstart, (size_t)(d + 1) * sizeof(int64_t));  // (a wait that yields inside a batch call)
    if (st) {
        free(h);
        return st;
    }
    // class_ptr[i] .. class_ptr[i+1] = entries of label i (i = 0..d); fill absent classes
    h[d + 1] = len;
    for (int64_t l = d; l >= 0; --l)
        if (h[l] < 0) h[l] = h[l + 1];
    *ent_out = idx_sorted;
    *class_ptr_host = h;
    return SDPSR_OK;
}

}  // namespace sdpsr
