// Partition kernels for gfx950: label gather (fill!/randomize!), projection onto L,
// signatures, and the canonical partition refinement.
//
// Reference semantics restated here:
//   fill!/randomize!      src/partitions.jl:68-75, src/abstract_part.jl:107-110
//   _clamp_round!         src/utils.jl:34-53
//   x .-= projL(x)        src/partitions.jl:161, src/utils.jl:62-66
//   Partition(M), refine! src/partitions.jl:24-35,44-66
//
// All of these are HBM-bound: one pass over n^2 entries each, 16-byte accesses per lane,
// grid sized to a few blocks per CU with grid-stride loops.
#include <cstdlib>
#include "sdpsr_internal.h"

namespace sdpsr {

static inline int grid_for(int64_t work_items, int block, int max_blocks = 256 * 8) {
    int64_t g = (work_items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > max_blocks) g = max_blocks;
    return (int)g;
}

// ---------------------------------------------------------------------------
// fill / randomize / clamp_round
// ---------------------------------------------------------------------------
__global__ void fill_f64_kernel(int64_t len, const uint32_t* __restrict__ L,
                                const double* __restrict__ values, uint32_t d, double* __restrict__ M,
                                uint32_t* __restrict__ bad_flag) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    bool bad = false;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < len; e += stride) {
        uint32_t l = L[e];
        if (l > d) {  // @assert length(values) == dim(P), src/partitions.jl:69: never read past `values`
            bad = true;
            l = 0;
        }
        M[e] = l ? values[l - 1] : 0.0;
    }
    if (bad) *bad_flag = 1u;
}

__global__ void randomize_f64_kernel(int64_t len, const uint32_t* __restrict__ L, uint64_t key,
                                     double* __restrict__ M) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < len; e += stride) {
        uint32_t l = L[e];
        M[e] = l ? sdpsr_class_uniform(key, l) : 0.0;
    }
}

__global__ void clamp_round_kernel(int64_t len, double* __restrict__ a, double atol,
                                   double scale) {
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < len; e += stride)
        a[e] = sdpsr_clamp_round(a[e], atol, scale);
}

void launch_fill_f64(hipStream_t s, int64_t len, const uint32_t* L, const double* values, int64_t d,
                     double* M, uint32_t* bad_flag) {
    fill_f64_kernel<<<grid_for(len, 256), 256, 0, s>>>(len, L, values, (uint32_t)d, M, bad_flag);
}
void launch_randomize_f64(hipStream_t s, int64_t len, const uint32_t* L, uint64_t key,
                          double* M) {
    randomize_f64_kernel<<<grid_for(len, 256), 256, 0, s>>>(len, L, key, M);
}
void launch_clamp_round(hipStream_t s, int64_t len, double* a, double atol, double scale) {
    clamp_round_kernel<<<grid_for(len, 256), 256, 0, s>>>(len, a, atol, scale);
}

// ---------------------------------------------------------------------------
// channel gathers for the square step.  One thread = 16 consecutive rows of one
// column of the padded ld x ld matrix; each channel gets one 16-byte store (int8) or
// four (f32).  Padding rows/columns are zero.
// ---------------------------------------------------------------------------
constexpr int GATHER_LUT = 2048;  // classes whose random bits are tabulated in LDS

template <int T>
__global__ void gather_i8_kernel(int64_t n, int64_t ld, const uint32_t* __restrict__ L,
                                 uint64_t key, int8_t* __restrict__ X, int dlut) {
    // dlut > 0: the labels are <= dlut <= GATHER_LUT and their 64 random bits come from an LDS
    // table (the per-entry hash costs three quarter-rate 64-bit multiplies, more than the
    // memory traffic of this kernel); the labels of a thread's 16 rows are read as four 16-byte
    // loads when the row count allows it
    __shared__ uint64_t lut[GATHER_LUT + 1];
    if (dlut > 0) {
        for (int i = threadIdx.x; i <= dlut; i += blockDim.x) lut[i] = i ? sdpsr_class_bits(key, (uint32_t)i) : 0ull;
        __syncthreads();
    }
    const bool vec = (n & 3) == 0;
    const int64_t chunks_per_col = ld / 16;
    const int64_t total = chunks_per_col * ld;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < total; c += stride) {
        const int64_t j = c / chunks_per_col;
        const int64_t i0 = (c - j * chunks_per_col) * 16;
        uint32_t out[T][4];
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int w = 0; w < 4; ++w) out[t][w] = 0;
        if (j < n) {
            uint32_t lab[16];
            if (vec && i0 + 16 <= n) {
                const uint4* p = reinterpret_cast<const uint4*>(L + i0 + j * n);
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const uint4 v = p[q4];
                    lab[4 * q4] = v.x;
                    lab[4 * q4 + 1] = v.y;
                    lab[4 * q4 + 2] = v.z;
                    lab[4 * q4 + 3] = v.w;
                }
            } else {
#pragma unroll
                for (int q = 0; q < 16; ++q) lab[q] = (i0 + q < n) ? L[i0 + q + j * n] : 0u;
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const uint32_t l = lab[q];
                const uint64_t bits = dlut > 0 ? lut[l] : (l ? sdpsr_class_bits(key, l) : 0ull);
#pragma unroll
                for (int t = 0; t < T; ++t)
                    out[t][q >> 2] |= (uint32_t)((bits >> (8 * t)) & 0xFFull) << (8 * (q & 3));
            }
        }
#pragma unroll
        for (int t = 0; t < T; ++t) {
            uint4 v = make_uint4(out[t][0], out[t][1], out[t][2], out[t][3]);
            *reinterpret_cast<uint4*>(X + (int64_t)t * ld * ld + i0 + j * ld) = v;
        }
    }
}

void launch_gather_i8(hipStream_t s, int64_t n, int64_t ld, int T, const uint32_t* L,
                      uint64_t key, int8_t* X, int64_t dmax) {
    // dmax: upper bound of the labels (0 = unknown): enables the LDS table of class bits
    const int dlut = (dmax > 0 && dmax <= GATHER_LUT) ? (int)dmax : 0;
    int g = grid_for(ld / 16 * ld, 256);
    switch (T) {
        case 1: gather_i8_kernel<1><<<g, 256, 0, s>>>(n, ld, L, key, X, dlut); break;
        case 2: gather_i8_kernel<2><<<g, 256, 0, s>>>(n, ld, L, key, X, dlut); break;
        case 3: gather_i8_kernel<3><<<g, 256, 0, s>>>(n, ld, L, key, X, dlut); break;
        case 4: gather_i8_kernel<4><<<g, 256, 0, s>>>(n, ld, L, key, X, dlut); break;
        case 5: gather_i8_kernel<5><<<g, 256, 0, s>>>(n, ld, L, key, X, dlut); break;
        case 6: gather_i8_kernel<6><<<g, 256, 0, s>>>(n, ld, L, key, X, dlut); break;
        case 7: gather_i8_kernel<7><<<g, 256, 0, s>>>(n, ld, L, key, X, dlut); break;
        default: gather_i8_kernel<8><<<g, 256, 0, s>>>(n, ld, L, key, X, dlut); break;
    }
}

// Channel gather from the PACKED lower triangle of symmetric labels (column j at offset
// j n - j (j - 1) / 2, rows j .. n-1): 64 x 64 tile pairs (I >= J).  The tile's entries are gathered
// once (coalesced along the packed columns) as 32-bit words holding the <= 4 channel bytes of the
// class, staged in LDS, and written twice: to X[I, J] and, transposed, to X[J, I], 16 rows per
// thread and channel (one 16-byte store each, bytes regrouped by v_perm).  Replaces
// unpack_symmetric_labels + gather_i8 inside the loop (the labels stay packed between the
// refinements of an iteration).  Padding rows / columns (>= n) are written as zero.
__device__ __forceinline__ void bytes_4x4_transpose(uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t o[4]) {
    // o[t] = byte t of a, b, c, d (a lowest)
    const uint32_t ab_lo = __builtin_amdgcn_perm(b, a, 0x05010400u);  // a0 b0 a1 b1
    const uint32_t ab_hi = __builtin_amdgcn_perm(b, a, 0x07030602u);  // a2 b2 a3 b3
    const uint32_t cd_lo = __builtin_amdgcn_perm(d, c, 0x05010400u);
    const uint32_t cd_hi = __builtin_amdgcn_perm(d, c, 0x07030602u);
    o[0] = __builtin_amdgcn_perm(cd_lo, ab_lo, 0x05040100u);  // a0 b0 c0 d0
    o[1] = __builtin_amdgcn_perm(cd_lo, ab_lo, 0x07060302u);
    o[2] = __builtin_amdgcn_perm(cd_hi, ab_hi, 0x05040100u);
    o[3] = __builtin_amdgcn_perm(cd_hi, ab_hi, 0x07060302u);
}

template <int T>
__global__ void __launch_bounds__(256)
gather_i8_sym_packed_kernel(int n, int64_t ld, const uint32_t* __restrict__ Lp, uint64_t key, int8_t* __restrict__ X, int dlut) {
    __shared__ uint32_t lut[GATHER_LUT + 1];
    __shared__ uint32_t tile[64][65];  // [column][row]: channel bytes of element (i0 + row, j0 + column)
    const int bi = blockIdx.x, bj = blockIdx.y;
    if (bi < bj) return;
    if (dlut > 0) {
        for (int i = threadIdx.x; i <= dlut; i += blockDim.x) lut[i] = i ? (uint32_t)sdpsr_class_bits(key, (uint32_t)i) : 0u;
        __syncthreads();
    }
    const int i0 = bi * 64, j0 = bj * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    uint32_t lab[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int i = i0 + tx, j = j0 + ty + 4 * q;
        lab[q] = 0xFFFFFFFFu;
        if (i < n && j < n) {
            const int hi = i > j ? i : j, lo = i > j ? j : i;  // (diagonal tiles: the upper half by symmetry)
            lab[q] = Lp[(int64_t)lo * n - (int64_t)lo * (lo - 1) / 2 + (hi - lo)];
        }
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const uint32_t l = lab[q];
        uint32_t v = 0u;
        if (l != 0xFFFFFFFFu) v = dlut > 0 ? lut[l] : (l ? (uint32_t)sdpsr_class_bits(key, l) : 0u);
        tile[ty + 4 * q][tx] = v;
    }
    __syncthreads();
    // 64 columns x 4 sixteen-row groups = 256 (column, group) pairs per orientation: one per thread
    const int col = threadIdx.x >> 2, seg = threadIdx.x & 3;
#pragma unroll
    for (int orient = 0; orient < 2; ++orient) {
        if (orient == 1 && bi == bj) break;
        uint32_t w[T][4];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            uint32_t e[4], o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int r = seg * 16 + g4 * 4 + k;
                e[k] = orient == 0 ? tile[col][r] : tile[r][col];  // straight: column `col`, rows r; transposed: row `col` of the tile
            }
            bytes_4x4_transpose(e[0], e[1], e[2], e[3], o);
#pragma unroll
            for (int t = 0; t < T; ++t) w[t][g4] = o[t];
        }
        // straight: X[i0 + seg*16 .., j0 + col]; transposed: X[j0 + seg*16 .., i0 + col]
        const int64_t r0 = (orient == 0 ? i0 : j0) + seg * 16, c0 = (orient == 0 ? j0 : i0) + col;
#pragma unroll
        for (int t = 0; t < T; ++t)
            *reinterpret_cast<uint4*>(X + (int64_t)t * ld * ld + r0 + c0 * ld) = make_uint4(w[t][0], w[t][1], w[t][2], w[t][3]);
    }
}
void launch_gather_i8_sym_packed(hipStream_t s, int64_t n, int64_t ld, int T, const uint32_t* Lp, uint64_t key, int8_t* X,
                                 int64_t dmax) {
    const int dlut = (dmax > 0 && dmax <= GATHER_LUT) ? (int)dmax : 0;
    const unsigned t = (unsigned)(ld / 64);  // ld is a multiple of 128
    dim3 g(t, t);
    switch (T) {
        case 1: gather_i8_sym_packed_kernel<1><<<g, 256, 0, s>>>((int)n, ld, Lp, key, X, dlut); break;
        case 2: gather_i8_sym_packed_kernel<2><<<g, 256, 0, s>>>((int)n, ld, Lp, key, X, dlut); break;
        case 4: gather_i8_sym_packed_kernel<4><<<g, 256, 0, s>>>((int)n, ld, Lp, key, X, dlut); break;
        default: break;
    }
}

__global__ void gather_f32_kernel(int64_t n, int64_t ld, int T, int vmax,
                                  const uint32_t* __restrict__ L, uint64_t key,
                                  float* __restrict__ X) {
    const int64_t chunks_per_col = ld / 4;
    const int64_t total = chunks_per_col * ld;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < total; c += stride) {
        const int64_t j = c / chunks_per_col;
        const int64_t i0 = (c - j * chunks_per_col) * 4;
        uint64_t bits[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t i = i0 + q;
            uint32_t l = (i < n && j < n) ? L[i + j * n] : 0u;
            bits[q] = l ? (sdpsr_class_bits(key, l) | (1ull << 63)) : 0ull;  // bit 63: nonzero class
        }
        for (int t = 0; t < T; ++t) {
            float4 v;
            v.x = bits[0] ? (float)sdpsr_class_small(bits[0], t, vmax) : 0.f;
            v.y = bits[1] ? (float)sdpsr_class_small(bits[1], t, vmax) : 0.f;
            v.z = bits[2] ? (float)sdpsr_class_small(bits[2], t, vmax) : 0.f;
            v.w = bits[3] ? (float)sdpsr_class_small(bits[3], t, vmax) : 0.f;
            *reinterpret_cast<float4*>(X + (int64_t)t * ld * ld + i0 + j * ld) = v;
        }
    }
}

void launch_gather_f32(hipStream_t s, int64_t n, int64_t ld, int T, int vmax, const uint32_t* L,
                       uint64_t key, float* X) {
    gather_f32_kernel<<<grid_for(ld / 4 * ld, 256), 256, 0, s>>>(n, ld, T, vmax, L, key, X);
}

__global__ void gather_f64_padded_kernel(int64_t n, int64_t ld, const uint32_t* __restrict__ L,
                                         uint64_t key, double* __restrict__ X) {
    const int64_t chunks_per_col = ld / 2;
    const int64_t total = chunks_per_col * ld;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < total; c += stride) {
        const int64_t j = c / chunks_per_col;
        const int64_t i0 = (c - j * chunks_per_col) * 2;
        double2 v;
        uint32_t l0 = (i0 < n && j < n) ? L[i0 + j * n] : 0u;
        uint32_t l1 = (i0 + 1 < n && j < n) ? L[i0 + 1 + j * n] : 0u;
        v.x = l0 ? sdpsr_class_uniform(key, l0) : 0.0;
        v.y = l1 ? sdpsr_class_uniform(key, l1) : 0.0;
        *reinterpret_cast<double2*>(X + i0 + j * ld) = v;
    }
}

void launch_gather_f64_padded(hipStream_t s, int64_t n, int64_t ld, const uint32_t* L,
                              uint64_t key, double* X) {
    gather_f64_padded_kernel<<<grid_for(ld / 2 * ld, 256), 256, 0, s>>>(n, ld, L, key, X);
}

template <typename T>
__global__ void pad_copy_kernel(int64_t n, int64_t ld, const T* __restrict__ src,
                                T* __restrict__ dst) {
    const int64_t total = ld * ld;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < total; c += stride) {
        const int64_t j = c / ld, i = c - j * ld;
        dst[c] = (i < n && j < n) ? src[i + j * n] : T(0);
    }
}
template <typename T>
__global__ void unpad_copy_kernel(int64_t n, int64_t ld, const T* __restrict__ src,
                                  T* __restrict__ dst) {
    const int64_t total = n * n;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < total; c += stride) {
        const int64_t j = c / n, i = c - j * n;
        dst[c] = src[i + j * ld];
    }
}
void launch_pad_copy(hipStream_t s, int64_t n, int64_t ld, const void* src, void* dst,
                     int elem_bytes) {
    int g = grid_for(ld * ld, 256);
    if (elem_bytes == 1)
        pad_copy_kernel<int8_t><<<g, 256, 0, s>>>(n, ld, (const int8_t*)src, (int8_t*)dst);
    else if (elem_bytes == 4)
        pad_copy_kernel<float><<<g, 256, 0, s>>>(n, ld, (const float*)src, (float*)dst);
    else
        pad_copy_kernel<double><<<g, 256, 0, s>>>(n, ld, (const double*)src, (double*)dst);
}
// dst (n x n) from the LOWER triangle of the padded src (ld x ld), mirrored: dst[i, j] = src[max(i, j), min(i, j)]
__global__ void unpad_mirror_lower_i32_kernel(int64_t n, int64_t ld, const int32_t* __restrict__ src, int32_t* __restrict__ dst) {
    const int64_t total = n * n;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < total; c += stride) {
        const int64_t j = c / n, i = c - j * n;
        dst[c] = i >= j ? src[i + j * ld] : src[j + i * ld];
    }
}
void launch_unpad_mirror_lower_i32(hipStream_t s, int64_t n, int64_t ld, const int32_t* src, int32_t* dst) {
    unpad_mirror_lower_i32_kernel<<<grid_for(n * n, 256), 256, 0, s>>>(n, ld, src, dst);
}
void launch_unpad_copy(hipStream_t s, int64_t n, int64_t ld, const void* src, void* dst,
                       int elem_bytes) {
    int g = grid_for(n * n, 256);
    if (elem_bytes == 4)
        unpad_copy_kernel<float><<<g, 256, 0, s>>>(n, ld, (const float*)src, (float*)dst);
    else
        unpad_copy_kernel<double><<<g, 256, 0, s>>>(n, ld, (const double*)src, (double*)dst);
}

// ---------------------------------------------------------------------------
// projection  y = x - U (U' x)
// ---------------------------------------------------------------------------
__device__ __forceinline__ double block_reduce_sum(double v, double* sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) sh[w] = v;
    __syncthreads();
    double r = 0;
    if (threadIdx.x == 0)
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r += sh[i];
    return r;  // valid in thread 0
}

// grid = (nblk, r).  partial[k * nblk + b]
__global__ void proj_coef_kernel(int64_t len, const double* __restrict__ U,
                                 const uint32_t* __restrict__ L, uint64_t key,
                                 const double* __restrict__ xin, double* __restrict__ partial) {
    __shared__ double sh[8];
    const int k = blockIdx.y;
    const double* Uk = U + (int64_t)k * len;
    // four independent strided streams per thread: the loads of one round are all in flight
    // before the first FMA (the kernel is bound by memory latency, not by the hash)
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    auto value = [&](int64_t e) -> double {
        if (xin) return xin[e];
        const uint32_t l = L[e];
        return l ? sdpsr_class_uniform(key, l) : 0.0;
    };
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; e + 3 * stride < len; e += 4 * stride) {
        const double u0 = Uk[e], u1 = Uk[e + stride], u2 = Uk[e + 2 * stride], u3 = Uk[e + 3 * stride];
        const double x0 = value(e), x1 = value(e + stride), x2 = value(e + 2 * stride), x3 = value(e + 3 * stride);
        a0 = fma(u0, x0, a0);
        a1 = fma(u1, x1, a1);
        a2 = fma(u2, x2, a2);
        a3 = fma(u3, x3, a3);
    }
    for (; e < len; e += stride) a0 = fma(Uk[e], value(e), a0);
    const double acc = (a0 + a1) + (a2 + a3);
    double r = block_reduce_sum(acc, sh);
    if (threadIdx.x == 0) partial[(int64_t)k * gridDim.x + blockIdx.x] = r;
}

__global__ void proj_coef_final_kernel(int nblk, const double* __restrict__ partial,
                                       double* __restrict__ coef) {
    __shared__ double sh[8];
    const int k = blockIdx.x;
    double acc = 0;
    for (int b = threadIdx.x; b < nblk; b += blockDim.x) acc += partial[(int64_t)k * nblk + b];
    double r = block_reduce_sum(acc, sh);
    if (threadIdx.x == 0) coef[k] = r;
}

void launch_proj_coef(hipStream_t s, int64_t len, int64_t r, const double* U, const uint32_t* L,
                      uint64_t key, const double* xin, double* partial, int nblk, double* coef) {
    if (r <= 0) return;
    dim3 g(nblk, (unsigned)r);
    proj_coef_kernel<<<g, 256, 0, s>>>(len, U, L, key, xin, partial);
    proj_coef_final_kernel<<<(unsigned)r, 256, 0, s>>>(nblk, partial, coef);
}

__global__ void proj_apply_kernel(int64_t len, int r, const double* __restrict__ U,
                                  const uint32_t* __restrict__ L, uint64_t key,
                                  const double* __restrict__ xin, const double* __restrict__ coef,
                                  double atol, double scale, int do_round,
                                  double* __restrict__ yout, uint64_t* __restrict__ sig) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
#pragma unroll 4
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < len; e += stride) {
        uint32_t l = L ? L[e] : 0u;
        double x;
        if (xin)
            x = xin[e];
        else
            x = l ? sdpsr_class_uniform(key, l) : 0.0;
        double p = 0;
        for (int k = 0; k < r; ++k) p = fma(U[(int64_t)k * len + e], coef[k], p);
        double y = x - p;
        // signature: the injective code of the rounded value (sdpsr_hash.h) -- no division / ldexp; raw bits when not rounding
        const uint64_t kcode = do_round ? sdpsr_round_key(y, atol, scale) : (uint64_t)__double_as_longlong(y);
        if (yout) yout[e] = do_round ? sdpsr_clamp_round(y, atol, scale) : y;
        if (sig) {
            uint64_t kb = kcode;
            uint64_t h = 0;
            if (l != 0 || kb != 0) {
                h = sdpsr_sig_mix(sdpsr_sig_start(l), kb);
                if (h == 0) h = 1;
            }
            sig[e] = h;
        }
    }
}

void launch_proj_apply(hipStream_t s, int64_t len, int64_t r, const double* U, const uint32_t* L,
                       uint64_t key, const double* xin, const double* coef, double atol,
                       double scale, int do_round, double* yout, uint64_t* sig) {
    proj_apply_kernel<<<grid_for(len, 256), 256, 0, s>>>(len, (int)r, U, L, key, xin, coef, atol,
                                                         scale, do_round, yout, sig);
}

// ---------------------------------------------------------------------------
// Projection on the lower triangle.  With symmetric labels (x symmetric) and symmetric basis
// matrices U_k the products U_k'x and the projected element need the entries i >= j only
// (off-diagonal ones count twice in the dot products): half the bytes and half the hashes of the
// projection step; the refinement then runs on the packed lower triangle like the one of the squares.
// ---------------------------------------------------------------------------
// (i, j), i >= j, of packed index e: column j starts at off(j) = j n - j (j - 1) / 2
__device__ __forceinline__ void packed_lower_ij(int n, int64_t e, uint32_t& i, uint32_t& j) {
    const float t = (float)(2 * n + 1);
    int jj = (int)((t - sqrtf(fmaxf(t * t - 8.0f * (float)e, 0.0f))) * 0.5f);
    jj = jj < 0 ? 0 : (jj > n - 1 ? n - 1 : jj);
    while ((int64_t)jj * n - (int64_t)jj * (jj - 1) / 2 > e) --jj;
    while (jj + 1 < n && (int64_t)(jj + 1) * n - (int64_t)(jj + 1) * jj / 2 <= e) ++jj;
    j = (uint32_t)jj;
    i = j + (uint32_t)(e - ((int64_t)jj * n - (int64_t)jj * (jj - 1) / 2));
}

// grid = (nblk, r): workgroups stride over the columns, threads over the rows i >= j of a column
__global__ void proj_coef_lower_kernel(int n, const double* __restrict__ U, const uint32_t* __restrict__ L, int lab_packed,
                                       uint64_t key, double* __restrict__ partial) {
    __shared__ double sh[8];
    const int k = blockIdx.y;
    const double* Uk = U + (int64_t)k * n * n;
    double a0 = 0, a1 = 0;
    for (int j = blockIdx.x; j < n; j += gridDim.x) {
        const double* Uj = Uk + (int64_t)j * n;
        // labels: the full matrix, or its packed lower triangle (column j at offset j n - j (j - 1) / 2, rows j ..)
        const uint32_t* Lj = lab_packed ? L + ((int64_t)j * n - (int64_t)j * (j - 1) / 2 - j) : L + (int64_t)j * n;
        int i = j + threadIdx.x;
        for (; i + (int)blockDim.x < n; i += 2 * blockDim.x) {
            const double u0 = Uj[i], u1 = Uj[i + blockDim.x];
            const uint32_t l0 = Lj[i], l1 = Lj[i + blockDim.x];
            const double x0 = l0 ? sdpsr_class_uniform(key, l0) : 0.0, x1 = l1 ? sdpsr_class_uniform(key, l1) : 0.0;
            a0 = fma(i == j ? u0 : 2.0 * u0, x0, a0);
            a1 = fma(2.0 * u1, x1, a1);
        }
        if (i < n) {
            const uint32_t l0 = Lj[i];
            const double u0 = Uj[i], x0 = l0 ? sdpsr_class_uniform(key, l0) : 0.0;
            a0 = fma(i == j ? u0 : 2.0 * u0, x0, a0);
        }
    }
    const double r = block_reduce_sum(a0 + a1, sh);
    if (threadIdx.x == 0) partial[(int64_t)k * gridDim.x + blockIdx.x] = r;
}
void launch_proj_coef_lower(hipStream_t s, int64_t n, int64_t r, const double* U, const uint32_t* L, int lab_packed,
                            uint64_t key, double* partial, int nblk, double* coef) {
    if (r <= 0) return;
    dim3 g(nblk, (unsigned)r);
    proj_coef_lower_kernel<<<g, 256, 0, s>>>((int)n, U, L, lab_packed, key, partial);
    proj_coef_final_kernel<<<(unsigned)r, 256, 0, s>>>(nblk, partial, coef);
}

// Symmetry probe of the basis matrices, riding on the full dot-product pass of the first
// iteration: with W[i,j] = w(i + j n) pseudo-random, <U_k, W - W'> = 0 for a symmetric U_k and a
// random number of size ~|U_k - U_k'| otherwise.  partial rows r .. 2r-1.
__global__ void proj_coef_probe_kernel(int64_t len, int n, const double* __restrict__ U, const uint32_t* __restrict__ L,
                                       uint64_t key, double* __restrict__ partial, int r) {
    __shared__ double sh[8];
    const int k = blockIdx.y;
    const double* Uk = U + (int64_t)k * len;
    double acc = 0, pa = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const uint64_t wkey = key ^ 0x5DEECE66D1CE4E5BULL;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < len; e += stride) {
        const double u = Uk[e];
        const uint32_t l = L[e];
        acc = fma(u, l ? sdpsr_class_uniform(key, l) : 0.0, acc);
        const uint32_t j = (uint32_t)e / (uint32_t)n, i = (uint32_t)e - j * (uint32_t)n;
        const uint64_t et = (uint64_t)j + (uint64_t)i * (uint32_t)n;
        const double w = (double)(sdpsr_fmix64(wkey + (uint64_t)e) >> 11) - (double)(sdpsr_fmix64(wkey + et) >> 11);
        pa = fma(u, w * (1.0 / 9007199254740992.0), pa);
    }
    double t = block_reduce_sum(acc, sh);
    if (threadIdx.x == 0) partial[(int64_t)k * gridDim.x + blockIdx.x] = t;
    __syncthreads();
    t = block_reduce_sum(pa, sh);
    if (threadIdx.x == 0) partial[(int64_t)(r + k) * gridDim.x + blockIdx.x] = t;
}
// coef[0..r) = U'x, coef[r..2r) = the symmetry probes
void launch_proj_coef_probe(hipStream_t s, int64_t len, int64_t n, int64_t r, const double* U, const uint32_t* L, uint64_t key,
                            double* partial, int nblk, double* coef) {
    if (r <= 0) return;
    dim3 g(nblk, (unsigned)r);
    proj_coef_probe_kernel<<<g, 256, 0, s>>>(len, (int)n, U, L, key, partial, (int)r);
    proj_coef_final_kernel<<<(unsigned)(2 * r), 256, 0, s>>>(nblk, partial, coef);
}

// proj_apply on the lower triangle, signatures packed (column j at offset j n - j (j - 1) / 2): the
// stand-alone form of SrcProj<R> with packed = 1 (sort path / r > 4)
__global__ void proj_apply_lower_kernel(int n, int r, const double* __restrict__ U, const uint32_t* __restrict__ L, int lab_packed,
                                        uint64_t key, const double* __restrict__ coef, double atol, double scale,
                                        uint64_t* __restrict__ sig) {
    const int64_t len = (int64_t)n * n;
    for (int j = blockIdx.x; j < n; j += gridDim.x) {
        const int64_t poff = (int64_t)j * n - (int64_t)j * (j - 1) / 2 - j;
        uint64_t* sj = sig + poff;
        for (int i = j + threadIdx.x; i < n; i += blockDim.x) {
            const int64_t e = i + (int64_t)j * n;
            const uint32_t l = lab_packed ? L[poff + i] : L[e];
            const double x = l ? sdpsr_class_uniform(key, l) : 0.0;
            double p = 0;
            for (int k = 0; k < r; ++k) p = fma(U[(int64_t)k * len + e], coef[k], p);
            const uint64_t kb = sdpsr_round_key(x - p, atol, scale);
            uint64_t h = 0;
            if (l != 0 || kb != 0) {
                h = sdpsr_sig_mix(sdpsr_sig_start(l), kb);
                if (h == 0) h = 1;
            }
            sj[i] = h;
        }
    }
}
void launch_proj_apply_lower(hipStream_t s, int64_t n, int64_t r, const double* U, const uint32_t* L, int lab_packed, uint64_t key,
                             const double* coef, double atol, double scale, uint64_t* sig) {
    const int g = (int)(n < 256 * 8 ? n : 256 * 8);
    proj_apply_lower_kernel<<<g, 256, 0, s>>>((int)n, (int)r, U, L, lab_packed, key, coef, atol, scale, sig);
}

// ---------------------------------------------------------------------------
// signatures
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint64_t finish_sig(uint32_t l, bool all_zero, uint64_t h) {
    if (l == 0 && all_zero) return 0;
    return h ? h : 1;
}

__global__ void sig_f64_kernel(int64_t len, const uint32_t* __restrict__ L,
                               const double* __restrict__ v, uint64_t* __restrict__ sig) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < len; e += stride) {
        uint32_t l = L ? L[e] : 0u;
        uint64_t kb = (uint64_t)__double_as_longlong(v[e]);
        sig[e] = finish_sig(l, kb == 0, sdpsr_sig_mix(sdpsr_sig_start(l), kb));
    }
}
void launch_sig_f64(hipStream_t s, int64_t len, const uint32_t* L, const double* v,
                    uint64_t* sig) {
    sig_f64_kernel<<<grid_for(len, 256), 256, 0, s>>>(len, L, v, sig);
}

// S = Part(a); S = refine!(S, Part(b)) in one pass: the canonical relabel of the value pairs
// (src/partitions.jl:145-146); label 0 only where both values are +0.0
__global__ void sig_f64_pair_kernel(int64_t len, const double* __restrict__ a, const double* __restrict__ b,
                                    uint64_t* __restrict__ sig) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < len; e += stride) {
        const uint64_t ka = (uint64_t)__double_as_longlong(a[e]);
        const uint64_t kb = (uint64_t)__double_as_longlong(b[e]);
        uint64_t h = sdpsr_sig_mix(sdpsr_sig_mix(sdpsr_sig_start(0u), ka), kb);
        sig[e] = finish_sig(0u, ka == 0 && kb == 0, h);
    }
}
// the pair signatures of the lower triangle, packed (column j at offset j n - j (j - 1) / 2)
__global__ void sig_f64_pair_lower_kernel(int n, const double* __restrict__ a, const double* __restrict__ b,
                                          uint64_t* __restrict__ sig) {
    for (int j = blockIdx.x; j < n; j += gridDim.x) {
        uint64_t* sj = sig + ((int64_t)j * n - (int64_t)j * (j - 1) / 2 - j);
        for (int i = j + threadIdx.x; i < n; i += blockDim.x) {
            const int64_t e = i + (int64_t)j * n;
            const uint64_t ka = (uint64_t)__double_as_longlong(a[e]);
            const uint64_t kb = (uint64_t)__double_as_longlong(b[e]);
            const uint64_t h = sdpsr_sig_mix(sdpsr_sig_mix(sdpsr_sig_start(0u), ka), kb);
            sj[i] = finish_sig(0u, ka == 0 && kb == 0, h);
        }
    }
}
void launch_sig_f64_pair(hipStream_t s, int64_t len, const double* a, const double* b, uint64_t* sig) {
    sig_f64_pair_kernel<<<grid_for(len, 256), 256, 0, s>>>(len, a, b, sig);
}

// v is a padded ld x ld matrix; output sig is dense n x n
__global__ void sig_f64_rounded_kernel(int64_t n, int64_t ld, const uint32_t* __restrict__ L,
                                       const double* __restrict__ v, double atol, double scale,
                                       uint64_t* __restrict__ sig) {
    const int64_t len = n * n;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < len; e += stride) {
        const int64_t j = e / n, i = e - j * n;
        uint32_t l = L[e];
        const uint64_t kb = sdpsr_round_key(v[i + j * ld], atol, scale);
        sig[e] = finish_sig(l, kb == 0, sdpsr_sig_mix(sdpsr_sig_start(l), kb));
    }
}
void launch_sig_f64_rounded(hipStream_t s, int64_t n, int64_t ld, const uint32_t* L,
                            const double* v, double atol, double scale, uint64_t* sig) {
    sig_f64_rounded_kernel<<<grid_for(n * n, 256), 256, 0, s>>>(n, ld, L, v, atol, scale, sig);
}

__global__ void sig_u32_kernel(int64_t len, const uint32_t* __restrict__ L,
                               const uint32_t* __restrict__ k, uint64_t* __restrict__ sig) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < len; e += stride) {
        uint32_t l = L ? L[e] : 0u;
        uint32_t kk = k[e];
        sig[e] = finish_sig(l, kk == 0, sdpsr_sig_mix(sdpsr_sig_start(l), (uint64_t)kk));
    }
}
void launch_sig_u32(hipStream_t s, int64_t len, const uint32_t* L, const uint32_t* k,
                    uint64_t* sig) {
    sig_u32_kernel<<<grid_for(len, 256), 256, 0, s>>>(len, L, k, sig);
}

// 64-bit integer keys (labels of an Int64 partition, or the hash-combined labels of several
// restarts): key 0 -> signature 0 (the zero class), any other key -> its mixed value
__global__ void sig_u64_kernel(int64_t len, const uint64_t* __restrict__ k, uint64_t* __restrict__ sig) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < len; e += stride) {
        const uint64_t kk = k[e];
        sig[e] = finish_sig(0u, kk == 0, sdpsr_sig_mix(sdpsr_sig_start(0u), kk));
    }
}
void launch_sig_u64(hipStream_t s, int64_t len, const uint64_t* k, uint64_t* sig) {
    sig_u64_kernel<<<grid_for(len, 256), 256, 0, s>>>(len, k, sig);
}

// grid (row chunks, columns): no 64-bit division per entry
template <typename CT>
__global__ void sig_channels_kernel(int64_t n, int64_t ld, int T, const uint32_t* __restrict__ L,
                                    const CT* __restrict__ C, uint64_t* __restrict__ sig,
                                    const uint32_t* __restrict__ nonsym_flag, int packed, int lab_packed) {
    const int64_t istride = (int64_t)gridDim.x * blockDim.x;
    // lower != 0: labels and products are symmetric, only entries i >= j get a signature (the
    // strict upper triangle gets the zero signature and is mirrored after the refinement; first
    // occurrences in column-major order always sit in the lower triangle, so the canonical
    // numbering is unchanged)
    const bool lower = nonsym_flag && *nonsym_flag == 0u;
    for (int64_t j = blockIdx.y; j < n; j += gridDim.y) {
        // packed != 0 (with lower): the signatures of the lower triangle are written densely,
        // column j at offset j n - j (j - 1) / 2, rows j .. n-1 -- the refinement then runs on
        // n (n + 1) / 2 entries in the same relative order; lab_packed: the labels come packed the same way
        const bool pk = lower && packed;
        const uint32_t* Lj = (pk && lab_packed) ? L + (j * n - j * (j - 1) / 2 - j) : L + j * n;
        uint64_t* sj = pk ? sig + (j * n - j * (j - 1) / 2 - j) : sig + j * n;
        const CT* Cj = C + j * ld;
#pragma unroll 2
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += istride) {
            if (lower && i < j) {
                if (!pk) sj[i] = 0ull;
                continue;
            }
            const uint32_t l = Lj[i];
            uint64_t h = sdpsr_sig_start(l);
            bool allz = true;
            for (int t = 0; t < T; t += 2) {  // two 32-bit channel values per 64-bit mixing step
                const int32_t c0 = (int32_t)Cj[(int64_t)t * ld * ld + i];  // exact: f32 channels hold integers < 2^24
                const int32_t c1 = (t + 1 < T) ? (int32_t)Cj[(int64_t)(t + 1) * ld * ld + i] : 0;
                allz = allz && (c0 == 0) && (c1 == 0);
                h = sdpsr_sig_mix(h, (uint64_t)(uint32_t)c0 | ((uint64_t)(uint32_t)c1 << 32));
            }
            sj[i] = finish_sig(l, allz, h);
        }
    }
}
static inline dim3 column_grid(int64_t n) {
    int64_t gx = (n + 1023) / 1024;  // ~4 rows per thread
    if (gx < 1) gx = 1;
    if (gx > 64) gx = 64;
    // one round of resident workgroups (8 per CU): the kernel strides over the columns
    int64_t gy = (256 * 8) / gx;
    if (gy < 1) gy = 1;
    if (gy > n) gy = n;
    return dim3((unsigned)gx, (unsigned)gy);
}
void launch_sig_i32(hipStream_t s, int64_t n, int64_t ld, int T, const uint32_t* L,
                    const int32_t* C, uint64_t* sig, const uint32_t* nonsym_flag, int packed, int lab_packed) {
    sig_channels_kernel<int32_t><<<column_grid(n), 256, 0, s>>>(n, ld, T, L, C, sig, nonsym_flag, packed, lab_packed);
}
void launch_sig_f32(hipStream_t s, int64_t n, int64_t ld, int T, const uint32_t* L,
                    const float* C, uint64_t* sig, const uint32_t* nonsym_flag, int packed, int lab_packed) {
    sig_channels_kernel<float><<<column_grid(n), 256, 0, s>>>(n, ld, T, L, C, sig, nonsym_flag, packed, lab_packed);
}

// ---------------------------------------------------------------------------
// Signature sources of the insert pass: the refinement either reads a signature array or computes
// each signature from the data the matching sig_* / proj_apply kernel would have read -- the same
// device functions, so the same bits -- and the 8 bytes per entry never travel through HBM.
// ---------------------------------------------------------------------------
// A source is either FLAT (signature of the linear index e) or walks the matrix by (row i, column j):
// the insert kernel locates the first entry of a thread's chunk once (a division, or the inversion
// of the packed lower-triangle numbering: a square root and two correction loops) and STEPS from
// one of its entries to the next (+256 rows, wrapping into the following columns) -- the per-entry
// inversion cost as much as the hash (packed passes took 100-118 us for half the entries of a
// 130 us full pass).
struct IjWalk {
    // lower: rows j .. n-1 of column j, column-major (packed lower triangle); else the full n x n square
    __device__ __forceinline__ static void locate(int n, bool lower, int64_t e, uint32_t& i, uint32_t& j) {
        if (lower) {
            packed_lower_ij(n, e, i, j);
        } else {
            j = (uint32_t)e / (uint32_t)n;
            i = (uint32_t)e - j * (uint32_t)n;
        }
    }
    __device__ __forceinline__ static void step(int n, bool lower, uint32_t& i, uint32_t& j, uint32_t by) {
        i += by;
        while (i >= (uint32_t)n && j < (uint32_t)n) {
            ++j;
            i = i - (uint32_t)n + (lower ? j : 0u);
        }
    }
};
struct SrcArray {
    static constexpr bool kIJ = false;
    static constexpr int kFetchBatch = 1;
    const uint64_t* __restrict__ sig;
    __device__ __forceinline__ bool walks() const { return false; }
    __device__ __forceinline__ bool lower() const { return false; }
    __device__ __forceinline__ int order() const { return 1; }
    __device__ __forceinline__ uint64_t at(uint32_t, uint32_t, int64_t) const { return 0; }
    __device__ __forceinline__ uint64_t operator()(int64_t e) const { return __builtin_nontemporal_load(&sig[e]); }
};
struct SrcPair {  // sig_f64_pair_kernel; packed: the lower triangle column by column (a, b symmetric)
    static constexpr bool kIJ = true;
    static constexpr int kFetchBatch = 8;  // refine_insert_kernel: entries loaded before their signatures are formed
    const double* __restrict__ a;
    const double* __restrict__ b;
    int n, packed;
    __device__ __forceinline__ bool walks() const { return packed != 0; }
    __device__ __forceinline__ bool lower() const { return true; }
    __device__ __forceinline__ int order() const { return n; }
    __device__ __forceinline__ uint64_t flat(int64_t e) const {
        const uint64_t ka = (uint64_t)__double_as_longlong(__builtin_nontemporal_load(&a[e]));
        const uint64_t kb = (uint64_t)__double_as_longlong(__builtin_nontemporal_load(&b[e]));
        const uint64_t h = sdpsr_sig_mix(sdpsr_sig_mix(sdpsr_sig_start(0u), ka), kb);
        return finish_sig(0u, ka == 0 && kb == 0, h);
    }
    // fetch / sig: the loads of an entry and the signature of what was loaded
    struct Raw {
        double a, b;
    };
    __device__ __forceinline__ Raw fetch(uint32_t i, uint32_t j, int64_t) const {
        const int64_t e = (int64_t)i + (int64_t)j * n;
        return Raw{__builtin_nontemporal_load(&a[e]), __builtin_nontemporal_load(&b[e])};
    }
    __device__ __forceinline__ uint64_t sig(const Raw& r) const {
        const uint64_t ka = (uint64_t)__double_as_longlong(r.a), kb = (uint64_t)__double_as_longlong(r.b);
        return finish_sig(0u, ka == 0 && kb == 0, sdpsr_sig_mix(sdpsr_sig_mix(sdpsr_sig_start(0u), ka), kb));
    }
    __device__ __forceinline__ uint64_t at(uint32_t i, uint32_t j, int64_t e) const { return sig(fetch(i, j, e)); }
    __device__ __forceinline__ uint64_t operator()(int64_t e) const {
        if (packed) {
            uint32_t i, j;
            packed_lower_ij(n, e, i, j);
            return at(i, j, e);
        }
        return flat(e);
    }
};
template <int R>
struct SrcProj {  // proj_apply_kernel with xin = nullptr, do_round = 1, sig only
    static constexpr bool kIJ = true;
    static constexpr int kFetchBatch = 1;
    const double* __restrict__ U;
    const uint32_t* L;  // may alias the label output of the refinement (read before the entry's own write)
    const double* __restrict__ coef;
    int64_t len;
    uint64_t key;
    double atol, scale;
    int n, packed;  // packed: the lower triangle column by column (symmetric labels and basis)
    int lab_packed;  // L is the packed lower triangle itself (label of packed entry e = L[e])
    __device__ __forceinline__ bool walks() const { return packed != 0; }
    __device__ __forceinline__ bool lower() const { return true; }
    __device__ __forceinline__ int order() const { return n; }
    __device__ __forceinline__ uint64_t flat(int64_t e) const { return flat(e, e); }
    __device__ __forceinline__ uint64_t flat(int64_t e, int64_t el) const {
        const uint32_t l = L[el];
        double u[R > 0 ? R : 1];
#pragma unroll
        for (int k = 0; k < R; ++k) u[k] = __builtin_nontemporal_load(&U[(int64_t)k * len + e]);
        const double x = l ? sdpsr_class_uniform(key, l) : 0.0;
        double p = 0;
#pragma unroll
        for (int k = 0; k < R; ++k) p = fma(u[k], coef[k], p);
        const uint64_t kb = sdpsr_round_key(x - p, atol, scale);
        uint64_t h = 0;
        if (l != 0 || kb != 0) {
            h = sdpsr_sig_mix(sdpsr_sig_start(l), kb);
            if (h == 0) h = 1;
        }
        return h;
    }
    struct Raw {
        uint32_t l;
        double u[R > 0 ? R : 1];
    };
    __device__ __forceinline__ Raw fetch(uint32_t i, uint32_t j, int64_t e) const {
        const int64_t ef = (int64_t)i + (int64_t)j * n;
        Raw r;
        r.l = L[lab_packed ? e : ef];
#pragma unroll
        for (int k = 0; k < R; ++k) r.u[k] = __builtin_nontemporal_load(&U[(int64_t)k * len + ef]);
        return r;
    }
    __device__ __forceinline__ uint64_t sig(const Raw& r) const {
        const double x = r.l ? sdpsr_class_uniform(key, r.l) : 0.0;
        double p = 0;
#pragma unroll
        for (int k = 0; k < R; ++k) p = fma(r.u[k], coef[k], p);
        const uint64_t kb = sdpsr_round_key(x - p, atol, scale);
        uint64_t h = 0;
        if (r.l != 0 || kb != 0) {
            h = sdpsr_sig_mix(sdpsr_sig_start(r.l), kb);
            if (h == 0) h = 1;
        }
        return h;
    }
    __device__ __forceinline__ uint64_t at(uint32_t i, uint32_t j, int64_t e) const { return sig(fetch(i, j, e)); }
    __device__ __forceinline__ uint64_t operator()(int64_t e) const {
        if (packed) {
            uint32_t i, j;
            packed_lower_ij(n, e, i, j);
            return at(i, j, e);
        }
        return flat(e);
    }
};
template <typename CT, int T>
struct SrcChan {  // sig_channels_kernel; packed: the lower triangle column by column
    static constexpr bool kIJ = true;
    static constexpr int kFetchBatch = T <= 2 ? 4 : 1;  // (measured at T = 2: 65 -> 58 us; all 8 with three workgroups per CU: 64 us; wider entries not measured)
    int n;
    int64_t ld;
    const uint32_t* L;
    const CT* __restrict__ C;
    int packed;
    int lab_packed;  // (with packed) L is the packed lower triangle itself
    __device__ __forceinline__ bool walks() const { return true; }  // C is ld-strided: (i, j) needed either way
    __device__ __forceinline__ bool lower() const { return packed != 0; }
    __device__ __forceinline__ int order() const { return n; }
    struct Raw {
        uint32_t l;
        CT c[T];
    };
    __device__ __forceinline__ Raw fetch(uint32_t i, uint32_t j, int64_t e) const {
        Raw r;
        r.l = lab_packed ? L[e] : L[(int64_t)j * n + i];
        const CT* Cij = C + (int64_t)j * ld + i;
#pragma unroll
        for (int t = 0; t < T; ++t) r.c[t] = __builtin_nontemporal_load(&Cij[(int64_t)t * ld * ld]);
        return r;
    }
    __device__ __forceinline__ uint64_t sig(const Raw& r) const {
        int32_t c[T + (T & 1)];
#pragma unroll
        for (int t = 0; t < T; ++t) c[t] = (int32_t)r.c[t];  // exact: f32 channels hold integers < 2^24
        if constexpr ((T & 1) != 0) c[T] = 0;
        uint64_t h = sdpsr_sig_start(r.l);
        bool allz = true;
#pragma unroll
        for (int t = 0; t < T; t += 2) {
            allz = allz && (c[t] == 0) && (c[t + 1] == 0);
            h = sdpsr_sig_mix(h, (uint64_t)(uint32_t)c[t] | ((uint64_t)(uint32_t)c[t + 1] << 32));
        }
        return finish_sig(r.l, allz, h);
    }
    __device__ __forceinline__ uint64_t at(uint32_t i, uint32_t j, int64_t e) const { return sig(fetch(i, j, e)); }
    __device__ __forceinline__ uint64_t operator()(int64_t e) const {
        uint32_t i, j;
        IjWalk::locate(n, packed != 0, e, i, j);
        return at(i, j, e);
    }
};

// Projection and square of ONE iteration refined together: signature of (old label, rounded
// projected value, channel values of the square) on the packed lower triangle.  Both random
// elements are drawn from the same partition S; the loop reaches the same fixed point as the
// reference's two refinements per iteration (a class is only ever split when it has to be), with
// one insert pass per iteration instead of two.
template <int R, int T>  // T = 2 or 4 channels
struct SrcJoint {
    static constexpr bool kIJ = true;
    static constexpr int kFetchBatch = 1;  // (2: 79 -> 99 us, 8: 114 us for SrcJoint<2, 2>)
    const double* __restrict__ U;
    const uint32_t* L;
    const double* __restrict__ coef;
    uint64_t key;
    double atol, scale;
    const int32_t* __restrict__ C;  // T channels, ld x ld each
    int64_t ld;
    int n, lab_packed;
    __device__ __forceinline__ bool walks() const { return true; }
    __device__ __forceinline__ bool lower() const { return true; }
    __device__ __forceinline__ int order() const { return n; }
    struct Raw {
        uint32_t l;
        double u[R > 0 ? R : 1];
        int32_t c[T];
    };
    __device__ __forceinline__ Raw fetch(uint32_t i, uint32_t j, int64_t e) const {
        static_assert(T == 2 || T == 4, "joint signatures: 2 or 4 channels");
        const int64_t ef = (int64_t)i + (int64_t)j * n;
        Raw r;
        r.l = lab_packed ? L[e] : L[ef];
#pragma unroll
        for (int k = 0; k < R; ++k) r.u[k] = __builtin_nontemporal_load(&U[(int64_t)k * n * n + ef]);
        const int32_t* Cij = C + (int64_t)j * ld + i;
#pragma unroll
        for (int t = 0; t < T; ++t) r.c[t] = __builtin_nontemporal_load(&Cij[(int64_t)t * ld * ld]);
        return r;
    }
    __device__ __forceinline__ uint64_t sig(const Raw& r) const {
        const double x = r.l ? sdpsr_class_uniform(key, r.l) : 0.0;
        double p = 0;
#pragma unroll
        for (int k = 0; k < R; ++k) p = fma(r.u[k], coef[k], p);
        const uint64_t kb = sdpsr_round_key(x - p, atol, scale);
        uint64_t h = sdpsr_sig_mix(sdpsr_sig_start(r.l), kb);
        bool allz = kb == 0;
#pragma unroll
        for (int t = 0; t < T; t += 2) {
            h = sdpsr_sig_mix(h, (uint64_t)(uint32_t)r.c[t] | ((uint64_t)(uint32_t)r.c[t + 1] << 32));
            allz = allz && r.c[t] == 0 && r.c[t + 1] == 0;
        }
        return finish_sig(r.l, allz, h);
    }
    __device__ __forceinline__ uint64_t at(uint32_t i, uint32_t j, int64_t e) const { return sig(fetch(i, j, e)); }
    __device__ __forceinline__ uint64_t operator()(int64_t e) const {
        uint32_t i, j;
        packed_lower_ij(n, e, i, j);
        return at(i, j, e);
    }
};

// ---------------------------------------------------------------------------
// Canonical refinement of 64-bit signatures.
//
//   pass A  every block dedups its REFINE_BLOCK entries in an LDS hash table (sig -> min
//           index), then publishes each distinct signature once to the global table
//           (write-once 64-bit CAS slots + atomicMin of the first index); the global slot
//           of every entry is parked in labels_out.
//   pass B  per block: count entries that are the first occurrence of their class.
//   scan    exclusive scan of the block counts (one block).
//   pass C  rank first occurrences inside the block -> label of the class (1-based, in
//           column-major first-occurrence order = the reference's canonical numbering).
//   pass D  labels_out[e] = label of its slot.
// ---------------------------------------------------------------------------
constexpr int REFINE_THREADS = 256;
constexpr int REFINE_PER_THREAD = 4;
constexpr int REFINE_BLOCK = REFINE_THREADS * REFINE_PER_THREAD;  // 1024 entries
constexpr uint32_t NO_SLOT = 0xFFFFFFFFu;
constexpr int MAX_PROBES = 512;
// Few-classes fast path: the first SMALL_K distinct signatures also append their global slot to
// a list (counters[LIST_OFF + i]); when the whole partition has <= SMALL_K classes one workgroup
// ranks those slots by first index and the three entry-level ranking passes exit at once.
constexpr uint32_t SMALL_K = 1024;
constexpr int LIST_OFF = 16;
// classes whose first-occurrence index the ranking passes also write to RefineWs::first_idx
constexpr uint32_t REFINE_FIRST_CAP = 65536;
uint32_t refine_first_cap() { return REFINE_FIRST_CAP; }

size_t refine_block_entries() { return REFINE_BLOCK; }
uint32_t refine_small_k() { return SMALL_K; }
size_t refine_counters_bytes() { return (size_t)(LIST_OFF + SMALL_K) * sizeof(uint32_t); }

__device__ __forceinline__ uint32_t global_find_or_insert(uint64_t sg, RefSlot* tab,
                                                          uint32_t mask, uint32_t* counters) {
    // (the overflow flag is checked once per chunk by the caller: after an overflow the host
    // repeats the pass with a larger table, nobody should keep walking a full one)
    uint32_t idx = (uint32_t)sg & mask;
    for (int probe = 0; probe < MAX_PROBES; ++probe) {
        // a probe sequence that gets long in a table that has overflowed meanwhile (the flag goes up at 75 %) stops:
        // the host repeats the pass with a larger table anyway.  (A refinement that jumps from a handful of classes to
        // ~len/2 -- partitions without symmetry -- used to walk up to MAX_PROBES slots of the full table for every
        // remaining entry: 1.0 and 1.5 ms per failed attempt at len = 524 800, configs[1].)
        if ((probe & 7) == 7 && __builtin_nontemporal_load(&counters[1])) return NO_SLOT;
        unsigned long long cur = tab[idx].sig;  // slots are write-once: a stale read can only be 0
        if (cur == sg) return idx;
        if (cur == 0ull) {
            unsigned long long old = atomicCAS(&tab[idx].sig, 0ull, (unsigned long long)sg);
            if (old == 0ull) {
                // (one atomic per wave for the lanes that are here together -- ballot, leader, prefix count -- measured: no gain where it
                // could matter, the first workgroups of the one-workgroup-per-CU kernel, and four more registers in every insert kernel:
                // SrcChan<int, 2> went from 5 to 4 resident workgroups per CU, 65 -> 81 us per refinement)
                const uint32_t cnt = atomicAdd(&counters[0], 1u);
                if (cnt < SMALL_K) counters[LIST_OFF + cnt] = idx;
                if (cnt + 1 > (mask >> 1) + (mask >> 2)) counters[1] = 1u;  // > 75% full
                return idx;
            }
            if (old == sg) return idx;
        }
        idx = (idx + 1) & mask;
    }
    counters[1] = 1u;  // overflow: host retries with a larger table
    return NO_SLOT;
}

// development aid (-DLK_TIMING and SDPSR_DEBUG): wall-clock stamps (100 MHz) of the phases of the first four chunks of the
// first / middle / last workgroup of an insert launch
#ifdef LK_TIMING
bool dbg_on();  // ctx.cpp (SDPSR_DEBUG)
__device__ long long* ri_dbg = nullptr;
#define RI_STAMP(i)                                                                                                      \
    do {                                                                                                                  \
        if (ri_dbg && threadIdx.x == 0 && ri_it < 4 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1 || blockIdx.x == gridDim.x / 2)) { \
            const int which = blockIdx.x == 0 ? 0 : (blockIdx.x == gridDim.x - 1 ? 2 : 1);                               \
            ri_dbg[(which * 4 + ri_it) * 8 + (i)] = wall_clock64();                                                       \
        }                                                                                                                 \
    } while (0)
#else
#define RI_STAMP(i)
#endif

// One workgroup dedups INSERT_CHUNK entries in an LDS table before touching the global one.
// An entry whose LDS probe sequence gets long (many distinct signatures in the chunk) goes to
// the global table directly instead.
constexpr int LDS_MAX_PROBES = 24;

// SRC: where the signatures come from (see the Src* functors); INSERT_PER_THREAD entries per
// thread and chunk (16 for the plain array, 8 for the computed sources: their loads and hashes
// of one chunk are all live before the first probe)
// (Round 4 measured a 1024-thread instance with an 8192-slot table in dynamic LDS for the mid regime of 1024 .. 4096
// classes, where the 2048-slot table thrashes and every entry goes to the L2-resident global table: 306 us against 316 us
// for the insert pass at 3000 classes, N = 4096 -- with thousands of distinct signatures the lookups are random LDS
// accesses, 4-5 lanes per bank, and cost what the L2 lookups cost; removed.  THREADS stays a parameter.)
template <class SRC, int INSERT_PER_THREAD, int LDS_SLOTS, int THREADS = REFINE_THREADS>
__global__ void __launch_bounds__(THREADS)
refine_insert_kernel(int64_t len, const SRC src,
                     uint32_t* __restrict__ slot_out, RefSlot* __restrict__ tab, uint32_t mask, uint32_t* counters) {
    constexpr int INSERT_CHUNK = THREADS * INSERT_PER_THREAD;
    // The LDS table lives across the chunks of a workgroup: a signature is published to the
    // global table only the first time the workgroup meets it (its chunks come in increasing
    // index order, so that chunk also holds the workgroup's smallest index of the class); later
    // chunks reuse the resolved global slot.  The table is reset when it gets half full.
    extern __shared__ __attribute__((aligned(16))) unsigned long long ri_dyn[];
    __shared__ unsigned long long l_sig_st[LDS_SLOTS <= 2048 ? LDS_SLOTS : 1];
    __shared__ uint32_t l_min_st[LDS_SLOTS <= 2048 ? LDS_SLOTS : 1];
    __shared__ uint32_t l_gslot_st[LDS_SLOTS <= 2048 ? LDS_SLOTS : 1];
    unsigned long long* l_sig = LDS_SLOTS <= 2048 ? l_sig_st : ri_dyn;
    uint32_t* l_min = LDS_SLOTS <= 2048 ? l_min_st : reinterpret_cast<uint32_t*>(ri_dyn + LDS_SLOTS);
    uint32_t* l_gslot = LDS_SLOTS <= 2048 ? l_gslot_st : reinterpret_cast<uint32_t*>(ri_dyn + LDS_SLOTS) + LDS_SLOTS;
    __shared__ uint32_t l_count, l_overflow, l_new;
    constexpr uint32_t PENDING = 0xFFFFFFFEu;
    const int64_t nchunk = (len + INSERT_CHUNK - 1) / INSERT_CHUNK;
    bool need_clear = true, bypass = false;
    // (Round 3 tried issuing the loads of a workgroup's NEXT chunk before hashing and probing the current one -- PMC
    // showed two thirds of the wave cycles in s_waitcnt on the chunk's own loads.  The prefetched registers took the
    // joint source from 71 to 172 VGPRs (2 waves per SIMD instead of 7) and the pass got SLOWER at every grid size
    // tried, 1-6 workgroups per CU: refine phase of theta_c32xk128 0.91-1.17 ms against 0.79.  Latency is hidden by
    // resident workgroups here, not by software pipelining; removed.
    // Two more forms measured and removed in round 3 (both bit-exact on the golden partitions): per-old-class candidate
    // lists in LDS compared word by word, so that only NEW classes are hashed (joint pass 94 us against 80, with 8 or 4
    // candidates per class: the extra LDS reads cost more than the nine 64-bit multiplies they replace); and the waves
    // of a workgroup running free inside a chunk -- creator lanes resolve their own slots, other waves spin on the LDS
    // word, one barrier per chunk instead of three -- 80.3 us against 79.6: the barriers are not what the waves wait for.)
    int ri_it = -1;
    (void)ri_it;
    for (int64_t blk = blockIdx.x; blk < nchunk; blk += gridDim.x) {
        ++ri_it;
        RI_STAMP(0);
        if (need_clear) {
            for (int i = threadIdx.x; i < LDS_SLOTS; i += THREADS) {
                l_sig[i] = 0ull;
                l_min[i] = 0xFFFFFFFFu;
            }
            if (threadIdx.x == 0) l_count = 0;
        }
        if (threadIdx.x == 0) {
            l_overflow = __builtin_nontemporal_load(&counters[1]);
            l_new = 0;
        }
        __syncthreads();
        RI_STAMP(1);
        if (l_overflow) return;  // uniform: the host repeats the pass with a larger table
        const int64_t base = blk * INSERT_CHUNK;
        if (bypass) {
            // many classes (the previous chunk half filled the LDS table on its own): the LDS level only costs probes,
            // every entry goes to the global table directly.  The table is write-once, so the usual case -- the signature
            // already sits in its home slot -- is a plain read: eight entries' home slots are read together, then their
            // minima (round 5; entry by entry this was two dependent L2 round trips per entry, 340 us at 3000 classes and
            // 16.7 M entries).  Whatever the read does not settle (an empty or foreign slot) takes the probing path.
            constexpr int BB = INSERT_PER_THREAD < 8 ? INSERT_PER_THREAD : 8;
#pragma unroll
            for (int q0 = 0; q0 < INSERT_PER_THREAD; q0 += BB) {
                uint64_t sb[BB];
                unsigned long long cur[BB];
                uint32_t outs[BB], mn[BB];
#pragma unroll
                for (int q = 0; q < BB; ++q) {
                    const int64_t e = base + (q0 + q) * THREADS + threadIdx.x;
                    sb[q] = (e < len) ? src(e) : 0ull;
                }
#pragma unroll
                for (int q = 0; q < BB; ++q) {  // the home slot, signature and minimum in one 16-byte gather
                    const uint4 rec = sb[q] ? *reinterpret_cast<const uint4*>(&tab[(uint32_t)sb[q] & mask]) : make_uint4(0u, 0u, 0u, 0u);
                    cur[q] = (unsigned long long)rec.x | ((unsigned long long)rec.y << 32);
                    mn[q] = rec.z;
                }
#pragma unroll
                for (int q = 0; q < BB; ++q) {
                    outs[q] = NO_SLOT;
                    if (sb[q]) {
                        if (cur[q] == sb[q]) {
                            outs[q] = (uint32_t)sb[q] & mask;
                        } else {
                            outs[q] = global_find_or_insert(sb[q], tab, mask, counters);
                            mn[q] = outs[q] != NO_SLOT ? tab[outs[q]].min : 0u;
                        }
                    }
                }
#pragma unroll
                for (int q = 0; q < BB; ++q) {
                    const int64_t e = base + (q0 + q) * THREADS + threadIdx.x;
                    if (outs[q] != NO_SLOT && mn[q] > (uint32_t)e) atomicMin(&tab[outs[q]].min, (uint32_t)e);
                    if (e < len) slot_out[e] = outs[q];
                }
            }
            continue;
        }
        // >= 0: LDS slot; -1: zero signature; <= -2: already resolved global slot (-2 - g)
        int myslot[INSERT_PER_THREAD];
        // all 16 signature loads are issued before the first probe: the probe loop contains LDS
        // atomics, which the compiler will not move global loads across
        uint64_t sgs[INSERT_PER_THREAD];
        bool walked = false;
        if constexpr (SRC::kIJ) {
            if (src.walks()) {  // uniform
                walked = true;
                const int nn = src.order();
                const bool low = src.lower();
                uint32_t wi = 0, wj = 0;
                if (base + threadIdx.x < len) IjWalk::locate(nn, low, base + threadIdx.x, wi, wj);
                // SRC::kFetchBatch entries are loaded before their signatures are formed.  `sgs[q] = sig(fetch(..))` entry by entry,
                // with the walk's column-wrap loop between two entries, compiles to load - wait - hash per entry: eight dependent
                // memory round trips per chunk (round 4, profiles/r04_insert_phases.txt: 9-10 us of a 14 us chunk for the pair
                // source, 4.5-7 of 10 for the channel source).  The batch is bounded by registers: all 8 entries at once take
                // the channel source from 71 to 138 registers and the joint source to 199 (2-3 resident workgroups per CU instead
                // of 5) and make them slower, 65 -> 73 and 82 -> 114 us.  Kept where it measured faster: the pair source with all 8
                // (71 -> 54 us, four workgroups per CU), the channel source with 4 (65 -> 58 us); the joint source stays at 1.
                constexpr int FB = SRC::kFetchBatch;
                static_assert(INSERT_PER_THREAD % FB == 0, "");
#pragma unroll
                for (int q0 = 0; q0 < INSERT_PER_THREAD; q0 += FB) {
                    typename SRC::Raw raws[FB];
#pragma unroll
                    for (int q = 0; q < FB; ++q) {
                        const int64_t e = base + (q0 + q) * THREADS + threadIdx.x;
                        if (e < len) raws[q] = src.fetch(wi, wj, e);
                        IjWalk::step(nn, low, wi, wj, THREADS);
                    }
#pragma unroll
                    for (int q = 0; q < FB; ++q) {
                        const int64_t e = base + (q0 + q) * THREADS + threadIdx.x;
                        sgs[q0 + q] = (e < len) ? src.sig(raws[q]) : 0ull;
                    }
                }
            }
        }
        if (!walked) {
#pragma unroll
            for (int q = 0; q < INSERT_PER_THREAD; ++q) {
                const int64_t e = base + q * THREADS + threadIdx.x;
                sgs[q] = (e < len) ? src(e) : 0ull;
            }
        }
        RI_STAMP(2);
#pragma unroll
        for (int q = 0; q < INSERT_PER_THREAD; ++q) {
            const int64_t e = base + q * THREADS + threadIdx.x;
            const uint64_t sg = sgs[q];
            myslot[q] = -1;
            if (sg) {
                uint32_t idx = (uint32_t)(sg >> 40) & (LDS_SLOTS - 1);
                int probes = 0;
                int placed = 0;  // 1: found, 2: inserted by this thread
                while (probes < LDS_MAX_PROBES) {
                    unsigned long long cur = l_sig[idx];
                    if (cur == sg) {
                        placed = 1;
                        break;
                    }
                    if (cur == 0ull) {
                        unsigned long long old = atomicCAS(&l_sig[idx], 0ull, (unsigned long long)sg);
                        if (old == 0ull) {
                            placed = 2;
                            break;
                        }
                        if (old == sg) {
                            placed = 1;
                            break;
                        }
                    }
                    idx = (idx + 1) & (LDS_SLOTS - 1);
                    ++probes;
                }
                if (placed) {
                    if (placed == 2) {
                        l_gslot[idx] = PENDING;
                        atomicAdd(&l_count, 1u);
                        l_new = 1u;
                    }
                    // l_min only decreases: a plain read that is already <= e makes the atomic a
                    // no-op (true for every entry after the first of its class in this thread's
                    // increasing index order, i.e. almost always when classes are few)
                    if (l_min[idx] > (uint32_t)e) atomicMin(&l_min[idx], (uint32_t)e);
                    myslot[q] = (int)idx;
                } else {
                    const uint32_t g = global_find_or_insert(sg, tab, mask, counters);
                    if (g != NO_SLOT) {
                        if (tab[g].min > (uint32_t)e) atomicMin(&tab[g].min, (uint32_t)e);
                        myslot[q] = -2 - (int)g;
                    }
                }
            }
        }
        RI_STAMP(3);
        __syncthreads();
        RI_STAMP(4);
        // publish the signatures this workgroup has not resolved yet (few classes: nothing new after the first chunks,
        // the scan of the table and its barrier are skipped; l_new is uniform after the barrier above)
        if (l_new) {
        for (int i = threadIdx.x; i < LDS_SLOTS; i += THREADS) {
            if (l_sig[i] != 0ull && l_gslot[i] == PENDING) {
                const uint32_t g = global_find_or_insert(l_sig[i], tab, mask, counters);
                // tab_min only ever decreases, so a (possibly stale) plain read that is already
                // <= our candidate proves the atomic cannot change anything: skip it.
                if (g != NO_SLOT) {
                    const uint32_t mine = l_min[i];
                    if (tab[g].min > mine) atomicMin(&tab[g].min, mine);
                }
                l_gslot[i] = g;
            }
        }
        __syncthreads();
        }
#pragma unroll
        for (int q = 0; q < INSERT_PER_THREAD; ++q) {
            const int64_t e = base + q * THREADS + threadIdx.x;
            if (e < len) {
                uint32_t out = NO_SLOT;
                if (myslot[q] >= 0) out = l_gslot[myslot[q]];
                else if (myslot[q] <= -2) out = (uint32_t)(-2 - myslot[q]);
                slot_out[e] = out;
            }
        }
        RI_STAMP(5);
        need_clear = l_count > LDS_SLOTS / 2;  // uniform (read after the barrier above)
        bypass = need_clear;
        __syncthreads();
        RI_STAMP(6);
    }
}

// ---------------------------------------------------------------------------
// The insert pass for the MID regime: ~500 .. 5000 classes.  There the 2048-slot LDS table of the kernel above
// thrashes and every entry becomes a 16-byte gather from the L2-resident global table: one L1 miss per lane, ~5-9 clocks
// per entry and CU, 270-300 us for 16.7 M entries (round 5, profiles/r05_refine_warm_3000_classes_kernel_stats.csv).
// Here ONE workgroup per CU keeps every signature it has met in 128 KB of LDS (MID_SLOTS x {8-byte signature, 4-byte
// word, 4-byte minimum}); a lookup is two LDS reads issued together, eight entries at a time, and the waves run FREE:
// no barrier after the table is cleared.
// The lane whose compare-and-swap puts a signature into the LDS table publishes it: global find-or-insert, minimum, then
// the slot's word = (ordinal of the workgroup's chunk << 20) | global slot.  An entry that finds the word still
// pending is set aside and looked at again behind the chunk's other entries (a bounded number of times; asking the global
// table itself is what thousands of lanes of one large new class would do at the same moment).  The minimum: the workgroup's chunks come in
// increasing index order, so an entry of chunk j cannot lower the minimum of a class published in an OLDER chunk (word
// ordinal < j) and takes its slot at once -- the steady state.  Otherwise (same or newer chunk: the waves drift apart)
// it lowers the slot's LDS minimum first and goes to the global minimum only if it did -- the LDS minimum only ever holds
// indices whose owners do go there.  A table that fills up (probe sequences beyond MID_MAX_PROBES) sends the extra
// signatures to the global table entry by entry.
// A short first launch (PER = 1: sixteen workgroups, one entry per thread, publishing behind a barrier) goes over the first
// 16 384 entries; the full launch preloads its LDS tables with what that published (launch_insert_mid).
// Measured and not kept (3000 classes, N = 4096, stamps below): a barrier per chunk with the new signatures published from a
// list (143 us against 133: every wave sits through every other wave's latencies); the work of a chunk ordered by kind --
// all home slots, then each lane walking only ITS entries that need probing, then the global home slots of all entries
// that need them in one batch -- with the entry picked by select chains (180 us: one lane in five is not in its home
// slot and the probing step, 6-12 us of a chunk, is where the time goes either way).
// ---------------------------------------------------------------------------
constexpr int MID_THREADS = 1024;
constexpr int MID_SLOTS = 8192;
constexpr int MID_PER = 8;
constexpr int MID_MAX_PROBES = 16;
constexpr int MID_FIRST_WGS = 16;  // workgroups that go first, 1024 entries each
constexpr uint32_t MID_PENDING = 0xFFFFFFFFu;
constexpr size_t MID_LDS_BYTES = (size_t)MID_SLOTS * 16;
constexpr int MID_MAX_LOG2CAP = 20;  // the word's low 20 bits; ordinals < 2^11 (len < 2^31, >= 1024-entry chunks, 256 workgroups)

__device__ __forceinline__ uint32_t mid_direct(uint64_t sg, uint32_t e, RefSlot* tab, uint32_t mask, uint32_t* counters) {
    const uint32_t g = global_find_or_insert(sg, tab, mask, counters);
    if (g != NO_SLOT && tab[g].min > e) atomicMin(&tab[g].min, e);
    return g;
}

// What a thread holds of an entry between its loads and its signature: the signature itself (array source) or the raw
// operands of a computed source (SRC::Raw -- the loads of the next chunk stay in flight as raw words, the hash runs when
// the chunk's turn comes).
template <class SRC> struct MidTraits {
    using Raw = typename SRC::Raw;
    static __device__ __forceinline__ uint64_t sig(const SRC& s, const Raw& r) { return s.sig(r); }
};
template <> struct MidTraits<SrcArray> {
    using Raw = uint64_t;
    static __device__ __forceinline__ uint64_t sig(const SrcArray&, const uint64_t& r) { return r; }
};
// the loads of the chunk at `base`, all unconditional (an entry beyond the end reads entry 0 and is masked later)
template <class SRC, int PER>
__device__ __forceinline__ void mid_fetch(const SRC& src, int64_t len, int64_t base, typename MidTraits<SRC>::Raw (&r)[PER]) {
    if constexpr (!SRC::kIJ) {
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int64_t e = base + q * MID_THREADS + threadIdx.x;
            r[q] = src(e < len ? e : len - 1);
        }
    } else {
        const bool walk = src.walks();  // uniform
        const int nn = src.order();
        const bool low = src.lower();
        uint32_t wi = 0, wj = 0;
        if (walk) {
            const int64_t e0 = base + threadIdx.x;
            IjWalk::locate(nn, low, e0 < len ? e0 : len - 1, wi, wj);
        }
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int64_t e = base + q * MID_THREADS + threadIdx.x;
            const bool valid = e < len;
            if (walk) {
                r[q] = src.fetch(valid ? wi : 0u, valid ? wj : 0u, valid ? e : 0);
                IjWalk::step(nn, low, wi, wj, MID_THREADS);
            } else {
                const int64_t ec = valid ? e : 0;
                r[q] = src.fetch((uint32_t)ec, 0u, ec);  // (flat: entry e of every operand)
            }
        }
    }
}

#ifdef LK_TIMING
#define MID_STAMP(i)                                                                                                        \
    do {                                                                                                                     \
        if (ri_dbg && (threadIdx.x == 0 || threadIdx.x == 512) && ord < 10 && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2)) \
            ri_dbg[(((blockIdx.x ? 1 : 0) * 2 + (threadIdx.x ? 1 : 0)) * 10 + ord) * 8 + (i)] = wall_clock64();                \
    } while (0)
#else
#define MID_STAMP(i)
#endif

// PER: entries per thread and chunk (8; 1 for the few workgroups that go first over the first 8192 entries)
template <class SRC, int PER>
__global__ void __launch_bounds__(MID_THREADS)
refine_insert_mid_kernel(int64_t len, const SRC src, uint32_t* __restrict__ slot_out, RefSlot* __restrict__ tab, uint32_t mask,
                         uint32_t* counters, int preload) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long mid_dyn[];
    unsigned long long* l_sig = mid_dyn;
    uint32_t* l_word = reinterpret_cast<uint32_t*>(mid_dyn + MID_SLOTS);
    uint32_t* l_min = l_word + MID_SLOTS;
    __shared__ uint32_t l_full, l_stop;
    constexpr int CHUNK = MID_THREADS * PER;
    const int64_t nchunk = (len + CHUNK - 1) / CHUNK;
    for (int i = threadIdx.x; i < MID_SLOTS; i += MID_THREADS) {
        l_sig[i] = 0ull;
        l_word[i] = MID_PENDING;
        l_min[i] = 0xFFFFFFFFu;
    }
    if (threadIdx.x == 0) {
        l_full = 0;
        l_stop = 0;
    }
    __syncthreads();
    uint32_t ord = 0;  // ordinal of this workgroup's chunk
    if (preload) {
        // The workgroups that went first (launch below) have published the classes of the first entries of the array, with
        // their final minima: no later index can lower them.  Every workgroup of the full launch starts with those in its LDS
        // table, as if met in a chunk before its first -- instead of meeting them one by one in its first chunk, eight steps of
        // two or three dependent L2 round trips each (48 us of the 122 us this kernel took at 3000 classes).
        for (uint32_t g = threadIdx.x; g <= mask; g += MID_THREADS) {
            const uint4 rec = *reinterpret_cast<const uint4*>(&tab[g]);
            const unsigned long long sg = (unsigned long long)rec.x | ((unsigned long long)rec.y << 32);
            if (!sg) continue;
            uint32_t idx = __umulhi((uint32_t)(sg >> 32), (uint32_t)MID_SLOTS);
            for (int probes = 0; probes < MID_MAX_PROBES; ++probes) {
                if (atomicCAS(&l_sig[idx], 0ull, sg) == 0ull) {  // (the global table holds a signature once)
                    l_word[idx] = g;  // ordinal 0
                    break;
                }
                idx = (idx + 1) & (MID_SLOTS - 1);
            }
        }
        if (threadIdx.x == 0 && counters[1]) l_stop = 1u;  // the table was too small already for the first entries: the host repeats the pass
        __syncthreads();
        if (l_stop) return;
        ord = 1;
    }
    // The signatures of the NEXT chunk are in flight while this one is looked up.  The loads are unconditional (index
    // clamped): a load under a branch makes the number of outstanding loads unknown to the compiler, and every wait
    // becomes a wait for all of them -- the prefetch included.
    // (A source whose raw entries of a chunk do not fit 32 registers loads them when the chunk's turn comes: the joint source's
    // 56 prefetched registers made SrcJoint<2, 2> spill, 107 us per launch.)
    using RawT = typename MidTraits<SRC>::Raw;
    constexpr bool PF = sizeof(RawT) * PER <= 128;
    RawT nxt[PER];
    if constexpr (PF) mid_fetch<SRC, PER>(src, len, (int64_t)blockIdx.x * CHUNK, nxt);
    for (int64_t blk = blockIdx.x; blk < nchunk; blk += gridDim.x, ++ord) {
        const int64_t base = blk * CHUNK;
        uint64_t sgs[PER];
        uint32_t home[PER];  // LDS slot: the home slot first, the slot found or taken after the probing step
        unsigned long long cur[PER];
        uint32_t word[PER];
        MID_STAMP(0);
        if constexpr (!PF) mid_fetch<SRC, PER>(src, len, base, nxt);
#pragma unroll
        for (int q = 0; q < PER; ++q) sgs[q] = (base + q * MID_THREADS + threadIdx.x < len) ? MidTraits<SRC>::sig(src, nxt[q]) : 0ull;
        if constexpr (PF) mid_fetch<SRC, PER>(src, len, (blk + gridDim.x) * CHUNK, nxt);
        // (the overflow flag: thread 0 reads it for the workgroup -- 4096 waves asking one address every chunk queue up there)
        if (threadIdx.x == 0 && counters[1]) l_stop = 1u;
        // the home slots of all entries, both words, before anything depends on them
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            home[q] = __umulhi((uint32_t)(sgs[q] >> 32), (uint32_t)MID_SLOTS);
            cur[q] = l_sig[home[q]];
            word[q] = l_word[home[q]];
        }
        const bool full = l_full != 0;
        const bool overflow = l_stop != 0;
#ifdef LK_TIMING
        if (cur[0] == 1ull) continue;  // (never: the stamp below waits for the LDS reads)
#endif
        MID_STAMP(1);
        uint32_t deferred = 0, lowered = 0;  // bit q: entry q waits for another lane to publish its signature / has lowered the slot's LDS minimum
        if constexpr (PER == 1) {
            // The workgroups that go first: one chunk each (gridDim.x = nchunk), every signature new to the workgroup, and with
            // few classes a thousand lanes meet the same one at the same moment.  Barrier-free, every lane that finds the word
            // pending asks the global table itself: 16 384 lanes on a handful of addresses (99 us at 3 classes).  Here the lanes
            // only leave their index in the slot's LDS minimum; after a barrier the lane that inserted the signature publishes
            // it with that minimum; after another everybody reads the word.
            const int64_t e = base + threadIdx.x;
            const uint64_t sg = sgs[0];
            uint32_t idx = home[0];
            int placed = (sg && cur[0] == sg) ? 1 : 0;
            if (sg && !placed) {
                for (int probes = 0; probes < MID_MAX_PROBES; ++probes) {
                    const unsigned long long c = l_sig[idx];
                    if (c == sg) {
                        placed = 1;
                        break;
                    }
                    if (c == 0ull) {
                        const unsigned long long old = atomicCAS(&l_sig[idx], 0ull, (unsigned long long)sg);
                        if (old == 0ull) {
                            placed = 2;
                            break;
                        }
                        if (old == sg) {
                            placed = 1;
                            break;
                        }
                    }
                    idx = (idx + 1) & (MID_SLOTS - 1);
                }
            }
            if (placed && l_min[idx] > (uint32_t)e) atomicMin(&l_min[idx], (uint32_t)e);
            __syncthreads();
            uint32_t out = NO_SLOT;
            if (placed == 2) {
                out = global_find_or_insert(sg, tab, mask, counters);
                if (out != NO_SLOT) {
                    const uint32_t mine = l_min[idx];
                    if (tab[out].min > mine) atomicMin(&tab[out].min, mine);
                    l_word[idx] = (ord << 20) | out;
                }
            } else if (sg && !placed) {
                out = mid_direct(sg, (uint32_t)e, tab, mask, counters);  // (a probe sequence beyond MID_MAX_PROBES)
            }
            __syncthreads();
            if (placed == 1) {
                const uint32_t w = l_word[idx];
                out = w != MID_PENDING ? (w & 0xFFFFFu) : NO_SLOT;  // (pending still: the global table has overflowed)
            }
            if (e < len) slot_out[e] = out;
            (void)full;
            (void)word;
            (void)overflow;
            return;
        }
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int64_t e = base + q * MID_THREADS + threadIdx.x;
            const uint64_t sg = sgs[q];
            uint32_t out = NO_SLOT;
            if (sg) {
                if (cur[q] == sg && (word[q] >> 20) < ord) {  // published in an older chunk: the steady state (a pending word is all ones)
                    out = word[q] & 0xFFFFFu;
                } else {
                    // probe (the table is insert-only, signatures are write-once)
                    uint32_t idx = home[q];
                    int placed = cur[q] == sg ? 1 : 0;  // 1: found, 2: inserted here
                    if (!placed) {
                        for (int probes = 0; probes < MID_MAX_PROBES; ++probes) {
                            const unsigned long long c = l_sig[idx];
                            if (c == sg) {
                                placed = 1;
                                break;
                            }
                            if (c == 0ull) {
                                if (full) break;  // (no new signatures into a table whose probe sequences have got long)
                                const unsigned long long old = atomicCAS(&l_sig[idx], 0ull, (unsigned long long)sg);
                                if (old == 0ull) {
                                    placed = 2;
                                    break;
                                }
                                if (old == sg) {
                                    placed = 1;
                                    break;
                                }
                            }
                            idx = (idx + 1) & (MID_SLOTS - 1);
                        }
                    }
                    if (!placed) {
                        if (!full) l_full = 1u;
                        out = mid_direct(sg, (uint32_t)e, tab, mask, counters);
                    } else {
                        const uint32_t w = placed == 2 ? MID_PENDING : l_word[idx];
                        if (w != MID_PENDING && (w >> 20) < ord) {
                            out = w & 0xFFFFFu;
                        } else {
                            // same or newer chunk, or not published yet: this entry may hold the class's smallest index
                            const bool lowers = l_min[idx] > (uint32_t)e && atomicMin(&l_min[idx], (uint32_t)e) > (uint32_t)e;
                            if (w != MID_PENDING) {
                                out = w & 0xFFFFFu;
                                if (lowers && tab[out].min > (uint32_t)e) atomicMin(&tab[out].min, (uint32_t)e);
                            } else if (placed == 2) {
                                out = global_find_or_insert(sg, tab, mask, counters);
                                if (out != NO_SLOT) {
                                    if (lowers && tab[out].min > (uint32_t)e) atomicMin(&tab[out].min, (uint32_t)e);
                                    l_word[idx] = (ord << 20) | out;
                                }
                            } else {
                                // Somebody else is publishing this signature right now.  Asking the global table too is what
                                // every lane of a LARGE new class would do at the same moment (a class that is absent from the
                                // entries the first workgroups saw -- ER(7) x K_72 is not vertex-transitive -- met by thousands of
                                // lanes of every workgroup in its first chunk: 90 us instead of 45).  The entry waits its turn
                                // after the chunk's other entries; the word is there by then.
                                home[q] = idx;
                                deferred |= 1u << q;
                                if (lowers) lowered |= 1u << q;
                                continue;
                            }
                        }
                    }
                }
            }
            if (e < len) slot_out[e] = out;
        }
        for (int tries = 0; deferred && tries < 64; ++tries) {
#pragma unroll
            for (int q = 0; q < PER; ++q) {
                if (!((deferred >> q) & 1u)) continue;
                const uint32_t w = l_word[home[q]];
                if (w == MID_PENDING) continue;
                const uint32_t e = (uint32_t)(base + q * MID_THREADS + threadIdx.x);
                const uint32_t out = w & 0xFFFFFu;
                if (((lowered >> q) & 1u) && tab[out].min > e) atomicMin(&tab[out].min, e);
                slot_out[e] = out;
                deferred &= ~(1u << q);
            }
            if (deferred) __builtin_amdgcn_s_sleep(16);
        }
        if (deferred) {  // (the publisher got no slot: the global table has overflowed, the host repeats the pass)
#pragma unroll
            for (int q = 0; q < PER; ++q) {
                if (!((deferred >> q) & 1u)) continue;
                const uint32_t e = (uint32_t)(base + q * MID_THREADS + threadIdx.x);
                const uint32_t out = global_find_or_insert(sgs[q], tab, mask, counters);
                if (out != NO_SLOT && ((lowered >> q) & 1u) && tab[out].min > e) atomicMin(&tab[out].min, e);
                slot_out[e] = out;
            }
        }
        MID_STAMP(2);
        if (overflow) return;  // the host repeats the pass with a larger table (each wave leaves on its own: no barriers in this loop)
    }
}

// thread t of the block owns entries base + 4t .. 4t+3 (index order matters here)
__device__ __forceinline__ int first_flags(int64_t len, int64_t base,
                                           const uint32_t* __restrict__ slot,
                                           const RefSlot* __restrict__ tab, int* flags,
                                           uint32_t* slots) {
    int cnt = 0;
    const int64_t e0 = base + (int64_t)threadIdx.x * REFINE_PER_THREAD;
#pragma unroll
    for (int q = 0; q < REFINE_PER_THREAD; ++q) {
        const int64_t e = e0 + q;
        uint32_t sl = (e < len) ? slot[e] : NO_SLOT;
        slots[q] = sl;
        int f = 0;
        if (sl != NO_SLOT) f = (tab[sl].min == (uint32_t)e);
        flags[q] = f;
        cnt += f;
    }
    return cnt;
}

__global__ void __launch_bounds__(REFINE_THREADS)
refine_count_kernel(int64_t len, const uint32_t* __restrict__ slot,
                    const RefSlot* __restrict__ tab, uint32_t* __restrict__ blk_cnt,
                    const uint32_t* __restrict__ counters) {
    __shared__ int sh[REFINE_THREADS / 64];
    if (counters[0] <= SMALL_K) return;  // ranked by refine_small_rank_kernel
    const int64_t nblk = (len + REFINE_BLOCK - 1) / REFINE_BLOCK;
    for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        int flags[REFINE_PER_THREAD];
        uint32_t slots[REFINE_PER_THREAD];
        int cnt = first_flags(len, blk * REFINE_BLOCK, slot, tab, flags, slots);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o, 64);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = cnt;
        __syncthreads();
        if (threadIdx.x == 0) {
            int t = 0;
            for (int w = 0; w < REFINE_THREADS / 64; ++w) t += sh[w];
            blk_cnt[blk] = (uint32_t)t;
        }
        __syncthreads();
    }
}

// exclusive scan of blk_cnt[0..nblk) in place; counters[2] = total
__global__ void __launch_bounds__(1024)
refine_scan_kernel(int64_t nblk, uint32_t* __restrict__ blk_cnt, uint32_t* counters) {
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry;
    if (counters[0] <= SMALL_K) return;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int64_t base = 0; base < nblk; base += 1024) {
        const int64_t i = base + threadIdx.x;
        uint32_t v = (i < nblk) ? blk_cnt[i] : 0u;
        uint32_t x = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            uint32_t y = __shfl_up(x, o, 64);
            if (lane >= o) x += y;
        }
        if (lane == 63) wsum[w] = x;
        __syncthreads();
        uint32_t woff = 0;
        for (int k = 0; k < w; ++k) woff += wsum[k];
        uint32_t incl = x + woff + carry;
        if (i < nblk) blk_cnt[i] = incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) counters[2] = carry;
}

__global__ void __launch_bounds__(REFINE_THREADS)
refine_rank_kernel(int64_t len, const uint32_t* __restrict__ slot,
                   const RefSlot* __restrict__ tab, const uint32_t* __restrict__ blk_off,
                   uint32_t* __restrict__ tab_lab, const uint32_t* __restrict__ counters, uint32_t* __restrict__ first_idx) {
    __shared__ int wsum[REFINE_THREADS / 64];
    if (counters[0] <= SMALL_K) return;
    const int64_t nblk = (len + REFINE_BLOCK - 1) / REFINE_BLOCK;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        int flags[REFINE_PER_THREAD];
        uint32_t slots[REFINE_PER_THREAD];
        int cnt = first_flags(len, blk * REFINE_BLOCK, slot, tab, flags, slots);
        int x = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            int y = __shfl_up(x, o, 64);
            if (lane >= o) x += y;
        }
        if (lane == 63) wsum[w] = x;
        __syncthreads();
        int woff = 0;
        for (int k = 0; k < w; ++k) woff += wsum[k];
        int excl = x - cnt + woff;
        uint32_t lab = blk_off[blk] + (uint32_t)excl;
#pragma unroll
        for (int q = 0; q < REFINE_PER_THREAD; ++q)
            if (flags[q]) {
                tab_lab[slots[q]] = ++lab;
                // first occurrence of class `lab` (its class representative for verify_* below)
                if (first_idx && lab <= REFINE_FIRST_CAP) first_idx[lab - 1] = (uint32_t)(blk * REFINE_BLOCK + (int64_t)threadIdx.x * REFINE_PER_THREAD + q);
            }
        __syncthreads();
    }
}

// <= SMALL_K classes: label of a class = 1 + number of classes with a smaller first index
__global__ void __launch_bounds__(1024)
refine_small_rank_kernel(const RefSlot* __restrict__ tab, uint32_t* __restrict__ tab_lab,
                         uint32_t* __restrict__ counters, uint32_t* __restrict__ first_idx) {
    __shared__ uint32_t s_min[SMALL_K];
    const uint32_t K = counters[0];
    if (K > SMALL_K) return;
    const uint32_t i = threadIdx.x;
    uint32_t slot = 0, mine = 0;
    if (i < K) {
        slot = counters[LIST_OFF + i];
        mine = tab[slot].min;
        s_min[i] = mine;
    }
    __syncthreads();
    if (i < K) {
        uint32_t rank = 0;
        for (uint32_t j = 0; j < K; ++j) rank += (s_min[j] < mine);
        tab_lab[slot] = rank + 1;
        if (first_idx) first_idx[rank] = mine;  // K <= SMALL_K <= REFINE_FIRST_CAP
    }
    if (i == 0) counters[2] = K;
}

// tab_sig = 0, tab_min = 0xFFFFFFFF, counters[0..16) = 0 in one launch
__global__ void refine_clear_kernel(int64_t cap, RefSlot* __restrict__ tab, uint32_t* __restrict__ counters) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t i = t0; i < cap; i += stride) {
        *reinterpret_cast<uint4*>(&tab[i]) = make_uint4(0u, 0u, 0xFFFFFFFFu, 0u);
    }
    if (t0 < LIST_OFF) counters[t0] = 0u;
}

// A pass the host is going to repeat (table overflow, or more classes than the one-workgroup
// ranking was launched for) must leave labels_out alone: a computed signature source may read the
// old labels from that very array.  slot may be labels_out itself (array source: in place).
__global__ void refine_label_kernel(int64_t len, const uint32_t* slot, uint32_t* labels_out,
                                    const uint32_t* __restrict__ tab_lab, const uint32_t* __restrict__ counters,
                                    int expect_small, uint32_t* __restrict__ host_counters, uint32_t host_seq) {
    // the last kernel of a refinement hands the counters (inserted, overflow flag, classes) to the host itself: a
    // store into pinned host memory instead of a separate 16-byte copy launch (~5 us + a launch gap per refinement)
    // ... and then the stamp: the host may go on as soon as it sees it (ctx_wait_word), the rest of this pass is behind it in the stream
    if (host_counters && blockIdx.x == 0 && threadIdx.x < 64) {
        if (threadIdx.x < 3) host_counters[threadIdx.x] = counters[threadIdx.x];
        __threadfence_system();
        if (threadIdx.x == 0) host_counters[3] = host_seq;
    }
    if (counters[1] || (expect_small && counters[0] > SMALL_K)) return;
    // four consecutive entries per thread and trip: one 16-byte load, four gathers in flight together, one 16-byte store (one
    // entry per trip was a load, a wait, a gather, a wait and a store, sixteen times per thread; round 4)
    typedef uint32_t rl_u32x4 __attribute__((ext_vector_type(4)));
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // (16-byte accesses only when both arrays allow them: the labels may be a caller's device pointer)
    const int64_t len4 = ((reinterpret_cast<uintptr_t>(slot) | reinterpret_cast<uintptr_t>(labels_out)) & 15) ? 0 : (len >> 2);
    const rl_u32x4* slot4 = reinterpret_cast<const rl_u32x4*>(slot);
    rl_u32x4* out4 = reinterpret_cast<rl_u32x4*>(labels_out);
    for (int64_t q = t0; q < len4; q += stride) {
        const rl_u32x4 sl = __builtin_nontemporal_load(&slot4[q]);
        rl_u32x4 o;
        o.x = (sl.x == NO_SLOT) ? 0u : tab_lab[sl.x];
        o.y = (sl.y == NO_SLOT) ? 0u : tab_lab[sl.y];
        o.z = (sl.z == NO_SLOT) ? 0u : tab_lab[sl.z];
        o.w = (sl.w == NO_SLOT) ? 0u : tab_lab[sl.w];
        out4[q] = o;
    }
    for (int64_t e = 4 * len4 + t0; e < len; e += stride) {
        const uint32_t sl = __builtin_nontemporal_load(&slot[e]);
        labels_out[e] = (sl == NO_SLOT) ? 0u : tab_lab[sl];
    }
}

// Tile pass shared by refine_label_sym_kernel and copy_check_symmetric_kernel: out = map(in) over an
// n x n column-major matrix, 64 x 64 tile pairs (I, J), I >= J, with the mirror tile compared
// through LDS.  16-byte accesses (4 rows per lane) when n % 4 == 0.  Returns true if some
// out[r, c] != out[c, r] in this workgroup's tiles.
template <bool VEC4, class MAP>
__device__ __forceinline__ bool sym_tile_pass(int64_t n, const uint32_t* in, uint32_t* out, const MAP& map,
                                              uint32_t (*tile)[65]) {
    const int64_t i0 = (int64_t)blockIdx.x * 64, j0 = (int64_t)blockIdx.y * 64;
    bool bad = false;
    if (VEC4) {
        const int vr = threadIdx.x & 15, cb = threadIdx.x >> 4;  // 16 row groups x 16 columns per round
        uint4 m[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {  // mirror tile: rows j0.., columns i0..
            const int64_t r = j0 + 4 * vr, cc = i0 + cb + 16 * q;
            m[q] = make_uint4(0u, 0u, 0u, 0u);
            if (r < n && cc < n) m[q] = *reinterpret_cast<const uint4*>(in + r + cc * n);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t r = j0 + 4 * vr, cc = i0 + cb + 16 * q;
            uint4 l = make_uint4(map(m[q].x), map(m[q].y), map(m[q].z), map(m[q].w));
            if (r < n && cc < n && i0 != j0) *reinterpret_cast<uint4*>(out + r + cc * n) = l;
            const int c = cb + 16 * q;
            tile[4 * vr + 0][c] = l.x;
            tile[4 * vr + 1][c] = l.y;
            tile[4 * vr + 2][c] = l.z;
            tile[4 * vr + 3][c] = l.w;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) m[q] = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (int q = 0; q < 4; ++q) {  // tile: rows i0.., columns j0..
            const int64_t r = i0 + 4 * vr, cc = j0 + cb + 16 * q;
            if (r < n && cc < n) m[q] = *reinterpret_cast<const uint4*>(in + r + cc * n);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t r = i0 + 4 * vr, cc = j0 + cb + 16 * q;
            if (r < n && cc < n) {
                const uint4 l = make_uint4(map(m[q].x), map(m[q].y), map(m[q].z), map(m[q].w));
                *reinterpret_cast<uint4*>(out + r + cc * n) = l;
                const int c = cb + 16 * q;  // out[r + k, cc] against out[cc, r + k] = mirror tile (row c, column 4 vr + k)
                bad = bad || l.x != tile[c][4 * vr + 0] || l.y != tile[c][4 * vr + 1] || l.z != tile[c][4 * vr + 2] ||
                      l.w != tile[c][4 * vr + 3];
            }
        }
    } else {
        const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
#pragma unroll 4
        for (int q = 0; q < 16; ++q) {
            const int64_t r = j0 + tx, cc = i0 + ty + 4 * q;
            uint32_t l = 0u;
            if (r < n && cc < n) {
                l = map(in[r + cc * n]);
                if (i0 != j0) out[r + cc * n] = l;
            }
            tile[tx][ty + 4 * q] = l;
        }
        __syncthreads();
#pragma unroll 4
        for (int q = 0; q < 16; ++q) {
            const int64_t r = i0 + tx, cc = j0 + ty + 4 * q;
            if (r < n && cc < n) {
                const uint32_t l = map(in[r + cc * n]);
                out[r + cc * n] = l;
                if (l != tile[ty + 4 * q][tx]) bad = true;
            }
        }
    }
    return bad;
}

struct MapSlotLabel {
    const uint32_t* __restrict__ tab_lab;
    __device__ __forceinline__ uint32_t operator()(uint32_t sl) const { return (sl == NO_SLOT) ? 0u : tab_lab[sl]; }
};
struct MapIdentity {
    __device__ __forceinline__ uint32_t operator()(uint32_t v) const { return v; }
};

// refine_label_kernel for an n x n label matrix, with the symmetry check of the NEW labels folded in
// (check_symmetric_kernel would read them again).  counters[3] = 1 if the labels are NOT symmetric
// (counters are cleared by refine_clear_kernel).
template <bool VEC4>
__global__ void __launch_bounds__(256)
refine_label_sym_kernel(int64_t n, const uint32_t* slot, uint32_t* labels_out,
                        const uint32_t* __restrict__ tab_lab, uint32_t* __restrict__ counters, int expect_small) {
    __shared__ uint32_t tile[64][65];
    if (counters[1] || (expect_small && counters[0] > SMALL_K)) return;
    if (blockIdx.x < blockIdx.y) return;  // lower triangle of tile pairs
    if (sym_tile_pass<VEC4>(n, slot, labels_out, MapSlotLabel{tab_lab}, tile)) counters[3] = 1u;
}

// Which sources have the one-workgroup-per-CU kernel (each costs two more instantiations of a large kernel; the ones the
// loop of admissible_subspace runs on the benchmark shapes and the array source)
template <class SRC> struct MidSource { static constexpr bool value = false; };
template <> struct MidSource<SrcArray> { static constexpr bool value = true; };
template <> struct MidSource<SrcPair> { static constexpr bool value = true; };
template <> struct MidSource<SrcChan<int32_t, 2>> { static constexpr bool value = true; };
// (Measured and not kept: the joint source.  SrcJoint<2, 2> with its raw entries prefetched spills (107 us per launch at
// N = 4104), without the prefetch it takes 78 - 88 + 8 us against 71 + 18 for the 256-thread kernel.  The channel source gains
// only with the deferral in the kernel: ER(7) x K_72 is not vertex-transitive, and its large classes that the first
// workgroups never saw took the full launch from 47 to 90 us.)

template <class SRC>
bool mid_set_attributes_kind() {
    bool ok = true;
    ok &= hipSuccess == hipFuncSetAttribute(reinterpret_cast<const void*>(&refine_insert_mid_kernel<SRC, MID_PER>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)MID_LDS_BYTES);
    ok &= hipSuccess == hipFuncSetAttribute(reinterpret_cast<const void*>(&refine_insert_mid_kernel<SRC, 1>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)MID_LDS_BYTES);
    return ok;
}

template <class SRC>
static void launch_insert_mid(hipStream_t s, int64_t len, const SRC& src, uint32_t* slot, const RefineWs& ws, size_t cap) {
    const int64_t nchunk = (len + MID_THREADS * MID_PER - 1) / (MID_THREADS * MID_PER);
    // A few workgroups first, over the first entries (as in launch_insert, and for the same reason: 256 workgroups publishing
    // the same signatures at the same moment are compare-and-swaps and minimum atomics on one address each; the first 16 384
    // entries hold all but a dozen of 3000 classes, with their smallest indices).  Sixteen workgroups with one entry per
    // thread: one round of probing and publishing each.  The full launch preloads its LDS tables with what they published.
    const bool first = nchunk > 4 && !(ws.mid & 2);
    if (first)
        refine_insert_mid_kernel<SRC, 1><<<MID_FIRST_WGS, MID_THREADS, MID_LDS_BYTES, s>>>((int64_t)MID_THREADS * MID_FIRST_WGS, src, slot, ws.tab,
                                                                                       (uint32_t)(cap - 1), ws.counters, 0);
#ifdef LK_TIMING
    long long* dbg = nullptr;
    if (dbg_on()) {
        hipMalloc(&dbg, 4 * 10 * 8 * 8);
        hipMemset(dbg, 0, 4 * 10 * 8 * 8);
        hipStreamSynchronize(s);
        hipMemcpyToSymbol(HIP_SYMBOL(ri_dbg), &dbg, sizeof(dbg));
    }
#endif
    refine_insert_mid_kernel<SRC, MID_PER><<<(unsigned)(nchunk < 256 ? nchunk : 256), MID_THREADS, MID_LDS_BYTES, s>>>(
        len, src, slot, ws.tab, (uint32_t)(cap - 1), ws.counters, first ? 1 : 0);
#ifdef LK_TIMING
    if (dbg) {
        hipStreamSynchronize(s);
        long long h[4 * 10 * 8];
        hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost);
        for (int w = 0; w < 4; ++w)
            for (int o = 0; o < 10; ++o) {
                const long long* t = &h[(w * 10 + o) * 8];
                if (!t[0]) continue;
                fprintf(stderr, "[mid timing] wg=%d wave=%d chunk=%d: start %8.2f us, home slots read +%6.2f, chunk done +%6.2f\n", w / 2, (w % 2) * 8, o,
                        (t[0] - h[0]) * 0.01, (t[1] - t[0]) * 0.01, (t[2] - t[0]) * 0.01);
            }
        long long* z = nullptr;
        hipMemcpyToSymbol(HIP_SYMBOL(ri_dbg), &z, sizeof(z));
        hipFree(dbg);
    }
#endif
}

// SLOTS: entries of the workgroup's LDS table (16 bytes each): 2048 -> four workgroups per CU,
// 1024 -> up to eight (the computed sources are bound by their hash arithmetic and by the
// barriers between the phases of a chunk: more resident workgroups overlap those phases)
template <class SRC, int PER, int SLOTS = 1024>
static void launch_insert(hipStream_t s, int g_chunks_cap, int64_t len, const SRC& src, uint32_t* slot, const RefineWs& ws,
                          size_t cap) {
    if constexpr (MidSource<SRC>::value) {
        if (ws.mid) {
            launch_insert_mid<SRC>(s, len, src, slot, ws, cap);
            return;
        }
    }
    const int64_t nchunk = (len + REFINE_THREADS * PER - 1) / (REFINE_THREADS * PER);
    // resident workgroups per CU, measured 2..8: 5 is the minimum of a flat curve (sdpsr_opts.insert_wgs_per_cu overrides)
    // (a fetch-first source holds its raw entries in registers: 126 of them, four workgroups per CU are resident)
    if (SLOTS == 1024) g_chunks_cap = 256 * (ws.insert_wgs_per_cu > 0 ? ws.insert_wgs_per_cu : (SRC::kFetchBatch >= 8 ? 4 : 5));
    const int g = (int)(nchunk < g_chunks_cap ? nchunk : g_chunks_cap);
#ifdef LK_TIMING
    long long* dbg = nullptr;
    if (dbg_on()) {
        hipMalloc(&dbg, 3 * 4 * 8 * 8);
        hipMemset(dbg, 0, 3 * 4 * 8 * 8);
        hipMemcpyToSymbol(HIP_SYMBOL(ri_dbg), &dbg, sizeof(dbg));
    }
#endif
    // Few classes: every workgroup meets the same few dozen signatures in its first chunk and publishes them to the global
    // table at the same moment -- ~1000 compare-and-swaps and minimum atomics per signature on one address each, 12-20 us
    // of the first chunk (why the pair source took 78 us at 34 classes and 55 us at 3).  ONE workgroup
    // goes first over the first chunk: it publishes what that chunk holds with their smallest indices, and the full
    // launch then finds those signatures by plain reads (slots are write-once) and minima it cannot lower.
    if (ws.log2cap <= 12 && nchunk > 4) {
        const int64_t first = (int64_t)REFINE_THREADS * PER;
        refine_insert_kernel<SRC, PER, SLOTS><<<1, REFINE_THREADS, 0, s>>>(first < len ? first : len, src, slot, ws.tab,
                                                                    (uint32_t)(cap - 1), ws.counters);
    }
    refine_insert_kernel<SRC, PER, SLOTS><<<g, REFINE_THREADS, 0, s>>>(len, src, slot, ws.tab,
                                                                (uint32_t)(cap - 1), ws.counters);
#ifdef LK_TIMING
    if (dbg) {
        hipStreamSynchronize(s);
        long long h[3 * 4 * 8];
        hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost);
        for (int w = 0; w < 3; ++w)
            for (int it = 0; it < 4; ++it) {
                const long long* t = &h[(w * 4 + it) * 8];
                if (!t[0]) continue;
                fprintf(stderr, "[insert timing] %s grid=%d len=%lld wg=%d chunk=%d:", __PRETTY_FUNCTION__ + 40, g, (long long)len, w, it);
                for (int i = 1; i < 7; ++i) fprintf(stderr, " %6.2f", t[i] ? (t[i] - t[0]) * 0.01 : -1.0);
                fprintf(stderr, "  (start rel. to wg0 chunk0 %.2f)\n", (t[0] - h[0]) * 0.01);
            }
        long long* z = nullptr;
        hipMemcpyToSymbol(HIP_SYMBOL(ri_dbg), &z, sizeof(z));
        hipFree(dbg);
    }
#endif
}

template <typename CT>
static bool launch_insert_chan(hipStream_t s, int gcap, int64_t len, const SigSource& q, uint32_t* slot, const RefineWs& ws,
                               size_t cap) {
    const CT* C = (const CT*)q.C;
    switch (q.T) {
        case 1: launch_insert<SrcChan<CT, 1>, 8>(s, gcap, len, SrcChan<CT, 1>{(int)q.n, q.ld, q.L, C, q.packed, q.lab_packed}, slot, ws, cap); return true;
        case 2: launch_insert<SrcChan<CT, 2>, 8>(s, gcap, len, SrcChan<CT, 2>{(int)q.n, q.ld, q.L, C, q.packed, q.lab_packed}, slot, ws, cap); return true;
        case 4: launch_insert<SrcChan<CT, 4>, 8>(s, gcap, len, SrcChan<CT, 4>{(int)q.n, q.ld, q.L, C, q.packed, q.lab_packed}, slot, ws, cap); return true;
        case 8: launch_insert<SrcChan<CT, 8>, 4>(s, gcap, len, SrcChan<CT, 8>{(int)q.n, q.ld, q.L, C, q.packed, q.lab_packed}, slot, ws, cap); return true;
        default: return false;
    }
}

// stand-alone form of SrcJoint (sort path); T = 2 or 4 channels
__global__ void sig_joint_lower_kernel(int n, int64_t ld, int r, int T, const double* __restrict__ U, const uint32_t* __restrict__ L,
                                       int lab_packed, uint64_t key, const double* __restrict__ coef, double atol, double scale,
                                       const int32_t* __restrict__ C, uint64_t* __restrict__ sig) {
    const int64_t len = (int64_t)n * n;
    for (int j = blockIdx.x; j < n; j += gridDim.x) {
        const int64_t poff = (int64_t)j * n - (int64_t)j * (j - 1) / 2 - j;
        for (int i = j + threadIdx.x; i < n; i += blockDim.x) {
            const int64_t e = i + (int64_t)j * n;
            const uint32_t l = lab_packed ? L[poff + i] : L[e];
            const double x = l ? sdpsr_class_uniform(key, l) : 0.0;
            double p = 0;
            for (int k = 0; k < r; ++k) p = fma(U[(int64_t)k * len + e], coef[k], p);
            const uint64_t kb = sdpsr_round_key(x - p, atol, scale);
            const int32_t* Cij = C + (int64_t)j * ld + i;
            uint64_t h = sdpsr_sig_mix(sdpsr_sig_start(l), kb);
            bool allz = kb == 0;
            for (int t = 0; t < T; t += 2) {
                const int32_t c0 = Cij[(int64_t)t * ld * ld], c1 = Cij[(int64_t)(t + 1) * ld * ld];
                h = sdpsr_sig_mix(h, (uint64_t)(uint32_t)c0 | ((uint64_t)(uint32_t)c1 << 32));
                allz = allz && c0 == 0 && c1 == 0;
            }
            sig[poff + i] = finish_sig(l, allz, h);
        }
    }
}

// ---------------------------------------------------------------------------
// "Does this step split any class?" without a relabel.  A refinement that ends with the dimension it
// started with (every confirm round; the first iteration on an already closed partition) leaves the
// labels as they are, so the insert / rank / label passes only establish that nothing changed.  That
// question needs no hash table: with the first-occurrence index of every class at hand (first_idx,
// written by the ranking pass of the refinement that made the labels) a tiny kernel evaluates the
// step's values at the class representatives, and one streaming pass compares every entry with the
// representative of its class -- raw values (rounded projection, channel products), no signature
// hashing, no atomics; flag[0] = 1 as soon as some entry differs (the caller then runs the full
// refinement).  Packed lower triangle, int32 channels (the default loop).
// ---------------------------------------------------------------------------
struct VerifyRef {  // one per class: the step's values at the class representative
    double x;        // the class's uniform draw (depends on the label only)
    uint64_t ybits;  // code of the rounded projected value there (sdpsr_round_key)
    int32_t c[4];    // channel products there
};
template <int R, int T, bool JOINT>
__global__ void verify_ref_kernel(int n, int64_t ld, int d, const uint32_t* __restrict__ first_idx, const double* __restrict__ U,
                                  uint64_t key, const double* __restrict__ coef, double atol, double scale,
                                  const int32_t* __restrict__ C, VerifyRef* __restrict__ ref, uint32_t* __restrict__ flag) {
    const int cls = blockIdx.x * blockDim.x + threadIdx.x;
    if (cls == 0) flag[0] = 0u;  // the verdict of the compare pass that follows in stream order
    if (cls >= d) return;
    uint32_t i, j;
    packed_lower_ij(n, (int64_t)first_idx[cls], i, j);
    const int64_t ef = (int64_t)i + (int64_t)j * n;
    VerifyRef r;
    r.x = 0;
    r.ybits = 0;
    if (JOINT) {
        r.x = sdpsr_class_uniform(key, (uint32_t)cls + 1u);
        double p = 0;
#pragma unroll
        for (int k = 0; k < R; ++k) p = fma(U[(int64_t)k * n * n + ef], coef[k], p);
        r.ybits = sdpsr_round_key(r.x - p, atol, scale);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) r.c[t] = t < T ? C[(int64_t)t * ld * ld + (int64_t)j * ld + i] : 0;
    ref[cls] = r;
}
template <int R, int T, bool JOINT>
__global__ void __launch_bounds__(256)
verify_lower_kernel(int n, int64_t ld, const uint32_t* __restrict__ Lp, const double* __restrict__ U, const double* __restrict__ coef,
                    double atol, double scale, const int32_t* __restrict__ C, const VerifyRef* __restrict__ ref,
                    uint32_t* __restrict__ flag) {
    bool bad = false;
    double cf[R > 0 ? R : 1];
#pragma unroll
    for (int k = 0; k < R; ++k) cf[k] = coef[k];
    const int64_t nn = (int64_t)n * n;
    for (int j = blockIdx.x; j < n; j += gridDim.x) {
        const int64_t poff = (int64_t)j * n - (int64_t)j * (j - 1) / 2 - j;
        const uint32_t* Lj = Lp + poff;
        const int32_t* Cj = C + (int64_t)j * ld;
        const double* Uj = U + (int64_t)j * n;
        // VB entries of a thread per trip: their labels and values in one batch of loads, then the representatives of their
        // classes in a second one (entry by entry it was two dependent memory round trips per entry, round 4)
        constexpr int VB = 4;
        for (int i0 = j + threadIdx.x; i0 < n; i0 += 256 * VB) {
            uint32_t l[VB];
            int32_t c[VB][T];
            double u[VB][R > 0 ? R : 1];
#pragma unroll
            for (int b = 0; b < VB; ++b) {
                const int i = i0 + 256 * b;
                const bool ok = i < n;
                l[b] = ok ? __builtin_nontemporal_load(&Lj[i]) : 0u;
#pragma unroll
                for (int t = 0; t < T; ++t) c[b][t] = ok ? __builtin_nontemporal_load(&Cj[(int64_t)t * ld * ld + i]) : 0;
                if (JOINT) {
#pragma unroll
                    for (int k = 0; k < R; ++k) u[b][k] = ok ? __builtin_nontemporal_load(&Uj[(int64_t)k * nn + i]) : 0.0;
                }
            }
            VerifyRef r[VB];
#pragma unroll
            for (int b = 0; b < VB; ++b) {
                r[b].x = 0;
                r[b].ybits = 0;
                r[b].c[0] = r[b].c[1] = r[b].c[2] = r[b].c[3] = 0;
                if (l[b]) r[b] = ref[l[b] - 1];  // label 0 (and the slots past the column): the zero class stays together only while every value is zero
            }
#pragma unroll
            for (int b = 0; b < VB; ++b) {
                if (JOINT) {
                    double p = 0;
#pragma unroll
                    for (int k = 0; k < R; ++k) p = fma(u[b][k], cf[k], p);
                    const uint64_t yb = sdpsr_round_key(r[b].x - p, atol, scale);
                    bad = bad || yb != r[b].ybits;
                }
#pragma unroll
                for (int t = 0; t < T; ++t) bad = bad || c[b][t] != r[b].c[t];
            }
        }
    }
    if (bad) flag[0] = 1u;
}

template <int R, int T, bool JOINT>
static void launch_verify_rt(hipStream_t s, const SigSource& q, int64_t d, const uint32_t* first_idx, void* ref, uint32_t* flag) {
    const int n = (int)q.n;
    verify_ref_kernel<R, T, JOINT><<<(unsigned)((d + 255) / 256), 256, 0, s>>>(n, q.ld, (int)d, first_idx, q.U, q.key, q.coef, q.atol, q.scale,
                                                                             (const int32_t*)q.C, (VerifyRef*)ref, flag);
    const int g = n < 256 * 8 ? n : 256 * 8;
    verify_lower_kernel<R, T, JOINT><<<g, 256, 0, s>>>(n, q.ld, q.L, q.U, q.coef, q.atol, q.scale, (const int32_t*)q.C,
                                                      (const VerifyRef*)ref, flag);
}
size_t verify_ref_bytes(int64_t d) { return (size_t)(d > 0 ? d : 1) * sizeof(VerifyRef); }

// "Can the projection step still split a class?"  x - U U'x of an x that is constant on the classes of S is constant on
// them for EVERY x exactly when every basis matrix U_k is (U_k in span(S) => U U'x in span(S)), and a finer S keeps the
// property: once it holds, the projection half of the loop (src/partitions.jl:159-164) cannot refine S any more.  The
// check compares the rounded codes of U_k (the rounding of the projection's own signatures, sdpsr_round_key) at every
// entry of the packed lower triangle with those at the class representative (first_idx, as the verify pass); label 0
// (structural zeros) needs U_k = 0.  flag[0] = 1 <=> some U_k is NOT constant on some class.
template <int R>
__global__ void uconst_ref_kernel(int n, int d, const uint32_t* __restrict__ first_idx, const double* __restrict__ U, double atol, double scale,
                                  uint64_t* __restrict__ ref, uint32_t* __restrict__ flag) {
    const int cls = blockIdx.x * blockDim.x + threadIdx.x;
    if (cls == 0) flag[0] = 0u;
    if (cls >= d) return;
    uint32_t i, j;
    packed_lower_ij(n, (int64_t)first_idx[cls], i, j);
    const int64_t ef = (int64_t)i + (int64_t)j * n;
#pragma unroll
    for (int k = 0; k < R; ++k) ref[(int64_t)cls * R + k] = sdpsr_round_key(U[(int64_t)k * n * n + ef], atol, scale);
}
template <int R>
__global__ void __launch_bounds__(256)
uconst_check_kernel(int n, const uint32_t* __restrict__ Lp, const double* __restrict__ U, double atol, double scale,
                    const uint64_t* __restrict__ ref, uint32_t* __restrict__ flag) {
    bool bad = false;
    const int64_t nn = (int64_t)n * n;
    const uint64_t zero_code = sdpsr_round_key(0.0, atol, scale);
    for (int j = blockIdx.x; j < n; j += gridDim.x) {
        const int64_t poff = (int64_t)j * n - (int64_t)j * (j - 1) / 2 - j;
        const uint32_t* Lj = Lp + poff;
        const double* Uj = U + (int64_t)j * n;
        constexpr int VB = 4;  // entries per trip: labels and values first, then the classes' reference codes (see verify_lower_kernel)
        for (int i0 = j + threadIdx.x; i0 < n; i0 += 256 * VB) {
            uint32_t l[VB];
            double uv[VB][R];
#pragma unroll
            for (int b = 0; b < VB; ++b) {
                const int i = i0 + 256 * b;
                const bool ok = i < n;
                l[b] = ok ? __builtin_nontemporal_load(&Lj[i]) : 0u;
#pragma unroll
                for (int k = 0; k < R; ++k) uv[b][k] = ok ? __builtin_nontemporal_load(&Uj[(int64_t)k * nn + i]) : 0.0;
            }
            uint64_t rc[VB][R];
#pragma unroll
            for (int b = 0; b < VB; ++b)
#pragma unroll
                for (int k = 0; k < R; ++k) rc[b][k] = l[b] ? ref[(int64_t)(l[b] - 1) * R + k] : zero_code;
#pragma unroll
            for (int b = 0; b < VB; ++b)
#pragma unroll
                for (int k = 0; k < R; ++k) bad = bad || sdpsr_round_key(uv[b][k], atol, scale) != rc[b][k];
        }
    }
    if (bad) flag[0] = 1u;
}
size_t uconst_ref_bytes(int64_t d, int64_t r) { return (size_t)(d > 0 ? d : 1) * (size_t)(r > 0 ? r : 1) * 8; }
// symmetric basis matrices U_k (n x n, column-major, r <= 4 of them), packed labels Lp of d <= REFINE_FIRST_CAP classes
// with their representatives first_idx; flag: a word the device can write (pinned host memory)
bool launch_basis_constant_on_classes(hipStream_t s, int64_t n, int64_t r, const double* U, const uint32_t* Lp, int64_t d,
                                      const uint32_t* first_idx, double atol, double scale, void* ref, uint32_t* flag) {
    if (r < 1 || r > 4 || d < 1 || d > (int64_t)REFINE_FIRST_CAP) return false;
    const unsigned gr = (unsigned)((d + 255) / 256);
    const int g = n < 256 * 8 ? (int)n : 256 * 8;
#define SDPSR_UCONST_CASE(RR)                                                                                          \
    case RR:                                                                                                           \
        uconst_ref_kernel<RR><<<gr, 256, 0, s>>>((int)n, (int)d, first_idx, U, atol, scale, (uint64_t*)ref, flag);      \
        uconst_check_kernel<RR><<<g, 256, 0, s>>>((int)n, Lp, U, atol, scale, (const uint64_t*)ref, flag);              \
        break;
    switch (r) {
        SDPSR_UCONST_CASE(1)
        SDPSR_UCONST_CASE(2)
        SDPSR_UCONST_CASE(3)
        SDPSR_UCONST_CASE(4)
    }
#undef SDPSR_UCONST_CASE
    return true;
}
// q: SIG_JOINT_I32 or SIG_CHAN_I32 on the packed lower triangle with packed labels (q.L = Lp); flag[0] = verdict.
// Returns false when there is no instance for the shape (the caller runs the refinement).
bool launch_verify_no_split(hipStream_t s, const SigSource& q, int64_t d, const uint32_t* first_idx, void* ref, uint32_t* flag) {
    if (!q.packed || !q.lab_packed || d < 1 || d > (int64_t)REFINE_FIRST_CAP || (q.T != 2 && q.T != 4)) return false;
    if (q.kind == SIG_CHAN_I32) {
        if (q.T == 2) launch_verify_rt<0, 2, false>(s, q, d, first_idx, ref, flag);
        else launch_verify_rt<0, 4, false>(s, q, d, first_idx, ref, flag);
        return true;
    }
    if (q.kind != SIG_JOINT_I32 || q.r < 0 || q.r > 4) return false;
#define SDPSR_VERIFY_CASE(RR)                                                       \
    case RR:                                                                        \
        if (q.T == 2) launch_verify_rt<RR, 2, true>(s, q, d, first_idx, ref, flag); \
        else launch_verify_rt<RR, 4, true>(s, q, d, first_idx, ref, flag);          \
        break;
    switch (q.r) {
        SDPSR_VERIFY_CASE(0)
        SDPSR_VERIFY_CASE(1)
        SDPSR_VERIFY_CASE(2)
        SDPSR_VERIFY_CASE(3)
        SDPSR_VERIFY_CASE(4)
    }
#undef SDPSR_VERIFY_CASE
    return true;
}

bool sig_source_fusable(const SigSource& q) {
    switch (q.kind) {
        case SIG_JOINT_I32: return q.r >= 0 && q.r <= 4 && (q.T == 2 || q.T == 4) && q.packed;
        case SIG_ARRAY: return true;
        case SIG_PAIR: return true;
        case SIG_PROJ: return q.r >= 0 && q.r <= 4;
        case SIG_CHAN_I32:
        case SIG_CHAN_F32: return q.T == 1 || q.T == 2 || q.T == 4 || q.T == 8;
        default: return false;
    }
}

// the signature array of a computed source (sort path, or a source the insert kernel has no
// instance for): the stand-alone kernels
void launch_sig_materialize(hipStream_t s, int64_t len, const SigSource& q, uint64_t* sig) {
    switch (q.kind) {
        case SIG_PAIR:
            if (q.packed) sig_f64_pair_lower_kernel<<<(int)(q.n < 2048 ? q.n : 2048), 256, 0, s>>>((int)q.n, q.a, q.b, sig);
            else launch_sig_f64_pair(s, len, q.a, q.b, sig);
            break;
        case SIG_PROJ:
            if (q.packed) launch_proj_apply_lower(s, q.n, q.r, q.U, q.L, q.lab_packed, q.key, q.coef, q.atol, q.scale, sig);
            else launch_proj_apply(s, len, q.r, q.U, q.L, q.key, nullptr, q.coef, q.atol, q.scale, 1, nullptr, sig);
            break;
        case SIG_JOINT_I32:
            sig_joint_lower_kernel<<<(int)(q.n < 2048 ? q.n : 2048), 256, 0, s>>>((int)q.n, q.ld, q.r, q.T, q.U, q.L, q.lab_packed, q.key, q.coef,
                                                                                 q.atol, q.scale, (const int32_t*)q.C, sig);
            break;
        case SIG_CHAN_I32: launch_sig_i32(s, q.n, q.ld, q.T, q.L, (const int32_t*)q.C, sig, q.zero_flag, q.packed, q.lab_packed); break;
        case SIG_CHAN_F32: launch_sig_f32(s, q.n, q.ld, q.T, q.L, (const float*)q.C, sig, q.zero_flag, q.packed, q.lab_packed); break;
        default: break;
    }
}

// does launch_refine run the one-workgroup-per-CU insert kernel for this source (RefineWs::mid)?
bool refine_mid_supports(const SigSource& q) {
    switch (q.kind) {
        case SIG_ARRAY:
        case SIG_PAIR: return true;
        case SIG_CHAN_I32: return q.T == 2;
        default: return false;
    }
}

void launch_refine(hipStream_t s, int64_t len, const SigSource& q, uint32_t* slot, uint32_t* labels_out,
                   const RefineWs& ws, int64_t sym_n) {
    const size_t cap = (size_t)1 << ws.log2cap;
    refine_clear_kernel<<<grid_for((int64_t)cap, 256), 256, 0, s>>>((int64_t)cap, ws.tab, ws.counters);
    const int64_t nblk = (len + REFINE_BLOCK - 1) / REFINE_BLOCK;
    // 116 VGPRs + 32 KiB of LDS: four workgroups are resident per CU; with few classes (LDS-level
    // work) launch exactly one round of resident workgroups -- a fifth per CU would run alone
    const int per_cu = (ws.log2cap <= 16) ? 4 : 5;  // many classes: bound by the global table, a few more workgroups help
    const int gcap = 256 * per_cu;
    switch (q.kind) {
        case SIG_PAIR: launch_insert<SrcPair, 8>(s, gcap, len, SrcPair{q.a, q.b, (int)q.n, q.packed}, slot, ws, cap); break;
        case SIG_PROJ:
            switch (q.r) {
                case 0: launch_insert<SrcProj<0>, 8>(s, gcap, len, SrcProj<0>{q.U, q.L, q.coef, q.packed ? q.n * q.n : len, q.key, q.atol, q.scale, (int)q.n, q.packed, q.lab_packed}, slot, ws, cap); break;
                case 1: launch_insert<SrcProj<1>, 8>(s, gcap, len, SrcProj<1>{q.U, q.L, q.coef, q.packed ? q.n * q.n : len, q.key, q.atol, q.scale, (int)q.n, q.packed, q.lab_packed}, slot, ws, cap); break;
                case 2: launch_insert<SrcProj<2>, 8>(s, gcap, len, SrcProj<2>{q.U, q.L, q.coef, q.packed ? q.n * q.n : len, q.key, q.atol, q.scale, (int)q.n, q.packed, q.lab_packed}, slot, ws, cap); break;
                case 3: launch_insert<SrcProj<3>, 8>(s, gcap, len, SrcProj<3>{q.U, q.L, q.coef, q.packed ? q.n * q.n : len, q.key, q.atol, q.scale, (int)q.n, q.packed, q.lab_packed}, slot, ws, cap); break;
                default: launch_insert<SrcProj<4>, 8>(s, gcap, len, SrcProj<4>{q.U, q.L, q.coef, q.packed ? q.n * q.n : len, q.key, q.atol, q.scale, (int)q.n, q.packed, q.lab_packed}, slot, ws, cap); break;
            }
            break;
        case SIG_JOINT_I32: {
            const int32_t* Cj = (const int32_t*)q.C;
#define SDPSR_JOINT_CASE(RR, TT)                                                                                                   \
    launch_insert<SrcJoint<RR, TT>, 8>(s, gcap, len, SrcJoint<RR, TT>{q.U, q.L, q.coef, q.key, q.atol, q.scale, Cj, q.ld, (int)q.n, \
                                                                      q.lab_packed}, slot, ws, cap)
            const int rr = q.r < 0 ? 0 : (q.r > 4 ? 4 : q.r);
            if (q.T == 2) {
                switch (rr) {
                    case 0: SDPSR_JOINT_CASE(0, 2); break;
                    case 1: SDPSR_JOINT_CASE(1, 2); break;
                    case 2: SDPSR_JOINT_CASE(2, 2); break;
                    case 3: SDPSR_JOINT_CASE(3, 2); break;
                    default: SDPSR_JOINT_CASE(4, 2); break;
                }
            } else {
                switch (rr) {
                    case 0: SDPSR_JOINT_CASE(0, 4); break;
                    case 1: SDPSR_JOINT_CASE(1, 4); break;
                    case 2: SDPSR_JOINT_CASE(2, 4); break;
                    case 3: SDPSR_JOINT_CASE(3, 4); break;
                    default: SDPSR_JOINT_CASE(4, 4); break;
                }
            }
#undef SDPSR_JOINT_CASE
            break;
        }
        case SIG_CHAN_I32: launch_insert_chan<int32_t>(s, gcap, len, q, slot, ws, cap); break;
        case SIG_CHAN_F32: launch_insert_chan<float>(s, gcap, len, q, slot, ws, cap); break;
        default:
            if (ws.mid) {
                launch_insert_mid<SrcArray>(s, len, SrcArray{q.sig}, slot, ws, cap);
            } else {
                launch_insert<SrcArray, 16, 2048>(s, gcap, len, SrcArray{q.sig}, slot, ws, cap);
            }
            break;
    }
    const int g2 = (int)(nblk < 256 * 8 ? nblk : 256 * 8);
    // ws.expect_small: the host predicts <= SMALL_K classes (from the previous refinement) and
    // launches the one-workgroup ranking only; it checks counters[0] afterwards and repeats the
    // pass with expect_small = 0 on a misprediction (the three general kernels would exit at once
    // anyway, but three empty launches cost ~15 us of a ~150 us refinement)
    refine_small_rank_kernel<<<1, 1024, 0, s>>>(ws.tab, ws.tab_lab, ws.counters, ws.first_idx);
    if (!ws.expect_small) {
        // more than SMALL_K classes: ranked from the table's side (kernels_refine_bucket.hip: one bit per first index, rank
        // records per 64 entries) -- work on the classes and on len / 64 words.  The entry-level count / scan / rank passes
        // stay for callers without that workspace.
        if (!(ws.rank_ws && launch_rank_slots(s, len, (int64_t)cap, ws.tab, ws.tab_lab, ws.counters, SMALL_K, ws.first_idx,
                                              REFINE_FIRST_CAP, ws.rank_ws, ws.rank_ws_bytes))) {
            refine_count_kernel<<<g2, REFINE_THREADS, 0, s>>>(len, slot, ws.tab, ws.blk_cnt, ws.counters);
            refine_scan_kernel<<<1, 1024, 0, s>>>(nblk, ws.blk_cnt, ws.counters);
            refine_rank_kernel<<<g2, REFINE_THREADS, 0, s>>>(len, slot, ws.tab, ws.blk_cnt,
                                                             ws.tab_lab, ws.counters, ws.first_idx);
        }
    }
    if (sym_n > 0 && sym_n * sym_n == len) {  // labels of an n x n matrix: the symmetry verdict comes with the label pass
        const unsigned t = (unsigned)((sym_n + 63) / 64);
        if (sym_n % 4 == 0) refine_label_sym_kernel<true><<<dim3(t, t), 256, 0, s>>>(sym_n, slot, labels_out, ws.tab_lab, ws.counters, ws.expect_small);
        else refine_label_sym_kernel<false><<<dim3(t, t), 256, 0, s>>>(sym_n, slot, labels_out, ws.tab_lab, ws.counters, ws.expect_small);
    } else {
        refine_label_kernel<<<grid_for(len, 256), 256, 0, s>>>(len, slot, labels_out, ws.tab_lab, ws.counters, ws.expect_small, ws.host_counters, ws.host_seq);
    }
}

// ---------------------------------------------------------------------------
// largest pair code l1 + l2 (d1 + 1) of refine!(P1, P2) (src/partitions.jl:63): what the reference
// stores into its label type T before renumbering -- InexactError when it exceeds typemax(T)
// (sdpsr_opts.label_bits).  out[0] must be zero.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
max_pair_code_kernel(int64_t len, const uint32_t* __restrict__ p1, const uint32_t* __restrict__ p2, unsigned long long d1p1,
                     unsigned long long* __restrict__ out) {
    unsigned long long m = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < len; e += stride) {
        const unsigned long long v = (unsigned long long)p1[e] + (unsigned long long)p2[e] * d1p1;
        m = v > m ? v : m;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long t = __shfl_down(m, o, 64);
        m = t > m ? t : m;
    }
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}
void launch_max_pair_code(hipStream_t s, int64_t len, const uint32_t* p1, const uint32_t* p2, uint64_t d1, uint64_t* out) {
    max_pair_code_kernel<<<grid_for(len, 256), 256, 0, s>>>(len, p1, p2, (unsigned long long)d1 + 1ull, (unsigned long long*)out);
}

// ---------------------------------------------------------------------------
// checksum of a label array: two position-weighted sums mod 2^64 (order of summation irrelevant)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
labels_checksum_kernel(int64_t len, const uint32_t* __restrict__ L, unsigned long long* __restrict__ partial) {
    __shared__ unsigned long long sh[2][4];
    unsigned long long h1 = 0, h2 = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < len; e += stride) {
        const unsigned long long l = (unsigned long long)L[e] + 1ull;
        const unsigned long long ee = (unsigned long long)e;
        h1 += l * (ee * 0x9E3779B97F4A7C15ull + 0xD1342543DE82EF95ull);
        h2 += (l * l + 0x27D4EB2F165667C5ull) * ((ee ^ (ee >> 13)) * 0xBF58476D1CE4E5B9ull + 0x94D049BB133111EBull);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        h1 += __shfl_down(h1, o, 64);
        h2 += __shfl_down(h2, o, 64);
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) {
        sh[0][w] = h1;
        sh[1][w] = h2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3];
        partial[gridDim.x + blockIdx.x] = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
    }
}
__global__ void __launch_bounds__(256)
labels_checksum_final_kernel(int nblk, const unsigned long long* __restrict__ partial, unsigned long long* __restrict__ out) {
    __shared__ unsigned long long sh[4];
    for (int which = 0; which < 2; ++which) {
        unsigned long long h = 0;
        for (int b = threadIdx.x; b < nblk; b += 256) h += partial[which * nblk + b];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) h += __shfl_down(h, o, 64);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = h;
        __syncthreads();
        if (threadIdx.x == 0) out[which] = sh[0] + sh[1] + sh[2] + sh[3];
        __syncthreads();
    }
}
// partial: 2 * 2048 words of scratch, out: 2 words (device)
void launch_labels_checksum(hipStream_t s, int64_t len, const uint32_t* L, uint64_t* partial, uint64_t* out) {
    const int nblk = grid_for(len, 256);
    labels_checksum_kernel<<<nblk, 256, 0, s>>>(len, L, (unsigned long long*)partial);
    labels_checksum_final_kernel<<<1, 256, 0, s>>>(nblk, (const unsigned long long*)partial, (unsigned long long*)out);
}

// ---------------------------------------------------------------------------
// symmetric label check
// ---------------------------------------------------------------------------
// 64 x 64 tiles through LDS: both the tile and its mirror image are read along columns
__global__ void __launch_bounds__(256)
check_symmetric_kernel(int64_t n, const uint32_t* __restrict__ L, uint32_t* flag) {
    __shared__ uint32_t tile[64][65];
    const int64_t i0 = (int64_t)blockIdx.x * 64, j0 = (int64_t)blockIdx.y * 64;
    if (i0 < j0) return;  // lower triangle of tiles only
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
    for (int c = ty; c < 64; c += 4) {  // mirror tile: rows j0.., columns i0..
        const int64_t r = j0 + tx, cc = i0 + c;
        tile[c][tx] = (r < n && cc < n) ? L[r + cc * n] : 0u;
    }
    __syncthreads();
    bool bad = false;
    for (int c = ty; c < 64; c += 4) {  // tile: rows i0.., columns j0..
        const int64_t r = i0 + tx, cc = j0 + c;
        if (r < n && cc < n && L[r + cc * n] != tile[tx][c]) bad = true;  // L[r,cc] vs L[cc,r]
    }
    if (bad) flag[0] = 1u;
}
// dst = src (n x n labels) with the symmetry check of the same tiles: one pass instead of a copy
// and a check.  flag[0] = epoch if NOT symmetric (no zeroing pass: the host compares with the
// epoch of this call).
template <bool VEC4>
__global__ void __launch_bounds__(256)
copy_check_symmetric_kernel(int64_t n, const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, uint32_t* flag,
                            uint32_t epoch) {
    __shared__ uint32_t tile[64][65];
    if (blockIdx.x < blockIdx.y) return;
    if (sym_tile_pass<VEC4>(n, src, dst, MapIdentity{}, tile)) flag[0] = epoch;
}
void launch_copy_check_symmetric(hipStream_t s, int64_t n, const uint32_t* src, uint32_t* dst, uint32_t* flag, uint32_t epoch) {
    const unsigned t = (unsigned)((n + 63) / 64);
    const bool v4 = n % 4 == 0 && (reinterpret_cast<uintptr_t>(src) % 16) == 0 && (reinterpret_cast<uintptr_t>(dst) % 16) == 0;
    if (v4) copy_check_symmetric_kernel<true><<<dim3(t, t), 256, 0, s>>>(n, src, dst, flag, epoch);
    else copy_check_symmetric_kernel<false><<<dim3(t, t), 256, 0, s>>>(n, src, dst, flag, epoch);
}
void launch_check_symmetric(hipStream_t s, int64_t n, const uint32_t* L, uint32_t* flag) {
    hipMemsetAsync(flag, 0, sizeof(uint32_t), s);
    const unsigned t = (unsigned)((n + 63) / 64);
    check_symmetric_kernel<<<dim3(t, t), 256, 0, s>>>(n, L, flag);
}

// Full symmetric label matrix from the packed lower triangle (column j at offset
// j n - j (j - 1) / 2, rows j .. n-1): both triangles are written along columns, the mirrored
// one through a 64 x 64 LDS tile.
__global__ void __launch_bounds__(256)
unpack_symmetric_labels_kernel(int64_t n, const uint32_t* __restrict__ Lp, uint32_t* __restrict__ L) {
    __shared__ uint32_t tile[64][65];
    const int64_t i0 = (int64_t)blockIdx.x * 64, j0 = (int64_t)blockIdx.y * 64;  // lower tile: rows i0.., cols j0..
    if (i0 < j0) return;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int c = ty; c < 64; c += 4) {
        const int64_t r = i0 + tx, cc = j0 + c;
        uint32_t v = 0u;
        if (r < n && cc < n && r >= cc) {
            v = Lp[cc * n - cc * (cc - 1) / 2 + (r - cc)];
            L[r + cc * n] = v;
        }
        tile[c][tx] = v;
    }
    __syncthreads();
    for (int c = ty; c < 64; c += 4) {  // destination: row j0 + tx, column i0 + c  (= entry (i0 + c, j0 + tx))
        const int64_t r = j0 + tx, cc = i0 + c;
        if (r < n && cc < n && r < cc) L[r + cc * n] = tile[tx][c];
    }
}
void launch_unpack_symmetric_labels(hipStream_t s, int64_t n, const uint32_t* Lp, uint32_t* L) {
    const unsigned t = (unsigned)((n + 63) / 64);
    unpack_symmetric_labels_kernel<<<dim3(t, t), 256, 0, s>>>(n, Lp, L);
}

// Lt[k + i*n] = L[i + k*n]: 64 x 64 label tiles through LDS (both sides coalesced)
__global__ void __launch_bounds__(256)
transpose_labels_kernel(int n, const uint32_t* __restrict__ L, uint32_t* __restrict__ Lt) {
    __shared__ uint32_t tile[64][65];
    const int bi = blockIdx.x * 64, bk = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 4 rows of the tile per pass
    for (int q = ty; q < 64; q += 4) {
        const int i = bi + tx, kk = bk + q;
        tile[q][tx] = (i < n && kk < n) ? L[i + (int64_t)kk * n] : 0u;
    }
    __syncthreads();
    for (int q = ty; q < 64; q += 4) {
        const int kk = bk + tx, i = bi + q;
        if (kk < n && i < n) Lt[kk + (int64_t)i * n] = tile[tx][q];
    }
}
void launch_transpose_labels(hipStream_t s, int64_t n, const uint32_t* L, uint32_t* Lt) {
    dim3 g((unsigned)((n + 63) / 64), (unsigned)((n + 63) / 64));
    transpose_labels_kernel<<<g, 256, 0, s>>>((int)n, L, Lt);
}

// ---------------------------------------------------------------------------
// Reduced-SDP assembly (README.md:57-60, test/sd_problems.jl:32-37): out = A * PMat with
// PMat[e, i] = 1[L[e] == i], i.e. the columns of A (m x n^2, column-major) summed per class.
// One wave per chunk of entries: lanes own the rows of A (m <= 64 per pass), the per-class
// accumulators sit in LDS, entries are walked in order (fixed summation order); chunk partials
// are added in chunk order by the second kernel.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
reduce_columns_kernel(int64_t len, int64_t chunk, int m, int d, const uint32_t* __restrict__ L,
                      const double* __restrict__ A, double* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) double s_bins[];  // [(d + 1)][mw]
    const int lane = threadIdx.x;
    const int64_t e0 = (int64_t)blockIdx.x * chunk;
    int64_t e1 = e0 + chunk;
    if (e1 > len) e1 = len;
    for (int r0 = 0; r0 < m; r0 += 64) {
        const int mw = (m - r0 < 64) ? m - r0 : 64;
        for (int t = lane; t < (d + 1) * mw; t += 64) s_bins[t] = 0.0;
        __syncthreads();
        for (int64_t eb = e0; eb < e1; eb += 64) {
            const uint32_t mylab = (eb + lane < e1) ? L[eb + lane] : 0u;
            const int lim = (int)((e1 - eb < 64) ? e1 - eb : 64);
            for (int u = 0; u < lim; ++u) {
                const uint32_t lb = __shfl(mylab, u, 64);
                if (lane < mw) s_bins[lb * mw + lane] += A[(int64_t)(eb + u) * m + r0 + lane];
            }
        }
        __syncthreads();
        for (int t = lane; t < d * mw; t += 64) {
            const int i = t / mw, r = t - i * mw;
            partial[((int64_t)blockIdx.x * d + i) * m + r0 + r] = s_bins[(i + 1) * mw + r];
        }
        __syncthreads();
    }
}
__global__ void reduce_columns_final_kernel(int64_t nchunks, int m, int d, const double* __restrict__ partial,
                                            double* __restrict__ out) {
    const int64_t total = (int64_t)m * d;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int64_t i = t / m, r = t - i * m;
    double acc = 0;
    for (int64_t ch = 0; ch < nchunks; ++ch) acc += partial[(ch * d + i) * m + r];
    out[r + i * m] = acc;  // m x d column-major
}
// per-device kernel attributes, set by sdpsr_create() (see gemm_set_device_attributes)
bool partition_set_device_attributes() {
    bool ok = true;
    ok &= hipSuccess == hipFuncSetAttribute(reinterpret_cast<const void*>(&reduce_columns_kernel),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    ok &= mid_set_attributes_kind<SrcArray>();
    ok &= mid_set_attributes_kind<SrcPair>();
    ok &= mid_set_attributes_kind<SrcChan<int32_t, 2>>();
    return ok;
}

int64_t reduce_columns_chunk(int64_t len, int64_t m, int64_t d) {
    int64_t chunk = 4096;
    while ((len + chunk - 1) / chunk * d * m * 8 > ((int64_t)64 << 20)) chunk *= 2;
    return chunk;
}
bool launch_reduce_columns(hipStream_t s, int64_t len, int64_t m, int64_t d, const uint32_t* L, const double* A,
                           double* partial, double* out) {
    const int mw = (int)(m < 64 ? m : 64);
    const size_t lds = (size_t)(d + 1) * mw * 8;
    if (lds > 60 * 1024) return false;
    const int64_t chunk = reduce_columns_chunk(len, m, d);
    const int64_t nch = (len + chunk - 1) / chunk;
    reduce_columns_kernel<<<(unsigned)nch, 64, lds, s>>>(len, chunk, (int)m, (int)d, L, A, partial);
    reduce_columns_final_kernel<<<(unsigned)((m * d + 255) / 256), 256, 0, s>>>(nch, (int)m, (int)d, partial, out);
    return true;
}

// ---------------------------------------------------------------------------
// small vector kernels of the device setup stage (src/partitions.jl:117-142)
// ---------------------------------------------------------------------------
// R[e + i*len] = A[i + e*m]
__global__ void transpose_rows_kernel(int64_t len, int m, const double* __restrict__ A, double* __restrict__ R) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < len; e += stride)
        for (int i = 0; i < m; ++i) R[e + (int64_t)i * len] = A[i + e * m];
}
void launch_transpose_rows(hipStream_t s, int64_t len, int64_t m, const double* A, double* R) {
    transpose_rows_kernel<<<grid_for(len, 256), 256, 0, s>>>(len, (int)m, A, R);
}
// out[k] = |V[:,k]|^2 ; grid (nblk, k)
__global__ void col_norms2_kernel(int64_t len, const double* __restrict__ V, double* __restrict__ partial) {
    __shared__ double sh[8];
    const double* v = V + (int64_t)blockIdx.y * len;
    double acc = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < len; e += stride) acc = fma(v[e], v[e], acc);
    double r = block_reduce_sum(acc, sh);
    if (threadIdx.x == 0) partial[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = r;
}
void launch_col_norms2(hipStream_t s, int64_t len, int64_t k, const double* V, double* partial, int nblk, double* out) {
    if (k <= 0) return;
    dim3 g(nblk, (unsigned)k);
    col_norms2_kernel<<<g, 256, 0, s>>>(len, V, partial);
    proj_coef_final_kernel<<<(unsigned)k, 256, 0, s>>>(nblk, partial, out);
}
__global__ void scale_copy_kernel(int64_t len, const double* __restrict__ v, double alpha, double* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < len; e += stride) out[e] = v[e] * alpha;
}
void launch_scale_copy(hipStream_t s, int64_t len, const double* v, double alpha, double* out) {
    scale_copy_kernel<<<grid_for(len, 256), 256, 0, s>>>(len, v, alpha, out);
}
// R[:, i] -= dots[i] * u   for every i (dots[i] = 0 leaves the row alone)
__global__ void rank1_update_kernel(int64_t len, int m, double* __restrict__ R, const double* __restrict__ u,
                                    const double* __restrict__ dots) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < len; e += stride) {
        const double ue = u[e];
        for (int i = 0; i < m; ++i) {
            const double dd = dots[i];
            if (dd != 0.0) R[e + (int64_t)i * len] = fma(-dd, ue, R[e + (int64_t)i * len]);
        }
    }
}
void launch_rank1_update(hipStream_t s, int64_t len, int64_t m, double* R, const double* u, const double* dots) {
    rank1_update_kernel<<<grid_for(len, 256), 256, 0, s>>>(len, (int)m, R, u, dots);
}
// out = clamp_round(a - b)
__global__ void sub_round_kernel(int64_t len, const double* __restrict__ a, const double* __restrict__ b, double atol,
                                 double scale, double* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < len; e += stride)
        out[e] = sdpsr_clamp_round(a[e] - b[e], atol, scale);
}
void launch_sub_round(hipStream_t s, int64_t len, const double* a, const double* b, double atol, double scale, double* out) {
    sub_round_kernel<<<grid_for(len, 256), 256, 0, s>>>(len, a, b, atol, scale, out);
}

}  // namespace sdpsr
