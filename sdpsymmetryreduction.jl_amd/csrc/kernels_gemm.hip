// C = A' * B on the gfx950 matrix cores, column-major operands (A: k x m, B: k x n, so both
// are read along their contiguous dimension), MFMA tiles staged through LDS.
//
// Serves  mul!(X2, X, X)           src/partitions.jl:172   (X symmetric => X*X = X'X)
//         Q' * A * Q               src/eigen_decomposition.jl:203 (A symmetric)
//         A * F (first columns)    src/eigen_decomposition.jl:333
//
// Variants: int8 -> int32 (v_mfma_i32_32x32x32_i8), f32 (v_mfma_f32_32x32x2_f32, exact f32
// fma chain), f64 (v_mfma_f64_16x16x4_f64).  One workgroup = 4 waves = one 128 x 128 tile
// of C; a wave owns a 64 x 64 sub-tile.  K is walked in tiles of KB bytes per operand
// row, register-staged and double-buffered in LDS (one barrier per K-tile); LDS rows are
// padded by 16 B so the ds_read_b128 fragment reads are bank-conflict free.
//
// Fragment rule used throughout: a lane reads 16 contiguous bytes of "its" operand row
// and uses them for 1 (i8), 4 (f32) or 2 (f64) consecutive MFMAs.  A and B fragments are
// cut the same way, so each MFMA multiplies matching k indices; the order in which k is
// consumed differs from the natural one, which only permutes an exact sum (integers) or
// the fp rounding order (f32/f64).
#include <cstdlib>
#include "sdpsr_internal.h"

namespace sdpsr {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef double v2d __attribute__((ext_vector_type(2)));
typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int BM = 128;   // rows of C per workgroup (i, from operand A)
constexpr int BN = 128;   // cols of C per workgroup (j, from operand B)
constexpr int NT = 256;   // threads

enum { KIND_I8 = 0, KIND_F32 = 1, KIND_F64 = 2 };

template <int KIND> struct GemmTraits;
template <> struct GemmTraits<KIND_I8> {
    typedef int8_t in_t;
    typedef int32_t out_t;
    static constexpr int KB = 64;  // bytes of K per operand row per K-tile
};
template <> struct GemmTraits<KIND_F32> {
    typedef float in_t;
    typedef float out_t;
    static constexpr int KB = 128;
};
template <> struct GemmTraits<KIND_F64> {
    typedef double in_t;
    typedef double out_t;
    static constexpr int KB = 128;
};

template <int KIND>
__global__ void __launch_bounds__(NT)
gemm_tn_kernel(int64_t k, const typename GemmTraits<KIND>::in_t* __restrict__ Ag, int64_t lda,
               const typename GemmTraits<KIND>::in_t* __restrict__ Bg, int64_t ldb,
               typename GemmTraits<KIND>::out_t* __restrict__ Cg, int64_t ldc, int64_t strideA,
               int64_t strideB, int64_t strideC, const uint32_t* __restrict__ nonsym_flag, int cmode) {
    typedef GemmTraits<KIND> TR;
    typedef typename TR::in_t in_t;
    typedef typename TR::out_t out_t;
    constexpr int KB = TR::KB;
    constexpr int ES = sizeof(in_t);
    constexpr int KE = KB / ES;          // k elements per K-tile
    constexpr int RS = KB + 16;          // padded LDS row stride (bytes)
    constexpr int CH = KB / 16;          // 16-byte chunks per row
    constexpr int LPT = BM * CH / NT;    // chunks per thread per operand (2 or 4)
    constexpr int OPB = BM * RS;         // bytes per operand tile in LDS

    extern __shared__ __attribute__((aligned(16))) char smem[];
    // layout: [buf][operand][row][RS]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wi = wave & 1;   // wave position along i
    const int wj = wave >> 1;  // wave position along j

    const int64_t i0 = (int64_t)blockIdx.x * BM;
    const int64_t j0 = (int64_t)blockIdx.y * BN;
    // symmetric product (C = X'X) whose consumer only reads the lower triangle: tiles above the
    // diagonal are skipped while the device flag says the labels are symmetric
    if (nonsym_flag && i0 < j0 && *nonsym_flag == 0u) return;
    const char* Ab = reinterpret_cast<const char*>(Ag + (int64_t)blockIdx.z * strideA + i0 * lda);
    const char* Bb = reinterpret_cast<const char*>(Bg + (int64_t)blockIdx.z * strideB + j0 * ldb);
    out_t* C = Cg + (int64_t)blockIdx.z * strideC;

    // staging assignment: chunk q = tid + s*NT -> row q / CH, chunk q % CH
    int srow[LPT], scol[LPT];
#pragma unroll
    for (int s = 0; s < LPT; ++s) {
        int q = tid + s * NT;
        srow[s] = q / CH;
        scol[s] = q % CH;
    }
    uint4 ra[LPT], rb[LPT];
    auto load_tile = [&](int64_t kt) {
        const int64_t kbyte = kt * KB;
#pragma unroll
        for (int s = 0; s < LPT; ++s) {
            ra[s] = *reinterpret_cast<const uint4*>(Ab + (int64_t)srow[s] * lda * ES + kbyte + scol[s] * 16);
            rb[s] = *reinterpret_cast<const uint4*>(Bb + (int64_t)srow[s] * ldb * ES + kbyte + scol[s] * 16);
        }
    };
    auto store_tile = [&](int buf) {
        char* base = smem + buf * 2 * OPB;
#pragma unroll
        for (int s = 0; s < LPT; ++s) {
            *reinterpret_cast<uint4*>(base + srow[s] * RS + scol[s] * 16) = ra[s];
            *reinterpret_cast<uint4*>(base + OPB + srow[s] * RS + scol[s] * 16) = rb[s];
        }
    };

    const int64_t nk = k / KE;

    if constexpr (KIND == KIND_I8 || KIND == KIND_F32) {
        // 32x32 MFMA tiles: 2 (j) x 2 (i) per wave.  MFMA "A" operand <- B tile (j),
        // MFMA "B" operand <- A tile (i): D[jj][ii], lanes run along i (contiguous in C).
        typedef typename std::conditional<KIND == KIND_I8, v16i, v16f>::type acc_t;
        acc_t acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0;
        const int r32 = lane & 31, h = lane >> 5;

        load_tile(0);
        store_tile(0);
        __syncthreads();
        for (int64_t kt = 0; kt < nk; ++kt) {
            const int buf = (int)(kt & 1);
            if (kt + 1 < nk) load_tile(kt + 1);
            const char* tA = smem + buf * 2 * OPB;  // rows i
            const char* tB = tA + OPB;              // rows j
            constexpr int NQ = KB / 32;             // 32-byte groups per row (2 halves x 16 B)
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                uint4 fi[2], fj[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    fi[t] = *reinterpret_cast<const uint4*>(tA + (wi * 64 + t * 32 + r32) * RS + q * 32 + h * 16);
                    fj[t] = *reinterpret_cast<const uint4*>(tB + (wj * 64 + t * 32 + r32) * RS + q * 32 + h * 16);
                }
                if constexpr (KIND == KIND_I8) {
#pragma unroll
                    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
                        for (int ti = 0; ti < 2; ++ti) {
                            v4i a = {(int)fj[tj].x, (int)fj[tj].y, (int)fj[tj].z, (int)fj[tj].w};
                            v4i b = {(int)fi[ti].x, (int)fi[ti].y, (int)fi[ti].z, (int)fi[ti].w};
                            acc[tj][ti] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[tj][ti], 0, 0, 0);
                        }
                } else {
                    const float* fjf0 = reinterpret_cast<const float*>(&fj[0]);
                    const float* fjf1 = reinterpret_cast<const float*>(&fj[1]);
                    const float* fif0 = reinterpret_cast<const float*>(&fi[0]);
                    const float* fif1 = reinterpret_cast<const float*>(&fi[1]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fjf0[e], fif0[e], acc[0][0], 0, 0, 0);
                        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fjf0[e], fif1[e], acc[0][1], 0, 0, 0);
                        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fjf1[e], fif0[e], acc[1][0], 0, 0, 0);
                        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fjf1[e], fif1[e], acc[1][1], 0, 0, 0);
                    }
                }
            }
            if (kt + 1 < nk) store_tile(buf ^ 1);
            __syncthreads();
        }
        // D[jj][ii]: ii = lane & 31, jj = (reg & 3) + 8 * (reg >> 2) + 4 * h
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int ti = 0; ti < 2; ++ti) {
                const int64_t ii = i0 + wi * 64 + ti * 32 + r32;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t jj = j0 + wj * 64 + tj * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    C[ii + jj * ldc] = acc[tj][ti][r];
                }
            }
    } else {
        // f64: 16x16x4 tiles, 4 (j) x 4 (i) per wave; lane (r16 = lane & 15, g = lane >> 4)
        v4d acc[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[a][b][r] = 0.0;
        const int r16 = lane & 15, g = lane >> 4;
        load_tile(0);
        store_tile(0);
        __syncthreads();
        for (int64_t kt = 0; kt < nk; ++kt) {
            const int buf = (int)(kt & 1);
            if (kt + 1 < nk) load_tile(kt + 1);
            const char* tA = smem + buf * 2 * OPB;
            const char* tB = tA + OPB;
            constexpr int NQ = KB / 64;  // 64-byte groups (4 lane groups x 16 B)
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                v2d fi[4], fj[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    fi[t] = *reinterpret_cast<const v2d*>(tA + (wi * 64 + t * 16 + r16) * RS + q * 64 + g * 16);
                    fj[t] = *reinterpret_cast<const v2d*>(tB + (wj * 64 + t * 16 + r16) * RS + q * 64 + g * 16);
                }
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int tj = 0; tj < 4; ++tj)
#pragma unroll
                        for (int ti = 0; ti < 4; ++ti)
                            acc[tj][ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(fj[tj][e], fi[ti][e], acc[tj][ti], 0, 0, 0);
            }
            if (kt + 1 < nk) store_tile(buf ^ 1);
            __syncthreads();
        }
        // D[jj][ii]: ii = lane & 15, jj = (lane >> 4) + 4 * reg
        if (cmode == 0) {
#pragma unroll
            for (int tj = 0; tj < 4; ++tj)
#pragma unroll
                for (int ti = 0; ti < 4; ++ti) {
                    const int64_t ii = i0 + wi * 64 + ti * 16 + r16;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int64_t jj = j0 + wj * 64 + tj * 16 + g + 4 * r;
                        C[ii + jj * ldc] = acc[tj][ti][r];
                    }
                }
        } else {  // C -= A'B: batches of 16 independent loads, then the stores
#pragma unroll
            for (int tj = 0; tj < 4; ++tj) {
                out_t cv[4][4];
#pragma unroll
                for (int ti = 0; ti < 4; ++ti)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        cv[ti][r] = C[(i0 + wi * 64 + ti * 16 + r16) + (j0 + wj * 64 + tj * 16 + g + 4 * r) * ldc];
#pragma unroll
                for (int ti = 0; ti < 4; ++ti)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        C[(i0 + wi * 64 + ti * 16 + r16) + (j0 + wj * 64 + tj * 16 + g + 4 * r) * ldc] = cv[ti][r] - acc[tj][ti][r];
            }
        }
    }
}


// ---------------------------------------------------------------------------
// Variant with direct global->LDS loads (all three dtypes) (LDS-DMA, global_load_lds_dwordx4): no VGPR
// staging and no ds_write pass.  K-tile = 128 bytes per operand row; one wave instruction
// fills 8 rows x 128 B of the lane-linear LDS image, the XOR swizzle that makes the
// ds_read_b128 fragment reads conflict-free is applied on the per-lane SOURCE address
// (slot s of row r holds global chunk s ^ ((r >> 1) & 7)) and undone on the read.
// Two LDS buffers (64 KiB): tile t+1 streams in while tile t feeds the MFMAs.
// ---------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

// global -> LDS DMA of 16 bytes per lane (1 KiB per wave instruction, lane-linear at lds_dst).
// Issued through inline asm ON PURPOSE: for the builtin the compiler books a pending LDS write on
// the VM counter and, unable to prove that the fragment reads of the OTHER buffer do not alias it,
// puts an `s_waitcnt vmcnt(0)` in front of the first ds_read of every K-tile -- the loads of tile
// t+1 were drained before tile t was touched and nothing overlapped (measured at N = 4096, int8,
// 4 channels, lower tiles: loads alone 0.124 ms, MFMAs alone 0.127 ms, together 0.217 ms).  The
// asm form is invisible to that bookkeeping; ordering is by the explicit counted waits + barriers
// of the K loops below.  M0 (the DMA destination base) is saved and restored in the same statement.
__device__ __forceinline__ void glds16(const void* gsrc, const void* lds_dst_generic) {
    const unsigned lds_dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_void_t*)lds_dst_generic);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

template <int KIND, int CMODE>  // CMODE 0: C = A'B, 1: C -= A'B
__global__ void __launch_bounds__(NT)
gemm_tn_dma_kernel(int64_t k, const typename GemmTraits<KIND>::in_t* __restrict__ Ag, int64_t lda,
                   const typename GemmTraits<KIND>::in_t* __restrict__ Bg, int64_t ldb,
                   typename GemmTraits<KIND>::out_t* __restrict__ Cg, int64_t ldc, int64_t strideA, int64_t strideB,
                   int64_t strideC, const uint32_t* __restrict__ nonsym_flag) {
    typedef typename GemmTraits<KIND>::in_t in_t;
    typedef typename GemmTraits<KIND>::out_t out_t;
    constexpr int ES = sizeof(in_t);
    constexpr int KB = 128;            // bytes of K per row per tile
    constexpr int KE = KB / ES;
    constexpr int OPB = BM * KB;       // 16 KiB per operand tile
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave & 1, wj = wave >> 1;
    // XCD-aware tile order (speed only): workgroups are dealt round-robin over the 8 XCDs, so
    // ids b and b+8 share an L2.  Give every XCD one contiguous range of the tile sequence and
    // walk the tiles in 8-row groups, so the workgroups resident on an XCD at any time share
    // operand rows through its L2.
    int bi = blockIdx.x, bj = blockIdx.y;
    const bool sym_lower = nonsym_flag && *nonsym_flag == 0u;  // uniform
    if (sym_lower) {
        // lower-triangle tiles only, row-major enumeration (bi, bj <= bi); the workgroups beyond
        // the triangle have nothing to do
        const int lin = blockIdx.y * gridDim.x + blockIdx.x;
        const int gm = gridDim.x;
        const int ntri = gm * (gm + 1) / 2;
        if (lin >= ntri) return;
        // workgroup lin runs on XCD lin % 8: give every XCD one contiguous, equally long run of
        // the row-major triangle sequence (balanced, and neighbours in the run share operand
        // panels through that XCD's L2); the last ntri % 8 tiles keep their own number
        const int per = ntri >> 3;
        const int t = (lin < 8 * per) ? (lin & 7) * per + (lin >> 3) : lin;
        int row = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
        while (row * (row + 1) / 2 > t) --row;
        while ((row + 1) * (row + 2) / 2 <= t) ++row;
        bi = row;
        bj = t - row * (row + 1) / 2;
    } else {
        const int gm = gridDim.x, gn = gridDim.y;
        const int nwg = gm * gn;
        if ((nwg & 7) == 0 && (gm & 7) == 0) {
            const int lin = blockIdx.y * gm + blockIdx.x;
            const int swz = (lin & 7) * (nwg >> 3) + (lin >> 3);
            const int per_group = 8 * gn;
            const int grp = swz / per_group, within = swz - grp * per_group;
            bi = grp * 8 + (within & 7);
            bj = within >> 3;
        }
    }
    const int64_t i0 = (int64_t)bi * BM;
    const int64_t j0 = (int64_t)bj * BN;
    const char* Ab = reinterpret_cast<const char*>(Ag + (int64_t)blockIdx.z * strideA + i0 * lda);
    const char* Bb = reinterpret_cast<const char*>(Bg + (int64_t)blockIdx.z * strideB + j0 * ldb);
    out_t* C = Cg + (int64_t)blockIdx.z * strideC;

    // this wave issues instructions ii = wave*4 .. wave*4+3 of each operand tile
    int64_t srcA[4], srcB[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int r = 8 * (wave * 4 + s) + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        srcA[s] = (int64_t)r * lda * ES + c * 16;
        srcB[s] = (int64_t)r * ldb * ES + c * 16;
    }
    auto issue = [&](int buf, int64_t kt) {
        const int64_t kb = kt * KB;
        char* base = smem + buf * 2 * OPB + (wave * 4) * 1024;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            glds16(Ab + srcA[s] + kb, base + s * 1024);
            glds16(Bb + srcB[s] + kb, base + OPB + s * 1024);
        }
    };
    const int64_t nk = k / KE;

    if constexpr (KIND == KIND_I8 || KIND == KIND_F32) {
        typedef typename std::conditional<KIND == KIND_I8, v16i, v16f>::type acc_t;
        acc_t acc[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0;
        const int r32 = lane & 31, h = lane >> 5;
        int rowA[2], rowB[2], swA[2], swB[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            rowA[t] = wi * 64 + t * 32 + r32;
            rowB[t] = wj * 64 + t * 32 + r32;
            swA[t] = (rowA[t] >> 1) & 7;
            swB[t] = (rowB[t] >> 1) & 7;
        }
        issue(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int64_t kt = 0; kt < nk; ++kt) {
            const int buf = (int)(kt & 1);
            if (kt + 1 < nk) issue(buf ^ 1, kt + 1);
            const char* tA = smem + buf * 2 * OPB;
            const char* tB = tA + OPB;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint4 fi[2], fj[2];
                const int ch = 2 * q + h;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    fi[t] = *reinterpret_cast<const uint4*>(tA + rowA[t] * KB + ((ch ^ swA[t]) << 4));
                    fj[t] = *reinterpret_cast<const uint4*>(tB + rowB[t] * KB + ((ch ^ swB[t]) << 4));
                }
                if constexpr (KIND == KIND_I8) {
#pragma unroll
                    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
                        for (int ti = 0; ti < 2; ++ti) {
                            v4i a = {(int)fj[tj].x, (int)fj[tj].y, (int)fj[tj].z, (int)fj[tj].w};
                            v4i b = {(int)fi[ti].x, (int)fi[ti].y, (int)fi[ti].z, (int)fi[ti].w};
                            acc[tj][ti] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[tj][ti], 0, 0, 0);
                        }
                } else {
                    const float* fjf0 = reinterpret_cast<const float*>(&fj[0]);
                    const float* fjf1 = reinterpret_cast<const float*>(&fj[1]);
                    const float* fif0 = reinterpret_cast<const float*>(&fi[0]);
                    const float* fif1 = reinterpret_cast<const float*>(&fi[1]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fjf0[e], fif0[e], acc[0][0], 0, 0, 0);
                        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fjf0[e], fif1[e], acc[0][1], 0, 0, 0);
                        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fjf1[e], fif0[e], acc[1][0], 0, 0, 0);
                        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fjf1[e], fif1[e], acc[1][1], 0, 0, 0);
                    }
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int ti = 0; ti < 2; ++ti) {
                const int64_t ii = i0 + wi * 64 + ti * 32 + r32;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t jj = j0 + wj * 64 + tj * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    C[ii + jj * ldc] = acc[tj][ti][r];
                }
            }
    } else {
        v4d acc[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[a][b][r] = 0.0;
        const int r16 = lane & 15, g = lane >> 4;
        int rowA[4], rowB[4], swA[4], swB[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            rowA[t] = wi * 64 + t * 16 + r16;
            rowB[t] = wj * 64 + t * 16 + r16;
            swA[t] = (rowA[t] >> 1) & 7;
            swB[t] = (rowB[t] >> 1) & 7;
        }
        issue(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int64_t kt = 0; kt < nk; ++kt) {
            const int buf = (int)(kt & 1);
            if (kt + 1 < nk) issue(buf ^ 1, kt + 1);
            const char* tA = smem + buf * 2 * OPB;
            const char* tB = tA + OPB;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                v2d fi[4], fj[4];
                const int ch = 4 * q + g;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    fi[t] = *reinterpret_cast<const v2d*>(tA + rowA[t] * KB + ((ch ^ swA[t]) << 4));
                    fj[t] = *reinterpret_cast<const v2d*>(tB + rowB[t] * KB + ((ch ^ swB[t]) << 4));
                }
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int tj = 0; tj < 4; ++tj)
#pragma unroll
                        for (int ti = 0; ti < 4; ++ti)
                            acc[tj][ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(fj[tj][e], fi[ti][e], acc[tj][ti], 0, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        if constexpr (CMODE == 0) {
#pragma unroll
            for (int tj = 0; tj < 4; ++tj)
#pragma unroll
                for (int ti = 0; ti < 4; ++ti) {
                    const int64_t ii = i0 + wi * 64 + ti * 16 + r16;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int64_t jj = j0 + wj * 64 + tj * 16 + g + 4 * r;
                        C[ii + jj * ldc] = acc[tj][ti][r];
                    }
                }
        } else {  // C -= A'B: batches of 16 independent loads, then the stores
#pragma unroll
            for (int tj = 0; tj < 4; ++tj) {
                out_t cv[4][4];
#pragma unroll
                for (int ti = 0; ti < 4; ++ti)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        cv[ti][r] = C[(i0 + wi * 64 + ti * 16 + r16) + (j0 + wj * 64 + tj * 16 + g + 4 * r) * ldc];
#pragma unroll
                for (int ti = 0; ti < 4; ++ti)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        C[(i0 + wi * 64 + ti * 16 + r16) + (j0 + wj * 64 + tj * 16 + g + 4 * r) * ldc] = cv[ti][r] - acc[tj][ti][r];
            }
        }
    }
}

// ---------------------------------------------------------------------------
// 256 x 256 C tile, 8 waves (2 x 4, a wave owns 128 x 64), int8 and f32 only.  Same LDS-DMA
// staging and swizzle as gemm_tn_dma_kernel; the point of the larger tile is the LDS pipe: with
// 128 x 128 / 4 waves every K-tile costs as many LDS-read cycles (64 KiB at 128 B/clk) as MFMA
// cycles (one wave per SIMD, 16 MFMAs), so the kernel sat at ~30 % of the int8 peak.  Here a K-tile
// is 192 KiB of fragment reads (1536 clk) against 2 waves x 32 MFMAs per SIMD (2048 clk), and
// the L2 -> LDS traffic per MFMA halves.  Two 64 KiB buffers, one workgroup per CU.
// ---------------------------------------------------------------------------
constexpr int BM2 = 256, NT2 = 512;

template <int KIND>
__global__ void __launch_bounds__(NT2)
gemm_tn_dma256_kernel(int64_t k, const typename GemmTraits<KIND>::in_t* __restrict__ Ag, int64_t lda,
                      const typename GemmTraits<KIND>::in_t* __restrict__ Bg, int64_t ldb,
                      typename GemmTraits<KIND>::out_t* __restrict__ Cg, int64_t ldc, int64_t strideA, int64_t strideB,
                      int64_t strideC, const uint32_t* __restrict__ nonsym_flag) {
    static_assert(KIND == KIND_I8 || KIND == KIND_F32, "int8 / f32 only");
    typedef typename GemmTraits<KIND>::in_t in_t;
    typedef typename GemmTraits<KIND>::out_t out_t;
    constexpr int ES = sizeof(in_t);
    constexpr int KB = 128;            // bytes of K per row per tile
    constexpr int KE = KB / ES;
    constexpr int OPB = BM2 * KB;      // 32 KiB per operand tile
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave & 1, wj = wave >> 1;
    int bi = blockIdx.x, bj = blockIdx.y;
    if (nonsym_flag && *nonsym_flag == 0u) {  // lower-triangle tiles only (see gemm_tn_dma_kernel)
        const int lin = blockIdx.y * gridDim.x + blockIdx.x;
        const int gm = gridDim.x;
        const int ntri = gm * (gm + 1) / 2;
        if (lin >= ntri) return;
        // workgroup lin runs on XCD lin % 8: give every XCD one contiguous, equally long run of
        // the row-major triangle sequence (balanced, and neighbours in the run share operand
        // panels through that XCD's L2); the last ntri % 8 tiles keep their own number
        const int per = ntri >> 3;
        const int t = (lin < 8 * per) ? (lin & 7) * per + (lin >> 3) : lin;
        int row = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
        while (row * (row + 1) / 2 > t) --row;
        while ((row + 1) * (row + 2) / 2 <= t) ++row;
        bi = row;
        bj = t - row * (row + 1) / 2;
    } else {
        // XCD-aware order (see gemm_tn_dma_kernel): the 32 workgroups resident on an XCD form an
        // 8 x 4 cluster of tiles and share operand panels through that XCD's L2
        const int gm = gridDim.x, gn = gridDim.y;
        const int nwg = gm * gn;
        if ((nwg & 7) == 0 && (gm & 7) == 0) {
            const int lin = blockIdx.y * gm + blockIdx.x;
            const int swz = (lin & 7) * (nwg >> 3) + (lin >> 3);
            const int per_group = 8 * gn;
            const int grp = swz / per_group, within = swz - grp * per_group;
            bi = grp * 8 + (within & 7);
            bj = within >> 3;
        }
    }
    const int64_t i0 = (int64_t)bi * BM2;
    const int64_t j0 = (int64_t)bj * BM2;
    const char* Ab = reinterpret_cast<const char*>(Ag + (int64_t)blockIdx.z * strideA + i0 * lda);
    const char* Bb = reinterpret_cast<const char*>(Bg + (int64_t)blockIdx.z * strideB + j0 * ldb);
    out_t* C = Cg + (int64_t)blockIdx.z * strideC;

    // an operand tile is 32 DMA instructions of 1 KiB (8 rows x 128 B); this wave issues 4 of them
    int64_t srcA[4], srcB[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int r = 8 * (wave * 4 + s) + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        srcA[s] = (int64_t)r * lda * ES + c * 16;
        srcB[s] = (int64_t)r * ldb * ES + c * 16;
    }
    auto issue = [&](int buf, int64_t kt) {
        const int64_t kb = kt * KB;
        char* base = smem + buf * 2 * OPB + (wave * 4) * 1024;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            glds16(Ab + srcA[s] + kb, base + s * 1024);
            glds16(Bb + srcB[s] + kb, base + OPB + s * 1024);
        }
    };
    const int64_t nk = k / KE;
    typedef typename std::conditional<KIND == KIND_I8, v16i, v16f>::type acc_t;
    acc_t acc[2][4];  // [tj: 32-col tiles of B side][ti: 32-row tiles of A side]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0;
    const int r32 = lane & 31, h = lane >> 5;
    int rowA[4], rowB[2], swA[4], swB[2];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        rowA[t] = wi * 128 + t * 32 + r32;
        swA[t] = (rowA[t] >> 1) & 7;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        rowB[t] = wj * 64 + t * 32 + r32;
        swB[t] = (rowB[t] >> 1) & 7;
    }
    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int64_t kt = 0; kt < nk; ++kt) {
        const int buf = (int)(kt & 1);
        if (kt + 1 < nk) issue(buf ^ 1, kt + 1);
        const char* tA = smem + buf * 2 * OPB;
        const char* tB = tA + OPB;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint4 fi[4], fj[2];
            const int ch = 2 * q + h;
#pragma unroll
            for (int t = 0; t < 4; ++t) fi[t] = *reinterpret_cast<const uint4*>(tA + rowA[t] * KB + ((ch ^ swA[t]) << 4));
#pragma unroll
            for (int t = 0; t < 2; ++t) fj[t] = *reinterpret_cast<const uint4*>(tB + rowB[t] * KB + ((ch ^ swB[t]) << 4));
            if constexpr (KIND == KIND_I8) {
#pragma unroll
                for (int tj = 0; tj < 2; ++tj)
#pragma unroll
                    for (int ti = 0; ti < 4; ++ti) {
                        v4i a = {(int)fj[tj].x, (int)fj[tj].y, (int)fj[tj].z, (int)fj[tj].w};
                        v4i b = {(int)fi[ti].x, (int)fi[ti].y, (int)fi[ti].z, (int)fi[ti].w};
                        acc[tj][ti] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[tj][ti], 0, 0, 0);
                    }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
                        for (int ti = 0; ti < 4; ++ti)
                            acc[tj][ti] = __builtin_amdgcn_mfma_f32_32x32x2f32(reinterpret_cast<const float*>(&fj[tj])[e],
                                                                               reinterpret_cast<const float*>(&fi[ti])[e],
                                                                               acc[tj][ti], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) {
            const int64_t ii = i0 + wi * 128 + ti * 32 + r32;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t jj = j0 + wj * 64 + tj * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                C[ii + jj * ldc] = acc[tj][ti][r];
            }
        }
}


// Dynamic-LDS limits are a per-device property of a kernel: sdpsr_create() calls this with the
// ctx's device current, so a process may hold ctxs on several GPUs (no process-global flags).
template <int KIND>
bool gemm_set_attributes_kind() {
    bool ok = true;
    constexpr int KBt = GemmTraits<KIND>::KB;
    ok &= hipSuccess == hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_kernel<KIND>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 2 * BM * (KBt + 16));
    ok &= hipSuccess == hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_dma_kernel<KIND, 0>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 2 * BM * 128);
    if constexpr (KIND == KIND_F64)
        ok &= hipSuccess == hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_dma_kernel<KIND, 1>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 2 * BM * 128);
    if constexpr (KIND == KIND_I8 || KIND == KIND_F32)
        ok &= hipSuccess == hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tn_dma256_kernel<KIND>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 2 * BM2 * 128);
    return ok;
}
bool gemm_set_device_attributes() {
    bool ok = true;
    ok &= gemm_set_attributes_kind<KIND_I8>();
    ok &= gemm_set_attributes_kind<KIND_F32>();
    ok &= gemm_set_attributes_kind<KIND_F64>();
    return ok;
}

template <int KIND>
static void launch_gemm(hipStream_t s, int64_t m, int64_t n, int64_t k,
                        const typename GemmTraits<KIND>::in_t* A, int64_t lda,
                        const typename GemmTraits<KIND>::in_t* B, int64_t ldb,
                        typename GemmTraits<KIND>::out_t* C, int64_t ldc, int batch,
                        int64_t strideA, int64_t strideB, int64_t strideC, const uint32_t* nonsym_flag = nullptr,
                        int cmode = 0) {
    constexpr int KB = GemmTraits<KIND>::KB;
    constexpr size_t lds = 2 * 2 * BM * (KB + 16);
    dim3 grid((unsigned)(m / BM), (unsigned)(n / BN), (unsigned)batch);
    constexpr int ESZ = sizeof(typename GemmTraits<KIND>::in_t);
    if ((k * ESZ) % 128 == 0 && ((lda * ESZ) % 16) == 0 && ((ldb * ESZ) % 16) == 0 &&
        ((strideA * ESZ) % 16) == 0 && ((strideB * ESZ) % 16) == 0 &&
        (reinterpret_cast<uintptr_t>(A) % 16) == 0 && (reinterpret_cast<uintptr_t>(B) % 16) == 0) {
        if constexpr (KIND == KIND_I8 || KIND == KIND_F32) {
            // 256 x 256 tiles pay off (6-15 % measured) once the launch has >= 4 workgroups per CU
            // (one resident workgroup per CU: fewer leave a ragged last round)
            const int64_t t2 = m / BM2;
            const int64_t wgs = (nonsym_flag ? t2 * (t2 + 1) / 2 : t2 * (n / BM2)) * batch;
            if (m % BM2 == 0 && n % BM2 == 0 && wgs >= 1024) {
                constexpr size_t lds256 = 2 * 2 * BM2 * 128;  // 128 KiB
                dim3 grid2((unsigned)(m / BM2), (unsigned)(n / BM2), (unsigned)batch);
                gemm_tn_dma256_kernel<KIND><<<grid2, NT2, lds256, s>>>(k, A, lda, B, ldb, C, ldc, strideA, strideB, strideC, nonsym_flag);
                return;
            }
        }
        constexpr size_t lds_dma = 2 * 2 * BM * 128;  // 64 KiB
        if constexpr (KIND == KIND_F64) {
            if (cmode) {
                gemm_tn_dma_kernel<KIND, 1><<<grid, NT, lds_dma, s>>>(k, A, lda, B, ldb, C, ldc, strideA, strideB, strideC, nonsym_flag);
                return;
            }
        }
        gemm_tn_dma_kernel<KIND, 0><<<grid, NT, lds_dma, s>>>(k, A, lda, B, ldb, C, ldc, strideA, strideB, strideC, nonsym_flag);
        return;
    }
    gemm_tn_kernel<KIND><<<grid, NT, lds, s>>>(k, A, lda, B, ldb, C, ldc, strideA, strideB, strideC, nonsym_flag, cmode);
}

void launch_gemm_tn_i8(hipStream_t s, int64_t m, int64_t n, int64_t k, const int8_t* A,
                       int64_t lda, const int8_t* B, int64_t ldb, int32_t* C, int64_t ldc,
                       int batch, int64_t strideA, int64_t strideB, int64_t strideC) {
    launch_gemm<KIND_I8>(s, m, n, k, A, lda, B, ldb, C, ldc, batch, strideA, strideB, strideC);
}
void launch_gemm_tn_f32(hipStream_t s, int64_t m, int64_t n, int64_t k, const float* A,
                        int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc,
                        int batch, int64_t strideA, int64_t strideB, int64_t strideC) {
    launch_gemm<KIND_F32>(s, m, n, k, A, lda, B, ldb, C, ldc, batch, strideA, strideB, strideC);
}
void launch_gemm_tn_f64(hipStream_t s, int64_t m, int64_t n, int64_t k, const double* A,
                        int64_t lda, const double* B, int64_t ldb, double* C, int64_t ldc,
                        int batch, int64_t strideA, int64_t strideB, int64_t strideC) {
    launch_gemm<KIND_F64>(s, m, n, k, A, lda, B, ldb, C, ldc, batch, strideA, strideB, strideC);
}
// C -= A' * B (the compact-WY updates of the back-transformation)
void launch_gemm_tn_f64_sub(hipStream_t s, int64_t m, int64_t n, int64_t k, const double* A, int64_t lda, const double* B,
                            int64_t ldb, double* C, int64_t ldc) {
    launch_gemm<KIND_F64>(s, m, n, k, A, lda, B, ldb, C, ldc, 1, 0, 0, 0, nullptr, 1);
}

// C = X'X with only the lower-triangle tiles computed while *nonsym_flag == 0 (device-side
// decision, no host round trip); the consumer must then read C[i,j] for i >= j only
void launch_gemm_tn_i8_sym(hipStream_t s, int64_t n, int64_t k, const int8_t* X, int64_t ldx, int32_t* C, int64_t ldc,
                           int batch, int64_t strideX, int64_t strideC, const uint32_t* nonsym_flag, int num_cus, int variant) {
    // the persistent 256 x 256 launch of kernels_gemm_sym.hip when the shapes allow it (num_cus = 0: never)
    if (nonsym_flag && variant != 1 &&
        launch_i8_symsquare(s, n, k, X, ldx, C, ldc, batch, strideX, strideC, nonsym_flag, num_cus, variant))
        return;
    launch_gemm<KIND_I8>(s, n, n, k, X, ldx, X, ldx, C, ldc, batch, strideX, strideX, strideC, nonsym_flag);
}
void launch_gemm_tn_f32_sym(hipStream_t s, int64_t n, int64_t k, const float* X, int64_t ldx, float* C, int64_t ldc,
                            int batch, int64_t strideX, int64_t strideC, const uint32_t* nonsym_flag) {
    launch_gemm<KIND_F32>(s, n, n, k, X, ldx, X, ldx, C, ldc, batch, strideX, strideX, strideC, nonsym_flag);
}


}  // namespace sdpsr
