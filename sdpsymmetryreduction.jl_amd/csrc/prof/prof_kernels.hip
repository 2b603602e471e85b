// Measurement kernels of libsdpsr_prof.so (not in the product library).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../sdpsr_hash.h"

namespace sdpsr {

// ---------------------------------------------------------------------------
// Shader-clock meter (diagnostic, sdpsr_profile_clock): one wave on a side stream samples
// (clock64 = shader cycles, wall_clock64 = 100 MHz) every ~20 us while the kernels under test run
// on the main stream.  The int8 squares are power-limited on this part: the clock they run at,
// not their cycle count, decides the rate (measured: 2.1-2.2 GHz under MFMAs alone, ~1.4 GHz
// under the full kernel).  Terminates on the flag, after ns samples or after max_ticks.
// ---------------------------------------------------------------------------
__global__ void clock_sampler_kernel(long long* __restrict__ buf, int ns, const unsigned* flag, long long max_ticks,
                                     int* __restrict__ count) {
    if (threadIdx.x != 0) return;
    const long long w0 = wall_clock64();
    long long next = w0;
    int i = 0;
    while (i < ns) {
        const long long w = wall_clock64();
        if (w >= next) {
            buf[2 * i] = clock64();
            buf[2 * i + 1] = w;
            ++i;
            next = w + 2000;
        }
        if (__builtin_nontemporal_load(flag) != 0u) break;
        if (w - w0 > max_ticks) break;
        __builtin_amdgcn_s_sleep(16);
    }
    *count = i;
}
__global__ void wall_marker_kernel(long long* out) { *out = wall_clock64(); }

void launch_clock_sampler(hipStream_t s, long long* buf, int ns, const unsigned* flag, long long max_ticks, int* count) {
    clock_sampler_kernel<<<1, 64, 0, s>>>(buf, ns, flag, max_ticks, count);
}
void launch_wall_marker(hipStream_t s, long long* out) { wall_marker_kernel<<<1, 1, 0, s>>>(out); }

// synthetic signatures with `nclasses` distinct non-zero values (measurement hook)
__global__ void fill_test_sig_kernel(int64_t len, int64_t nclasses, uint64_t* __restrict__ sig) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < len; e += stride) {
        uint64_t cls = sdpsr_fmix64((uint64_t)e * 0x9E3779B97F4A7C15ULL + 17) % (uint64_t)nclasses;
        sig[e] = sdpsr_fmix64(cls + 1) | 1ull;
    }
}
void launch_fill_test_sig(hipStream_t s, int64_t len, int64_t nclasses, uint64_t* sig) {
    fill_test_sig_kernel<<<(int)((len + 255) / 256 < 2048 ? (len + 255) / 256 : 2048), 256, 0, s>>>(len, nclasses, sig);
}



// Stage 1 of the two-stage form (prof.cpp: sdpsr_profile_band_reduce): after the panel's QR, V = the reflectors with their
// unit diagonal and zeros above written out (what the level-3 updates take), and the panel keeps only R.
__global__ void band_panel_split_kernel(int64_t m, int b, double* __restrict__ P, int64_t ldp, double* __restrict__ V, int64_t ldv) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    if (i >= m || j >= b) return;
    const double v = P[i + (int64_t)j * ldp];
    V[i + (int64_t)j * ldv] = i > j ? v : (i == j ? 1.0 : 0.0);
    if (i > j) P[i + (int64_t)j * ldp] = 0.0;
}
void launch_band_panel_split(hipStream_t s, int64_t m, int b, double* P, int64_t ldp, double* V, int64_t ldv) {
    band_panel_split_kernel<<<dim3((unsigned)((m + 255) / 256), (unsigned)b), 256, 0, s>>>(m, b, P, ldp, V, ldv);
}

// ---------------------------------------------------------------------------
// Stage 2 of a two-stage tridiagonalisation, built to be MEASURED ("costed, not built"): symmetric band
// (bandwidth b) -> tridiagonal by Householder bulge chasing (Bischof / Lang / Sun's SBR scheme; LAPACK's dsytrd_sb2st runs
// the same sweeps).  Sweep j annihilates column j below the first subdiagonal with a reflector on rows j+1 .. j+b and
// chases the bulge down the band: step k works on the reflector rows I_k = j+1+kb .. j+(k+1)b and touches three b x b
// blocks -- L = A(I_k, I_{k-1}) (reflector from its first column, applied from the left), D = A(I_k, I_k) (two-sided),
// R = A(I_{k+1}, I_k) (from the right: the next bulge).  Step k of sweep j + 1 overlaps the blocks of steps k .. k+2 of
// sweep j, so sweep j + 1 runs three steps behind sweep j: a chain of ~3 n dependent steps whatever the parallelism.
// One workgroup per sweep (workgroup w: sweeps w, w + G, ...; all G resident), the hand-off through a progress word per
// sweep.  Every access to the matrix and the progress words is an agent-scope relaxed atomic (sc1: served by the memory
// side, coherent across XCDs without fences -- MI355X_MICROARCH.md's "sc1 stores and loads both sides" form; the
// fence form costs 1.7 + 1.7 us per hop on this part).  Dense n x n storage (lower triangle referenced), results d, e.
// A poll that lasts longer than ~2 s raises `abort_flag` and every workgroup leaves: no hang whatever goes wrong.
// ---------------------------------------------------------------------------
constexpr int BC_T = 256;
constexpr unsigned BC_DONE = 0x7FFFFFFFu;
__device__ __forceinline__ double bc_ld(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void bc_st(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int B>
__global__ void __launch_bounds__(BC_T)
band_chase_kernel(int n, double* __restrict__ A, int64_t ld, unsigned* __restrict__ progress, unsigned* __restrict__ abort_flag,
                  double* __restrict__ dout, double* __restrict__ eout) {
    extern __shared__ __attribute__((aligned(16))) double bc_smem[];  // three (B x (B + 1)) blocks
    double (*Lb)[B + 1] = reinterpret_cast<double (*)[B + 1]>(bc_smem);
    double (*Db)[B + 1] = reinterpret_cast<double (*)[B + 1]>(bc_smem + B * (B + 1));
    double (*Rb)[B + 1] = reinterpret_cast<double (*)[B + 1]>(bc_smem + 2 * B * (B + 1));
    __shared__ double v[B], wv[B], red[BC_T / 64];
    __shared__ double s_tau, s_beta;
    __shared__ int s_go;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    auto block_sum = [&](double x) -> double {  // sum over the workgroup, result in every thread
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
        __syncthreads();
        if (lane == 0) red[wave] = x;
        __syncthreads();
        double s = 0;
        for (int k = 0; k < BC_T / 64; ++k) s += red[k];
        return s;
    };
    for (int j = blockIdx.x; j < n - 2; j += gridDim.x) {
        const int r0 = j + 1;
        for (int k = 0;; ++k) {
            const int i0 = r0 + k * B;             // first reflector row
            if (i0 > n - 1) break;
            const int m = (n - i0 < B) ? n - i0 : B;  // reflector length
            if (m < 2) break;                      // a single row: nothing to annihilate, nothing to chase
            // ---- wait for sweep j - 1 to be three steps ahead ----
            if (j > 0) {
                if (t == 0) {
                    int go = 1;
                    const long long t0 = wall_clock64();
                    for (;;) {
                        const unsigned p = __hip_atomic_load(&progress[j - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (p >= (unsigned)(k + 3)) break;
                        if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { go = 0; break; }
                        if (wall_clock64() - t0 > 200000000ll) {  // 2 s at 100 MHz
                            __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            go = 0;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(2);
                    }
                    s_go = go;
                }
                __syncthreads();
                if (!s_go) return;
            }
            const int c0 = (k == 0) ? j : i0 - B;   // first column of the left block (k = 0: the single column j)
            const int mc = (k == 0) ? 1 : B;        // its columns
            const int i1 = i0 + m;                  // first row below
            const int mr = (n - i1 < B) ? (n - i1 > 0 ? n - i1 : 0) : B;  // rows of the block below
            // ---- load L (m x mc), D (m x m, lower triangle mirrored), R (mr x m).  From the second step on L is the R block
            // the previous step of this sweep left in LDS (it is still stored: later sweeps read it from memory) ----
            if (k == 0) {
                for (int e = t; e < m * mc; e += BC_T) {
                    const int r = e % m, cc = e / m;
                    Lb[r][cc] = bc_ld(&A[(int64_t)(i0 + r) + (int64_t)(c0 + cc) * ld]);
                }
            }
            for (int e = t; e < m * m; e += BC_T) {
                const int r = e % m, cc = e / m;
                if (r >= cc) {
                    const double x = bc_ld(&A[(int64_t)(i0 + r) + (int64_t)(i0 + cc) * ld]);
                    Db[r][cc] = x;
                    Db[cc][r] = x;
                }
            }
            for (int e = t; e < mr * m; e += BC_T) {
                const int r = e % mr, cc = e / mr;
                Rb[r][cc] = bc_ld(&A[(int64_t)(i1 + r) + (int64_t)(i0 + cc) * ld]);
            }
            __syncthreads();
            // ---- reflector from the first column of L: H x = beta e1, H = I - tau v v', v(0) = 1 ----
            double part = 0;
            for (int r = 1 + t; r < m; r += BC_T) part += Lb[r][0] * Lb[r][0];
            const double xnorm2 = block_sum(part);
            if (t == 0) {
                const double alpha = Lb[0][0];
                if (xnorm2 == 0.0) {
                    s_tau = 0.0;
                    s_beta = alpha;
                } else {
                    const double beta = -copysign(sqrt(alpha * alpha + xnorm2), alpha);
                    s_tau = (beta - alpha) / beta;
                    s_beta = beta;
                    wv[0] = 1.0 / (alpha - beta);
                }
            }
            __syncthreads();
            const double tau = s_tau;
            if (t < m) v[t] = (t == 0) ? 1.0 : (tau != 0.0 ? Lb[t][0] * wv[0] : 0.0);
            __syncthreads();
            if (tau != 0.0) {
                // L <- H L: column 0 becomes (beta, 0, ...); the other columns: c -= tau v (v'c)
                for (int cc = 1 + wave; cc < mc; cc += BC_T / 64) {
                    double dsum = 0;
                    for (int r = lane; r < m; r += 64) dsum += v[r] * Lb[r][cc];
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) dsum += __shfl_xor(dsum, o, 64);
                    for (int r = lane; r < m; r += 64) Lb[r][cc] -= tau * dsum * v[r];
                }
                if (t < m) Lb[t][0] = (t == 0) ? s_beta : 0.0;
                // D <- H D H: w = tau D v, w += -(tau/2)(w'v) v, D -= v w' + w v'
                if (t < m) {
                    double sum = 0;
                    for (int cc = 0; cc < m; ++cc) sum += Db[t][cc] * v[cc];
                    wv[t] = tau * sum;
                }
                __syncthreads();
                double pv = (t < m) ? wv[t] * v[t] : 0.0;
                const double wdotv = block_sum(pv);
                if (t < m) wv[t] -= 0.5 * tau * wdotv * v[t];
                __syncthreads();
                for (int e = t; e < m * m; e += BC_T) {
                    const int r = e % m, cc = e / m;
                    Db[r][cc] -= v[r] * wv[cc] + wv[r] * v[cc];
                }
                // R <- R H: row r: R(r, :) -= tau (R(r, :) v) v'
                for (int r = t; r < mr; r += BC_T) {
                    double sum = 0;
                    for (int cc = 0; cc < m; ++cc) sum += Rb[r][cc] * v[cc];
                    sum *= tau;
                    for (int cc = 0; cc < m; ++cc) Rb[r][cc] -= sum * v[cc];
                }
                __syncthreads();
            }
            // ---- store (lower triangle of D) ----
            for (int e = t; e < m * mc; e += BC_T) {
                const int r = e % m, cc = e / m;
                bc_st(&A[(int64_t)(i0 + r) + (int64_t)(c0 + cc) * ld], Lb[r][cc]);
            }
            for (int e = t; e < m * m; e += BC_T) {
                const int r = e % m, cc = e / m;
                if (r >= cc) bc_st(&A[(int64_t)(i0 + r) + (int64_t)(i0 + cc) * ld], Db[r][cc]);
            }
            for (int e = t; e < mr * m; e += BC_T) {
                const int r = e % mr, cc = e / mr;
                bc_st(&A[(int64_t)(i1 + r) + (int64_t)(i0 + cc) * ld], Rb[r][cc]);
            }
            // every storing wave's stores have left before the progress word moves
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t == 0) __hip_atomic_store(&progress[j], (unsigned)(k + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            {  // this step's R is the next step's L
                double (*tmp)[B + 1] = Lb;
                Lb = Rb;
                Rb = tmp;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t == 0) __hip_atomic_store(&progress[j], BC_DONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    (void)dout;
    (void)eout;
}
__global__ void band_extract_kernel(int n, const double* __restrict__ A, int64_t ld, double* __restrict__ d, double* __restrict__ e) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) d[i] = A[(int64_t)i + (int64_t)i * ld];
    if (i < n - 1) e[i] = A[(int64_t)(i + 1) + (int64_t)i * ld];
}
// returns false for an unsupported bandwidth
bool launch_band_chase(hipStream_t s, int n, int b, double* A, int64_t ld, unsigned* progress, unsigned* abort_flag, double* d, double* e,
                       int max_wgs) {
    if (n < 3) return true;
    const int G = (n - 2 < max_wgs) ? n - 2 : max_wgs;
    if (b == 16) band_chase_kernel<16><<<G, BC_T, 3 * 16 * 17 * 8, s>>>(n, A, ld, progress, abort_flag, d, e);
    else if (b == 32) band_chase_kernel<32><<<G, BC_T, 3 * 32 * 33 * 8, s>>>(n, A, ld, progress, abort_flag, d, e);
    else if (b == 64) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&band_chase_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 64 * 65 * 8) != hipSuccess) return false;
        band_chase_kernel<64><<<G, BC_T, 3 * 64 * 65 * 8, s>>>(n, A, ld, progress, abort_flag, d, e);
    } else return false;
    band_extract_kernel<<<(n + 255) / 256, 256, 0, s>>>(n, A, ld, d, e);
    return true;
}

}  // namespace sdpsr
