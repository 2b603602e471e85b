// Householder tridiagonalisation A = Q T Q' of the generic algebra element for gfx950
// (first phase of eigen(A), src/eigen_decomposition.jl:246; LAPACK dsytrd/dlatrd, lower).
//
// Design (MI355X-first, not a translation of dlatrd's BLAS-2 call sequence):
//  * only the LOWER triangle of the trailing matrix is stored, read and updated.  The big
//    product p = A v of every column reads each stored entry ONCE and uses it twice
//    (p_r += a_rc v_c and p_c += a_rc v_r): half the HBM bytes of a full-matrix product, and the
//    two triangles can never drift apart in rounding.  Tiles of 128 rows x 64 columns, a
//    workgroup walks down a 64-column strip; the "direct" sums (rows of the tile) leave per
//    tile, the "transposed" sums (columns of the strip) stay in registers for the whole walk.
//    Every partial sum goes to its own slot (no atomics, bitwise reproducible); the next
//    launch adds them up;
//  * two launches per column, both spread over the whole chip:
//      form_kernel (j):   sum the partial products of column j-1, finish W(:, j-1) (corrections
//                         with the panel V, W), then form the updated column
//                         a_j = A(:,j) - V W(j,:)' - W V(j,:)' and its partial squared norms;
//      symv_kernel (j):   every workgroup redundantly finishes the reflector scalars
//                         (beta, tau, scale) from the partial norms and computes its share of
//                         p = A v, W'v, V'v and v'p  (v is never materialised: v_r = a_rj * scale);
//    a kernel boundary (~1.5 us) is the cheapest chip-wide synchronisation on this part, cheaper
//    than any in-kernel grid barrier (4-5 us), so the launches are built into a hipGraph once
//    per shape and replayed;
//  * the panel [V | W] is kept row-major (64 doubles per row), which is at once the coalesced
//    layout for the per-row corrections and the K-contiguous operand layout of the MFMA kernel;
//  * after NB = 32 columns the trailing matrix gets the rank-2NB update A -= V W' + W V' on the
//    matrix cores: v_mfma_f64_16x16x4_f64, 128 x 128 tiles of the lower triangle, K = 64, both
//    operands streamed global -> LDS directly (same staging as kernels_gemm.hip).
// Output is LAPACK-compatible (d, e, tau, reflectors below the subdiagonal of A), so the
// tridiagonal solve (rocSOLVER stedc) and the back-transformation plug in unchanged.
#include <cstdlib>
#include <chrono>
#include <cstdio>
#include <vector>
#include <cstdlib>

#include "sdpsr_internal.h"
#include "jacobi64.h"

namespace sdpsr {

// development aid (-DLK_TIMING, and SDPSR_DEBUG set at run time; launches one by one: SDPSR_FLAG_NO_GRAPH): wall-clock
// stamps (100 MHz) of the phases of the form kernel's first / middle / last workgroup
#ifdef LK_TIMING
bool dbg_on();  // ctx.cpp (SDPSR_DEBUG)
__device__ long long* sy_dbg = nullptr;
#define SY_STAMP(i)                                                                                                   \
    do {                                                                                                               \
        if (sy_dbg && threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1 || blockIdx.x == gridDim.x / 2)) { \
            const int which = blockIdx.x == 0 ? 0 : (blockIdx.x == gridDim.x - 1 ? 2 : 1);                            \
            sy_dbg[((int64_t)j * 3 + which) * 16 + (i)] = wall_clock64();                                              \
        }                                                                                                              \
    } while (0)
#else
#define SY_STAMP(i)
#endif

constexpr int SY_NB = 32;            // panel width
constexpr int SY_PW = 2 * SY_NB;     // doubles per panel row: [V(r, 0..NB) | W(r, 0..NB)]
constexpr int SY_THREADS = 256;
constexpr int SY_TR = 128;           // tile rows of the symmetric product
constexpr int SY_TC = 64;            // strip width (tile columns)
constexpr int SY_FROWS = 64;         // rows per workgroup of the form kernel
constexpr int SY_MAXSEG = 40;        // segments of a strip (rows of the transposed-partials buffer)
constexpr int SY_MAXVAV = 8704;      // slots of per-workgroup v'Av partials (one-launch form: 2 parities x 2 per tile, 64 blocks)

// wave-wide sum on the DPP path (no LDS round trips): every lane ends with the same bits
template <int CTRL>
__device__ __forceinline__ double sr_dpp(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double sr_readlane(double x, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l), __builtin_amdgcn_readlane(__double2loint(x), l));
}
__device__ __forceinline__ double sr_wave_sum(double x) {
    x += sr_dpp<0xB1>(x);   // quad_perm [1,0,3,2]
    x += sr_dpp<0x4E>(x);   // quad_perm [2,3,0,1]
    x += sr_dpp<0x124>(x);  // row_ror:4
    x += sr_dpp<0x128>(x);  // row_ror:8: every lane holds the sum of its row of 16
    return (sr_readlane(x, 0) + sr_readlane(x, 16)) + (sr_readlane(x, 32) + sr_readlane(x, 48));
}

struct SytrdArgs {
    double* A;          // n x n, leading dimension ld (multiple of 128), lower triangle referenced
    int64_t ld;
    int n;
    double* PT;         // ld x SY_PW, row-major panel
    double* PTc;        // SY_PW x ld: the same panel column-major (what a row's owner thread of the form kernel reads)
    double* Pdir;       // (ld / 64) x ld: direct partial products, one row per strip
    double* Ptr;        // SY_MAXSEG x ld: transposed partial products, one row per segment
    double* Gpart;      // (ld / 64 + 1) x SY_PW: partial panel dots [V'v | W'v], one row per 64 matrix rows
    double* part_norm;  // per form-workgroup partial sum of a[r]^2, r >= j+2
    double* part_vav;   // per symv-workgroup partial of v'(A v)
    double* scal;       // [0] tau, [1] beta, [2] scale of the reflector in flight
    double* d;
    double* e;
    double* tau;
};

// geometry of the symv launch of column j (shared by the launch and by the form kernel that sums
// its partials)
struct SymvGeom {
    int S0;      // first strip with a column > j
    int nstr;    // strips S0 .. S0 + nstr - 1
    int SEG;     // tiles per workgroup
    int maxseg;  // segments of the longest strip
    int G;       // panel-dot workgroups (0 when the panel is empty)
    int grid;    // nstr * maxseg tile workgroups (those past a strip's last segment idle) + G
};
static SymvGeom symv_geometry(int n, int j, int cf) {
    SymvGeom g;
    const int nt128 = (n + SY_TR - 1) / SY_TR;
    const int nstrips = (n + SY_TC - 1) / SY_TC;
    g.S0 = (j + 1) / SY_TC;
    g.nstr = nstrips - g.S0;
    long tiles = 0;
    for (int S = g.S0; S < nstrips; ++S) tiles += nt128 - (S >> 1);
    // ~1.5 workgroups per CU, each streaming `seg` tiles back to back (round 3, N = 4096, whole tridiagonalisation in the
    // panel form with 256 / 384 / 512 / 768 / 1024 / 2048 workgroups aimed at: 72.8 / 72.1 / 74.2 / 75.9 / 77.3 / 77.7 ms)
    int seg = (int)((tiles + 383) / 384);
    if (seg < 1) seg = 1;
    if (seg > 8) seg = 8;
    g.SEG = seg;
    g.maxseg = (nt128 - (g.S0 >> 1) + seg - 1) / seg;
    const int m = n - j - 1;
    g.G = cf > 0 ? (m + 63) / 64 : 0;  // 64 rows per panel-dot workgroup
    g.grid = g.nstr * g.maxseg + g.G;
    return g;
}

// ---------------------------------------------------------------------------
// form_kernel: rows r in [j, n); a workgroup owns SY_FROWS = 64 rows.
//   do_finish: column jf = j-1 (panel column cf) gets its W column: the partial products of
//              symv(jf) are summed here (geometry S0 / SEG / G of that launch).
//   do_form:   column j is updated with the finished panel columns and its norm partials
//              are produced.  n_vav = number of part_vav entries written by symv(jf).
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(SY_THREADS)
sytrd_form_kernel(SytrdArgs a, int j, int cf, int do_finish, int do_form, int n_vav, int S0, int SEG, int G) {
    __shared__ double s_raw[SY_PW], s_Gf[SY_PW], s_Mf[SY_PW], s_graw[4][SY_PW];
    __shared__ double s_psum[4][SY_FROWS], s_sf[4][SY_FROWS], s_sm[4][SY_FROWS];
    __shared__ double s_scal[4];  // tau, alpha2, p0[j], sf of row j
    __shared__ double s_vavw[4];
    const int tid = threadIdx.x;
    const int lane = tid & 63, q = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: scalar bases, 32-bit lane offsets
    const int64_t ld = a.ld;
    const int n = a.n;
    const int jf = j - 1;
    const int rbase = j + blockIdx.x * SY_FROWS;
    const int r = rbase + lane;
    const int nt128 = (n + SY_TR - 1) / SY_TR;

    SY_STAMP(0);
    // ---- loads first, ALL of them, into registers, in batches of a static shape: a value that is consumed (or a loop whose
    // trip count the compiler does not know) between two loads makes the later load wait for a memory round trip of its
    // own, and a round trip is ~1 us on the critical path of the column (round 4: this kernel had seven of them in a row
    // before its first barrier, 8.3 us per launch).  Orders beyond 4096 finish the longer sums in loops behind the batch.
    const double pre_tau = do_finish ? a.scal[0] : 0.0, pre_scale = do_finish ? a.scal[2] : 0.0;
    const bool rok = r < n;
    const int rr = rok ? r : j;  // (an address that exists)
    double pre_raw = 0, pre_x = 0, mf = 0, pjk = 0, x1 = 0, x2 = 0;
    double tvd[16], tvt[4], tvg[16], tvv[8], pcv[8], pcw[8];
    const int Sr = rr / SY_TC;
    const int cnt = (do_finish && rok && Sr >= S0 + q) ? (Sr - S0 - q) / 4 + 1 : 0;   // slots of direct partial products of this wave
    const int nseg = (nt128 - (Sr >> 1) + SEG - 1) / SEG;
    const int cnt2 = (do_finish && rok && nseg > q) ? (nseg - 1 - q) / 4 + 1 : 0;      // ... of transposed ones
    const int gk = lane, gpart = q;
    if (q == 0) {
        if (do_finish) pre_raw = a.A[rr + (int64_t)jf * ld];
        if (do_form) pre_x = a.A[rr + (int64_t)j * ld];
    }
    if (do_finish) {
        // (slots past the last one are read too, and dropped: the addresses stay inside the workspace -- Pdir, Ptr and the
        // records are followed by the 64 ld doubles of the column-major panel -- and plain strides cost no address arithmetic)
        {
            const double* pd = a.Pdir + (int64_t)(S0 + q) * ld + rr;
            const double* pt = a.Ptr + (int64_t)q * ld + rr;
            const int64_t st = 4 * ld;
#pragma unroll
            for (int u = 0; u < 16; ++u) tvd[u] = pd[u * st];
#pragma unroll
            for (int u = 0; u < 4; ++u) tvt[u] = pt[u * st];
        }
        {
            // (64 rows: Gpart has ld / 64 + 1; at smaller orders the rows past G run into the records behind it)
            const double* pg = a.Gpart + gpart * SY_PW + gk;
#pragma unroll
            for (int u = 0; u < 16; ++u) tvg[u] = pg[u * 4 * SY_PW];
        }
        // v'Av partials: all 256 threads
#pragma unroll
        for (int u = 0; u < 8; ++u) tvv[u] = a.part_vav[64 * q + SY_THREADS * u + lane];  // (SY_MAXVAV >= 2048 slots exist)
        if (q == 3) {  // row j: product partials
            const int Sj = j / SY_TC;
            const int nsegj = (nt128 - (Sj >> 1) + SEG - 1) / SEG;
            x1 = a.Pdir[(int64_t)(S0 + lane) * ld + j];
            x2 = a.Ptr[(int64_t)lane * ld + j];
        }
    }
    if (do_form && tid < SY_PW)  // row j of the panel with the halves swapped: [W(j,:) | V(j,:)]
        mf = a.PT[(int64_t)j * SY_PW + (tid ^ SY_NB)];
    // the panel entries of row r from the column-major copy: a thread owns a row (coalesced loads, no reduction across
    // lanes), wave q takes the panel columns 8q .. 8q+7 of V and of W
    if (cf > 0) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = 8 * q + u;
            const bool need = k < cf;
            pcv[u] = need ? a.PTc[(int64_t)k * ld + rr] : 0.0;
            pcw[u] = need ? a.PTc[(int64_t)(SY_NB + k) * ld + rr] : 0.0;
        }
        if (q == 2 && do_finish) pjk = a.PT[(int64_t)j * SY_PW + lane];  // row j, lane = panel entry
    }
    SY_STAMP(1);
    // ---- consumption
    double vav = 0, psj = 0;
    if (do_finish) {
#pragma unroll
        for (int u = 0; u < 16; ++u) tvd[u] = u < cnt ? tvd[u] : 0.0;
#pragma unroll
        for (int u = 0; u < 4; ++u) tvt[u] = u < cnt2 ? tvt[u] : 0.0;
        double ps = ((((tvd[0] + tvd[1]) + (tvd[2] + tvd[3])) + ((tvd[4] + tvd[5]) + (tvd[6] + tvd[7]))) +
                     (((tvd[8] + tvd[9]) + (tvd[10] + tvd[11])) + ((tvd[12] + tvd[13]) + (tvd[14] + tvd[15])))) +
                    ((tvt[0] + tvt[1]) + (tvt[2] + tvt[3]));
        for (int b0 = 16; b0 < cnt; ++b0) ps += a.Pdir[(int64_t)(S0 + q + 4 * b0) * ld + rr];   // (orders beyond 4096)
        for (int b0 = 4; b0 < cnt2; ++b0) ps += a.Ptr[(int64_t)(q + 4 * b0) * ld + rr];
        s_psum[q][lane] = ps;
        double g = 0;
#pragma unroll
        for (int u = 0; u < 16; ++u) tvg[u] = gpart + 4 * u < G ? tvg[u] : 0.0;
        g = (((tvg[0] + tvg[1]) + (tvg[2] + tvg[3])) + ((tvg[4] + tvg[5]) + (tvg[6] + tvg[7]))) +
            (((tvg[8] + tvg[9]) + (tvg[10] + tvg[11])) + ((tvg[12] + tvg[13]) + (tvg[14] + tvg[15])));
        for (int b = gpart + 64; b < G; b += 4) g += a.Gpart[b * SY_PW + gk];
        s_graw[gpart][gk] = g;
#pragma unroll
        for (int u = 0; u < 8; ++u) tvv[u] = tid + SY_THREADS * u < n_vav ? tvv[u] : 0.0;
        vav = ((tvv[0] + tvv[1]) + (tvv[2] + tvv[3])) + ((tvv[4] + tvv[5]) + (tvv[6] + tvv[7]));
        for (int b = tid + SY_THREADS * 8; b < n_vav; b += SY_THREADS) vav += a.part_vav[b];
        if (q == 3) {
            const int Sj = j / SY_TC;
            const int nsegj = (nt128 - (Sj >> 1) + SEG - 1) / SEG;
            psj = (S0 + lane <= Sj ? x1 : 0.0) + (lane < nsegj ? x2 : 0.0);
        }
    }
    if (do_form && tid < SY_PW) s_Mf[tid] = ((tid & (SY_NB - 1)) < cf) ? mf : 0.0;
    if (!rok) {
        pre_raw = 0.0;
        pre_x = 0.0;
    }
    if (do_finish) {
        vav = sr_wave_sum(vav);
        if (lane == 0) s_vavw[q] = vav;
    }
    SY_STAMP(2);
    __syncthreads();
    SY_STAMP(3);
    if (do_finish && tid < SY_PW) s_raw[tid] = (s_graw[0][tid] + s_graw[1][tid]) + (s_graw[2][tid] + s_graw[3][tid]);
    SY_STAMP(4);
    __syncthreads();
    SY_STAMP(5);
    if (do_finish) {
        // multiplier of PT[r][k] in  V(r,:) (W'v) + W(r,:) (V'v):  the other half's dot
        if (tid < SY_PW) s_Gf[tid] = ((tid & (SY_NB - 1)) < cf) ? s_raw[tid ^ SY_NB] : 0.0;
        if (q == 1) {  // fixed-shape tree reductions (bitwise reproducible)
            double gg = (lane < cf) ? s_raw[lane] * s_raw[lane + SY_NB] : 0.0;
            gg = sr_wave_sum(gg);
            if (lane == 0) {
                const double vav = (s_vavw[0] + s_vavw[1]) + (s_vavw[2] + s_vavw[3]);
                const double tau = pre_tau;
                const double dot = tau * (vav - 2.0 * gg);  // p'v with p = tau (A v - V g1 - W g2)
                s_scal[0] = tau;
                s_scal[1] = -0.5 * tau * dot;
            }
        }
        if (q == 3) {  // row j: needed by every row of the form part
            psj = sr_wave_sum(psj);
            if (lane == 0) s_scal[2] = psj;
        }
    }
    SY_STAMP(6);
    __syncthreads();
    SY_STAMP(7);
    // ---- panel dots of the 64 rows
    {
        double sf = 0, sm = 0;
        if (cf > 0) {
            // the 32 multipliers of this wave in one batch of LDS reads, then two independent chains per sum (round 4: one LDS
            // round trip per fma pair, behind a branch on do_finish / do_form each, was 0.9 us of the 6.3 us of a launch)
            double gf[16], gm[16];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                gf[u] = s_Gf[8 * q + u];
                gf[8 + u] = s_Gf[SY_NB + 8 * q + u];
                gm[u] = s_Mf[8 * q + u];
                gm[8 + u] = s_Mf[SY_NB + 8 * q + u];
            }
            double sf2 = 0, sm2 = 0;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                sf = fma(pcv[u], gf[u], sf);
                sf2 = fma(pcw[u], gf[8 + u], sf2);
                sm = fma(pcv[u], gm[u], sm);
                sm2 = fma(pcw[u], gm[8 + u], sm2);
            }
            sf += sf2;
            sm += sm2;
            if (q == 2 && do_finish) {
                double sfj = pjk * s_Gf[lane];
                sfj = sr_wave_sum(sfj);
                if (lane == 0) s_scal[3] = sfj;
            }
        } else if (tid == 0) {
            s_scal[3] = 0.0;
        }
        s_sf[q][lane] = (r < n) ? sf : 0.0;
        s_sm[q][lane] = (r < n) ? sm : 0.0;
    }
    SY_STAMP(8);
    __syncthreads();
    SY_STAMP(9);
    if (q != 0) return;
    double sq = 0;
    if (r < n) {
        double wnew = 0, v = 0;
        if (do_finish) {
            const double tau = s_scal[0], alpha2 = s_scal[1], scale = pre_scale;
            v = (r == jf + 1) ? 1.0 : pre_raw * scale;
            const double p0 = s_psum[0][lane] + s_psum[1][lane] + s_psum[2][lane] + s_psum[3][lane];
            wnew = tau * (p0 - ((s_sf[0][lane] + s_sf[1][lane]) + (s_sf[2][lane] + s_sf[3][lane]))) + alpha2 * v;
            a.PT[(int64_t)r * SY_PW + cf] = v;
            a.PT[(int64_t)r * SY_PW + SY_NB + cf] = wnew;
            a.PTc[(int64_t)cf * ld + r] = v;
            a.PTc[(int64_t)(SY_NB + cf) * ld + r] = wnew;
            // LAPACK storage of the finished reflector: v below the subdiagonal of column jf
            if (r >= jf + 2) a.A[r + (int64_t)jf * ld] = v;
        }
        if (do_form) {
            double x = pre_x;
            double sm = (s_sm[0][lane] + s_sm[1][lane]) + (s_sm[2][lane] + s_sm[3][lane]);
            if (do_finish) {
                // row j of the column that this launch finishes: V(j, cf) = v_jf[jf+1] = 1
                const double wj = s_scal[0] * (s_scal[2] - s_scal[3]) + s_scal[1];
                sm += v * wj + wnew;
            }
            x -= sm;
            a.A[r + (int64_t)j * ld] = x;
            if (r == j) a.d[j] = x;
            if (r >= j + 2) sq = x * x;
        }
    }
    if (do_form) {
        sq = sr_wave_sum(sq);
        if (lane == 0) a.part_norm[blockIdx.x] = sq;
    }
    SY_STAMP(10);
}

// ---------------------------------------------------------------------------
// symv_kernel(j): reflector scalars of column j, then the symmetric product on the lower triangle.
//   1-D grid: workgroup b < nstr * maxseg is segment b % maxseg of strip S0 + b / maxseg (workgroups
//   past the strip's last segment idle); the last G workgroups are the panel dots.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(SY_THREADS, 2)
sytrd_symv_kernel(SytrdArgs a, int j, int cf, int n_norm, int S0, int SEG, int nstr, int maxseg, int G) {
    __shared__ double s_y[2][4][SY_TR];
    __shared__ double s_vr[2][SY_TR];
    __shared__ double s_red[4];
    const int tid = threadIdx.x;
    const int lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: scalar column bases
    const int64_t ld = a.ld;
    const int n = a.n;
    const double* __restrict__ colj = a.A + (int64_t)j * ld;
    const int nt128 = (n + SY_TR - 1) / SY_TR;
    const int ntile_wg = nstr * maxseg;
    const bool tile_wg = (int)blockIdx.x < ntile_wg;
    const int sy = tile_wg ? (int)blockIdx.x / maxseg : 0;
    const int sx = tile_wg ? (int)blockIdx.x - sy * maxseg : (int)blockIdx.x - ntile_wg;  // segment / dot index
    const int S = S0 + sy;
    const int Ib = (S >> 1) + sx * SEG;
    int Ie = Ib + SEG;
    if (Ie > nt128) Ie = nt128;
    const bool has_tiles = tile_wg && Ib < nt128;
    const bool dot_wg = !tile_wg && sx < G;
    const int col0 = SY_TC * S;

    // ---- every load that does not depend on the reflector scalars is issued first: the first
    // tile, the raw column entries of its rows and of the strip's columns, the norm partials
    double2 x[16];
    double2 rr = {0.0, 0.0};
    double raw_c = 0.0;
    if (has_tiles) {
        const unsigned voff = (unsigned)(SY_TR * Ib + 2 * lane) * 8u;  // per-lane byte offset; column bases stay scalar
        const char* __restrict__ cbase = reinterpret_cast<const char*>(a.A + (int64_t)(col0 + 16 * w) * ld);
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) x[cc] = *reinterpret_cast<const double2*>(cbase + (int64_t)cc * ld * 8 + voff);
        rr = *reinterpret_cast<const double2*>(reinterpret_cast<const char*>(colj) + voff);
        if (lane < 16) raw_c = colj[col0 + 16 * w + lane];
    } else if (dot_wg) {
        // panel dots [V'v | W'v]: 64 rows per workgroup, 16 per wave, lane = panel entry
        const int rb = j + 1 + sx * 64 + w * 16;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int r2 = rb + u;
            const int rs = r2 < n ? r2 : j;
            x[u].x = a.PT[(int64_t)rs * SY_PW + lane];  // (the tile registers double as panel-row / raw-entry pairs)
            x[u].y = colj[rs];
        }
    }
    // reflector scalars: every wave reduces the norm partials itself (same loads, same fixed-shape
    // tree: bitwise the same result in every wave and workgroup) -- no LDS broadcast, no barrier
    double scale;
    {
        double xn2 = a.part_norm[lane];  // (64 slots are there at every order: part_vav follows part_norm)
        const double alpha = colj[j + 1];
        xn2 = lane < n_norm ? xn2 : 0.0;
        for (int b = lane + 64; b < n_norm; b += 64) xn2 += a.part_norm[b];  // (orders beyond 4096)
        xn2 = sr_wave_sum(xn2);
        double beta, tau;
        if (xn2 == 0.0) {  // dlarfg: H = I
            tau = 0.0;
            beta = alpha;
            scale = 0.0;
        } else {
            beta = -copysign(sqrt(alpha * alpha + xn2), alpha);
            tau = (beta - alpha) / beta;
            scale = 1.0 / (alpha - beta);
        }
        if (blockIdx.x == 0 && tid == 0) {
            a.e[j] = beta;
            a.tau[j] = tau;
            a.scal[0] = tau;
            a.scal[1] = beta;
            a.scal[2] = scale;
        }
    }
    // v_r = 0 (r <= j), 1 (r = j+1), a_rj * scale (below): never materialised
    auto vval = [&](int r, double raw) -> double { return r <= j ? 0.0 : (r == j + 1 ? 1.0 : raw * scale); };
    double vav = 0;
    if (has_tiles) {
        // v of the wave's 16 columns: computed by lanes 0..15, read back lane by lane (scalar operands)
        double vc_l = 0.0;
        if (lane < 16) {
            const int cidx = col0 + 16 * w + lane;
            vc_l = (cidx < n) ? vval(cidx, raw_c) : 0.0;
        }
        double vcol[16];
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) {
            const int lo = __builtin_amdgcn_readlane(__double2loint(vc_l), cc);
            const int hi = __builtin_amdgcn_readlane(__double2hiint(vc_l), cc);
            vcol[cc] = __hiloint2double(hi, lo);
        }
        double z[16];
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) z[cc] = 0.0;
        // two register sets: the loads of tile I+1 are in flight while tile I is consumed
        double2 xn[16];
        double2 rrn = {0.0, 0.0};
        const char* __restrict__ cbase = reinterpret_cast<const char*>(a.A + (int64_t)(col0 + 16 * w) * ld);
        auto consume = [&](int I, int buf, const double2 (&xx)[16], const double2 rrr) {
            const int row0 = SY_TR * I + 2 * lane;
            const double vr0 = (row0 < n) ? vval(row0, rrr.x) : 0.0;
            const double vr1 = (row0 + 1 < n) ? vval(row0 + 1, rrr.y) : 0.0;
            double y0 = 0, y1 = 0;
            if (SY_TR * I >= col0 + SY_TC) {  // tile strictly below the diagonal
#pragma unroll
                for (int cc = 0; cc < 16; ++cc) {
                    const double vcc = vcol[cc];
                    y0 = fma(xx[cc].x, vcc, y0);
                    y1 = fma(xx[cc].y, vcc, y1);
                    z[cc] = fma(xx[cc].x, vr0, z[cc]);
                    z[cc] = fma(xx[cc].y, vr1, z[cc]);
                }
            } else {  // the tile meets the diagonal: rows >= columns for the direct part, > for the transposed
#pragma unroll
                for (int cc = 0; cc < 16; ++cc) {
                    const int cidx = col0 + 16 * w + cc;
                    const double vcc = vcol[cc];
                    const double x0 = (row0 >= cidx) ? xx[cc].x : 0.0;
                    const double x1 = (row0 + 1 >= cidx) ? xx[cc].y : 0.0;
                    y0 = fma(x0, vcc, y0);
                    y1 = fma(x1, vcc, y1);
                    z[cc] = fma((row0 > cidx) ? xx[cc].x : 0.0, vr0, z[cc]);
                    z[cc] = fma((row0 + 1 > cidx) ? xx[cc].y : 0.0, vr1, z[cc]);
                }
            }
            s_y[buf][w][2 * lane] = y0;
            s_y[buf][w][2 * lane + 1] = y1;
            if (w == 0) {
                s_vr[buf][2 * lane] = vr0;
                s_vr[buf][2 * lane + 1] = vr1;
            }
            __syncthreads();
            if (tid < SY_TR) {
                const double tot = (s_y[buf][0][tid] + s_y[buf][1][tid]) + (s_y[buf][2][tid] + s_y[buf][3][tid]);
                a.Pdir[(int64_t)S * ld + SY_TR * I + tid] = tot;
                vav = fma(tot, s_vr[buf][tid], vav);
            }
        };
        auto fetch = [&](int I, double2 (&xx)[16], double2& rrr) {
            const unsigned voff = (unsigned)(SY_TR * I + 2 * lane) * 8u;
#pragma unroll
            for (int cc = 0; cc < 16; ++cc) xx[cc] = *reinterpret_cast<const double2*>(cbase + (int64_t)cc * ld * 8 + voff);
            rrr = *reinterpret_cast<const double2*>(reinterpret_cast<const char*>(colj) + voff);
        };
        for (int I = Ib; I < Ie; I += 2) {
            if (I + 1 < Ie) fetch(I + 1, xn, rrn);
            consume(I, 0, x, rr);
            if (I + 1 < Ie) {
                if (I + 2 < Ie) fetch(I + 2, x, rr);
                consume(I + 1, 1, xn, rrn);
            }
        }
        // transposed sums: reduce-scatter over the 64 lanes (rows), 16 columns per lane
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const int half = 8 >> st, step = 32 >> st;
            const bool upper = (lane & step) != 0;
#pragma unroll
            for (int i = 0; i < half; ++i) {
                const double send = upper ? z[i] : z[i + half];
                const double keep = upper ? z[i + half] : z[i];
                z[i] = keep + __shfl_xor(send, step, 64);
            }
        }
        z[0] += __shfl_xor(z[0], 2, 64);
        z[0] += __shfl_xor(z[0], 1, 64);
        {
            const int cc = (((lane >> 5) & 1) << 3) | (((lane >> 4) & 1) << 2) | (((lane >> 3) & 1) << 1) | ((lane >> 2) & 1);
            const double vsel = __shfl(vc_l, cc, 64);  // all lanes active: the source lanes 0..15 must be
            if ((lane & 3) == 0) {
                a.Ptr[(int64_t)sx * ld + col0 + 16 * w + cc] = z[0];
                vav = fma(z[0], vsel, vav);
            }
        }
    } else if (dot_wg) {
        const int rb = j + 1 + sx * 64 + w * 16;
        double acc = 0;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int r2 = rb + u;
            acc = fma(x[u].x, (r2 < n) ? vval(r2, x[u].y) : 0.0, acc);
        }
        s_y[0][w][lane] = acc;
        __syncthreads();
        if (tid < SY_PW)
            a.Gpart[sx * SY_PW + tid] = (s_y[0][0][tid] + s_y[0][1][tid]) + (s_y[0][2][tid] + s_y[0][3][tid]);
    }
    vav = sr_wave_sum(vav);
    if (lane == 0) s_red[w] = vav;
    __syncthreads();
    if (tid == 0) a.part_vav[blockIdx.x] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

// ---------------------------------------------------------------------------
// rank-2NB update of the trailing matrix on the matrix cores:
//   A[r, s] -= sum_c V[r,c] W[s,c] + W[r,c] V[s,c],   r >= s >= j1   (lower triangle only)
// = PT[r, :] . rot(PT[s, :]) with the halves of the second row swapped.  One workgroup = one
// 128 x 128 tile (4 waves x 64 x 64, v_mfma_f64_16x16x4_f64), K = 64 in four 128-byte K-tiles,
// operands global -> LDS directly with the source-side XOR swizzle of kernels_gemm.hip.
// ---------------------------------------------------------------------------
typedef double sy_v2d __attribute__((ext_vector_type(2)));
typedef double sy_v4d __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void sy_lds_void_t;
typedef const __attribute__((address_space(1))) void sy_gbl_void_t;

// 16-byte-per-lane global -> LDS DMA through inline asm (see glds16 in kernels_gemm.hip: the
// builtin makes the compiler drain the VM counter in front of every LDS read, which would void the
// counted waits below)
__device__ __forceinline__ void sy_glds16(const void* gsrc, const void* lds_dst_generic) {
    const unsigned lds_dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(sy_lds_void_t*)lds_dst_generic);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

__global__ void __launch_bounds__(SY_THREADS)
sytrd_syr2k_mfma_kernel(SytrdArgs a, int j1, int T0) {
    constexpr int KB = 128;        // bytes of K per row per tile
    constexpr int OPB = 128 * KB;  // 16 KiB per operand tile
    constexpr int ROWB = SY_PW * 8;  // bytes per panel row (512)
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave & 1, wj = wave >> 1;
    int bi, bj;
    {
        const int t = blockIdx.x;
        int row = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
        while (row * (row + 1) / 2 > t) --row;
        while ((row + 1) * (row + 2) / 2 <= t) ++row;
        bi = row;
        bj = t - row * (row + 1) / 2;
    }
    const int i0 = 128 * (T0 + bi), j0 = 128 * (T0 + bj);
    const char* Ab = reinterpret_cast<const char*>(a.PT) + (int64_t)i0 * ROWB;
    const char* Bb = reinterpret_cast<const char*>(a.PT) + (int64_t)j0 * ROWB;
    int srcA[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int r = 8 * (wave * 4 + s) + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        srcA[s] = r * ROWB + c * 16;
    }
    auto issue = [&](int buf, int kt) {
        const int kbA = kt * KB;
        const int kbB = (kt * KB + SY_NB * 8) & (ROWB - 1);  // second operand: halves swapped
        char* base = smem + buf * 2 * OPB + (wave * 4) * 1024;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            sy_glds16(Ab + srcA[s] + kbA, base + s * 1024);
            sy_glds16(Bb + srcA[s] + kbB, base + OPB + s * 1024);
        }
    };
    constexpr int nk = ROWB / KB;  // 4
    sy_v4d acc[4][4];
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[x][y][r] = 0.0;
    const int r16 = lane & 15, g = lane >> 4;
    int rowA[4], rowB[4], swA[4], swB[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        rowA[t] = wi * 64 + t * 16 + r16;
        rowB[t] = wj * 64 + t * 16 + r16;
        swA[t] = (rowA[t] >> 1) & 7;
        swB[t] = (rowB[t] >> 1) & 7;
    }
    // the tile of A that will be updated is requested first of all (64 entries per lane; one wave
    // per SIMD, so the registers are there): its latency hides behind the whole product
    const int64_t ld = a.ld;
    const int n = a.n;
    double cv[4][4][4];
#pragma unroll
    for (int tj = 0; tj < 4; ++tj)
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) {
            const int ii = i0 + wi * 64 + ti * 16 + r16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int jj = j0 + wj * 64 + tj * 16 + g + 4 * r;
                const bool ok = ii >= jj && jj >= j1 && ii < n;
                cv[tj][ti][r] = ok ? a.A[ii + (int64_t)jj * ld] : 0.0;
            }
        }
    // K = 64 is only four K-tiles: all of them are requested up front (4 x 32 KiB of LDS) and
    // consumed behind counted waits, so the workgroup pays the global -> LDS latency once
#pragma unroll
    for (int kt = 0; kt < nk; ++kt) issue(kt, kt);
#pragma unroll
    for (int kt = 0; kt < nk; ++kt) {
        if (kt == 0) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        else if (kt == 1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (kt == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const char* tA = smem + kt * 2 * OPB;
        const char* tB = tA + OPB;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            sy_v2d fi[4], fj[4];
            const int ch = 4 * q + g;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                fi[t] = *reinterpret_cast<const sy_v2d*>(tA + rowA[t] * KB + ((ch ^ swA[t]) << 4));
                fj[t] = *reinterpret_cast<const sy_v2d*>(tB + rowB[t] * KB + ((ch ^ swB[t]) << 4));
            }
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int tj = 0; tj < 4; ++tj)
#pragma unroll
                    for (int ti = 0; ti < 4; ++ti)
                        acc[tj][ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(fj[tj][e], fi[ti][e], acc[tj][ti], 0, 0, 0);
        }
    }
    // D[jj][ii]: ii = lane & 15 (row of A, contiguous), jj = (lane >> 4) + 4 * reg
#pragma unroll
    for (int tj = 0; tj < 4; ++tj)
#pragma unroll
        for (int ti = 0; ti < 4; ++ti) {
            const int ii = i0 + wi * 64 + ti * 16 + r16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int jj = j0 + wj * 64 + tj * 16 + g + 4 * r;
                if (ii >= jj && jj >= j1 && ii < n) a.A[ii + (int64_t)jj * ld] = cv[tj][ti][r] - acc[tj][ti][r];
            }
        }
}

// ---------------------------------------------------------------------------
// Orders n <= SR_NMAX: ONE launch per column (round 3).  At these orders the panel form above is pure latency -- two
// dependent launches per column, 13.4 us per column at n = 1024 -- and its two chip-wide dependencies per column (the
// norm of the updated column, then p = A v) can be folded into one:
//   * the whole symmetric trailing matrix is kept (both triangles), a workgroup owns ROWS of it (row i -> workgroup
//     i mod G, two rows per wave): the product y_i = A(i, :) v of an owned row is complete inside one wave, there are no
//     partial products to add up across workgroups;
//   * the update is the unblocked one, A <- A - v w' - w v' (dsytd2), applied LAZILY: launch j applies the update of
//     column j-1 to its rows in the same pass that multiplies them by v_j (one read + one write of the trailing
//     matrix per column: n^3/3 x 16 bytes in all, 5.7 GB at n = 1024, from the L2 of the XCD that owns the row --
//     that is why this form stops at SR_NMAX);
//   * every workgroup forms the reflector of column j redundantly from row j (= column j) of the previous version
//     and the full vectors v_{j-1}, y_{j-1} = A v_{j-1} and the per-workgroup partial dots y'v left by launch j-1
//     (w_{j-1} = tau y - tau^2/2 (y'v) v needs the chip-wide scalar y'v: the sum of G partials, every workgroup adds
//     them in the same order).  Same loads, same instruction sequence: bitwise the same v_j in every workgroup.
// So the chain per column is one kernel boundary + (vector loads -> wave reduction, block reduction -> row pass from registers,
// whose loads were issued at the top of the kernel).  The two products of the update are rounded separately and
// added (a - (v_i w_c + w_i v_c)): the stored matrix stays bitwise symmetric.
// (Measured and not kept, round 3: the last 128 columns inside ONE workgroup with the 128 x 128 block in LDS -- dsytd2 with
// four barriers per column -- takes as long as the 128 launches it replaces, 4 us per column: one CU's LDS moves the
// block three times per column.)
// (Also measured and not kept: writing the trailing matrix back every SECOND column only -- a launch that finds two pending
// rank-2 updates applies both in registers, the older pair (v, w) read ready-made -- i.e. a quarter less trailing-matrix
// traffic: 17.3 ms at n = 2048 with or without the skipped stores, against 15.8 for this kernel (two more vectors per
// launch in LDS).  The writes ride along with the reads; they are not what a column waits for.)
// Memory: line a of the allocation (A + a ld) is row a = column a.  Reflector j is stored LAPACK-style on line j,
// positions >= j+2, one launch late (launch j still reads line j as the matrix row).
// ---------------------------------------------------------------------------
constexpr int SR_NMAX = 2048;
constexpr int SR_THREADS = 256;
constexpr int SR_RPW = 2;  // rows per wave

struct SytrdRowArgs {
    double* A;
    int64_t ld;
    int n;
    int G;         // workgroups = ceil(n / 8)
    double* vbuf;  // 2 x ld: v_j (absolute row index), by parity of j
    double* ybuf;  // 2 x ld: A_j v_j
    double* pdot;  // 2 x 256: per-workgroup partial of (A_j v_j)'v_j
    double* d;
    double* e;
    double* tau;
};

// Wave-wide sum on the DPP path (quad permutes, row rotations, four read-lanes) instead of six ds_bpermute round
// trips of a 64-bit value: the kernel below is one dependent chain with three of these in it.  All 64 lanes must be
// active.  The result is uniform.

// NCH: 128-column chunks of a row that can be active (ceil(ld / 128) at most)
template <int NCH>
__global__ void __launch_bounds__(SR_THREADS)
sytrd_row_kernel(SytrdRowArgs a, int j) {
    constexpr int NT = NCH / 2 > 0 ? NCH / 2 : 1;  // 256-entry slices of the vectors
    __shared__ double s_vp[NCH * 128], s_wp[NCH * 128], s_vj[NCH * 128];
    __shared__ double s_red[3][4];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = a.n, G = a.G, b = blockIdx.x;
    const int64_t ld = a.ld;
    const int k0 = (j + 1) >> 7, c0 = k0 << 7;  // first active chunk / its first column
    const int par = j & 1, pp = par ^ 1;
    double* __restrict__ A = a.A;

    // ---- loads that depend on nothing: the owned rows, the vectors of the previous column, row j
    int rowi[SR_RPW];
    double2 xr[SR_RPW][NCH];
    {
        int first = j + 1 + ((b - (j + 1)) % G + G) % G;
#pragma unroll
        for (int r = 0; r < SR_RPW; ++r) {
            const int i = first + G * (wv + 4 * r);
            rowi[r] = i < n ? i : -1;
            const double* __restrict__ line = A + (int64_t)(i < n ? i : 0) * ld + c0 + 2 * lane;
#pragma unroll
            for (int k = 0; k < NCH; ++k) {
                xr[r][k] = double2{0.0, 0.0};
                if (i < n && c0 + 128 * k < n) xr[r][k] = *reinterpret_cast<const double2*>(line + 128 * k);
                if (c0 + 128 * k + 2 * lane >= n) xr[r][k].x = 0.0;  // padding columns may hold anything
                if (c0 + 128 * k + 2 * lane + 1 >= n) xr[r][k].y = 0.0;
            }
        }
    }
    double vpv[NT], ypv[NT], rjv[NT];
    const double* __restrict__ vprev = a.vbuf + (int64_t)pp * ld;
    const double* __restrict__ yprev = a.ybuf + (int64_t)pp * ld;
    const double* __restrict__ linej = A + (int64_t)j * ld;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int i = c0 + tid + 256 * t;
        const bool in = i > j && i < n;
        vpv[t] = (in && j > 0) ? vprev[i] : 0.0;
        ypv[t] = (in && j > 0) ? yprev[i] : 0.0;
        rjv[t] = in ? linej[i] : 0.0;
    }
    const double tau_p = j > 0 ? a.tau[j - 1] : 0.0;
    const double yp_j = j > 0 ? yprev[j] : 0.0;  // entry j of the previous product (v_{j-1}(j) = 1)
    const double ajj = linej[j];
    double pd = 0.0;  // every wave adds up all G partials itself: no barrier in front of w_{j-1}
    if (j > 0) {
        // four loads, then four selects, then the sum: `if (..) pd += load` compiles to a load, a wait and an add per slice,
        // i.e. up to three more memory round trips behind the one of the matrix rows (round 4; all 256 slots exist)
        double t4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) t4[q] = a.pdot[pp * 256 + 64 * q + lane];
#pragma unroll
        for (int q = 0; q < 4; ++q) t4[q] = 64 * q + lane < G ? t4[q] : 0.0;
        pd = (t4[0] + t4[1]) + (t4[2] + t4[3]);
    }

    // ---- s = y'v of the previous column, then w_{j-1} (LDS) and the updated column x
    const double sdot = sr_wave_sum(pd);
    const double hts = 0.5 * tau_p * tau_p * sdot;  // w = tau y - (tau^2 s / 2) v
    const double wp_j = tau_p * yp_j - hts;
    double xv[NT];
    double xn2 = 0.0, alpha = 0.0;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int i = c0 + tid + 256 * t;
        const double wpi = tau_p * ypv[t] - hts * vpv[t];
        s_vp[tid + 256 * t] = vpv[t];
        s_wp[tid + 256 * t] = wpi;
        xv[t] = rjv[t] - (__dmul_rn(vpv[t], wp_j) + wpi);  // - (v_i w_j + w_i v_j), v_j = 1
        if (i > j + 1 && i < n) xn2 = fma(xv[t], xv[t], xn2);
        if (i == j + 1) alpha = xv[t];
    }
    // norm of x below its leading entry and the leading entry itself, to every thread
    xn2 = sr_wave_sum(xn2);
    alpha = sr_wave_sum(alpha);  // one lane holds it, the others 0
    if (lane == 0) {
        s_red[1][wv] = xn2;
        s_red[2][wv] = alpha;
    }
    __syncthreads();
    xn2 = (s_red[1][0] + s_red[1][1]) + (s_red[1][2] + s_red[1][3]);
    alpha = (s_red[2][0] + s_red[2][1]) + (s_red[2][2] + s_red[2][3]);
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
    double* __restrict__ vout = a.vbuf + (int64_t)par * ld;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int i = c0 + tid + 256 * t;
        double v = 0.0;
        if (i == j + 1) v = 1.0;
        else if (i > j + 1 && i < n) v = xv[t] * scale;
        s_vj[tid + 256 * t] = v;
        if (b == 0 && i > j && i < n) vout[i] = v;
    }
    if (b == 0 && tid == 0) {
        a.d[j] = ajj - (wp_j + wp_j);
        a.e[j] = beta;
        a.tau[j] = tau;
    }
    // reflector j-1 to its LAPACK place (line j-1, positions >= j+1): nobody reads that line any more
    if (j > 0 && b == (G > 1 ? 1 : 0)) {
        double* __restrict__ ref = A + (int64_t)(j - 1) * ld;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int i = c0 + tid + 256 * t;
            if (i > j && i < n) ref[i] = vpv[t];
        }
    }
    __syncthreads();

    // ---- owned rows: update of column j-1, product with v_j
    double pdw = 0.0;
#pragma unroll
    for (int r = 0; r < SR_RPW; ++r) {
        const int i = rowi[r];
        if (i < 0) continue;  // wave-uniform
        const double vi = s_vp[i - c0], wi = s_wp[i - c0], vji = s_vj[i - c0];
        double acc0 = 0.0, acc1 = 0.0;
        double* __restrict__ line = A + (int64_t)i * ld + c0 + 2 * lane;
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
            if (c0 + 128 * k < n) {
                const double2 vp2 = *reinterpret_cast<const double2*>(&s_vp[128 * k + 2 * lane]);
                const double2 wp2 = *reinterpret_cast<const double2*>(&s_wp[128 * k + 2 * lane]);
                const double2 vj2 = *reinterpret_cast<const double2*>(&s_vj[128 * k + 2 * lane]);
                double2 x = xr[r][k];
                x.x = x.x - (__dmul_rn(vi, wp2.x) + __dmul_rn(wi, vp2.x));
                x.y = x.y - (__dmul_rn(vi, wp2.y) + __dmul_rn(wi, vp2.y));
                acc0 = fma(x.x, vj2.x, acc0);
                acc1 = fma(x.y, vj2.y, acc1);
                if (j > 0) *reinterpret_cast<double2*>(line + 128 * k) = x;
            }
        }
        const double y = sr_wave_sum(acc0 + acc1);
        if (lane == 0) a.ybuf[(int64_t)par * ld + i] = y;
        pdw = fma(y, vji, pdw);
    }
    if (lane == 0) s_red[0][wv] = pdw;
    __syncthreads();
    if (tid == 0) a.pdot[par * 256 + b] = (s_red[0][0] + s_red[0][1]) + (s_red[0][2] + s_red[0][3]);
}

// upper triangle <- lower triangle (the callers promise the lower one only): line a, positions < a
__global__ void __launch_bounds__(256)
sytrd_row_mirror_kernel(int n, int64_t ld, double* __restrict__ A) {
    __shared__ double tile[64][65];
    // tile pair (I, J), I > J: block (rows 64 I.., line 64 J..) is valid, its mirror is written; I == J in place
    int t = blockIdx.x, I = 0;
    while (t > I) {
        t -= I + 1;
        ++I;
    }
    const int J = t;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int c = ty; c < 64; c += 4) {
        const int line = 64 * J + c, pos = 64 * I + tx;
        tile[c][tx] = (line < n && pos < n) ? A[(int64_t)line * ld + pos] : 0.0;
    }
    __syncthreads();
    for (int c = ty; c < 64; c += 4) {
        const int line = 64 * I + c, pos = 64 * J + tx;
        if (line < n && pos < n && pos < line) A[(int64_t)line * ld + pos] = tile[tx][c];
    }
}

// ---------------------------------------------------------------------------
// host driver.  A: n x n inside an ld x ld allocation, ld a multiple of 128 (tiles read whole
// 128-row / 64-column blocks; entries outside the lower triangle of the leading n x n part are
// read but never used).
// ---------------------------------------------------------------------------
size_t sytrd_workspace_doubles(int64_t n, int64_t ld) {
    (void)n;
    return (size_t)ld * SY_PW + (size_t)(ld / SY_TC + 1) * ld + (size_t)SY_MAXSEG * ld + (size_t)(ld / 64 + 1) * SY_PW +
           (size_t)(ld / SY_FROWS + 8) + SY_MAXVAV + 64 + (size_t)ld * SY_PW;  // (+ the one-launch form's column-major panel)
}

static SytrdArgs sytrd_args(int64_t n, double* A, int64_t ld, double* d, double* e, double* tau, double* ws) {
    SytrdArgs a;
    a.A = A;
    a.ld = ld;
    a.n = (int)n;
    a.PT = ws;
    a.Pdir = a.PT + ld * SY_PW;
    a.Ptr = a.Pdir + (ld / SY_TC + 1) * ld;
    a.Gpart = a.Ptr + (size_t)SY_MAXSEG * ld;
    a.part_norm = a.Gpart + (ld / 64 + 1) * SY_PW;
    a.part_vav = a.part_norm + (ld / SY_FROWS + 8);
    a.scal = a.part_vav + SY_MAXVAV;
    a.PTc = a.scal + 64;
    a.d = d;
    a.e = e;
    a.tau = tau;
    return a;
}

// per-device kernel attributes, set by sdpsr_create() with the ctx's device current
bool sytrd_set_device_attributes() {
    bool ok = true;
    ok &= hipSuccess == hipFuncSetAttribute(reinterpret_cast<const void*>(&sytrd_syr2k_mfma_kernel),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    return ok;
}

// The launch sequence of a tridiagonalisation goes either to a stream or, node by node, into a
// hipGraph built with the explicit node API.  No stream capture: while ANY stream of the process
// is capturing, legacy-stream operations of OTHER threads fail with hipErrorStreamCaptureImplicit
// on this runtime (seen: hipBLASLt's initialisation and rocSOLVER's stedc in a second ctx's thread
// failing or silently returning wrong eigenvalues while this thread captured).
struct SytrdEmitter {
    hipStream_t s = nullptr;
    hipGraph_t graph = nullptr;  // non-null: add nodes (a linear chain) instead of launching
    hipGraphNode_t last = nullptr;
    bool ok = true;

    template <typename... Args>
    void kernel(const void* fn, unsigned grid, unsigned block, size_t shmem, Args... args) {
        void* argv[] = {(void*)&args...};
        if (!graph) {
            if (hipLaunchKernel(fn, dim3(grid), dim3(block), argv, shmem, s) != hipSuccess) ok = false;
            return;
        }
        hipKernelNodeParams p{};
        p.func = const_cast<void*>(fn);
        p.gridDim = dim3(grid);
        p.blockDim = dim3(block);
        p.sharedMemBytes = (unsigned)shmem;
        p.kernelParams = argv;  // copied by the call
        p.extra = nullptr;
        hipGraphNode_t node = nullptr;
        if (hipGraphAddKernelNode(&node, graph, last ? &last : nullptr, last ? 1 : 0, &p) != hipSuccess) ok = false;
        else last = node;
    }
    void zero(void* dst, size_t bytes) {  // bytes % 4 == 0
        if (!graph) {
            if (hipMemsetAsync(dst, 0, bytes, s) != hipSuccess) ok = false;
            return;
        }
        hipMemsetParams m{};
        m.dst = dst;
        m.elementSize = 4;
        m.width = bytes / 4;
        m.height = 1;
        m.pitch = bytes;
        m.value = 0;
        hipGraphNode_t node = nullptr;
        if (hipGraphAddMemsetNode(&node, graph, last ? &last : nullptr, last ? 1 : 0, &m) != hipSuccess) ok = false;
        else last = node;
    }
    void copy8(void* dst, const void* src) {
        if (!graph) {
            if (hipMemcpyAsync(dst, src, 8, hipMemcpyDeviceToDevice, s) != hipSuccess) ok = false;
            return;
        }
        hipGraphNode_t node = nullptr;
        if (hipGraphAddMemcpyNode1D(&node, graph, last ? &last : nullptr, last ? 1 : 0, dst, src, 8, hipMemcpyDeviceToDevice) != hipSuccess)
            ok = false;
        else last = node;
    }
};

static void emit_symv(SytrdEmitter& em, const SytrdArgs& a, int j, int cf, int n_norm, const SymvGeom& g) {
    em.kernel(reinterpret_cast<const void*>(&sytrd_symv_kernel), (unsigned)g.grid, SY_THREADS, 0, a, j, cf, n_norm, g.S0, g.SEG,
              g.nstr, g.maxseg, g.G);
}
static void emit_form(SytrdEmitter& em, unsigned grid, const SytrdArgs& a, int j, int cf, int do_finish, int do_form, int n_vav,
                      int S0, int SEG, int G) {
    em.kernel(reinterpret_cast<const void*>(&sytrd_form_kernel), grid, SY_THREADS, 0, a, j, cf, do_finish, do_form, n_vav, S0, SEG, G);
}

// stop > 0 (a multiple of 128): only columns 0 .. stop-1; the trailing matrix A(stop:, stop:) is left updated (lower
// triangle), column `stop` not formed -- the row form takes over from there
static bool emit_sytrd(SytrdEmitter& em, int64_t n64, double* A, int64_t ld, double* d, double* e, double* tau, double* ws,
                       int stop = -1) {
    const int n = (int)n64;
#ifdef LK_TIMING
    long long* dbg = nullptr;
    struct DbgDump {
        long long*& dbg; hipStream_t s; int n;
        ~DbgDump() {
            if (!dbg) return;
            hipStreamSynchronize(s);
            std::vector<long long> h((size_t)(n + 64) * 3 * 16);
            hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost);
            for (int j : {5, 20, 31, 33, 1029, 1950}) {
                if (j >= n) continue;
                for (int wch = 0; wch < 3; ++wch) {
                    const long long* t = &h[((size_t)j * 3 + wch) * 16];
                    fprintf(stderr, "[form timing] j=%d wg=%d:", j, wch);
                    for (int i = 1; i < 11; ++i) fprintf(stderr, " %6.2f", t[i] ? (t[i] - t[0]) * 0.01 : -1.0);
                    fprintf(stderr, "  (t0 rel. to wg0 %.2f)\n", (t[0] - h[(size_t)j * 3 * 16]) * 0.01);
                }
            }
            long long* z = nullptr;
            hipMemcpyToSymbol(HIP_SYMBOL(sy_dbg), &z, sizeof(z));
            hipFree(dbg);
        }
    } dump{dbg, em.s, n};
    if (!em.graph && dbg_on()) {
        hipMalloc(&dbg, (size_t)(n + 64) * 3 * 16 * 8);
        hipMemset(dbg, 0, (size_t)(n + 64) * 3 * 16 * 8);
        hipMemcpyToSymbol(HIP_SYMBOL(sy_dbg), &dbg, sizeof(dbg));
    }
#endif
    SytrdArgs a = sytrd_args(n64, A, ld, d, e, tau, ws);
    em.zero(ws, sytrd_workspace_doubles(n, ld) * sizeof(double));
    if (n == 1) {
        em.copy8(d, A);
        return em.ok;
    }
    int cf = 0;  // panel column of the column currently being formed
    // column 0: plain form (no panel yet)
    emit_form(em, (unsigned)((n + SY_FROWS - 1) / SY_FROWS), a, 0, 0, 0, 1, 0, 0, 1, 0);
    for (int j = 0; j <= n - 2; ++j) {
        // reflector of column j + symmetric product
        const int n_norm = (n - j + SY_FROWS - 1) / SY_FROWS;
        const SymvGeom g = symv_geometry(n, j, cf);
        emit_symv(em, a, j, cf, n_norm, g);
        const int n_vav = g.grid;
        const int jn = j + 1;
        const unsigned nb_form = (unsigned)((n - jn + SY_FROWS - 1) / SY_FROWS);
        if (cf + 1 < SY_NB && jn <= n - 1) {
            // finish W(:, cf) and form column j+1 inside the same panel
            emit_form(em, nb_form, a, jn, cf, 1, 1, n_vav, g.S0, g.SEG, g.G);
            ++cf;
        } else {
            // panel complete: finish W, update the trailing matrix, start a new panel
            emit_form(em, nb_form, a, jn, cf, 1, 0, n_vav, g.S0, g.SEG, g.G);
            if (n - jn > 0 && cf + 1 == SY_NB) {
                const int T0 = jn / 128;
                const int nt = (n + 127) / 128 - T0;
                em.kernel(reinterpret_cast<const void*>(&sytrd_syr2k_mfma_kernel), (unsigned)(nt * (nt + 1) / 2), SY_THREADS,
                          128 * 1024, a, jn, T0);
            }
            if (jn == stop) return em.ok;
            if (n - jn > 0) emit_form(em, nb_form, a, jn, 0, 0, 1, 0, 0, 1, 0);
            cf = 0;
        }
    }
    return em.ok;
}

// one launch per column (sytrd_row_kernel): n <= SR_NMAX
static bool emit_sytrd_rows(SytrdEmitter& em, int64_t n64, double* A, int64_t ld, double* d, double* e, double* tau, double* ws) {
    const int n = (int)n64;
    if (n == 1) {
        em.copy8(d, A);
        return em.ok;
    }
    SytrdRowArgs a;
    a.A = A;
    a.ld = ld;
    a.n = n;
    a.G = (n + 7) / 8;
    a.vbuf = ws;
    a.ybuf = ws + 2 * ld;
    a.pdot = ws + 4 * ld;
    a.d = d;
    a.e = e;
    a.tau = tau;
    const int T = (n + 63) / 64;
    em.kernel(reinterpret_cast<const void*>(&sytrd_row_mirror_kernel), (unsigned)(T * (T + 1) / 2), 256, 0, n, ld, A);
    const int nch_all = (n + 127) / 128;
    for (int j = 0; j <= n - 2; ++j) {
        const int nch = nch_all - ((j + 1) >> 7);  // chunks still active
        const void* fn = nch <= 2   ? reinterpret_cast<const void*>(&sytrd_row_kernel<2>)
                         : nch <= 4 ? reinterpret_cast<const void*>(&sytrd_row_kernel<4>)
                         : nch <= 8 ? reinterpret_cast<const void*>(&sytrd_row_kernel<8>)
                                    : reinterpret_cast<const void*>(&sytrd_row_kernel<16>);
        em.kernel(fn, (unsigned)a.G, SR_THREADS, 0, a, j);
    }
    em.copy8(d + (n - 1), A + (int64_t)(n - 1) * ld + (n - 1));  // tau(n-2) = 0: the last diagonal entry is final
    return em.ok;
}
// The panel columns with ONE launch per column (kernels_sytrd_look.hip): columns 0 .. stop-1, stop a multiple of 128 below n;
// the trailing matrix A(stop:, stop:) is left updated (lower triangle) for the row form.  The buffers of the two-launch form
// are reused: the partial products take Pdir, the column Ptr, the records Gpart / part_norm / part_vav.
#ifdef LK_TIMING
void sytrd_look_debug_buffer(long long* p);
#endif
static bool emit_sytrd_look(SytrdEmitter& em, int64_t n64, double* A, int64_t ld, double* d, double* e, double* tau, double* ws, int stop) {
    const int n = (int)n64;
    SytrdArgs a = sytrd_args(n64, A, ld, d, e, tau, ws);
    em.zero(ws, sytrd_workspace_doubles(n, ld) * sizeof(double));
    SytrdLookArgs la;
    la.A = A;
    la.ld = ld;
    la.n = n;
    la.PT = a.PT;
    la.PTc = a.PTc;
    la.P = a.Pdir;
    la.acol = a.Ptr;
    la.rec_norm = a.part_norm;
    la.rec_dots = a.Gpart;
    la.rec_vaz = a.part_vav;
    la.vaz_cap = SY_MAXVAV / 2;
    la.d = d;
    la.e = e;
    la.tau = tau;
    const int nb = (n + 127) / 128;
    auto tiles = [&](int j) {
        const int t = nb - j / 128;
        return t * (t + 1) / 2;
    };
#ifdef LK_TIMING
    long long* dbg = nullptr;
    if (!em.graph && dbg_on()) {
        hipMalloc(&dbg, (size_t)(stop + 64) * 3 * 16 * 8);
        hipMemset(dbg, 0, (size_t)(stop + 64) * 3 * 16 * 8);
        sytrd_look_debug_buffer(dbg);
    }
#endif
    for (int j0 = 0; j0 < stop; j0 += SY_NB) {
        for (int cf = 0; cf < SY_NB; ++cf) {
            const int j = j0 + cf;
            em.kernel(sytrd_look_kernel_fn(cf == 0 ? 0 : 1, ld), (unsigned)tiles(j), SY_THREADS, 0, la, j, cf, cf == 0 ? 0 : 2 * tiles(j - 1));
        }
        const int j1 = j0 + SY_NB;
        em.kernel(sytrd_look_kernel_fn(2, ld), (unsigned)(nb - j1 / 128), SY_THREADS, 0, la, j1, SY_NB, 2 * tiles(j1 - 1));
        const int T0 = j1 / 128;
        const int nt = nb - T0;
        em.kernel(reinterpret_cast<const void*>(&sytrd_syr2k_mfma_kernel), (unsigned)(nt * (nt + 1) / 2), SY_THREADS, 128 * 1024, a, j1, T0);
    }
#ifdef LK_TIMING
    if (dbg) {
        hipStreamSynchronize(em.s);
        std::vector<long long> h((size_t)(stop + 64) * 3 * 16);
        hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost);
        for (int j : {5, 20, 31, 32, 33, 1029, 1055, 1056, 1950, 1983, 1984}) {
            if (j >= stop + 1) continue;
            for (int wch = 0; wch < 3; ++wch) {
                const long long* t = &h[((size_t)j * 3 + wch) * 16];
                fprintf(stderr, "[look timing] j=%d wg=%d:", j, wch);
                for (int i = 1; i < 10; ++i) fprintf(stderr, " %6.2f", t[i] ? (t[i] - t[0]) * 0.01 : -1.0);
                fprintf(stderr, "  (us since the first stamp; t0 rel. to wg0 %.2f)\n", (t[0] - h[(size_t)j * 3 * 16]) * 0.01);
            }
        }
        sytrd_look_debug_buffer(nullptr);
        hipFree(dbg);
    }
#endif
    return em.ok;
}

// the default: row form up to SR_NMAX; beyond, the panel form until the trailing matrix has come down to SR_NMAX
// (there a column of the two-launch panel form is two launch latencies, 13-14 us, against one latency + 16 m^2 bytes of the row form)
// form: 0 = panels (two launches per column) at every order, 1 = row form + two-launch panels above SR_NMAX (default),
//       2 = row form + one-launch panels above SR_NMAX (SDPSR_FLAG_SYTRD_ONE_LAUNCH; measured slower)
static int sytrd_form(const sdpsr_ctx* c, int64_t n, int64_t ld) {
    if (ld < n || (c && (c->opts.flags & SDPSR_FLAG_SYTRD_PANELS))) return 0;
    if (ld <= 8192 && c && (c->opts.flags & SDPSR_FLAG_SYTRD_ONE_LAUNCH)) return 2;
    return 1;
}
static bool emit_sytrd_default(SytrdEmitter& em, int64_t n, double* A, int64_t ld, double* d, double* e, double* tau, double* ws, int form) {
    if (form == 0) return emit_sytrd(em, n, A, ld, d, e, tau, ws);
    if (n <= SR_NMAX) return emit_sytrd_rows(em, n, A, ld, d, e, tau, ws);
    const int64_t j0 = (n - SR_NMAX + 127) / 128 * 128;
    if (form == 2 ? !emit_sytrd_look(em, n, A, ld, d, e, tau, ws, (int)j0) : !emit_sytrd(em, n, A, ld, d, e, tau, ws, (int)j0)) return false;
    return emit_sytrd_rows(em, n - j0, A + j0 * ld + j0, ld, d + j0, e + j0, tau + j0, ws);
}

static void launch_sytrd_direct(hipStream_t s, int64_t n64, double* A, int64_t ld, double* d, double* e, double* tau,
                                double* ws, int form) {
    SytrdEmitter em;
    em.s = s;
    emit_sytrd_default(em, n64, A, ld, d, e, tau, ws, form);
}

// measurement hook: only the symv launches of a full tridiagonalisation (same grid shapes and
// memory traffic, numerically meaningless because form/syr2k are skipped)
void launch_sytrd_symv_sweep(hipStream_t s, int64_t n64, double* A, int64_t ld, double* d, double* e,
                             double* tau, double* ws) {
    const int n = (int)n64;
    SytrdArgs a = sytrd_args(n64, A, ld, d, e, tau, ws);
    hipMemsetAsync(ws, 0, sytrd_workspace_doubles(n, ld) * sizeof(double), s);
    for (int j = 0; j <= n - 2; ++j) {
        const int cf = j % SY_NB;
        const int n_norm = (n - j + SY_FROWS - 1) / SY_FROWS;
        SytrdEmitter em;
        em.s = s;
        emit_symv(em, a, j, cf, n_norm, symv_geometry(n, j, cf));
    }
}

// Most of the ~2n + n/32 launches of the tridiagonalisation are shorter than the cost of
// launching them from the host (2-4 us of kernel against 3-5 us of launch): the launch sequence of
// one problem shape is built once as a hipGraph (explicit nodes, see SytrdEmitter) and replayed.  Key = every value baked into
// the nodes (order, leading dimension, all pointers); the ctx's buffers are grow-only, so the key
// is stable across calls.  A few graphs are kept (generic elements alternate between two or
// three buffers).
// The cache belongs to the ctx (sdpsr_ctx::sytrd_graphs): no process-global state, and calls on
// one ctx are serialised by contract, so no lock is needed.
struct SytrdGraph {
    int64_t n = 0, ld = 0;
    int form = 0;
    const void *A = nullptr, *d = nullptr, *e = nullptr, *tau = nullptr, *ws = nullptr;
    hipGraphExec_t exec = nullptr;
    uint64_t last_use = 0;
};
struct SytrdGraphCache {
    SytrdGraph slots[6];
    uint64_t clock = 0;
    // what the cache did (sytrd_graph_cache_stats; traced under SDPSR_DEBUG): a miss builds and instantiates a graph of
    // ~2 n nodes on the host, several times the cost of the solve it then replays
    uint64_t hits = 0, misses = 0;
    double instantiate_ms = 0;
};
bool dbg_on();  // ctx.cpp (SDPSR_DEBUG)
void sytrd_graph_cache_stats(const SytrdGraphCache* g, uint64_t* hits, uint64_t* misses, double* instantiate_ms) {
    if (hits) *hits = g ? g->hits : 0;
    if (misses) *misses = g ? g->misses : 0;
    if (instantiate_ms) *instantiate_ms = g ? g->instantiate_ms : 0.0;
}
void sytrd_graph_cache_destroy(SytrdGraphCache* g) {
    if (!g) return;
    for (auto& sl : g->slots)
        if (sl.exec) hipGraphExecDestroy(sl.exec);
    delete g;
}

void launch_sytrd(sdpsr_ctx* c, int64_t n64, double* A, int64_t ld, double* d, double* e, double* tau, double* ws) {
    hipStream_t s = c->stream;
    const bool no_graph = (c->opts.flags & SDPSR_FLAG_NO_GRAPH) != 0;  // profiling: per-kernel statistics of the launches
    const int form = sytrd_form(c, n64, ld);
    if (no_graph || n64 < 64) {
        launch_sytrd_direct(s, n64, A, ld, d, e, tau, ws, form);
        return;
    }
    if (!c->sytrd_graphs) c->sytrd_graphs = new SytrdGraphCache();
    SytrdGraphCache& gc = *c->sytrd_graphs;
    ++gc.clock;
    SytrdGraph* slot = nullptr;
    for (auto& g : gc.slots)
        if (g.exec && g.n == n64 && g.ld == ld && g.form == form && g.A == A && g.d == d && g.e == e && g.tau == tau && g.ws == ws) slot = &g;
    if (slot) ++gc.hits;
    if (!slot) {
        ++gc.misses;
        const auto t_build = std::chrono::steady_clock::now();
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
        bool ok = hipGraphCreate(&graph, 0) == hipSuccess && graph;
        if (ok) {
            SytrdEmitter em;
            em.graph = graph;
            ok = emit_sytrd_default(em, n64, A, ld, d, e, tau, ws, form);
        }
        if (ok) ok = hipGraphInstantiate(&victim->exec, graph, nullptr, nullptr, 0) == hipSuccess;
        if (graph) hipGraphDestroy(graph);
        if (!ok) {  // graph construction failed: plain launches
            victim->exec = nullptr;
            (void)hipGetLastError();
            launch_sytrd_direct(s, n64, A, ld, d, e, tau, ws, form);
            return;
        }
        victim->n = n64;
        victim->ld = ld;
        victim->form = form;
        victim->A = A;
        victim->d = d;
        victim->e = e;
        victim->tau = tau;
        victim->ws = ws;
        slot = victim;
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_build).count();
        gc.instantiate_ms += ms;
        if (dbg_on())
            fprintf(stderr, "[sdpsr] sytrd graph cache MISS: n=%lld ld=%lld form=%d built + instantiated in %.2f ms (hits %llu, misses %llu)\n",
                    (long long)n64, (long long)ld, form, ms, (unsigned long long)gc.hits, (unsigned long long)gc.misses);
    }
    slot->last_use = gc.clock;
    if (hipGraphLaunch(slot->exec, s) != hipSuccess) {
        (void)hipGetLastError();
        launch_sytrd_direct(s, n64, A, ld, d, e, tau, ws, form);
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
// pairs of a round come from register arithmetic and the step ping-pongs between two LDS copies.
//
// jacobi64_fill_pairs / jacobi64_sweeps are the workgroup-level core, shared by the single-problem
// kernel below and by the batched eigen_decomposition kernel (kernels_batched.hip includes this
// file's declarations through jacobi64.h).
template <bool PP>
__global__ void __launch_bounds__(JAC_MAXTHREADS)
small_syev_jacobi64_kernel(int n, double* __restrict__ Ag, int64_t lda, double* __restrict__ wout,
                           int* __restrict__ info) {
    extern __shared__ __attribute__((aligned(16))) double sA[];  // A, V and their ping-pong twins (m x m, ld ldl)
    __shared__ double s_red[2 * JAC_MAXTHREADS / 64];
    __shared__ int s_rank[64];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int m = (n + 1) & ~1;
    const int ldl = m | 1;
    double* sV = sA + (size_t)ldl * m;
    for (int e = tid; e < m * m; e += nthr) {
        const int j = e / m, i = e - j * m;
        const int ii = i > j ? i : j, jj = i > j ? j : i;
        sA[i + j * ldl] = (ii < n) ? Ag[ii + (int64_t)jj * lda] : 0.0;
        sV[i + j * ldl] = (i == j) ? 1.0 : 0.0;
    }
    int sweep;
    if (PP) {
        __syncthreads();
        sweep = jacobi64_sweeps_pp(n, m, ldl, sA, sV, sV + (size_t)ldl * m, sV + 2 * (size_t)ldl * m, s_red);
    } else {
        int* s_pq = reinterpret_cast<int*>(sV + (size_t)ldl * m);
        jacobi64_fill_pairs(m, s_pq);
        __syncthreads();
        sweep = jacobi64_sweeps(n, m, ldl, sA, sV, s_pq, s_red);
    }
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

bool small_syev_set_device_attributes() {
    bool ok = true;
    ok &= hipSuccess == hipFuncSetAttribute(reinterpret_cast<const void*>(&small_syev_jacobi64_kernel<true>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 136 * 1024);
    ok &= hipSuccess == hipFuncSetAttribute(reinterpret_cast<const void*>(&small_syev_jacobi64_kernel<false>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 136 * 1024);
    ok &= hipSuccess == hipFuncSetAttribute(reinterpret_cast<const void*>(&small_syev_jacobi_kernel),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    return ok;
}

bool launch_small_syev(hipStream_t s, int64_t n, double* A, int64_t lda, double* w, double* Vtmp, int* info) {
    if (n < 1 || n > JAC_MAXN) return false;
    size_t lds = (size_t)((n | 1) * n + 8) * sizeof(double);
    if (n <= 64) {
        const int half = (int)((n + 1) / 2), mm = 2 * half;
        int threads = (half * half + 63) / 64 * 64;
        if (threads < 64) threads = 64;
        const size_t lds64 = 4 * (size_t)(mm | 1) * mm * sizeof(double) + 64;
        small_syev_jacobi64_kernel<true><<<1, threads, lds64, s>>>((int)n, A, lda, w, info);
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
