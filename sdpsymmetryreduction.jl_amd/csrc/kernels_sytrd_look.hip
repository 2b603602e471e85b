// Panel form of the Householder tridiagonalisation with ONE launch per column (round 4; LAPACK dsytrd / dlatrd, lower;
// first phase of eigen(A), src/eigen_decomposition.jl:246).
//
// The two-launch form of kernels_sytrd.hip has two chip-wide dependencies per column: the norm of the updated column
// a_j (-> the reflector v_j), then the product A0 v_j.  The second one waits for the first only through a SCALING:
//     v_j = (a_j - beta e) / (alpha - beta)      (e: unit vector of the pivot row, alpha = a_j[pivot])
//     A0 v_j = (A0 a_j - beta A0 e) / (alpha - beta)
// and A0 e is a stored column of A0.  So launch j multiplies the panel-start matrix by the UNNORMALISED column,
// z_j = A0 a_j, and leaves the norm of a_j as per-workgroup partial sums beside it; launch j + 1 finishes the reflector
// (beta, tau), y_j = A0 v_j, w_j and the next column a_{j+1} from those, redundantly in every workgroup:
//   * a workgroup is one 128 x 128 tile (I, J) of the lower triangle, so it needs the vectors on the 256 rows of the
//     blocks I and J only: their partial products z (one slot per partner block, written by the tiles of launch j,
//     summed here in a fixed order), their rows of the panel [V | W], their entries of column j + 1;
//   * every chip-wide sum of the step is a sum of records the previous launch left behind, added up in the same order
//     by every workgroup (bitwise the same scalars everywhere):  |a_j|^2,  V'a_j and W'a_j (-> V'v, W'v),
//     a_j' A0 a_j  (-> y'v = (a'z - 2 beta z[pivot] + beta^2 A0[pivot, pivot]) / (alpha - beta)^2);
//   * the tile (I, I) owns the rows of block I: it stores v_j, w_j into the panel, the reflector into A (LAPACK
//     storage), the column a_{j+1} and the records of its rows.
// One kernel boundary per column instead of two; the rank-64 update of the trailing matrix after 32 columns is the MFMA
// kernel of kernels_sytrd.hip (a `finish` launch of the diagonal tiles completes the panel's last column first).
// Cost of the redundancy: a tile reads, besides its 128 KiB of the matrix, <= 124 KiB of panel rows (the stored columns only,
// from a column-major copy of the panel: a thread owns a row, its loads are coalesced and its dots need no reduction) and
// <= 64 KiB of partial products, from L2 / Infinity Cache.
#include "sdpsr_internal.h"

namespace sdpsr {

// development aid (-DLK_TIMING): wall-clock stamps (100 MHz) of the phases of three workgroups of a launch
#ifdef LK_TIMING
__device__ long long* lk_dbg = nullptr;
#define LK_STAMP(i)                                                                                                   \
    do {                                                                                                               \
        if (lk_dbg && tid == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1 || blockIdx.x == gridDim.x / 2)) {  \
            const int which = blockIdx.x == 0 ? 0 : (blockIdx.x == gridDim.x - 1 ? 2 : 1);                            \
            lk_dbg[((int64_t)j * 3 + which) * 16 + (i)] = wall_clock64();                                              \
        }                                                                                                              \
    } while (0)
#else
#define LK_STAMP(i)
#endif

constexpr int LK_NB = 32;  // panel width (= SY_NB of kernels_sytrd.hip: same panel layout, same trailing update)
constexpr int LK_PW = 2 * LK_NB;
constexpr int LK_T = 128;
constexpr int LK_THREADS = 256;

__device__ __forceinline__ double lk_wave_sum(double x) {  // butterfly: every lane ends with the same bits
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}

// MODE 0: first column of a panel (nothing to finish; a_j is the stored column)
// MODE 1: finish column j - 1 (panel column cf - 1), form a_j, multiply
// MODE 2: finish column j - 1 only (last column of a panel; grid = the diagonal tiles)
// NS: slots of partial products a row can have (blocks of the matrix: 32 up to ld = 4096, else 64) -- every load of the
// prologue is issued in ONE batch of a static shape (a loop with a load per trip is a memory round trip per trip on the
// critical path of the column)
template <int MODE, int NS>
__global__ void __launch_bounds__(LK_THREADS, 1)
sytrd_look_kernel(SytrdLookArgs a, int j, int cf, int n_vaz_prev) {
    constexpr int NVZ = NS == 32 ? 5 : 17;  // records of a'(A0 a) per thread: 2 per tile, NS (NS + 1) / 2 tiles
    constexpr bool FIN = MODE != 0;
    constexpr bool PROD = MODE != 2;
    __shared__ double s_graw[4][LK_PW], s_cv[LK_PW], s_pj[LK_PW];
    __shared__ double s_a[256];
    __shared__ double s_y[4][LK_T], s_zt[LK_T];
    __shared__ double s_vavw[4], s_scal[8], s_red[4][4];
    __shared__ double s_acc[4][LK_PW];
    const int tid = threadIdx.x;
    const int lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t ld = a.ld;
    const int n = a.n;
    const int nb = (n + LK_T - 1) / LK_T;
    const int nbmax = (int)(ld / LK_T);
    const int Bmin = j / LK_T;
    const int nbact = nb - Bmin;
    int bi, bj;
    {
        int t = blockIdx.x;
        if (t < nbact) {
            bi = bj = t;
        } else {
            t -= nbact;  // strictly lower pairs: t = bi (bi - 1) / 2 + bj, bi > bj
            int row = (int)((sqrtf(8.0f * (float)t + 1.0f) + 1.0f) * 0.5f);
            while (row * (row - 1) / 2 > t) --row;
            while ((row + 1) * row / 2 <= t) ++row;
            bi = row;
            bj = t - row * (row - 1) / 2;
        }
    }
    const bool diag = bi == bj;
    const int I0 = LK_T * (Bmin + bi), J0 = LK_T * (Bmin + bj);
    const int par = j & 1, pp = par ^ 1;
    const int jf = j - 1, cfm = cf - 1;  // the column being finished and its place in the panel
    const int r = tid < LK_T ? I0 + tid : J0 + tid - LK_T;

    LK_STAMP(0);
    // ---- every global load is issued here, before the first barrier
    double2 x[32];
    if constexpr (PROD) {
        const char* cbase = reinterpret_cast<const char*>(a.A + (int64_t)(J0 + 32 * w) * ld) + (size_t)(I0 + 2 * lane) * 8;
#pragma unroll
        for (int cc = 0; cc < 32; ++cc) x[cc] = *reinterpret_cast<const double2*>(cbase + (int64_t)cc * ld * 8);
    }
    double zr = 0;  // z_{j-1}[r] = (A0 a_{j-1})[r]: the partial products of launch j - 1, one per partner block
    double aprev = 0;
    double pc[LK_PW];  // row r of the panel [V | W], read from its column-major copy (thread per row: coalesced, no reduction)
    double pr[32];     // diagonal tile: 32 of its rows per wave, row-major (lane = panel entry), for the records of the block
    double xn2 = 0, alpha = 0, a0jj = 0, vav = 0, psj = 0;
    // column j of A0: a_j before its update, and A0[:, pivot] of column j - 1
    const double rawcol = (r >= j && r < n) ? a.A[r + (int64_t)j * ld] : 0.0;
    if constexpr (FIN) {
        // loads only, into registers: a value that is consumed (or stored to LDS) between two loads makes the later one wait
        // for a memory round trip of its own
        const int Bp = jf / LK_T;
        const int cnt = nb - Bp;
        const double* Pp = a.P + ((int64_t)pp * nbmax + Bp) * ld + r;
        double tvz[NS], tvd[NS / 4], tvv[NVZ];
#pragma unroll
        for (int u = 0; u < NS; ++u) tvz[u] = Pp[(int64_t)(u < cnt ? u : 0) * ld];
        aprev = a.acol[(int64_t)pp * ld + r];
        const bool panel = cfm > 0;
        const int k = tid & (LK_PW - 1), part = tid >> 6;
#pragma unroll
        for (int k = 0; k < LK_NB; ++k) {
            const bool need = k < cfm;  // a stored panel column (uniform)
            pc[k] = need ? a.PTc[(int64_t)k * ld + r] : 0.0;
            pc[LK_NB + k] = need ? a.PTc[(int64_t)(LK_NB + k) * ld + r] : 0.0;
        }
        if (PROD && diag && panel) {
#pragma unroll
            for (int i = 0; i < 32; ++i) pr[i] = a.PT[(int64_t)(I0 + 32 * w + i) * LK_PW + lane];
        }
        const double pjv = (panel && tid < LK_PW) ? a.PT[(int64_t)j * LK_PW + tid] : 0.0;
        // panel dots of a_{j-1}: 256 threads add the records of the diagonal tiles, four parts per entry
#pragma unroll
        for (int u = 0; u < NS / 4; ++u) {
            const int b = part + 4 * u;
            tvd[u] = (panel && b < cnt) ? a.rec_dots[((int64_t)pp * nbmax + Bp + b) * LK_PW + k] : 0.0;
        }
        // |a_{j-1}[pivot+1:]|^2: every wave adds the records itself (same loads, same tree: the same bits everywhere)
        xn2 = (lane < cnt) ? a.rec_norm[(int64_t)pp * nbmax + Bp + lane] : 0.0;
        alpha = a.acol[(int64_t)pp * ld + j];
        a0jj = a.A[j + (int64_t)j * ld];
#pragma unroll
        for (int u = 0; u < NVZ; ++u) {
            const int b = tid + LK_THREADS * u;
            tvv[u] = b < n_vaz_prev ? a.rec_vaz[(int64_t)pp * a.vaz_cap + b] : 0.0;
        }
        if (w == 3) psj = (lane < cnt) ? a.P[((int64_t)pp * nbmax + Bp + lane) * ld + j] : 0.0;  // z_{j-1}[pivot]
        LK_STAMP(1);
        // ---- consumption
#pragma unroll
        for (int u = 0; u < NS; ++u) tvz[u] = u < cnt ? tvz[u] : 0.0;
#pragma unroll
        for (int h = NS / 2; h > 0; h >>= 1)
#pragma unroll
            for (int u = 0; u < h; ++u) tvz[u] += tvz[u + h];
        zr = tvz[0];
#pragma unroll
        for (int h = NS / 8; h > 0; h >>= 1)
#pragma unroll
            for (int u = 0; u < h; ++u) tvd[u] += tvd[u + h];
        s_graw[part][k] = tvd[0];
        if (tid < LK_PW) s_pj[tid] = pjv;
#pragma unroll
        for (int u = 0; u < NVZ; ++u) vav += tvv[u];
    }

    double beta = 0, tau = 0, scale = 0, sf = 0, sm = 0;
    if constexpr (FIN) {
        xn2 = lk_wave_sum(xn2);
        if (xn2 == 0.0) {  // dlarfg: H = I
            tau = 0.0;
            beta = alpha;
            scale = 0.0;
        } else {
            beta = -copysign(sqrt(alpha * alpha + xn2), alpha);
            tau = (beta - alpha) / beta;
            scale = 1.0 / (alpha - beta);
        }
        vav = lk_wave_sum(vav);
        if (lane == 0) s_vavw[w] = vav;
        if (w == 3) {
            psj = lk_wave_sum(psj);
            if (lane == 0) s_scal[2] = psj;
        }
        LK_STAMP(2);
        __syncthreads();  // B1
        if (tid < LK_PW) {
            // (X'v)[k] = (X'a - beta X[pivot, k]) / (alpha - beta);  k < 32: X = V, k >= 32: X = W
            const double raw = (s_graw[0][tid] + s_graw[1][tid]) + (s_graw[2][tid] + s_graw[3][tid]);
            s_cv[tid] = ((tid & (LK_NB - 1)) < cfm && cfm > 0) ? (raw - beta * s_pj[tid]) * scale : 0.0;
        }
        __syncthreads();  // B2
        LK_STAMP(3);
        if (w == 1) {
            double gg = (lane < cfm) ? s_cv[lane] * s_cv[lane + LK_NB] : 0.0;
            gg = lk_wave_sum(gg);
            if (lane == 0) {
                const double vaz = (s_vavw[0] + s_vavw[1]) + (s_vavw[2] + s_vavw[3]);
                const double zj = s_scal[2];
                const double yv = ((vaz - 2.0 * beta * zj) + beta * beta * a0jj) * scale * scale;  // y'v, y = A0 v
                const double dot = tau * (yv - 2.0 * gg);                                       // w~'v, w~ = tau (y - V c1 - W c2)
                s_scal[1] = -0.5 * tau * dot;
            }
        }
        if (w == 2) {  // row `pivot` of the panel against the multipliers (needed by every row below)
            double sfj = ((lane & (LK_NB - 1)) < cfm) ? s_pj[lane] * s_cv[lane ^ LK_NB] : 0.0;
            sfj = lk_wave_sum(sfj);
            if (lane == 0) s_scal[3] = sfj;
        }
        // V(r,:) (W'v) + W(r,:) (V'v): the multiplier of an entry is the other half's dot;  V(r,:) W(j,:)' + W(r,:) V(j,:)':
        // the other half's entry of row j.  (entries beyond the stored columns are zero in pc and in s_cv)
        if (cfm > 0) {
#pragma unroll
            for (int k = 0; k < LK_PW; ++k) {
                sf = fma(pc[k], s_cv[k ^ LK_NB], sf);
                sm = fma(pc[k], s_pj[k ^ LK_NB], sm);
            }
        }
        __syncthreads();  // B3
        LK_STAMP(4);
    }
    // ---- the row of this thread: v_{j-1}, w_{j-1}, then a_j
    double v = 0, wv = 0, anew = 0;
    const bool live = r >= j && r < n;
    if constexpr (FIN) {
        const double alpha2 = s_scal[1];
        const double yj = (s_scal[2] - beta * a0jj) * scale;
        const double wj = tau * (yj - s_scal[3]) + alpha2;  // w_{j-1}[pivot]; v_{j-1}[pivot] = 1
        if (live) {
            v = (r == j) ? 1.0 : aprev * scale;
            const double y = (zr - beta * rawcol) * scale;
            wv = tau * (y - sf) + alpha2 * v;
            anew = rawcol - (sm + (v * wj + wv));
        }
        if (diag && tid < LK_T && live) {
            a.PT[(int64_t)r * LK_PW + cfm] = v;
            a.PT[(int64_t)r * LK_PW + LK_NB + cfm] = wv;
            a.PTc[(int64_t)cfm * ld + r] = v;
            a.PTc[(int64_t)(LK_NB + cfm) * ld + r] = wv;
            if (r >= j + 1) a.A[r + (int64_t)jf * ld] = v;  // LAPACK storage of the finished reflector
        }
        if (blockIdx.x == 0 && tid == 0) {
            a.e[jf] = beta;
            a.tau[jf] = tau;
        }
    } else {
        anew = rawcol;
    }
    if constexpr (!PROD) return;
    const double aprod = (r >= j + 1 && r < n) ? anew : 0.0;
    s_a[tid] = aprod;
    if (diag && tid < LK_T) {
        if (live) a.acol[(int64_t)par * ld + r] = anew;
        if (r == j) a.d[j] = anew;
        // records of the rows of this block: |a_j[j+2:]|^2 and the panel dots with the column just finished
        double sq = (r >= j + 2 && r < n) ? anew * anew : 0.0;
        double pv = v * aprod, pw = wv * aprod;
        sq = lk_wave_sum(sq);
        pv = lk_wave_sum(pv);
        pw = lk_wave_sum(pw);
        if (lane == 0) {
            s_red[w][0] = sq;
            s_red[w][1] = pv;
            s_red[w][2] = pw;
        }
    }
    LK_STAMP(5);
    __syncthreads();  // B4
    LK_STAMP(6);
    // ---- the tile: direct sums (rows of I against a[J]) and transposed sums (columns of J against a[I])
    const bool ragged = I0 + LK_T > n;
    if (ragged) {
#pragma unroll
        for (int cc = 0; cc < 32; ++cc) {
            const bool cok = J0 + 32 * w + cc < n;
            if (!cok || I0 + 2 * lane >= n) x[cc].x = 0.0;
            if (!cok || I0 + 2 * lane + 1 >= n) x[cc].y = 0.0;
        }
    }
    const double aI0 = s_a[2 * lane], aI1 = s_a[2 * lane + 1];
    const double* saJ = s_a + LK_T + 32 * w;
    double y0 = 0, y1 = 0;
    double zc[32];
    if (!diag) {
#pragma unroll
        for (int cc = 0; cc < 32; ++cc) {
            const double aj = saJ[cc];
            y0 = fma(x[cc].x, aj, y0);
            y1 = fma(x[cc].y, aj, y1);
            zc[cc] = fma(x[cc].y, aI1, x[cc].x * aI0);
        }
    } else {  // rows >= columns for the direct part, > for the transposed one
#pragma unroll
        for (int cc = 0; cc < 32; ++cc) {
            const int col = 32 * w + cc, row0 = 2 * lane;
            const double aj = saJ[cc];
            const double d0 = row0 >= col ? x[cc].x : 0.0, d1 = row0 + 1 >= col ? x[cc].y : 0.0;
            const double t0 = row0 > col ? x[cc].x : 0.0, t1 = row0 + 1 > col ? x[cc].y : 0.0;
            y0 = fma(d0, aj, y0);
            y1 = fma(d1, aj, y1);
            zc[cc] = fma(t1, aI1, t0 * aI0);
        }
    }
    // transposed sums: reduce-scatter over the 64 lanes (rows), 32 columns per lane
#pragma unroll
    for (int st = 0; st < 5; ++st) {
        const int half = 16 >> st, step = 32 >> st;
        const bool upper = (lane & step) != 0;
#pragma unroll
        for (int i = 0; i < half; ++i) {
            const double send = upper ? zc[i] : zc[i + half];
            const double keep = upper ? zc[i + half] : zc[i];
            zc[i] = keep + __shfl_xor(send, step, 64);
        }
    }
    zc[0] += __shfl_xor(zc[0], 1, 64);
    {
        const int cc = (((lane >> 5) & 1) << 4) | (((lane >> 4) & 1) << 3) | (((lane >> 3) & 1) << 2) | (((lane >> 2) & 1) << 1) |
                       ((lane >> 1) & 1);
        if ((lane & 1) == 0) s_zt[32 * w + cc] = zc[0];
    }
    s_y[w][2 * lane] = y0;
    s_y[w][2 * lane + 1] = y1;
    // the diagonal tile's panel dots with the stored columns
    if (FIN && diag && cfm > 0) {
        double acc = 0;
#pragma unroll
        for (int i = 0; i < 32; ++i) acc = fma(pr[i], s_a[32 * w + i], acc);
        s_acc[w][lane] = acc;
    }
    LK_STAMP(7);
    __syncthreads();  // B5
    LK_STAMP(8);
    const int slotI = Bmin + bi, slotJ = Bmin + bj;
    if (tid < LK_T) {
        double tot = (s_y[0][tid] + s_y[1][tid]) + (s_y[2][tid] + s_y[3][tid]);
        double vz;
        if (diag) {
            tot += s_zt[tid];
            vz = s_a[tid] * tot;
        } else {
            vz = 2.0 * s_a[tid] * tot;
        }
        a.P[((int64_t)par * nbmax + slotJ) * ld + I0 + tid] = tot;
        vz = lk_wave_sum(vz);
        if (lane == 0) a.rec_vaz[(int64_t)par * a.vaz_cap + 2 * blockIdx.x + w] = vz;
    } else if (!diag) {
        a.P[((int64_t)par * nbmax + slotI) * ld + J0 + tid - LK_T] = s_zt[tid - LK_T];
    }
    if (diag) {
        if (tid == 0) a.rec_norm[(int64_t)par * nbmax + slotI] = s_red[0][0] + s_red[1][0];
        if (tid < LK_PW) {
            double g = 0.0;
            if (FIN && cfm > 0 && (tid & (LK_NB - 1)) < cfm) g = (s_acc[0][tid] + s_acc[1][tid]) + (s_acc[2][tid] + s_acc[3][tid]);
            if (FIN && tid == cfm) g = s_red[0][1] + s_red[1][1];
            if (FIN && tid == LK_NB + cfm) g = s_red[0][2] + s_red[1][2];
            a.rec_dots[((int64_t)par * nbmax + slotI) * LK_PW + tid] = g;
        }
    }
    LK_STAMP(9);
}

#ifdef LK_TIMING
void sytrd_look_debug_buffer(long long* p) { hipMemcpyToSymbol(HIP_SYMBOL(lk_dbg), &p, sizeof(p)); }
#endif
const void* sytrd_look_kernel_fn(int mode, int64_t ld) {
    if (ld <= 32 * LK_T)
        return mode == 0   ? reinterpret_cast<const void*>(&sytrd_look_kernel<0, 32>)
               : mode == 1 ? reinterpret_cast<const void*>(&sytrd_look_kernel<1, 32>)
                           : reinterpret_cast<const void*>(&sytrd_look_kernel<2, 32>);
    return mode == 0   ? reinterpret_cast<const void*>(&sytrd_look_kernel<0, 64>)
           : mode == 1 ? reinterpret_cast<const void*>(&sytrd_look_kernel<1, 64>)
                       : reinterpret_cast<const void*>(&sytrd_look_kernel<2, 64>);
}

}  // namespace sdpsr
