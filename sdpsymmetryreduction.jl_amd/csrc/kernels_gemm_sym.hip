// C = X'X for T int8 channel matrices X (ld x ld, symmetric labels => X symmetric, so X'X = X X), the random square of
// the refinement loop:   mul!(X2, X, X)   src/partitions.jl:172
//
// One PERSISTENT launch: at most one workgroup per CU (8 waves, 128 KiB of LDS), every workgroup walks a static list of
// jobs.  Why not the 128 x 128 tiles of kernels_gemm.hip (round 3's product launch, 92 - 98 us at N = 4096, 2 channels):
// its counters (profiles/r04_pmc.json, i8x2_lower_128tiles_counters) say a tile takes 62.7 k clocks where its 512 MFMAs per wave
// need 32.8 k (two workgroups per CU), because the global -> LDS feed is at its per-CU limit -- 64 KiB per K-tile of
// 128 bytes and CU in 1960 clocks = 33 B/clk/CU = 70 GB/s per CU, the rate MI355X_MICROARCH.md gives for LDS gathers
// served by the L2 -- and that the 1056 tiles take three rounds on the 512 resident slots although they are 2.06 rounds
// of work.  Hence
//   * 256 x 256 macro-tiles (a wave owns 128 x 64): half the feed bytes per MFMA, 32 B/clk/CU at the full MFMA rate;
//   * the diagonal macro-tiles two per job, their 2 x 36 lower 32 x 32 blocks dealt 9 to a wave (a "full" wave has 8), so
//     that N = 4096, 2 channels is 240 + 16 = 256 jobs of 1.0 / 1.125 tile times: ONE round on 256 CUs, no tail;
//   * quarter tiles (128 x 128, 8 waves of 64 x 32) for the ragged last 128 rows of an order that is an odd multiple of
//     128 (N = 4104 -> 4224: 256 big jobs + 66 quarter tiles in a short second round);
//   * the K walk in a ring of four 64-byte LDS stages filled by LDS-DMA with counted waits (vmcnt(N), never a drain
//     inside the loop), so the feed does not idle between a stage's arrival and the next issue.
// 76 - 78 us at N = 4096, 2 channels (0.36 of the nominal int8 peak, MfmaUtil 0.40, 193 MB moved for 168 MB algorithmic).
// Where the rest goes, from ablation builds (profiles/r04_i8sym_ablation.txt): the MFMAs alone take 46 us (the matrix
// pipe under random int8 operands runs at ~1.6 GHz under the board's power cap), fragment reads + barriers add 7, the
// DMA 14, the 69 MB of results 9.5 -- the stages add up instead of overlapping, which is what a power budget does.
// The device flag decides as before: *nonsym_flag != 0 => every macro-tile of the full squares (no diagonal jobs).
//
// Exact integers (int32 accumulation of int8 products, |sum| <= 2^14 ld): the order in which K is consumed is free.
#include "sdpsr_internal.h"

namespace sdpsr {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void_t;

// 16 bytes per lane global -> LDS, lane-linear at the wave-uniform LDS byte address `lds_dst` (see glds16 in
// kernels_gemm.hip for why this is inline asm and how it is ordered: explicit counted waits + barriers below)
__device__ __forceinline__ void glds16_u(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

// Source-side XOR swizzle of the 16-byte chunks of an LDS row (slot p of row r holds global chunk p ^ swz(r)) that
// makes the ds_read_b128 fragment reads conflict-free (lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} of the b128
// banking, MI355X_MICROARCH.md): rows of 128 bytes alternate between the two halves of the 64 banks, so the 8 rows of a
// group with the same parity need 8 different slots; rows of 64 bytes repeat every 4 rows, the 4 rows of a group with
// the same r mod 4 have (r >> 2) in {0,3,5,6} or {1,2,4,7} and need 4 different slots.
template <int KB> __device__ __forceinline__ int sym_swz(int r) {
    if constexpr (KB == 128) return (r >> 1) & 7;
    else return ((r >> 3) ^ (r >> 2)) & 3;
}

// Diagonal macro-tile: the 36 blocks (r, c), c <= r, of its 8 x 8 grid of 32 x 32 blocks, dealt to four waves.
// fr[]: the row blocks a wave reads (a fragment serves as either MFMA operand: both are cut the same way);
// tr[] / tc[]: positions in fr[] of the row / column block of the wave's nine accumulators.
struct DiagRole {
    int nf;
    int fr[7];
    int tr[9];
    int tc[9];
};
__device__ constexpr DiagRole kDiagRoles[4] = {
    {6, {3, 4, 5, 0, 1, 2, 0}, {0, 0, 0, 1, 1, 1, 2, 2, 2}, {3, 4, 5, 3, 4, 5, 3, 4, 5}},  // rows 3-5 x columns 0-2
    {7, {6, 7, 0, 1, 2, 3, 4}, {0, 0, 0, 1, 1, 1, 0, 1, 1}, {2, 3, 4, 2, 3, 4, 5, 5, 6}},  // rows 6-7 x columns 0-2, (6,3) (7,3) (7,4)
    {5, {0, 1, 2, 3, 4, 0, 0}, {0, 1, 1, 2, 2, 2, 3, 4, 4}, {0, 0, 1, 0, 1, 2, 3, 3, 4}},  // triangle of rows 0-2, (3,3) (4,3) (4,4)
    {5, {3, 4, 5, 6, 7, 0, 0}, {2, 2, 2, 3, 3, 3, 4, 4, 4}, {0, 1, 2, 1, 2, 3, 2, 3, 4}},  // (5,3..5) (6,4..6) (7,5..7)
};

struct SymSquareArgs {
    const int8_t* X;
    int32_t* C;
    int64_t ldx, ldc, strideX, strideC;
    int k;  // rows of X (bytes of K per operand row), a multiple of the stage width
    int m;  // whole macro-tile rows: ld / 256 (rounded down)
    int rem;  // 1: ld is an odd multiple of 128 (a last unit row / column of quarter tiles)
    int T;  // channels
    const uint32_t* nonsym_flag;
};

constexpr int SYM_NT = 512;

template <int KB> struct SymCfg {
    static constexpr int NS = (KB == 64) ? 4 : 2;  // LDS ring stages (128 KiB either way)
    static constexpr int D = NS - 1;               // stages in flight ahead of the one being multiplied
    static constexpr int CH = KB / 16;             // 16-byte chunks per LDS row
    static constexpr int RPI = 1024 / KB;          // rows per DMA wave-instruction
    static constexpr int IPW = 256 / RPI / 8;      // DMA instructions per wave, operand panel and stage
    static constexpr int OPB = 256 * KB;           // bytes of one operand panel per stage
    static constexpr int STAGE = 2 * OPB;
    static constexpr int NQ = KB / 32;             // MFMA K-groups (32 bytes) per stage
};

// what a wave needs to walk K for one job
struct SymLane {
    unsigned lds0;  // LDS byte address of the ring
    int wave, lane_off, r32, h;
};

// One job of one wave: ROLE -1 = its 128 x 64 part of a full macro-tile (rows from panel A, columns from panel B),
// ROLE -2 = its 64 x 32 part of a 128 x 128 "quarter" tile (the ragged last 128 rows / columns of a matrix whose order
// is an odd multiple of 128: panels of 128 rows, half the DMA instructions),
// ROLE 0..3 = nine blocks of a diagonal macro-tile (waves 0-3: the tile of panel A, waves 4-7: that of panel B).
// Every wave of the workgroup passes the same nk + 1 barriers whatever its role.
template <int KB, int ROLE, bool LATE>
__device__ __forceinline__ void sym_job(const SymLane& L, const char* smem, const int8_t* pA, const int8_t* pB, int32_t* Cout, int64_t ldc,
                                        int nk, const int (&soff)[ROLE == -2 ? 1 : SymCfg<KB>::IPW], const int (&coff)[SymCfg<KB>::NQ]) {
    typedef SymCfg<KB> CF;
    constexpr bool QUARTER = ROLE == -2;
    constexpr int NACC = QUARTER ? 2 : (ROLE < 0 ? 8 : 9);
    constexpr int NIS = QUARTER ? 1 : CF::IPW;  // DMA instructions per wave, panel and stage
    const int wave = L.wave;
    const int wi = wave & 1, wj = wave >> 1;
    v16i acc[NACC];
#pragma unroll
    for (int t = 0; t < NACC; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0;

    auto issue = [&](int slot, int kt) {
        const unsigned dst = L.lds0 + (unsigned)(slot * CF::STAGE + wave * NIS * 1024);
        const char* ga = reinterpret_cast<const char*>(pA) + (int64_t)kt * KB;
        const char* gb = reinterpret_cast<const char*>(pB) + (int64_t)kt * KB;
#pragma unroll
        for (int s = 0; s < NIS; ++s) {
            glds16_u(ga + soff[s], dst + s * 1024);
            glds16_u(gb + soff[s], dst + CF::OPB + s * 1024);
        }
    };
#pragma unroll
    for (int st = 0; st < CF::D; ++st)
        if (st < nk) issue(st, st);

    // fragments of one 32-byte K-group and the MFMAs on them, by role
    constexpr int NFR = QUARTER ? 3 : (ROLE < 0 ? 6 : kDiagRoles[ROLE < 0 ? 0 : ROLE].nf);
    auto load = [&](uint4 (&F)[NFR], const char* sA, const char* sB, int q) {
        if constexpr (QUARTER) {
#pragma unroll
            for (int t = 0; t < 2; ++t) F[t] = *reinterpret_cast<const uint4*>(sA + (wi * 64 + t * 32) * KB + L.lane_off + coff[q]);
            F[2] = *reinterpret_cast<const uint4*>(sB + (wj * 32) * KB + L.lane_off + coff[q]);
        } else if constexpr (ROLE < 0) {
#pragma unroll
            for (int t = 0; t < 4; ++t) F[t] = *reinterpret_cast<const uint4*>(sA + (wi * 128 + t * 32) * KB + L.lane_off + coff[q]);
#pragma unroll
            for (int t = 0; t < 2; ++t) F[4 + t] = *reinterpret_cast<const uint4*>(sB + (wj * 64 + t * 32) * KB + L.lane_off + coff[q]);
        } else {
            constexpr DiagRole R = kDiagRoles[ROLE < 0 ? 0 : ROLE];
            const char* sP = (wave < 4) ? sA : sB;
#pragma unroll
            for (int x = 0; x < R.nf; ++x) F[x] = *reinterpret_cast<const uint4*>(sP + R.fr[x] * 32 * KB + L.lane_off + coff[q]);
        }
    };
    auto mma = [&](const uint4 (&F)[NFR]) {
        if constexpr (QUARTER) {
#pragma unroll
            for (int ti = 0; ti < 2; ++ti) {
                v4i av = {(int)F[2].x, (int)F[2].y, (int)F[2].z, (int)F[2].w};
                v4i bv = {(int)F[ti].x, (int)F[ti].y, (int)F[ti].z, (int)F[ti].w};
                acc[ti] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, bv, acc[ti], 0, 0, 0);
            }
        } else if constexpr (ROLE < 0) {
#pragma unroll
            for (int tj = 0; tj < 2; ++tj)
#pragma unroll
                for (int ti = 0; ti < 4; ++ti) {
                    v4i av = {(int)F[4 + tj].x, (int)F[4 + tj].y, (int)F[4 + tj].z, (int)F[4 + tj].w};
                    v4i bv = {(int)F[ti].x, (int)F[ti].y, (int)F[ti].z, (int)F[ti].w};
                    acc[tj * 4 + ti] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, bv, acc[tj * 4 + ti], 0, 0, 0);
                }
        } else {
            constexpr DiagRole R = kDiagRoles[ROLE < 0 ? 0 : ROLE];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const uint4 fa = F[R.tc[t]], fb = F[R.tr[t]];
                v4i av = {(int)fa.x, (int)fa.y, (int)fa.z, (int)fa.w};
                v4i bv = {(int)fb.x, (int)fb.y, (int)fb.z, (int)fb.w};
                acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, bv, acc[t], 0, 0, 0);
            }
        }
    };

    // The two waves of a SIMD (w and w + 4) walk K half a stage apart: waves 4-7 ("late") keep the fragments of a
    // stage's last K-group in registers across the barrier and multiply them right behind it, while waves 0-3 wait for
    // their first reads of the new stage.
    constexpr bool late = LATE;
    uint4 Fc[NFR];  // the late group's carried fragments
    for (int kt = 0; kt < nk; ++kt) {
        // stage kt has landed once at most the D - 1 younger stages of this wave are outstanding
        if (kt + CF::D - 1 < nk) {
            if constexpr (QUARTER) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // (D - 1) * 2 * NIS
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if constexpr (late) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the carried fragments are out of the ring
        __syncthreads();  // everybody's part of stage kt is in; nobody reads stage kt - 1 any more
        if (kt + CF::D < nk) issue((kt + CF::D) % CF::NS, kt + CF::D);
        const char* sA = smem + (kt % CF::NS) * CF::STAGE;
        const char* sB = sA + CF::OPB;
        if constexpr (!late) {
#pragma unroll
            for (int q = 0; q < CF::NQ; ++q) {
                uint4 F[NFR];
                load(F, sA, sB, q);
                mma(F);
            }
        } else {
            if (kt > 0) mma(Fc);
#pragma unroll
            for (int q = 0; q < CF::NQ - 1; ++q) {
                uint4 F[NFR];
                load(F, sA, sB, q);
                mma(F);
            }
            load(Fc, sA, sB, CF::NQ - 1);
        }
    }
    if constexpr (late) {
        if (nk > 0) mma(Fc);
    }
    __syncthreads();  // the ring is free for the next job's first stages

    // ---- results: D[jj][ii], ii = lane & 31, jj = (reg & 3) + 8 (reg >> 2) + 4 h ----
    if (!Cout) return;
    if constexpr (QUARTER) {
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) {
            int32_t* Ct = Cout + (wi * 64 + ti * 32 + L.r32) + (int64_t)(wj * 32 + 4 * L.h) * ldc;
#pragma unroll
            for (int r = 0; r < 16; ++r) Ct[(int64_t)((r & 3) + 8 * (r >> 2)) * ldc] = acc[ti][r];
        }
    } else if constexpr (ROLE < 0) {
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int ti = 0; ti < 4; ++ti) {
                int32_t* Ct = Cout + (wi * 128 + ti * 32 + L.r32) + (int64_t)(wj * 64 + tj * 32 + 4 * L.h) * ldc;
#pragma unroll
                for (int r = 0; r < 16; ++r) Ct[(int64_t)((r & 3) + 8 * (r >> 2)) * ldc] = acc[tj * 4 + ti][r];
            }
    } else {
        constexpr DiagRole R = kDiagRoles[ROLE < 0 ? 0 : ROLE];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            int32_t* Ct = Cout + (R.fr[R.tr[t]] * 32 + L.r32) + (int64_t)(R.fr[R.tc[t]] * 32 + 4 * L.h) * ldc;
#pragma unroll
            for (int r = 0; r < 16; ++r) Ct[(int64_t)((r & 3) + 8 * (r >> 2)) * ldc] = acc[t][r];
        }
    }
}

template <int KB>
__global__ void __launch_bounds__(SYM_NT) i8_symsquare_kernel(SymSquareArgs a) {
    typedef SymCfg<KB> CF;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    SymLane L;
    L.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    L.r32 = lane & 31;
    L.h = lane >> 5;
    L.lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_void_t*)smem);
    L.lane_off = L.r32 * KB;
    const int wave = L.wave;

    const bool sym = a.nonsym_flag && *a.nonsym_flag == 0u;  // uniform
    const int m = a.m, T = a.T;                 // m whole macro-tile rows; a.rem = 1: 128 more rows / columns
    const int tri = m * (m - 1) / 2;
    const int ndt = sym ? T * m : 0;            // diagonal macro-tiles ...
    const int ND = (ndt + 1) >> 1;              // ... two per job
    const int NF = sym ? T * tri : T * m * m;   // full macro-tiles
    const int nqc = a.rem ? (sym ? 2 * m + 1 : 4 * m + 1) : 0;  // quarter tiles per channel: the last unit row (+ column)
    const int NQ4 = T * nqc;
    const int NJ = NF + ND + NQ4;
    const int nk = a.k / KB;

    // DMA sources of this lane inside a panel (K offset excluded)
    int soff[CF::IPW];
#pragma unroll
    for (int s = 0; s < CF::IPW; ++s) {
        const int r = (wave * CF::IPW + s) * CF::RPI + lane / CF::CH;
        const int c = (lane % CF::CH) ^ sym_swz<KB>(r);
        soff[s] = r * (int)a.ldx + c * 16;
    }
    int soffq[1];  // quarter tiles: this wave's 16 rows of a 128-row panel
    {
        const int r = wave * CF::RPI + lane / CF::CH;
        soffq[0] = r * (int)a.ldx + ((lane % CF::CH) ^ sym_swz<KB>(r)) * 16;
    }
    // fragment reads: row r32 of a 32-row block, chunk 2 q + h, un-swizzled
    int coff[CF::NQ];
#pragma unroll
    for (int q = 0; q < CF::NQ; ++q) coff[q] = ((2 * q + L.h) ^ sym_swz<KB>(L.r32)) << 4;

    for (int job = blockIdx.x; job < NJ; job += gridDim.x) {
        if (job < NF) {
            // workgroup b sits on XCD b % 8: every XCD gets one contiguous run of the tile sequence (neighbours in
            // the run share operand panels through that XCD's L2); the last NF % 8 jobs keep their number
            const int per = NF >> 3;
            const int f = (job < 8 * per) ? (job & 7) * per + (job >> 3) : job;
            int c, I, J;
            if (sym) {
                c = f / tri;
                const int t = f - c * tri;  // strictly lower macro-tiles, row-major: (I, J), J < I, I = 1 .. m - 1
                I = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)t)) * 0.5f);
                while (I * (I - 1) / 2 > t) --I;
                while ((I + 1) * I / 2 <= t) ++I;
                J = t - I * (I - 1) / 2;
            } else {
                c = f / (m * m);
                const int t = f - c * m * m;
                J = t / m;
                I = t - J * m;
            }
            const int8_t* pA = a.X + (int64_t)c * a.strideX + (int64_t)I * 256 * a.ldx;
            const int8_t* pB = a.X + (int64_t)c * a.strideX + (int64_t)J * 256 * a.ldx;
            int32_t* cA = a.C + (int64_t)c * a.strideC + (int64_t)I * 256 + (int64_t)J * 256 * a.ldc;
            if (wave < 4) sym_job<KB, -1, false>(L, smem, pA, pB, cA, a.ldc, nk, soff, coff);
            else sym_job<KB, -1, true>(L, smem, pA, pB, cA, a.ldc, nk, soff, coff);
        } else if (job >= NF + ND) {
            // quarter tile of the ragged border: unit row 2 m x unit column u (u <= 2 m), or -- full squares only --
            // unit row u - (2 m + 1) x unit column 2 m
            const int q = job - NF - ND;
            const int c = q / nqc, u = q - c * nqc;
            const int ur = u <= 2 * m ? 2 * m : u - (2 * m + 1), uc = u <= 2 * m ? u : 2 * m;
            const int8_t* pA = a.X + (int64_t)c * a.strideX + (int64_t)ur * 128 * a.ldx;
            const int8_t* pB = a.X + (int64_t)c * a.strideX + (int64_t)uc * 128 * a.ldx;
            int32_t* cQ = a.C + (int64_t)c * a.strideC + (int64_t)ur * 128 + (int64_t)uc * 128 * a.ldc;
            if (wave < 4) sym_job<KB, -2, false>(L, smem, pA, pB, cQ, a.ldc, nk, soffq, coff);
            else sym_job<KB, -2, true>(L, smem, pA, pB, cQ, a.ldc, nk, soffq, coff);
        } else {
            const int d1 = 2 * (job - NF), d2 = d1 + 1;
            const int c1 = d1 / m, I1 = d1 - c1 * m;
            const int8_t* pA = a.X + (int64_t)c1 * a.strideX + (int64_t)I1 * 256 * a.ldx;
            int32_t* cA = a.C + (int64_t)c1 * a.strideC + (int64_t)I1 * 256 * (1 + a.ldc);
            const int8_t* pB = pA;
            int32_t* cB = nullptr;
            if (d2 < ndt) {
                const int c2 = d2 / m, I2 = d2 - c2 * m;
                pB = a.X + (int64_t)c2 * a.strideX + (int64_t)I2 * 256 * a.ldx;
                cB = a.C + (int64_t)c2 * a.strideC + (int64_t)I2 * 256 * (1 + a.ldc);
            }
            // waves 0-3: the tile of panel A, waves 4-7 (half a stage behind): that of panel B
            switch (wave) {
                case 0: sym_job<KB, 0, false>(L, smem, pA, pB, cA, a.ldc, nk, soff, coff); break;
                case 1: sym_job<KB, 1, false>(L, smem, pA, pB, cA, a.ldc, nk, soff, coff); break;
                case 2: sym_job<KB, 2, false>(L, smem, pA, pB, cA, a.ldc, nk, soff, coff); break;
                case 3: sym_job<KB, 3, false>(L, smem, pA, pB, cA, a.ldc, nk, soff, coff); break;
                case 4: sym_job<KB, 0, true>(L, smem, pA, pB, cB, a.ldc, nk, soff, coff); break;
                case 5: sym_job<KB, 1, true>(L, smem, pA, pB, cB, a.ldc, nk, soff, coff); break;
                case 6: sym_job<KB, 2, true>(L, smem, pA, pB, cB, a.ldc, nk, soff, coff); break;
                default: sym_job<KB, 3, true>(L, smem, pA, pB, cB, a.ldc, nk, soff, coff); break;
            }
        }
    }
}

bool gemm_sym_set_device_attributes() {
    bool ok = true;
    ok &= hipSuccess == hipFuncSetAttribute(reinterpret_cast<const void*>(&i8_symsquare_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    return ok;
}

// Does the persistent launch pay for an n x n problem with T channels?  Its big jobs (lower-triangle macro-tiles of
// the whole 256-row part of the matrix, the diagonal ones two per job) come in rounds of one per CU, and a job is
// 35 - 40 us of work at N = 4096: the launch pays when the rounds are well filled (N = 4096, 2 channels: 256 jobs on 256
// CUs; N = 8192: 1024; N = 4104 -> ld 4224: the same 256 jobs plus 66 quarter tiles of the last 128 rows, a short
// second round).  A sparsely filled last round of big jobs or a small problem keeps the 128 x 128 tiles of
// kernels_gemm.hip, which spread over more CUs.
bool i8_symsquare_pays(int64_t n, int T, int num_cus) {
    if (num_cus < 1) return false;
    const int64_t m = ((n + 127) / 128 * 128) / 256;
    const int64_t jobs = (int64_t)T * m * (m - 1) / 2 + ((int64_t)T * m + 1) / 2;
    const int64_t rounds = (jobs + num_cus - 1) / num_cus;
    return 4 * jobs >= 3 * (int64_t)num_cus && 100 * jobs >= 85 * rounds * num_cus;
}

// true if the persistent kernel serves this launch (ld a multiple of 256, aligned operands)
bool launch_i8_symsquare(hipStream_t s, int64_t n, int64_t k, const int8_t* X, int64_t ldx, int32_t* C, int64_t ldc, int batch,
                         int64_t strideX, int64_t strideC, const uint32_t* nonsym_flag, int num_cus, int variant) {
    const bool forced = variant == 64;  // tests / measurements: the persistent launch at any size
    if (num_cus < 1 || !(forced || i8_symsquare_pays(n, batch, num_cus)) || n < 256 || (n % 128) != 0 || (k % 128) != 0 || (ldx % 16) != 0 || (strideX % 16) != 0 ||
        (reinterpret_cast<uintptr_t>(X) % 16) != 0 || n > 65536 || ldx > 65536)
        return false;
    SymSquareArgs a;
    a.X = X;
    a.C = C;
    a.ldx = ldx;
    a.ldc = ldc;
    a.strideX = strideX;
    a.strideC = strideC;
    a.k = (int)k;
    a.m = (int)(n / 256);
    a.rem = (n % 256) != 0 ? 1 : 0;
    a.T = batch;
    a.nonsym_flag = nonsym_flag;
    const int64_t jobs = (int64_t)batch * (a.m * a.m + (a.rem ? 4 * a.m + 1 : 0));  // upper bound (the full squares)
    const unsigned grid = (unsigned)std::min<int64_t>(jobs, num_cus);
    i8_symsquare_kernel<64><<<grid, SYM_NT, 128 * 1024, s>>>(a);
    return true;
}

}  // namespace sdpsr
