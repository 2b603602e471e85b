// Canonical relabel for the many-classes regime of refine! / Partition(M) (src/partitions.jl:24-35,44-66) by
// GROUPING, not sorting.  With ~n^2/2 distinct signatures (problems without symmetry: BASELINE configs[1]) a hash
// table is far larger than any cache, and the 64-bit radix sort of kernels_refine_sort.hip (hipCUB; kept behind
// refine_path = 2) moves ~250 bytes per entry in eight passes.  The canonical numbering only needs, for every entry,
// the FIRST index of its class in the column-major scan; equal signatures only have to meet, not to be ordered:
//
//   count    histogram of the entries over NBT ~ len / 2048 (<= 2^15) hash buckets (LDS tables, added to a global one);
//            first[e] = UNSET (ZERO for signature 0)
//   starts   exclusive sum: where every bucket starts; cursors of the two scatter levels
//   scatter  (signature, index) to its bucket in TWO levels of <= 128 x <= 256 buckets.  A workgroup stages its 8192
//            entries in LDS ordered by bucket, reserves one range per bucket with one global atomic, and writes whole
//            runs (64 - 128 entries: full lines).  (One level of 2048 buckets with one store per entry ran at the ~65 G
//            transactions/s of scattered 8-byte writes, 0.48 ms at 16.7 M entries; measured, replaced.)
//   resolve  one workgroup per bucket: LDS table (signature -> smallest index), in as many sub-passes over the
//            bucket's entries (by further hash bits) as the distinct signatures need; every entry that is not the first
//            of its class stores first[index] = that first index -- the one random scatter of the method
//   rank     in index order: firsts counted per block, block offsets by one scan, labels of the firsts = their rank + 1;
//            then every other entry copies the label of its first (first[e] < e: the one random gather)
//
// Signature 0 is the structurally-zero class (label 0, not counted).  Results are the canonical numbering whatever
// order the atomics put entries in (only minima of indices are taken).
#include "sdpsr_internal.h"
#include "sdpsr_hash.h"

namespace sdpsr {

constexpr int BK_MAX_LGT = 15;        // <= 32768 buckets in all (LDS histogram of 128 KiB, dynamic)
constexpr int BK_TS = 4096;           // slots of the resolver's LDS table (48 KiB: three workgroups per CU)
constexpr int BK_RTHREADS = 256;      // resolver workgroup
constexpr int BK_RB = 4096;           // entries per block of the rank passes (256 threads x 16)
constexpr int SC_THREADS = 1024;      // scatter workgroup
constexpr int SC_PER = 8;             // entries per thread
constexpr int SC_CH = SC_THREADS * SC_PER;  // 8192 entries staged per workgroup
constexpr int SC_MAXB = 256;          // buckets of one scatter level
constexpr uint32_t BK_UNSET = 0xFFFFFFFFu, BK_ZERO = 0xFFFFFFFEu;

// hash bits: [63 .. 64 - lgt] bucket (level 1: the top lg1 of them), [43 .. 32] table slot, [51 .. 44] sub-pass
__device__ __forceinline__ uint64_t bk_hash(uint64_t sg) { return sdpsr_fmix64(sg ^ 0x6A09E667F3BCC909ULL); }

__global__ void __launch_bounds__(1024)
bk_count_kernel(int64_t len, const uint64_t* __restrict__ sig, int lgt, uint32_t* __restrict__ hist, uint32_t* __restrict__ first) {
    extern __shared__ uint32_t h[];  // 1 << lgt words
    const int NBT = 1 << lgt;
    for (int i = threadIdx.x; i < NBT; i += 1024) h[i] = 0u;
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * 1024;
    for (int64_t e = (int64_t)blockIdx.x * 1024 + threadIdx.x; e < len; e += stride) {
        const uint64_t sg = sig[e];
        first[e] = sg ? BK_UNSET : BK_ZERO;
        if (sg) atomicAdd(&h[(uint32_t)(bk_hash(sg) >> (64 - lgt))], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NBT; i += 1024)
        if (h[i]) atomicAdd(&hist[i], h[i]);
}

// exclusive sum of v[0 .. m) in place by ONE workgroup of 1024 threads; total -> *total_out (if not null)
__global__ void __launch_bounds__(1024)
bk_scan_kernel(int64_t m, uint32_t* __restrict__ v, uint32_t* __restrict__ total_out) {
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t carry_s;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0u;
    __syncthreads();
    // tiles of 1024 x 8 consecutive elements: a thread owns 8 consecutive ones
    for (int64_t base = 0; base < m; base += 8192) {
        const int64_t i0 = base + (int64_t)threadIdx.x * 8;
        uint32_t x[8];
        uint32_t s = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            x[q] = (i0 + q < m) ? v[i0 + q] : 0u;
            s += x[q];
        }
        uint32_t incl = s;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(incl, o, 64);
            if (lane >= o) incl += y;
        }
        if (lane == 63) wsum[w] = incl;
        __syncthreads();
        uint32_t woff = 0;
        for (int k = 0; k < w; ++k) woff += wsum[k];
        uint32_t run = carry_s + woff + incl - s;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (i0 + q < m) v[i0 + q] = run;
            run += x[q];
        }
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = run;
        __syncthreads();
    }
    if (threadIdx.x == 0 && total_out) *total_out = carry_s;
}

// cursors of the scatter levels from the bucket starts (start[NBT] = total is written by the scan)
__global__ void bk_cursors_kernel(int lgt, int lg2, const uint32_t* __restrict__ start, uint32_t* __restrict__ cur1, uint32_t* __restrict__ cur2) {
    const int NBT = 1 << lgt;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < NBT; i += gridDim.x * blockDim.x) {
        cur2[i] = start[i];
        if ((i & ((1 << lg2) - 1)) == 0) cur1[i >> lg2] = start[i];
    }
}

// One scatter level.  LEVEL 1: input = the signature array (index = position), bucket = top lg1 hash bits, all of len.
// LEVEL 2: input = level 1's output; workgroup -> (level-1 bucket b1, chunk of it); bucket = the next lg2 hash bits.
// LDS (dynamic): signatures 8 B, indices 4 B, bucket ids 1 B per staged entry, then the per-bucket words.
template <int LEVEL>
__global__ void __launch_bounds__(SC_THREADS)
bk_scatter_kernel(int64_t len, const uint64_t* __restrict__ in_sig, const uint32_t* __restrict__ in_idx, int lg1, int lg2,
                  const uint32_t* __restrict__ start, uint32_t* __restrict__ cur, uint64_t* __restrict__ out_sig, uint32_t* __restrict__ out_idx) {
    extern __shared__ __attribute__((aligned(16))) char sc_smem[];
    uint64_t* s_sig = reinterpret_cast<uint64_t*>(sc_smem);
    uint32_t* s_idx = reinterpret_cast<uint32_t*>(sc_smem + (size_t)SC_CH * 8);
    uint8_t* s_b = reinterpret_cast<uint8_t*>(sc_smem + (size_t)SC_CH * 12);
    uint32_t* s_cnt = reinterpret_cast<uint32_t*>(sc_smem + (size_t)SC_CH * 13);
    uint32_t* s_lbase = s_cnt + SC_MAXB;
    uint32_t* s_gbase = s_lbase + SC_MAXB;
    __shared__ int64_t seg_lo, seg_hi;
    __shared__ int seg_b1;
    const int lgt = lg1 + lg2;
    const int B = LEVEL == 1 ? (1 << lg1) : (1 << lg2);
    // ---- which entries ----
    int64_t lo, hi;  // [lo, hi) of the input array
    int b1 = 0;
    if (LEVEL == 1) {
        lo = (int64_t)blockIdx.x * SC_CH;
        hi = lo + SC_CH < len ? lo + SC_CH : len;
        if (lo >= len) return;
    } else {
        if (threadIdx.x == 0) {
            // chunk number blockIdx.x in the concatenation of the level-1 buckets' chunk lists
            int64_t c = blockIdx.x;
            seg_b1 = -1;
            const int B1 = 1 << lg1;
            for (int k = 0; k < B1; ++k) {
                const int64_t a = start[(int64_t)k << lg2], z = start[(int64_t)(k + 1) << lg2];
                const int64_t nch = (z - a + SC_CH - 1) / SC_CH;
                if (c < nch) {
                    seg_b1 = k;
                    seg_lo = a + c * SC_CH;
                    seg_hi = seg_lo + SC_CH < z ? seg_lo + SC_CH : z;
                    break;
                }
                c -= nch;
            }
        }
        __syncthreads();
        if (seg_b1 < 0) return;  // uniform
        b1 = seg_b1;
        lo = seg_lo;
        hi = seg_hi;
    }
    for (int i = threadIdx.x; i < B; i += SC_THREADS) s_cnt[i] = 0u;
    __syncthreads();
    // ---- load, bucket, place inside the bucket ----
    uint64_t sg[SC_PER];
    uint32_t ix[SC_PER], loc[SC_PER];
    int bk[SC_PER];
#pragma unroll
    for (int q = 0; q < SC_PER; ++q) {
        const int64_t e = lo + q * SC_THREADS + threadIdx.x;
        sg[q] = (e < hi) ? in_sig[e] : 0ull;
        ix[q] = LEVEL == 1 ? (uint32_t)e : ((e < hi) ? in_idx[e] : 0u);
    }
#pragma unroll
    for (int q = 0; q < SC_PER; ++q) {
        bk[q] = -1;
        if (sg[q]) {
            const uint64_t hh = bk_hash(sg[q]);
            bk[q] = LEVEL == 1 ? (int)(hh >> (64 - lg1)) : (int)((hh >> (64 - lgt)) & (uint64_t)(B - 1));
            loc[q] = atomicAdd(&s_cnt[bk[q]], 1u);
        }
    }
    __syncthreads();
    // ---- local starts (one wave scans the <= 256 counts) and one global reservation per bucket ----
    if (threadIdx.x < 64) {
        uint32_t run = 0;
        for (int b0 = 0; b0 < B; b0 += 64) {
            const int b = b0 + threadIdx.x;
            const uint32_t cn = b < B ? s_cnt[b] : 0u;
            uint32_t incl = cn;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t y = __shfl_up(incl, o, 64);
                if ((int)threadIdx.x >= o) incl += y;
            }
            if (b < B) {
                s_lbase[b] = run + incl - cn;
                s_gbase[b] = cn ? atomicAdd(&cur[LEVEL == 1 ? b : ((b1 << lg2) + b)], cn) : 0u;
            }
            run += __shfl(incl, 63, 64);
        }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < SC_PER; ++q)
        if (bk[q] >= 0) {
            const uint32_t p = s_lbase[bk[q]] + loc[q];
            s_sig[p] = sg[q];
            s_idx[p] = ix[q];
            s_b[p] = (uint8_t)bk[q];
        }
    __syncthreads();
    // ---- runs out: consecutive staged entries of a bucket go to consecutive addresses ----
    const uint32_t staged = s_lbase[B - 1] + s_cnt[B - 1];
    for (uint32_t p = threadIdx.x; p < staged; p += SC_THREADS) {
        const int b = s_b[p];
        const uint32_t g = s_gbase[b] + (p - s_lbase[b]);
        out_sig[g] = s_sig[p];
        out_idx[g] = s_idx[p];
    }
}

// find-or-insert in the resolver's LDS table; returns the slot, or -1 when the probe sequence says the table is too full
__device__ __forceinline__ int bk_table_insert(unsigned long long* t_sig, uint32_t* t_cnt, uint32_t* t_ovf, uint64_t sg, uint64_t hh) {
    uint32_t sl = (uint32_t)(hh >> 32) & (BK_TS - 1);
    for (int probes = 0; probes < BK_TS / 2; ++probes) {
        const unsigned long long cur = t_sig[sl];
        if (cur == sg) return (int)sl;
        if (cur == 0ull) {
            const unsigned long long old = atomicCAS(&t_sig[sl], 0ull, (unsigned long long)sg);
            if (old == 0ull) {
                if (atomicAdd(t_cnt, 1u) + 1 > (BK_TS / 4) * 3) *t_ovf = 1u;
                return (int)sl;
            }
            if (old == sg) return (int)sl;
        }
        sl = (sl + 1) & (BK_TS - 1);
    }
    *t_ovf = 1u;  // (a table this full is being abandoned anyway)
    return -1;
}

// One workgroup per bucket: first[idx] = smallest index of the entry's class, for the entries that are not it.
constexpr int BK_RPER = 12;  // entries per thread of the one-sweep form (<= 3072 per bucket: the table's 75 %)
__global__ void __launch_bounds__(BK_RTHREADS)
bk_resolve_kernel(int NB, const uint32_t* __restrict__ bstart, const uint64_t* __restrict__ bsig, const uint32_t* __restrict__ bidx,
                  uint32_t* __restrict__ first, uint32_t* __restrict__ fail) {
    __shared__ unsigned long long t_sig[BK_TS];
    __shared__ uint32_t t_min[BK_TS];
    __shared__ uint32_t t_cnt, t_ovf;
    for (int bk = blockIdx.x; bk < NB; bk += gridDim.x) {
        const uint32_t s0 = bstart[bk], n = bstart[bk + 1] - s0;
        if (n == 0) continue;  // uniform
        __syncthreads();  // the previous bucket's readers are done with the table
        for (int i = threadIdx.x; i < BK_TS; i += BK_RTHREADS) {
            t_sig[i] = 0ull;
            t_min[i] = 0xFFFFFFFFu;
        }
        if (threadIdx.x == 0) {
            t_cnt = 0u;
            t_ovf = 0u;
        }
        bool done = false;
        if (n <= BK_RPER * BK_RTHREADS) {
            // ---- the usual bucket: every entry in registers (all loads in flight at once), one sweep ----
            uint64_t sg[BK_RPER];
            uint32_t ix[BK_RPER];
            int sl[BK_RPER];
#pragma unroll
            for (int q = 0; q < BK_RPER; ++q) {
                const uint32_t i = q * BK_RTHREADS + threadIdx.x;
                sg[q] = i < n ? bsig[s0 + i] : 0ull;
                ix[q] = i < n ? bidx[s0 + i] : 0u;
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < BK_RPER; ++q) {
                sl[q] = -1;
                if (sg[q]) {
                    sl[q] = bk_table_insert(t_sig, &t_cnt, &t_ovf, sg[q], bk_hash(sg[q]));
                    if (sl[q] >= 0 && t_min[sl[q]] > ix[q]) atomicMin(&t_min[sl[q]], ix[q]);
                }
            }
            __syncthreads();
            if (!t_ovf) {  // uniform
#pragma unroll
                for (int q = 0; q < BK_RPER; ++q)
                    if (sl[q] >= 0) {
                        const uint32_t m = t_min[sl[q]];
                        if (m != ix[q]) first[ix[q]] = m;
                    }
                done = true;
            }
        }
        if (done) continue;
        // ---- large or crowded bucket: sub-passes over its entries by further hash bits, two sweeps each.  The number of
        // sub-passes follows the DISTINCT signatures, not the entries (a bucket of 260 000 entries of one class -- a
        // refinement that ends with few classes on a ctx whose previous one ended with many -- is one sub-pass): start
        // with one, double whenever a sub-pass overfills the table ----
        uint32_t npass = 1;
        for (uint32_t sp = 0; sp < npass;) {
            __syncthreads();
            for (int i = threadIdx.x; i < BK_TS; i += BK_RTHREADS) {
                t_sig[i] = 0ull;
                t_min[i] = 0xFFFFFFFFu;
            }
            if (threadIdx.x == 0) {
                t_cnt = 0u;
                t_ovf = 0u;
            }
            __syncthreads();
            for (uint32_t i0 = 0; i0 < n; i0 += 8 * BK_RTHREADS) {  // chunks of 2048 entries; an overfull table stops the sweep
                uint64_t sgs[8];
                uint32_t ixs[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {  // all of the chunk's loads in flight before the first probe
                    const uint32_t i = i0 + q * BK_RTHREADS + threadIdx.x;
                    sgs[q] = i < n ? bsig[s0 + i] : 0ull;
                    ixs[q] = i < n ? bidx[s0 + i] : 0u;
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    if (!sgs[q]) continue;
                    const uint64_t hh = bk_hash(sgs[q]);
                    if (((uint32_t)(hh >> 44) & (npass - 1)) != sp) continue;
                    const int sl = bk_table_insert(t_sig, &t_cnt, &t_ovf, sgs[q], hh);
                    if (sl >= 0 && t_min[sl] > ixs[q]) atomicMin(&t_min[sl], ixs[q]);
                }
                __syncthreads();
                if (t_ovf) break;  // uniform
            }
            __syncthreads();
            if (t_ovf) {  // uniform: more distinct signatures in this sub-pass than the table takes: split finer, start over
                if (npass >= 256) {  // 2^8 sub-passes of 3072 distinct signatures each in ONE bucket: the hash bits do not spread
                    if (threadIdx.x == 0) *fail = 1u;  // these signatures; reported (counters[1]), never a silently wrong partition
                    return;
                }
                npass <<= 1;
                sp = 0;
                continue;
            }
            for (uint32_t i0 = 0; i0 < n; i0 += 8 * BK_RTHREADS) {
                uint64_t sgs[8];
                uint32_t ixs[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const uint32_t i = i0 + q * BK_RTHREADS + threadIdx.x;
                    sgs[q] = i < n ? bsig[s0 + i] : 0ull;
                    ixs[q] = i < n ? bidx[s0 + i] : 0u;
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    if (!sgs[q]) continue;
                    const uint64_t hh = bk_hash(sgs[q]);
                    if (((uint32_t)(hh >> 44) & (npass - 1)) != sp) continue;
                    uint32_t sl = (uint32_t)(hh >> 32) & (BK_TS - 1);
                    while (t_sig[sl] != sgs[q]) sl = (sl + 1) & (BK_TS - 1);
                    const uint32_t m = t_min[sl];
                    if (m != ixs[q]) first[ixs[q]] = m;
                }
            }
            ++sp;
        }
    }
}

// Rank passes in index order.  A block covers BK_RB = 4096 consecutive entries as 16 rows of 256: thread t reads entry
// base + 256 q + t of row q (coalesced); ranks inside the block come from wave ballots and a 16 x 4 table of wave counts.
__device__ __forceinline__ int bk_block_flags(int64_t len, int64_t base, const uint32_t* __restrict__ first, uint32_t (&f)[16], int (&lanepre)[16],
                                              int* s_wcnt /* [16][4] -> exclusive prefix in row-major order, [64] = total */) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int64_t e = base + q * 256 + threadIdx.x;
        f[q] = e < len ? first[e] : BK_ZERO;
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const unsigned long long b = __ballot(f[q] == BK_UNSET);
        lanepre[q] = __popcll(b & ((1ull << lane) - 1ull));
        if (lane == 0) s_wcnt[q * 4 + w] = __popcll(b);
    }
    __syncthreads();
    if (threadIdx.x < 64) {  // exclusive prefix of the 64 wave counts (row-major = index order)
        const int v = s_wcnt[threadIdx.x];
        int incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int y = __shfl_up(incl, o, 64);
            if ((int)threadIdx.x >= o) incl += y;
        }
        s_wcnt[threadIdx.x] = incl - v;
        if (threadIdx.x == 63) s_wcnt[64] = incl;
    }
    __syncthreads();
    return s_wcnt[64];
}

__global__ void __launch_bounds__(256)
bk_first_count_kernel(int64_t len, const uint32_t* __restrict__ first, uint32_t* __restrict__ blk_cnt) {
    __shared__ int sh[4];
    const int64_t base = (int64_t)blockIdx.x * BK_RB;
    int cnt = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int64_t e = base + q * 256 + threadIdx.x;
        if (e < len) cnt += (first[e] == BK_UNSET);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) blk_cnt[blockIdx.x] = (uint32_t)(sh[0] + sh[1] + sh[2] + sh[3]);
}

__global__ void __launch_bounds__(256)
bk_label_first_kernel(int64_t len, const uint32_t* __restrict__ first, const uint32_t* __restrict__ blk_off, uint32_t* __restrict__ labels,
                      uint32_t* __restrict__ first_idx, uint32_t first_cap) {
    __shared__ int s_wcnt[65];
    const int w = threadIdx.x >> 6;
    const int64_t base = (int64_t)blockIdx.x * BK_RB;
    uint32_t f[16];
    int lanepre[16];
    bk_block_flags(len, base, first, f, lanepre, s_wcnt);
    const uint32_t off = blk_off[blockIdx.x];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int64_t e = base + q * 256 + threadIdx.x;
        if (e >= len) continue;
        if (f[q] == BK_UNSET) {
            const uint32_t lab = off + (uint32_t)(s_wcnt[q * 4 + w] + lanepre[q]) + 1u;
            labels[e] = lab;
            if (first_idx && lab <= first_cap) first_idx[lab - 1] = (uint32_t)e;
        } else if (f[q] == BK_ZERO) {
            labels[e] = 0u;
        }
    }
}

__global__ void __launch_bounds__(256)
bk_label_rest_kernel(int64_t len, const uint32_t* __restrict__ first, uint32_t* labels, const uint32_t* __restrict__ total, uint32_t* __restrict__ counters) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const uint32_t d = total[0];
        counters[0] = d;
        counters[1] = total[1];  // the resolver's failure word
        counters[2] = d;
    }
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < len; e += stride) {
        const uint32_t f = first[e];
        if (f < BK_ZERO) labels[e] = labels[f];  // f < e: written by bk_label_first_kernel
    }
}

// bucket bits: ~2048 entries per bucket, at most 2^13 buckets; two scatter levels from 2^8 buckets on
static void bk_plan(int64_t len, int* lg1, int* lg2) {
    int lgt = 4;
    while (lgt < BK_MAX_LGT && (int64_t(1) << (lgt + 1)) * 2048 <= len) ++lgt;
    if (lgt <= 7) {
        *lg1 = lgt;
        *lg2 = 0;
    } else {
        *lg1 = lgt / 2;
        *lg2 = lgt - lgt / 2;
    }
}

constexpr size_t SC_LDS_BYTES = (size_t)SC_CH * 13 + 3 * SC_MAXB * 4;

void refine_bucket_set_device_attributes() {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&bk_count_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (1 << BK_MAX_LGT) * 4);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&bk_scatter_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SC_LDS_BYTES);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&bk_scatter_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SC_LDS_BYTES);
}

// workspace: two (signature, index) buffers (len u64 + len u32 each) | first (len u32) | start (NBT + 1) | cursors | blk_cnt | total
size_t refine_bucketed_workspace_bytes(int64_t len) {
    const int64_t nrb = (len + BK_RB - 1) / BK_RB;
    return (size_t)len * 28 + (size_t)((1 << BK_MAX_LGT) + 1) * 4 * 3 + (size_t)(nrb + 1) * 4 + 1024 + 12 * 256;
}

// labels_out: canonical labels; counters[0] = counters[2] = number of classes, counters[1] = 0 (1: a bucket could not be resolved,
// labels_out is then not a partition of the input and the caller must fail); first_idx (may be null):
// first-occurrence index of class l at [l - 1] for l <= first_cap
bool launch_refine_bucketed(hipStream_t s, int64_t len, const uint64_t* sig, uint32_t* labels_out, void* ws, size_t ws_bytes,
                            uint32_t* counters, uint32_t* first_idx, uint32_t first_cap) {
    if (len < 1 || len >= (int64_t(1) << 31) || ws_bytes < refine_bucketed_workspace_bytes(len)) return false;
    int lg1, lg2;
    bk_plan(len, &lg1, &lg2);
    const int lgt = lg1 + lg2;
    const int64_t nbt = int64_t(1) << lgt;
    const int64_t nrb = (len + BK_RB - 1) / BK_RB;
    auto align = [](char* p) { return (char*)(((uintptr_t)p + 255) & ~uintptr_t(255)); };
    char* p = align((char*)ws);
    uint64_t* sigA = (uint64_t*)p;
    p = align(p + (size_t)len * 8);
    uint64_t* sigB = (uint64_t*)p;
    p = align(p + (size_t)len * 8);
    uint32_t* idxA = (uint32_t*)p;
    p = align(p + (size_t)len * 4);
    uint32_t* idxB = (uint32_t*)p;
    p = align(p + (size_t)len * 4);
    uint32_t* first = (uint32_t*)p;
    p = align(p + (size_t)len * 4);
    uint32_t* start = (uint32_t*)p;
    p = align(p + (size_t)(nbt + 1) * 4);
    uint32_t* cur1 = (uint32_t*)p;
    p = align(p + (size_t)(nbt + 1) * 4);
    uint32_t* cur2 = (uint32_t*)p;
    p = align(p + (size_t)(nbt + 1) * 4);
    uint32_t* blk_cnt = (uint32_t*)p;
    p = align(p + (size_t)(nrb + 1) * 4);
    uint32_t* total = (uint32_t*)p;
    if (hipMemsetAsync(start, 0, (size_t)(nbt + 1) * 4, s) != hipSuccess || hipMemsetAsync(total, 0, 8, s) != hipSuccess) return false;
    const int cgrid = (int)std::min<int64_t>((len + 1023) / 1024, 512);
    bk_count_kernel<<<cgrid, 1024, (size_t)nbt * 4, s>>>(len, sig, lgt, start, first);
    bk_scan_kernel<<<1, 1024, 0, s>>>(nbt, start, start + nbt);
    bk_cursors_kernel<<<(unsigned)((nbt + 255) / 256), 256, 0, s>>>(lgt, lg2, start, cur1, cur2);
    const unsigned g1 = (unsigned)((len + SC_CH - 1) / SC_CH);
    bk_scatter_kernel<1><<<g1, SC_THREADS, SC_LDS_BYTES, s>>>(len, sig, nullptr, lg1, lg2, start, cur1, sigA, idxA);
    const uint64_t* bsig = sigA;
    const uint32_t* bidx = idxA;
    if (lg2 > 0) {
        const unsigned g2 = g1 + (1u << lg1);  // every level-1 bucket may end in a partial chunk
        bk_scatter_kernel<2><<<g2, SC_THREADS, SC_LDS_BYTES, s>>>(len, sigA, idxA, lg1, lg2, start, cur2, sigB, idxB);
        bsig = sigB;
        bidx = idxB;
    }
    const int rgrid = (int)std::min<int64_t>(nbt, 256 * 3);
    bk_resolve_kernel<<<rgrid, BK_RTHREADS, 0, s>>>((int)nbt, start, bsig, bidx, first, total + 1);
    bk_first_count_kernel<<<(unsigned)nrb, 256, 0, s>>>(len, first, blk_cnt);
    bk_scan_kernel<<<1, 1024, 0, s>>>(nrb, blk_cnt, total);
    bk_label_first_kernel<<<(unsigned)nrb, 256, 0, s>>>(len, first, blk_cnt, labels_out, first_idx, first_cap);
    int64_t g = (len + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;
    bk_label_rest_kernel<<<(unsigned)g, 256, 0, s>>>(len, first, labels_out, total, counters);
    return true;
}

}  // namespace sdpsr
