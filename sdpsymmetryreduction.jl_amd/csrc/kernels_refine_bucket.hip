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
//
// Round 5 (the b2_* kernels below; the two-level bk_* front end stays for len > 22 M entries):
//   * ONE scatter level: ~len / 5632 (<= 4032) buckets by range reduction of the hash's high word.  The count pass and the
//     scatter pass split the entries the same way (workgroup w owns one span), so the scatter's write offsets are exact
//     sums of the count pass's per-workgroup histograms: no cursor atomics.  A workgroup stages 8192 entries at a time
//     in LDS ordered by bucket and writes 12-byte records (hash, index) in runs.  The record carries the HASH (a
//     bijection of the signature), so the resolver never hashes.
//   * the resolver holds a whole bucket (<= 8192 records) in REGISTERS (512 threads x 16), two workgroups per CU, and
//     walks it in 4-8 sub-passes over a 4096-slot LDS table (sub-pass = hash bits 13.., slot = hash bits 0..11, bucket =
//     high word: disjoint); a sub-pass first compacts its records into an LDS list, then the threads walk the list
//     densely with their compare-and-swaps in flight together.  Buckets beyond the register file (skewed inputs) go to
//     a list that a second launch resolves with 16 workgroups per bucket, one sub-pass each; what that cannot take is
//     reported and the host repeats the refinement through the radix sort.
//   * ranks from an L2-RESIDENT structure instead of a gather from the label array: per 64 entries one 16-byte record
//     {bit per entry "is the first of its class", number of firsts before the word}; label(e) = rank(first(e)) + 1 is
//     one small gather for the entries that are not first and pure ballot arithmetic for those that are -- the 128-byte
//     line per 4-byte label of the old bk_label_rest pass (1.2 GB at N = 4096) is gone.
// N = 4096, len / 2 classes: 0.72 -> 0.48 ms, 172 -> 99 bytes per entry (PMC); see DESIGN.md section 5.
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

// hash bits: [63 .. 64 - lgt] bucket (level 1: the top lg1 of them), [43 .. 32] table slot, [31 .. 24] sub-pass (disjoint: ADVICE r4)
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
                    if (((uint32_t)(hh >> 24) & (npass - 1)) != sp) continue;
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
                    if (((uint32_t)(hh >> 24) & (npass - 1)) != sp) continue;
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

// ---------------------------------------------------------------------------
// Round 5 front end: one scatter level, register-resident resolver
// ---------------------------------------------------------------------------
constexpr int B2_THREADS = 1024;
constexpr int B2_CAP = 8192;                     // the largest bucket the register form takes (512 threads x 16 records)
constexpr int B2_MEAN = 5632;                    // mean bucket size aimed at.  Classes of k entries make the bucket sizes vary by
                                                 // sqrt(k * mean): the cap is 3 sigma away at k = 128 (the grouping runs from len / 128
                                                 // classes on), 30 at k = 2; 6656 sent 1 % of the buckets of a 64-entries-per-class
                                                 // input to the second launch, more than it takes
constexpr int B2_MAXNB = 4032;                   // buckets of the one scatter level (three LDS words each in the scatter)
constexpr int B2_TS = 4096;                      // slots of the resolver's LDS table (48 KiB with the minima)
constexpr int B2_CH = 8192;                      // entries staged per scatter workgroup
constexpr int B2_MAXBIG = 64;                    // buckets beyond B2_CAP that the second launch resolves
constexpr int B2_BIGSP = 16;                     // its sub-passes = workgroups per such bucket
constexpr int64_t B2_MAXLEN = (int64_t)B2_MAXNB * B2_MEAN;

__device__ __forceinline__ uint32_t b2_bucket(uint64_t hh, uint32_t NB) { return (uint32_t)(((hh >> 32) * (uint64_t)NB) >> 32); }
static int b2_nb(int64_t len) {
    int64_t nb = (len + B2_MEAN - 1) / B2_MEAN;
    if (nb < 2) nb = 2;  // two at least: the resolver's "empty" marker is a hash value of the OTHER end of the range
    return (int)nb;
}

// Work split of the count and scatter kernels: workgroup w owns the `span` consecutive entries from w * span (a multiple
// of B2_CH; ~one workgroup per CU), so the count pass can hand the scatter pass its exact write offsets:
// cnt[w][b] -> off[w][b] = start[b] + sum of cnt[w'][b] over w' < w.  No cursor atomics (round 5, first form: one
// reservation per workgroup, chunk and bucket = 2.6 M atomics on 40 cache lines of cursors, 240 us for the pass),
// and the records of a bucket end up in index order of their chunks.
__global__ void __launch_bounds__(1024)
b2_count_kernel(int64_t len, const uint64_t* __restrict__ sig, uint32_t NB, int64_t span, uint32_t* __restrict__ hist,
                uint32_t* __restrict__ cnt, uint32_t* __restrict__ first) {
    extern __shared__ uint32_t h[];  // NB words
    for (uint32_t i = threadIdx.x; i < NB; i += 1024) h[i] = 0u;
    __syncthreads();
    const int64_t lo = (int64_t)blockIdx.x * span;
    const int64_t hi = lo + span < len ? lo + span : len;
    for (int64_t e0 = lo + (int64_t)threadIdx.x * 4; e0 < hi; e0 += 4096) {
        // four consecutive entries per thread and trip: 32 bytes in, 16 bytes out
        uint64_t sg[4];
        uint32_t fo[4];
        if (e0 + 3 < hi) {
            typedef unsigned long long bk_u64x2 __attribute__((ext_vector_type(2)));
            const bk_u64x2 a = __builtin_nontemporal_load(reinterpret_cast<const bk_u64x2*>(sig + e0));
            const bk_u64x2 b = __builtin_nontemporal_load(reinterpret_cast<const bk_u64x2*>(sig + e0 + 2));
            sg[0] = a.x; sg[1] = a.y; sg[2] = b.x; sg[3] = b.y;
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) sg[q] = e0 + q < hi ? sig[e0 + q] : 0ull;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            fo[q] = sg[q] ? BK_UNSET : BK_ZERO;
            if (sg[q]) atomicAdd(&h[b2_bucket(bk_hash(sg[q]), NB)], 1u);
        }
        if (e0 + 3 < hi) *reinterpret_cast<uint4*>(first + e0) = make_uint4(fo[0], fo[1], fo[2], fo[3]);
        else
            for (int q = 0; q < 4 && e0 + q < hi; ++q) first[e0 + q] = fo[q];
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < NB; i += 1024) {
        cnt[(size_t)blockIdx.x * NB + i] = h[i];
        if (h[i]) atomicAdd(&hist[i], h[i]);
    }
}

// cnt[w][b] -> where workgroup w writes its first record of bucket b.  One wave per bucket walks the G workgroups.
__global__ void __launch_bounds__(256)
b2_offsets_kernel(uint32_t NB, uint32_t G, const uint32_t* __restrict__ start, uint32_t* __restrict__ cnt) {
    const uint32_t b = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (b >= NB) return;
    uint32_t run = start[b];
    for (uint32_t w0 = 0; w0 < G; w0 += 64) {
        const uint32_t w = w0 + lane;
        const uint32_t v = w < G ? cnt[(size_t)w * NB + b] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t y = __shfl_up(incl, o, 64);
            if (lane >= o) incl += y;
        }
        if (w < G) cnt[(size_t)w * NB + b] = run + incl - v;
        run += __shfl(incl, 63, 64);
    }
}

// A workgroup stages B2_CH consecutive entries at a time in LDS, ordered by bucket, and writes each bucket's run behind
// its previous one.  Records are three dwords (hash low, hash high, index), written dword by dword: a run of k entries
// is 12 k contiguous bytes.  LDS (dynamic): records 12 B, bucket ids 2 B per staged entry, three words per bucket.
__global__ void __launch_bounds__(B2_THREADS)
b2_scatter_kernel(int64_t len, const uint64_t* __restrict__ sig, uint32_t NB, int64_t span, const uint32_t* __restrict__ off,
                  uint32_t* __restrict__ rec) {
    extern __shared__ __attribute__((aligned(16))) char b2_smem[];
    uint32_t* s_rec = reinterpret_cast<uint32_t*>(b2_smem);
    uint16_t* s_b = reinterpret_cast<uint16_t*>(b2_smem + (size_t)B2_CH * 12);
    uint32_t* s_cnt = reinterpret_cast<uint32_t*>(b2_smem + (size_t)B2_CH * 14);
    uint32_t* s_lbase = s_cnt + NB;
    uint32_t* s_gbase = s_lbase + NB;
    __shared__ uint32_t s_wsum[B2_THREADS / 64];
    const int64_t wlo = (int64_t)blockIdx.x * span;
    if (wlo >= len) return;
    const int64_t whi = wlo + span < len ? wlo + span : len;
    constexpr int PER = B2_CH / B2_THREADS;  // 8
    for (uint32_t i = threadIdx.x; i < NB; i += B2_THREADS) {
        s_gbase[i] = off[(size_t)blockIdx.x * NB + i];
        s_cnt[i] = 0u;
    }
    uint64_t nxt[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {  // (unconditional loads, index clamped: see refine_insert_mid_kernel)
        const int64_t e = wlo + q * B2_THREADS + threadIdx.x;
        const uint64_t v = __builtin_nontemporal_load(&sig[e < whi ? e : whi - 1]);
        nxt[q] = (e < whi) ? v : 0ull;
    }
    __syncthreads();
    for (int64_t lo = wlo; lo < whi; lo += B2_CH) {
        uint64_t hh[PER];
        uint32_t loc[PER];
        int bk[PER];
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            hh[q] = nxt[q];
            bk[q] = -1;
            if (hh[q]) {
                hh[q] = bk_hash(hh[q]);
                bk[q] = (int)b2_bucket(hh[q], NB);
                loc[q] = atomicAdd(&s_cnt[bk[q]], 1u);
            }
        }
        __syncthreads();
        // ---- exclusive sum of the <= 4096 counts (a thread owns four) ----
        {
            const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
            uint32_t c[4], sum = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t b = 4 * threadIdx.x + k;
                c[k] = b < NB ? s_cnt[b] : 0u;
                sum += c[k];
            }
            uint32_t incl = sum;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t y = __shfl_up(incl, o, 64);
                if (lane >= o) incl += y;
            }
            if (lane == 63) s_wsum[w] = incl;
            __syncthreads();
            uint32_t ex = incl - sum;
            for (int k = 0; k < w; ++k) ex += s_wsum[k];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t b = 4 * threadIdx.x + k;
                if (b < NB) s_lbase[b] = ex;
                ex += c[k];
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < PER; ++q)
            if (bk[q] >= 0) {
                const uint32_t p = s_lbase[bk[q]] + loc[q];
                s_rec[3 * p] = (uint32_t)hh[q];
                s_rec[3 * p + 1] = (uint32_t)(hh[q] >> 32);
                s_rec[3 * p + 2] = (uint32_t)(lo + q * B2_THREADS + threadIdx.x);
                s_b[p] = (uint16_t)bk[q];
            }
        // the next chunk's signatures travel while this one is written out
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int64_t e = lo + B2_CH + q * B2_THREADS + threadIdx.x;
            const uint64_t v = __builtin_nontemporal_load(&sig[e < whi ? e : whi - 1]);
            nxt[q] = (e < whi) ? v : 0ull;
        }
        __syncthreads();
        const uint32_t staged = s_lbase[NB - 1] + s_cnt[NB - 1];
        for (uint32_t i = threadIdx.x; i < 3 * staged; i += B2_THREADS) {
            const uint32_t p = i / 3u, q = i - 3u * p;
            const uint32_t b = s_b[p];
            const uint32_t g = s_gbase[b] + (p - s_lbase[b]);
            rec[(size_t)g * 3 + q] = s_rec[i];
        }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < NB; i += B2_THREADS) {
            s_gbase[i] += s_cnt[i];
            s_cnt[i] = 0u;
        }
        __syncthreads();
    }
}

// find-or-insert of a hash in the resolver's table from slot `sl` on (empty marker: a value no entry of this bucket can have)
__device__ __forceinline__ int b2_insert_from(unsigned long long* t_sig, uint32_t* t_ovf, uint32_t sl, uint64_t hh, unsigned long long empty) {
    for (int probes = 0; probes < 512; ++probes) {
        const unsigned long long old = atomicCAS(&t_sig[sl], empty, (unsigned long long)hh);
        if (old == empty || old == hh) return (int)sl;
        sl = (sl + 1) & (B2_TS - 1);
    }
    *t_ovf = 1u;
    return -1;
}
__device__ __forceinline__ int b2_insert(unsigned long long* t_sig, uint32_t* t_ovf, uint64_t hh, unsigned long long empty) {
    return b2_insert_from(t_sig, t_ovf, (uint32_t)hh & (B2_TS - 1), hh, empty);
}

// big[0] = number of buckets handed to the second launch, big[1 ..] their ids
__device__ __forceinline__ void b2_defer(uint32_t* big, uint32_t* fail, int bk) {
    const uint32_t k = atomicAdd(&big[0], 1u);
    if (k < B2_MAXBIG) big[1 + k] = (uint32_t)bk;
    else *fail = 1u;
}

// One workgroup per bucket: first[idx] = smallest index of the entry's class, for the entries that are not it.
// The bucket sits in REGISTERS: 512 threads x 32 records (three words each).  One workgroup per CU is all the LDS
// allows anyway (table + list = 144 KiB), and eight waves have 256 registers each -- with 1024 threads x 16 records
// the 48 record words spilled, and fetching the high word and the index per sub-pass instead cost two exposed L2 round
// trips per sub-pass.  A sub-pass (hash bits 13 ..) first COMPACTS its records into an LDS list, then the threads walk
// the list densely, eight records each: eight compare-and-swaps in flight, the minima as atomics without a return value,
// one barrier, eight reads.  (First form of round 5: every thread filtered its records per sub-pass -- a quarter of the
// lanes alive in each of 64 dependent LDS round trips per wave: 54 us per bucket.)  The slots a sub-pass used are reset
// by their users.
constexpr int B2_RTHREADS = 512;
constexpr int B2_REPT = B2_CAP / B2_RTHREADS;      // 32 records per thread
constexpr int B2_LIST = 2048;                      // records of one sub-pass (its list in LDS)
constexpr int B2_LPT = B2_LIST / B2_RTHREADS;      // 8 list records per thread
constexpr int B2_SUBMEAN = 1792;                   // sub-passes are chosen for at most this many records on average
constexpr size_t B2_RES_LDS = (size_t)B2_TS * 12 + (size_t)B2_LIST * 12;
#ifdef LK_TIMING
bool dbg_on();  // ctx.cpp (SDPSR_DEBUG)
__device__ long long* b2_dbg = nullptr;  // development aid: wall-clock stamps (100 MHz) of the first two buckets of workgroups 0 and 100
#define B2_STAMP(i)                                                                                                        \
    do {                                                                                                                   \
        if (b2_dbg && threadIdx.x == 0 && b2_it < 2 && (blockIdx.x == 0 || blockIdx.x == 100))                              \
            b2_dbg[((blockIdx.x ? 1 : 0) * 2 + b2_it) * 32 + (i)] = wall_clock64();                                          \
    } while (0)
#else
#define B2_STAMP(i)
#endif
__global__ void __launch_bounds__(B2_RTHREADS) __attribute__((amdgpu_waves_per_eu(4, 4)))  // two workgroups per CU
b2_resolve_kernel(int NB, const uint32_t* __restrict__ bstart, const uint32_t* __restrict__ rec, uint32_t* __restrict__ first,
                  uint32_t* __restrict__ big, uint32_t* __restrict__ fail) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long b2_tab[];  // B2_TS hashes, B2_TS minima, the list
    unsigned long long* t_sig = b2_tab;
    uint32_t* t_min = reinterpret_cast<uint32_t*>(b2_tab + B2_TS);
    uint32_t* l_lo = t_min + B2_TS;
    uint32_t* l_hi = l_lo + B2_LIST;
    uint32_t* l_ix = l_hi + B2_LIST;
    __shared__ uint32_t t_ovf;
    __shared__ uint32_t s_wsum[B2_RTHREADS / 64 + 1];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    auto clear_table = [&](unsigned long long empty) {
        for (int i = threadIdx.x; i < B2_TS; i += B2_RTHREADS) {
            t_sig[i] = empty;
            t_min[i] = 0xFFFFFFFFu;
        }
        if (threadIdx.x == 0) t_ovf = 0u;
    };
    clear_table(blockIdx.x == 0 ? ~0ull : 0ull);
    __syncthreads();
    int b2_it = -1;
    (void)b2_it;
    for (int bk = blockIdx.x; bk < NB; bk += gridDim.x) {
        ++b2_it;
        B2_STAMP(0);
        const uint32_t s0 = bstart[bk], ncur = bstart[bk + 1] - s0;
        const unsigned long long empty = bk == 0 ? ~0ull : 0ull;  // hash 0 lives in bucket 0, hash ~0 in the last one
        if (ncur == 0) continue;  // uniform
        if (ncur > (uint32_t)B2_CAP) {
            if (threadIdx.x == 0) b2_defer(big, fail, bk);
            continue;
        }
        // all of the bucket's loads in flight together; slots past ncur read the records behind the bucket -- the
        // workspace has B2_CAP records of slack -- and are never used
        uint32_t hl[B2_REPT], hh[B2_REPT], ix[B2_REPT];
        {
            const uint32_t* base = rec + (size_t)s0 * 3;
            const uint32_t o = threadIdx.x * 3u;
#pragma unroll
            for (int q = 0; q < B2_REPT; ++q) {
                hl[q] = base[o + (uint32_t)q * (3u * B2_RTHREADS)];
                hh[q] = base[o + (uint32_t)q * (3u * B2_RTHREADS) + 1u];
                ix[q] = base[o + (uint32_t)q * (3u * B2_RTHREADS) + 2u];
            }
        }
        uint32_t nsp = 1;
        while (ncur > nsp * (uint32_t)B2_SUBMEAN) nsp <<= 1;  // <= 8 sub-passes; the table at most 44 % full on average
        bool dirty = false;  // the table holds slots of unknown state (a sub-pass was broken off)
        for (uint32_t sp = 0; sp < nsp; ++sp) {
            // ---- compact the sub-pass's records into the list ----
            uint32_t mine = 0;
#pragma unroll
            for (int q = 0; q < B2_REPT; ++q) {
                const uint32_t i = q * B2_RTHREADS + threadIdx.x;
                mine += (i < ncur && ((hl[q] >> 13) & (nsp - 1)) == sp) ? 1u : 0u;
            }
            uint32_t incl = mine;
#pragma unroll
            for (int o2 = 1; o2 < 64; o2 <<= 1) {
                const uint32_t y = __shfl_up(incl, o2, 64);
                if (lane >= o2) incl += y;
            }
            if (lane == 63) s_wsum[wv] = incl;
            __syncthreads();
            B2_STAMP(1 + sp * 6);
            uint32_t pos = incl - mine, total = 0;
            for (int k = 0; k < B2_RTHREADS / 64; ++k) {
                const uint32_t v = s_wsum[k];
                if (k < wv) pos += v;
                total += v;
            }
            // The list takes B2_LIST records.  A sub-pass that has more (classes of many entries make the sub-pass sizes vary by
            // sqrt(entries per class * mean): 2 % of the sub-passes at 64 entries per class) goes through it in BATCHES: all
            // batches insert, then all batches look their minima up (the slot found again by probing) -- the table's limit is
            // the number of DISTINCT hashes of the sub-pass, which such inputs are far from.
            const uint32_t nbatch = (total + B2_LIST - 1) / B2_LIST;  // uniform
            const uint32_t npass = nbatch > 1 ? 2u : 1u;
            bool broke = false;
            for (uint32_t pass = 0; pass < npass && !broke; ++pass)
                for (uint32_t bt = 0; bt < nbatch; ++bt) {
                    const uint32_t lo = bt * (uint32_t)B2_LIST;
                    const uint32_t cntb = total - lo < (uint32_t)B2_LIST ? total - lo : (uint32_t)B2_LIST;
                    uint32_t pp = pos;
#pragma unroll
                    for (int q = 0; q < B2_REPT; ++q) {
                        const uint32_t i = q * B2_RTHREADS + threadIdx.x;
                        if (i < ncur && ((hl[q] >> 13) & (nsp - 1)) == sp) {
                            if (pp - lo < (uint32_t)B2_LIST) {  // lo <= pp < lo + B2_LIST
                                l_lo[pp - lo] = hl[q];
                                l_hi[pp - lo] = hh[q];
                                l_ix[pp - lo] = ix[q];
                            }
                            ++pp;
                        }
                    }
                    __syncthreads();
                    B2_STAMP(2 + sp * 6);
                    // ---- four records per thread ----
                    uint32_t e_lo[B2_LPT], e_hi[B2_LPT], e_ix[B2_LPT];
                    int sl[B2_LPT];
                    unsigned long long old[B2_LPT];
                    bool need[B2_LPT];
#pragma unroll
                    for (int k = 0; k < B2_LPT; ++k) {
                        const uint32_t i = k * B2_RTHREADS + threadIdx.x;
                        const uint32_t ic = i < cntb ? i : 0u;  // (every value defined on every path: nothing lives across the loops)
                        e_lo[k] = l_lo[ic];
                        e_hi[k] = l_hi[ic];
                        e_ix[k] = l_ix[ic];
                        sl[k] = i < cntb ? (int)(e_lo[k] & (B2_TS - 1)) : -1;
                        old[k] = empty;
                        need[k] = false;
                    }
                    if (pass == 0) {
                        // ---- insert.  Two rounds: the threads' first records, a barrier, then the rest.  A plain read comes
                        // before every atomic: a slot that already holds the hash needs no compare-and-swap, a minimum that is
                        // already smaller no atomicMin.  Classes of many entries put dozens of lanes on ONE LDS address, and
                        // same-address LDS atomics retire one lane at a time; after the first round most records find their
                        // class in the table and a minimum below their own index.
                        bool any = false;
                        auto insert_range = [&](const int k0, const int k1) {
#pragma unroll
                            for (int k = k0; k < k1; ++k)
                                if (sl[k] >= 0) old[k] = t_sig[sl[k]];
#pragma unroll
                            for (int k = k0; k < k1; ++k)
                                if (sl[k] >= 0 && old[k] == empty) old[k] = atomicCAS(&t_sig[sl[k]], empty, (unsigned long long)e_lo[k] | ((unsigned long long)e_hi[k] << 32));
                            any = false;
#pragma unroll
                            for (int k = k0; k < k1; ++k) {
                                need[k] = sl[k] >= 0 && old[k] != empty && old[k] != ((unsigned long long)e_lo[k] | ((unsigned long long)e_hi[k] << 32));
                                any = any || need[k];
                            }
                            for (int probes = 0; any; ++probes) {  // occupied by another hash: the next slots, the records' probes together
                                if (probes >= 512) {
                                    t_ovf = 1u;
#pragma unroll
                                    for (int k = k0; k < k1; ++k)
                                        if (need[k]) sl[k] = -1;
                                    break;
                                }
#pragma unroll
                                for (int k = k0; k < k1; ++k)
                                    if (need[k]) {
                                        sl[k] = (sl[k] + 1) & (B2_TS - 1);
                                        old[k] = t_sig[sl[k]];
                                        if (old[k] == empty) old[k] = atomicCAS(&t_sig[sl[k]], empty, (unsigned long long)e_lo[k] | ((unsigned long long)e_hi[k] << 32));
                                    }
                                any = false;
#pragma unroll
                                for (int k = k0; k < k1; ++k) {
                                    need[k] = need[k] && old[k] != empty && old[k] != ((unsigned long long)e_lo[k] | ((unsigned long long)e_hi[k] << 32));
                                    any = any || need[k];
                                }
                            }
#pragma unroll
                            for (int k = k0; k < k1; ++k)
                                if (sl[k] >= 0 && t_min[sl[k]] > e_ix[k]) atomicMin(&t_min[sl[k]], e_ix[k]);  // (only ever decreases)
                        };
                        insert_range(0, 1);
                        __syncthreads();
                        insert_range(1, B2_LPT);
                        // (measured and not kept: the record that claimed a slot STORES its index, and after one more barrier the
                        // others lower it where they must -- fewer atomics, but the barrier and the extra live registers cost
                        // more: 178 -> 220 us)
                        __syncthreads();
                        B2_STAMP(3 + sp * 6);
                        if (t_ovf) {  // uniform: more distinct hashes than the table takes
                            if (threadIdx.x == 0) b2_defer(big, fail, bk);
                            dirty = true;
                            broke = true;
                            break;
                        }
                    } else {
                        // ---- second pass of a batched sub-pass: every hash is in the table, find its slot again ----
#pragma unroll
                        for (int k = 0; k < B2_LPT; ++k)
                            if (sl[k] >= 0) {
                                const unsigned long long v = (unsigned long long)e_lo[k] | ((unsigned long long)e_hi[k] << 32);
                                while (t_sig[sl[k]] != v) sl[k] = (sl[k] + 1) & (B2_TS - 1);
                            }
                    }
                    if (pass + 1 == npass) {
                        // ---- every record that is not the first of its class learns which one is ----
                        uint32_t mn[B2_LPT];
#pragma unroll
                        for (int k = 0; k < B2_LPT; ++k) mn[k] = sl[k] >= 0 ? t_min[sl[k]] : 0u;
#pragma unroll
                        for (int k = 0; k < B2_LPT; ++k)
                            if (sl[k] >= 0 && mn[k] != e_ix[k]) first[e_ix[k]] = mn[k];
                        B2_STAMP(4 + sp * 6);
                    }
                    __syncthreads();  // the list is free for the next batch; all look-ups of this one are done
                    B2_STAMP(5 + sp * 6);
                    if (nbatch == 1) {
                        // ---- the slots this sub-pass used go back to empty (by their users; the next insert is a barrier away) ----
#pragma unroll
                        for (int k = 0; k < B2_LPT; ++k)
                            if (sl[k] >= 0) {
                                t_sig[sl[k]] = empty;
                                t_min[sl[k]] = 0xFFFFFFFFu;
                            }
                    }
                }
            if (broke) break;
            if (nbatch > 1) {  // uniform: a batched sub-pass leaves the whole table to be cleared
                clear_table(empty);
                __syncthreads();
            }
        }
        B2_STAMP(30);
        if (dirty || empty != 0ull) {  // uniform.  Bucket 0's table was filled with ~0: every later bucket of this workgroup wants 0
            __syncthreads();
            clear_table(0ull);
        }
        __syncthreads();
    }
}

// Buckets beyond the register file, or whose sub-passes overflowed: 16 workgroups per bucket, workgroup = one
// sub-pass (hash bits 13 .. 16), two sweeps over the bucket's records.  A deferred bucket's earlier partial stores
// are overwritten here: every non-first entry of every sub-pass is stored again, first entries were never stored.
__global__ void __launch_bounds__(B2_THREADS)
b2_resolve_big_kernel(const uint32_t* __restrict__ bstart, const uint32_t* __restrict__ rec, uint32_t* __restrict__ first,
                      const uint32_t* __restrict__ big, uint32_t* __restrict__ fail) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long b2_tab[];  // B2_TS hashes, then B2_TS minima
    unsigned long long* t_sig = b2_tab;
    uint32_t* t_min = reinterpret_cast<uint32_t*>(b2_tab + B2_TS);
    __shared__ uint32_t t_ovf;
    const uint32_t nbig = big[0] < (uint32_t)B2_MAXBIG ? big[0] : (uint32_t)B2_MAXBIG;
    const uint32_t bi = blockIdx.x / B2_BIGSP, sp = blockIdx.x % B2_BIGSP;
    if (bi >= nbig) return;
    const uint32_t bk = big[1 + bi];
    const uint32_t s0 = bstart[bk], n = bstart[bk + 1] - s0;
    const unsigned long long empty = bk == 0 ? ~0ull : 0ull;
    for (int i = threadIdx.x; i < B2_TS; i += B2_THREADS) {
        t_sig[i] = empty;
        t_min[i] = 0xFFFFFFFFu;
    }
    if (threadIdx.x == 0) t_ovf = 0u;
    __syncthreads();
    constexpr int PER = 8;
    for (uint32_t i0 = 0; i0 < n; i0 += PER * B2_THREADS) {
        uint32_t hl[PER], hh[PER], ix[PER];
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const uint32_t i = i0 + q * B2_THREADS + threadIdx.x;
            const uint32_t* r = rec + (size_t)(s0 + (i < n ? i : 0u)) * 3;
            hl[q] = r[0];
            hh[q] = r[1];
            ix[q] = r[2];
        }
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const uint32_t i = i0 + q * B2_THREADS + threadIdx.x;
            if (i < n && ((hl[q] >> 13) & (B2_BIGSP - 1)) == sp) {
                const int sl = b2_insert(t_sig, &t_ovf, (uint64_t)hl[q] | ((uint64_t)hh[q] << 32), empty);
                if (sl >= 0 && t_min[sl] > ix[q]) atomicMin(&t_min[sl], ix[q]);
            }
        }
        __syncthreads();
        if (t_ovf) {  // uniform.  More than ~6000 distinct signatures in one of 64 sub-passes of one bucket: reported,
            if (threadIdx.x == 0) *fail = 1u;  // the host repeats the refinement through the radix sort
            return;
        }
    }
    for (uint32_t i0 = 0; i0 < n; i0 += PER * B2_THREADS) {
        uint32_t hl[PER], hh[PER], ix[PER];
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const uint32_t i = i0 + q * B2_THREADS + threadIdx.x;
            const uint32_t* r = rec + (size_t)(s0 + (i < n ? i : 0u)) * 3;
            hl[q] = r[0];
            hh[q] = r[1];
            ix[q] = r[2];
        }
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const uint32_t i = i0 + q * B2_THREADS + threadIdx.x;
            if (i < n && ((hl[q] >> 13) & (B2_BIGSP - 1)) == sp) {
                const unsigned long long v = (uint64_t)hl[q] | ((uint64_t)hh[q] << 32);
                uint32_t sl = hl[q] & (B2_TS - 1);
                while (t_sig[sl] != v) sl = (sl + 1) & (B2_TS - 1);
                const uint32_t m = t_min[sl];
                if (m != ix[q]) first[ix[q]] = m;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Back end (both front ends): ranks from one bit per entry + a prefix count per 64 entries
// ---------------------------------------------------------------------------
// rank record of a word of 64 entries: its first-bits and the number of firsts before it -- ONE 16-byte gather per look-up
// (the address unit retires about one cache line per clock and CU: three separate gathers per entry were 40 us of this pass)
struct __attribute__((aligned(16))) BkRank {
    unsigned long long bits;
    uint32_t before;
    uint32_t pad;
};
__device__ __forceinline__ uint32_t bk_rank_of(uint32_t t, const BkRank* __restrict__ rk) {
    const uint4 r = *reinterpret_cast<const uint4*>(&rk[t >> 6]);
    const unsigned long long b = (unsigned long long)r.x | ((unsigned long long)r.y << 32);
    return r.z + (uint32_t)__popcll(b & ((1ull << (t & 63u)) - 1ull));
}
// A block covers BK_RB = 4096 consecutive entries = 64 words of 64; wave w owns words 16 w .. 16 w + 15.
// bits[word] = ballot(first == UNSET); pre[word] = firsts of the BLOCK before the word; blk_cnt[block] = firsts of the block.
__global__ void __launch_bounds__(256)
bk_bits_kernel(int64_t len, const uint32_t* __restrict__ first, BkRank* __restrict__ rk, uint32_t* __restrict__ blk_cnt) {
    __shared__ uint32_t s_w[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t base = (int64_t)blockIdx.x * BK_RB + (int64_t)w * 1024;
    uint32_t f[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int64_t e = base + k * 64 + lane;
        f[k] = e < len ? __builtin_nontemporal_load(&first[e]) : BK_ZERO;
    }
    unsigned long long mine_bits = 0ull;
    uint32_t mine_pre = 0u, run = 0u;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const unsigned long long b = __ballot(f[k] == BK_UNSET);
        if (lane == k) {
            mine_bits = b;
            mine_pre = run;
        }
        run += (uint32_t)__popcll(b);
    }
    if (lane == 0) s_w[w] = run;
    __syncthreads();
    uint32_t woff = 0;
    for (int k = 0; k < w; ++k) woff += s_w[k];
    const int64_t word = (int64_t)blockIdx.x * 64 + w * 16 + lane;
    if (lane < 16 && word * 64 < len) {
        BkRank r;
        r.bits = mine_bits;
        r.before = woff + mine_pre;  // + the earlier blocks' firsts: bk_rank_finish_kernel
        r.pad = 0u;
        rk[word] = r;
    }
    if (threadIdx.x == 0) blk_cnt[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

// labels in index order: label(e) = 1 + number of first entries before first(e)
// rk[word].before = firsts of the earlier blocks + firsts of the block before the word
__global__ void __launch_bounds__(256)
bk_rank_finish_kernel(int64_t nwords, const uint32_t* __restrict__ blk_off, BkRank* __restrict__ rk) {
    const int64_t w = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (w < nwords) rk[w].before += blk_off[w >> 6];
}
// A wave owns 64 consecutive entries per trip (one word of the bit structure): a FIRST entry's rank is the word's prefix
// count + the ballot bits below its lane -- no lookup at all; only the other entries look their first's rank up (three
// small loads, L2).  Blocks of 4096 entries as in bk_bits_kernel: wave w of a block owns words 16 w .. 16 w + 15.
__global__ void __launch_bounds__(256)
bk_label_kernel(int64_t len, const uint32_t* __restrict__ first, uint32_t* __restrict__ labels, const BkRank* __restrict__ rk,
                const uint32_t* __restrict__ total, uint32_t* __restrict__ counters, uint32_t* __restrict__ host_counters,
                uint32_t* __restrict__ first_idx, uint32_t first_cap, uint32_t host_seq) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const uint32_t d = total[0];
        counters[0] = d;
        counters[1] = total[1];  // the resolvers' failure word
        counters[2] = d;
        if (host_counters) {  // pinned host memory: no 16-byte copy launch behind the pass
            host_counters[0] = d;
            host_counters[1] = total[1];
            host_counters[2] = d;
            __threadfence_system();
            host_counters[3] = host_seq;  // (the stamp: see refine_label_kernel)
        }
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t nrb = (len + BK_RB - 1) / BK_RB;
    for (int64_t blk = blockIdx.x; blk < nrb; blk += gridDim.x) {
        const int64_t base = blk * BK_RB + (int64_t)w * 1024;
        uint32_t f[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int64_t e = base + k * 64 + lane;
            f[k] = e < len ? __builtin_nontemporal_load(&first[e]) : BK_ZERO;
        }
        // ranks of the entries that are not first: their lookups in flight together
        uint32_t o[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            o[k] = 0u;
            if (f[k] < BK_ZERO) o[k] = bk_rank_of(f[k], rk) + 1u;
        }
        uint32_t run = rk[blk * 64 + w * 16].before;  // firsts before this wave's first word (uniform)
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int64_t e = base + k * 64 + lane;
            const unsigned long long b = __ballot(f[k] == BK_UNSET);
            if (f[k] == BK_UNSET) {
                o[k] = run + (uint32_t)__popcll(b & ((1ull << lane) - 1ull)) + 1u;
                if (first_idx && o[k] <= first_cap) first_idx[o[k] - 1] = (uint32_t)e;
            }
            run += (uint32_t)__popcll(b);
            if (e < len) __builtin_nontemporal_store(o[k], &labels[e]);
        }
    }
}

// ---------------------------------------------------------------------------
// Ranks of the classes of a hash-table refinement from the TABLE's side (kernels_partition.hip, more than SMALL_K
// classes): one bit per first index, the rank records of the pass above, label of a slot = rank of its first index + 1.
// Work on the d classes and on len / 64 words -- the entry-level count / scan / rank passes they replace read every
// entry's slot and gathered its minimum twice (140 us at 16.7 M entries).
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
rs_mark_kernel(int64_t cap, const RefSlot* __restrict__ tab, uint32_t* __restrict__ bitmap,
               const uint32_t* __restrict__ counters, uint32_t small_k) {
    if (counters[0] <= small_k || counters[1]) return;  // ranked by the one-workgroup kernel / a pass the host repeats
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t sl = (int64_t)blockIdx.x * 256 + threadIdx.x; sl < cap; sl += stride)
        if (tab[sl].sig) {
            const uint32_t m = tab[sl].min;
            atomicOr(&bitmap[m >> 5], 1u << (m & 31u));
        }
}
// one wave per block of 64 words
__global__ void __launch_bounds__(64)
rs_words_kernel(int64_t nwords, const unsigned long long* __restrict__ bitmap, BkRank* __restrict__ rk, uint32_t* __restrict__ blk_cnt,
                const uint32_t* __restrict__ counters, uint32_t small_k) {
    if (counters[0] <= small_k || counters[1]) return;
    const int64_t w = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const unsigned long long b = w < nwords ? bitmap[w] : 0ull;
    const uint32_t c = (uint32_t)__popcll(b);
    uint32_t incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = __shfl_up(incl, o, 64);
        if ((int)threadIdx.x >= o) incl += y;
    }
    if (w < nwords) {
        BkRank r;
        r.bits = b;
        r.before = incl - c;
        r.pad = 0u;
        rk[w] = r;
    }
    if (threadIdx.x == 63) blk_cnt[blockIdx.x] = incl;
}
__global__ void __launch_bounds__(256)
rs_assign_kernel(int64_t cap, const RefSlot* __restrict__ tab, uint32_t* __restrict__ tab_lab,
                 const BkRank* __restrict__ rk, const uint32_t* __restrict__ total, uint32_t* __restrict__ counters, uint32_t small_k,
                 uint32_t* __restrict__ first_idx, uint32_t first_cap) {
    if (counters[0] <= small_k || counters[1]) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) counters[2] = total[0];
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t sl = (int64_t)blockIdx.x * 256 + threadIdx.x; sl < cap; sl += stride)
        if (tab[sl].sig) {
            const uint32_t m = tab[sl].min;
            const uint32_t lab = bk_rank_of(m, rk) + 1u;
            tab_lab[sl] = lab;
            if (first_idx && lab <= first_cap) first_idx[lab - 1] = m;
        }
}
size_t refine_rank_slots_workspace_bytes(int64_t len) {
    const int64_t nw = (len + 63) / 64, nb = (nw + 63) / 64;
    return (size_t)nw * 8 + 256 + (size_t)nb * 64 * sizeof(BkRank) + 256 + (size_t)(nb + 1) * 4 + 256 + 64;
}
// tab_lab[slot] = canonical label of the slot's class, counters[2] = number of classes, first_idx as in the other passes.
// Every kernel returns at once when the one-workgroup ranking (<= small_k classes) has done the job or the table overflowed.
bool launch_rank_slots(hipStream_t s, int64_t len, int64_t cap, const RefSlot* tab, uint32_t* tab_lab,
                       uint32_t* counters, uint32_t small_k, uint32_t* first_idx, uint32_t first_cap, void* ws, size_t ws_bytes) {
    if (ws_bytes < refine_rank_slots_workspace_bytes(len) || len >= (int64_t(1) << 32)) return false;
    const int64_t nw = (len + 63) / 64, nb = (nw + 63) / 64;
    auto align = [](char* p) { return (char*)(((uintptr_t)p + 255) & ~uintptr_t(255)); };
    char* p = align((char*)ws);
    unsigned long long* bitmap = (unsigned long long*)p;
    p = align(p + (size_t)nw * 8);
    BkRank* rk = (BkRank*)p;
    p = align(p + (size_t)nb * 64 * sizeof(BkRank));
    uint32_t* blk_cnt = (uint32_t*)p;
    p = align(p + (size_t)(nb + 1) * 4);
    uint32_t* total = (uint32_t*)p;
    if (hipMemsetAsync(bitmap, 0, (size_t)nw * 8, s) != hipSuccess) return false;
    const unsigned gs = (unsigned)std::min<int64_t>((cap + 255) / 256, 2048);
    rs_mark_kernel<<<gs, 256, 0, s>>>(cap, tab, (uint32_t*)bitmap, counters, small_k);
    rs_words_kernel<<<(unsigned)nb, 64, 0, s>>>(nw, bitmap, rk, blk_cnt, counters, small_k);
    bk_scan_kernel<<<1, 1024, 0, s>>>(nb, blk_cnt, total);
    bk_rank_finish_kernel<<<(unsigned)((nb * 64 + 255) / 256), 256, 0, s>>>(nw, blk_cnt, rk);
    rs_assign_kernel<<<gs, 256, 0, s>>>(cap, tab, tab_lab, rk, total, counters, small_k, first_idx, first_cap);
    return true;
}

// ---------------------------------------------------------------------------
// Distinct-signature estimate from a sample (which relabel path a refinement takes is decided from it, not by
// trial and overflow): m <= 65536 entries at stratified positions (one per stride, jittered: no position twice)
// go through a 2^17-slot global table with a count per slot; the finishing workgroup stores
// {sampled non-zero entries, distinct, singletons, doubletons} into pinned host memory.
// ---------------------------------------------------------------------------
constexpr int SMP_LOG2 = 17;
constexpr int64_t SMP_M = 65536;
__global__ void __launch_bounds__(256)
bk_sample_kernel(int64_t len, const uint64_t* __restrict__ sig, int64_t m, int64_t stride, unsigned long long* __restrict__ tab,
                 uint32_t* __restrict__ cnt) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    int64_t e = i * stride + (int64_t)(sdpsr_fmix64((uint64_t)i + 0x5DEECE66DULL) % (uint64_t)stride);
    if (e >= len) e = len - 1;
    const uint64_t sg = sig[e];
    if (!sg) return;
    const uint64_t hh = bk_hash(sg);
    uint32_t sl = (uint32_t)(hh >> 40) & ((1u << SMP_LOG2) - 1);
    for (;;) {  // m <= half the slots: terminates
        const unsigned long long cur = tab[sl];
        if (cur == sg) break;
        if (cur == 0ull) {
            const unsigned long long old = atomicCAS(&tab[sl], 0ull, (unsigned long long)sg);
            if (old == 0ull || old == sg) break;
        }
        sl = (sl + 1) & ((1u << SMP_LOG2) - 1);
    }
    atomicAdd(&cnt[sl], 1u);
}
__global__ void __launch_bounds__(1024)
bk_sample_finish_kernel(const uint32_t* __restrict__ cnt, uint32_t* __restrict__ host_out) {
    __shared__ uint32_t acc[4];
    if (threadIdx.x < 4) acc[threadIdx.x] = 0u;
    __syncthreads();
    uint32_t tot = 0, d = 0, f1 = 0, f2 = 0;
    for (int i = threadIdx.x; i < (1 << SMP_LOG2); i += 1024) {
        const uint32_t c = cnt[i];
        tot += c;
        d += c > 0;
        f1 += c == 1;
        f2 += c == 2;
    }
    atomicAdd(&acc[0], tot);
    atomicAdd(&acc[1], d);
    atomicAdd(&acc[2], f1);
    atomicAdd(&acc[3], f2);
    __syncthreads();
    if (threadIdx.x < 4) host_out[threadIdx.x] = acc[threadIdx.x];
}
size_t refine_sample_workspace_bytes() { return ((size_t)1 << SMP_LOG2) * 12; }
// host_out (pinned): [0] non-zero entries sampled, [1] distinct, [2] seen once, [3] seen twice; returns the sample size
int64_t launch_refine_sample(hipStream_t s, int64_t len, const uint64_t* sig, void* ws, uint32_t* host_out) {
    unsigned long long* tab = (unsigned long long*)ws;
    uint32_t* cnt = (uint32_t*)((char*)ws + ((size_t)8 << SMP_LOG2));
    if (hipMemsetAsync(ws, 0, refine_sample_workspace_bytes(), s) != hipSuccess) return 0;
    const int64_t m = len < SMP_M ? len : SMP_M;
    const int64_t stride = len / m;
    bk_sample_kernel<<<(unsigned)((m + 255) / 256), 256, 0, s>>>(len, sig, m, stride, tab, cnt);
    bk_sample_finish_kernel<<<1, 1024, 0, s>>>(cnt, host_out);
    return m;
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

constexpr size_t B2_SC_LDS_MAX = (size_t)B2_CH * 14 + 3 * (size_t)B2_MAXNB * 4;
constexpr size_t B2_TAB_LDS = (size_t)B2_TS * 12;
constexpr int B2_WGS = 256;  // workgroups of the count / scatter passes (one per CU: 136 KiB of LDS each)

bool refine_bucket_set_device_attributes() {
    bool ok = true;
    ok &= hipFuncSetAttribute(reinterpret_cast<const void*>(&bk_count_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (1 << BK_MAX_LGT) * 4) == hipSuccess;
    ok &= hipFuncSetAttribute(reinterpret_cast<const void*>(&bk_scatter_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SC_LDS_BYTES) == hipSuccess;
    ok &= hipFuncSetAttribute(reinterpret_cast<const void*>(&bk_scatter_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SC_LDS_BYTES) == hipSuccess;
    ok &= hipFuncSetAttribute(reinterpret_cast<const void*>(&b2_scatter_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)B2_SC_LDS_MAX) == hipSuccess;
    ok &= hipFuncSetAttribute(reinterpret_cast<const void*>(&b2_resolve_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)B2_RES_LDS) == hipSuccess;
    ok &= hipFuncSetAttribute(reinterpret_cast<const void*>(&b2_resolve_big_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)B2_TAB_LDS) == hipSuccess;
    return ok;
}

// workspace: front end (one level: 12-byte records; two levels: two (signature, index) buffers of 12 bytes per entry) |
// first (len u32) | bits (u64 per 64 entries) | pre (u32 per 64 entries) | start, cursors | blk_cnt | total, big list
size_t refine_bucketed_workspace_bytes(int64_t len) {
    const int64_t nrb = (len + BK_RB - 1) / BK_RB;
    const size_t front = len <= B2_MAXLEN ? (size_t)(len + B2_CAP) * 12 : (size_t)len * 24;
    return front + (size_t)len * 4 + (size_t)(nrb * 64) * 16 + (size_t)((1 << BK_MAX_LGT) + 1) * 4 * 3 + (size_t)(nrb + 1) * 4 + (size_t)B2_WGS * B2_MAXNB * 4 + 1024 + 16 * 256;
}

// labels_out: canonical labels; counters[0] = counters[2] = number of classes, counters[1] = 0 (1: some bucket could not be
// resolved -- labels_out is then not a partition of the input; the caller repeats the refinement through the radix sort);
// first_idx (may be null): first-occurrence index of class l at [l - 1] for l <= first_cap
bool launch_refine_bucketed(hipStream_t s, int64_t len, const uint64_t* sig, uint32_t* labels_out, void* ws, size_t ws_bytes,
                            uint32_t* counters, uint32_t* first_idx, uint32_t first_cap, uint32_t* host_counters, uint32_t host_seq) {
    if (len < 1 || len >= (int64_t(1) << 31) || ws_bytes < refine_bucketed_workspace_bytes(len)) return false;
    const bool one_level = len <= B2_MAXLEN;
    int lg1 = 0, lg2 = 0;
    if (!one_level) bk_plan(len, &lg1, &lg2);
    const int64_t nbt = one_level ? b2_nb(len) : (int64_t(1) << (lg1 + lg2));
    const int64_t nrb = (len + BK_RB - 1) / BK_RB;
    auto align = [](char* p) { return (char*)(((uintptr_t)p + 255) & ~uintptr_t(255)); };
    char* p = align((char*)ws);
    char* front = p;
    p = align(p + (one_level ? (size_t)(len + B2_CAP) * 12 : (size_t)len * 24));
    uint32_t* first = (uint32_t*)p;
    p = align(p + (size_t)len * 4);
    BkRank* rk = (BkRank*)p;
    p = align(p + (size_t)nrb * 64 * sizeof(BkRank));
    uint32_t* start = (uint32_t*)p;  // nbt + 1 words, the cursors and the small words right behind (one memset)
    uint32_t* cur1 = start + (nbt + 1);
    uint32_t* cur2 = cur1 + (nbt + 1);
    uint32_t* total = cur2 + (nbt + 1);  // [0] classes, [1] failure word, [8 ..] the list of deferred buckets
    uint32_t* big = total + 8;
    const size_t zero_words = (size_t)(nbt + 1) * 3 + 8 + 1 + B2_MAXBIG;
    p = align(p + zero_words * 4);
    uint32_t* blk_cnt = (uint32_t*)p;
    p = align(p + (size_t)(nrb + 1) * 4);
    char* cntp = p;  // one level: counts / offsets per (workgroup, bucket)
    p = align(p + (size_t)B2_WGS * B2_MAXNB * 4);
    if (hipMemsetAsync(start, 0, zero_words * 4, s) != hipSuccess) return false;
    if (one_level) {
        const uint32_t NB = (uint32_t)nbt;
        uint32_t* rec = (uint32_t*)front;
        // workgroup w of the count and scatter passes owns entries [w span, (w + 1) span)
        int64_t span = (len + B2_WGS - 1) / B2_WGS;
        span = (span + B2_CH - 1) / B2_CH * B2_CH;
        const unsigned G = (unsigned)((len + span - 1) / span);
        uint32_t* cnt = (uint32_t*)cntp;
        b2_count_kernel<<<G, 1024, (size_t)NB * 4, s>>>(len, sig, NB, span, start, cnt, first);
        bk_scan_kernel<<<1, 1024, 0, s>>>(nbt, start, start + nbt);
        b2_offsets_kernel<<<(NB + 3) / 4, 256, 0, s>>>(NB, G, start, cnt);
        b2_scatter_kernel<<<G, B2_THREADS, (size_t)B2_CH * 14 + 3 * (size_t)NB * 4, s>>>(len, sig, NB, span, cnt, rec);
#ifdef LK_TIMING
        long long* dbg = nullptr;
        if (dbg_on()) {
            hipMalloc(&dbg, 4 * 32 * 8);
            hipMemset(dbg, 0, 4 * 32 * 8);
            hipMemcpyToSymbol(HIP_SYMBOL(b2_dbg), &dbg, sizeof(dbg));
        }
#endif
        b2_resolve_kernel<<<(unsigned)std::min<int64_t>(NB, 512), B2_RTHREADS, B2_RES_LDS, s>>>((int)NB, start, rec, first, big, total + 1);
#ifdef LK_TIMING
        if (dbg) {
            hipStreamSynchronize(s);
            long long h[4 * 32];
            hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost);
            for (int w = 0; w < 4; ++w) {
                const long long* t = &h[w * 32];
                if (!t[0]) continue;
                fprintf(stderr, "[resolve timing] wg=%d bucket=%d (us from the bucket's start; per sub-pass: scan, compacted, inserted, looked up, barrier):", w / 2 ? 100 : 0, w % 2);
                for (int i = 1; i < 31; ++i)
                    if (t[i]) fprintf(stderr, " [%d]%.2f", i, (t[i] - t[0]) * 0.01);
                fprintf(stderr, "\n");
            }
            long long* z = nullptr;
            hipMemcpyToSymbol(HIP_SYMBOL(b2_dbg), &z, sizeof(z));
            hipFree(dbg);
        }
#endif
        b2_resolve_big_kernel<<<B2_MAXBIG * B2_BIGSP, B2_THREADS, B2_TAB_LDS, s>>>(start, rec, first, big, total + 1);
    } else {
        const int lgt = lg1 + lg2;
        uint64_t* sigA = (uint64_t*)front;
        uint64_t* sigB = sigA + len;
        uint32_t* idxA = (uint32_t*)(sigB + len);
        uint32_t* idxB = idxA + len;
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
    }
    bk_bits_kernel<<<(unsigned)nrb, 256, 0, s>>>(len, first, rk, blk_cnt);
    bk_scan_kernel<<<1, 1024, 0, s>>>(nrb, blk_cnt, total);
    bk_rank_finish_kernel<<<(unsigned)((nrb * 64 + 255) / 256), 256, 0, s>>>((len + 63) / 64, blk_cnt, rk);
    const int64_t g = std::min<int64_t>(nrb, 256 * 8);
    bk_label_kernel<<<(unsigned)g, 256, 0, s>>>(len, first, labels_out, rk, total, counters, host_counters, first_idx, first_cap, host_seq);
    return true;
}

}  // namespace sdpsr
