// Batched small-N mode: `count` independent runs of eigen_decomposition(P, A; atol)
// (src/eigen_decomposition.jl:236-273) on one partition of order n <= 64, one workgroup per run,
// every CU busy.  The reference's robustness pin runs it 10 000 times on a 64 x 64 partition
// (test/numerical_issues.jl:85-94); one problem per launch leaves 255 of 256 CUs idle.
//
// Per run, entirely inside one workgroup (matrices in LDS):
//   generic element #1 (randomize!, :242)  ->  parallel Jacobi eigen (:246, jacobi64.h)
//   -> EigenDecomposition ctor: sort, cluster by |dv| > atol (:19-40)
//   -> generic element #2 (:259), Q'AQ (:203), block max-norms (:177-193)
//   -> otsu_threshold + log_histogram (:83-139) -> IntDisjointSets merges (:205-217, union by
//      rank with path compression exactly as DataStructures.jl, executed by one thread)
//   -> __isconsistent (:163-167).
// Outputs per run: status (OK / NUMERICAL_INCONSISTENCY / NOT_CONVERGED), number of
// eigenspaces, number of isomorphism classes.
#include "sdpsr_internal.h"
#include "jacobi64.h"

namespace sdpsr {

struct BatchedEigdecArgs {
    int n, d, count;
    const uint32_t* L;     // n x n labels, column-major
    const double* values;  // optional: 2 * count * d explicit class values (run r: rows 2r, 2r+1)
    uint64_t seed, stream_base;
    double atol;
    int32_t* status;
    int32_t* neig;
    int32_t* nclasses;
};

constexpr int BE_THREADS = 1024;

__device__ __forceinline__ double be_class_value(const BatchedEigdecArgs& a, int run, int e, uint64_t key, uint32_t l) {
    if (l == 0u) return 0.0;
    if (a.values) return a.values[((int64_t)(2 * run + e)) * a.d + (l - 1)];
    return sdpsr_class_uniform(key, l);
}

__global__ void __launch_bounds__(BE_THREADS)
eigdec_batched64_kernel(BatchedEigdecArgs a) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ double s_red[BE_THREADS / 64 + 2];
    __shared__ double s_w[64], s_edges[17], s_scal[4];
    __shared__ int s_rank[64], s_space[64], s_dim[64], s_par[64], s_rk[64], s_kp[64], s_info[4];
    __shared__ unsigned int s_cnt[16];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int n = a.n;
    const int m = (n + 1) & ~1, half = m >> 1, ldl = m | 1;
    double* __restrict__ B0 = smem;
    double* __restrict__ B1 = B0 + (size_t)ldl * m;
    double* __restrict__ B2 = B1 + (size_t)ldl * m;
    unsigned long long* __restrict__ norms = reinterpret_cast<unsigned long long*>(B2 + (size_t)ldl * m);
    int* __restrict__ s_pq = reinterpret_cast<int*>(norms + (size_t)n * n);
    jacobi64_fill_pairs(m, s_pq);

    for (int run = blockIdx.x; run < a.count; run += gridDim.x) {
        const uint64_t key1 = sdpsr_stream_key(a.seed, a.stream_base + 2 * (uint64_t)run);
        const uint64_t key2 = sdpsr_stream_key(a.seed, a.stream_base + 2 * (uint64_t)run + 1);
        __syncthreads();
        // ---- generic element #1 and its eigendecomposition
        for (int e = tid; e < m * m; e += nthr) {
            const int j = e / m, i = e - j * m;
            double v = 0.0;
            if (i < n && j < n) v = be_class_value(a, run, 0, key1, a.L[i + (int64_t)j * n]);
            B0[i + j * ldl] = v;
            B1[i + j * ldl] = (i == j) ? 1.0 : 0.0;
        }
        __syncthreads();
        const int sweeps = jacobi64_sweeps(n, m, ldl, B0, B1, s_pq, s_red);
        // ---- ascending order (ties by index), V sorted into B2
        for (int i = tid; i < n; i += nthr) {
            const double li = B0[i + i * ldl];
            int rk = 0;
            for (int j = 0; j < n; ++j) {
                const double lj = B0[j + j * ldl];
                rk += (lj < li) || (lj == li && j < i);
            }
            s_rank[i] = rk;
            s_w[rk] = li;
        }
        if (tid < 64) s_dim[tid] = 0;
        __syncthreads();
        for (int e = tid; e < n * n; e += nthr) {
            const int j = e / n, i = e - j * n;
            B2[i + s_rank[j] * ldl] = B1[i + j * ldl];
        }
        // ---- EigenDecomposition ctor: a new eigenspace where !(|w[i] - w[i-1]| <= atol)
        if (tid < 64) {
            const bool bnd = tid > 0 && tid < n && !(fabs(s_w[tid] - s_w[tid - 1]) <= a.atol);
            const unsigned long long mask = __ballot(bnd);
            const unsigned long long upto = (tid == 63) ? ~0ull : ((2ull << tid) - 1ull);
            const int sp = __popcll(mask & upto);
            if (tid < n) {
                s_space[tid] = sp;
                atomicAdd(&s_dim[sp], 1);
            }
            if (tid == 0) s_info[0] = __popcll(mask) + 1;
        }
        __syncthreads();
        const int neig = s_info[0];
        // ---- generic element #2 -> B0;  T = A2 V -> B1
        for (int e = tid; e < n * n; e += nthr) {
            const int j = e / n, i = e - j * n;
            B0[i + j * ldl] = be_class_value(a, run, 1, key2, a.L[i + (int64_t)j * n]);
        }
        for (int e = tid; e < neig * neig; e += nthr) norms[e] = 0ull;
        if (tid < 16) s_cnt[tid] = 0u;
        __syncthreads();
        for (int e = tid; e < n * n; e += nthr) {
            const int j = e / n, i = e - j * n;
            double acc = 0.0;
            for (int k = 0; k < n; ++k) acc = fma(B0[k + i * ldl], B2[k + j * ldl], acc);  // A2 symmetric: row i = column i
            B1[i + j * ldl] = acc;
        }
        __syncthreads();
        // ---- M = V' T; block max-norms over (rows in E_i, columns in E_j), i <= j, equal dimensions
        for (int e = tid; e < n * n; e += nthr) {
            const int bcol = e / n, arow = e - bcol * n;
            const int sa = s_space[arow], sb = s_space[bcol];
            if (sa > sb || s_dim[sa] != s_dim[sb]) continue;
            double acc = 0.0;
            for (int k = 0; k < n; ++k) acc = fma(B2[k + arow * ldl], B1[k + bcol * ldl], acc);
            atomicMax(&norms[sa * neig + sb], (unsigned long long)__double_as_longlong(fabs(acc)));
        }
        __syncthreads();
        // end_norm[i, j] = end_norm[j, i] (:185-188)
        for (int e = tid; e < neig * neig; e += nthr) {
            const int j = e / neig, i = e - j * neig;  // entry (i, j) at i * neig + j?  use (r, c) = (i, j)
            if (i > j) norms[i * neig + j] = norms[j * neig + i];
        }
        __syncthreads();
        // ---- otsu_threshold(end_infnorm; atol): min / max, 16 log-spaced bins (:83-139)
        {
            double mn = INFINITY, mx = 0.0;
            for (int e = tid; e < neig * neig; e += nthr) {
                const double x = __longlong_as_double((long long)norms[e]);
                mn = fmin(mn, x);
                mx = fmax(mx, x);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                mn = fmin(mn, __shfl_down(mn, o, 64));
                mx = fmax(mx, __shfl_down(mx, o, 64));
            }
            if ((tid & 63) == 0) {
                s_red[tid >> 6] = mn;
            }
            __syncthreads();
            if (tid == 0) {
                double t = INFINITY;
                for (int k = 0; k < nthr / 64; ++k) t = fmin(t, s_red[k]);
                s_scal[0] = t;
            }
            __syncthreads();
            if ((tid & 63) == 0) s_red[tid >> 6] = mx;
            __syncthreads();
            if (tid == 0) {
                double t = 0.0;
                for (int k = 0; k < nthr / 64; ++k) t = fmax(t, s_red[k]);
                s_scal[1] = t;
            }
            __syncthreads();
        }
        constexpr int NB = 16;  // max(ceil(-log10(eps(Float64))), 4)
        if (tid <= NB) {
            double mnv = s_scal[0];
            if (mnv < a.atol) mnv = a.atol;
            const double l0 = log(mnv), l1 = log(s_scal[1]);
            const double tt = (tid == NB) ? l1 : l0 + (l1 - l0) * (double)tid / (double)NB;
            s_edges[tid] = exp(tt);
        }
        __syncthreads();
        for (int e = tid; e < neig * neig; e += nthr) {
            const double x = __longlong_as_double((long long)norms[e]);
            int f = NB + 1;  // something(findfirst(b -> b > x, edges), NB + 1), 1-based
            for (int i = 0; i <= NB; ++i)
                if (s_edges[i] > x) {
                    f = i + 1;
                    break;
                }
            int bin = f - 1;
            bin = bin < 1 ? 1 : (bin > NB ? NB : bin);
            atomicAdd(&s_cnt[bin - 1], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            double total = 0;
            for (int i = 0; i < NB; ++i) total += (double)s_cnt[i];
            double w[NB], mu[NB];
            double cw = 0, cm = 0;
            for (int i = 0; i < NB; ++i) {
                const double p = (double)s_cnt[i] / total;
                cw += p;
                cm += log(s_edges[i]) * p;
                w[i] = cw;
                mu[i] = cm;
            }
            const double muT = mu[NB - 1];
            int best = 0;
            double bestv = -INFINITY;
            bool have_nan = false;
            for (int i = 0; i < NB - 1; ++i) {
                const double num = muT * w[i] - mu[i];
                const double s2 = num * num / (w[i] * (1 - w[i]));
                if (s2 != s2) {  // Julia argmax returns the first NaN
                    if (!have_nan) {
                        best = i;
                        have_nan = true;
                    }
                } else if (!have_nan && s2 > bestv) {
                    bestv = s2;
                    best = i;
                }
            }
            const double thr = s_edges[best + 1];
            // ---- IntDisjointSets: union by rank, path compression (:205-217)
            for (int i = 0; i < neig; ++i) {
                s_par[i] = i;
                s_rk[i] = 0;
            }
            auto find = [&](int x) {
                int r = x;
                while (s_par[r] != r) r = s_par[r];
                while (s_par[x] != r) {
                    const int nx = s_par[x];
                    s_par[x] = r;
                    x = nx;
                }
                return r;
            };
            for (int i = 0; i < neig; ++i)
                for (int j = i + 1; j < neig; ++j)
                    if (__longlong_as_double((long long)norms[i * neig + j]) >= thr) {
                        int x = find(i), y = find(j);
                        if (x == y) continue;
                        if (s_rk[x] < s_rk[y]) {
                            const int t = x;
                            x = y;
                            y = t;
                        } else if (s_rk[x] == s_rk[y]) {
                            ++s_rk[x];
                        }
                        s_par[y] = x;
                    }
            // ---- __isconsistent (:163-167): every root is the first member of its class
            int ncls = 0;
            bool ok = true;
            for (int i = 0; i < neig; ++i) s_kp[i] = find(i);
            for (int i = 0; i < neig; ++i) {
                const int r = s_kp[i];
                int first = 0;
                while (s_kp[first] != r) ++first;
                if (first == i) {  // first occurrence of this root
                    ++ncls;
                    if (r != i) ok = false;
                }
            }
            int st = SDPSR_OK;
            if (sweeps >= 40) st = SDPSR_NOT_CONVERGED;
            else if (!ok) st = SDPSR_NUMERICAL_INCONSISTENCY;
            a.status[run] = st;
            a.neig[run] = neig;
            a.nclasses[run] = ncls;
        }
    }
}

size_t batched_eigdec_lds_bytes(int64_t n) {
    const int64_t m = (n + 1) & ~int64_t(1), half = m / 2, ldl = m | 1;
    return (size_t)(3 * ldl * m) * 8 + (size_t)n * n * 8 + (size_t)(m - 1) * half * 4 + 64;
}

bool batched_set_device_attributes() {
    bool ok = true;
    ok &= hipSuccess == hipFuncSetAttribute(reinterpret_cast<const void*>(&eigdec_batched64_kernel),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    return ok;
}

// status/neig/nclasses: device arrays of `count` ints.
void launch_eigdec_batched64(hipStream_t s, int64_t n, int64_t d, int64_t count, const uint32_t* L, const double* values,
                             uint64_t seed, uint64_t stream_base, double atol, int32_t* status, int32_t* neig,
                             int32_t* nclasses, int num_cus) {
    BatchedEigdecArgs a;
    a.n = (int)n;
    a.d = (int)d;
    a.count = (int)count;
    a.L = L;
    a.values = values;
    a.seed = seed;
    a.stream_base = stream_base;
    a.atol = atol;
    a.status = status;
    a.neig = neig;
    a.nclasses = nclasses;
    const size_t lds = batched_eigdec_lds_bytes(n);
    // workgroups per CU by LDS (160 KiB per CU; the 1024-thread workgroup caps it at 2)
    int per_cu = (int)((160 * 1024) / (lds + 2048));
    per_cu = per_cu < 1 ? 1 : (per_cu > 2 ? 2 : per_cu);
    int64_t grid = (int64_t)num_cus * per_cu;
    if (grid > count) grid = count;
    eigdec_batched64_kernel<<<(unsigned)grid, BE_THREADS, lds, s>>>(a);
}

}  // namespace sdpsr
