// Host-side declarations shared by the orchestration translation units (ctx.cpp, primitives.cpp,
// loop.cpp, eigdec.cpp, compress.cpp, blockdiag.cpp, complex.cpp) and by the measurement library
// (prof/).  Nothing here is part of the ABI.
#pragma once
#include <algorithm>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "sdpsr_internal.h"

namespace sdpsr {

// SDPSR_DEBUG=1: host wall-clock marks on stderr (relative to the ctx's previous mark); the only
// environment variable the library reads, and it changes no result
bool dbg_on();
void dbg_mark(sdpsr_ctx* c, const char* what);

struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int dev) {
        hipGetDevice(&prev);
        if (prev != dev) hipSetDevice(dev);
    }
    ~DeviceGuard() {
        int cur;
        hipGetDevice(&cur);
        if (prev >= 0 && cur != prev) hipSetDevice(prev);
    }
};

template <typename T>
const T* in_dev(sdpsr_ctx* c, const char* name, const T* p, size_t count, int mem, int* st) {
    if (mem == SDPSR_MEM_DEVICE || p == nullptr) return p;
    T* d = (T*)ctx_buf(c, name, count * sizeof(T));
    if (!d) {
        *st = SDPSR_OUT_OF_MEMORY;
        return nullptr;
    }
    if (hipMemcpyAsync(d, p, count * sizeof(T), hipMemcpyHostToDevice, c->stream) != hipSuccess) {
        *st = ctx_fail(c, SDPSR_HIP_ERROR, std::string("H2D copy of ") + name);
        return nullptr;
    }
    c->h2d_bytes += count * sizeof(T);
    return d;
}

template <typename T>
T* out_dev(sdpsr_ctx* c, const char* name, T* p, size_t count, int mem, int* st) {
    if (mem == SDPSR_MEM_DEVICE) return p;
    T* d = (T*)ctx_buf(c, name, count * sizeof(T));
    if (!d) *st = SDPSR_OUT_OF_MEMORY;
    return d;
}

template <typename T>
int out_finish(sdpsr_ctx* c, T* host, const T* dev, size_t count, int mem) {
    // outputs are complete on return in both memory spaces (ordering rule of sdpsr.h)
    if (mem != SDPSR_MEM_DEVICE) {
        HIP_TRY(c, hipMemcpyAsync(host, dev, count * sizeof(T), hipMemcpyDeviceToHost, c->stream));
        c->d2h_bytes += count * sizeof(T);
    }
    HIP_TRY(c, ctx_sync_stream(c, c->stream));
    return SDPSR_OK;
}

inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

bool ctx_ensure_side(sdpsr_ctx* c);
int d2h_sync(sdpsr_ctx* c, void* host, const void* dev, size_t bytes);
// Upload of a small host array.  Up to 32 KiB go through a ring of pinned slots and stay
// stream-ordered (the caller's buffer is free on return, no host wait)
int h2d_sync(sdpsr_ctx* c, void* dev, const void* host, size_t bytes);
inline int ceil_log2(uint64_t x) {
    int l = 0;
    while ((uint64_t(1) << l) < x) ++l;
    return l;
}


double round_scale(const sdpsr_ctx* c, double atol);
bool label_overflows(const sdpsr_ctx* c, uint64_t value);
int label_overflow_fail(sdpsr_ctx* c, const char* where, uint64_t value);
uint64_t next_key(sdpsr_ctx* c);
int check_len(sdpsr_ctx* c, int64_t len);
// reduce.cpp / batch.cpp: the entry points with the memory spaces of the inputs and of the outputs named separately
int jordan_reduce_impl(sdpsr_ctx* c, int64_t n, const double* CL, const double* X0L, const double* U, int64_t r, double atol, double epsilon,
                       uint32_t* P_out, int64_t* dim_out, int32_t* iters_out, int32_t* nblocks, int64_t* sum_sq, int64_t* sum_s, double* blks,
                       int64_t blks_capacity, double* Q_hat, int64_t qhat_capacity, double* phase_ms, int mem_in, int mem_out);
int jordan_reduce_batch_impl(sdpsr_ctx* c, int32_t R, const uint64_t* seeds, int64_t n, const double* CL, const double* X0L, const double* U,
                             int64_t r, int hint, double atol, double epsilon, uint32_t* const* P_out, int64_t* dim_out, int32_t* iters_out,
                             int32_t* nblocks, int64_t* sum_sq, int64_t* sum_s, double* const* blks, const int64_t* blks_capacity,
                             int32_t* status, int mem_in, int mem_out);

// ---- phase timing with events; collected after the syncs the loop needs anyway ----
struct PhaseTimer {
    sdpsr_ctx* c;
    double acc[SDPSR_T_COUNT] = {};
    struct Pending {
        int slot;
        hipEvent_t a, b;
    };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> pool;
    bool enabled;
    PhaseTimer(sdpsr_ctx* ctx, bool en) : c(ctx), enabled(en) {}
    ~PhaseTimer() {
        for (auto e : pool) hipEventDestroy(e);
        for (auto& p : pending) {
            hipEventDestroy(p.a);
            hipEventDestroy(p.b);
        }
    }
    hipEvent_t get() {
        if (!pool.empty()) {
            hipEvent_t e = pool.back();
            pool.pop_back();
            return e;
        }
        hipEvent_t e;
        hipEventCreate(&e);
        return e;
    }
    int cur_slot = -1;
    hipEvent_t cur_a{};
    void begin(int slot) {
        if (!enabled) return;
        cur_slot = slot;
        cur_a = get();
        hipEventRecord(cur_a, c->stream);
    }
    void end() {
        if (!enabled || cur_slot < 0) return;
        hipEvent_t b = get();
        hipEventRecord(b, c->stream);
        pending.push_back({cur_slot, cur_a, b});
        cur_slot = -1;
    }
    void collect() {  // intervals whose end event has not completed yet stay pending
        std::vector<Pending> later;
        for (auto& p : pending) {
            float ms = 0;
            const hipError_t e = hipEventElapsedTime(&ms, p.a, p.b);
            if (e == hipErrorNotReady) {
                later.push_back(p);
                continue;
            }
            if (e == hipSuccess) acc[p.slot] += ms;
            pool.push_back(p.a);
            pool.push_back(p.b);
        }
        (void)hipGetLastError();
        pending.swap(later);
    }
};

// start / stop events of a whole entry point (phase_ms[SDPSR_T_TOTAL]); freed on every exit path
struct TotalEvents {
    hipEvent_t a = nullptr, b = nullptr;
    TotalEvents(bool enabled, hipStream_t s) {
        if (!enabled) return;
        if (hipEventCreate(&a) != hipSuccess) a = nullptr;
        if (hipEventCreate(&b) != hipSuccess) b = nullptr;
        if (a) hipEventRecord(a, s);
    }
    ~TotalEvents() {
        if (a) hipEventDestroy(a);
        if (b) hipEventDestroy(b);
    }
    float stop(hipStream_t s) {  // records the end, waits for it, returns the milliseconds
        float ms = 0;
        if (!a || !b) return ms;
        hipEventRecord(b, s);
        hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
        return ms;
    }
    TotalEvents(const TotalEvents&) = delete;
    TotalEvents& operator=(const TotalEvents&) = delete;
};

// canonical refinement of a signature source / array (primitives.cpp)
// early: return as soon as the label pass has REPORTED the class count (ctx_wait_word) -- the pass itself is still running; only
// for a caller that goes on in stream order and waits for the stream before its own return
int refine_signatures(sdpsr_ctx* c, int64_t len, const SigSource& src_in, uint32_t* labels, int64_t* nparts, int64_t sym_n = 0,
                      uint32_t* symflag_dev = nullptr, int* sym_out = nullptr, bool early = false);
int refine_signatures(sdpsr_ctx* c, int64_t len, const uint64_t* sig, uint32_t* labels, int64_t* nparts, int64_t sym_n = 0,
                      uint32_t* symflag_dev = nullptr, int* sym_out = nullptr, bool early = false);

// loop.cpp / blockdiag.cpp: the bodies of the entry points, chainable without host synchronisation in between
// (reduce.cpp: sdpsr_jordan_reduce)
int admissible_subspace_impl(sdpsr_ctx* c, int64_t n, const double* CL, const double* X0L, const double* U, int64_t r, double atol,
                             uint32_t* P_out, int64_t* dim_out, int32_t* iters_out, double* phase_ms, int mem, int mem_out,
                             bool final_sync, int* labels_sym_out);
// trusted_symmetric: the caller made the labels and knows; in_place (device labels only): no copy into ctx buffer
// "bd_labels" -- P itself serves phase 2 (c->bd_labels_ext) until the caller ends that arrangement
int block_diagonalize_impl(sdpsr_ctx* c, int64_t n, const uint32_t* P, int64_t d, double epsilon, int32_t* nblocks, int64_t* sum_sq,
                           int64_t* sum_s, double* phase_ms, int mem, bool trusted_symmetric, bool final_sync, bool in_place = false);

// ---- blockDiagonalize: host pieces and drivers (eigdec.cpp, compress.cpp) ----
struct EigInfo {
    std::vector<double> vals;
    std::vector<int> ptrs;  // 0-based boundaries, size neig+1
    std::vector<int> kpart; // root of every eigenspace
    // "bd_t" holds T = A2 Q for the ONE generic element A2 whose couplings decided the classes
    // (false after extra coupling elements: the decisive block may come from any of them)
    bool t_valid = false;
};


// Source of "generic elements" for the dense driver: the label gather (gen == nullptr,
// randomize!(A, P)) or a compressed representation B = W' A W of it (module-compression driver).
struct ElemGen {
    // writes an (n_eff x n_eff, leading dimension ld_eff, zero padded) symmetric matrix
    std::function<int(double* dst)> make;
    // optional: start making the NEXT element into dst on a side stream (returns 0 if started),
    // and make the main stream wait for it
    std::function<int(double* dst)> prefetch;
    std::function<int()> join;
    // optional: mark the fork point on the main stream NOW; a later prefetch() starts from this
    // point (so that the eigensolver can be enqueued first and its launches do not wait behind the
    // prefetch's host-side launch work)
    std::function<int()> fork;
};
int make_element(sdpsr_ctx* c, const ElemGen* gen, int64_t n, int64_t ld, const uint32_t* L, double* dst);
double otsu_threshold(const std::vector<double>& X, double atol);
int isomorphism_classes(sdpsr_ctx* c, const std::vector<double>& norms, int neig, double atol, std::vector<int>& kpart);
void class_structure(const std::vector<int>& kpart, std::vector<int>& roots, std::vector<std::vector<int>>& members);
int eigen_decomposition_device(sdpsr_ctx* c, int64_t n, const uint32_t* L, double atol, EigInfo& info, PhaseTimer& tm,
                               const ElemGen* gen = nullptr, int64_t expect_dim = -1);
// status used internally when a driver of diagonalize hands over to the dense one
constexpr int DRIVER_FALLBACK = -1000;
int driver_fallback(sdpsr_ctx* c, const std::string& why);
int dense_diagonalize(sdpsr_ctx* c, int64_t n, const uint32_t* L, const ElemGen* gen, double atol, EigInfo& info,
                      std::vector<int32_t>& sizes, int64_t& S1, int64_t& S, PhaseTimer& tm, int64_t expect_dim = -1);
int gemm_tn_splitk(sdpsr_ctx* c, int64_t m, int64_t n, int64_t k, const double* A, int64_t lda, const double* B, int64_t ldb,
                   double* C, int64_t ldc);
// host_C (optional, pinned host memory, mp x np like C): filled by the product itself when the exact-shape kernel
// runs; *host_filled says whether it did (the padded split-K path does not)
int gram_tn(sdpsr_ctx* c, int64_t ma, int64_t nb, int64_t k, const double* A, int64_t lda, const double* B, int64_t ldb, double* C,
            int64_t mp, int64_t np, double* host_C = nullptr, bool* host_filled = nullptr);
int compressed_diagonalize(sdpsr_ctx* c, int64_t n, const uint32_t* L, int64_t d, double atol, EigInfo& info,
                           std::vector<int32_t>& sizes, int64_t& S1, int64_t& S, PhaseTimer& tm);
bool compression_eligible(const sdpsr_ctx* c, int64_t n, int64_t d);

// small_eigen_host.cpp: the compressed problem (order w <= 64) on the host
int host_syev(int n, const double* A, int lda, double* w, double* Z, int ldz);  // host_syev.cpp
void host_count_edges17(const double* x, size_t n, const double* ed, int64_t* hist);
void host_gemm_tn(int m, int n, int k, const double* A, int lda, const double* B, int ldb, double* C, int ldc);
int murota_small_host(sdpsr_ctx* c, int w, const double* B1, const std::function<int(double*)>& next_element, double atol,
                      int64_t expect_dim, std::vector<int32_t>& sizes, int64_t& S1, int64_t& S, std::vector<double>& Qs);

// ---- kernel launchers used by the host files only (kernels_module.hip, kernels_sytrd.hip, ...) ----
void launch_symmetrize(hipStream_t s, int64_t m, int64_t ld, double* B);
void launch_extract_symmetric(hipStream_t s, int64_t m, int64_t mp, const double* src, int64_t lds_, double* dst);
void launch_splitk_reduce(hipStream_t s, int64_t len, int Z, int64_t stride, const double* P, double* C);
size_t gram_small_partial_doubles(int64_t k, int ma, int nb);
void launch_gram_small(hipStream_t s, int64_t k, int ma, int nb, const double* A, int64_t lda, const double* B, int64_t ldb,
                       double* partials, double* C, int64_t ldc, int mp, int np, double* host_C = nullptr);
size_t label_spmm_partial_doubles(int64_t n, int w);
bool launch_label_spmm_multi(hipStream_t s, int64_t n, const uint32_t* L, const uint64_t* keys, int G, int64_t d, const double* W,
                             int64_t ldw, int w, double* partials, double* Y, int64_t ldy);
bool launch_label_spmm(hipStream_t s, int64_t n, const uint32_t* L, uint64_t key, int64_t d, const double* W,
                       int64_t ldw, int w, double* partials, double* Y, int64_t ldy);
void launch_tall_times_small(hipStream_t s, int64_t n, int64_t ldi, const double* In, int kk, const double* S,
                             int lds_, int ncols, double alpha, double beta, double* out, int64_t ldo);
void launch_transpose_rows(hipStream_t s, int64_t len, int64_t m, const double* A, double* R);
void launch_col_norms2(hipStream_t s, int64_t len, int64_t k, const double* V, double* partial, int nblk, double* out);
void launch_scale_copy(hipStream_t s, int64_t len, const double* v, double alpha, double* out);
void launch_rank1_update(hipStream_t s, int64_t len, int64_t m, double* R, const double* u, const double* dots);
void launch_sub_round(hipStream_t s, int64_t len, const double* a, const double* b, double atol, double scale, double* out);
void launch_normalize_columns(hipStream_t s, int64_t n, int64_t ld, double* H, int64_t hstride, const double* X,
                         int64_t ldx, int nruns, double* norm0);
void launch_random_vector(hipStream_t s, int64_t n, uint64_t key, double* x);
void launch_fill_test_sig(hipStream_t s, int64_t len, int64_t nclasses, uint64_t* sig);
size_t sytrd_workspace_doubles(int64_t n, int64_t ld);
void launch_sytrd(sdpsr_ctx* c, int64_t n, double* A, int64_t ld, double* d, double* e, double* tau, double* ws);
void launch_sytrd_symv_sweep(hipStream_t s, int64_t n, double* A, int64_t ld, double* d, double* e, double* tau, double* ws);
bool launch_small_syev(hipStream_t s, int64_t n, double* A, int64_t lda, double* w, double* Vtmp, int* info);

}  // namespace sdpsr

#define CHECK_CTX(c) \
    if (!(c)) return SDPSR_BAD_ARGUMENT; \
    sdpsr::DeviceGuard _dg((c)->device); \
    (c)->err.clear();
