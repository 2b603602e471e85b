// R independent random restarts of one reduction in ONE call on ONE host thread (include/sdpsr.h:
// sdpsr_jordan_reduce_batch).  A reduction is a chain of ~60 launches with eight host round trips (refinement counters,
// verify verdicts, the module growth's small Gram matrices, the compressed eigenproblem): while the host waits for one
// restart's verdict the GPU runs dry unless another restart's launches are queued.  Every restart gets its own ctx
// (stream, buffers, random streams) and its own FIBER (ucontext) of the calling thread; every host wait of the library
// goes through ctx_sync_stream, which inside a batch polls the stream and, while it is busy, switches to the next
// restart's fiber.  No threads, no locks, no process-global state: the scheduler lives on the caller's stack.
// Reference: the restarts are the "try again" of src/eigen_decomposition.jl:264-270 / src/diagonalize.jl:4-9 and the
// independent draws of the loop (src/partitions.jl:154-185), run side by side instead of one after the other.
#include <sys/mman.h>
#include <ucontext.h>
#include <unistd.h>

#include <algorithm>
#include <cstring>
#include <functional>
#include <memory>
#include <new>
#include <vector>

#include "host_internal.h"

using namespace sdpsr;

namespace {

struct Sched;
// a fiber's stack: its own mapping with an inaccessible page below it (an overflow inside HIP or a solver library
// faults there instead of running into the heap)
struct FiberStack {
    void* base = nullptr;
    size_t bytes = 0;  // whole mapping, guard page included
    size_t guard = 0;
    FiberStack() = default;
    FiberStack(const FiberStack&) = delete;
    FiberStack& operator=(const FiberStack&) = delete;
    bool map(size_t usable) {
        const long pg = sysconf(_SC_PAGESIZE);
        guard = pg > 0 ? (size_t)pg : 4096;
        bytes = usable + guard;
        void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_STACK, -1, 0);
        if (p == MAP_FAILED) return false;
        if (mprotect(p, guard, PROT_NONE) != 0) {
            munmap(p, bytes);
            return false;
        }
        base = p;
        return true;
    }
    void* sp() const { return (char*)base + guard; }
    size_t size() const { return bytes - guard; }
    ~FiberStack() {
        if (base) munmap(base, bytes);
    }
};
struct Fiber {
    ucontext_t uc;
    FiberStack stack;
    Sched* sched = nullptr;
    std::function<int()> body;
    int status = SDPSR_OK;
    bool done = false;
};
struct Sched {
    ucontext_t main;
    std::unique_ptr<Fiber[]> fibers;  // (addresses handed to makecontext: never moved)
};

constexpr size_t kFiberStack = size_t(1) << 20;

void fiber_entry(unsigned lo, unsigned hi) {
    Fiber* f = reinterpret_cast<Fiber*>(((uintptr_t)hi << 32) | (uintptr_t)lo);
    try {  // nothing may unwind past the fiber's first frame
        f->status = f->body();
    } catch (const std::bad_alloc&) {
        f->status = SDPSR_OUT_OF_MEMORY;
    } catch (...) {
        f->status = SDPSR_HIP_ERROR;
    }
    f->done = true;
    swapcontext(&f->uc, &f->sched->main);
}

void fiber_yield(void* arg) {
    Fiber* f = static_cast<Fiber*>(arg);
    swapcontext(&f->uc, &f->sched->main);
}

}  // namespace

namespace sdpsr {
int jordan_reduce_batch_impl(sdpsr_ctx* c, int32_t R, const uint64_t* seeds, int64_t n, const double* CL, const double* X0L, const double* U,
                             int64_t r, int hint, double atol, double epsilon, uint32_t* const* P_out, int64_t* dim_out, int32_t* iters_out,
                             int32_t* nblocks, int64_t* sum_sq, int64_t* sum_s, double* const* blks, const int64_t* blks_capacity,
                             int32_t* status, int mem_in, int mem_out) {
    CHECK_CTX(c);
    if (R < 1 || R > 64 || !dim_out || !status || (blks && !blks_capacity)) return ctx_fail(c, SDPSR_BAD_ARGUMENT, "bad arguments");
    if (c->yield_fn) return ctx_fail(c, SDPSR_BAD_STATE, "sdpsr_jordan_reduce_batch called from inside a batch");
    // every restart reports: nothing below may leave status[] as the caller initialised it (a failure before the
    // fibers start looked like R clean restarts of dimension 0 to a caller that only reads status[])
    for (int i = 0; i < R; ++i) status[i] = SDPSR_BAD_STATE;
    if (!CL || !X0L || n < 1 || r < 0 || (r > 0 && !U)) return ctx_fail(c, SDPSR_BAD_ARGUMENT, "bad arguments");
    // Host arrays are uploaded ONCE, into this ctx's input buffers; all R restarts read them there (the inputs are
    // read-only).  (Round 4 handed `mem` to every restart: R uploads of C_L, X0, U -- 3 x 134 MB each at N = 4096.)
    if (mem_in != SDPSR_MEM_DEVICE) {
        const int64_t len = n * n;
        int st0 = check_len(c, len);
        if (st0) return st0;
        const double* dCL = in_dev(c, "adm_cl", CL, (size_t)len, mem_in, &st0);
        const double* dX0 = in_dev(c, "adm_x0", X0L, (size_t)len, mem_in, &st0);
        const double* dU = r > 0 ? in_dev(c, "adm_u", U, (size_t)len * (size_t)r, mem_in, &st0) : nullptr;
        if (st0) return st0;
        CL = dCL;
        X0L = dX0;
        U = dU;
        mem_in = SDPSR_MEM_DEVICE;  // (the copies are ordered on ctx's stream, which every restart's stream waits for below)
    }
    // ctxs of restarts 1 .. R - 1: same device, same options, created once and kept
    while ((int)c->batch_children.size() < R - 1) {
        sdpsr_ctx* ch = nullptr;
        sdpsr_opts o = c->opts;
        const int st = sdpsr_create(c->device, c->seed + 0x9E3779B97F4A7C15ULL * (c->batch_children.size() + 1), &o, &ch);
        if (st) return ctx_fail(c, st, "sdpsr_jordan_reduce_batch: could not create the ctx of a restart");
        c->batch_children.push_back(ch);
    }
    Sched sched;
    sched.fibers.reset(new (std::nothrow) Fiber[R]);
    if (!sched.fibers) return ctx_fail(c, SDPSR_OUT_OF_MEMORY, "fibers");
    std::vector<sdpsr_ctx*> ctxs(R);
    for (int i = 0; i < R; ++i) {  // everything that can fail first: no ctx is switched to fiber waits before all stacks exist
        ctxs[i] = i == 0 ? c : c->batch_children[i - 1];
        if (!sched.fibers[i].stack.map(kFiberStack)) return ctx_fail(c, SDPSR_OUT_OF_MEMORY, "fiber stack");
    }
    // the restarts' streams inherit what the caller has ordered before ctx's stream (sdpsr_wait_stream / a shared stream):
    // device-resident inputs produced by the caller's own kernels are complete for every restart, not only for restart 0
    for (int i = 1; i < R; ++i) {
        const int st = sdpsr_wait_stream(ctxs[i], c->stream);
        if (st) return ctx_fail(c, st, "sdpsr_jordan_reduce_batch: could not order a restart's stream behind the ctx's");
    }
    for (int i = 0; i < R; ++i) {
        sdpsr_ctx* ci = ctxs[i];
        if (seeds) sdpsr_set_seed(ci, seeds[i]);
        ci->hint_symmetric_basis = hint;
        status[i] = SDPSR_OK;
        Fiber& f = sched.fibers[i];
        f.sched = &sched;
        f.body = [=]() {
            return jordan_reduce_impl(ci, n, CL, X0L, U, r, atol, epsilon, P_out ? P_out[i] : nullptr, dim_out + i,
                                      iters_out ? iters_out + i : nullptr, nblocks ? nblocks + i : nullptr, sum_sq ? sum_sq + i : nullptr,
                                      sum_s ? sum_s + i : nullptr, blks ? blks[i] : nullptr, blks ? blks_capacity[i] : 0, nullptr, 0, nullptr,
                                      mem_in, mem_out);
        };
        getcontext(&f.uc);
        f.uc.uc_stack.ss_sp = f.stack.sp();
        f.uc.uc_stack.ss_size = f.stack.size();
        f.uc.uc_link = nullptr;
        const uintptr_t pf = reinterpret_cast<uintptr_t>(&f);
        makecontext(&f.uc, reinterpret_cast<void (*)()>(fiber_entry), 2, (unsigned)(pf & 0xFFFFFFFFu), (unsigned)(pf >> 32));
        ci->yield_fn = fiber_yield;
        ci->yield_arg = &f;
    }
    // round robin over the restarts that are still running; a fiber comes back when it has to wait for its stream
    for (int left = R; left > 0;) {
        for (int i = 0; i < R; ++i) {
            Fiber& f = sched.fibers[i];
            if (f.done) continue;
            swapcontext(&sched.main, &f.uc);
            if (f.done) {
                --left;
                status[i] = f.status;
            }
        }
    }
    int first_bad = SDPSR_OK;
    for (int i = 0; i < R; ++i) {
        ctxs[i]->yield_fn = nullptr;
        ctxs[i]->yield_arg = nullptr;
        if (status[i] != SDPSR_OK && first_bad == SDPSR_OK) {
            first_bad = status[i];
            if (i > 0) c->err = "restart " + std::to_string(i) + ": " + ctxs[i]->err;
        }
    }
    return first_bad;
}
}  // namespace sdpsr

extern "C" int sdpsr_jordan_reduce_batch(sdpsr_ctx* c, int32_t R, const uint64_t* seeds, int64_t n, const double* CL, const double* X0L,
                                         const double* U, int64_t r, double atol, double epsilon, uint32_t* const* P_out,
                                         int64_t* dim_out, int32_t* iters_out, int32_t* nblocks, int64_t* sum_sq, int64_t* sum_s,
                                         double* const* blks, const int64_t* blks_capacity, int32_t* status, int mem) {
    CHECK_CTX(c);
    const int hint = c->hint_symmetric_basis;  // one-shot hint of the caller: every restart solves the same problem
    c->hint_symmetric_basis = 0;
    return jordan_reduce_batch_impl(c, R, seeds, n, CL, X0L, U, r, hint, atol, epsilon, P_out, dim_out, iters_out, nblocks, sum_sq, sum_s, blks,
                                    blks_capacity, status, mem, mem);
}

// blkSizes of restart `restart` of the last batch call on ctx (as sdpsr_block_sizes for a single call)
extern "C" int sdpsr_batch_block_sizes(sdpsr_ctx* c, int32_t restart, int32_t* blk_sizes) {
    CHECK_CTX(c);
    if (!blk_sizes || restart < 0 || restart > (int32_t)c->batch_children.size()) return ctx_fail(c, SDPSR_BAD_ARGUMENT, "bad arguments");
    sdpsr_ctx* ci = restart == 0 ? c : c->batch_children[restart - 1];
    const int st = sdpsr_block_sizes(ci, blk_sizes);
    if (st && ci != c) c->err = "restart " + std::to_string(restart) + ": " + ci->err;
    return st;
}
