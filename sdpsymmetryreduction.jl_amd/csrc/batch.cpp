// R independent random restarts of one reduction in ONE call on ONE host thread (include/sdpsr.h:
// sdpsr_jordan_reduce_batch).  A reduction is a chain of ~60 launches with eight host round trips (refinement counters,
// verify verdicts, the module growth's small Gram matrices, the compressed eigenproblem): while the host waits for one
// restart's verdict the GPU runs dry unless another restart's launches are queued.  Every restart gets its own ctx
// (stream, buffers, random streams) and its own FIBER (ucontext) of the calling thread; every host wait of the library
// goes through ctx_sync_stream, which inside a batch polls the stream and, while it is busy, switches to the next
// restart's fiber.  No threads, no locks, no process-global state: the scheduler lives on the caller's stack.
// Reference: the restarts are the "try again" of src/eigen_decomposition.jl:264-270 / src/diagonalize.jl:4-9 and the
// independent draws of the loop (src/partitions.jl:154-185), run side by side instead of one after the other.
#include <ucontext.h>

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
struct Fiber {
    ucontext_t uc;
    std::unique_ptr<char[]> stack;
    Sched* sched = nullptr;
    std::function<int()> body;
    int status = SDPSR_OK;
    bool done = false;
};
struct Sched {
    ucontext_t main;
    std::vector<Fiber> fibers;
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

extern "C" int sdpsr_jordan_reduce_batch(sdpsr_ctx* c, int32_t R, const uint64_t* seeds, int64_t n, const double* CL, const double* X0L,
                                         const double* U, int64_t r, double atol, double epsilon, uint32_t* const* P_out,
                                         int64_t* dim_out, int32_t* iters_out, int32_t* nblocks, int64_t* sum_sq, int64_t* sum_s,
                                         double* const* blks, const int64_t* blks_capacity, int32_t* status, int mem) {
    CHECK_CTX(c);
    if (R < 1 || R > 64 || !dim_out || !status || (blks && !blks_capacity)) return ctx_fail(c, SDPSR_BAD_ARGUMENT, "bad arguments");
    if (c->yield_fn) return ctx_fail(c, SDPSR_BAD_STATE, "sdpsr_jordan_reduce_batch called from inside a batch");
    // ctxs of restarts 1 .. R - 1: same device, same options, created once and kept
    while ((int)c->batch_children.size() < R - 1) {
        sdpsr_ctx* ch = nullptr;
        sdpsr_opts o = c->opts;
        const int st = sdpsr_create(c->device, c->seed + 0x9E3779B97F4A7C15ULL * (c->batch_children.size() + 1), &o, &ch);
        if (st) return ctx_fail(c, st, "sdpsr_jordan_reduce_batch: could not create the ctx of a restart");
        c->batch_children.push_back(ch);
    }
    const int hint = c->hint_symmetric_basis;  // one-shot hint of the caller: every restart solves the same problem
    Sched sched;
    sched.fibers.resize(R);
    std::vector<sdpsr_ctx*> ctxs(R);
    for (int i = 0; i < R; ++i) {  // everything that can fail first: no ctx is switched to fiber waits before all stacks exist
        ctxs[i] = i == 0 ? c : c->batch_children[i - 1];
        sched.fibers[i].stack.reset(new (std::nothrow) char[kFiberStack]);
        if (!sched.fibers[i].stack) return ctx_fail(c, SDPSR_OUT_OF_MEMORY, "fiber stack");
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
            return sdpsr_jordan_reduce(ci, n, CL, X0L, U, r, atol, epsilon, P_out ? P_out[i] : nullptr, dim_out + i,
                                       iters_out ? iters_out + i : nullptr, nblocks ? nblocks + i : nullptr, sum_sq ? sum_sq + i : nullptr,
                                       sum_s ? sum_s + i : nullptr, blks ? blks[i] : nullptr, blks ? blks_capacity[i] : 0, nullptr, 0, nullptr,
                                       mem);
        };
        getcontext(&f.uc);
        f.uc.uc_stack.ss_sp = f.stack.get();
        f.uc.uc_stack.ss_size = kFiberStack;
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
