// Context of libsdpsr_hip.so: lifecycle entry points of include/sdpsr.h, the grow-only device
// buffers, pinned staging and the small helpers every stage uses.
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <functional>
#include <numeric>

#include "host_internal.h"

using namespace sdpsr;
// ---------------------------------------------------------------------------
// ctx helpers
// ---------------------------------------------------------------------------
int ctx_fail(sdpsr_ctx* c, int status, const std::string& msg) {
    if (c) c->err = std::string(sdpsr_status_string(status)) + ": " + msg;
    return status;
}

void* ctx_buf(sdpsr_ctx* c, const char* name, size_t bytes) {
    if (bytes == 0) bytes = 16;
    // ~100 calls per reduction: the usual call finds its buffer large enough -- no std::string is built for that
    auto it = c->bufs.find(std::string_view(name));
    if (it != c->bufs.end() && it->second.bytes >= bytes) return it->second.p;
    DevBuf& b = it != c->bufs.end() ? it->second : c->bufs[std::string(name)];
    if (b.p) {
        ctx_sync_stream(c, c->stream);
        hipFree(b.p);
        b.p = nullptr;
        b.bytes = 0;
    }
    size_t want = bytes + (bytes >> 3);  // slack so that slowly growing sizes do not realloc
    want = (want + 255) & ~size_t(255);
    if (hipMalloc(&b.p, want) != hipSuccess) {
        b.p = nullptr;
        if (hipMalloc(&b.p, bytes) != hipSuccess) {
            b.p = nullptr;
            ctx_fail(c, SDPSR_OUT_OF_MEMORY, std::string("hipMalloc ") + name + " " + std::to_string(bytes));
            return nullptr;
        }
        want = bytes;
    }
    b.bytes = want;
    return b.p;
}


namespace sdpsr {
// Small host<->device transfers go through a growable pinned staging area: a hipMemcpyAsync
// to or from pageable memory costs milliseconds of host time on this stack.
void* ctx_pinned(sdpsr_ctx* c, size_t bytes) {  // shared with eigen.cpp
    if (c->pinned_bytes >= bytes) return c->pinned;
    ctx_sync_stream(c, c->stream);
    if (c->pinned) hipHostFree(c->pinned);
    c->pinned = nullptr;
    c->pinned_bytes = 0;
    size_t want = std::max<size_t>(bytes + bytes / 4, 1 << 16);
    if (hipHostMalloc(&c->pinned, want, hipHostMallocCoherent) != hipSuccess) {  // (coherent: kernels report into it while they run, ctx_wait_word)
        c->pinned = nullptr;
        return nullptr;
    }
    c->pinned_bytes = want;
    return c->pinned;
}

// the side stream of the ctx and the events of its three users (generic-element prefetch of the drivers, first half of
// the back-transformation, projection beside the square): created together, on first use
bool ctx_ensure_side(sdpsr_ctx* c) {
    if (!c->side_stream && hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking) != hipSuccess) {
        c->side_stream = nullptr;
        (void)hipGetLastError();
        return false;
    }
    hipEvent_t* evs[] = {&c->ev_fork, &c->ev_join, &c->ev_bt_fork, &c->ev_bt_join};
    for (hipEvent_t* e : evs)
        if (!*e && hipEventCreateWithFlags(e, hipEventDisableTiming) != hipSuccess) {
            *e = nullptr;
            (void)hipGetLastError();
            return false;
        }
    return true;
}

bool dbg_on() { return getenv("SDPSR_DEBUG") != nullptr; }
void dbg_mark(sdpsr_ctx* c, const char* what) {
    if (!dbg_on()) return;
    const double now = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
    fprintf(stderr, "[sdpsr] +%8.3f ms  %s\n", c && c->dbg_last_ms > 0 ? now - c->dbg_last_ms : 0.0, what);
    if (c) c->dbg_last_ms = now;
}

int d2h_sync(sdpsr_ctx* c, void* host, const void* dev, size_t bytes) {
    void* p = ctx_pinned(c, bytes);
    if (!p) return ctx_fail(c, SDPSR_OUT_OF_MEMORY, "pinned staging");
    HIP_TRY(c, hipMemcpyAsync(p, dev, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, ctx_sync_stream(c, c->stream));
    c->d2h_bytes += bytes;
    memcpy(host, p, bytes);
    return SDPSR_OK;
}
// Upload of a small host array.  Up to H2D_SLOT bytes go through a ring of pinned slots and stay
// stream-ordered (the caller's buffer is free on return, no host wait); the stream is only
// synchronised when the ring wraps, so a slot is never rewritten while its copy is in flight.
constexpr size_t H2D_SLOT = 32 * 1024;
constexpr int H2D_SLOTS = 32;
int h2d_sync(sdpsr_ctx* c, void* dev, const void* host, size_t bytes) {
    c->h2d_bytes += bytes;
    if (bytes <= H2D_SLOT) {
        if (!c->h2d_ring && hipHostMalloc(&c->h2d_ring, H2D_SLOT * H2D_SLOTS, hipHostMallocDefault) != hipSuccess) {
            c->h2d_ring = nullptr;
            return ctx_fail(c, SDPSR_OUT_OF_MEMORY, "pinned upload ring");
        }
        if (c->h2d_ring_next == H2D_SLOTS) {  // every stream that may still read a slot
            HIP_TRY(c, ctx_sync_stream(c, c->stream));
            if (c->side_stream && c->side_stream != c->stream) HIP_TRY(c, ctx_sync_stream(c, c->side_stream));
            if (c->main_shadow && c->main_shadow != c->stream) HIP_TRY(c, ctx_sync_stream(c, c->main_shadow));
            c->h2d_ring_next = 0;
        }
        void* slot = (char*)c->h2d_ring + (size_t)c->h2d_ring_next++ * H2D_SLOT;
        memcpy(slot, host, bytes);
        HIP_TRY(c, hipMemcpyAsync(dev, slot, bytes, hipMemcpyHostToDevice, c->stream));
        return SDPSR_OK;
    }
    void* p = ctx_pinned(c, bytes);
    if (!p) return ctx_fail(c, SDPSR_OUT_OF_MEMORY, "pinned staging");
    memcpy(p, host, bytes);
    HIP_TRY(c, hipMemcpyAsync(dev, p, bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, ctx_sync_stream(c, c->stream));
    return SDPSR_OK;
}
// carries the rounding rule into the kernels (sdpsr_hash.h: negative = truncate like the reference)
double round_scale(const sdpsr_ctx* c, double atol) {
    const double sc = std::pow(10.0, std::floor(-std::log10(atol)));
    return c->opts.round_mode == SDPSR_ROUND_TRUNC ? -sc : sc;
}

// sdpsr_opts.label_bits: the reference's label type T = UInt8/16/32 cannot hold `value`
// (T(l + 1) at src/partitions.jl:29, the pair code at :63 -> InexactError)
bool label_overflows(const sdpsr_ctx* c, uint64_t value) {
    const int b = c->opts.label_bits;
    return b > 0 && b < 64 && value > ((uint64_t(1) << b) - 1);
}
int label_overflow_fail(sdpsr_ctx* c, const char* where, uint64_t value) {
    return ctx_fail(c, SDPSR_LABEL_OVERFLOW, std::string(where) + ": " + std::to_string(value) + " does not fit the " +
                                                 std::to_string(c->opts.label_bits) + "-bit label type (InexactError in the reference)");
}

uint64_t next_key(sdpsr_ctx* c) { return sdpsr_stream_key(c->seed, c->stream_counter++); }

int check_len(sdpsr_ctx* c, int64_t len) {
    if (len < 1 || len >= (int64_t)0xFFFFFFF0ll)
        return ctx_fail(c, SDPSR_BAD_ARGUMENT, "len out of range [1, 2^32-16)");
    return SDPSR_OK;
}

}  // namespace sdpsr

// ---------------------------------------------------------------------------
// lifecycle
// ---------------------------------------------------------------------------
extern "C" {

const char* sdpsr_status_string(int s) {
    switch (s) {
        case SDPSR_OK: return "OK";
        case SDPSR_INVALID_DECOMPOSITION_FIELD: return "INVALID_DECOMPOSITION_FIELD";
        case SDPSR_NUMERICAL_INCONSISTENCY: return "NUMERICAL_INCONSISTENCY";
        case SDPSR_DIMENSION_MISMATCH: return "DIMENSION_MISMATCH";
        case SDPSR_LABEL_OVERFLOW: return "LABEL_OVERFLOW";
        case SDPSR_BAD_ARGUMENT: return "BAD_ARGUMENT";
        case SDPSR_HIP_ERROR: return "HIP_ERROR";
        case SDPSR_SOLVER_ERROR: return "SOLVER_ERROR";
        case SDPSR_OUT_OF_MEMORY: return "OUT_OF_MEMORY";
        case SDPSR_NOT_CONVERGED: return "NOT_CONVERGED";
        case SDPSR_BAD_STATE: return "BAD_STATE";
    }
    return "UNKNOWN";
}

int sdpsr_version(void) { return SDPSR_VERSION_MAJOR * 1000 + SDPSR_VERSION_MINOR; }

int sdpsr_create(int device_id, uint64_t seed, const sdpsr_opts* opts, sdpsr_ctx** out) {
    if (!out) return SDPSR_BAD_ARGUMENT;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return SDPSR_HIP_ERROR;
    if (device_id < 0 || device_id >= ndev) return SDPSR_BAD_ARGUMENT;
    sdpsr_ctx* c = new sdpsr_ctx();
    c->device = device_id;
    c->seed = seed;
    if (opts) {
        size_t sz = std::min<size_t>(opts->struct_size ? opts->struct_size : sizeof(sdpsr_opts), sizeof(sdpsr_opts));
        memcpy(&c->opts, opts, sz);
    }
    c->opts.struct_size = sizeof(sdpsr_opts);
    if (c->opts.square_mode == SDPSR_SQUARE_AUTO) c->opts.square_mode = SDPSR_SQUARE_I8;
    if (c->opts.channels <= 0) {
        // default: 2 channels and one confirm round -- a false stop needs (confirm_rounds + 1) consecutive
        // squares that miss every needed split, each with probability <= (2/256)^channels: the same
        // (2/256)^4 as 4 channels without a confirm round, at 2 (I + 1) instead of 4 I channel squares
        c->opts.channels = 2;
        if (c->opts.confirm_rounds < 1 && c->opts.square_mode != SDPSR_SQUARE_F64) c->opts.confirm_rounds = 1;
    }
    if (c->opts.channels > 8) c->opts.channels = 8;
    if (c->opts.confirm_rounds < 0) c->opts.confirm_rounds = 0;
    if (c->opts.round_mode != SDPSR_ROUND_NEAREST && c->opts.round_mode != SDPSR_ROUND_TRUNC) {
        delete c;
        return SDPSR_BAD_ARGUMENT;
    }
    if (c->opts.label_bits != 0 && c->opts.label_bits != 8 && c->opts.label_bits != 16 && c->opts.label_bits != 32) {
        delete c;
        return SDPSR_BAD_ARGUMENT;
    }
    if (c->opts.max_iters <= 0) c->opts.max_iters = 10000;
    DeviceGuard dg(device_id);
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return SDPSR_HIP_ERROR;
    }
    c->own_stream = true;
    if (hipDeviceGetAttribute(&c->num_cus, hipDeviceAttributeMultiprocessorCount, device_id) != hipSuccess || c->num_cus < 1)
        c->num_cus = 256;
    // per-device kernel attributes (dynamic LDS above 64 KiB); cheap and idempotent
    bool attrs_ok = true;
    attrs_ok &= gemm_set_device_attributes();
    attrs_ok &= gemm_sym_set_device_attributes();
    attrs_ok &= refine_bucket_set_device_attributes();
    attrs_ok &= blockdiag_set_device_attributes();
    attrs_ok &= module_set_device_attributes();
    attrs_ok &= partition_set_device_attributes();
    attrs_ok &= sytrd_set_device_attributes();
    attrs_ok &= small_syev_set_device_attributes();
    attrs_ok &= stedc_set_device_attributes();
    attrs_ok &= batched_set_device_attributes();
    attrs_ok &= backtransform_set_device_attributes();
    attrs_ok &= complex_set_device_attributes();
    if (hipGetLastError() != hipSuccess || !attrs_ok) {  // a failed LDS opt-in would surface later as an opaque launch error
        hipStreamDestroy(c->stream);
        delete c;
        return SDPSR_HIP_ERROR;
    }
    c->pinned_bytes = 1 << 16;
    if (hipHostMalloc((void**)&c->pinned_small, 256, hipHostMallocDefault) != hipSuccess) c->pinned_small = nullptr;
    if (hipHostMalloc(&c->pinned, c->pinned_bytes, hipHostMallocCoherent) != hipSuccess) {
        hipStreamDestroy(c->stream);
        delete c;
        return SDPSR_OUT_OF_MEMORY;
    }
    *out = c;
    return SDPSR_OK;
}

void sdpsr_destroy(sdpsr_ctx* c) {
    if (!c) return;
    for (sdpsr_ctx* ch : c->batch_children) sdpsr_destroy(ch);
    c->batch_children.clear();
    DeviceGuard dg(c->device);
    ctx_sync_stream(c, c->stream);
    destroy_handle(c);
    sytrd_graph_cache_destroy(c->sytrd_graphs);
    for (auto& kv : c->bufs)
        if (kv.second.p) hipFree(kv.second.p);
    if (c->pinned) hipHostFree(c->pinned);
    if (c->h2d_ring) hipHostFree(c->h2d_ring);
    if (c->pinned_small) hipHostFree(c->pinned_small);
    if (c->ev_fork) hipEventDestroy(c->ev_fork);
    if (c->ev_join) hipEventDestroy(c->ev_join);
    if (c->ev_bt_fork) hipEventDestroy(c->ev_bt_fork);
    if (c->ev_bt_join) hipEventDestroy(c->ev_bt_join);
    if (c->ev_wait) hipEventDestroy(c->ev_wait);
    if (c->side_stream) hipStreamDestroy(c->side_stream);
    if (c->own_stream && c->stream) hipStreamDestroy(c->stream);
    delete c;
}

const char* sdpsr_last_error(const sdpsr_ctx* c) { return c ? c->err.c_str() : "null ctx"; }

int sdpsr_set_stream(sdpsr_ctx* c, void* hip_stream) {
    CHECK_CTX(c);
    ctx_sync_stream(c, c->stream);
    if (c->own_stream && c->stream) hipStreamDestroy(c->stream);
    if (hip_stream) {
        c->stream = (hipStream_t)hip_stream;
        c->own_stream = false;
    } else {
        HIP_TRY(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        c->own_stream = true;
    }
    return SDPSR_OK;
}

int sdpsr_synchronize(sdpsr_ctx* c) {
    CHECK_CTX(c);
    HIP_TRY(c, ctx_sync_stream(c, c->stream));
    return SDPSR_OK;
}

int sdpsr_wait_stream(sdpsr_ctx* c, void* hip_stream) {
    CHECK_CTX(c);
    if ((hipStream_t)hip_stream == c->stream) return SDPSR_OK;
    if (!c->ev_wait) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_wait, hipEventDisableTiming));
    HIP_TRY(c, hipEventRecord(c->ev_wait, (hipStream_t)hip_stream));
    HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_wait, 0));
    return SDPSR_OK;
}

int sdpsr_hint_symmetric_basis(sdpsr_ctx* c, int yes) {
    if (!c) return SDPSR_BAD_ARGUMENT;
    c->hint_symmetric_basis = yes & 3;
    return SDPSR_OK;
}

int sdpsr_transfer_bytes(sdpsr_ctx* c, uint64_t* h2d, uint64_t* d2h) {
    if (!c) return SDPSR_BAD_ARGUMENT;
    uint64_t a = c->h2d_bytes, b = c->d2h_bytes;
    for (const sdpsr_ctx* ch : c->batch_children) {
        a += ch->h2d_bytes;
        b += ch->d2h_bytes;
    }
    if (h2d) *h2d = a;
    if (d2h) *d2h = b;
    return SDPSR_OK;
}

int sdpsr_set_seed(sdpsr_ctx* c, uint64_t seed) {
    if (!c) return SDPSR_BAD_ARGUMENT;
    c->seed = seed;
    c->stream_counter = 0;
    return SDPSR_OK;
}

int sdpsr_dimension_trajectory(sdpsr_ctx* c, int64_t* dims, int32_t capacity, int32_t* count) {
    if (!c || capacity < 0 || (capacity > 0 && !dims)) return SDPSR_BAD_ARGUMENT;
    const int32_t have = (int32_t)std::min<size_t>(c->adm_dims.size(), 0x7FFFFFFF);
    if (count) *count = have;
    for (int32_t i = 0; i < have && i < capacity; ++i) dims[i] = c->adm_dims[i];
    return SDPSR_OK;
}

}  // extern "C"
