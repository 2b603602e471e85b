// AbstractPartition primitives of include/sdpsr.h (Partition ctor, refine!, fill!, randomize!,
// _clamp_round!, projection) and the N x N squares, plus the canonical refinement of a signature
// source that the loop shares with them.  Reference: src/partitions.jl:24-75, src/utils.jl:34-66.
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

namespace sdpsr {

// ---- canonical refinement of a signature array --------------------------------
// sym_n > 0: the new labels (an sym_n x sym_n matrix) are also checked for symmetry on the device
// and the verdict rides back with the counters (same synchronisation): *sym_out = 1 if symmetric.
// src: where the signatures come from (sdpsr_internal.h: SigSource).  A computed source is
// evaluated inside the insert kernel; it is written out as an array (src.sig: len entries of
// scratch) only for the sort path or when the insert kernel has no instance for it.
int refine_signatures(sdpsr_ctx* c, int64_t len, const SigSource& src_in, uint32_t* labels,
                      int64_t* nparts, int64_t sym_n, uint32_t* symflag_dev, int* sym_out, bool early) {
    SigSource src = src_in;
    auto materialize = [&]() -> bool {
        if (src.kind == SIG_ARRAY) return true;
        if (!src.sig) return false;
        launch_sig_materialize(c->stream, len, src, src.sig);
        src.kind = SIG_ARRAY;
        return true;
    };
    const bool no_fuse = (c->opts.flags & SDPSR_FLAG_REFINE_NO_FUSE) != 0;  // always through the array
    if ((no_fuse || !sig_source_fusable(src)) && !materialize())
        return ctx_fail(c, SDPSR_BAD_ARGUMENT, "refine: signature source needs scratch");
    // slots of the insert pass: in place for an array source (nothing reads the old labels), a
    // scratch array for a computed source (it may read the old labels from `labels` on a repeated pass)
    uint32_t* slot = (src.kind == SIG_ARRAY) ? labels : (uint32_t*)ctx_buf(c, "ref_slots", (size_t)len * 4);
    if (!slot) return SDPSR_OUT_OF_MEMORY;
    const int full = std::max(12, ceil_log2((uint64_t)len * 2));
    int log2cap = std::min(full, std::max(12, c->table_log2_hint));
    const int64_t rb = (int64_t)refine_block_entries();
    const int64_t nblk = (len + rb - 1) / rb;
    bool mispredicted = false, sampled = false, cub_fallback = false;
    c->first_idx_labels = nullptr;  // whatever happens below, the old representatives no longer describe `labels`
    // Which relabel?  Few classes: a hash table (LDS level + a global one that stays in L2).  Many classes (problems
    // without symmetry: ~len / 2 distinct signatures): a table that large is one global atomic per entry into memory no
    // cache holds -- the bucketed grouping of kernels_refine_bucket.hip streams instead.  The choice is made BEFORE a
    // pass runs to its end: from the class count the previous refinement of this call ended with (table_log2_hint), and,
    // when a table built on that prediction overflows (the pass stops within its first chunks then), from a SAMPLE of
    // the signatures: 65 536 stratified entries through a small table give the distinct count d_s and the numbers f1,
    // f2 of signatures seen once / twice; Chao's estimate d_s + f1 (f1 - 1) / (2 (f2 + 1)) of the total is within a few
    // percent where it matters (classes of ~2 entries: f1 ~ m, f2 ~ m^2 / len) and errs low only for partitions whose
    // small classes hide behind huge ones -- then the table sized from it overflows once more and the grouping runs.
    // (Rounds 3-4 walked a ladder of 2^12 -> 2^16 -> 2^20 slots: 12.7 ms of failing passes on a fresh N = 4096 problem
    // without symmetry.)  refine_path 2 / 3 force the hipCUB sort / the bucketed grouping at any size (comparison, tests)
    const bool forced_relabel = c->opts.refine_path == 2 || c->opts.refine_path == 3;
    const bool sort_ok = ((len >= (int64_t(1) << 18) && c->opts.refine_path != 1) || forced_relabel) && len < (int64_t(1) << 31);
    // the grouping's time does not depend on the class count (0.4 ms at 16.7 M entries); the table path's does
    const int64_t group_from = std::max<int64_t>(int64_t(1) << 16, len >> 7);
    // (table_log2_hint = ceil(log2(8 d + 1)) of the previous refinement's class count d)
    bool use_sort = sort_ok && (forced_relabel || (c->table_log2_hint >= 4 && (int64_t(1) << (c->table_log2_hint - 4)) >= group_from));
    for (;;) {
        if (use_sort) {
            // the hand-written bucketed grouping (kernels_refine_bucket.hip); hipCUB's radix sort behind refine_path = 2
            // and as the fallback for signatures the grouping reports it cannot resolve
            const bool cub = c->opts.refine_path == 2 || cub_fallback;
            const size_t wsb = cub ? refine_sorted_workspace_bytes(len) : refine_bucketed_workspace_bytes(len);
            void* wsp = ctx_buf(c, "ref_sort_ws", wsb);
            uint32_t* counters = (uint32_t*)ctx_buf(c, "ref_counters", refine_counters_bytes());
            uint32_t* firsts = (uint32_t*)ctx_buf(c, "ref_first", (size_t)refine_first_cap() * 4);
            uint32_t* h = (uint32_t*)ctx_pinned(c, 64);
            if (!wsp || !counters || !firsts || !h) return SDPSR_OUT_OF_MEMORY;
            if (!materialize()) return ctx_fail(c, SDPSR_BAD_ARGUMENT, "refine: signature source needs scratch");
            const uint32_t seq = ++c->report_seq ? c->report_seq : ++c->report_seq;  // (never 0: the pinned words start as zeros)
            if (cub ? !launch_refine_sorted(c->stream, len, src.sig, labels, wsp, wsb, counters)
                    : !launch_refine_bucketed(c->stream, len, src.sig, labels, wsp, wsb, counters, firsts, refine_first_cap(), h, seq))
                return ctx_fail(c, SDPSR_HIP_ERROR, "sorted / bucketed refinement failed");
            if (cub) HIP_TRY(c, hipMemcpyAsync(h, counters, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));  // (the grouping's label pass stores them itself)
            if (sym_n > 0 && symflag_dev) {
                launch_check_symmetric(c->stream, sym_n, labels, symflag_dev);
                HIP_TRY(c, hipMemcpyAsync(h + 8, symflag_dev, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
            }
            // (early: the caller only needs the label pass's report and waits for the stream itself before it returns)
            if (early && !cub && !(sym_n > 0 && symflag_dev)) HIP_TRY(c, ctx_wait_word(c, c->stream, h + 3, seq));
            else HIP_TRY(c, ctx_sync_stream(c, c->stream));
            HIP_TRY(c, hipGetLastError());
            if (h[1]) {
                // a bucket the grouping could not resolve (signatures that do not spread over its sub-passes): the radix sort
                // has no such case.  (The grouping reads the signature ARRAY, never the old labels: nothing to restore.)
                if (cub) return ctx_fail(c, SDPSR_HIP_ERROR, "sorted refinement reported a failure");
                cub_fallback = true;
                continue;
            }
            if (sym_n > 0 && symflag_dev && sym_out) *sym_out = h[8] ? 0 : 1;
            *nparts = h[2];
            c->table_log2_hint = std::min(full, std::max(12, ceil_log2((uint64_t)h[2] * 8 + 1)));
            if (!cub && h[2] <= refine_first_cap()) c->first_idx_labels = labels;  // "ref_first" describes these labels
            return SDPSR_OK;
        }
        // Up to ~5000 classes (array source and the loop's computed sources): one workgroup per CU with every signature in LDS (refine_insert_mid_kernel; built for
        // the regime of 500 .. 5000 classes, it also beats the 2048-slot workgroup tables below that: 0.096 against 0.135 ms at 34
        // classes, N = 4096).  The global table then sees each signature once per workgroup, not once per entry: a quarter of
        // the slots do (half full at most), and the label pass gathers from 32 KB of labels instead of 128 KB.
        // (refine_path 4 / 6: never / without the workgroups that go first)
        const bool mid = refine_mid_supports(src) && log2cap <= 16 && len >= (int64_t(1) << 20) && c->opts.refine_path != 4;
        const int tab_log2 = (mid && !mispredicted && !sampled && log2cap >= 14) ? log2cap - 2 : log2cap;
        const size_t cap = size_t(1) << tab_log2;
        RefineWs ws;
        ws.tab = (RefSlot*)ctx_buf(c, "ref_tab", cap * sizeof(RefSlot));
        ws.tab_lab = (uint32_t*)ctx_buf(c, "ref_tab_lab", cap * 4);
        ws.blk_cnt = (uint32_t*)ctx_buf(c, "ref_blk_cnt", (nblk + 1) * 4);
        ws.counters = (uint32_t*)ctx_buf(c, "ref_counters", refine_counters_bytes());
        ws.first_idx = (uint32_t*)ctx_buf(c, "ref_first", (size_t)refine_first_cap() * 4);
        if (!ws.tab || !ws.tab_lab || !ws.blk_cnt || !ws.counters || !ws.first_idx)
            return SDPSR_OUT_OF_MEMORY;
        ws.log2cap = tab_log2;
        if (log2cap > 12 || mispredicted || sampled) {  // (a table of 2^12 slots holds at most 3072 classes: mostly the one-workgroup ranking)
            ws.rank_ws_bytes = refine_rank_slots_workspace_bytes(len);
            ws.rank_ws = ctx_buf(c, "ref_rank_ws", ws.rank_ws_bytes);
            if (!ws.rank_ws) return SDPSR_OUT_OF_MEMORY;
        }
        ws.mid = mid ? (c->opts.refine_path == 6 ? 3 : 1) : 0;
        ws.insert_wgs_per_cu = c->opts.insert_wgs_per_cu;
        ws.nblk = (int)nblk;
        // hint 12 <=> last dim <= 512; a table grown after an overflow in this call holds more than 0.75 * 2^12 classes
        ws.expect_small = (!mispredicted && !sampled && c->table_log2_hint <= 12 && log2cap <= 12) ? 1 : 0;
        const bool sym_fused = sym_n > 0 && sym_n * sym_n == len;  // verdict in counters[3], same read-back
        uint32_t* h = (uint32_t*)ctx_pinned(c, 64);
        if (!h) return ctx_fail(c, SDPSR_OUT_OF_MEMORY, "pinned staging");
        ws.host_counters = sym_fused ? nullptr : h;  // the plain label pass stores the counters into the pinned buffer itself
        ws.host_seq = ++c->report_seq ? c->report_seq : ++c->report_seq;
        launch_refine(c->stream, len, src, slot, labels, ws, sym_fused ? sym_n : 0);
        if (!ws.host_counters) HIP_TRY(c, hipMemcpyAsync(h, ws.counters, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        if (sym_n > 0 && symflag_dev && !sym_fused) {
            launch_check_symmetric(c->stream, sym_n, labels, symflag_dev);  // flag = 1 if NOT symmetric
            HIP_TRY(c, hipMemcpyAsync(h + 8, symflag_dev, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        }
        if (early && ws.host_counters && !(sym_n > 0 && symflag_dev)) HIP_TRY(c, ctx_wait_word(c, c->stream, h + 3, ws.host_seq));
        else HIP_TRY(c, ctx_sync_stream(c, c->stream));
        if (sym_fused) h[8] = h[3];
        if (sym_n > 0 && (symflag_dev || sym_fused) && sym_out) *sym_out = h[8] ? 0 : 1;
        HIP_TRY(c, hipGetLastError());
        if (!h[1] && ws.expect_small && h[0] > refine_small_k()) {  // more classes than predicted: general ranking
            mispredicted = true;
            // the count is exact now: the size the hint would have asked for (a 2^12 table three quarters full probes long chains)
            log2cap = std::min(full, std::max(log2cap, ceil_log2((uint64_t)h[0] * 8 + 1)));
            continue;
        }
        if (h[1]) {  // table too small for this many classes
            if (log2cap >= full) return ctx_fail(c, SDPSR_HIP_ERROR, "refine hash table overflow at full size");
            if (len < (int64_t(1) << 16)) {  // small inputs: the full-size table is a few hundred KB
                log2cap = full;
                continue;
            }
            if (sampled) {  // the estimate was too low (small classes hidden behind huge ones)
                if (sort_ok) use_sort = true;
                else log2cap = full;
                continue;
            }
            // how many classes are there?  (A computed source is written out as an array first: the sample reads it, and so
            // does whichever relabel follows -- the grouping needs it anyway, the table path takes it as its source.)
            if (!materialize()) {
                log2cap = std::min(full, log2cap + 6);
                continue;
            }
            slot = labels;  // array source: the slots go in place, nothing reads the old labels any more
            void* sws = ctx_buf(c, "ref_sample", refine_sample_workspace_bytes());
            uint32_t* hs = (uint32_t*)ctx_pinned(c, 1024);
            if (!sws || !hs) return SDPSR_OUT_OF_MEMORY;
            hs += 224;  // its own pinned words (counters at 0, verify verdict at 128, basis verdict at 192)
            const int64_t m = launch_refine_sample(c->stream, len, src.sig, sws, hs);
            if (m <= 0) return ctx_fail(c, SDPSR_HIP_ERROR, "refine: sample launch failed");
            HIP_TRY(c, ctx_sync_stream(c, c->stream));
            HIP_TRY(c, hipGetLastError());
            sampled = true;
            const double nz = hs[0], ds = hs[1], f1 = hs[2], f2 = hs[3];
            double est = ds;
            if (m < len) {
                est = ds + f1 * std::max(f1 - 1.0, 0.0) / (2.0 * (f2 + 1.0));
                est = std::min(est, nz * (double)len / (double)m + 1.0);  // no more classes than non-zero entries
            }
            if (dbg_on()) fprintf(stderr, "[sdpsr] refine: sample of %lld: %g non-zero, %g distinct, %g once, %g twice -> ~%.3g classes\n",
                                  (long long)m, nz, ds, f1, f2, est);
            if (sort_ok && est >= (double)group_from) {
                use_sort = true;
                continue;
            }
            // a table with <= 1/3 of its slots taken by the estimate (the flag goes up at 3/4), at least four times the last one
            log2cap = std::min(full, std::max(log2cap + 2, ceil_log2((uint64_t)(est * 3.0) + 1)));
            continue;
        }
        *nparts = h[2];
        c->table_log2_hint = std::min(full, std::max(12, ceil_log2((uint64_t)h[2] * 8 + 1)));
        if (h[2] <= refine_first_cap()) c->first_idx_labels = labels;  // "ref_first" describes these labels
        return SDPSR_OK;
    }
}

int refine_signatures(sdpsr_ctx* c, int64_t len, const uint64_t* sig, uint32_t* labels,
                      int64_t* nparts, int64_t sym_n, uint32_t* symflag_dev, int* sym_out, bool early) {
    SigSource src;
    src.kind = SIG_ARRAY;
    src.sig = const_cast<uint64_t*>(sig);
    return refine_signatures(c, len, src, labels, nparts, sym_n, symflag_dev, sym_out, early);
}

}  // namespace sdpsr

extern "C" {

// ---------------------------------------------------------------------------
// primitives
// ---------------------------------------------------------------------------
int sdpsr_partition_from_f64(sdpsr_ctx* c, int64_t len, const double* M, uint32_t* labels,
                             int64_t* nparts, int mem) {
    CHECK_CTX(c);
    if (!M || !labels || !nparts) return ctx_fail(c, SDPSR_BAD_ARGUMENT, "null pointer");
    int st = check_len(c, len);
    if (st) return st;
    const double* dM = in_dev(c, "prim_in_a", M, len, mem, &st);
    uint32_t* dL = out_dev(c, "prim_out", labels, len, mem, &st);
    uint64_t* sig = (uint64_t*)ctx_buf(c, "sig", len * 8);
    if (st || !sig) return st ? st : SDPSR_OUT_OF_MEMORY;
    launch_sig_f64(c->stream, len, nullptr, dM, sig);
    st = refine_signatures(c, len, sig, dL, nparts);
    if (st) return st;
    if (label_overflows(c, (uint64_t)*nparts)) return label_overflow_fail(c, "Partition{T}(M)", (uint64_t)*nparts);
    return out_finish(c, labels, dL, len, mem);
}

int sdpsr_partition_from_u32(sdpsr_ctx* c, int64_t len, const uint32_t* in, uint32_t* labels,
                             int64_t* nparts, int mem) {
    CHECK_CTX(c);
    if (!in || !labels || !nparts) return ctx_fail(c, SDPSR_BAD_ARGUMENT, "null pointer");
    int st = check_len(c, len);
    if (st) return st;
    const uint32_t* dI = in_dev(c, "prim_in_a", in, len, mem, &st);
    uint32_t* dL = out_dev(c, "prim_out", labels, len, mem, &st);
    uint64_t* sig = (uint64_t*)ctx_buf(c, "sig", len * 8);
    if (st || !sig) return st ? st : SDPSR_OUT_OF_MEMORY;
    launch_sig_u32(c->stream, len, nullptr, dI, sig);
    st = refine_signatures(c, len, sig, dL, nparts);
    if (st) return st;
    if (label_overflows(c, (uint64_t)*nparts)) return label_overflow_fail(c, "Partition{T}(M)", (uint64_t)*nparts);
    return out_finish(c, labels, dL, len, mem);
}

int sdpsr_partition_from_u64(sdpsr_ctx* c, int64_t len, const uint64_t* in, uint32_t* labels, int64_t* nparts, int mem) {
    CHECK_CTX(c);
    if (!in || !labels || !nparts) return ctx_fail(c, SDPSR_BAD_ARGUMENT, "null pointer");
    int st = check_len(c, len);
    if (st) return st;
    const uint64_t* dI = in_dev(c, "prim_in_a", in, len, mem, &st);
    uint32_t* dL = out_dev(c, "prim_out", labels, len, mem, &st);
    uint64_t* sig = (uint64_t*)ctx_buf(c, "sig", len * 8);
    if (st || !sig) return st ? st : SDPSR_OUT_OF_MEMORY;
    launch_sig_u64(c->stream, len, dI, sig);
    st = refine_signatures(c, len, sig, dL, nparts);
    if (st) return st;
    if (label_overflows(c, (uint64_t)*nparts)) return label_overflow_fail(c, "Partition{T}(M)", (uint64_t)*nparts);
    return out_finish(c, labels, dL, len, mem);
}

int sdpsr_refine(sdpsr_ctx* c, int64_t len, uint32_t* p1, int64_t* d1, const uint32_t* p2,
                 int64_t d2, int mem) {
    CHECK_CTX(c);
    (void)d2;
    if (!p1 || !p2 || !d1) return ctx_fail(c, SDPSR_BAD_ARGUMENT, "null pointer");
    int st = check_len(c, len);
    if (st) return st;
    const uint32_t* d1in = in_dev(c, "prim_in_a", (const uint32_t*)p1, len, mem, &st);
    const uint32_t* d2in = in_dev(c, "prim_in_b", p2, len, mem, &st);
    uint32_t* dL = (mem == SDPSR_MEM_DEVICE) ? p1 : (uint32_t*)ctx_buf(c, "prim_out", len * 4);
    uint64_t* sig = (uint64_t*)ctx_buf(c, "sig", len * 8);
    if (st || !sig || !dL) return st ? st : SDPSR_OUT_OF_MEMORY;
    if (c->opts.label_bits) {
        // P1.matrix .+= P2.matrix .* (dim(P1) + 1) is stored in P1's label type before the renumbering
        // (src/partitions.jl:63): the largest pair code decides, exactly as in the reference
        uint64_t* dmax = (uint64_t*)ctx_buf(c, "prim_flag", 64);
        if (!dmax) return SDPSR_OUT_OF_MEMORY;
        HIP_TRY(c, hipMemsetAsync(dmax, 0, 8, c->stream));
        launch_max_pair_code(c->stream, len, d1in, d2in, (uint64_t)*d1, dmax);
        uint64_t hmax = 0;
        st = d2h_sync(c, &hmax, dmax, 8);
        if (st) return st;
        if (label_overflows(c, hmax)) return label_overflow_fail(c, "refine!: pair code l1 + l2 * (dim(P1) + 1)", hmax);
    }
    launch_sig_u32(c->stream, len, d1in, d2in, sig);
    st = refine_signatures(c, len, sig, dL, d1);
    if (st) return st;
    return out_finish(c, p1, dL, len, mem);
}

int sdpsr_partition_checksum(sdpsr_ctx* c, int64_t len, const uint32_t* labels, uint64_t* out, int mem) {
    CHECK_CTX(c);
    if (!labels || !out) return ctx_fail(c, SDPSR_BAD_ARGUMENT, "null pointer");
    int st = check_len(c, len);
    if (st) return st;
    const uint32_t* dL = in_dev(c, "chk_labels", labels, (size_t)len, mem, &st);
    uint64_t* scratch = (uint64_t*)ctx_buf(c, "chk_scratch", (size_t)(2 * 2048 + 2) * 8);
    if (st || !scratch) return st ? st : SDPSR_OUT_OF_MEMORY;
    launch_labels_checksum(c->stream, len, dL, scratch, scratch + 2 * 2048);
    HIP_TRY(c, hipGetLastError());
    return d2h_sync(c, out, scratch + 2 * 2048, 16);
}

int sdpsr_fill(sdpsr_ctx* c, int64_t len, const uint32_t* labels, const double* values, int64_t d,
               double* M, int mem) {
    CHECK_CTX(c);
    if (!labels || !M || (d > 0 && !values)) return ctx_fail(c, SDPSR_BAD_ARGUMENT, "null pointer");
    int st = check_len(c, len);
    if (st) return st;
    const uint32_t* dL = in_dev(c, "prim_in_a", labels, len, mem, &st);
    const double* dV = in_dev(c, "prim_in_b", values, (size_t)std::max<int64_t>(d, 1), mem, &st);
    double* dM = out_dev(c, "prim_out", M, len, mem, &st);
    if (st) return st;
    // labels beyond d never index `values` (the kernel writes 0.0 there and raises the flag)
    uint32_t* flag = (uint32_t*)ctx_buf(c, "prim_flag", 64);
    if (!flag || !c->pinned_small) return SDPSR_OUT_OF_MEMORY;
    HIP_TRY(c, hipMemsetAsync(flag, 0, 4, c->stream));
    launch_fill_f64(c->stream, len, dL, dV, d, dM, flag);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(c->pinned_small, flag, 4, hipMemcpyDeviceToHost, c->stream));
    st = out_finish(c, M, dM, len, mem);
    if (st) return st;
    if (c->pinned_small[0]) return ctx_fail(c, SDPSR_BAD_ARGUMENT, "fill: a label exceeds d = length(values)");
    return SDPSR_OK;
}

int sdpsr_randomize(sdpsr_ctx* c, int64_t len, const uint32_t* labels, double* M, int mem) {
    CHECK_CTX(c);
    if (!labels || !M) return ctx_fail(c, SDPSR_BAD_ARGUMENT, "null pointer");
    int st = check_len(c, len);
    if (st) return st;
    const uint32_t* dL = in_dev(c, "prim_in_a", labels, len, mem, &st);
    double* dM = out_dev(c, "prim_out", M, len, mem, &st);
    if (st) return st;
    launch_randomize_f64(c->stream, len, dL, next_key(c), dM);
    HIP_TRY(c, hipGetLastError());
    return out_finish(c, M, dM, len, mem);
}

int sdpsr_clamp_round(sdpsr_ctx* c, int64_t len, double* a, double atol, int mem) {
    CHECK_CTX(c);
    if (!a || !(atol > 0)) return ctx_fail(c, SDPSR_BAD_ARGUMENT, "null pointer or atol <= 0");
    int st = check_len(c, len);
    if (st) return st;
    double* dA = (mem == SDPSR_MEM_DEVICE) ? a : (double*)in_dev(c, "prim_in_a", (const double*)a, len, mem, &st);
    if (st) return st;
    launch_clamp_round(c->stream, len, dA, atol, round_scale(c, atol));
    HIP_TRY(c, hipGetLastError());
    return out_finish(c, a, dA, len, mem);
}

int sdpsr_project_out(sdpsr_ctx* c, int64_t len, double* x, const double* U, int64_t r, int mem) {
    CHECK_CTX(c);
    if (!x || (r > 0 && !U) || r < 0) return ctx_fail(c, SDPSR_BAD_ARGUMENT, "bad arguments");
    int st = check_len(c, len);
    if (st) return st;
    double* dX = (mem == SDPSR_MEM_DEVICE) ? x : (double*)in_dev(c, "prim_in_a", (const double*)x, len, mem, &st);
    const double* dU = in_dev(c, "prim_in_b", U, (size_t)len * std::max<int64_t>(r, 1), mem, &st);
    const int nblk = 2048;
    double* partial = (double*)ctx_buf(c, "proj_partial", (size_t)std::max<int64_t>(r, 1) * nblk * 8);
    double* coef = (double*)ctx_buf(c, "proj_coef", (size_t)std::max<int64_t>(r, 1) * 8);
    if (st || !partial || !coef) return st ? st : SDPSR_OUT_OF_MEMORY;
    launch_proj_coef(c->stream, len, r, dU, nullptr, 0, dX, partial, nblk, coef);
    launch_proj_apply(c->stream, len, r, dU, nullptr, 0, dX, coef, 0, 1, 0, dX, nullptr);
    HIP_TRY(c, hipGetLastError());
    return out_finish(c, x, dX, len, mem);
}

// ---------------------------------------------------------------------------
// squares / products
// ---------------------------------------------------------------------------
}  // extern "C"

template <typename TI, typename TO, typename F>
static int square_generic(sdpsr_ctx* c, int64_t n, const TI* X, TO* X2, int mem, F launch) {
    if (!X || !X2 || n < 1) return ctx_fail(c, SDPSR_BAD_ARGUMENT, "bad arguments");
    int st = SDPSR_OK;
    const int64_t ld = round_up(n, 128);
    const TI* dX = in_dev(c, "sq_in", X, (size_t)n * n, mem, &st);
    TI* Xp = (TI*)ctx_buf(c, "sq_xpad", (size_t)ld * ld * sizeof(TI));
    TO* Cp = (TO*)ctx_buf(c, "sq_cpad", (size_t)ld * ld * sizeof(TO));
    TO* dC = out_dev(c, "sq_out", X2, (size_t)n * n, mem, &st);
    if (st || !Xp || !Cp) return st ? st : SDPSR_OUT_OF_MEMORY;
    launch_pad_copy(c->stream, n, ld, dX, Xp, sizeof(TI));
    launch(c->stream, ld, ld, ld, Xp, ld, Xp, ld, Cp, ld, 1, 0, 0, 0);
    launch_unpad_copy(c->stream, n, ld, Cp, dC, sizeof(TO));
    HIP_TRY(c, hipGetLastError());
    return out_finish(c, X2, dC, (size_t)n * n, mem);
}

extern "C" {

int sdpsr_square_f64(sdpsr_ctx* c, int64_t n, const double* X, double* X2, int mem) {
    CHECK_CTX(c);
    return square_generic<double, double>(c, n, X, X2, mem, launch_gemm_tn_f64);
}
int sdpsr_square_f32(sdpsr_ctx* c, int64_t n, const float* X, float* X2, int mem) {
    CHECK_CTX(c);
    return square_generic<float, float>(c, n, X, X2, mem, launch_gemm_tn_f32);
}
int sdpsr_square_i8(sdpsr_ctx* c, int64_t n, const int8_t* X, int32_t* X2, int mem) {
    CHECK_CTX(c);
    return square_generic<int8_t, int32_t>(c, n, X, X2, mem, launch_gemm_tn_i8);
}

// `batch` symmetric int8 matrices squared in ONE launch, as the loop squares its channel matrices (lower-triangle
// tiles only, the persistent macro-tile launch by sdpsr_opts.square_kernel); results mirrored into full matrices
int sdpsr_square_i8_symmetric(sdpsr_ctx* c, int64_t n, int64_t batch, const int8_t* X, int32_t* X2, int mem) {
    CHECK_CTX(c);
    if (!X || !X2 || n < 1 || batch < 1 || batch > 8) return ctx_fail(c, SDPSR_BAD_ARGUMENT, "bad arguments");
    int st = SDPSR_OK;
    const int sk = c->opts.square_kernel;
    const int64_t ld = round_up(n, 128);
    const int8_t* dX = in_dev(c, "sq_in", X, (size_t)batch * n * n, mem, &st);
    int8_t* Xp = (int8_t*)ctx_buf(c, "sq_xpad", (size_t)batch * ld * ld);
    int32_t* Cp = (int32_t*)ctx_buf(c, "sq_cpad", (size_t)batch * ld * ld * 4);
    int32_t* dC = out_dev(c, "sq_out", X2, (size_t)batch * n * n, mem, &st);
    uint32_t* zflag = (uint32_t*)ctx_buf(c, "sq_zero", 64);
    if (st || !Xp || !Cp || !zflag) return st ? st : SDPSR_OUT_OF_MEMORY;
    HIP_TRY(c, hipMemsetAsync(zflag, 0, 64, c->stream));
    for (int64_t b = 0; b < batch; ++b) launch_pad_copy(c->stream, n, ld, dX + b * n * n, Xp + b * ld * ld, 1);
    launch_gemm_tn_i8_sym(c->stream, ld, ld, Xp, ld, Cp, ld, (int)batch, ld * ld, ld * ld, zflag, c->num_cus, sk);
    for (int64_t b = 0; b < batch; ++b) launch_unpad_mirror_lower_i32(c->stream, n, ld, Cp + b * ld * ld, dC + b * n * n);
    HIP_TRY(c, hipGetLastError());
    return out_finish(c, X2, dC, (size_t)batch * n * n, mem);
}

int sdpsr_gemm_tn_f64(sdpsr_ctx* c, int64_t m, int64_t n, int64_t k, const double* A, int64_t lda,
                      const double* B, int64_t ldb, double* C, int64_t ldc, int mem) {
    CHECK_CTX(c);
    if (!A || !B || !C || m < 1 || n < 1 || k < 1 || lda < k || ldb < k || ldc < m)
        return ctx_fail(c, SDPSR_BAD_ARGUMENT, "bad arguments");
    int st = SDPSR_OK;
    const int64_t mp = round_up(m, 128), np = round_up(n, 128), kp = round_up(k, 16);
    const double* dA = in_dev(c, "g_a", A, (size_t)lda * m, mem, &st);
    const double* dB = in_dev(c, "g_b", B, (size_t)ldb * n, mem, &st);
    double* dC = out_dev(c, "g_c", C, (size_t)ldc * n, mem, &st);
    double* Ap = (double*)ctx_buf(c, "g_ap", (size_t)kp * mp * 8);
    double* Bp = (double*)ctx_buf(c, "g_bp", (size_t)kp * np * 8);
    double* Cp = (double*)ctx_buf(c, "g_cp", (size_t)mp * np * 8);
    if (st || !Ap || !Bp || !Cp) return st ? st : SDPSR_OUT_OF_MEMORY;
    HIP_TRY(c, hipMemsetAsync(Ap, 0, (size_t)kp * mp * 8, c->stream));
    HIP_TRY(c, hipMemsetAsync(Bp, 0, (size_t)kp * np * 8, c->stream));
    HIP_TRY(c, hipMemcpy2DAsync(Ap, kp * 8, dA, lda * 8, k * 8, m, hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(c, hipMemcpy2DAsync(Bp, kp * 8, dB, ldb * 8, k * 8, n, hipMemcpyDeviceToDevice, c->stream));
    launch_gemm_tn_f64(c->stream, mp, np, kp, Ap, kp, Bp, kp, Cp, mp, 1, 0, 0, 0);
    HIP_TRY(c, hipMemcpy2DAsync(dC, ldc * 8, Cp, mp * 8, m * 8, n, hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(c, hipGetLastError());
    return out_finish(c, C, dC, (size_t)ldc * n, mem);
}

}  // extern "C"
