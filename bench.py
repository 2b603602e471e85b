#!/usr/bin/env python3
"""bench.py -- reductions/s of the Jordan-reduction hot path on MI355X.

One "step" = one reduction = admissible_subspace (device loop) + blockDiagonalize
(eigen_decomposition + irreducible_decomposition + basis_image) of BASELINE.json configs[3]:
a synthetic Jordan algebra of order N = 4096 with 34 basis matrices.  Inputs (C_L, X0_L, U) are
resident in HBM before the timed region; the setup stage (src/partitions.jl:117-142) is outside
the hot path.

Three instances of configs[3] are measured; `value` is the first, the other two are reported in
`workloads` of the same JSON line (each with reductions/s, iteration count, per-phase ms):
  closed_scheme     the 34-class scheme circulant(Z_32) (x) {I, J-I}_128 handed over as the SDP
                    data itself (C takes a distinct value per class): the loop confirms the
                    closure in ONE iteration, all blocks have size 1;
  theta_c32xk128    theta'-type SDP (C = ones, A = [adjacency; I], test/sd_problems.jl:22-26 form)
                    of the Cartesian product C_32 [] K_128: the loop starts from {diagonal, edges,
                    non-edges} and needs 5 iterations to reach the same 34-class scheme;
  theta_er7xk72     the same construction on ER(7) [] K_72, N = 4104: 5 iterations, 36 classes,
                    NON-commutative algebra, blocks [2,2,2,2,3] twice.

N ranks (one per GPU): every rank runs an independent random restart of the same reduction (its
own seed), the ranks agree on the partition (128-bit checksums all-gathered; labels travel only
on disagreement: MIN/MAX all-reduce, then the hash-meet with the device relabel), then each rank
block-diagonalises.  value = restarts finished by all ranks per second ("weak": per-GPU work is fixed).

`python bench.py --gpus N` without a launcher (no WORLD_SIZE in the environment) starts the N ranks
itself: N fresh child processes, one per GPU, created BEFORE this process imports torch or touches
a GPU (never a re-exec of a process that has); rank 0's JSON line is the output.  Under
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` the ranks are already there.

Prints ONE JSON line (rank 0).
"""
import argparse
import ctypes as C
import csv
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TF = 157.3   # MI355X_MICROARCH.md: dense fp32 matrix peak
FP64_MFMA_PEAK_TF = 78.6    # public MI355X spec (fp64 matrix); not in the local guide
I8_MFMA_PEAK_TOPS = 5000.0  # ~2x bf16 dense (MI355X_MICROARCH.md matrix-core table)
HBM_PEAK_GBS = 8000.0
ATOL = 1.4901161193847656e-08


def cpu_baseline_theta(pr, n_sample):
    """The CPU oracle on the instance that ITERATES (theta' SDP of C_32 [] K_k, 5 iterations of the loop), at a
    reduced order n_sample = 32 k: the loop side of the path gets a measured CPU figure too."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import sdpsr_oracle as O
    k = max(1, n_sample // 32)
    Cv, A, b, Ls, d = pr.theta_prime_product_problem(pr.cycle_adjacency(32), pr.symmetric_circulant_labels(32), k, seed=1)
    setup = O.admissible_setup(Cv, A, b)
    t0 = time.perf_counter()
    P = O.admissible_subspace(Cv, A, b, rng=np.random.default_rng(0), setup=setup)
    t1 = time.perf_counter()
    O.block_diagonalize(P, rng=np.random.default_rng(1))
    t2 = time.perf_counter()
    assert np.array_equal(P.matrix, Ls)
    return {"adm_s": t1 - t0, "bd_s": t2 - t1, "n": 32 * k, "dim": int(d)}


def cpu_baseline(pr, n_sample, seed):
    """The CPU oracle (NumPy/SciPy restatement, NOT Julia) timed on the host cores: one full
    reduction of the headline workload's generator at order n_sample."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import sdpsr_oracle as O
    try:
        from threadpoolctl import threadpool_info
        thr = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        thr = os.cpu_count() or 1
    Ls, d = pr.synthetic_jordan_partition(n_sample, seed=seed)
    Cv, A, b = pr.partition_as_sdp(Ls, seed=1)
    setup = O.admissible_setup(Cv, A, b)
    t0 = time.perf_counter()
    P = O.admissible_subspace(Cv, A, b, rng=np.random.default_rng(0), setup=setup)
    t1 = time.perf_counter()
    sizes, blks, _ = O.block_diagonalize(P, rng=np.random.default_rng(1))
    t2 = time.perf_counter()
    assert np.array_equal(P.matrix, Ls)
    return {"adm_s": t1 - t0, "bd_s": t2 - t1, "threads": thr, "n": n_sample, "dim": int(d)}


PROFILE_ROUND = "r05"
BENCH_N = 4096  # set from --n: the committed rocprofv3 summary looked up is the one of the order being run


def rocprof_average_us(pattern, summary=None):
    """Average duration (us) of a kernel in this round's committed rocprofv3 --kernel-trace --stats
    summary of the bench command AT THE ORDER BEING RUN (profiles/r05_bench_n<N>_kernel_stats.csv), or of another
    committed summary of this round (`summary`: the part of the file name after the round), or None."""
    try:
        path = sorted(glob.glob(os.path.join(ROOT, "profiles", f"{PROFILE_ROUND}_{summary}" if summary else f"{PROFILE_ROUND}_bench_n{BENCH_N}_kernel_stats*.csv")))[-1]
        for row in csv.DictReader(open(path)):
            if pattern in row["Name"]:
                return round(float(row["AverageNs"]) / 1e3, 2)
    except Exception:
        pass
    return None


def pmc_traffic(key):
    """HBM bytes per launch from this round's separate rocprofv3 --pmc passes (FETCH_SIZE x 2 +
    WRITE_SIZE, MI355X_MICROARCH.md HBM section), or None when the pass is not in profiles/."""
    try:
        pj = json.load(open(os.path.join(ROOT, "profiles", f"{PROFILE_ROUND}_pmc.json")))
        return round(pj[key]["traffic_bytes_per_launch"]), f"profiles/{PROFILE_ROUND}_pmc.json:" + key
    except Exception:
        return None, None


class Workload:
    """One instance: device-resident inputs of the loop + expectations for the checks."""

    def __init__(self, pkg, dev, name, note, Cv, A, b, labels, d, blocks):
        import torch
        self.name, self.note = name, note
        t_setup = time.perf_counter()
        setup = pkg.admissible_setup(Cv, A, b)
        self.setup_ms = (time.perf_counter() - t_setup) * 1e3  # the host stage outside the timed region (src/partitions.jl:117-142)
        n, CL, X0L, U = setup
        self.host = (CL, X0L, np.asfortranarray(U) if U.shape[1] else None)  # what a caller of the host-array interface holds
        # the host setup knows whether the basis matrices are symmetric (they are whenever the
        # constraint matrices are); the loop is told before every call (sdpsr_hint_symmetric_basis)
        self.hint = int(getattr(setup, "hint", 0))
        self.n, self.d, self.blocks = n, int(d), sorted(blocks)
        self.r = U.shape[1]
        t_up = time.perf_counter()
        self.tCL = torch.from_numpy(CL).to(dev)
        self.tX0 = torch.from_numpy(X0L).to(dev)
        self.tU = torch.from_numpy(np.ascontiguousarray(U.T)).to(dev) if self.r else None  # (r, n^2): rows = columns of U
        torch.cuda.synchronize()
        self.upload_ms = (time.perf_counter() - t_up) * 1e3
        self.tP = torch.empty(n * n, dtype=torch.int32, device=dev)
        self.golden = torch.from_numpy(np.ascontiguousarray(labels.ravel(order="F")).astype(np.int32)).to(dev)
        self.blk = None  # device buffer of the block images (sized by the first step)
        self.dev = dev


def launch_ranks(n_ranks, argv):
    """Parent of `bench.py --gpus N`: N child processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set,
    stdout and stderr inherited (rank 0 prints the JSON line).  This process never initialises a GPU."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n_ranks), "LOCAL_WORLD_SIZE": str(n_ranks),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    # poll all children: the moment one exits non-zero the others are killed (a rank that died leaves its peers waiting
    # in a collective until the backend's timeout); only fresh children were started, nothing is re-executed
    rc = 0
    while True:
        codes = [p.poll() for p in procs]
        bad = [c for c in codes if c not in (None, 0)]
        if bad:
            rc = bad[0]
            for p in procs:
                if p.poll() is None:
                    p.kill()
            for p in procs:
                p.wait()
            break
        if all(c == 0 for c in codes):
            break
        time.sleep(0.05)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--mode", default="i8", choices=["i8", "f32", "f64"])
    ap.add_argument("--cpu-n", type=int, default=-1, help="order of the CPU-baseline reduction (-1 = the headline order, 0 = skip)")
    ap.add_argument("--eig-driver", type=int, default=0, help="0 auto (module compression when dim(P) << n), 4 dense eigensolver forced")
    ap.add_argument("--no-graph", action="store_true", help="dense driver: launch the tridiagonalisation's kernels one by one (per-kernel rocprofv3 statistics)")
    ap.add_argument("--flags", type=int, default=0, help="sdpsr_opts.flags (include/sdpsr.h: SDPSR_FLAG_*) for A/B runs")
    ap.add_argument("--insert-wgs", type=int, default=0, help="sdpsr_opts.insert_wgs_per_cu (measurement knob)")
    ap.add_argument("--channels", type=int, default=0, help="sdpsr_opts.channels (0 = default: 2 + one confirm round)")
    ap.add_argument("--timers-in-timed-region", action="store_true",
                    help="record the per-phase HIP events inside the timed steps (default: in a separate instrumented pass)")
    ap.add_argument("--skip-roofline", action="store_true", help="only the timed steps (clean rocprofv3 kernel statistics)")
    ap.add_argument("--workload", default="closed_scheme", choices=["closed_scheme", "theta_c32xk128", "theta_er7xk72"],
                    help="instance run in the timed region (the other two are measured after it, rank 0); at --n 8192 the same "
                         "generators give C_32 [] K_256 and ER(7) [] K_144 (N = 8208)")
    ap.add_argument("--restarts-per-gpu", type=int, default=1,
                    help="independent random restarts per step and GPU, run by ONE sdpsr_jordan_reduce_batch call (fibers of one host "
                         "thread, one stream each); 1 = the headline definition (one reduction at a time)")
    ap.add_argument("--cpu-samples", type=int, default=1, help="samples of the CPU-oracle reduction at the headline order (the median is reported; one takes ~50 s)")
    ap.add_argument("--square-kernel", type=int, default=0, help="sdpsr_opts.square_kernel (0 default, 1 = 128 x 128 tiles, 64 = persistent forced)")
    args = ap.parse_args()
    global BENCH_N
    BENCH_N = args.n
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist
    from __graft_entry__ import load_package
    pkg = load_package()
    pr = pkg.problems
    L = pkg._lib

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # test hooks (single-GPU boxes): SDPSR_BENCH_SAME_DEVICE=1 puts every rank on cuda:0 and
    # SDPSR_BENCH_BACKEND=gloo swaps RCCL (which refuses two ranks on one device) for gloo, so
    # that the world_size > 1 control flow can be exercised where only one GPU is visible
    if os.environ.get("SDPSR_BENCH_SAME_DEVICE"):
        local = 0
    backend = os.environ.get("SDPSR_BENCH_BACKEND", "nccl")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
        else:
            dist.init_process_group(backend)
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")
    n = args.n

    def build(name, nn=None):
        n = nn or args.n
        if name == "closed_scheme":
            Ls, d = pr.synthetic_jordan_partition(n, seed=1)
            Cv, A, b = pr.partition_as_sdp(Ls, seed=1)
            return Workload(pkg, dev, name, f"partition_as_sdp of circulant Z_32 (x) K_{n // 32} (already closed)", Cv, A, b, Ls, d, [1] * d)
        if name == "theta_c32xk128":
            k = n // 32
            Cv, A, b, Ls, d = pr.theta_prime_product_problem(pr.cycle_adjacency(32), pr.symmetric_circulant_labels(32), k, seed=1)
            return Workload(pkg, dev, name, f"theta' SDP of C_32 [] K_{k} (C = ones, A = [adjacency; I])", Cv, A, b, Ls, d, [1] * d)
        gold = np.load(os.path.join(ROOT, "tests", "golden", "golden_partitions.npz"))["er7_P"].astype(np.int64)
        ke = max(1, round(n / 57))  # 72 at n = 4096 (N = 4104), 144 at n = 8192 (N = 8208)
        Cv, A, b, Ls, d = pr.theta_prime_product_problem(pr.er_graph_adjacency(7), gold, ke, seed=1)
        return Workload(pkg, dev, name, f"theta' SDP of ER(7) [] K_{ke}, N = {57 * ke}, non-commutative", Cv, A, b, Ls, d, [2, 2, 2, 2, 3] * 2)

    mode = {"i8": L.SQUARE_I8, "f32": L.SQUARE_F32, "f64": L.SQUARE_F64}[args.mode]
    opt_flags = args.flags | (L.FLAG_NO_GRAPH if args.no_graph else 0)
    ctx = pkg.Context(device=local, seed=1000 + rank, square_mode=mode, eig_driver=args.eig_driver, flags=opt_flags, channels=args.channels, insert_wgs_per_cu=args.insert_wgs,
                      square_kernel=args.square_kernel)
    RPG = max(1, args.restarts_per_gpu)

    def vp(t):
        return C.c_void_p(t.data_ptr()) if t is not None else None

    class Acc:
        def __init__(self):
            self.phase = np.zeros(L.T_COUNT)
            self.iters = 0
            self.meets = 0  # agreement steps that needed the hash-meet (ranks ended on different partitions)
            self.bd_adopted = 0  # steps in which this rank's blockDiagonalize failed and another rank's block sizes were adopted
            self.bd_all_failed = 0  # steps in which every rank failed and all of them drew again

    # test hook: SDPSR_BENCH_FORCE_DISAGREE=1 makes rank 1 report a COARSER partition in the warm-up steps (classes 1
    # and 2 merged, canonically relabelled on the device), so that the MIN/MAX all-reduce and the hash-meet with the
    # device relabel run; the agreed partition must again be the generator's closure (the `check` below)
    force_disagree = bool(os.environ.get("SDPSR_BENCH_FORCE_DISAGREE"))
    # test hook: SDPSR_BENCH_FORCE_BD_FAIL=r makes rank r report DimensionMismatch for its blockDiagonalize in the
    # warm-up steps: the winner selection of SURVEY 8(e)(ii) must hand it another rank's block sizes
    force_bd_fail = int(os.environ.get("SDPSR_BENCH_FORCE_BD_FAIL", "-1"))

    def batch_step(w, cx, R, check=False):
        """R independent restarts of the reduction in ONE call (sdpsr_jordan_reduce_batch); returns the statuses"""
        lib = cx._lib
        if getattr(w, "bP", None) is None or len(w.bP) != R:
            w.bP = [torch.empty(w.n * w.n, dtype=torch.int32, device=dev) for _ in range(R)]
            w.bblk = [None] * R
        pP = (C.c_void_p * R)(*[t.data_ptr() for t in w.bP])
        pb = (C.c_void_p * R)(*[(t.data_ptr() if t is not None else 0) for t in w.bblk])
        caps = (C.c_int64 * R)(*[(t.numel() if t is not None else 0) for t in w.bblk])
        dd, it, nb = (C.c_int64 * R)(), (C.c_int32 * R)(), (C.c_int32 * R)()
        ssq, ss, st = (C.c_int64 * R)(), (C.c_int64 * R)(), (C.c_int32 * R)()
        if w.hint:
            lib.sdpsr_hint_symmetric_basis(cx._h, w.hint)
        lib.sdpsr_jordan_reduce_batch(cx._h, R, None, w.n, vp(w.tCL), vp(w.tX0), vp(w.tU), w.r, ATOL, ATOL, C.cast(pP, C.c_void_p), dd, it, nb, ssq, ss,
                                      C.cast(pb, C.c_void_p), caps, st, L.MEM_DEVICE)
        for i in range(R):
            if st[i] == 0 and dd[i] * ssq[i] > caps[i]:  # first step (or a changed size): the buffer for the next one
                w.bblk[i] = torch.empty(max(1, dd[i] * ssq[i]), dtype=torch.float64, device=dev)
            if check and st[i] == 0:
                assert dd[i] == w.d and bool((w.bP[i] == w.golden).all()), "partition differs from the generator's closure"
                assert ssq[i] == sum(x * x for x in w.blocks), (ssq[i], w.blocks)
        return [int(x) for x in st], sum(int(x) for x in it)

    def one_step(w, acc, cx, check=False, collective=True, timers=True):
        lib = cx._lib
        dd = C.c_int64(0)
        it = C.c_int32(0)
        nb = C.c_int32(0)
        ssq = C.c_int64(0)
        ss = C.c_int64(0)
        ms = (C.c_double * L.T_COUNT)()
        tp = (lambda a: C.cast(a, C.c_void_p)) if timers else (lambda a: None)  # NULL: no phase events in the stream
        if w.hint:
            lib.sdpsr_hint_symmetric_basis(cx._h, w.hint)
        if not (world > 1 and collective):
            # one rank: the whole reduction in ONE call (sdpsr_jordan_reduce: the partition stays on the device, no host
            # synchronisation between admissible_subspace, blockDiagonalize and basis_image); the image buffer of the
            # previous step is offered, a first step (or a changed size) fetches the images with sdpsr_block_images
            cap = w.blk.numel() if w.blk is not None else 0
            cx.check(lib.sdpsr_jordan_reduce(cx._h, w.n, vp(w.tCL), vp(w.tX0), vp(w.tU), w.r, ATOL, ATOL, vp(w.tP), C.byref(dd), C.byref(it),
                                             C.byref(nb), C.byref(ssq), C.byref(ss), vp(w.blk), cap, None, 0, tp(ms), L.MEM_DEVICE))
            if dd.value * ssq.value > cap:
                w.blk = torch.empty(max(1, dd.value * ssq.value), dtype=torch.float64, device=dev)
                ms2 = (C.c_double * L.T_COUNT)()
                cx.check(lib.sdpsr_block_images(cx._h, vp(w.blk), None, tp(ms2), L.MEM_DEVICE))
                for i in range(1, L.T_COUNT):
                    ms[i] += ms2[i]
            acc.iters += it.value
            for i in range(L.T_COUNT):
                acc.phase[i] += ms[i]
            if check:
                assert dd.value == w.d and bool((w.tP == w.golden).all()), "partition differs from the generator's closure"
        else:
            cx.check(lib.sdpsr_admissible_subspace(cx._h, w.n, vp(w.tCL), vp(w.tX0), vp(w.tU), w.r, ATOL, vp(w.tP), C.byref(dd),
                                                   C.byref(it), tp(ms), L.MEM_DEVICE))
            acc.iters += it.value
            for i in range(L.T_COUNT):
                acc.phase[i] += ms[i]
            # agree the partition across the restarts (canonical labels: equal w.p. 1): 128-bit
            # checksums computed on the device are all-gathered; the 64 MiB label matrix itself
            # only travels if they differ: MIN/MAX all-reduce, and -- a rank's draws missed a split --
            # the meet of the partitions through one SUM all-reduce of hashed labels and the
            # library's canonical relabel on the device (parallel.agree_partition)
            if force_disagree and check and rank == 1:
                coarse = torch.where(w.tP == 2, torch.ones_like(w.tP), w.tP)
                relab, dcoarse = pkg.relabel_keys(coarse.to(torch.int64), ctx=cx)
                w.tP.copy_(relab)
                dd.value = dcoarse
            agreed, lab = pkg.parallel.agree_partition(
                w.tP, lambda sig: pkg.relabel_keys(sig, ctx=cx), checksum=lambda t: pkg.partition_checksum(t, ctx=cx))
            if not agreed:
                acc.meets += 1
                w.tP.copy_(lab)
                dd.value = int(lab.max().item())
            if check:
                assert dd.value == w.d and bool((w.tP == w.golden).all()), "partition differs from the generator's closure"
            # blockDiagonalize is randomized; the reference's answer to NumericalInconsistency / DimensionMismatch is "try
            # again" (src/eigen_decomposition.jl:264-270).  The ranks' tries have run side by side: the lowest rank
            # whose status is OK wins and broadcasts its block sizes (parallel.agree_block_diagonalization, SURVEY
            # 8(e)(ii)); a rank that failed adopts them (its images are the winner's to deliver); if all failed, all draw again
            for attempt in range(5):
                ms1 = (C.c_double * L.T_COUNT)()
                st_bd = lib.sdpsr_block_diagonalize(cx._h, w.n, vp(w.tP), dd.value, ATOL, C.byref(nb), C.byref(ssq), C.byref(ss),
                                                    tp(ms1), L.MEM_DEVICE)
                if st_bd not in (0, 2, 3):
                    cx.check(st_bd)
                if check and force_bd_fail == rank:
                    st_bd = 3
                my_sizes = []
                if st_bd == 0:
                    sz = np.zeros(nb.value, dtype=np.int32)
                    cx.check(lib.sdpsr_block_sizes(cx._h, sz.ctypes.data_as(C.c_void_p)))
                    my_sizes = [int(x) for x in sz]
                winner, agreed_sizes, _ = pkg.parallel.agree_block_diagonalization(st_bd, my_sizes, device=dev)
                for i in range(1, L.T_COUNT):
                    acc.phase[i] += ms1[i]
                if winner >= 0:
                    break
                acc.bd_all_failed += 1
            else:
                raise RuntimeError("five consecutive randomized failures on every rank")
            if st_bd == 0:
                if w.blk is None or w.blk.numel() < dd.value * ssq.value:
                    w.blk = torch.empty(max(1, dd.value * ssq.value), dtype=torch.float64, device=dev)
                ms2 = (C.c_double * L.T_COUNT)()
                cx.check(lib.sdpsr_block_images(cx._h, vp(w.blk), None, tp(ms2), L.MEM_DEVICE))
                for i in range(1, L.T_COUNT):
                    acc.phase[i] += ms2[i]
            else:
                acc.bd_adopted += 1
            if check:
                assert sorted(agreed_sizes) == w.blocks, (sorted(agreed_sizes), w.blocks)
            return len(agreed_sizes)
        if check:
            sizes = np.zeros(nb.value, dtype=np.int32)
            cx.check(lib.sdpsr_block_sizes(cx._h, sizes.ctypes.data_as(C.c_void_p)))
            assert sorted(int(s) for s in sizes) == w.blocks, (sorted(sizes), w.blocks)
        return nb.value

    def retrying(fn):
        """blockDiagonalize is randomized; the reference's answer to NumericalInconsistency /
        DimensionMismatch is "try again" (src/eigen_decomposition.jl:264-270) -- counted, not hidden."""
        for attempt in range(5):
            try:
                return fn(), attempt
            except (pkg.NumericalInconsistency, pkg.DimensionMismatch):
                continue
        raise RuntimeError("five consecutive randomized failures")

    def fence(cx):
        cx.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def phases(acc, steps):
        return {k: round(acc.phase[i] / steps, 3) for k, i in
                (("project", L.T_PROJECT), ("square", L.T_SQUARE), ("refine", L.T_REFINE), ("eigen", L.T_EIGEN),
                 ("iso_QtAQ", L.T_ISO), ("irreducible", L.T_IRRED), ("basis_image", L.T_IMAGE))}

    # ---- the timed region: K reductions of the headline instance ----
    w0 = build(args.workload)
    acc = Acc()

    def batched_step(w, a, cx, check=False):
        """RPG restarts per rank in one call.  Across ranks AND within a rank the restarts' partitions are agreed as in the
        one-restart flow (parallel.agree_partitions: the R x world table of checksums; on a difference the meet of all
        of them), then the block-diagonalisation winner with the sizes this rank actually got.  Returns the number of this
        rank's restarts that did NOT deliver a reduction (randomized failures: counted, never asserted on)."""
        sts, its = batch_step(w, cx, RPG, check=check and world == 1)
        a.iters += its / RPG
        failed = sum(1 for x in sts if x != 0)
        if world > 1:
            if force_disagree and check and rank == 1:  # test hook: restart 1 of rank 1 reports a coarser partition
                t = w.bP[min(1, RPG - 1)]
                coarse = torch.where(t == 2, torch.ones_like(t), t)
                relab, _ = pkg.relabel_keys(coarse.to(torch.int64), ctx=cx)
                t.copy_(relab)
            agreed, lab = pkg.parallel.agree_partitions(w.bP, lambda sig: pkg.relabel_keys(sig, ctx=cx),
                                                        checksum=lambda t: pkg.partition_checksum(t, ctx=cx))
            valid = [True] * RPG  # restart i's block images describe the agreed partition
            if not agreed:
                a.meets += 1
                for i, t in enumerate(w.bP):
                    valid[i] = bool((t == lab).all())  # (a restart the meet refined: its images belong to a partition that was replaced)
                    if not valid[i]:
                        t.copy_(lab)
                        if sts[i] == 0:
                            failed += 1
            if check:
                assert all(bool((t == w.golden).all()) for t in w.bP), "partition differs from the generator's closure"
            # the first restart of this rank whose blockDiagonalize succeeded offers the sizes it actually got
            my_sizes, st_mine = [], 3
            if check and force_bd_fail == rank:
                sts = [3] * RPG
                failed = RPG
            for i, x in enumerate(sts):
                if x == 0 and valid[i]:
                    sz = np.zeros(256, dtype=np.int32)
                    if cx._lib.sdpsr_batch_block_sizes(cx._h, i, sz.ctypes.data_as(C.c_void_p)) == 0:
                        my_sizes = [int(v) for v in sz if v > 0]
                        st_mine = 0
                        break
            win, agreed_sizes, _ = pkg.parallel.agree_block_diagonalization(st_mine, my_sizes, device=dev)
            if win < 0:
                a.bd_all_failed += 1
            elif st_mine != 0:
                a.bd_adopted += 1
            if check and win >= 0:
                assert sorted(agreed_sizes) == w.blocks, (sorted(agreed_sizes), w.blocks)
        return failed

    for _ in range(args.warmup):
        if RPG > 1:
            batched_step(w0, acc, ctx, check=True)
        else:
            retrying(lambda: one_step(w0, acc, ctx, check=True))
    acc_warm = acc
    acc = Acc()
    retries = 0
    failed_restarts = 0  # batched steps: restarts whose status was not 0 -- they do not count as reductions
    fence(ctx)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if RPG > 1:
            failed_restarts += batched_step(w0, acc, ctx)  # a failed restart is not a reduction; the step's other restarts stand
        else:
            _, rt = retrying(lambda: one_step(w0, acc, ctx, timers=args.timers_in_timed_region))
            retries += rt
    fence(ctx)
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        # one-restart flow: a rank that adopted another rank's block sizes delivered no images for that step
        fr = torch.tensor([float(failed_restarts + (acc.bd_adopted if RPG == 1 else 0))], dtype=torch.float64, device=dev)
        dist.all_reduce(fr, op=dist.ReduceOp.SUM)
        failed_restarts = int(fr.item())
    acc_timed = acc
    if not args.timers_in_timed_region and RPG == 1:
        # the per-phase HIP events (about thirty records per reduction) stay out of the timed region: the phase
        # split comes from an instrumented pass of the same number of steps right after it
        acc_timed = Acc()
        for _ in range(args.steps):
            retrying(lambda: one_step(w0, acc_timed, ctx))
        acc_timed.iters = acc.iters
    if RPG > 1:
        batched_step(w0, Acc(), ctx, check=True)
    else:
        retrying(lambda: one_step(w0, Acc(), ctx, check=True))  # results still correct after the timed region

    def measure(w, cx, steps):
        """rank 0 only, no collectives: reductions/s of another instance / driver"""
        a = Acc()
        retrying(lambda: one_step(w, a, cx, check=True, collective=False))
        retrying(lambda: one_step(w, a, cx, check=True, collective=False))
        a = Acc()
        cx.synchronize()
        torch.cuda.synchronize()
        t = time.perf_counter()
        rts = 0
        for _ in range(steps):
            _, rt = retrying(lambda: one_step(w, a, cx, collective=False, timers=args.timers_in_timed_region))
            rts += rt
        cx.synchronize()
        torch.cuda.synchronize()
        el = (time.perf_counter() - t) / steps
        if not args.timers_in_timed_region:  # phase split from an instrumented pass (see the timed region above)
            its = a.iters
            a = Acc()
            for _ in range(steps):
                retrying(lambda: one_step(w, a, cx, collective=False))
            a.iters = its
        return {"value": round(1.0 / el, 3), "unit": "reductions/s", "ms_per_step": round(el * 1e3, 3), "steps": steps,
                "N": w.n, "dim": w.d, "blocks": w.blocks if len(set(w.blocks)) > 1 else f"{len(w.blocks)} x size {w.blocks[0]}",
                "iterations_per_reduction": a.iters / steps, "randomized_retries": rts, "phase_ms_per_step": phases(a, steps),
                "instance": w.note}

    total_red = args.steps * world * RPG - failed_restarts  # reductions DELIVERED in the timed region
    workloads = {args.workload: {"value": round(total_red / dt, 3), "unit": "reductions/s", "ms_per_step": round(dt / args.steps * 1e3, 3),
                                 "steps": args.steps, "N": w0.n, "dim": w0.d, "blocks": f"{len(w0.blocks)} x size {w0.blocks[0]}" if len(set(w0.blocks)) == 1 else w0.blocks,
                                 "iterations_per_reduction": acc_timed.iters / max(1, args.steps), "randomized_retries": retries,
                                 "phase_ms_per_step": phases(acc_timed, args.steps), "instance": w0.note, "timed_region": True}}
    meets_warmup = acc_warm.meets
    variants = {}
    kernels = {}
    roof = None
    cpu = None
    if rank == 0 and not args.skip_roofline and args.eig_driver == 0 and n in (4096, 8192) and RPG == 1:
        for name in ("closed_scheme", "theta_c32xk128", "theta_er7xk72"):
            if name != args.workload:
                wk = build(name)
                workloads[name] = measure(wk, ctx, 10)
                del wk
                torch.cuda.empty_cache()
        # the headline instance with the dense eigensolver forced (hand-written tridiagonalisation, own tridiagonal
        # divide and conquer, own compact-WY back-transformation on the full n x n generic element)
        ctx_d = pkg.Context(device=local, seed=2000, square_mode=mode, eig_driver=4)
        try:
            variants["dense_eigensolver"] = measure(w0, ctx_d, 3)
            variants["dense_eigensolver"]["note"] = "eig_driver=4: diagonalize on the full n x n generic element"
        finally:
            ctx_d.close()
        # two independent restarts IN FLIGHT on the one GPU: two host threads, each with its own ctx (own stream, own
        # buffers, own seed) and its own copy of the headline instance.  Not the headline figure (that is one reduction
        # at a time): it shows how much of a step is host round trips and launch gaps that a second restart fills
        import threading
        for nth in (2, 4):
            wks = [build(args.workload) for _ in range(nth)]
            cxs = [pkg.Context(device=local, seed=3000 + i, square_mode=mode, flags=opt_flags, channels=args.channels) for i in range(nth)]
            try:
                for wk, cx in zip(wks, cxs):
                    retrying(lambda: one_step(wk, Acc(), cx, check=True, collective=False))
                    retrying(lambda: one_step(wk, Acc(), cx, check=True, collective=False))
                steps2 = max(10, args.steps)
                errs = []
                gate = threading.Barrier(nth + 1)

                def run(wk, cx):
                    try:
                        a2 = Acc()
                        gate.wait()
                        for _ in range(steps2):
                            retrying(lambda: one_step(wk, a2, cx, collective=False, timers=False))
                        cx.synchronize()
                    except Exception as exc:  # noqa: BLE001
                        errs.append(repr(exc))

                ths = [threading.Thread(target=run, args=(wk, cx)) for wk, cx in zip(wks, cxs)]
                for th in ths:
                    th.start()
                torch.cuda.synchronize()
                gate.wait()
                t2 = time.perf_counter()
                for th in ths:
                    th.join()
                torch.cuda.synchronize()
                el2 = time.perf_counter() - t2
                if not errs:
                    for wk, cx in zip(wks, cxs):
                        retrying(lambda: one_step(wk, Acc(), cx, check=True, collective=False))  # still the generator's closure
                    variants["%d_restarts_in_flight" % nth] = {"value": round(nth * steps2 / el2, 3), "unit": "reductions/s", "threads": nth, "steps_per_thread": steps2,
                                                          "note": "%d ctx threads on one GPU, one reduction at a time each; informational (the headline is one ctx)" % nth}
                else:
                    variants["%d_restarts_in_flight" % nth] = {"error": errs[0][:200]}
            finally:
                for cx in cxs:
                    cx.close()
                del wks
                torch.cuda.empty_cache()

        if n == 4096:
            # BASELINE configs[4]: the same generator at N = 8192 end to end (one reduction at a time, this ctx)
            try:
                w8 = build("closed_scheme", 8192)
                variants["n8192_closed_scheme"] = measure(w8, ctx, 5)
                variants["n8192_closed_scheme"]["note"] = "configs[4] scale: circulant Z_32 (x) K_256 handed over as SDP data, N = 8192; bench.py --n 8192 times all three instances"
                del w8
                torch.cuda.empty_cache()
            except Exception as exc:  # noqa: BLE001
                variants["n8192_closed_scheme"] = {"error": repr(exc)[:200]}
        # the same WITHOUT host threads: R restarts in one sdpsr_jordan_reduce_batch call on one ctx (fibers of this thread)
        for R in (2, 4):
            try:
                wb = build(args.workload)
                for _ in range(3):
                    batch_step(wb, ctx, R, check=True)
                stepsb = max(10, args.steps)
                ctx.synchronize()
                torch.cuda.synchronize()
                tb = time.perf_counter()
                failed = 0
                for _ in range(stepsb):
                    sts, _ = batch_step(wb, ctx, R)
                    failed += sum(1 for x in sts if x != 0)
                torch.cuda.synchronize()
                elb = time.perf_counter() - tb
                batch_step(wb, ctx, R, check=True)
                variants["batch_%d_restarts_one_call" % R] = {"value": round(R * stepsb / elb, 3), "unit": "reductions/s", "restarts_per_call": R, "calls": stepsb,
                                                             "randomized_failures": failed,
                                                             "note": "sdpsr_jordan_reduce_batch: %d restarts per call, one host thread, one ctx; informational" % R}
                del wb
                torch.cuda.empty_cache()
            except Exception as exc:  # noqa: BLE001
                variants["batch_%d_restarts_one_call" % R] = {"error": repr(exc)[:200]}

    # ---- what a caller of the HOST-array interface pays (the reference's seam hands host arrays to admissible_subspace on
    # every call, src/partitions.jl:109-116; the Julia shim passes SDPSR_MEM_HOST): C_L, X0_L, U uploaded, labels and
    # block images downloaded inside the call.  One call; a 4-restart batch from host arrays (ONE upload for the four);
    # and the problem handle (upload once, outside the calls) with host outputs.
    if rank == 0 and not args.skip_roofline and RPG == 1:
        try:
            CLh, X0h, Uh = w0.host
            nn = w0.n
            hP = np.zeros(nn * nn, dtype=np.uint32)
            hblk = np.zeros(max(1, w0.d * sum(x * x for x in w0.blocks)))
            dd, it, nb, ssq, ss = C.c_int64(0), C.c_int32(0), C.c_int32(0), C.c_int64(0), C.c_int64(0)
            hp = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None  # noqa: E731

            def host_call():
                if w0.hint:
                    ctx._lib.sdpsr_hint_symmetric_basis(ctx._h, w0.hint)
                return ctx._lib.sdpsr_jordan_reduce(ctx._h, nn, hp(CLh), hp(X0h), hp(Uh), w0.r, ATOL, ATOL, hp(hP), C.byref(dd), C.byref(it), C.byref(nb),
                                                    C.byref(ssq), C.byref(ss), hp(hblk), hblk.size, None, 0, None, L.MEM_HOST)
            for _ in range(2):
                host_call()
            assert dd.value == w0.d and np.array_equal(hP.view(np.int32), w0.golden.cpu().numpy())
            b0 = ctx.transfer_bytes()
            stepsh = 5
            th = time.perf_counter()
            okh = sum(1 for _ in range(stepsh) if host_call() == 0)
            elh = time.perf_counter() - th
            b1 = ctx.transfer_bytes()
            variants["host_arrays_one_call"] = {"value": round(okh / elh, 3), "unit": "reductions/s", "ms_per_call": round(elh / stepsh * 1e3, 3), "calls": stepsh,
                                                "h2d_MB_per_call": round((b1[0] - b0[0]) / stepsh / 1e6, 1), "d2h_MB_per_call": round((b1[1] - b0[1]) / stepsh / 1e6, 1),
                                                "note": "sdpsr_jordan_reduce with SDPSR_MEM_HOST for every array (what the Julia shim does): C_L, X0_L, U up, labels and block images down, "
                                                        "pageable host memory; never the headline value"}
            R4 = 4
            Ps4 = [np.zeros(nn * nn, dtype=np.uint32) for _ in range(R4)]
            bl4 = [np.zeros(hblk.size) for _ in range(R4)]
            pP4 = (C.c_void_p * R4)(*[a.ctypes.data for a in Ps4])
            pb4 = (C.c_void_p * R4)(*[a.ctypes.data for a in bl4])
            cap4 = (C.c_int64 * R4)(*[a.size for a in bl4])
            d4, st4 = (C.c_int64 * R4)(), (C.c_int32 * R4)()

            def host_batch():
                if w0.hint:
                    ctx._lib.sdpsr_hint_symmetric_basis(ctx._h, w0.hint)
                ctx._lib.sdpsr_jordan_reduce_batch(ctx._h, R4, None, nn, hp(CLh), hp(X0h), hp(Uh), w0.r, ATOL, ATOL, C.cast(pP4, C.c_void_p), d4, None, None, None, None,
                                                   C.cast(pb4, C.c_void_p), cap4, st4, L.MEM_HOST)
                return sum(1 for x in st4 if x == 0)
            host_batch()
            b0 = ctx.transfer_bytes()
            th = time.perf_counter()
            okb = sum(host_batch() for _ in range(stepsh))
            elb = time.perf_counter() - th
            b1 = ctx.transfer_bytes()
            variants["host_arrays_batch_4"] = {"value": round(okb / elb, 3), "unit": "reductions/s", "ms_per_call": round(elb / stepsh * 1e3, 3), "restarts_per_call": R4, "calls": stepsh,
                                               "h2d_MB_per_call": round((b1[0] - b0[0]) / stepsh / 1e6, 1), "d2h_MB_per_call": round((b1[1] - b0[1]) / stepsh / 1e6, 1),
                                               "note": "sdpsr_jordan_reduce_batch, 4 restarts from host arrays: ONE upload of C_L, X0_L, U for the four (round 4: one per restart); "
                                                       "every restart's labels and block images go back to the host"}
            hprob = C.c_void_p()
            tc = time.perf_counter()
            ctx.check(ctx._lib.sdpsr_problem_create(ctx._h, nn, hp(CLh), hp(X0h), hp(Uh), w0.r, w0.hint, L.MEM_HOST, C.byref(hprob)))
            create_ms = (time.perf_counter() - tc) * 1e3
            try:
                def prob_batch():
                    ctx._lib.sdpsr_problem_reduce_batch(ctx._h, hprob, R4, None, ATOL, ATOL, C.cast(pP4, C.c_void_p), d4, None, None, None, None, C.cast(pb4, C.c_void_p), cap4, st4,
                                                        L.MEM_HOST)
                    return sum(1 for x in st4 if x == 0)
                prob_batch()
                b0 = ctx.transfer_bytes()
                th = time.perf_counter()
                okp = sum(prob_batch() for _ in range(stepsh))
                elp = time.perf_counter() - th
                b1 = ctx.transfer_bytes()
                variants["problem_handle_batch_4"] = {"value": round(okp / elp, 3), "unit": "reductions/s", "ms_per_call": round(elp / stepsh * 1e3, 3), "restarts_per_call": R4,
                                                      "calls": stepsh, "create_ms": round(create_ms, 2), "h2d_MB_per_call": round((b1[0] - b0[0]) / stepsh / 1e6, 3),
                                                      "d2h_MB_per_call": round((b1[1] - b0[1]) / stepsh / 1e6, 1),
                                                      "note": "sdpsr_problem_create once (create_ms: the upload), then sdpsr_problem_reduce_batch: nothing is uploaded per call; "
                                                              "labels and block images of the four restarts still go back to host arrays"}
            finally:
                ctx._lib.sdpsr_problem_destroy(hprob)
        except Exception as exc:  # noqa: BLE001
            variants["host_arrays_one_call"] = variants.get("host_arrays_one_call") or {"error": repr(exc)[:300]}

    # ---- roofline leg: per-launch duration of the hot kernels, HIP events on ctx's stream ----
    lib = ctx._lib

    def prof(kind, nn, aux=0, reps=10):
        v = C.c_double(0)
        ctx.check(L.load_prof_library().sdpsr_profile_kernel(ctx._h, kind, nn, aux, reps, C.byref(v)))
        return v.value

    if rank == 0 and not args.skip_roofline:
        d, r = w0.d, w0.r
        flops = 2.0 * n ** 3
        # the int8 square is launched exactly as the product path launches it: all its channels in
        # one launch (so that the HIP-event duration agrees with rocprofv3's average for the kernel)
        TP = args.channels if args.channels > 0 else 2  # channels of one product launch (sdpsr_opts.channels default)
        for name, kind, peak, unit, batch in (("square_i8", 0, I8_MFMA_PEAK_TOPS, "TOP/s", TP), ("square_f32", 1, FP32_MFMA_PEAK_TF, "TFLOP/s", 1),
                                              ("gemm_f64", 2, FP64_MFMA_PEAK_TF, "TFLOP/s", 1)):
            ms = prof(kind, n, aux=batch)
            ach = batch * flops / (ms * 1e-3) / 1e12
            kernels[name] = {"ms_per_launch": round(ms, 4), "channels_per_launch": batch, "achieved": round(ach, 2), "peak": peak, "unit": unit,
                             "frac": round(ach / peak, 4), "bound": "mfma", "algorithmic": "2*N^3 per channel, full square"}
        # north-star bar of BASELINE.json: the partition-square step at N = 8192 against the fp32 MFMA
        # peak (>= 40 % asked).  Measured on the fp32 square kernel (same code path as
        # square_mode="f32") and on the default int8 square (4 channels, full launch).
        if n != 8192:
            f8 = 2.0 * 8192 ** 3
            ms8 = prof(1, 8192, aux=1, reps=3)
            a8 = f8 / (ms8 * 1e-3) / 1e12
            kernels["square_f32_n8192"] = {"ms_per_launch": round(ms8, 3), "achieved": round(a8, 2), "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s",
                                           "frac": round(a8 / FP32_MFMA_PEAK_TF, 4), "bound": "mfma"}
            ms8 = prof(0, 8192, aux=4, reps=3)
            a8 = 4 * f8 / (ms8 * 1e-3) / 1e12
            kernels["square_i8_n8192"] = {"ms_per_launch": round(ms8, 3), "channels_per_launch": 4, "achieved": round(a8, 2), "peak": I8_MFMA_PEAK_TOPS,
                                          "unit": "TOP/s", "frac": round(a8 / I8_MFMA_PEAK_TOPS, 4), "bound": "mfma"}
        # partition refinement (src/partitions.jl:44-66): 16 B per entry algorithmic (8 value + 4 old
        # + 4 new label), in the three regimes of the configs
        for cls in sorted({int(d), 3000, n * n // 2}):
            ms = prof(3, n, aux=cls, reps=5)
            gbs = 16.0 * n * n / (ms * 1e-3) / 1e9
            kernels[f"refine_{cls}_classes"] = {"ms": round(ms, 4), "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                                                "bound": "hbm", "algorithmic_bytes": 16 * n * n,
                                                # the probe's own summary (tools/refine_probe.py warm under rocprofv3, N = 4096; the first / full launches of
                                                # the one-workgroup-per-CU insert kernel share a row there)
                                                "rocprof_avg_us": ({k: rocprof_average_us(k, f"refine_warm_{cls}_classes_kernel_stats.csv")
                                                                    for k in (("b2_count_kernel", "b2_scatter_kernel", "b2_resolve_kernel", "bk_bits_kernel", "bk_label_kernel")
                                                                              if cls == n * n // 2 else
                                                                              ("refine_insert_mid_kernel<sdpsr::SrcArray, 1>", "refine_insert_mid_kernel<sdpsr::SrcArray, 8>", "refine_label_kernel"))}
                                                                   if n == 4096 else None)}
        ms = prof(4, n, aux=max(r, 1), reps=5)  # gather+project+signature: (4 + 8r)*2 read + 8 write per entry
        gbs = ((4.0 + 8.0 * max(r, 1)) * 2 + 8.0) * n * n / (ms * 1e-3) / 1e9
        kernels["project_sig"] = {"ms": round(ms, 4), "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "bound": "hbm"}
        # basis_image of the headline instance (every block 1 x 1): class sums of a PAIR of vectors (x = sum_k q_k and a
        # sign-randomised copy for the self-check) + contraction; the projection formula runs when the check fails
        ph = phases(acc_timed, args.steps)
        S1 = len(w0.blocks)
        kernels["basis_image"] = {"ms": ph["basis_image"], "bound": "LDS read-modify-write latency (class_sums2_kernel) / L2 gathers (fallback)",
                                  "algorithmic_bytes": round(4.0 * n * n), "unit": "ms",
                                  "note": "phase timer (HIP events) of sdpsr_block_images, average over the steps: commutative shortcut "
                                          "(bi_signed_sums + class_sums2 + bi_contract_check) plus, in the ~15 % of the steps whose invariance check fails, "
                                          "the two-stage projection kernels (basis_image_rows + basis_image_blocks)",
                                  "rocprof_avg_us": {"class_sums2_kernel": rocprof_average_us("class_sums2_kernel"),
                                                     "bi_contract_check_kernel": rocprof_average_us("bi_contract_check_kernel"),
                                                     "basis_image_rows_kernel (fallback)": rocprof_average_us("basis_image_rows_kernel")}}
        wdim = min(64, int(d))  # the compressed eigenproblem of the module-compression driver: w = dim <S>x <= dim(P)
        # solved on the host inside the read-back the driver makes anyway (small_eigen_host.cpp), while the device forms
        # the second generic element; the one-workgroup Jacobi kernel it replaced (SDPSR_FLAG_SMALL_EIGEN_ON_DEVICE) beside it
        ms_h = prof(10, wdim, reps=20)
        ms_d = prof(8, wdim, reps=10)
        kernels["small_eigen_host"] = {"ms": round(ms_h, 4), "order": wdim, "where": "one host core (Householder + implicit QL), overlapped with the "
                                       "device's second generic element", "device_jacobi_one_workgroup_ms": round(ms_d, 4)}
        # Y = A(v) W straight from the labels on v_mfma_f64_16x16x4_f64 (w = the compressed order): the time is the
        # whole product (transposed copy of W + MFMA kernel + reduction of the partial sums); executed flop =
        # 2 N^2 * 16*ceil(w/16) (whole 16-column tiles), useful = 2 N^2 w
        ms = prof(9, n, aux=wdim | (1 << 8) | (int(d) << 12), reps=20)
        fl_exec = 2.0 * n * n * 16 * ((wdim + 15) // 16)
        kernels["label_spmm_mfma"] = {"ms": round(ms, 4), "w": wdim, "bound": "mfma", "unit": "TFLOP/s", "peak": FP64_MFMA_PEAK_TF,
                                      "achieved": round(fl_exec / (ms * 1e-3) / 1e12, 2), "frac": round(fl_exec / (ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TF, 4),
                                      "useful": round(2.0 * n * n * wdim / (ms * 1e-3) / 1e12, 2), "algorithmic_bytes": 4 * n * n,
                                      "rocprof_avg_us": {"label_spmm_mfma_kernel": rocprof_average_us("label_spmm_mfma_kernel"),
                                                         "label_spmm_reduce_kernel": rocprof_average_us("label_spmm_reduce_kernel"),
                                                         "transpose_w_rowmajor_kernel": rocprof_average_us("transpose_w_rowmajor_kernel")}}
        # dense driver: the symmetric-product kernel of the tridiagonalisation, launched once per
        # column j; algorithmic bytes of launch j = 8 * (n-j-1)^2 / 2 (the LOWER triangle of the
        # trailing matrix is read once), 8*n*(2n-1)/12 on average
        ms = prof(5, n)
        avg_bytes = 8.0 * n * (2 * n - 1) / 12.0
        gbs = avg_bytes / (ms * 1e-3) / 1e9
        tr, src = pmc_traffic("sytrd_symv")
        kernels["sytrd_symv"] = {"ms": round(ms, 5), "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                                 "bound": "hbm", "launches_per_reduction": n - 1, "algorithmic_bytes_per_launch": round(avg_bytes), "traffic": tr, "traffic_source": src,
                                 "note": "dense driver only; back-to-back launches from the host (not graph-replayed): includes host launch cost"}
        ms6 = prof(6, n, reps=2)
        kernels["sytrd_total"] = {"ms": round(ms6, 3), "note": "whole tridiagonalisation (graph replay): panel form (symv + form + MFMA syr2k launches) for the first n - 2048 columns, one launch per column (sytrd_row_kernel) for the last 2048"}
        # the row form alone at the orders where the dense driver is the default: 16 (n-j)^2 bytes per column
        # (trailing matrix read + written once), n^3 / 3 * 16 per reduction
        for nr in (1024, 2048):
            msr = prof(6, nr, reps=3)
            by = 16.0 * nr ** 3 / 3.0
            kernels["sytrd_rows_n%d" % nr] = {"ms": round(msr, 3), "us_per_column": round(msr * 1e3 / (nr - 1), 2), "bound": "latency" if nr <= 1024 else "hbm",
                                              "achieved": round(by / (msr * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(by / (msr * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                              "algorithmic_bytes_per_reduction": round(by), "note": "sytrd_row_kernel: one launch per column, graph replay"}
        # `roofline`: the kernel that carries the O(N^3) work of the default path and its only
        # MFMA-bound one, launched as the product path launches it: TP channels, lower-triangle tiles
        # of the symmetric product.  frac = EXECUTED ops / peak (hardware efficiency); the figure
        # judged against the algorithmic 2*N^3 per square is reported beside it.
        ms_tri = prof(0, n, aux=100 + TP)
        # executed multiply-adds of that launch.  Persistent launch (kernels_gemm_sym.hip; chosen by i8_symsquare_pays,
        # restated here): 256 x 256 macro-tiles of the lower triangle, the diagonal ones only their 36 lower 32 x 32
        # blocks of 64; else the 128 x 128 tiles of kernels_gemm.hip: (T + 1) / (2 T) of the square
        mt = (n + 255) // 256
        jobs = TP * mt * (mt - 1) // 2 + (TP * mt + 1) // 2
        rounds = -(-jobs // 256)
        persistent = (4 * jobs >= 3 * 256) and (100 * jobs >= 85 * rounds * 256) and args.square_kernel != 1
        if persistent or args.square_kernel == 64:
            exec_frac = (64.0 * mt * (mt - 1) / 2 + 36.0 * mt) / (64.0 * mt * mt) * (256.0 * mt / n) ** 2
            kname, kpat = "i8_symsquare_kernel (persistent: 256 x 256 macro-tiles of the lower triangle, diagonal tiles two per job)", "i8_symsquare_kernel"
        else:
            Tt = (n + 127) // 128
            exec_frac = (Tt + 1) / (2.0 * Tt)
            kname, kpat = "gemm_tn_dma_kernel<i8> (128 x 128 lower-triangle tiles)", "gemm_tn_dma_kernel<0,"
        alg_rate = TP * flops / (ms_tri * 1e-3) / 1e12
        tr, src = pmc_traffic(f"i8x{TP}_lower")
        ki8 = kernels["square_i8"]
        roof = {"kernel": f"{kname}; random squares: {TP} channels per launch of the symmetric product",
                "bound": "mfma", "achieved": round(alg_rate * exec_frac, 2), "peak": I8_MFMA_PEAK_TOPS, "unit": "TOP/s",
                "frac": round(alg_rate * exec_frac / I8_MFMA_PEAK_TOPS, 4), "traffic": tr, "traffic_source": src,
                "ms_per_launch": round(ms_tri, 4), "channels_per_launch": TP, "executed_ops_per_launch": TP * flops * exec_frac,
                "algorithmic_ops_per_launch": TP * flops, "algorithmic_rate_2N3": round(alg_rate, 2),
                "algorithmic_frac_2N3": round(alg_rate / I8_MFMA_PEAK_TOPS, 4),
                "algorithmic_bytes_per_launch": TP * (n * n + 4 * n * n),
                "rocprof_avg_us": rocprof_average_us(kpat),
                "full_square_kernel": {"ms_per_launch": ki8["ms_per_launch"], "achieved": ki8["achieved"], "frac": ki8["frac"]},
                "algorithmic": f"2*N^3 int8 multiply-adds (as ops) per channel and square (SURVEY 8d); one launch = {TP} channels (the "
                               f"reference does ONE square per iteration: {TP} channels + one confirm round are this build's redundancy for 8-bit draws); "
                               "executed = the lower-triangle tiles' share of that (see executed_ops_per_launch); algorithmic bytes per launch = channels x "
                               "(N^2 int8 read + N^2 int32 written)"}
        # Clock under this launch (round 5, profiles/r05_power_under_kernels.txt): hwmon reads 1302 W of the 1400 W cap and a
        # median sclk of 1938 MHz while the kernel loops; GRBM_GUI_ACTIVE per launch over the same launch's duration gives
        # 2081-2169 MHz.  The in-kernel clock64 sampler of rounds 2-4 read 1.5-1.7 GHz -- and slowed the kernel by 8 % while
        # it ran: it is the outlier, and `frac_at_measured_clock` (0.51-0.53 in rounds 3-4) went with it.  At the ~2.08 GHz
        # the two hardware readings agree on, frac is 0.42 of the peak at the running clock; the nominal-clock figure is `frac`.
        roof["clock_under_launch"] = {"hwmon_sclk_mhz_median": 1938, "grbm_cycles_over_duration_mhz": [2081, 2169], "socket_power_w_median": 1302,
                                      "power_cap_w": 1400, "source": "profiles/r05_power_under_kernels.txt (tools/gpu/clock_story.sh)",
                                      "frac_at_2080_mhz": round(roof["frac"] * 2400.0 / 2080.0, 4)}
        cpu_n = (n if n <= 4096 else 0) if args.cpu_n < 0 else args.cpu_n  # (N = 8192 on the CPU takes ~7 minutes: the theta leg below stands in)
        if cpu_n > 0:
            samples = [cpu_baseline(pr, cpu_n, seed=1) for _ in range(max(1, args.cpu_samples))]
            samples.sort(key=lambda c: c["adm_s"] + c["bd_s"])
            cb = samples[len(samples) // 2]
            tot = cb["adm_s"] + cb["bd_s"]
            cpu = {"value": round(1.0 / tot, 6), "unit": "reductions/s", "cores": cb["threads"], "kind": "port", "n": cb["n"],
                   "samples": [round(c["adm_s"] + c["bd_s"], 2) for c in samples],
                   "sample": f"{'median of ' + str(len(samples)) + ' full oracle reductions' if len(samples) > 1 else 'ONE full oracle reduction'} (NumPy/SciPy restatement, not Julia; "
                             f"--cpu-samples K takes the median of K) of the headline instance at N={cb['n']}, dim {cb['dim']}: "
                             f"admissible_subspace {cb['adm_s']:.2f} s + blockDiagonalize {cb['bd_s']:.2f} s, measured (no extrapolation)"}
            # like for like: the CPU leg runs the reference's algorithm (dense eigh on the N x N generic element); the
            # GPU headline runs module compression, the commutative basis_image shortcut and the verify shortcut --
            # algorithmic shortcuts the reference does not have.  The GPU number on the reference's own algorithm is
            # variants.dense_eigensolver.
            de = variants.get("dense_eigensolver", {}).get("value")
            cpu["like_for_like"] = {"gpu_dense_eigensolver_over_cpu": round(de * tot, 1) if de else None,
                                    "gpu_headline_over_cpu": round(total_red / dt * tot, 1),
                                    "note": "the headline path uses module compression (a w = dim(P) eigenproblem instead of the reference's dense "
                                            "N x N eigh); the first ratio compares like with like (eig_driver = 4 against the CPU restatement)"}
        if args.cpu_n != 0:
            # the instance that ITERATES, at a reduced order: the loop side on the CPU.  Also the whole CPU leg of the orders
            # the full oracle reduction is not run at (--n 8192): cpu_baseline is never null in a default run
            try:
                ct = cpu_baseline_theta(pr, 1024)
                leg = {"value": round(1.0 / (ct["adm_s"] + ct["bd_s"]), 5), "unit": "reductions/s", "n": ct["n"], "dim": ct["dim"],
                       "sample": f"one oracle reduction of theta' of C_32 [] K_32 (N = 1024, 5 loop iterations): admissible_subspace "
                                 f"{ct['adm_s']:.2f} s + blockDiagonalize {ct['bd_s']:.2f} s"}
            except Exception as exc:  # noqa: BLE001
                leg = {"error": repr(exc)[:200]}
            if cpu is None:
                try:
                    from threadpoolctl import threadpool_info
                    thr = max([p_.get("num_threads", 1) for p_ in threadpool_info()] + [1])
                except Exception:  # noqa: BLE001
                    thr = os.cpu_count() or 1
                cpu = {"value": leg.get("value"), "unit": "reductions/s", "cores": thr, "kind": "port", "n": 1024,
                       "sample": "the full oracle reduction is not run at this order (minutes); value = the theta' instance at N = 1024: " + leg.get("sample", "failed")}
            cpu["theta_c32xk32_n1024"] = leg
    if rank == 0:
        out = {
            "metric": f"N x N SDP reductions/sec (admissible_subspace+blockDiagonalize) at N={n}",
            "value": round(total_red / dt, 4), "unit": "reductions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"i8": "int8 square (int32 acc) + f64 eigen", "f32": "f32 square + f64 eigen", "f64": "f64"}[args.mode],
            "data": "synthetic",
            "config": {"workload": f"configs[3]: synthetic Jordan algebra N={w0.n}, {w0.d} basis matrices, instance '{args.workload}' ({w0.note}), "
                                   f"square_mode={args.mode}, {args.channels or 2} channels" + ("" if args.channels else " + 1 confirm round"), "N": w0.n, "dim": w0.d, "restarts_per_step": world * RPG, "restarts_per_gpu": RPG,
                       "restarts_not_delivered": failed_restarts,
                       "iterations_per_reduction": acc_timed.iters / max(1, args.steps)},
            "phase_ms_per_step": phases(acc_timed, args.steps),
            "phase_ms_source": "HIP events inside the timed steps" if args.timers_in_timed_region else
                               "an instrumented pass of the same steps right after the timed region (the timed steps carry no phase events)",
            "partition_meets": {"warmup": meets_warmup, "timed": acc.meets,
                                "note": "agreement steps in which the ranks' partitions differed and the hash-meet ran (rank 0's count)"},
            "block_diagonalization_winner": {"adopted_warmup": acc_warm.bd_adopted, "adopted_timed": acc.bd_adopted,
                                             "all_failed": acc_warm.bd_all_failed + acc.bd_all_failed,
                                             "note": "multi-rank runs (SURVEY 8(e)(ii)): steps in which this rank's blockDiagonalize failed and it adopted the "
                                                     "block sizes of the lowest rank that succeeded (rank 0's count) / steps in which every rank failed and all drew again"},
            "setup_ms": {"host_setup": round(w0.setup_ms, 1), "upload": round(w0.upload_ms, 1),
                         "note": "outside the timed region: admissible_setup on the host (QR of A', C_L, min-norm x0: src/partitions.jl:117-142) and the upload of "
                                 "C_L, X0_L, U; variants.host_arrays_* time the calls that carry the upload"},
            "workloads": workloads, "roofline": roof, "kernels": kernels, "variants": variants, "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
