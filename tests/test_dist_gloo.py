"""world_size-2 gloo tests (CPU) of the restart-agreement step used by bench.py --gpus N."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from __graft_entry__ import load_package
    pkg = load_package()
    from sdpsr_amd import parallel as par  # noqa
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 24
    Ls, d = pkg.problems.synthetic_jordan_partition(n, seed=3)
    flat = torch.from_numpy(np.ascontiguousarray(Ls.ravel(order="F")).astype(np.int64))
    # 1) equal canonical labels on every rank: agreement without a meet
    ok, out = par.agree_partition(flat.clone(), par.relabel_numpy)
    res = {"agree_equal": ok and bool((out == flat).all())}
    # 2) rank 1 missed a split (merges classes 1 and 2): the meet restores the finer partition
    mine = flat.clone()
    if rank == 1:
        mine[mine == 2] = 1
        # keep it canonical like a real (coarser) result would be
        mine, _ = par.relabel_numpy(mine)
    ok2, meet = par.agree_partition(mine, par.relabel_numpy)
    res["meet_is_finer"] = (not ok2) and bool((meet == flat).all())
    # 3) zero class survives the hash combination
    z = flat.clone()
    z[:7] = 0
    if rank == 1:
        z[z == 3] = 4
        z, _ = par.relabel_numpy(z)
    else:
        z, _ = par.relabel_numpy(z)
    _, meet2 = par.agree_partition(z, par.relabel_numpy)
    res["zero_kept"] = bool((meet2[:7] == 0).all()) and bool((meet2[7:] != 0).all())
    res["seeds_differ"] = par.restart_seed(5, 0) != par.restart_seed(5, 1)
    # 4) SURVEY 8(e)(ii): blockDiagonalize failed on rank 0 (DimensionMismatch = 3), rank 1 succeeded: the lowest rank
    #    with status 0 wins, its block sizes and Q_hat are broadcast
    sizes = [2, 2, 2, 2, 3] if rank == 1 else [1, 1]
    qmine = torch.full((6, 11), float(rank + 1), dtype=torch.float64)
    win, got_sizes, qh = par.agree_block_diagonalization(3 if rank == 0 else 0, sizes, q_hat=lambda sz: qmine if rank == 1 else torch.zeros(6, sum(sz), dtype=torch.float64))
    res["winner_is_rank1"] = win == 1 and got_sizes == [2, 2, 2, 2, 3] and bool((qh == 2.0).all())
    # both succeed: rank 0 wins; both fail: -1, the caller retries
    win0, s0, _ = par.agree_block_diagonalization(0, [5 + rank], group=None)
    res["lowest_rank_wins"] = win0 == 0 and s0 == [5]
    winx, sx, _ = par.agree_block_diagonalization(2, [1])
    res["all_failed"] = winx == -1 and sx is None
    # more block sizes than the one-collective record holds (QAP-type partitions have a few dozen blocks; 100 here): every
    # rank must fall through to the all-reduce + broadcast form together, also the rank whose own list is short / failed
    big = list(range(1, 101))
    winb, sb, _ = par.agree_block_diagonalization(0 if rank == 1 else 3, big if rank == 1 else [7])
    res["long_form"] = winb == 1 and sb == big
    winc, sc, _ = par.agree_block_diagonalization(0, big if rank == 0 else [7, 7])
    res["long_form_rank0_wins"] = winc == 0 and sc == big
    # 5) R restarts per rank (bench.py --restarts-per-gpu R): the R x world table of checksums.  All equal: agreement
    #    without a meet.  Restart 1 of rank 1 missed a split: the meet over all four restarts -- across the ranks and
    #    within rank 1 -- restores the finer partition on every rank.  A miss inside ONE rank only (rank 0's two restarts
    #    differ, rank 1's agree with rank 0's first) is caught as well.
    okr, outr = par.agree_partitions([flat.clone(), flat.clone()], par.relabel_numpy)
    res["table_equal"] = okr and bool((outr == flat).all())
    coarse = flat.clone()
    coarse[coarse == 2] = 1
    coarse, _ = par.relabel_numpy(coarse)
    mine2 = [flat.clone(), coarse.clone() if rank == 1 else flat.clone()]
    okm, meetr = par.agree_partitions(mine2, par.relabel_numpy)
    res["table_meet_across_ranks"] = (not okm) and bool((meetr == flat).all())
    mine3 = [flat.clone(), coarse.clone() if rank == 0 else flat.clone()]
    okw, meetw = par.agree_partitions(mine3, par.relabel_numpy)
    res["table_meet_within_rank"] = (not okw) and bool((meetw == flat).all())
    # two different coarsenings on the two ranks: the meet is finer than both (= the common refinement)
    c2 = flat.clone()
    c2[c2 == 4] = 3
    c2, _ = par.relabel_numpy(c2)
    okx, meetx = par.agree_partitions([coarse.clone() if rank == 0 else c2.clone()] * 2, par.relabel_numpy)
    res["table_meet_of_two_coarsenings"] = (not okx) and bool((meetx == flat).all())
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_agreement_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, res in got:
        assert all(res.values()), (rank, res)
