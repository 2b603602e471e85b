"""Independent random restarts across the GPUs of one node.

One process per GPU (``torch.distributed``; backend "nccl" is RCCL on ROCm, "gloo" in the CPU
tests).  Every rank runs the reduction with its own seed; the canonical label matrices are
equal with probability 1.  The ranks first compare a 128-bit checksum of their label matrix
(``sdpsr_partition_checksum`` on the device, two int64 words all-gathered): a reduction takes
~2.5 ms at N=4096, two all-reduces over the 64 MiB labels would cost about as much over xGMI.
Only if the checksums differ (a split missed by one rank's draws) the labels themselves travel:
MIN/MAX all-reduce to confirm, then the meet of the partitions -- a universal hash
``sum_r a_r * label_r mod 2^64`` summed with one more all-reduce and canonically relabelled.
"""
from __future__ import annotations

import numpy as np

_ODD = [0x9E3779B97F4A7C15, 0xBF58476D1CE4E5B9, 0x94D049BB133111EB, 0xD6E8FEB86659FD93,
        0xC2B2AE3D27D4EB4F, 0x165667B19E3779F9, 0x27D4EB2F165667C5, 0x85EBCA77C2B2AE63]


_BD_RECORD = 64  # int64 words of a rank's record in agree_block_diagonalization: status, count, <= 62 block sizes


def restart_seed(base_seed: int, rank: int, step: int = 0) -> int:
    return (base_seed * 0x9E3779B97F4A7C15 + rank * 0xD1342543DE82EF95 + step) & (2 ** 64 - 1)


def _signed(x):
    x &= 2 ** 64 - 1
    return x - 2 ** 64 if x >= 2 ** 63 else x


def torch_checksum(labels):
    """Position-weighted checksum with plain torch ops (wraps mod 2^64): the stand-in for
    ``sdpsr_partition_checksum`` where the HIP library is not loaded (CPU gloo tests)."""
    import torch
    idx = torch.arange(labels.numel(), dtype=torch.int64, device=labels.device)
    l = labels.reshape(-1).to(torch.int64) + 1
    # same formula and constants as labels_checksum_kernel (csrc/kernels_partition.hip)
    h1 = (l * (idx * _signed(0x9E3779B97F4A7C15) + _signed(0xD1342543DE82EF95))).sum()
    h2 = ((l * l + _signed(0x27D4EB2F165667C5)) *
          ((idx ^ (idx >> 13)) * _signed(0xBF58476D1CE4E5B9) + _signed(0x94D049BB133111EB))).sum()
    return int(h1.item()), int(h2.item())


def checksums_agree(words, group=None, device=None):
    """All-gather two 64-bit words per rank; True iff every rank holds the same pair."""
    import torch
    import torch.distributed as dist
    if device is not None and torch.device(device).type != "cpu" and dist.get_backend(group) == "gloo":
        device = None  # gloo has no all_gather for device tensors (the CPU-backend test hook of bench.py): 16 bytes via the host
    mine = torch.tensor([_signed(words[0]), _signed(words[1])], dtype=torch.int64, device=device)
    world = dist.get_world_size(group)
    got = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(got, mine, group=group)
    return all(bool((g == got[0]).all()) for g in got)


def agree_partition(labels, relabel, group=None, checksum=None):
    """labels: flat torch integer tensor (column-major label matrix of this rank).
    relabel(sig_int64_tensor) -> (labels_tensor, nparts): canonical relabel of arbitrary 64-bit
    keys (0 stays 0) -- on the GPU this is the refine kernel.
    checksum(labels) -> (word0, word1): defaults to ``torch_checksum``; on the GPU pass
    ``lambda t: pkg.partition_checksum(t, ctx)``.
    Returns (agreed_without_meet, labels)."""
    import torch
    import torch.distributed as dist
    words = (checksum or torch_checksum)(labels)
    if checksums_agree(words, group=group, device=labels.device):
        return True, labels
    lo = labels.clone()
    hi = labels.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    if bool((lo == hi).all()):
        return True, labels
    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    # zero must stay zero only where every rank has zero: hash label+1 and subtract the all-zero key
    a = _signed(_ODD[rank % len(_ODD)] * (2 * (rank // len(_ODD)) + 1))
    sig = (labels.to(torch.int64) + 1) * a  # wraps mod 2^64
    dist.all_reduce(sig, op=dist.ReduceOp.SUM, group=group)
    zero_key = 0
    for r in range(world):
        zero_key += _ODD[r % len(_ODD)] * (2 * (r // len(_ODD)) + 1)
    sig = sig - _signed(zero_key)
    new_labels, _ = relabel(sig)
    return False, new_labels


def _slot_multiplier(k):
    return _ODD[k % len(_ODD)] * (2 * (k // len(_ODD)) + 1)


def agree_partitions(labels_list, relabel, group=None, checksum=None):
    """The agreement step for R restarts PER RANK (``bench.py --restarts-per-gpu R``, ``sdpsr_jordan_reduce_batch``): every
    rank holds R label tensors (independent draws of src/partitions.jl:154-185).  The R x world table of 128-bit
    checksums is all-gathered (2 R words per rank); if all its rows are equal the restarts agree -- across ranks AND
    within each rank.  Otherwise every restart's labels enter the meet: a universal hash with one odd multiplier per
    (rank, restart) slot, summed over the rank's own restarts locally and over the ranks with ONE all-reduce, then the
    canonical relabel -- a restart whose draws missed a split is refined by the others, whichever rank it ran on.
    Returns (agreed_without_meet, labels); after a meet every restart adopts ``labels``."""
    import torch
    import torch.distributed as dist
    R = len(labels_list)
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = labels_list[0].device
    words = []
    for t in labels_list:
        w0, w1 = (checksum or torch_checksum)(t)
        words += [_signed(w0), _signed(w1)]
    gdev = None if (dev.type != "cpu" and dist.get_backend(group) == "gloo") else dev
    mine = torch.tensor(words, dtype=torch.int64, device=gdev)
    got = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(got, mine, group=group)
    table = torch.stack(got).reshape(world * R, 2).cpu()
    if bool((table == table[0]).all()):
        return True, labels_list[0]
    sig = None
    for i, t in enumerate(labels_list):
        term = (t.to(torch.int64) + 1) * _signed(_slot_multiplier(rank * R + i))  # wraps mod 2^64
        sig = term if sig is None else sig + term
    dist.all_reduce(sig, op=dist.ReduceOp.SUM, group=group)
    zero_key = sum(_slot_multiplier(k) for k in range(world * R))
    new_labels, _ = relabel(sig - _signed(zero_key))
    return False, new_labels


def agree_block_diagonalization(status, blk_sizes, q_hat=None, group=None, device=None):
    """``blockDiagonalize`` is randomized and the reference's answer to ``NumericalInconsistency`` /
    ``DimensionMismatch`` is "try again" (src/eigen_decomposition.jl:264-270, src/diagonalize.jl:4-9).  With one restart
    per rank the tries have already run side by side: the LOWEST rank whose status is 0 wins and every rank ends up with
    its ``blkSizes`` -- one all-gather of a 64-word record per rank (status, count, sizes) in the usual case, a one-integer MIN
    all-reduce + two broadcasts when a rank has more than 62 blocks -- (and, if given, its ``Q_hat`` tensor, which must have the same shape on
    every rank once the sizes are known -- pass ``q_hat`` as a callable ``sizes -> tensor`` to allocate it late).
    Returns (winner_rank, blk_sizes, q_hat); winner_rank = -1 when every rank failed (the caller retries with fresh
    draws).  ``status``: this rank's sdpsr status (0 = ok); ``blk_sizes``: this rank's sizes (ignored unless it wins)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if device is not None and torch.device(device).type != "cpu" and dist.get_backend(group) == "gloo":
        device = None
    sizes_in = [int(x) for x in blk_sizes] if int(status) == 0 else []
    if len(sizes_in) <= _BD_RECORD - 2:
        # the usual case in ONE collective: every rank contributes a fixed record [status, count, sizes ...]; each rank then
        # reads the winner's sizes out of the gathered records itself (a MIN all-reduce + two broadcasts would be three
        # latencies of a small collective per reduction, ~10 % of a 0.9 ms step over xGMI)
        rec = torch.zeros(_BD_RECORD, dtype=torch.int64, device=device)
        rec[0] = int(status)
        rec[1] = len(sizes_in)
        if sizes_in:
            rec[2:2 + len(sizes_in)] = torch.as_tensor(sizes_in, dtype=torch.int64)
        got = [torch.empty_like(rec) for _ in range(world)]
        dist.all_gather(got, rec, group=group)
        recs = torch.stack(got).cpu()
        if bool((recs[:, 1] >= 0).all()):  # (no rank took the long form: count = -1 marks it)
            ok = [r for r in range(world) if int(recs[r, 0]) == 0]
            if not ok:
                return -1, None, None
            winner = ok[0]
            out_sizes = [int(x) for x in recs[winner, 2:2 + int(recs[winner, 1])].tolist()]
            q = None
            if q_hat is not None:
                q = q_hat(out_sizes) if callable(q_hat) else q_hat
                dist.broadcast(q, src=dist.get_global_rank(group, winner) if group is not None else winner, group=group)
            return winner, out_sizes, q
    else:
        # more block sizes than a record holds: tell the others (count = -1), then the three-step form below
        rec = torch.zeros(_BD_RECORD, dtype=torch.int64, device=device)
        rec[0] = int(status)
        rec[1] = -1
        got = [torch.empty_like(rec) for _ in range(world)]
        dist.all_gather(got, rec, group=group)
    pick = torch.tensor([rank if int(status) == 0 else world], dtype=torch.int64, device=device)
    dist.all_reduce(pick, op=dist.ReduceOp.MIN, group=group)
    winner = int(pick.item())
    if winner >= world:
        return -1, None, None
    src = dist.get_global_rank(group, winner) if group is not None else winner
    cnt = torch.tensor([len(blk_sizes) if rank == winner else 0], dtype=torch.int64, device=device)
    dist.broadcast(cnt, src=src, group=group)
    sizes = torch.zeros(int(cnt.item()), dtype=torch.int64, device=device)
    if rank == winner:
        sizes.copy_(torch.as_tensor([int(x) for x in blk_sizes], dtype=torch.int64))
    dist.broadcast(sizes, src=src, group=group)
    out_sizes = [int(x) for x in sizes.cpu().tolist()]
    q = None
    if q_hat is not None:
        q = q_hat(out_sizes) if callable(q_hat) else q_hat
        dist.broadcast(q, src=src, group=group)
    return winner, out_sizes, q


def relabel_numpy(sig):
    """CPU canonical relabel used by the gloo tests (first occurrence order, 0 stays 0)."""
    import torch
    flat = sig.cpu().numpy().astype(np.int64)
    uniq, first, inv = np.unique(flat, return_index=True, return_inverse=True)
    order = np.argsort(first, kind="stable")
    rank = np.zeros(len(uniq), dtype=np.int64)
    k = 0
    for u in order:
        if uniq[u] == 0:
            continue
        k += 1
        rank[u] = k
    return torch.from_numpy(rank[inv.reshape(-1)]).to(sig.device), k
