"""Utterance sharding across the GPUs of a node, and the gather of the feature matrices.

The hot path shards embarrassingly: utterances share nothing (no cross-utterance state in
``compute_full``, reference compute.py:574-607; local CMVN and deltas are per utterance).
One process per GPU (``torch.distributed``, backend ``"nccl"`` = RCCL over xGMI on the GPU
box, ``"gloo"`` in CPU tests); rank r takes a contiguous block of utterances so that the
concatenation of the shards keeps the input order.  The only collective is the optional
gather of the ``(rows, num_coeffs)`` feature matrices.

The reference has no distributed code at all (its only parallelism is DataLoader worker
processes, command_line.py:594); this module is the node-level driver the north star asks for.
"""
from typing import List, Optional, Sequence, Tuple

import numpy as np

__all__ = ["all_reduce_stats", "gather_rows", "shard_bounds", "compute_full_sharded"]


def shard_bounds(num_items: int, world_size: int, rank: int) -> Tuple[int, int]:
    """``[lo, hi)`` of the contiguous block of items owned by `rank`

    Blocks differ in size by at most one item; earlier ranks get the larger blocks.
    """
    if not 0 <= rank < world_size:
        raise ValueError("rank out of range")
    base, extra = divmod(num_items, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_rows(local, group=None, dst: Optional[int] = None):
    """Concatenate every rank's ``(rows_r, C)`` tensor along axis 0, in rank order

    Equal row counts take one ``all_gather_into_tensor`` (a single RCCL collective per call:
    on xGMI the per-link bandwidth, not launch count, bounds it, so shards are gathered whole).
    Ragged row counts first exchange the counts, then gather shards padded to the longest.
    With `dst` set, only that rank returns the result (others return None) -- the collective
    is the same all-gather; the feature matrices are small next to the audio they came from.

    Works on CPU tensors with the ``gloo`` backend (tests) and on GPU tensors with ``nccl``.
    """
    import torch
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized():
        return local
    world = dist.get_world_size(group)
    if world == 1:
        return local
    if local.dim() != 2:
        raise ValueError("expected a 2-D (rows, coeffs) tensor")
    local = local.contiguous()
    counts = torch.empty(world, dtype=torch.int64, device=local.device)
    dist.all_gather_into_tensor(
        counts, torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device), group=group
    )
    counts = counts.cpu().tolist()
    longest, C = max(counts), local.shape[1]
    if min(counts) == longest:
        out = torch.empty((world * longest, C), dtype=local.dtype, device=local.device)
        if longest:
            dist.all_gather_into_tensor(out, local, group=group)
    else:
        padded = torch.zeros((longest, C), dtype=local.dtype, device=local.device)
        padded[: local.shape[0]] = local
        stacked = torch.empty((world * longest, C), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(stacked, padded, group=group)
        out = torch.cat([stacked[r * longest : r * longest + counts[r]] for r in range(world)])
    if dst is not None and dist.get_rank(group) != dst:
        return None
    return out


def all_reduce_stats(standardize, group=None) -> None:
    """Sum the statistics a ``Standardize`` (``CMVN``) accumulated on every rank, in place

    Global (corpus-level) normalisation with the corpus sharded by utterance: every rank calls
    ``accumulate`` on its own feature matrices (reference post.py:193-212), then this -- one
    all-reduce of the float64 ``[2, C + 1]`` table of sums, sums of squares and the count
    (about 1 KB) -- after which ``apply`` uses the same statistics on every rank.  A rank that
    accumulated nothing (an empty shard) contributes zeros.
    """
    import torch
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else "cpu"
    stats = standardize._stats
    width = torch.tensor([0 if stats is None else stats.shape[1]], dtype=torch.int64, device=dev)
    widest = width.clone()
    dist.all_reduce(widest, op=dist.ReduceOp.MAX, group=group)
    widest = int(widest.item())
    # every rank takes part in both collectives before anyone raises, so a mismatch cannot
    # leave the others waiting
    ok = torch.tensor([int(stats is None or stats.shape[1] == widest)], dtype=torch.int64, device=dev)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
    if not int(ok.item()):
        raise ValueError("ranks accumulated statistics of different widths")
    if widest == 0:
        return
    table = torch.zeros((2, widest), dtype=torch.float64) if stats is None else torch.from_numpy(
        np.ascontiguousarray(stats, dtype=np.float64)
    )
    table = table.to(dev)
    dist.all_reduce(table, op=dist.ReduceOp.SUM, group=group)
    standardize._stats = table.cpu().numpy()


def compute_full_sharded(computer, signals: Sequence, gather: bool = True, group=None):
    """``compute_full`` of `signals` with the work split over the ranks of `group`

    Every rank passes the same list of host signals (or just its own needs to be valid:
    only ``signals[lo:hi]`` of the rank's block are touched).  Returns

    * with ``gather=True``: the list of all feature matrices (numpy), on every rank;
    * with ``gather=False``: ``(lo, feats)`` -- the rank's block start and its own list.
    """
    import torch
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    else:
        world, rank = 1, 0
    lo, hi = shard_bounds(len(signals), world, rank)
    mine = [np.asarray(s) for s in signals[lo:hi]]
    feats = computer.compute_full_batch(mine) if mine else []
    if not gather or world == 1:
        return feats if gather else (lo, feats)
    C = computer.num_coeffs
    dtype = mine[0].dtype if mine else np.float32
    local = np.concatenate(feats) if feats else np.zeros((0, C), dtype)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else "cpu"
    rows = gather_rows(torch.from_numpy(np.ascontiguousarray(local)).to(dev), group=group)
    rows = rows.cpu().numpy()
    # frame counts are a pure function of the lengths, so every rank can split the rows
    counts = [computer.num_frames(len(s)) for s in signals]
    bounds = np.concatenate([[0], np.cumsum(counts)])
    return [rows[bounds[i] : bounds[i + 1]] for i in range(len(signals))]
