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


def gather_rows(local, group=None, dst: Optional[int] = None, counts: Optional[Sequence[int]] = None):
    """Concatenate every rank's ``(rows_r, C)`` tensor along axis 0, in rank order

    Equal row counts take one ``all_gather_into_tensor`` (a single RCCL collective per call:
    on xGMI the per-link bandwidth, not launch count, bounds it, so shards are gathered whole).
    Ragged row counts: every rank's block is broadcast straight into its rows of the result
    (``world`` broadcasts queued back to back; no padding to the longest shard, no second copy).
    `counts`: the row count of every rank when the caller knows them already (saves the exchange).
    With `dst` set, only that rank returns the result (others return None) -- the collectives
    are the same; the feature matrices are small next to the audio they came from.

    Works on CPU tensors with the ``gloo`` backend (tests) and on GPU tensors with ``nccl`` -- the
    device buffer the kernels wrote is what is gathered, nothing passes through the host.
    The C ABI has the same operation for callers without torch: ``pds_gather_rows``.
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
    rank = dist.get_rank(group)
    if counts is None:
        gathered = torch.empty(world, dtype=torch.int64, device=local.device)
        dist.all_gather_into_tensor(
            gathered, torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device), group=group
        )
        counts = gathered.cpu().tolist()
    counts = [int(c) for c in counts]
    if len(counts) != world or counts[rank] != local.shape[0]:
        raise ValueError("counts do not describe this rank's rows")
    C = local.shape[1]
    out = torch.empty((sum(counts), C), dtype=local.dtype, device=local.device)
    if min(counts) == max(counts):
        if counts[0]:
            dist.all_gather_into_tensor(out, local, group=group)
    else:
        bounds = np.concatenate([[0], np.cumsum(counts)])
        out[bounds[rank] : bounds[rank + 1]] = local
        for r in range(world):
            if counts[r]:
                dist.broadcast(out[bounds[r] : bounds[r + 1]], src=dist.get_global_rank(group, r) if group else r,
                               group=group)
    if dst is not None and rank != dst:
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

    Every rank passes a list of the same length; only the rank's own block ``signals[lo:hi]``
    is read (the other entries may be placeholders).  Returns

    * with ``gather=True``: the list of all feature matrices (numpy), on every rank;
    * with ``gather=False``: ``(lo, feats)`` -- the rank's block start and its own list.

    The rank's utterances are packed into one device buffer, one launch computes their features,
    and that device buffer is what the ranks exchange (:func:`gather_rows`); the per-utterance
    frame counts and the feature dtype travel in one small all-gather in front of it.
    """
    import torch
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    else:
        world, rank = 1, 0
    lo, hi = shard_bounds(len(signals), world, rank)
    mine = [np.asarray(s) for s in signals[lo:hi]]
    if not gather or world == 1:
        feats = computer.compute_full_batch(mine) if mine else []
        return feats if gather else (lo, feats)
    on_gpu = dist.get_backend(group) == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")
    C = computer.num_coeffs
    packed = getattr(computer, "compute_packed", None)
    if mine and on_gpu and packed is not None and mine[0].dtype in (np.float32, np.float64):
        lengths = np.asarray([len(s) for s in mine], dtype=np.int64)
        offsets = np.concatenate([[0], np.cumsum(lengths)[:-1]]).astype(np.int64)
        flat = np.concatenate(mine) if lengths.sum() else np.zeros(0, mine[0].dtype)
        local, row_off = packed(torch.from_numpy(flat).to(dev), offsets, lengths)
        own_counts = np.diff(np.asarray(row_off, dtype=np.int64))
    elif mine:  # (CPU groups in tests, computers without a packed interface, integer samples)
        feats = computer.compute_full_batch(mine)
        own_counts = np.asarray([f.shape[0] for f in feats], dtype=np.int64)
        local = torch.from_numpy(np.ascontiguousarray(np.concatenate(feats))).to(dev)
    else:
        local, own_counts = None, np.zeros(0, np.int64)
    # one all-gather of [dtype code, frame counts of the rank's utterances (padded to the largest
    # shard)]: a rank with an empty shard learns the dtype of the others' features from it
    codes = {torch.float32: 1, torch.float64: 2}
    largest = shard_bounds(len(signals), world, 0)
    largest = largest[1] - largest[0]
    note = torch.zeros(1 + largest, dtype=torch.int64)
    note[0] = codes[local.dtype] if local is not None else 0
    note[1 : 1 + len(own_counts)] = torch.from_numpy(own_counts)
    notes = torch.empty(world * (1 + largest), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(notes, note.to(dev), group=group)
    notes = notes.cpu().numpy().reshape(world, 1 + largest)
    code = int(notes[:, 0].max())
    if code == 0:
        return [np.zeros((0, C), np.float32) for _ in signals]
    if any(c not in (0, code) for c in notes[:, 0]):
        raise ValueError("ranks hold signals of different dtypes")
    dtype = torch.float32 if code == 1 else torch.float64
    if local is None:
        local = torch.zeros((0, C), dtype=dtype, device=dev)
    counts: List[int] = []
    for r in range(world):
        rlo, rhi = shard_bounds(len(signals), world, r)
        counts.extend(int(c) for c in notes[r, 1 : 1 + rhi - rlo])
    per_rank = [int(notes[r, 1:].sum()) for r in range(world)]
    rows = gather_rows(local, group=group, counts=per_rank).cpu().numpy()
    bounds = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    return [rows[bounds[i] : bounds[i + 1]] for i in range(len(signals))]
