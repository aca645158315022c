"""Sharding and gather logic with world_size 2 over gloo (CPU, no GPU needed)."""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from pydrobert_speech_amd.dist import shard_bounds
from tests.conftest import ROOT


def test_shard_bounds_cover_everything_in_order():
    for n in (0, 1, 7, 8, 9, 1024, 65536 + 3):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
    with pytest.raises(ValueError):
        shard_bounds(4, 2, 2)


WORKER = textwrap.dedent(
    """
    import os, sys
    import numpy as np
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.environ["PDS_ROOT"])
    from pydrobert_speech_amd.dist import gather_rows, shard_bounds, compute_full_sharded

    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    # equal shards
    local = torch.full((5, 3), float(rank)) + torch.arange(5.0)[:, None]
    out = gather_rows(local)
    assert out.shape == (5 * world, 3)
    for r in range(world):
        assert torch.equal(out[5 * r : 5 * r + 5], torch.full((5, 3), float(r)) + torch.arange(5.0)[:, None])
    # ragged shards, including an empty one; dst-only result
    n = [4, 0][rank] if world == 2 else rank + 1
    local = torch.arange(n * 2, dtype=torch.float64).reshape(n, 2) + 100 * rank
    out = gather_rows(local, dst=0)
    if rank == 0:
        want = torch.cat([torch.arange(k * 2, dtype=torch.float64).reshape(k, 2) + 100 * r
                          for r, k in enumerate(([4, 0] if world == 2 else range(1, world + 1)))])
        assert torch.equal(out, want), (out, want)
    else:
        assert out is None

    # the sharded driver with a stand-in computer (the real one needs a GPU): features are a
    # deterministic function of the signal, so order and row bookkeeping are checked exactly
    class FakeComputer:
        num_coeffs = 2
        def num_frames(self, n):
            return n // 10
        def compute_full_batch(self, sigs):
            return [np.stack([s[: (len(s) // 10) * 10].reshape(-1, 10).sum(1),
                              np.full(len(s) // 10, len(s), s.dtype)], 1) for s in sigs]
    rng = np.random.default_rng(0)
    signals = [rng.standard_normal(n).astype(np.float32) for n in (100, 0, 57, 230, 10, 999, 31)]
    feats = compute_full_sharded(FakeComputer(), signals)
    want = FakeComputer().compute_full_batch(signals)
    assert len(feats) == len(want)
    for a, b in zip(feats, want):
        assert a.shape == b.shape and np.array_equal(a, b)
    lo, own = compute_full_sharded(FakeComputer(), signals, gather=False)
    assert lo == shard_bounds(len(signals), world, rank)[0]
    # fewer signals than ranks, float64: the rank with the empty shard takes the dtype from the others
    one = [rng.standard_normal(120)]
    feats = compute_full_sharded(FakeComputer(), one)
    assert len(feats) == 1 and feats[0].dtype == np.float64
    assert np.array_equal(feats[0], FakeComputer().compute_full_batch(one)[0])
    assert compute_full_sharded(FakeComputer(), []) == []
    # only the rank's own block is read: everything else may be a placeholder
    lo, hi = shard_bounds(len(signals), world, rank)
    holes = [s if lo <= i < hi else None for i, s in enumerate(signals)]
    feats = compute_full_sharded(FakeComputer(), holes)
    for a, b in zip(feats, want):
        assert a.shape == b.shape and np.array_equal(a, b)
    # ragged gather with counts known in advance
    n = rank + 2
    out = gather_rows(torch.full((n, 3), float(rank)), counts=[r + 2 for r in range(world)])
    assert out.shape == (sum(r + 2 for r in range(world)), 3) and float(out[-1, 0]) == world - 1
    # global CMVN statistics: the sum over ranks of what each accumulated; a rank with an empty
    # shard (no statistics yet) contributes zeros
    from pydrobert_speech_amd.dist import all_reduce_stats
    from pydrobert_speech_amd.post import Standardize
    cmvn = Standardize()
    if rank == 0:
        cmvn._stats = np.array([[1.0, 2.0, 3.0, 10.0], [4.0, 5.0, 6.0, 0.0]])
    all_reduce_stats(cmvn)
    assert np.array_equal(cmvn._stats, [[1.0, 2.0, 3.0, 10.0], [4.0, 5.0, 6.0, 0.0]]), cmvn._stats
    cmvn._stats = cmvn._stats * (rank + 1)
    all_reduce_stats(cmvn)
    total = sum(r + 1 for r in range(world))
    assert np.array_equal(cmvn._stats, np.array([[1.0, 2.0, 3.0, 10.0], [4.0, 5.0, 6.0, 0.0]]) * total)
    if rank == 1:
        cmvn._stats = np.zeros((2, 7))
    try:
        all_reduce_stats(cmvn)
    except ValueError as err:
        assert "different widths" in str(err)
    else:
        raise AssertionError("width mismatch not reported")
    empty = Standardize()
    all_reduce_stats(empty)
    assert empty._stats is None
    dist.barrier()
    dist.destroy_process_group()
    open(os.path.join(os.environ["PDS_OUT"], f"rank{rank}.ok"), "w").write("ok")
    """
)


def test_gather_and_sharded_driver_world_size_2(tmp_path):
    import socket

    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    with socket.socket() as sock:  # a free rendezvous port
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, PDS_ROOT=ROOT, PDS_OUT=str(tmp_path), MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    res = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
         "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
        env=env, capture_output=True, text=True, timeout=300,
    )
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    assert (tmp_path / "rank0.ok").exists() and (tmp_path / "rank1.ok").exists()
