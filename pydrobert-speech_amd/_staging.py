"""Pinned host staging for callers that hold their signals as numpy arrays.

``compute_full_batch`` of the short-integration computer (and any other caller without a native feed) spent
its time on the host: one ``np.concatenate`` into pageable memory, a pageable upload, a pageable download.
:class:`PinnedStaging` keeps two pinned buffers per thread of use -- samples up, features down -- packs the
utterances into the first with a few threads (numpy's copies release the GIL; one memcpy stream does not
keep up with the link) and copies asynchronously on the current stream.  The STFT computer has the native
ring of ``csrc/feed.hip`` for the same job; this is the small version for everything else.
"""
import threading
from concurrent.futures import ThreadPoolExecutor
from typing import List, Sequence, Tuple

import numpy as np

from . import _native

__all__ = ["PinnedStaging", "MAX_BYTES"]

MAX_BYTES = 1 << 30        # beyond this a batch takes the pageable path (pinning a GiB costs more than it saves)
_MIN_BYTES = 1 << 20        # below this the pageable path is as fast
_THREADS = 8
_POOL = None
_POOL_LOCK = threading.Lock()


def _pool() -> ThreadPoolExecutor:
    """One copy pool for the process (not one per computer)"""
    global _POOL
    with _POOL_LOCK:
        if _POOL is None:
            _POOL = ThreadPoolExecutor(max_workers=_THREADS, thread_name_prefix="pds-pack")
        return _POOL


class PinnedStaging:
    """Two pinned buffers (grown on demand, reused); one batch at a time"""

    def __init__(self):
        self._lock = threading.Lock()
        self._up = None
        self._down = None

    def _buffer(self, which: str, nbytes: int):
        torch = _native.require_device()
        buf = getattr(self, which)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(max(nbytes, 1 << 22), dtype=torch.uint8, pin_memory=True)
            setattr(self, which, buf)
        return buf

    def try_acquire(self, nbytes: int) -> bool:
        """True when this batch should go through the pinned buffers (and the caller now holds them)"""
        if nbytes < _MIN_BYTES or nbytes > MAX_BYTES:
            return False
        return self._lock.acquire(blocking=False)

    def release(self) -> None:
        self._lock.release()

    def upload(self, signals: Sequence[np.ndarray], offsets: np.ndarray, dtype: np.dtype, device):
        """The utterances back to back (converted to `dtype`) as a 1-D tensor on `device`"""
        torch = _native.require_device()
        dtype = np.dtype(dtype)
        total = int(offsets[-1])
        host = self._buffer("_up", total * dtype.itemsize)[: total * dtype.itemsize].numpy().view(dtype)
        jobs: List[Tuple[int, int]] = []
        # utterances dealt to the threads in contiguous runs of about equal bytes
        per = max(1, total // _THREADS)
        lo = 0
        for b in range(len(signals)):
            if int(offsets[b + 1]) - int(offsets[lo]) >= per or b == len(signals) - 1:
                jobs.append((lo, b + 1))
                lo = b + 1

        def copy(run):
            for b in range(run[0], run[1]):
                host[int(offsets[b]) : int(offsets[b + 1])] = np.asarray(signals[b]).reshape(-1)

        if len(jobs) > 1:
            list(_pool().map(copy, jobs))
        else:
            for run in jobs:
                copy(run)
        return torch.from_numpy(host).to(device, non_blocking=True)

    def download(self, feats) -> np.ndarray:
        """A fresh, caller-owned numpy array with the rows of the GPU tensor `feats`"""
        torch = _native.require_device()
        nbytes = feats.numel() * feats.element_size()
        pinned = self._buffer("_down", nbytes)[:nbytes].view(feats.dtype).view(feats.shape)
        pinned.copy_(feats, non_blocking=True)
        torch.cuda.current_stream(feats.device).synchronize()
        return pinned.numpy().copy()
