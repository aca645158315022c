"""Host feed: batches of HOST signals through an STFT computer's fused kernel and back

The reference's callers hold their audio on the host (``compute_full`` takes a numpy array, reference
compute.py:574; ``signals-to-torch-feat-dir`` reads files, command_line.py:337-607).  For them the rate of
the path is what it makes of PCIe: the kernel needs 0.27 ms for a batch the link needs ~12 ms to deliver.
:class:`HostFeed` binds the native staging ring of ``csrc/feed.hip`` (``pds_feed_*``): pinned host buffers
for samples and features, their device twins and one stream per slot, so that batch k + 1 uploads while
batch k computes and batch k - 1 downloads; samples travel as they are stored (int16 PCM stays int16 until
a frame is loaded: half the bytes of float32, a quarter of the reference drivers' float64).

    feed = HostFeed(computer, np.int16, slot_samples=1 << 26, slot_utts=1024)
    for feats_list in feed.run(batches):          # batches: iterable of lists of 1-D numpy arrays
        ...                                       # list of (T, C) float32 arrays, one per utterance
"""
import collections
import ctypes
from typing import Iterable, Iterator, List, Optional, Sequence

import numpy as np

from . import _native

__all__ = ["HostFeed"]

_FORMATS = {np.dtype(np.float32): 0, np.dtype(np.float64): 1, np.dtype(np.int16): 2}


class Ticket:
    """One submitted batch: which slot holds it and how many utterances it has"""

    def __init__(self, slot: int, n_utts: int):
        self.slot, self.n_utts = slot, n_utts


class HostFeed:
    """Ring of pinned staging slots in front of an :class:`STFTFrameComputer` (``pds_feed_*``)

    `dtype`: how the samples are stored on the host and travel to the device -- ``float32``, ``float64``
    (rounded to float32 as a frame is loaded, like the reference's drivers' arrays) or ``int16`` (converted as
    a frame is loaded).  `slot_samples` / `slot_utts`: capacity of one batch.  `slots`: depth of the ring
    (3 keeps the three engines busy).  `copy_threads`: host threads that copy a batch into its staging buffer.
    `feature_cols`: float32 columns per row the slots' feature buffers have room for (default ``num_coeffs``;
    more when a `post` callable widens the rows).  `direct`: ``True`` -- the kernel reads the samples from the
    pinned host buffer and writes the features to the pinned host buffer itself, so upload and download run
    concurrently (int16: 160 M frames/s host to host against 117 M staged); ``False`` -- DMA copies either side of
    the kernel; ``None`` (default) -- direct for float32 and int16 samples, staged for float64 (where it measured
    faster).  Features are float32.
    """

    def __init__(self, computer, dtype=np.float32, slot_samples: int = 1 << 26, slot_utts: int = 1024, slots: int = 3,
                 copy_threads: int = 8, device=None, feature_cols: int = 0, direct: Optional[bool] = None):
        torch = _native.require_device()
        self._lib = _native.lib()
        self.dtype = np.dtype(dtype)
        if self.dtype not in _FORMATS:
            raise TypeError("HostFeed: samples must be float32, float64 or int16")
        # (no reference to the computer itself: it keeps its feeds in a dict, and a cycle would leave their pinned
        # buffers to the cyclic collector)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self._plan = computer._native_plan(self.device)  # (kept alive: the feed uses its tables)
        self.slot_samples, self.slot_utts, self.slots = int(slot_samples), int(slot_utts), int(slots)
        self.copy_threads = int(copy_threads)
        self.num_coeffs = computer.num_coeffs
        handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _native.check(self._lib.pds_feed_create(self._plan.handle, _FORMATS[self.dtype], self.slot_samples,
                                                    self.slot_utts, self.slots, int(feature_cols), ctypes.byref(handle)),
                          "pds_feed_create")
        self.feature_cols = max(int(feature_cols), self.num_coeffs)
        self.direct = (self.dtype != np.float64) if direct is None else bool(direct)
        _native.check(self._lib.pds_feed_set_direct(handle, int(self.direct)), "pds_feed_set_direct")
        self._handle = handle
        self.slot_rows = int(self._lib.pds_feed_slot_rows(handle))
        self._torch = torch

    def close(self):
        handle, self._handle = getattr(self, "_handle", None), None
        if handle:
            self._lib.pds_feed_destroy(handle)

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- one batch ------------------------------------------------------------------------------------------

    def acquire(self):
        """The ring's next slot and a numpy view of its pinned sample buffer (blocks until the slot is free)"""
        slot, ptr = ctypes.c_int32(), ctypes.c_void_p()
        _native.check(self._lib.pds_feed_acquire(self._handle, ctypes.byref(slot), ctypes.byref(ptr)), "pds_feed_acquire")
        buf = (ctypes.c_char * (self.slot_samples * self.dtype.itemsize)).from_address(ptr.value)
        return slot.value, np.frombuffer(buf, dtype=self.dtype)

    def submit(self, signals: Sequence[np.ndarray], preemphasis: float = 0.0, post=None, nframes=None,
               pad_left: Optional[int] = None) -> Ticket:
        """Queue one batch of host signals (1-D arrays of the feed's dtype); returns its ticket at once

        `post`: optional ``callable(feats, row_offsets) -> tensor`` run on the slot's stream between the kernel
        and the download -- post-processors over the packed rows (``Deltas.apply_rows``, ...); its float32 result
        (at most ``slot_rows * feature_cols`` elements) is what comes back.  `nframes` / `pad_left`: frames to emit
        per signal and the left reflection, as ``STFTFrameComputer.launch`` takes them (default: ``compute_full``'s).
        """
        torch = self._torch
        n = len(signals)
        arrays = [np.ascontiguousarray(s, dtype=self.dtype).reshape(-1) for s in signals]
        lengths = np.asarray([a.shape[0] for a in arrays], dtype=np.int64)
        slot, _ = self.acquire()
        ptrs = (ctypes.c_void_p * max(n, 1))(*[a.ctypes.data for a in arrays])
        try:
            return self._submit(slot, arrays, ptrs, lengths, n, preemphasis, post, nframes, pad_left)
        except Exception:
            # the ring hands its slots out in order: pass the slot on as an empty batch instead of losing it
            with torch.cuda.device(self.device):
                if self._lib.pds_feed_submit(self._handle, slot, None, 0, 0.0, 1) == 0:
                    self.collect(Ticket(slot, 0))
            raise

    def _submit(self, slot, arrays, ptrs, lengths, n, preemphasis, post, nframes=None, pad_left=None) -> Ticket:
        torch = self._torch
        with torch.cuda.device(self.device):
            _native.check(self._lib.pds_feed_pack(self._handle, slot, ptrs, lengths.ctypes.data, n, self.copy_threads),
                          "pds_feed_pack")
            if nframes is None and pad_left is None:
                rc = self._lib.pds_feed_submit(self._handle, slot, lengths.ctypes.data, n, float(preemphasis),
                                               0 if post is not None else 1)
            else:
                counts = None if nframes is None else np.ascontiguousarray(nframes, dtype=np.int64).reshape(-1)
                if counts is not None and counts.shape[0] != n:
                    raise ValueError("HostFeed.submit: one frame count per signal")
                rc = self._lib.pds_feed_submit_frames(self._handle, slot, lengths.ctypes.data,
                                                      None if counts is None else counts.ctypes.data, n,
                                                      -1 if pad_left is None else int(pad_left), float(preemphasis),
                                                      0 if post is not None else 1)
            _native.check(rc, "pds_feed_submit")
            if post is not None:
                d_ptr, rows, offs, stream = ctypes.c_void_p(), ctypes.c_int64(), ctypes.c_void_p(), ctypes.c_void_p()
                _native.check(self._lib.pds_feed_device_view(self._handle, slot, ctypes.byref(d_ptr), ctypes.byref(rows),
                                                             ctypes.byref(offs), ctypes.byref(stream)), "pds_feed_device_view")
                row_offsets = np.ctypeslib.as_array(ctypes.cast(offs.value, ctypes.POINTER(ctypes.c_int64)), shape=(n + 1,)).copy()
                ext = torch.cuda.ExternalStream(stream.value, device=self.device)
                with torch.cuda.stream(ext):
                    feats = _device_view(torch, d_ptr.value, (rows.value, self.num_coeffs), self.device)
                    res = post(feats, row_offsets).to(torch.float32).contiguous()
                    if res.numel() > self.slot_rows * self.feature_cols:
                        raise ValueError("HostFeed: the post-processed rows do not fit the slot's host buffer")
                    _native.check(self._lib.pds_feed_download(self._handle, slot, res.data_ptr(), res.numel() * 4),
                                  "pds_feed_download")
                    res.record_stream(ext)
                ticket = Ticket(slot, n)
                ticket.post_shape = tuple(res.shape)
                return ticket
        return Ticket(slot, n)

    def collect(self, ticket: Ticket, copy: bool = True, release: bool = True):
        """Wait for a batch: ``(features, row_offsets)`` -- float32 ``(rows, num_coeffs)`` and ``n_utts + 1`` offsets

        ``copy=False``: the arrays are views of the slot's pinned buffers -- valid until :func:`release` (pass
        ``release=False`` and release the ticket yourself), after which the ring hands the slot out again.
        """
        h, offs, rows = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_int64()
        _native.check(self._lib.pds_feed_collect(self._handle, ticket.slot, ctypes.byref(h), ctypes.byref(offs),
                                                 ctypes.byref(rows)), "pds_feed_collect")
        shape = getattr(ticket, "post_shape", None) or (rows.value, self.num_coeffs)
        count = int(np.prod(shape))
        if count:
            feats = np.ctypeslib.as_array(ctypes.cast(h.value, ctypes.POINTER(ctypes.c_float)), shape=(count,)).reshape(shape)
        else:
            feats = np.zeros(shape, np.float32)
        row_offsets = np.ctypeslib.as_array(ctypes.cast(offs.value, ctypes.POINTER(ctypes.c_int64)), shape=(ticket.n_utts + 1,))
        if copy:
            feats, row_offsets = feats.copy(), row_offsets.copy()
        if release:
            self.release(ticket)
        return feats, row_offsets

    def collect_into(self, ticket: Ticket, out: np.ndarray) -> np.ndarray:
        """Wait for a batch and copy its features into the head of the C-contiguous float32 array `out` with the
        feed's copying threads (``pds_feed_unpack``); returns the batch's row offsets and releases the slot"""
        feats, row_offsets = self.collect(ticket, copy=False, release=False)
        if out.dtype != np.float32 or not out.flags.c_contiguous or out.size < feats.size:
            self.release(ticket)
            raise ValueError("HostFeed.collect_into: `out` must be C-contiguous float32 with room for the batch")
        try:
            if feats.size:
                _native.check(self._lib.pds_feed_unpack(self._handle, ticket.slot, out.ctypes.data, feats.size * 4,
                                                        self.copy_threads), "pds_feed_unpack")
            return row_offsets.copy()
        finally:
            self.release(ticket)

    def release(self, ticket: Ticket) -> None:
        _native.check(self._lib.pds_feed_release(self._handle, ticket.slot), "pds_feed_release")

    # -- a stream of batches --------------------------------------------------------------------------------

    def run(self, batches: Iterable[Sequence[np.ndarray]], preemphasis: float = 0.0, post=None,
            copy: bool = True) -> Iterator[List[np.ndarray]]:
        """Features of every batch, in order: lists of ``(T, C)`` float32 arrays, one per utterance

        Up to ``slots - 1`` batches are in flight behind the one being collected.  ``copy=False``: the arrays are
        views of the slot's pinned buffer, valid until the generator is advanced again (write them out, then ask
        for the next batch) -- no copy on the way out.
        """
        pending = collections.deque()
        held = None

        def take():
            nonlocal held
            ticket = pending.popleft()
            res = self._split(*self.collect(ticket, copy=copy, release=copy))
            held = None if copy else ticket
            return res

        def let_go():
            nonlocal held
            if held is not None:
                self.release(held)
                held = None

        try:
            for batch in batches:
                if len(pending) >= self.slots - 1 and pending:
                    yield take()
                    let_go()
                pending.append(self.submit(batch, preemphasis, post))
            while pending:
                yield take()
                let_go()
        finally:
            let_go()
            while pending:  # (the consumer stopped early: drain the ring)
                self.collect(pending.popleft(), copy=False)

    @staticmethod
    def _split(feats, rows):
        if feats.ndim == 2 and feats.shape[0] == rows[-1]:
            return [feats[rows[b] : rows[b + 1]] for b in range(len(rows) - 1)]
        return [feats]  # (a post-processor changed the row count: the caller knows how)


def _device_view(torch, ptr: int, shape, device):
    """float32 tensor over device memory the feed owns (no copy, no ownership)"""
    count = int(np.prod(shape))
    if count == 0:
        return torch.empty(shape, dtype=torch.float32, device=device)

    class _Holder:
        __cuda_array_interface__ = {"shape": tuple(int(v) for v in shape), "typestr": "<f4", "data": (int(ptr), False),
                                    "version": 2, "strides": None}

    return torch.as_tensor(_Holder(), device=device)
