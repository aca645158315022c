"""Frame computers: the drop-in boundary of the STFT filter-bank hot path.

Host side of ``ShortTimeFourierTransformFrameComputer`` (reference compute.py:229-607).
The constructor derives the same quantities the reference's does (frame length/shift,
frame style, window, DFT size, per-filter truncated responses) on the host, condenses
the per-filter spectrum walk of ``_compute_frame`` (compute.py:416-460) into a sparse
bin-weight table, and hands both to a native *plan* (``include/pds_amd.h``).  All
per-frame arithmetic -- framing with symmetric reflection, windowing, the real DFT,
``|X|^2`` / ``|X|``, filter integration, log and energy -- runs in HIP kernels on an
MI355X; there is no CPU path in this module.
"""
import abc
import ctypes
import os
import threading
from typing import List, Mapping, Optional, Sequence, Tuple, Union

import numpy as np

from . import _native, config
from ._staging import MAX_BYTES as _STAGING_MAX_BYTES, PinnedStaging
from .alias import AliasedFactory, alias_factory_subclass_from_arg
from .filters import GammaWindow, HannWindow, LinearFilterBank, WindowFunction

__all__ = [
    "PackedLayout",
    "bin_weight_table",
    "fold_spectrum_index",
    "frame_by_frame_calculation",
    "FrameComputer",
    "LinearFilterBankFrameComputer",
    "ShortTimeFourierTransformFrameComputer",
    "STFTFrameComputer",
]

_MAX_UTTS_PER_CALL = 65535  # grid.y limit of the batch kernels
# compute_full_batch of host signals: staging slots of the pinned ring (feed.HostFeed) and the batch size from which
# it is used
_FEED_SLOT_SAMPLES = 1 << 24
_FEED_SLOT_UTTS = 4096
_FEED_MIN_SAMPLES = 1 << 22


class FrameComputer(AliasedFactory):
    """Turns a signal into a (num_frames, num_coeffs) feature matrix

    Interface of the reference's ``FrameComputer`` (compute.py:48-178): streaming with
    :func:`compute_chunk` / :func:`finalize`, or all at once with :func:`compute_full`.
    """

    @abc.abstractproperty
    def frame_style(self) -> str:
        """``'causal'`` or ``'centered'`` (compute.py:74-85)"""

    @abc.abstractproperty
    def sampling_rate(self) -> float:
        pass

    @abc.abstractproperty
    def frame_length(self) -> int:
        pass

    @property
    def frame_length_ms(self) -> float:
        return self.frame_length * 1000 / self.sampling_rate

    @abc.abstractproperty
    def frame_shift(self) -> int:
        pass

    @property
    def frame_shift_ms(self) -> float:
        return self.frame_shift * 1000 / self.sampling_rate

    @abc.abstractproperty
    def num_coeffs(self) -> int:
        pass

    @abc.abstractproperty
    def started(self) -> bool:
        """True between the first :func:`compute_chunk` and :func:`finalize`"""

    @abc.abstractmethod
    def compute_chunk(self, chunk: np.ndarray) -> np.ndarray:
        pass

    @abc.abstractmethod
    def finalize(self) -> np.ndarray:
        pass

    def compute_full(self, signal: np.ndarray) -> np.ndarray:
        return frame_by_frame_calculation(self, signal)


class LinearFilterBankFrameComputer(FrameComputer):
    """Computers with one coefficient per filter of a bank, optionally preceded by energy

    Reference: compute.py:181-218.
    """

    def __init__(self, bank: Union[LinearFilterBank, Mapping, str], include_energy: bool = False):
        self._bank = alias_factory_subclass_from_arg(LinearFilterBank, bank)
        self._include_energy = bool(include_energy)

    @property
    def bank(self) -> LinearFilterBank:
        return self._bank

    @property
    def includes_energy(self) -> bool:
        return self._include_energy

    @property
    def num_coeffs(self) -> int:
        return self._bank.num_filts + int(self._include_energy)


def fold_spectrum_index(k, dft_size: int):
    """Half-spectrum bin that full-spectrum index `k` lands on in the reference's walk

    ``_compute_frame`` (compute.py:423-455) consumes a filter's taps in alternating
    segments: ``half = len(rfft)`` bins walking up, then ``half - 2 + half % 2`` bins
    walking back down from index ``half - 2 + half % 2``.  For ``dft_size`` divisible by
    four this revisits the Nyquist bin (the reference tests ``half % 2`` where
    ``dft_size % 2`` was meant); parity with the reference requires reproducing it.
    """
    half = dft_size // 2 + 1
    back = half - 2 + half % 2
    r = np.asarray(k) % (half + back)
    return np.where(r < half, r, back - (r - half))


def bin_weight_table(
    starts: Sequence[int],
    truncated: Sequence[np.ndarray],
    dft_size: int,
    is_real: bool,
    use_power: bool,
) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """CSR table ``W`` with ``feature[f] = sum_b W[f, b] * P[b]`` (before the log)

    ``P`` is ``|X|^2`` (`use_power`) or ``|X|`` over the half spectrum.  Only moduli
    enter the reference's sums (compute.py:221-226, 434-451), so every tap contributes
    ``|H|^2`` or ``|H|`` to the bin it lands on; real banks are doubled
    (compute.py:456-457).

    Returns
    -------
    row_ptr : int32[F + 1]
    col : int32[nnz]
    val : float64[nnz]
    """
    half = dft_size // 2 + 1
    row_ptr = [0]
    cols: List[np.ndarray] = []
    vals: List[np.ndarray] = []
    for start, taps in zip(starts, truncated):
        gains = np.abs(np.asarray(taps)) ** (2 if use_power else 1)
        bins = fold_spectrum_index(int(start) + np.arange(len(gains)), dft_size)
        dense = np.bincount(bins, weights=gains, minlength=half).astype(np.float64)
        if is_real:
            dense *= 2
        nz = np.flatnonzero(dense)
        cols.append(nz.astype(np.int32))
        vals.append(dense[nz])
        row_ptr.append(row_ptr[-1] + len(nz))
    col = np.concatenate(cols) if cols else np.zeros(0, np.int32)
    val = np.concatenate(vals) if vals else np.zeros(0, np.float64)
    return np.asarray(row_ptr, dtype=np.int32), col.astype(np.int32), val.astype(np.float64)


class PackedLayout:
    """Geometry of a packed batch: where each utterance starts, how many frames it yields
    and where its rows go.  Built by :func:`STFTFrameComputer.prepare_layout`."""

    def __init__(self, B, extent, nframes, row_offsets, d_meta):
        self.B = B
        self.extent = extent  # samples of signal buffer the batch spans
        self.nframes = nframes  # host int64[B]
        self.row_offsets = row_offsets  # host int64[B + 1]
        self.d_meta = d_meta  # device int64[4, B]: offsets, lengths, nframes, row offsets
        # share of the (utterance, frame < longest) grid that exists: ragged batches (< 0.9) take the launch
        # that deals the existing chunks evenly (pds_stft_batch_ragged_f32)
        top = int(nframes.max()) if B else 0
        self.fill = float(nframes.sum()) / (top * B) if top else 1.0
        # chunk prefix sums of the stretch-scheduled launches (statics + deltas, fused CMVN sums), per fused geometry
        # of the plan: index data like d_meta, computed on the device once per layout and read-only after
        self._chunk_prefix = {}

    def chunk_prefix(self, plan, lib, torch, stream):
        """device int64[B + 1]: chunks of the plan's frames-per-wave in front of every utterance
        (``pds_stft_prepare_chunk_prefix``); cached -- a layout's index data never changes"""
        key = (int(plan.kernel_kind), self.d_meta.device)  # (the frames per wave follow from the fused geometry)
        hit = self._chunk_prefix.get(key)
        if hit is None:
            hit = torch.empty(self.B + 1, dtype=torch.int64, device=self.d_meta.device)
            _native.check(lib.pds_stft_prepare_chunk_prefix(plan.handle, self.d_meta[2].data_ptr(), self.B,
                                                            hit.data_ptr(), stream), "pds_stft_prepare_chunk_prefix")
            self._chunk_prefix[key] = hit
        return hit

    @property
    def total_rows(self) -> int:
        return int(self.row_offsets[-1])


class _NativePlan:
    """Owner of a ``pds_stft_plan``"""

    def __init__(self, desc: _native.StftDesc, window, row_ptr, col, val):
        _native.require_device()
        lib = _native.lib()
        self._keep = (
            np.ascontiguousarray(window, np.float64),
            np.ascontiguousarray(row_ptr, np.int32),
            np.ascontiguousarray(col, np.int32),
            np.ascontiguousarray(val, np.float64),
        )
        handle = ctypes.c_void_p()
        rc = lib.pds_stft_plan_create(
            ctypes.byref(desc), *(a.ctypes.data for a in self._keep), ctypes.byref(handle)
        )
        _native.check(rc, "pds_stft_plan_create")
        self.handle = handle
        self.kernel_kind = lib.pds_stft_plan_kernel_kind(handle)
        self.has_f64in = bool(lib.pds_stft_plan_has_f64in(handle))
        self.has_i16in = bool(lib.pds_stft_plan_has_i16in(handle))
        self.has_fused_deltas = bool(lib.pds_stft_plan_has_fused_deltas(handle))
        self.has_fused_cmvn = bool(lib.pds_stft_plan_has_fused_cmvn(handle))


    def __del__(self):
        handle, self.handle = getattr(self, "handle", None), None
        if handle:
            try:
                _native.lib().pds_stft_plan_destroy(handle)
            except Exception:  # interpreter shutdown
                pass


class ShortTimeFourierTransformFrameComputer(LinearFilterBankFrameComputer):
    """STFT filter-bank features computed by fused HIP kernels (alias ``stft``)

    Same constructor arguments, defaults and derived quantities as the reference class
    (compute.py:288-362).  Per frame: window, DFT, multiply by each filter's frequency
    response, sum ``|.|^2`` or ``|.|``, optional log; optional energy in column 0.

    Besides the reference's one-signal methods there is a batch interface,
    :func:`compute_full_batch` / :func:`compute_packed`, which is what the GPU is for.
    float32 signals take the fused FFT kernel; float64 signals are computed in float64
    (the reference's internal precision) by the generic kernels (an FFT in LDS for power-of-two
    transform sizes, a direct DFT otherwise).
    """

    aliases = {"stft"}
    #: :func:`launch` / :func:`compute_packed` take a ``preemphasis`` coefficient (batch driver)
    fuses_preemphasis = True

    def __init__(
        self,
        bank: Union[LinearFilterBank, Mapping, str],
        frame_length_ms: Optional[float] = None,
        frame_shift_ms: Optional[float] = 10,
        frame_style: Optional[str] = None,
        include_energy: bool = False,
        pad_to_nearest_power_of_two: bool = True,
        window_function: Optional[Union[WindowFunction, Mapping, str]] = None,
        use_log: bool = True,
        use_power: bool = False,
        kaldi_shift: bool = False,
    ):
        bank = alias_factory_subclass_from_arg(LinearFilterBank, bank)
        super().__init__(bank, include_energy=include_energy)
        self._rate = bank.sampling_rate
        self._frame_shift = int(0.001 * frame_shift_ms * self._rate)
        self._log = use_log
        self._power = use_power
        self._real = bank.is_real
        self._kaldi_shift = kaldi_shift
        if frame_style is None:
            frame_style = "centered" if bank.is_zero_phase else "causal"
        elif frame_style not in ("centered", "causal"):
            raise ValueError('Invalid frame style: "{}"'.format(frame_style))
        self._frame_style = frame_style
        if frame_length_ms is None:
            # longest impulse response, but at least one DFT bin inside the narrowest
            # filter (compute.py:319-330)
            longest = max(right - left for left, right in bank.supports)
            narrowest = min(right - left for left, right in bank.supports_hz)
            self._frame_length = max(longest, int(np.ceil(2 * self._rate / narrowest)))
        else:
            self._frame_length = int(0.001 * frame_length_ms * bank.sampling_rate)
        if window_function is None:
            window_function = GammaWindow() if frame_style == "causal" else HannWindow()
        else:
            window_function = alias_factory_subclass_from_arg(WindowFunction, window_function)
        self._window = window_function.get_impulse_response(self._frame_length)
        if pad_to_nearest_power_of_two:
            self._dft_size = int(2 ** np.ceil(np.log2(self._frame_length)))
        else:
            self._dft_size = self._frame_length
        # kept under the reference's private names: its torch mirror reads them
        # (torch.py:386-399)
        self._filt_start_idxs = []
        self._truncated_filts = []
        for filt_idx in range(bank.num_filts):
            start_idx, truncated = bank.get_truncated_response(filt_idx, self._dft_size)
            self._filt_start_idxs.append(start_idx)
            self._truncated_filts.append(truncated)
        if self._frame_style == "causal":
            self._pad_left = 0
        elif self._kaldi_shift:
            self._pad_left = self._frame_length // 2 - self._frame_shift // 2
        else:
            self._pad_left = (self._frame_length + 1) // 2 - 1
        self._log_floor = float(config.LOG_FLOOR_VALUE)  # snapshot, see config.py
        self._row_ptr, self._col, self._val = bin_weight_table(
            self._filt_start_idxs, self._truncated_filts, self._dft_size, self._real, self._power
        )
        self._plans = {}  # device index -> _NativePlan (tables live on one GPU)
        self._feeds = {}  # (device index, sample dtype) -> feed.HostFeed of compute_full_batch
        self._feed_lock = threading.Lock()  # a staging ring has one feeding thread: others take the plain path
        self._staging = PinnedStaging()  # pinned buffers for the batches the ring does not serve (float64 arithmetic)
        self._reset_stream()

    # ---- properties ---------------------------------------------------------------

    @property
    def frame_style(self) -> str:
        return self._frame_style

    @property
    def sampling_rate(self) -> float:
        return self._rate

    @property
    def frame_length(self) -> int:
        return self._frame_length

    @property
    def frame_shift(self) -> int:
        return self._frame_shift

    @property
    def started(self) -> bool:
        return self._started

    @property
    def kaldi_shift(self) -> bool:
        return self._kaldi_shift

    @property
    def dft_size(self) -> int:
        return self._dft_size

    @property
    def pad_left(self) -> int:
        """Samples of symmetric reflection before the first frame (compute.py:582-587)"""
        return self._pad_left

    @property
    def bin_weights(self) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """``(row_ptr, col, val)`` of the bin-weight table handed to the kernels"""
        return self._row_ptr, self._col, self._val

    @property
    def kernel_kind(self) -> int:
        """The DFT size if float32 input runs the fused FFT kernel, 0 for the generic kernels"""
        return self._native_plan().kernel_kind

    def num_frames(self, num_samples: int) -> int:
        """Frames :func:`compute_full` yields for a signal of that many samples"""
        if num_samples < self._frame_length // 2 + 1:
            return 0
        return max(0, (num_samples + self._frame_shift // 2) // self._frame_shift)

    # ---- native plumbing ----------------------------------------------------------

    def _native_plan(self, device=None) -> _NativePlan:
        """The plan for `device` (a torch device or index; default: the current device)"""
        torch = _native.require_device()
        index = torch.cuda.current_device() if device is None else torch.device(device).index
        if index is None:
            index = torch.cuda.current_device()
        if index not in self._plans:
            desc = _native.StftDesc(
                frame_length=self._frame_length,
                frame_shift=self._frame_shift,
                dft_size=self._dft_size,
                pad_left=self._pad_left,
                num_filts=self._bank.num_filts,
                nnz=len(self._col),
                use_power=int(bool(self._power)),
                use_log=int(bool(self._log)),
                include_energy=int(self._include_energy),
                reserved=0,
                log_floor=self._log_floor,
            )
            with torch.cuda.device(index):  # the plan's tables are allocated on the current device
                self._plans[index] = _NativePlan(desc, self._window, self._row_ptr, self._col, self._val)
        return self._plans[index]

    def prepare_layout(self, offsets, lengths, nframes=None, device="cuda") -> "PackedLayout":
        """Describe a packed batch once; reuse it for every launch with that geometry

        `offsets` / `lengths` (host sequences) locate each utterance in the packed signal
        buffer; `nframes` defaults to :func:`num_frames` of each length.  The returned
        object owns the small device-side index arrays the kernel reads.
        """
        torch = _native.require_device()
        lengths = np.asarray(lengths, dtype=np.int64).reshape(-1)
        offsets = np.asarray(offsets, dtype=np.int64).reshape(-1)
        B = len(lengths)
        if nframes is None:
            nframes = np.array([self.num_frames(int(n)) for n in lengths], dtype=np.int64)
        else:
            nframes = np.asarray(nframes, dtype=np.int64).reshape(-1)
        if len(offsets) != B or len(nframes) != B:
            raise ValueError("offsets, lengths and nframes must have the same length")
        if B and (offsets.min() < 0 or lengths.min() < 0 or nframes.min() < 0):
            raise ValueError("negative offset, length or frame count")
        if B and ((nframes > 0) & (lengths <= 0)).any():
            raise ValueError("cannot emit frames for an empty utterance")
        row_offsets = np.zeros(B + 1, dtype=np.int64)
        np.cumsum(nframes, out=row_offsets[1:])
        meta = np.stack([offsets, lengths, nframes, row_offsets[:-1]]) if B else np.zeros((4, 0), np.int64)
        return PackedLayout(
            B=B,
            extent=int((offsets + lengths).max()) if B else 0,
            nframes=nframes,
            row_offsets=row_offsets,
            d_meta=torch.from_numpy(np.ascontiguousarray(meta)).to(device),
        )

    def launch(self, signal, layout: "PackedLayout", out=None, pad_left=None, generic=False,
               preemphasis: float = 0.0):
        """Queue the batch kernel for `layout` on the current stream; returns the output

        `preemphasis`: coefficient of :class:`pydrobert_speech_amd.pre.Preemphasize` to apply
        to every utterance while its frames are loaded (0 = none) -- equivalent to, and one
        pass over the signal cheaper than, pre-emphasising first.

        `signal`: contiguous 1-D float32/float64 GPU tensor; `out`: optional
        ``(>= total_rows, >= num_coeffs)`` tensor of the same dtype with unit column
        stride (its row stride may be larger, e.g. to leave room for deltas).  Nothing
        here synchronises or allocates besides `out` when it is not given.

        An int16 `signal` (PCM as a WAV file holds it) gives float32 features equal to those of
        ``signal.to(torch.float32)``: the samples are converted as their frames are loaded
        (``pds_stft_batch_i16in``), half the bytes of float32 samples over PCIe and out of HBM.
        """
        torch = _native.require_device()
        lib = _native.lib()
        if not signal.is_cuda or signal.dim() != 1 or not signal.is_contiguous():
            raise ValueError("signal must be a contiguous 1-D tensor on the GPU")
        plan = self._native_plan(signal.device)
        if layout.extent > signal.numel():
            raise ValueError("an utterance lies outside the signal buffer")
        f64in = signal.dtype == torch.float64 and config.FLOAT64_ARITHMETIC == "float32" and not generic
        if signal.dtype == torch.int16 and (generic or not plan.has_i16in):
            # (no fused int16-input kernel for this transform size: convert first)
            return self.launch(signal.to(torch.float32), layout, out=out, pad_left=pad_left, generic=generic,
                               preemphasis=preemphasis)

        def rounded_first():
            # opt-in float32 arithmetic for float64 data where no fused float64-input kernel serves the
            # call (transform size without one, or a filter table too large for LDS): round the samples first
            feats = self.launch(signal.to(torch.float32), layout, pad_left=pad_left, preemphasis=preemphasis)
            if out is None:
                return feats.to(torch.float64)
            out[: feats.shape[0], : feats.shape[1]] = feats
            return out

        if f64in and not plan.has_f64in:
            return rounded_first()
        if f64in and preemphasis and (out is None or out.dtype == torch.float64):
            # (float64 features with fused pre-emphasis: float32 features from the same kernel, widened)
            feats = self.launch(signal, layout, pad_left=pad_left, preemphasis=preemphasis,
                                out=torch.empty((layout.total_rows, self.num_coeffs), dtype=torch.float32,
                                                device=signal.device))
            if out is None:
                return feats.to(torch.float64)
            out[: feats.shape[0], : feats.shape[1]] = feats
            return out
        out_dtype = signal.dtype
        if f64in:
            # the fused kernel rounds the samples as it loads them and stores float32 or float64 features
            out_dtype = torch.float64 if out is None else out.dtype
            if out_dtype not in (torch.float32, torch.float64):
                raise ValueError("out has the wrong dtype, shape or strides")
            is64 = int(out_dtype == torch.float64)

            def fn(*args):
                return lib.pds_stft_batch_f64in(*args[:11], is64, *args[11:])
        elif signal.dtype == torch.float32:
            fn = lib.pds_stft_batch_f32_generic if generic else lib.pds_stft_batch_f32
            # (with a fused pre-emphasis the stretch schedule exists for the row-segment kernels of N = 512 / 1024: the
            # launch takes it where it can and the round-robin order otherwise, same values)
            if not generic and plan.kernel_kind and layout.fill < 0.9 and config.RAGGED_SCHEDULING:
                # (per launch, from the caching allocator: a workspace kept on the layout was shared by launches on
                # different streams, and chunk_prefix_kernel rewrites it every time)
                d_work = torch.empty(min(layout.B, _MAX_UTTS_PER_CALL) + 1, dtype=torch.int64, device=signal.device)
                work = d_work.data_ptr()

                def fn(*args):
                    return lib.pds_stft_batch_ragged_f32(*args[:10], work, *args[10:])
        elif signal.dtype == torch.float64:
            fn = lib.pds_stft_batch_f64
        elif signal.dtype == torch.int16:
            fn = lib.pds_stft_batch_i16in
            out_dtype = torch.float32
            if layout.fill < 0.9 and config.RAGGED_SCHEDULING:
                d_work = torch.empty(min(layout.B, _MAX_UTTS_PER_CALL) + 1, dtype=torch.int64, device=signal.device)
                work = d_work.data_ptr()

                def fn(*args):
                    return lib.pds_stft_batch_ragged_i16in(*args[:10], work, *args[10:])
        else:
            raise TypeError("signal must be float32, float64 or int16")
        total = layout.total_rows
        if out is None:
            out = torch.empty((total, self.num_coeffs), dtype=out_dtype, device=signal.device)
        elif (
            out.dtype != out_dtype
            or out.dim() != 2
            or out.shape[0] < total
            or out.shape[1] < self.num_coeffs
            or out.stride(1) != 1
        ):
            raise ValueError("out has the wrong dtype, shape or strides")
        if total == 0:
            return out
        stream = torch.cuda.current_stream(signal.device).cuda_stream
        pad = -1 if pad_left is None else int(pad_left)
        meta = layout.d_meta
        with torch.cuda.device(signal.device):
            for lo in range(0, layout.B, _MAX_UTTS_PER_CALL):
                hi = min(layout.B, lo + _MAX_UTTS_PER_CALL)
                rc = fn(
                    plan.handle,
                    signal.data_ptr(),
                    meta[0, lo:].data_ptr(),
                    meta[1, lo:].data_ptr(),
                    meta[2, lo:].data_ptr(),
                    meta[3, lo:].data_ptr(),
                    hi - lo,
                    int(layout.nframes[lo:hi].max()),
                    pad,
                    float(preemphasis),
                    out.data_ptr(),
                    out.stride(0),
                    stream,
                )
                if rc != 0 and f64in and lo == 0:
                    return rounded_first()  # (e.g. the bank's table does not fit in LDS beside the waves' areas)
                if rc != 0 and signal.dtype == torch.int16 and lo == 0:
                    return self.launch(signal.to(torch.float32), layout, out=out, pad_left=pad_left,
                                       preemphasis=preemphasis)
                _native.check(rc, "pds_stft_batch")
        return out

    def launch_with_deltas(self, signal, layout: "PackedLayout", deltas, out=None, pad_left=None,
                           fused: Optional[bool] = None, preemphasis: float = 0.0):
        """Statics and ``deltas`` (a :class:`pydrobert_speech_amd.post.Deltas`) of a packed batch

        Returns the ``(total_rows, (K + 1) num_coeffs)`` float32 tensor: row r holds the features of
        frame r followed by its order-1 .. order-K deltas: :func:`launch` into rows of that stride
        followed by ``deltas.apply_rows``.  ``fused=True`` asks for the one-launch form
        (``pds_stft_deltas_batch_f32``: transform sizes 512 and 1024, mel-like banks, ``Deltas(1 or 2)``
        with the default context window and padding): the statics are the same bit for bit, the deltas
        are formed in float32 from coefficients held in registers (within a few float32 ulps of the
        statics of the float64-accumulated ones) and nothing is read back from memory.  It is what runs
        when ``fused`` is left at ``None`` and the plan has it (1.34 x the rate of the two launches on
        BASELINE.json configs[2]); ``fused=False`` keeps the two launches, whose deltas are numpy's bit
        for bit.

        `preemphasis` as in :func:`launch`; float64 signals take the one launch too when
        ``config.FLOAT64_ARITHMETIC == "float32"`` (samples rounded as the frame is loaded, after the
        pre-emphasis: ``pds_stft_deltas_batch``) -- the reference drivers' chain float64 audio ->
        Preemphasize -> compute_full -> Deltas (command_line.py:345-350) as one kernel.  Otherwise their
        statics are computed in float64 (:func:`launch`) and rounded into the float32 result.  int16 signals
        (16-bit PCM) take the one launch too.
        """
        if fused is None:
            fused = True
        torch = _native.require_device()
        lib = _native.lib()
        C, K = self.num_coeffs, deltas.num_deltas
        total = layout.total_rows
        if out is None:
            out = torch.empty((total, (K + 1) * C), dtype=torch.float32, device=signal.device)
        elif (out.dtype != torch.float32 or out.dim() != 2 or out.shape[0] < total or out.shape[1] < (K + 1) * C
              or out.stride(1) != 1):
            raise ValueError("out has the wrong dtype, shape or strides")
        plan = self._native_plan(signal.device)
        filts = deltas._filts[1:]
        f64in = signal.dtype == torch.float64 and config.FLOAT64_ARITHMETIC == "float32" and plan.has_f64in
        i16in = signal.dtype == torch.int16 and plan.has_i16in
        if signal.dtype not in (torch.float32, torch.float64, torch.int16):
            raise TypeError("signal must be float32, float64 or int16")
        fused = (
            fused and plan.has_fused_deltas and (signal.dtype == torch.float32 or f64in or i16in) and K in (1, 2) and deltas.concatenate
            and deltas._pad_mode == "edge" and not deltas._pad_kwargs
            and [len(f) for f in filts] == [5, 9][:K] and total > 0
        )
        if fused:
            if not signal.is_cuda or signal.dim() != 1 or not signal.is_contiguous():
                raise ValueError("signal must be a contiguous 1-D tensor on the GPU")
            if layout.extent > signal.numel():
                raise ValueError("an utterance lies outside the signal buffer")
            taps = np.ascontiguousarray(np.concatenate(filts), dtype=np.float64)
            stream = torch.cuda.current_stream(signal.device).cuda_stream
            pad = -1 if pad_left is None else int(pad_left)
            meta = layout.d_meta
            with torch.cuda.device(signal.device):
                # (one call: the layout's prepared chunk prefix sums; more: a workspace the kernel in front fills)
                single = layout.B <= _MAX_UTTS_PER_CALL
                work = (layout.chunk_prefix(plan, lib, torch, stream) if single else
                        torch.empty(min(layout.B, _MAX_UTTS_PER_CALL) + 1, dtype=torch.int64, device=signal.device))
                for lo in range(0, layout.B, _MAX_UTTS_PER_CALL):
                    hi = min(layout.B, lo + _MAX_UTTS_PER_CALL)
                    rc = lib.pds_stft_deltas_batch(
                        plan.handle, signal.data_ptr(), {torch.float64: 1, torch.int16: 2}.get(signal.dtype, 0), meta[0, lo:].data_ptr(),
                        meta[1, lo:].data_ptr(), meta[2, lo:].data_ptr(), meta[3, lo:].data_ptr(), hi - lo,
                        int(layout.nframes[lo:hi].max()), pad, float(preemphasis), K, 2, taps.ctypes.data,
                        work.data_ptr(), int(single), out.data_ptr(), out.stride(0), stream,
                    )
                    if rc == -1 and lo == 0:
                        fused = False  # (not served, e.g. a filter table that does not fit in LDS): the two launches below
                        break
                    _native.check(rc, "pds_stft_deltas_batch")
        if not fused:
            if signal.dtype in (torch.float32, torch.int16) or f64in:
                self.launch(signal, layout, out=out, pad_left=pad_left, preemphasis=preemphasis)
            elif total:
                # float64 arithmetic (config.FLOAT64_ARITHMETIC == "float64"): float64 statics, rounded into the rows
                out[:total, :C] = self.launch(signal, layout, pad_left=pad_left, preemphasis=preemphasis)
            if total:
                deltas.apply_rows(out[:, :C], layout.row_offsets, out=out)
        return out[:, : (K + 1) * C] if out.shape[1] != (K + 1) * C else out

    def launch_with_cmvn(self, signal, layout: "PackedLayout", cmvn, out=None, pad_left=None,
                         fused: Optional[bool] = None, out_dtype=None, feats_out=None):
        """Features of a packed batch standardised per utterance (``cmvn``: a :class:`pydrobert_speech_amd.post.CMVN`
        without global statistics): :func:`launch` followed by ``cmvn.apply_rows``, float64 result like the
        reference's (``post.py:250-295``), float32 with ``out_dtype=torch.float32``.

        ``fused=True`` asks for the form in which the STFT launch takes the sums itself where the plan has it
        (``pds_stft_cmvn_batch_f32``: every wave adds what it stores to float64 sums of its own, the normalising
        kernel adds an utterance's pieces in a fixed order and reads the features once instead of twice: 770 MB of
        HBM traffic instead of 835 on BASELINE.json configs[4]).  It is NOT the default: measured on that workload
        the sums cost the (vector-bound) STFT kernel 2.9 us and save the (write-bound) normalising kernel 3.1 us, and
        the step comes out 1 % slower than the two calls (0.289 against 0.285 ms).  ``fused=None`` / ``False``,
        float64 signals, or a plan / call the kernel does not serve: the two calls.
        `feats_out`: optional ``(>= total_rows, >= num_coeffs)`` float32 tensor that receives the features
        themselves (else they live in a temporary).
        """
        torch = _native.require_device()
        lib = _native.lib()
        if cmvn.have_stats:
            raise ValueError("launch_with_cmvn standardises locally; global statistics are set")
        C, total = self.num_coeffs, layout.total_rows
        out_dtype = torch.float64 if out_dtype is None else out_dtype
        if out_dtype not in (torch.float32, torch.float64):
            raise ValueError("out_dtype must be float32 or float64")
        if out is None:
            out = torch.empty((total, C), dtype=out_dtype, device=signal.device)
        elif out.dtype != out_dtype or out.dim() != 2 or out.shape[0] < total or out.shape[1] < C or out.stride(1) != 1:
            raise ValueError("out has the wrong dtype, shape or strides")
        plan = self._native_plan(signal.device)
        use = bool(fused) and plan.has_fused_cmvn and signal.dtype == torch.float32 and total > 0 \
            and layout.B <= _MAX_UTTS_PER_CALL
        if use:
            if not signal.is_cuda or signal.dim() != 1 or not signal.is_contiguous():
                raise ValueError("signal must be a contiguous 1-D tensor on the GPU")
            if layout.extent > signal.numel():
                raise ValueError("an utterance lies outside the signal buffer")
            stream = torch.cuda.current_stream(signal.device).cuda_stream
            pad = -1 if pad_left is None else int(pad_left)
            meta = layout.d_meta
            with torch.cuda.device(signal.device):
                prefix = layout.chunk_prefix(plan, lib, torch, stream)
                plen = int(lib.pds_stft_cmvn_partials_len(plan.handle, layout.B))
                partials = torch.empty(plen, dtype=torch.float64, device=signal.device)  # (scratch of this launch)
                if feats_out is None:
                    feats = torch.empty((total, C), dtype=torch.float32, device=signal.device)
                else:
                    feats = feats_out
                    if (feats.dtype != torch.float32 or feats.dim() != 2 or feats.shape[0] < total or feats.shape[1] < C
                            or feats.stride(1) != 1 or not feats.is_cuda):
                        raise ValueError("feats_out has the wrong dtype, shape or strides")
                stats = torch.empty((layout.B, 2, C), dtype=torch.float64, device=signal.device)
                zero_var = torch.zeros(1, dtype=torch.int32, device=signal.device)
                rc = lib.pds_stft_cmvn_batch_f32(
                    plan.handle, signal.data_ptr(), meta[0].data_ptr(), meta[1].data_ptr(), meta[2].data_ptr(),
                    meta[3].data_ptr(), layout.B, int(layout.nframes.max()), pad, int(cmvn._norm_var), prefix.data_ptr(),
                    1, partials.data_ptr(), plen, feats.data_ptr(), feats.stride(0), stats.data_ptr(), out.data_ptr(),
                    int(out_dtype == torch.float64), out.stride(0), zero_var.data_ptr(), stream,
                )
            if rc == 0:
                cmvn._last_zero_var = zero_var
                return out[:total, :C] if out.shape != (total, C) else out
            if rc != -1:  # (PDS_ERR_INVALID = not served, e.g. no room in LDS for the waves' sums: the two calls below)
                _native.check(rc, "pds_stft_cmvn_batch")
        if feats_out is not None and signal.dtype == torch.float32:
            feats = self.launch(signal, layout, pad_left=pad_left, out=feats_out)[:total, :C]
        else:
            feats = self.launch(signal, layout, pad_left=pad_left)
        if feats.dtype != torch.float32:
            feats = feats.to(torch.float32)
        res = cmvn.apply_rows(feats, layout.row_offsets, out_dtype=out_dtype)
        if total:
            out[:total, :C] = res
        return out[:total, :C] if out.shape != (total, C) else out

    def compute_packed(self, signal, offsets, lengths, nframes=None, pad_left=None, out=None,
                       generic=False, preemphasis: float = 0.0):
        """:func:`prepare_layout` + :func:`launch` in one call

        Returns ``(feats, row_offsets)``: the ``(total_rows, num_coeffs)`` GPU tensor and
        the ``B + 1`` host row offsets of the utterances inside it.
        """
        layout = self.prepare_layout(offsets, lengths, nframes, device=signal.device)
        feats = self.launch(signal, layout, out=out, pad_left=pad_left, generic=generic,
                            preemphasis=preemphasis)
        return feats, layout.row_offsets

    @staticmethod
    def _compute_dtype(dtype) -> np.dtype:
        dtype = np.dtype(dtype)
        return dtype if dtype in (np.float32, np.float64) else np.dtype(np.float64)

    def _run_host_signal(self, signal: np.ndarray, nframes: int, pad_left: Optional[int]):
        # one host signal -> one host feature matrix (dtype of `signal`)
        torch = _native.require_device()
        in_dtype = signal.dtype
        if nframes <= 0:
            return np.empty((0, self.num_coeffs), dtype=in_dtype)
        # one host signal (compute_full, and every chunk of the streaming interface): through a small staging ring in
        # direct mode -- the kernel reads the pinned copy of the signal and writes the pinned features itself, no upload /
        # download calls: 145 -> 80 us per call for 10 s of audio (tools/latency.py)
        fed = self._one_signal_through_feed(signal, in_dtype, nframes, pad_left)
        if fed is not None:
            return fed
        work = np.array(signal, dtype=self._compute_dtype(in_dtype), copy=True, order="C")
        d_sig = torch.from_numpy(work).to("cuda")
        feats, _ = self.compute_packed(d_sig, [0], [len(work)], [nframes], pad_left)
        res = feats.cpu().numpy()
        return res if res.dtype == in_dtype else res.astype(in_dtype)

    # ---- the reference's one-signal interface ---------------------------------------

    def compute_full(self, signal):
        """Features of a whole signal (reference compute.py:574-607)

        `signal` is a 1-D float array; the result is a new ``(num_frames, num_coeffs)``
        array of the same dtype.  A 1-D ``torch`` tensor on the GPU is accepted too and
        then the result stays on the GPU.
        """
        if self.started:
            raise ValueError("Already started computing frames")
        if not isinstance(signal, np.ndarray) and getattr(signal, "is_cuda", False):
            return self.compute_full_batch([signal])[0]
        signal = np.asarray(signal)
        if signal.ndim != 1:
            raise ValueError("signal must be 1-dimensional")
        return self._run_host_signal(signal, self.num_frames(len(signal)), None)

    def _one_signal_through_feed(self, signal, in_dtype, nframes, pad_left):
        from .feed import HostFeed  # (imports this module)

        torch = _native.require_device()
        f64 = in_dtype == np.float64 and config.FLOAT64_ARITHMETIC == "float32"
        n = len(signal)
        if not config.HOST_FEED or not (in_dtype == np.float32 or f64) or n > _FEED_SLOT_SAMPLES:
            return None
        plan = self._native_plan()
        if not plan.kernel_kind or (f64 and not plan.has_f64in):
            return None
        if not self._feed_lock.acquire(blocking=False):
            return None  # (another thread is in a staging ring of this computer: the plain path has no shared state)
        try:
            key = (torch.cuda.current_device(), np.dtype(in_dtype), "one")
            feed = self._feeds.get(key)
            if feed is None or feed.slot_samples < n:
                if feed is not None:
                    feed.close()
                size = 1 << 20
                while size < n:
                    size <<= 1
                feed = self._feeds[key] = HostFeed(self, in_dtype, slot_samples=size, slot_utts=1, slots=1, copy_threads=2)
            if nframes > feed.slot_rows:
                return None
            feats, _ = feed.collect(feed.submit([signal], nframes=[nframes], pad_left=pad_left))
        finally:
            self._feed_lock.release()
        return feats if feats.dtype == in_dtype else feats.astype(in_dtype)

    def _full_batch_through_feed(self, signals, lengths, in_dtype, preemphasis):
        """Host signals through the pinned staging ring (``feed.HostFeed``): slot-sized pieces of the batch upload,
        compute and download concurrently instead of one concatenate + pageable copy each way (measured 3 x the
        rate on 1024 x 10 s).  Float32 signals (float64 ones when the configuration asks for float32 arithmetic)
        of a plan with a fused kernel, batches of at least a few seconds of audio; else ``None``: the plain path."""
        from .feed import HostFeed  # (imports this module)

        torch = _native.require_device()
        total = int(sum(lengths))
        f64 = in_dtype == np.float64 and config.FLOAT64_ARITHMETIC == "float32"
        if not config.HOST_FEED or total < _FEED_MIN_SAMPLES or not (in_dtype == np.float32 or f64):
            return None
        if max(lengths) > _FEED_SLOT_SAMPLES:
            return None
        plan = self._native_plan()
        if not plan.kernel_kind or (f64 and not plan.has_f64in):
            return None
        if not self._feed_lock.acquire(blocking=False):
            return None
        try:
            key = (torch.cuda.current_device(), np.dtype(in_dtype))
            feed = self._feeds.get(key)
            if feed is None:
                feed = self._feeds[key] = HostFeed(self, in_dtype, slot_samples=_FEED_SLOT_SAMPLES, slot_utts=_FEED_SLOT_UTTS,
                                                   slots=3, copy_threads=min(16, os.cpu_count() or 1))
            # slot-sized runs of consecutive utterances
            pieces, lo, acc = [], 0, 0
            for b, n in enumerate(lengths):
                if b > lo and (acc + n > _FEED_SLOT_SAMPLES or b - lo >= _FEED_SLOT_UTTS):
                    pieces.append((lo, b))
                    lo, acc = b, 0
                acc += n
            pieces.append((lo, len(lengths)))
            rows = np.zeros(len(lengths) + 1, dtype=np.int64)
            np.cumsum([self.num_frames(n) for n in lengths], out=rows[1:])
            C = self.num_coeffs
            out = np.empty((int(rows[-1]), C), dtype=np.float32)
            pending = []

            def drain():
                ticket, (a, _) = pending.pop(0)
                feed.collect_into(ticket, out[rows[a]:])

            try:
                for piece in pieces:
                    if len(pending) >= feed.slots - 1:
                        drain()
                    pending.append((feed.submit(signals[piece[0] : piece[1]], preemphasis), piece))
                while pending:
                    drain()
            except BaseException:
                for ticket, _ in pending:  # (leave the ring free for the next call)
                    try:
                        feed.collect(ticket, copy=False)
                    except Exception:
                        pass
                raise
        finally:
            self._feed_lock.release()
        if out.dtype != in_dtype:
            out = out.astype(in_dtype)
        return [out[rows[b] : rows[b + 1]] for b in range(len(lengths))]

    def compute_full_batch(self, signals: Sequence, preemphasis: float = 0.0) -> list:
        """:func:`compute_full` of many signals in one launch (optionally pre-emphasised)

        `signals` is a sequence of 1-D arrays (numpy, or torch tensors already on the
        GPU) of one dtype.  Returns the list of feature matrices, in order; numpy in,
        numpy out; GPU tensors in, GPU tensors (views of one buffer) out.
        """
        if self.started:
            raise ValueError("Already started computing frames")
        torch = _native.require_device()
        if len(signals) == 0:
            return []
        on_gpu = bool(getattr(signals[0], "is_cuda", False))
        lengths = [int(s.shape[0]) for s in signals]
        offsets = np.zeros(len(signals) + 1, dtype=np.int64)
        np.cumsum(lengths, out=offsets[1:])
        if on_gpu:
            in_dtype = None
            packed = torch.cat([s.reshape(-1) for s in signals]) if len(signals) > 1 else signals[0].contiguous()
            if packed.dtype not in (torch.float32, torch.float64):
                raise TypeError("GPU signals must be float32 or float64")
        else:
            in_dtype = np.asarray(signals[0]).dtype
            through_feed = self._full_batch_through_feed(signals, lengths, in_dtype, preemphasis)
            if through_feed is not None:
                return through_feed
            work = np.dtype(self._compute_dtype(in_dtype))
            nbytes = int(offsets[-1]) * work.itemsize
            if nbytes > _STAGING_MAX_BYTES and len(signals) > 1:  # (pieces the pinned buffers hold)
                signals, half = list(signals), len(signals) // 2
                return (self.compute_full_batch(signals[:half], preemphasis)
                        + self.compute_full_batch(signals[half:], preemphasis))
            if self._staging.try_acquire(nbytes):
                # pinned buffers both ways, the utterances packed by a few threads (_staging.py)
                try:
                    packed = self._staging.upload(signals, offsets, work, torch.device("cuda", torch.cuda.current_device()))
                    feats, rows = self.compute_packed(packed, offsets[:-1], lengths, preemphasis=preemphasis)
                    feats = self._staging.download(feats)
                finally:
                    self._staging.release()
                if feats.dtype != in_dtype:
                    feats = feats.astype(in_dtype)
                return [feats[rows[i] : rows[i + 1]] for i in range(len(signals))]
            host = np.concatenate(
                [np.asarray(s, dtype=work).reshape(-1) for s in signals]
            ) if offsets[-1] else np.zeros(0, work)
            packed = torch.from_numpy(host).to("cuda")
        feats, rows = self.compute_packed(packed, offsets[:-1], lengths, preemphasis=preemphasis)
        if on_gpu:
            return [feats[rows[i] : rows[i + 1]] for i in range(len(signals))]
        feats = feats.cpu().numpy()
        if feats.dtype != in_dtype:
            feats = feats.astype(in_dtype)
        return [feats[rows[i] : rows[i + 1]] for i in range(len(signals))]

    # ---- streaming (reference compute.py:462-572) -----------------------------------
    #
    # The reference keeps a ring buffer of one frame.  Here the not-yet-consumed tail of
    # the stream is kept as an array (`_carry`) together with the amount of left
    # reflection its first frame still needs (`_carry_pad`); completed frames are sent
    # to the batch kernel.  Emission rules are the reference's: a frame is emitted once
    # its right edge is available (compute.py:471-480), and `finalize` emits
    # ``(buffered + S // 2 - consumed_pad) // S`` more with symmetric right padding
    # (compute.py:546-561).

    def _reset_stream(self):
        self._started = False
        self._first_frame = True
        self._carry = np.zeros(0, dtype=np.float64)
        self._carry_pad = self._pad_left
        self._skip = 0
        self._chunk_dtype = np.dtype(np.float64)

    def compute_chunk(self, chunk: np.ndarray) -> np.ndarray:
        chunk = np.asarray(chunk)
        self._chunk_dtype = chunk.dtype
        self._started = True
        if self._skip:
            # frame_shift > frame_length: samples between frames are dropped
            drop = min(self._skip, len(chunk))
            chunk = chunk[drop:]
            self._skip -= drop
        L, S = self._frame_length, self._frame_shift
        work = np.concatenate([self._carry.astype(chunk.dtype, copy=False), chunk])
        avail, cp = len(work), self._carry_pad
        k = max(0, (avail + cp - L) // S + 1)
        feats = self._run_host_signal(work, k, cp) if k else np.empty(
            (0, self.num_coeffs), dtype=chunk.dtype
        )
        if k:
            self._first_frame = False
            nxt = k * S - cp  # start of the next frame inside `work`
            if nxt >= 0:
                self._skip = max(0, nxt - avail)
                work, cp = work[min(nxt, avail) :], 0
            else:
                cp = -nxt
        self._carry, self._carry_pad = work, cp
        return feats

    def finalize(self) -> np.ndarray:
        S = self._frame_shift
        buffered = len(self._carry)
        if self._first_frame:
            cp = self._pad_left
            num_frames = (buffered + S // 2) // S
        else:
            cp = self._carry_pad
            num_frames = (buffered + cp + S // 2 - self._pad_left) // S
        dtype = self._chunk_dtype
        if num_frames >= 1 and buffered:
            feats = self._run_host_signal(self._carry.astype(dtype, copy=False), num_frames, cp)
        else:
            feats = np.empty((0, self.num_coeffs), dtype=dtype)
        self._reset_stream()
        return feats


STFTFrameComputer = ShortTimeFourierTransformFrameComputer


def frame_by_frame_calculation(computer: FrameComputer, signal: np.ndarray, chunk_size: int = 2 ** 10):
    """Feed `signal` through ``compute_chunk`` in pieces of `chunk_size`, then ``finalize``

    Reference: compute.py:1002-1039.
    """
    if computer.started:
        raise ValueError("Already started computing frames")
    pieces = []
    while len(signal):
        pieces.append(computer.compute_chunk(signal[:chunk_size]))
        signal = signal[chunk_size:]
    pieces.append(computer.finalize())
    return np.concatenate(pieces)
