"""``ShortIntegrationFrameComputer`` (alias ``si``): filter, rectify, integrate over short windows.

Host side of the reference's second frame computer (compute.py:613-996; SURVEY.md section 8(f)
rank 4).  The constructor derives what the reference's does -- frame shift, frame style, the
2S-sample integration window, the longest filter support ``M``, the translation that aligns the
filters inside it, the DFT size it would have used -- and turns every filter into ``M`` FIR taps
(the impulse response rolled into place and clamped, compute.py:711-722).  The arithmetic

    y_f[i] = sum_k taps[f][k] sig[i + start - k],   out[t][f] = log(max(sum_m w[m] |y_f[tS + m]|^p, floor))

runs in ``csrc/si.hip`` on the GPU; there is no CPU path.  The reference evaluates the same sums
with a streaming overlap-save FFT; how many frames a stream yields depends on that block
arithmetic (compute.py:781-850) and is restated in :func:`num_frames` / the streaming methods.
"""
import ctypes
from typing import List, Mapping, Optional, Sequence, Union

import numpy as np

from . import _native, config
from ._staging import PinnedStaging
from .alias import alias_factory_subclass_from_arg
from .compute import _MAX_UTTS_PER_CALL, LinearFilterBankFrameComputer
from .filters import GammaWindow, HannWindow, LinearFilterBank, WindowFunction

__all__ = ["ShortIntegrationFrameComputer", "SIFrameComputer"]


class _SiPlan:
    """Owner of a ``pds_si_plan``"""

    def __init__(self, desc, taps, window):
        _native.require_device()
        lib = _native.lib()
        self._keep = (np.ascontiguousarray(taps), np.ascontiguousarray(window, np.float64))
        handle = ctypes.c_void_p()
        rc = lib.pds_si_plan_create(
            ctypes.byref(desc), self._keep[0].ctypes.data, self._keep[1].ctypes.data, ctypes.byref(handle)
        )
        _native.check(rc, "pds_si_plan_create")
        self.handle = handle

    def __del__(self):
        handle, self.handle = getattr(self, "handle", None), None
        if handle:
            try:
                _native.lib().pds_si_plan_destroy(handle)
            except Exception:  # interpreter shutdown
                pass


class ShortIntegrationFrameComputer(LinearFilterBankFrameComputer):
    """Short-time integration of the rectified outputs of a filter bank

    Same constructor arguments, properties and streaming interface as the reference's class
    (compute.py:613-672): `bank`, `frame_shift_ms` (also the half length of the integration
    window), `frame_style` (default: ``'centered'`` for zero-phase banks, else ``'causal'``),
    `include_energy`, `pad_to_nearest_power_of_two` (only :attr:`dft_size` depends on it here),
    `window_function` (default: Gamma for causal frames, Hann otherwise), `use_power`, `use_log`.
    """

    aliases = {"si"}

    def __init__(
        self,
        bank: Union[LinearFilterBank, Mapping, str],
        frame_shift_ms: float = 10,
        frame_style: Optional[str] = None,
        include_energy: bool = False,
        pad_to_nearest_power_of_two: bool = True,
        window_function: Optional[Union[WindowFunction, Mapping, str]] = None,
        use_power: bool = False,
        use_log: bool = True,
    ):
        bank = alias_factory_subclass_from_arg(LinearFilterBank, bank)
        super().__init__(bank, include_energy=include_energy)
        rate = bank.sampling_rate
        S = int(0.001 * frame_shift_ms * rate)
        if frame_style is None:
            frame_style = "centered" if bank.is_zero_phase else "causal"
        elif frame_style not in ("centered", "causal"):
            raise ValueError('Invalid frame style: "{}"'.format(frame_style))
        if window_function is None:
            window_function = GammaWindow() if frame_style == "causal" else HannWindow()
        else:
            window_function = alias_factory_subclass_from_arg(WindowFunction, window_function)
        supports = list(bank.supports)
        if frame_style == "centered":
            # every filter is re-centred on the middle of the longest support (compute.py:682-686)
            M = max(right - left for left, right in supports)
            translation = M // 2
        else:
            # shift right until no filter starts before 0; that stretch counts as support
            # (compute.py:687-696)
            translation = max([0] + [-left for left, _ in supports])
            M = max([0] + [right for _, right in supports]) + translation
        frame_length = M + S - 1
        narrowest_hz = min(hi - lo for lo, hi in bank.supports_hz)
        # the transform the reference convolves with: long enough for a frame and fine enough to
        # resolve the narrowest filter (compute.py:698-706)
        dft_size = max(frame_length, int(np.ceil(2 * rate / narrowest_hz)))
        if pad_to_nearest_power_of_two:
            dft_size = int(2 ** np.ceil(np.log2(dft_size)))
        rows = []
        if include_energy:
            dirac = np.zeros(M, dtype=np.float64)  # a unit impulse returns the (translated) signal
            dirac[translation] = 1
            rows.append(dirac)
        for f, (left, right) in enumerate(supports):
            response = bank.get_impulse_response(f, dft_size)
            shift = translation - (left + right) // 2 + 1 if frame_style == "centered" else translation
            rows.append(np.roll(response, shift)[:M])
        taps = np.stack(rows) if rows else np.zeros((0, M))
        self._real = bool(bank.is_real)
        self._taps = np.ascontiguousarray(taps.real if self._real else taps.astype(np.complex128))
        self._rate, self._frame_shift, self._frame_style = rate, S, frame_style
        self._max_support, self._translation = M, translation
        self._frame_length, self._dft_size = frame_length, dft_size
        self._window = np.asarray(window_function.get_impulse_response(2 * S), dtype=np.float64).reshape(2, S)
        self._power, self._log = bool(use_power), bool(use_log)
        self._log_floor = float(config.LOG_FLOOR_VALUE)
        skip = translation - S if frame_style == "centered" else translation
        # samples consumed before integration starts / virtual zeros in front (compute.py:859-865)
        self._skip0, self._lead = (skip, 0) if skip >= 0 else (0, -skip)
        self._plans = {}  # device index -> _SiPlan (tables live on one GPU)
        self._staging = PinnedStaging()  # pinned buffers of compute_full_batch (allocated on first use)
        self._reset_stream()

    # -- properties (compute.py:746-778) ------------------------------------------------------

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
    def dft_size(self) -> int:
        return self._dft_size

    @property
    def fft_size(self) -> int:
        """Transform size of the float32 overlap-save kernel for this bank (0: direct filtering)"""
        return int(_native.lib().pds_si_plan_fft_size(self._native_plan().handle))

    @property
    def taps(self) -> np.ndarray:
        """``(num_coeffs, max_support)`` FIR taps, the energy impulse first if included"""
        return self._taps

    # -- frame bookkeeping --------------------------------------------------------------------

    def _tail_frames(self, waiting: int, skip_left: int) -> int:
        """Frames ``finalize`` adds when `waiting` samples are buffered (compute.py:826-850)"""
        S = self._frame_shift
        borrowed = S if self._frame_style == "centered" else 0
        buf_len = self._translation - skip_left + waiting - borrowed
        want = max(0, (buf_len + S // 2) // S)
        if want < 1:
            return 0
        pad_right = (want - 1) * S + self._frame_length - buf_len
        pad_raw = pad_right - min(skip_left, pad_right)
        return min(want, max(0, (waiting + pad_raw) // S - 1))

    def num_frames(self, num_samples: int) -> int:
        """Rows ``compute_full`` returns for a signal of `num_samples` samples"""
        S = self._frame_shift
        consumed = min(self._skip0, num_samples)
        waiting = self._lead + num_samples - consumed
        first = max(0, waiting // S - 1)  # compute.py:792
        return first + self._tail_frames(waiting - first * S, self._skip0 - consumed)

    # -- device ----------------------------------------------------------------------------------

    def _native_plan(self, device=None) -> _SiPlan:
        """The plan for `device` (a torch device or index; default: the current device)"""
        torch = _native.require_device()
        index = torch.cuda.current_device() if device is None else torch.device(device).index
        if index is None:
            index = torch.cuda.current_device()
        if index not in self._plans:
            desc = _native.SiDesc(
                frame_shift=self._frame_shift, max_support=self._max_support, num_coeffs=self.num_coeffs,
                taps_complex=int(not self._real), use_power=int(self._power), use_log=int(self._log),
                reserved=0, reserved2=0, log_floor=self._log_floor,
            )
            taps = self._taps if self._real else self._taps.view(np.float64)
            with torch.cuda.device(index):  # the plan's tables are allocated on the current device
                self._plans[index] = _SiPlan(desc, taps, self._window.reshape(-1))
        return self._plans[index]

    def _launch(self, signal, meta, lo, hi, max_frames, start, out, direct=False):
        """One native call for utterances ``lo:hi`` of the device index rows `meta`"""
        torch = _native.require_device()
        lib = _native.lib()
        plan = self._native_plan(signal.device)
        stream = torch.cuda.current_stream(signal.device).cuda_stream
        args = (plan.handle, signal.data_ptr(), meta[0, lo:].data_ptr(), meta[1, lo:].data_ptr(),
                meta[2, lo:].data_ptr(), meta[3, lo:].data_ptr(), hi - lo, max_frames, start)
        if signal.dtype == torch.float64:
            rc = lib.pds_si_batch_f64(*args, out.data_ptr(), out.stride(0), stream)
        else:
            # float32: the overlap-save FFT form when the plan has it (it needs scratch memory),
            # else -- or on request -- direct time-domain filtering
            need = 0 if direct else int(lib.pds_si_scratch_len(plan.handle, hi - lo, max_frames))
            scratch = torch.empty(need, dtype=torch.float32, device=signal.device) if need else None
            rc = lib.pds_si_batch_f32(*args, scratch.data_ptr() if need else None, out.data_ptr(),
                                      out.stride(0), stream)
        _native.check(rc, "pds_si_batch")

    def compute_packed(self, signal, offsets, lengths, nframes=None, first_frame: int = 0, out=None,
                       direct: bool = False):
        """Features of a packed batch that is already on the GPU

        `signal`: contiguous 1-D float32 / float64 GPU tensor; utterance b is
        ``signal[offsets[b] : offsets[b] + lengths[b]]``.  `nframes` defaults to
        :func:`num_frames` of each length; `first_frame` continues every utterance at that frame
        (streaming); `direct` forces time-domain filtering for float32 too (cross-checks).
        Returns ``(feats, row_offsets)`` like the STFT computer's method.
        """
        torch = _native.require_device()
        if not signal.is_cuda or signal.dim() != 1 or not signal.is_contiguous():
            raise ValueError("signal must be a contiguous 1-D tensor on the GPU")
        if signal.dtype not in (torch.float32, torch.float64):
            raise TypeError("signal must be float32 or float64")
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
        if B and int((offsets + lengths).max()) > signal.numel():
            raise ValueError("an utterance lies outside the signal buffer")
        rows = np.zeros(B + 1, dtype=np.int64)
        np.cumsum(nframes, out=rows[1:])
        total = int(rows[-1])
        if out is None:
            out = torch.empty((total, self.num_coeffs), dtype=signal.dtype, device=signal.device)
        elif (out.dtype != signal.dtype or out.dim() != 2 or out.shape[0] < total
              or out.shape[1] < self.num_coeffs or out.stride(1) != 1):
            raise ValueError("out has the wrong dtype, shape or strides")
        if total == 0:
            return out, rows
        meta = torch.from_numpy(np.stack([offsets, lengths, nframes, rows[:-1]])).to(signal.device)
        start = self._skip0 - self._lead + int(first_frame) * self._frame_shift
        with torch.cuda.device(signal.device):
            for lo in range(0, B, _MAX_UTTS_PER_CALL):
                hi = min(B, lo + _MAX_UTTS_PER_CALL)
                self._launch(signal, meta, lo, hi, int(nframes[lo:hi].max()), start, out, direct)
        return out, rows

    @staticmethod
    def _work_dtype(dtype) -> np.dtype:
        dtype = np.dtype(dtype)
        if not np.issubdtype(dtype, np.floating):
            raise ValueError("Chunk must be a float type")  # compute.py:863-864
        return dtype if dtype in (np.float32, np.float64) else np.dtype(np.float32)

    def _frames_of(self, host: np.ndarray, nframes: int, first_frame: int = 0) -> np.ndarray:
        torch = _native.require_device()
        in_dtype = host.dtype
        if nframes <= 0:
            return np.empty((0, self.num_coeffs), dtype=in_dtype)
        work = np.ascontiguousarray(host, dtype=self._work_dtype(in_dtype))
        feats, _ = self.compute_packed(torch.from_numpy(work).to("cuda"), [0], [len(work)], [nframes], first_frame)
        return feats.cpu().numpy().astype(in_dtype, copy=False)

    def compute_full(self, signal: np.ndarray) -> np.ndarray:
        if self._started:
            raise ValueError("Already started computing frames")
        if getattr(signal, "is_cuda", False):
            feats, _ = self.compute_packed(signal.contiguous(), [0], [signal.numel()])
            return feats
        signal = np.asarray(signal)
        self._work_dtype(signal.dtype)
        return self._frames_of(signal.reshape(-1), self.num_frames(signal.size))

    def compute_full_batch(self, signals: Sequence) -> List[np.ndarray]:
        """``[compute_full(s) for s in signals]`` with one launch for the whole list"""
        torch = _native.require_device()
        if not len(signals):
            return []
        if self._started:
            raise ValueError("Already started computing frames")
        in_dtype = np.asarray(signals[0]).dtype
        work = self._work_dtype(in_dtype)
        lengths = np.asarray([np.asarray(s).size for s in signals], dtype=np.int64)
        offsets = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
        staging = self._staging
        if staging.try_acquire(int(offsets[-1]) * work.itemsize):
            # pinned buffers both ways, the utterances packed by a few threads (_staging.py)
            try:
                d_sig = staging.upload(signals, offsets, work, torch.device("cuda", torch.cuda.current_device()))
                feats, rows = self.compute_packed(d_sig, offsets[:-1], lengths)
                feats = staging.download(feats).astype(in_dtype, copy=False)
            finally:
                staging.release()
            return [feats[rows[b] : rows[b + 1]] for b in range(len(signals))]
        host = np.zeros(0, work)
        if offsets[-1]:
            host = np.concatenate([np.asarray(s, dtype=work).reshape(-1) for s in signals])
        feats, rows = self.compute_packed(torch.from_numpy(host).to("cuda"), offsets[:-1], lengths)
        feats = feats.cpu().numpy().astype(in_dtype, copy=False)
        return [feats[rows[b] : rows[b + 1]] for b in range(len(signals))]

    # -- streaming (compute.py:781-885) ---------------------------------------------------------
    #
    # The reference keeps a DFT-sized ring of input and per-block accumulators.  Here the part of
    # the stream that frames not yet emitted can still reach is kept as one host array; a call
    # emits exactly the frames the reference's block arithmetic emits and computes them with the
    # batch kernel, continued at the right frame.

    def _reset_stream(self):
        self._started = False
        self._tail = None       # samples from stream position _tail_at on
        self._tail_at = 0
        self._done = 0          # frames emitted so far
        self._waiting = 0       # integrated samples received but not yet framed (x_rem + y_rem)
        self._skip_left = 0
        self._stream_dtype = None

    def _emit(self, count: int) -> np.ndarray:
        if count <= 0:
            return np.empty((0, self.num_coeffs), dtype=self._stream_dtype)
        S = self._frame_shift
        # shift the frame origin so that the kept tail starts at stream position 0 for the kernel
        start = self._skip0 - self._lead + self._done * S - self._tail_at
        work = np.ascontiguousarray(self._tail, dtype=self._work_dtype(self._stream_dtype))
        torch = _native.require_device()
        d_sig = torch.from_numpy(work).to("cuda") if len(work) else torch.zeros(1, dtype=torch.from_numpy(work).dtype, device="cuda")
        out = torch.empty((count, self.num_coeffs), dtype=d_sig.dtype, device="cuda")
        meta = torch.tensor([[0], [len(work)], [count], [0]], dtype=torch.int64, device="cuda")
        self._launch(d_sig, meta, 0, 1, count, start, out)
        self._done += count
        # samples before the first one the next frame's filters can reach are no longer needed
        keep_from = max(self._tail_at, self._skip0 - self._lead + self._done * S - (self._max_support - 1))
        if keep_from > self._tail_at:
            self._tail = self._tail[keep_from - self._tail_at :]
            self._tail_at = keep_from
        return out.cpu().numpy().astype(self._stream_dtype, copy=False)

    def compute_chunk(self, chunk: np.ndarray) -> np.ndarray:
        chunk = np.asarray(chunk).reshape(-1)
        if self._started:
            if chunk.dtype != self._stream_dtype:
                raise ValueError("Chunk does not share a type with previous chunks")
        else:
            self._work_dtype(chunk.dtype)
            self._reset_stream()
            self._stream_dtype = chunk.dtype
            self._tail = np.zeros(0, dtype=chunk.dtype)
            self._skip_left, self._waiting = self._skip0, self._lead
            self._started = True
        consumed = min(self._skip_left, len(chunk))
        self._skip_left -= consumed
        self._waiting += len(chunk) - consumed
        self._tail = np.concatenate([self._tail, chunk])
        count = max(0, self._waiting // self._frame_shift - 1)  # compute.py:792
        self._waiting -= count * self._frame_shift
        return self._emit(count)

    def finalize(self) -> np.ndarray:
        if not self._started:
            return np.empty((0, self.num_coeffs), dtype=np.float64)
        feats = self._emit(self._tail_frames(self._waiting, self._skip_left))
        self._reset_stream()
        return feats


SIFrameComputer = ShortIntegrationFrameComputer
