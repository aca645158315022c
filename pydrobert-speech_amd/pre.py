"""Pre-processors: ``Preemphasize`` and ``Dither`` (reference pre.py:39-149), on the GPU.

These sit immediately before the hot path in the reference's drivers
(command_line.py:127-128, 346-348).  Both are one element-wise pass over the signal;
pre-emphasis can instead be fused into the STFT kernels' frame load
(``STFTFrameComputer.compute_full_batch(signals, preemphasis=0.97)``), which saves the pass.
Intermediate arithmetic is float64 and the result is cast back to the input dtype, as in the
reference.  ``Dither`` draws its noise from a counter-based generator on the device (Philox),
so it matches the reference statistically, not sample for sample (the reference's own test
checks the standard deviation only, tests/test_pre.py:6-12).
"""
import abc
import warnings
from typing import Optional

import numpy as np

from . import _native
from .alias import AliasedFactory

__all__ = ["Dither", "PreProcessor", "Preemphasize"]

_AXIS_DEP_MSG = (
    "Specifying axis in preprocessor.apply is deprecated. "
    "Preprocessors should be applied to 1D signals only."
)


class PreProcessor(AliasedFactory):
    """A transform applied to a signal before framing (reference pre.py:39-64)"""

    @abc.abstractmethod
    def apply(self, signal, axis: Optional[int] = None, in_place: bool = False):
        pass


def _device_copy(signal):
    """(tensor on the GPU in float32/float64, was_gpu, original numpy dtype or None)"""
    torch = _native.require_device()
    if getattr(signal, "is_cuda", False):
        t = signal if signal.dtype in (torch.float32, torch.float64) else signal.to(torch.float64)
        return t.contiguous(), True, None
    arr = np.asarray(signal)
    work = arr if arr.dtype in (np.float32, np.float64) else arr.astype(np.float64)
    return torch.from_numpy(np.array(work, order="C", copy=True)).to("cuda"), False, arr.dtype


def _finish(t, was_gpu, dtype, original, in_place):
    if was_gpu:
        return t
    res = t.cpu().numpy()
    if res.dtype != dtype:
        res = res.astype(dtype)
    if in_place and isinstance(original, np.ndarray) and original.flags.writeable:
        original[...] = res
        return original
    return res


class Dither(PreProcessor):
    """Add Gaussian noise of standard deviation `coeff` (aliases ``dither``, ``dithering``)

    `seed` makes the noise reproducible; each :func:`apply` advances it.
    """

    aliases = {"dither", "dithering"}

    def __init__(self, coeff: float = 1.0, seed: Optional[int] = None):
        super().__init__()
        self.coeff = coeff
        self._seed = int(np.random.SeedSequence(seed).generate_state(1, np.uint64)[0])

    def apply(self, signal, axis: Optional[int] = None, in_place: bool = False):
        if axis is not None:
            warnings.warn(_AXIS_DEP_MSG, DeprecationWarning)
        torch = _native.require_device()
        t, was_gpu, dtype = _device_copy(signal)
        shape = tuple(t.shape)
        if axis is not None and len(shape) > 1:
            # one noise value per position along `axis`, shared by the other axes
            # (reference pre.py:96-99): dither a vector of that length and broadcast-add
            noise = torch.zeros(shape[axis], dtype=t.dtype, device=t.device)
            noise = self._launch(noise, noise)
            view = [1] * len(shape)
            view[axis] = shape[axis]
            out = t + noise.reshape(view)
        else:
            out = self._launch(t, t if (was_gpu and in_place) else torch.empty_like(t))
        return _finish(out, was_gpu, dtype, signal, in_place)

    def _launch(self, src, dst):
        torch = _native.require_device()
        lib = _native.lib()
        fn = lib.pds_dither_f32 if src.dtype == torch.float32 else lib.pds_dither_f64
        self._seed = (self._seed * 6364136223846793005 + 1442695040888963407) % (1 << 64)
        with torch.cuda.device(src.device):
            rc = fn(src.data_ptr(), src.numel(), float(self.coeff), self._seed, dst.data_ptr(),
                    torch.cuda.current_stream(src.device).cuda_stream)
        _native.check(rc, "pds_dither")
        return dst


class Preemphasize(PreProcessor):
    """``new[i] = old[i] - coeff * old[i-1]``, ``new[0] = old[0]`` (aliases ``preemphasize``,
    ``preemphasis``, ``preemph``)"""

    aliases = {"preemphasize", "preemphasis", "preemph"}

    def __init__(self, coeff: float = 0.97):
        super().__init__()
        self.coeff = coeff

    def apply(self, signal, axis: Optional[int] = None, in_place: bool = False):
        if axis is not None:
            warnings.warn(_AXIS_DEP_MSG, DeprecationWarning)
        torch = _native.require_device()
        lib = _native.lib()
        t, was_gpu, dtype = _device_copy(signal)
        if t.dim() == 0 or t.numel() == 0:
            return _finish(t.clone(), was_gpu, dtype, signal, in_place)
        moved = axis not in (-1, None) and t.dim() > 1
        if moved:
            t = t.movedim(axis, -1).contiguous()
        rows, n = t.numel() // t.shape[-1], t.shape[-1]
        out = torch.empty_like(t)
        offs = torch.arange(rows, dtype=torch.int64, device=t.device) * n
        lens = torch.full((rows,), n, dtype=torch.int64, device=t.device)
        fn = lib.pds_preemphasize_f32 if t.dtype == torch.float32 else lib.pds_preemphasize_f64
        with torch.cuda.device(t.device):
            for lo in range(0, rows, 65535):
                hi = min(rows, lo + 65535)
                rc = fn(t.data_ptr(), offs[lo:].data_ptr(), lens[lo:].data_ptr(), hi - lo, n,
                        float(self.coeff), out.data_ptr(),
                        torch.cuda.current_stream(t.device).cuda_stream)
                _native.check(rc, "pds_preemphasize")
        if moved:
            out = out.movedim(-1, axis)
        return _finish(out, was_gpu, dtype, signal, in_place)

    def apply_packed(self, signal, offsets, lengths):
        """Pre-emphasise every utterance of a packed GPU buffer (see ``compute_packed``)"""
        torch = _native.require_device()
        lib = _native.lib()
        if not getattr(signal, "is_cuda", False) or signal.dim() != 1:
            raise ValueError("signal must be a 1-D GPU tensor")
        signal = signal.contiguous()
        offs = torch.as_tensor(np.asarray(offsets, dtype=np.int64)).to(signal.device)
        lens_h = np.asarray(lengths, dtype=np.int64)
        lens = torch.as_tensor(lens_h).to(signal.device)
        out = signal.clone()  # gaps between utterances keep their contents
        fn = lib.pds_preemphasize_f32 if signal.dtype == torch.float32 else lib.pds_preemphasize_f64
        B = len(lens_h)
        with torch.cuda.device(signal.device):
            for lo in range(0, B, 65535):
                hi = min(B, lo + 65535)
                rc = fn(signal.data_ptr(), offs[lo:].data_ptr(), lens[lo:].data_ptr(), hi - lo,
                        int(lens_h[lo:hi].max()) if hi > lo else 0, float(self.coeff),
                        out.data_ptr(), torch.cuda.current_stream(signal.device).cuda_stream)
                _native.check(rc, "pds_preemphasize")
        return out
