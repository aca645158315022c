"""``torch.nn.Module`` faces of the computers and processors (reference torch.py:73-520).

The reference ships a PyTorch port of its numpy code so that features can be computed inside a
``torch`` pipeline; its drivers wrap every object in one (command_line.py:553-577).  Here the
computation already lives on the device, so the modules are thin: a tensor in (CPU or GPU, any
floating dtype), the same kernels as the numpy-facing classes, a tensor out on the input's
device with the input's dtype.  No autograd: like the reference's, these are feature
extractors, and the kernels have no backward.

* ``PyTorchSTFTFrameComputer.from_stft_frame_computer(computer)``
* ``PyTorchSIFrameComputer.from_si_frame_computer(computer)``
* ``PyTorchPreemphasize(coeff)`` / ``.from_preemphasize(p)``, ``PyTorchDither(coeff)`` / ``.from_dither(d)``
* ``PyTorchPostProcessorWrapper.from_postprocessor(p)``
"""
import torch

from .compute import ShortTimeFourierTransformFrameComputer
from .post import PostProcessor
from .pre import Dither, Preemphasize
from .si import ShortIntegrationFrameComputer

__all__ = [
    "PyTorchDither",
    "PyTorchPostProcessorWrapper",
    "PyTorchPreemphasize",
    "PyTorchSIFrameComputer",
    "PyTorchSTFTFrameComputer",
    "PyTorchShortIntegrationFrameComputer",
    "PyTorchShortTimeFourierTransformFrameComputer",
    "pytorch_dither",
    "pytorch_preemphasize",
]


def _on_device(sig: torch.Tensor) -> torch.Tensor:
    if not sig.is_floating_point():
        raise ValueError("expected a floating-point tensor")
    work = sig if sig.dtype in (torch.float32, torch.float64) else sig.to(torch.float32)
    return work.to("cuda") if not work.is_cuda else work


def _like(result: torch.Tensor, sig: torch.Tensor) -> torch.Tensor:
    return result.to(device=sig.device, dtype=sig.dtype)


def pytorch_preemphasize(sig: torch.Tensor, coeff: float = 0.97) -> torch.Tensor:
    """``out[i] = sig[i] - coeff * sig[i - 1]`` along the last axis (torch.py:73-76)"""
    with torch.no_grad():
        return _like(Preemphasize(coeff).apply(_on_device(sig)), sig)


def pytorch_dither(sig: torch.Tensor, coeff: float = 1.0) -> torch.Tensor:
    """``sig + coeff * N(0, 1)`` (torch.py:103-105); noise from the device's counter-based generator"""
    with torch.no_grad():
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())  # follows torch.manual_seed, like randn_like
        return _like(Dither(coeff, seed=seed).apply(_on_device(sig)), sig)


class PyTorchPreemphasize(torch.nn.Module):
    def __init__(self, coeff: float = 0.97) -> None:
        super().__init__()
        self.coeff = coeff

    @classmethod
    def from_preemphasize(cls, preemphasize: Preemphasize):
        return cls(preemphasize.coeff)

    def forward(self, sig: torch.Tensor) -> torch.Tensor:
        return pytorch_preemphasize(sig, self.coeff)


class PyTorchDither(torch.nn.Module):
    def __init__(self, coeff: float = 1.0):
        super().__init__()
        self.coeff = coeff

    @classmethod
    def from_dither(cls, dither: Dither):
        return cls(dither.coeff)

    def forward(self, sig: torch.Tensor) -> torch.Tensor:
        return pytorch_dither(sig, self.coeff)


class _ComputerModule(torch.nn.Module):
    """A frame computer as a module: ``forward(signal)`` = ``compute_full(signal)``"""

    def __init__(self, computer):
        super().__init__()
        self.computer = computer

    def forward(self, signal: torch.Tensor) -> torch.Tensor:
        if signal.dim() != 1:
            raise RuntimeError("Expected signal to be 1-dimensional")  # torch.py:172-173
        with torch.no_grad():
            return _like(self.computer.compute_full(_on_device(signal).contiguous()), signal)

    # tables live in the native plan, not in torch buffers (the reference's SI module does the
    # same, torch.py:504-509)
    def state_dict(self, *args, **kwargs):
        return {}

    def load_state_dict(self, *args, **kwargs):
        return None


class PyTorchShortTimeFourierTransformFrameComputer(_ComputerModule):
    @classmethod
    def from_stft_frame_computer(cls, computer: ShortTimeFourierTransformFrameComputer):
        if not isinstance(computer, ShortTimeFourierTransformFrameComputer):
            raise TypeError("expected a ShortTimeFourierTransformFrameComputer")
        return cls(computer)


class PyTorchShortIntegrationFrameComputer(_ComputerModule):
    @classmethod
    def from_si_frame_computer(cls, computer: ShortIntegrationFrameComputer):
        if not isinstance(computer, ShortIntegrationFrameComputer):
            raise TypeError("expected a ShortIntegrationFrameComputer")
        return cls(computer)


class PyTorchPostProcessorWrapper(torch.nn.Module):
    """``forward(feats)`` = ``postprocessor.apply(feats)`` (default axis, torch.py:435-472)"""

    def __init__(self, postprocessor: PostProcessor):
        super().__init__()
        self.postprocessor = postprocessor

    @classmethod
    def from_postprocessor(cls, postprocessor: PostProcessor):
        return cls(postprocessor)

    def forward(self, sig: torch.Tensor) -> torch.Tensor:
        with torch.no_grad():
            return _like(self.postprocessor.apply(_on_device(sig)), sig)


PyTorchSTFTFrameComputer = PyTorchShortTimeFourierTransformFrameComputer
PyTorchSIFrameComputer = PyTorchShortIntegrationFrameComputer
