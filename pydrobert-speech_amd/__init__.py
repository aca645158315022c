"""MI355X-native STFT filter-bank features behind the pydrobert-speech plugin API.

One hot path of sdrobert/pydrobert-speech, rebuilt for gfx950:
``ShortTimeFourierTransformFrameComputer.compute_full`` (framing, window, real DFT,
``|.|^2``, filter-bank integration, log / energy) and the ``Deltas`` / ``CMVN``
post-processors, computed by hand-written HIP kernels reached through a C ABI
(``include/pds_amd.h``) with ``ctypes``.  The class names, aliases, constructor
arguments and JSON ``alias`` configuration surface are the reference's, so

>>> from pydrobert_speech_amd.alias import alias_factory_subclass_from_arg
>>> from pydrobert_speech_amd.compute import FrameComputer
>>> computer = alias_factory_subclass_from_arg(FrameComputer, json.load(open("fbank.json")))
>>> feats = computer.compute_full(signal)

works as it does with ``pydrobert.speech``.  There is no CPU fallback: computing
without the built library or without a HIP device raises.
"""
from . import alias, compute, config, filters, post, pre, scales, util  # noqa: F401
from . import si  # noqa: F401  (registers the "si" computer with the FrameComputer aliases)
from . import feed  # noqa: F401

# the reference keeps both computers in one module (compute.py:613, 996)
compute.ShortIntegrationFrameComputer = si.ShortIntegrationFrameComputer
compute.SIFrameComputer = si.SIFrameComputer
from ._native import LIB_PATH, NativeError  # noqa: F401

__all__ = ["alias", "compute", "config", "feed", "filters", "post", "pre", "scales", "si", "util"]
# ``pydrobert_speech_amd.torch`` (nn.Module faces) and ``.command_line`` import torch: on demand
__version__ = "0.1.0"
