"""Frequency scales: where a bank's filters go (evaluated on the host at construction).

The four maps of the reference (scales.py:39-171) with the same aliases and constructor
arguments, so that ``{"name": "tri", "scaling_function": "mel"}`` style configs resolve:

=========  =======================================================  ====================
alias      Hertz -> scale                                            arguments
=========  =======================================================  ====================
linear     ``(f - low_hz) * slope_hz``                               low_hz, slope_hz=1
octave     ``log2(f / low_hz)``                                      low_hz > 0
mel        ``1127 ln(1 + f / 700)`` (O'Shaughnessy)                  --
bark       Traunmueller's ``26.81 f / (1960 + f) - 0.53`` with the   --
           end corrections below 2 and above 20.1 Bark
=========  =======================================================  ====================

Vertices are computed one scalar at a time by the banks, so these functions receive and
return plain floats (numpy scalars from ``numpy.log`` / ``numpy.exp``); arrays work for the
branch-free ones too.
"""
import abc

import numpy as np

from .alias import AliasedFactory

__all__ = ["BarkScaling", "LinearScaling", "MelScaling", "OctaveScaling", "ScalingFunction"]

_MEL_FACTOR, _MEL_CORNER = 1127.0, 700.0
_BARK_A, _BARK_B, _BARK_C = 26.81, 1960.0, 0.53
_BARK_D = 26.28  # = A - C, spelled out so the inverse rounds like the reference's
_BARK_LO, _BARK_HI = 2.0, 20.1
_TINY = 1e-10


class ScalingFunction(AliasedFactory):
    """An invertible map between Hertz and a (perceptual or not) scale"""

    @abc.abstractmethod
    def hertz_to_scale(self, hertz: float) -> float:
        """Position of frequency `hertz` on the scale"""

    @abc.abstractmethod
    def scale_to_hertz(self, scale: float) -> float:
        """Frequency at position `scale`"""


class LinearScaling(ScalingFunction):
    """Scale 0 at `low_hz`, growing by `slope_hz` per Hertz"""

    aliases = {"linear", "uniform"}

    def __init__(self, low_hz: float, slope_hz: float = 1.0):
        self.low_hz, self.slope_hz = low_hz, slope_hz

    def hertz_to_scale(self, hertz):
        return (hertz - self.low_hz) * self.slope_hz

    def scale_to_hertz(self, scale):
        return scale / self.slope_hz + self.low_hz


class OctaveScaling(ScalingFunction):
    """Octaves above `low_hz` (which must be positive)"""

    aliases = {"octave"}

    def __init__(self, low_hz: float):
        if low_hz <= 0:
            raise ValueError("low_hz must be positive")
        self.low_hz = low_hz

    def _anchor(self):
        return max(_TINY, self.low_hz)

    def hertz_to_scale(self, hertz):
        return np.log2(hertz / self._anchor())

    def scale_to_hertz(self, scale):
        return (2 ** scale) * self._anchor()


class MelScaling(ScalingFunction):
    """The mel scale in O'Shaughnessy's closed form"""

    aliases = {"mel"}

    def hertz_to_scale(self, hertz):
        return _MEL_FACTOR * np.log(1 + hertz / _MEL_CORNER)

    def scale_to_hertz(self, scale):
        return _MEL_CORNER * (np.exp(scale / _MEL_FACTOR) - 1.0)


class BarkScaling(ScalingFunction):
    """The Bark scale after Traunmueller, with his corrections at both ends"""

    aliases = {"bark"}

    def hertz_to_scale(self, hertz):
        z = _BARK_A * hertz / (_BARK_B + hertz) - _BARK_C
        if z < _BARK_LO:
            z = z + 0.15 * (_BARK_LO - z)
        elif z > _BARK_HI:
            z = z + 0.22 * (z - _BARK_HI)
        return z

    def scale_to_hertz(self, scale):
        z = scale
        if scale < _BARK_LO:  # undo z + 0.15 (2 - z)
            z = (20.0 * scale - 6.0) / 17.0
        elif scale > _BARK_HI:  # undo z + 0.22 (z - 20.1)
            z = (50.0 * scale + 221.1) / 61.0
        return _BARK_B * (z + _BARK_C) / (_BARK_D - z)
