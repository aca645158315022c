"""Frequency scales used to place filters (reference: scales.py:39-171)

Evaluated on the host, once, when a bank is constructed.
"""
import abc

import numpy as np

from .alias import AliasedFactory

__all__ = ["BarkScaling", "LinearScaling", "MelScaling", "OctaveScaling", "ScalingFunction"]


class ScalingFunction(AliasedFactory):
    """Invertible map between Hertz and some perceptual (or not) scale"""

    @abc.abstractmethod
    def scale_to_hertz(self, scale: float) -> float:
        pass

    @abc.abstractmethod
    def hertz_to_scale(self, hertz: float) -> float:
        pass


class LinearScaling(ScalingFunction):
    """``scale = (hertz - low_hz) * slope_hz`` (reference scales.py:53-78)"""

    aliases = {"linear", "uniform"}

    def __init__(self, low_hz: float, slope_hz: float = 1.0):
        self.low_hz = low_hz
        self.slope_hz = slope_hz

    def scale_to_hertz(self, scale):
        return scale / self.slope_hz + self.low_hz

    def hertz_to_scale(self, hertz):
        return (hertz - self.low_hz) * self.slope_hz


class OctaveScaling(ScalingFunction):
    """``scale = log2(hertz / low_hz)`` (reference scales.py:81-104)"""

    aliases = {"octave"}

    def __init__(self, low_hz: float):
        if low_hz <= 0:
            raise ValueError("low_hz must be positive")
        self.low_hz = low_hz

    def scale_to_hertz(self, scale):
        return (2 ** scale) * max(1e-10, self.low_hz)

    def hertz_to_scale(self, hertz):
        return np.log2(hertz / max(1e-10, self.low_hz))


class MelScaling(ScalingFunction):
    """O'Shaughnessy mel: ``1127 ln(1 + f / 700)`` (reference scales.py:107-125)"""

    aliases = {"mel"}

    def scale_to_hertz(self, scale):
        return 700.0 * (np.exp(scale / 1127.0) - 1.0)

    def hertz_to_scale(self, hertz):
        return 1127.0 * np.log(1 + hertz / 700.0)


class BarkScaling(ScalingFunction):
    """Traunmueller's Bark approximation with end corrections (reference scales.py:128-171)"""

    aliases = {"bark"}

    def scale_to_hertz(self, scale):
        if scale < 2:
            z = (20.0 * scale - 6.0) / 17.0
        elif scale > 20.1:
            z = (50.0 * scale + 221.1) / 61.0
        else:
            z = scale
        return 1960.0 * (z + 0.53) / (26.28 - z)

    def hertz_to_scale(self, hertz):
        z = 26.81 * hertz / (1960.0 + hertz) - 0.53
        if z < 2:
            return z + 0.15 * (2.0 - z)
        if z > 20.1:
            return z + 0.22 * (z - 20.1)
        return z
