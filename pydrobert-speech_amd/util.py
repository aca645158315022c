"""Small helpers shared by the banks (reference: util.py:108-115)"""
import numpy as np

__all__ = ["angular_to_hertz", "hertz_to_angular"]


def hertz_to_angular(hertz: float, samp_rate: float) -> float:
    """cycles/sec -> radians/sample"""
    return hertz * 2 * np.pi / samp_rate


def angular_to_hertz(angle: float, samp_rate: float) -> float:
    """radians/sample -> cycles/sec"""
    return angle * samp_rate / (2 * np.pi)
