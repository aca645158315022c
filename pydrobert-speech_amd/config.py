"""Package constants (reference: config.py:27-52)

`LOG_FLOOR_VALUE` is read when an STFT plan is created (the reference reads the module
global at call time, compute.py:397,459; a plan snapshots it -- change the value, then
build a new computer).  `EFFECTIVE_SUPPORT_THRESHOLD` is read when a bank is built.
"""

__all__ = ["EFFECTIVE_SUPPORT_THRESHOLD", "LOG_FLOOR_VALUE", "USE_FFTPACK"]

#: kept for API compatibility; the DFT is always computed on the GPU here
USE_FFTPACK: bool = False

#: magnitude below which a filter response counts as zero (reference config.py:43)
EFFECTIVE_SUPPORT_THRESHOLD: float = 5e-4

#: floor applied before every logarithm (reference config.py:52)
LOG_FLOOR_VALUE: float = 1e-5
