"""Helpers shared by the banks, and the signal readers either side of the hot path.

* ``hertz_to_angular`` / ``angular_to_hertz`` (reference util.py:108-115), ``gauss_quant``
  (util.py:54-104) and ``circshift_fourier`` (util.py:118-185): host-side helpers the reference
  exports beside them;
* ``read_signal`` (reference util.py:362-510): same selection rules, arguments and error
  behaviour for the sources that need nothing outside this image -- ``wav`` (scipy, else the
  standard ``wave`` module), ``npy``, ``npz``, ``pt``, ``file`` (``numpy.fromfile``).  Kaldi
  tables / objects, HDF5, NIST SPHERE and libsndfile types are recognised and rejected with
  the ``ImportError`` their reader would raise in an environment without the optional package
  (pydrobert-kaldi, h5py, soundfile are absent here; SPHERE decoding is out of scope).

Torch files are loaded with ``weights_only=True``: a signal or feature file is a tensor, and
nothing from the file is executed.
"""
import math
import re
from typing import Any, BinaryIO, Optional, Union

import numpy as np

__all__ = ["angular_to_hertz", "circshift_fourier", "gauss_quant", "hertz_to_angular", "read_signal",
           "SIGNAL_SOURCES"]


def hertz_to_angular(hertz: float, samp_rate: float) -> float:
    """cycles/sec -> radians/sample"""
    return hertz * 2 * np.pi / samp_rate


def angular_to_hertz(angle: float, samp_rate: float) -> float:
    """radians/sample -> cycles/sec"""
    return angle * samp_rate / (2 * np.pi)


def _standard_normal_quantile(p: float) -> float:
    """z with Phi(z) = p, by Newton steps on ``math.erfc`` from a bracketing start (no scipy)"""
    if not 0.0 <= p <= 1.0:
        return float("nan")
    if p in (0.0, 1.0):
        return -math.inf if p == 0.0 else math.inf
    tail = min(p, 1.0 - p)  # solve in the lower tail, mirror afterwards
    z = -math.sqrt(-2.0 * math.log(tail))  # below the root: Phi(z) < tail for every tail < 1/2
    for _ in range(60):
        cdf = 0.5 * math.erfc(-z / math.sqrt(2.0))
        pdf = math.exp(-0.5 * z * z) / math.sqrt(2.0 * math.pi)
        step = (cdf - tail) / pdf
        z -= step
        if abs(step) <= 1e-15 * max(1.0, abs(z)):
            break
    return z if p < 0.5 else -z


def gauss_quant(p: float, mu: float = 0, std: float = 1) -> float:
    """Quantile function (inverse CDF) of a Gaussian with mean `mu` and deviation `std`

    ``scipy.special.ndtri`` when scipy is importable (the reference uses ``scipy.stats.norm.ppf``,
    the same function; util.py:73-78), else Newton iterations on the error function (the
    reference falls back to a rational approximation good to ~1e-8 instead; util.py:54-70).
    """
    try:
        from scipy.special import ndtri

        z = float(ndtri(p))
    except ImportError:
        z = _standard_normal_quantile(float(p))
    return z * std + mu


def circshift_fourier(filt: np.ndarray, shift: float, start_idx: int = 0, dft_size: Optional[int] = None,
                      copy: bool = True) -> np.ndarray:
    """Shift a filter circularly by `shift` samples in time, given its frequency response

    The shift theorem: bin k of the response is multiplied by ``exp(-2 pi i k shift / dft_size)``.
    `filt` may be a truncated response whose first entry is bin `start_idx` of a `dft_size`-point
    transform (default ``len(filt) + start_idx``; bins wrap modulo `dft_size`).  Returns a
    complex128 array; with ``copy=False`` a complex128 `filt` is modified in place and returned
    (reference util.py:118-185 -- which takes ``shift % dft_size`` before applying the default and
    so needs `dft_size` given; here the default is resolved first).
    """
    if dft_size is None:
        dft_size = len(filt) + start_idx
    shift = shift % dft_size
    bins = np.arange(start_idx, start_idx + len(filt)) % dft_size
    ramp = np.exp(-2j * np.pi * shift / dft_size * bins)
    if copy or filt.dtype != np.complex128:
        return filt * ramp
    filt *= ramp
    return filt


# ------------------------------------------------------------------ readers ----------


def _cast(data, dtype):
    return data.astype(dtype) if dtype else data


def _read_wav(source, dtype, key, **kwargs):
    try:
        from scipy.io import wavfile
    except ImportError:
        wavfile = None
    if wavfile is not None:
        return _cast(wavfile.read(source, **kwargs)[1], dtype)
    import wave

    with wave.open(source, **kwargs) as handle:
        channels, width = handle.getnchannels(), handle.getsampwidth()
        raw = handle.readframes(handle.getnframes())
    data = np.frombuffer(raw, dtype="<i%d" % width)
    if len(data) % channels:
        raise IOError("Number of channels do not evenly divide wave samples")
    if channels > 1:
        data = data.reshape(-1, channels)
    return _cast(data, dtype)


def _read_npy(source, dtype, key, **kwargs):
    return _cast(np.load(source, **kwargs), dtype)


def _read_npz(source, dtype, key, **kwargs):
    with np.load(source, **kwargs) as archive:
        return _cast(archive[key if key else "arr_0"], dtype)


def _read_pt(source, dtype, key, **kwargs):
    import torch

    kwargs.setdefault("weights_only", True)
    return _cast(torch.load(source, map_location="cpu", **kwargs).numpy(), dtype)


def _read_raw(source, dtype, key, **kwargs):
    if dtype:
        kwargs["dtype"] = dtype
    return np.fromfile(source, **kwargs)


def _needs(package, what):
    def reader(source, dtype, key, **kwargs):
        raise ImportError(f"reading {what} requires the package '{package}', which is not available")

    return reader


#: ``force_as`` value -> reader ``(source, dtype, key, **kwargs)``
SIGNAL_SOURCES = {
    "wav": _read_wav,
    "npy": _read_npy,
    "npz": _read_npz,
    "pt": _read_pt,
    "file": _read_raw,
    "table": _needs("pydrobert-kaldi", "a Kaldi table"),
    "kaldi": _needs("pydrobert-kaldi", "a Kaldi object"),
    "hdf5": _needs("h5py", "an HDF5 archive"),
    "soundfile": _needs("soundfile", "libsndfile audio"),
    "sph": _needs("soundfile", "a NIST SPHERE file"),
}

_BY_SUFFIX = (("wav", "wav"), ("hdf5", "hdf5"), ("npy", "npy"), ("npz", "npz"), ("pt", "pt"), ("sph", "sph"))


def _source_kind(rfilename: str) -> Optional[str]:
    """Selection order of the reference (util.py:338-359): Kaldi rspecifier, suffix, pipe"""
    if re.match(r"^(ark|scp)(,\w+)*:", rfilename):
        return "table"
    for suffix, kind in _BY_SUFFIX:
        if rfilename.endswith("." + suffix):
            return kind
    if rfilename.endswith("|"):
        return "kaldi"
    return None


def read_signal(
    rfilename: Union[str, BinaryIO],
    dtype=None,
    key: Any = None,
    force_as: Optional[str] = None,
    **kwargs,
) -> np.ndarray:
    """Read an array (a signal, a feature matrix, CMVN statistics) from a file

    `rfilename` is a path or, with `force_as` set, an open binary file.  The source type
    comes from `force_as` or from the name: ``.wav``, ``.npy``, ``.npz`` (entry `key`, default
    ``'arr_0'``), ``.pt``; ``force_as='file'`` reads raw binary with :func:`numpy.fromfile`.
    `dtype`, if set, is the type of the returned array; other keyword arguments go to the
    underlying reader.

    Raises :class:`ValueError` for a stream without `force_as` or an unknown `force_as`,
    :class:`IOError` when no rule matches the name (there is no catch-all, as in the
    reference since v0.2.0), :class:`ImportError` for a recognised type whose package is absent.
    """
    if not isinstance(rfilename, str):
        if force_as is None:
            raise ValueError("cannot infer type from IO stream. Set force_as")
        if force_as in ("kaldi", "table"):
            raise ValueError("kaldi types can't be inferred without a string rspecifier")
    elif force_as is None:
        force_as = _source_kind(rfilename)
        if force_as is None:
            raise IOError(f"Unable to infer how to read '{rfilename}'. Set force_as")
    reader = SIGNAL_SOURCES.get(force_as)
    if reader is None:
        raise ValueError(f"force_as ('{force_as}') is not one of {sorted(SIGNAL_SOURCES)}.")
    return reader(rfilename, dtype, key, **kwargs)
