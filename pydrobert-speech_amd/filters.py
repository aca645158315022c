"""Filter banks and window functions: the init-time tables of the STFT hot path.

Everything here runs once, on the host, in float64, when a frame computer is built
(reference: filters.py).  The GPU only ever sees the results: a window vector and a
sparse bin-weight table derived from ``get_truncated_response``.  Integer bin bounds
depend on float64 rounding of expressions such as ``ceil(width * hz / rate)``, so those
are evaluated on Python scalars in the reference's operation order (cited per method);
per-bin values are evaluated vectorised.

Banks: :class:`TriangularOverlappingFilterBank` (``tri``), :class:`Fbank` (``fbank``),
:class:`GaborFilterBank` (``gabor``), :class:`ComplexGammatoneFilterBank`
(``gammatone``).  Windows: Bartlett, Blackman, Hamming, Hann, Gamma.
"""
import abc
import math
from typing import Mapping, Optional, Tuple, Union

import numpy as np

from . import config
from .alias import AliasedFactory, alias_factory_subclass_from_arg
from .scales import MelScaling, ScalingFunction
from .util import angular_to_hertz, hertz_to_angular

__all__ = [
    "BartlettWindow",
    "BlackmanWindow",
    "ComplexGammatoneFilterBank",
    "Fbank",
    "GaborFilterBank",
    "GammaWindow",
    "HammingWindow",
    "HannWindow",
    "LinearFilterBank",
    "TriangularOverlappingFilterBank",
    "WindowFunction",
]

_TWO_PI = 2 * np.pi


def _half_width(width: int) -> int:
    # number of DFT bins in [0, pi] (reference filters.py:406-410)
    return (width + 1) // 2 if width % 2 else width // 2 + 1


def _uniform_scale_points(scaling_function, low_hz, high_hz, count, offset=0.0):
    # `count` points spaced uniformly on the scale between low_hz and high_hz, mapped
    # back to Hz, one scalar call per point (reference filters.py:300-306, 711-722)
    lo = scaling_function.hertz_to_scale(low_hz)
    hi = scaling_function.hertz_to_scale(high_hz)
    delta = (hi - lo) / (count - 1 if offset == 0.0 else count)
    return tuple(
        scaling_function.scale_to_hertz(lo + delta * (idx + offset)) for idx in range(count)
    )


class LinearFilterBank(AliasedFactory):
    """A fixed collection of LTI filters, lowest frequency first

    Interface of the reference's ``LinearFilterBank`` (filters.py:49-237).
    """

    @abc.abstractproperty
    def is_real(self) -> bool:
        """Whether the impulse responses are real"""

    @abc.abstractproperty
    def is_analytic(self) -> bool:
        """Whether the filters have (approximately) no negative-frequency part"""

    @abc.abstractproperty
    def is_zero_phase(self) -> bool:
        """Whether the frequency responses are real and even-centred in time"""

    @abc.abstractproperty
    def num_filts(self) -> int:
        """Number of filters"""

    @abc.abstractproperty
    def sampling_rate(self) -> float:
        """Samples per second of the target signals"""

    @abc.abstractproperty
    def supports_hz(self) -> Tuple[Tuple[float, float], ...]:
        """Per filter, the (low, high) Hz bounds outside which the response is ~0"""

    @abc.abstractproperty
    def supports(self) -> Tuple[Tuple[float, float], ...]:
        """Per filter, the (first, last) sample outside which the impulse response is ~0"""

    @property
    def supports_ms(self) -> Tuple[Tuple[float, float], ...]:
        rate = self.sampling_rate
        return tuple((lo * 1000 / rate, hi * 1000 / rate) for lo, hi in self.supports)

    @abc.abstractmethod
    def get_impulse_response(self, filt_idx: int, width: int) -> np.ndarray:
        """The filter in time, aliased into a buffer of `width` samples"""

    @abc.abstractmethod
    def get_frequency_response(
        self, filt_idx: int, width: int, half: bool = False
    ) -> np.ndarray:
        """The 2pi-periodised filter on a `width`-point DFT grid (``[0, pi]`` if `half`)"""

    @abc.abstractmethod
    def get_truncated_response(self, filt_idx: int, width: int) -> Tuple[int, np.ndarray]:
        """``(first_bin, values)`` of the non-zero stretch of the frequency response

        Real banks return the stretch inside ``[0, pi]``; complex banks may return a
        stretch that runs past `width` and wraps (reference filters.py:190-237).
        """


# --- triangular banks ---------------------------------------------------------------


class _TriangularVertexBank(LinearFilterBank):
    # shared state of the two banks whose filters are triangles between consecutive
    # vertex triples (reference filters.py:298-342 and 494-540)

    def _set_vertices(self, vertices, sampling_rate, analytic):
        self._vertices = tuple(vertices)
        self._rate = sampling_rate
        self._analytic = analytic

    @property
    def is_real(self) -> bool:
        return not self._analytic

    @property
    def is_analytic(self) -> bool:
        return self._analytic

    @property
    def is_zero_phase(self) -> bool:
        return True

    @property
    def num_filts(self) -> int:
        return len(self._vertices) - 2

    @property
    def sampling_rate(self) -> float:
        return self._rate

    @property
    def centers_hz(self) -> Tuple[float, ...]:
        """Frequency of maximal gain of each filter"""
        return self._vertices[1:-1]

    @property
    def supports_hz(self):
        return tuple(zip(self._vertices[:-2], self._vertices[2:]))

    def _triple(self, filt_idx):
        return self._vertices[filt_idx : filt_idx + 3]

    def _angular_triple(self, filt_idx):
        return tuple(hertz_to_angular(v, self._rate) for v in self._triple(filt_idx))

    def _bin_bounds(self, filt_idx, width):
        # reference filters.py:429-432 / 614-617
        left, _, right = self._triple(filt_idx)
        left_idx = int(np.ceil(width * left / self._rate))
        right_idx = int(width * right / self._rate)
        assert self._rate * (left_idx - 1) / width <= left
        assert self._rate * (right_idx + 1) / width >= right, width
        return left_idx, right_idx

    def _gain(self, filt_idx, bins, width):  # pragma: no cover - abstract
        raise NotImplementedError

    def get_frequency_response(self, filt_idx, width, half=False):
        left_idx, right_idx = self._bin_bounds(filt_idx, width)
        size = _half_width(width) if half else width
        res = np.zeros(size, dtype=np.float64)
        bins = np.arange(left_idx, min(size, right_idx + 1))
        if len(bins):
            gain = self._gain(filt_idx, bins, width)
            res[bins] = gain
            if not half and not self._analytic:
                res[-bins] = gain
        return res


class TriangularOverlappingFilterBank(_TriangularVertexBank):
    """Triangles in Hz whose vertices are uniform on a scale (``tri`` / ``triangular``)

    Reference: filters.py:240-440.
    """

    aliases = {"tri", "triangular"}

    def __init__(
        self,
        scaling_function: Union[ScalingFunction, Mapping, str],
        num_filts: int = 40,
        high_hz: Optional[float] = None,
        low_hz: float = 20.0,
        sampling_rate: float = 16000,
        analytic: bool = False,
    ):
        scaling_function = alias_factory_subclass_from_arg(ScalingFunction, scaling_function)
        nyquist = sampling_rate / 2
        if high_hz is None:
            high_hz = nyquist
        # 1 Hz of slack for serialisation round-off (reference filters.py:292-297)
        if not (0 <= low_hz < high_hz <= nyquist + 1):
            raise ValueError("Invalid frequency range: ({:.2f},{:.2f}".format(low_hz, high_hz))
        high_hz = min(high_hz, nyquist)
        self._set_vertices(
            _uniform_scale_points(scaling_function, low_hz, high_hz, num_filts + 2),
            sampling_rate,
            analytic,
        )

    @property
    def supports(self):
        # envelope bound 2(w_r - w_l) / ((w_c - w_l)(w_r - w_c) t^2 pi), reference
        # filters.py:345-358
        out = []
        for filt_idx in range(self.num_filts):
            left, mid, right = self._angular_triple(filt_idx)
            K = np.sqrt(8 * (right - left) / np.pi)
            K /= np.sqrt(config.EFFECTIVE_SUPPORT_THRESHOLD)
            K /= np.sqrt(mid - left) * np.sqrt(right - mid)
            K = int(np.ceil(K))
            out.append((-K // 2 - 1, K // 2 + 1))
        return tuple(out)

    def get_impulse_response(self, filt_idx, width):
        # closed-form inverse transform of a triangle (reference filters.py:360-393)
        left, mid, right = self._angular_triple(filt_idx)
        if right - mid > mid - left:
            denom, div_term = right - mid, mid - left
        else:
            denom, div_term = mid - left, right - mid
        denom *= (int(self._analytic) + 1) * np.pi
        t = np.arange(1, width + 1, dtype=np.float64)
        carrier = (lambda a: np.exp(1j * a * t)) if self._analytic else (lambda a: np.cos(a * t))
        numer = (right - left) / div_term * carrier(mid)
        numer = numer - (right - mid) / div_term * carrier(left)
        numer = numer - (mid - left) / div_term * carrier(right)
        val = numer / t ** 2
        res = np.zeros(width, dtype=np.complex128 if self._analytic else np.float64)
        inner = val[: width - 1]  # t = 1 .. width - 1
        if width > 1:
            res[1:width] += inner
            res[width - 1 : 0 : -1] += np.conj(inner)
        res[0] += val[width - 1]
        dc = mid / div_term * (right ** 2 - left ** 2)
        dc += right / div_term * (left ** 2 - mid ** 2)
        dc += left / div_term * (mid ** 2 - right ** 2)
        res[0] += dc / 2
        res /= denom
        return res

    def _gain(self, filt_idx, bins, width):
        left, mid, right = self._triple(filt_idx)
        hz = self._rate * bins / width
        return np.where(hz <= mid, (hz - left) / (mid - left), (right - hz) / (right - mid))

    def get_truncated_response(self, filt_idx, width):
        # reference filters.py:423-440 (length 1 + right - left even past `width`)
        left_idx, right_idx = self._bin_bounds(filt_idx, width)
        res = np.zeros(1 + right_idx - left_idx, dtype=np.float64)
        bins = np.arange(left_idx, min(width, right_idx + 1))
        if len(bins):
            res[bins - left_idx] = self._gain(filt_idx, bins, width)
        return left_idx, res


class Fbank(_TriangularVertexBank):
    """Kaldi/HTK-style bank: square root of triangles in mel (``fbank``)

    The square root is there because this package squares *after* filtering
    (reference filters.py:443-626).
    """

    aliases = {"fbank"}

    def __init__(
        self,
        num_filts: int = 40,
        high_hz: Optional[float] = None,
        low_hz: float = 20.0,
        sampling_rate: float = 16000,
        analytic: bool = False,
    ):
        if low_hz < 0 or (high_hz and (high_hz <= low_hz or high_hz > sampling_rate // 2)):
            raise ValueError("Invalid frequency range: ({:.2f},{:.2f}".format(low_hz, high_hz))
        if high_hz is None:
            high_hz = sampling_rate // 2
        self._mel = MelScaling()
        self._set_vertices(
            _uniform_scale_points(self._mel, low_hz, high_hz, num_filts + 2),
            sampling_rate,
            analytic,
        )

    @property
    def supports(self):
        # reference filters.py:542-560
        eps = config.EFFECTIVE_SUPPORT_THRESHOLD
        out = []
        for filt_idx in range(self.num_filts):
            left, mid, right = self._angular_triple(filt_idx)
            K = right - left + 2 * ((right - mid) * (mid - left)) ** 2
            K /= eps ** 2 * np.pi
            K /= (right - mid) * (mid - left)
            K /= np.sqrt(eps)
            K /= np.sqrt(mid - left) * np.sqrt(right - mid)
            K **= 0.3333
            K = int(np.ceil(K))
            out.append((-K // 2 - 1, K // 2 + 1))
        return tuple(out)

    def get_impulse_response(self, filt_idx, width):
        # inverse DFT of the sampled response (reference filters.py:562-569)
        if self.is_analytic:
            return np.fft.ifft(self.get_frequency_response(filt_idx, width, half=False))
        return np.fft.irfft(self.get_frequency_response(filt_idx, width, half=True), n=width)

    def _triangle_in_mel(self, filt_idx, bins, width):
        to_mel = self._mel.hertz_to_scale
        left_mel, mid_mel, right_mel = (to_mel(v) for v in self._triple(filt_idx))
        mel = to_mel(self._rate * bins / width)
        rising = (mel - left_mel) / (mid_mel - left_mel)
        falling = (right_mel - mel) / (right_mel - mid_mel)
        return np.where(mel <= mid_mel, rising, falling)

    def _gain(self, filt_idx, bins, width):
        return self._triangle_in_mel(filt_idx, bins, width) ** 0.5

    def get_truncated_response(self, filt_idx, width):
        # reference filters.py:604-626 (length clipped at `width`, unlike `tri`)
        left_idx, right_idx = self._bin_bounds(filt_idx, width)
        stop = min(width, right_idx + 1)
        res = np.zeros(stop - left_idx, dtype=np.float64)
        bins = np.arange(left_idx, stop)
        if len(bins):
            res[:] = self._triangle_in_mel(filt_idx, bins, width)
        return left_idx, res ** 0.5


# --- complex banks ------------------------------------------------------------------


def _check_complex_bank_range(low_hz, high_hz, sampling_rate):
    # reference filters.py:702-707 / 983-988
    if low_hz < 0 or (high_hz and (high_hz <= low_hz or high_hz > sampling_rate // 2)):
        raise ValueError("Invalid frequency range: ({:.2f},{:.2f}".format(low_hz, high_hz))


class GaborFilterBank(LinearFilterBank):
    r"""Gabor filters whose neighbours cross at their ERB / 3 dB points (``gabor``)

    :math:`\hat f(\omega) = C\sqrt{2\sigma}\pi^{1/4} e^{-\sigma^2(\xi-\omega)^2/2}`.
    Reference: filters.py:629-900.
    """

    aliases = {"gabor"}

    def __init__(
        self,
        scaling_function: Union[ScalingFunction, Mapping, str],
        num_filts: int = 40,
        high_hz: Optional[float] = None,
        low_hz: float = 20.0,
        sampling_rate: float = 16000,
        scale_l2_norm: bool = False,
        erb: bool = False,
    ):
        scaling_function = alias_factory_subclass_from_arg(ScalingFunction, scaling_function)
        _check_complex_bank_range(low_hz, high_hz, sampling_rate)
        self._scale_l2_norm = scale_l2_norm
        self._erb = erb
        self._rate = sampling_rate
        if high_hz is None:
            high_hz = sampling_rate // 2
        # neighbouring filters intersect at points uniform on the scale, with a half
        # step of margin at either end (reference filters.py:714-722)
        edges = _uniform_scale_points(scaling_function, low_hz, high_hz, num_filts + 1, 0.5)
        log_2, log_pi = np.log(2), np.log(np.pi)
        t_const = -2 * np.log(config.EFFECTIVE_SUPPORT_THRESHOLD)
        f_const = t_const
        if scale_l2_norm:
            f_const += log_2 + 0.5 * log_pi
            t_const -= 0.5 * log_pi
        else:
            t_const -= log_2 + log_pi
        bandwidth = np.sqrt(np.pi) / 2 if erb else np.sqrt(3 / 10 * np.log(10))
        centers_hz, centers_ang, stds = [], [], []
        supports_ang, wrap_widths, supports = [], [], []
        self._wrap_below = False
        for lo_edge, hi_edge in zip(edges[:-1], edges[1:]):
            center_hz = (lo_edge + hi_edge) / 2
            center_ang = hertz_to_angular(center_hz, sampling_rate)
            std = bandwidth / hertz_to_angular(center_hz - lo_edge, sampling_rate)
            log_std = np.log(std)
            if scale_l2_norm:
                reach = np.sqrt(log_std + f_const) / std
                wrap_reach = np.sqrt(log_std + f_const + log_2) / std
                samps = int(np.ceil(std * np.sqrt(t_const - log_std)))
            else:
                reach = np.sqrt(f_const) / std
                wrap_reach = np.sqrt(f_const + log_2) / std
                samps = int(np.ceil(std * np.sqrt(t_const - 2 * log_std)))
            if center_ang - reach < 0:
                self._wrap_below = True
            centers_hz.append(center_hz)
            centers_ang.append(center_ang)
            stds.append(std)
            supports_ang.append((center_ang - reach, center_ang + reach))
            wrap_widths.append(2 * wrap_reach)
            supports.append((-samps, samps))
        self._centers_hz = tuple(centers_hz)
        self._centers_ang = tuple(centers_ang)
        self._stds = tuple(stds)
        self._supports_ang = tuple(supports_ang)
        self._wrap_supports_ang = tuple(wrap_widths)
        self._supports = tuple(supports)
        self._supports_hz = tuple(
            (angular_to_hertz(lo, sampling_rate), angular_to_hertz(hi, sampling_rate))
            for lo, hi in supports_ang
        )

    @property
    def is_real(self) -> bool:
        return False

    @property
    def is_analytic(self) -> bool:
        return not self._wrap_below

    @property
    def is_zero_phase(self) -> bool:
        return True

    @property
    def num_filts(self) -> int:
        return len(self._centers_hz)

    @property
    def sampling_rate(self) -> float:
        return self._rate

    @property
    def centers_hz(self):
        return self._centers_hz

    @property
    def supports_hz(self):
        return self._supports_hz

    @property
    def supports(self):
        return self._supports

    @property
    def scaled_l2_norm(self) -> bool:
        return self._scale_l2_norm

    @property
    def erb(self) -> bool:
        return self._erb

    def get_impulse_response(self, filt_idx, width):
        # reference filters.py:823-839
        center_ang, std = self._centers_ang[filt_idx], self._stds[filt_idx]
        if self._scale_l2_norm:
            const_term = -0.5 * np.log(std) - 0.25 * np.log(np.pi)
        else:
            const_term = -0.5 * np.log(2 * np.pi) - np.log(std)
        t = np.arange(width + 1)
        val = np.exp(-(t ** 2) / (2 * std ** 2) + const_term + 1j * center_ang * t)
        res = np.zeros(width, dtype=np.complex128)
        res[:width] += val[:width]
        if width:
            # conj(val[t]) lands at index -t for t = 1..width (t = width wraps to 0)
            np.add.at(res, (-t[1:]) % width, np.conj(val[1:]))
        return res

    def _periodised_gaussian(self, filt_idx, bins, width, periods):
        center_ang, std = self._centers_ang[filt_idx], self._stds[filt_idx]
        if self._scale_l2_norm:
            const_term = 0.5 * np.log(2 * std) + 0.25 * np.log(np.pi)
        else:
            const_term = 0
        num_term = -(std ** 2) / 2
        res = np.zeros(len(bins), dtype=np.float64)
        for period in periods:
            omega = (bins / width + period) * 2 * np.pi
            res += np.exp(num_term * (center_ang - omega) ** 2 + const_term)
        return res

    def get_frequency_response(self, filt_idx, width, half=False):
        # reference filters.py:841-868
        lowest, highest = self._supports_ang[filt_idx]
        size = _half_width(width) if half else width
        periods = range(
            -1 - int(max(-lowest, 0) / _TWO_PI), 2 + int(highest / _TWO_PI)
        )
        return self._periodised_gaussian(filt_idx, np.arange(size), width, periods)

    def get_truncated_response(self, filt_idx, width):
        # reference filters.py:870-900; when even the half-threshold support spans a
        # whole period, every bin is "in support"
        if self._wrap_supports_ang[filt_idx] >= _TWO_PI:
            return 0, self.get_frequency_response(filt_idx, width)
        lowest, highest = self._supports_ang[filt_idx]
        left_idx = int(np.ceil(width * lowest / _TWO_PI))
        right_idx = int(width * highest / _TWO_PI)
        periods = range(-int(max(-lowest, 0) / _TWO_PI), 1 + int(highest / _TWO_PI))
        bins = np.arange(left_idx, right_idx + 1)
        return left_idx % width, self._periodised_gaussian(filt_idx, bins, width, periods)


class ComplexGammatoneFilterBank(LinearFilterBank):
    r"""Gammatone filters with complex carriers (``gammatone`` / ``tonebank``)

    :math:`h(t) = c\,t^{n-1} e^{-\alpha t + i\xi t}u(t)`,
    :math:`H(\omega) = c\,(n-1)!\,/\,(\alpha + i(\omega - \xi))^n`.
    Reference: filters.py:903-1211.
    """

    aliases = {"gammatone", "tonebank"}

    def __init__(
        self,
        scaling_function: Union[ScalingFunction, Mapping, str],
        num_filts: int = 40,
        high_hz: Optional[float] = None,
        low_hz: float = 20.0,
        sampling_rate: float = 16000,
        order: int = 4,
        max_centered: bool = False,
        scale_l2_norm: bool = False,
        erb: bool = False,
    ):
        scaling_function = alias_factory_subclass_from_arg(ScalingFunction, scaling_function)
        _check_complex_bank_range(low_hz, high_hz, sampling_rate)
        if not isinstance(order, int) or order <= 0:
            raise ValueError("order must be a positive integer")
        self._scale_l2_norm = scale_l2_norm
        self._erb = erb
        self._order = order
        self._rate = sampling_rate
        if high_hz is None:
            high_hz = sampling_rate // 2
        edges = _uniform_scale_points(scaling_function, low_hz, high_hz, num_filts + 1, 0.5)
        log_eps = np.log(config.EFFECTIVE_SUPPORT_THRESHOLD)
        log_double_fact = np.log(math.factorial(2 * order - 2))
        log_fact = np.log(math.factorial(order - 1))
        log_2 = np.log(2)
        if erb:
            alpha_const = log_2 * (2 * order - 1)
            alpha_const += 2 * log_fact
            alpha_const -= log_double_fact
        else:
            alpha_const = -0.5 * np.log(4 * (2 ** (1 / order)) - 4)
        self._centers_hz, self._xis, self._alphas, self._cs = [], [], [], []
        self._offsets, self._supports = [], []
        self._supports_ang, self._wrap_supports_ang = [], []
        self._wrap_below = False
        for lo_edge, hi_edge in zip(edges[:-1], edges[1:]):
            center_hz = (lo_edge + hi_edge) / 2
            xi = hertz_to_angular(center_hz, sampling_rate)
            log_alpha = alpha_const + np.log(hertz_to_angular(hi_edge - lo_edge, sampling_rate))
            alpha = np.exp(log_alpha)
            if scale_l2_norm:
                log_c = 0.5 * (log_2 + log_alpha + log_double_fact)
                log_c -= order * (log_alpha + log_2)
            else:
                log_c = order * log_alpha - log_fact
            offset = -(order - 1) / alpha if max_centered else 0
            supp_a = (2 / order) * (log_c + log_fact - log_eps)
            wrap_supp_a = supp_a + (2 / order) * log_2
            supp_b = np.exp(2 * log_alpha)
            reach = (np.exp(supp_a) - supp_b) ** 0.5
            wrap_reach = (np.exp(wrap_supp_a) - supp_b) ** 0.5
            self._centers_hz.append(center_hz)
            self._xis.append(xi)
            self._alphas.append(alpha)
            self._cs.append(np.exp(log_c))
            self._offsets.append(offset)
            self._supports.append(self._temporal_support(len(self._xis) - 1))
            self._supports_ang.append((xi - reach, xi + reach))
            if xi - reach < 0:
                self._wrap_below = True
            self._wrap_supports_ang.append(2 * wrap_reach)
        for name in (
            "_centers_hz", "_xis", "_alphas", "_cs", "_offsets", "_supports",
            "_supports_ang", "_wrap_supports_ang",
        ):
            setattr(self, name, tuple(getattr(self, name)))
        self._supports_hz = tuple(
            (angular_to_hertz(lo, sampling_rate), angular_to_hertz(hi, sampling_rate))
            for lo, hi in self._supports_ang
        )

    @property
    def is_real(self) -> bool:
        return False

    @property
    def is_analytic(self) -> bool:
        return not self._wrap_below

    @property
    def is_zero_phase(self) -> bool:
        return False

    @property
    def num_filts(self) -> int:
        return len(self._centers_hz)

    @property
    def order(self) -> int:
        return self._order

    @property
    def sampling_rate(self) -> float:
        return self._rate

    @property
    def centers_hz(self):
        return self._centers_hz

    @property
    def supports_hz(self):
        return self._supports_hz

    @property
    def supports(self):
        return self._supports

    @property
    def scaled_l2_norm(self) -> bool:
        return self._scale_l2_norm

    @property
    def erb(self) -> bool:
        return self._erb

    def _h(self, t, idx):
        # impulse response at (possibly fractional) sample t (reference filters.py:1163-1174)
        offset = self._offsets[idx]
        if t <= offset:
            return 0j
        r = np.log(self._cs[idx]) + (self._order - 1) * np.log(t - offset)
        r += (-self._alphas[idx] + 1j * self._xis[idx]) * (t - offset)
        return np.exp(r)

    def _H(self, omega, idx):
        # frequency response at angular frequencies omega (reference filters.py:1176-1185)
        numer = np.exp(-1j * omega * self._offsets[idx]) * self._cs[idx]
        numer = numer * math.factorial(self._order - 1)
        return numer / (self._alphas[idx] + 1j * (omega - self._xis[idx])) ** self._order

    def _temporal_support(self, idx):
        # Newton iteration down the envelope's tail until it drops under the threshold
        # (reference filters.py:1187-1211)
        alpha, c, offset, n = self._alphas[idx], self._cs[idx], self._offsets[idx], self._order
        eps = config.EFFECTIVE_SUPPORT_THRESHOLD
        if n == 1:
            right = int(np.ceil((np.log(c) - np.log(eps) / alpha)))
        else:
            right = (n - 1 + np.sqrt((n - 1) / 2)) / alpha
            mag = np.abs(self._h(right, idx))
            while mag > eps:
                slope = c * np.exp(-alpha * right) * right ** (n - 2)
                slope *= (n - 1) - alpha * right
                right -= mag / slope
                mag = np.abs(self._h(right, idx))
        return (int(np.floor(offset)), int(np.ceil(right) + offset))

    def get_impulse_response(self, filt_idx, width):
        # reference filters.py:1116-1125
        left_sup, right_sup = self.supports[filt_idx]
        res = np.zeros(width, dtype=np.complex128)
        for period in range(int(np.floor(left_sup / width)), int(np.ceil(right_sup / width)) + 1):
            for idx in range(width):
                res[idx] += self._h(period * width + idx, filt_idx)
        return res

    def get_frequency_response(self, filt_idx, width, half=False):
        # reference filters.py:1127-1144
        left_sup, right_sup = self._supports_ang[filt_idx]
        size = _half_width(width) if half else width
        res = np.zeros(size, dtype=np.complex128)
        omega = np.arange(size, dtype=np.float64) * 2 * np.pi / width
        for period in range(
            int(np.floor(left_sup / 2 / np.pi)), int(np.ceil(right_sup / 2 / np.pi)) + 1
        ):
            res += self._H(omega + 2 * np.pi * period, filt_idx)
        return res

    def get_truncated_response(self, filt_idx, width):
        # reference filters.py:1146-1161
        left_sup, right_sup = self._supports_ang[filt_idx]
        if right_sup - left_sup + self._wrap_supports_ang[filt_idx] >= _TWO_PI:
            return 0, self.get_frequency_response(filt_idx, width)
        left_idx = int(np.ceil(width * left_sup / _TWO_PI))
        right_idx = int(width * right_sup / _TWO_PI)
        omega = np.arange(left_idx, right_idx + 1, dtype=np.float64)
        omega *= 2 * np.pi / width
        return left_idx % width, self._H(omega, filt_idx)


# --- windows ------------------------------------------------------------------------


class WindowFunction(AliasedFactory):
    """A real low-pass window (reference filters.py:1217-1234)"""

    @abc.abstractmethod
    def get_impulse_response(self, width: int) -> np.ndarray:
        """The window as a float64 vector of length `width`"""


class _NumpyWindow(WindowFunction):
    # numpy window divided by (gain at DC) * max(1, width - 1), so that the window sums
    # to about one (reference filters.py:1247-1298)
    _shape = None
    _dc_gain = 1.0

    def get_impulse_response(self, width):
        window = type(self)._shape(width)
        window /= self._dc_gain * max(1, width - 1)
        return window


class BartlettWindow(_NumpyWindow):
    aliases = {"bartlett", "triangular", "tri"}
    _shape = staticmethod(np.bartlett)
    _dc_gain = 0.5

    def get_impulse_response(self, width):
        # the reference divides by ``max(1, width - 1) / 2`` (filters.py:1249); keep
        # that exact operation order
        window = np.bartlett(width)
        window /= max(1, width - 1) / 2
        return window


class BlackmanWindow(_NumpyWindow):
    aliases = {"blackman", "black"}
    _shape = staticmethod(np.blackman)
    _dc_gain = 0.42


class HammingWindow(_NumpyWindow):
    aliases = {"hamming"}
    _shape = staticmethod(np.hamming)
    _dc_gain = 0.54


class HannWindow(_NumpyWindow):
    aliases = {"hanning", "hann"}
    _shape = staticmethod(np.hanning)
    _dc_gain = 0.5


class GammaWindow(WindowFunction):
    r"""Time-reversed Gamma envelope :math:`t^{n-1}e^{-\alpha t}` peaking at ``peak * width``

    Default window of causal frames (reference filters.py:1301-1349).
    """

    aliases = {"gamma"}

    def __init__(self, order: int = 4, peak: float = 0.75):
        self.order = order
        self.peak = peak

    def get_impulse_response(self, width):
        if width <= 0:
            return np.array([], dtype=float)
        if width == 1:
            return np.array([1], dtype=float)
        peak = self.peak * width
        ret = np.arange(width - 1, -1, -1, dtype=float)
        if self.order > 1:
            alpha = (self.order - 1) / (width - peak)
            offs = width - 1  # the last sample (t = 0) stays 0
        else:
            alpha = 5 / width
            offs = width
        ln_c = self.order * np.log(alpha)
        ln_c -= np.log(math.factorial(self.order - 1))
        ret[:offs] = ret[:offs] ** (self.order - 1) * np.exp(-alpha * ret[:offs] + ln_c)
        return ret
