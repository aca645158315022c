"""TEST INFRASTRUCTURE -- CPU restatement of the reference's ShortIntegrationFrameComputer.

Not part of the product: only ``tests/`` (and ``bench.py``'s ``cpu_baseline`` leg) may import this.
Parity status: **pinned** by ``tests/golden/si.npz`` (outputs of the reference itself for five
configurations x seven lengths x {f32, f64} and two chunkings; ``tests/golden/make_golden_si.py``).

The reference (compute.py:613-996) computes the features with a streaming overlap-save
convolution.  Stripped of the buffering, what it computes for a whole signal is

    y_f[i]  = sum_{k < M} g_f[k] * sig[i + start - k]        (sig = 0 outside the signal)
    z_f[i]  = |y_f[i]|^2  or  |y_f[i]|                        (compute.py:909-912)
    out[t]  = sum_{m < 2S} window[m] * z_f[t S + m]           (compute.py:913-932, 980-988)
    out[t]  = log(max(out[t], floor))                         (compute.py:989-990)

with ``g_f`` the translated impulse response clamped to the longest support ``M``
(compute.py:711-722), ``S`` the frame shift, and ``start`` the stream position of the first
integrated sample: the samples "skipped" at the start of the stream (compute.py:859-865,
872-885) minus the virtual zeros a short translation puts in front of it.  This is the form the
reference's own test checks the streaming code against (tests/test_compute.py:129-171).  What
remains of the buffering is the NUMBER of frames, which depends on the block arithmetic of
``compute_chunk`` / ``finalize`` and is restated literally in :func:`frame_count`.
"""
from dataclasses import dataclass
from typing import List, Tuple

import numpy as np

LOG_FLOOR = 1e-5  # config.LOG_FLOOR_VALUE (config.py:27-29)


@dataclass
class SiParams:
    frame_shift: int          # S
    max_support: int          # M
    translation: int          # compute.py:682-696
    dft_size: int             # only the frame count depends on it (block arithmetic)
    taps: np.ndarray          # [C, M] complex or real: energy "dirac" first if include_energy
    window: np.ndarray        # [2 S]
    centered: bool
    use_power: bool
    use_log: bool

    @property
    def frame_length(self) -> int:  # compute.py:698
        return self.max_support + self.frame_shift - 1


def stream_start(p: SiParams) -> Tuple[int, int]:
    """``(skip, lead)``: samples consumed before integration starts, virtual leading zeros

    compute.py:859-865: centered frames skip ``translation - S`` samples, or, if that is
    negative, count that many zeros as already buffered; causal frames skip ``translation``.
    """
    skip = p.translation - p.frame_shift if p.centered else p.translation
    return (skip, 0) if skip >= 0 else (0, -skip)


def _chunk_frames(num_raw: int, y_rem: int, S: int) -> int:
    """compute.py:792: frames a chunk yields once `num_raw` samples wait to be integrated"""
    return max(0, (num_raw + y_rem) // S - 1)


def frame_count(n: int, p: SiParams) -> int:
    """Rows of ``compute_full`` for a signal of `n` samples (compute.py:781-855)"""
    S = p.frame_shift
    skip, lead = stream_start(p)
    consumed = min(skip, n)                       # _handle_skip (compute.py:872-885)
    skip_left = skip - consumed
    num_raw = lead + n - consumed                 # compute.py:790
    first = _chunk_frames(num_raw, 0, S)
    waiting = num_raw - first * S                 # x_rem + y_rem after the chunk
    borrowed = S if p.centered else 0             # compute.py:831-838
    buf_len = p.translation - skip_left + waiting - borrowed
    want = max(0, (buf_len + S // 2) // S)        # compute.py:841
    if want < 1:
        return first
    pad_right = (want - 1) * S + p.frame_length - buf_len  # compute.py:843-844
    pad_raw = pad_right - min(skip_left, pad_right)
    # the zero chunk continues the same stream: `waiting` samples are already there
    more = max(0, (waiting + pad_raw) // S - 1)
    return first + min(want, more)


def stream_frame_counts(chunk_lengths: List[int], p: SiParams) -> List[int]:
    """Frames returned by each ``compute_chunk`` call and by ``finalize`` (last entry)"""
    S = p.frame_shift
    skip, lead = stream_start(p)
    waiting, counts = lead, []
    for n in chunk_lengths:
        consumed = min(skip, n)
        skip -= consumed
        waiting += n - consumed
        k = max(0, waiting // S - 1)
        counts.append(k)
        waiting -= k * S
    borrowed = S if p.centered else 0
    buf_len = p.translation - skip + waiting - borrowed
    want = max(0, (buf_len + S // 2) // S)
    last = 0
    if want >= 1:
        pad_right = (want - 1) * S + p.frame_length - buf_len
        pad_raw = pad_right - min(skip, pad_right)
        last = min(want, max(0, (waiting + pad_raw) // S - 1))
    return counts + [last]


def features(signal: np.ndarray, num_frames: int, p: SiParams, first_frame: int = 0) -> np.ndarray:
    """Frames ``first_frame .. first_frame + num_frames`` of the closed form, float64 arithmetic

    The result is cast to the signal's dtype (compute.py:797, 857).
    """
    S, M = p.frame_shift, p.max_support
    C = p.taps.shape[0]
    sig = np.asarray(signal, dtype=np.float64)
    out = np.empty((num_frames, C), dtype=np.float64)
    if num_frames:
        skip, lead = stream_start(p)
        start = skip - lead
        lo = first_frame * S                       # first integrated sample needed
        count = (num_frames + 1) * S               # integrated samples needed
        # sig[lo + start - (M - 1) .. lo + start + count - 1], zeros outside the signal
        a = lo + start - (M - 1)
        seg = np.zeros(count + M - 1, dtype=np.float64)
        s0, s1 = max(a, 0), min(a + len(seg), len(sig))
        if s1 > s0:
            seg[s0 - a : s1 - a] = sig[s0:s1]
        win = sliding = np.lib.stride_tricks.sliding_window_view
        for c in range(C):
            y = np.convolve(seg, p.taps[c])[M - 1 : M - 1 + count]
            z = (y * y.conj()).real if p.use_power else np.abs(y)
            frames = win(z, 2 * S)[::S][:num_frames]
            out[:, c] = frames @ p.window
        if p.use_log:
            out = np.log(np.maximum(out, LOG_FLOOR))
    return out.astype(np.asarray(signal).dtype if np.asarray(signal).dtype.kind == "f" else np.float64)


def compute_full(signal: np.ndarray, p: SiParams) -> np.ndarray:
    """``ShortIntegrationFrameComputer.compute_full`` (compute.py:852-855)"""
    return features(signal, frame_count(len(signal), p), p)
