"""Post-processors on the hot path: ``Deltas`` and ``Standardize`` / ``CMVN``.

Same classes, aliases, arguments and error behaviour as the reference's (post.py:38-491);
the element-wise and stencil arithmetic runs in HIP kernels (``csrc/post.hip``) -- there
is no CPU path.  Host code only decides shapes, pads for the non-default ``pad_mode`` s,
and finalises the ``O(num_coeffs)`` mean/scale vectors of CMVN.

``Stack`` moves data only: views for a single tensor, one copy
kernel for a packed ragged batch.  CMVN statistics files (``rfilename``, ``save``) go through
``util.read_signal`` / numpy.
"""
import abc
import warnings
from typing import Callable, Optional, Union

import numpy as np

from . import _native
from .alias import AliasedFactory

__all__ = ["CMVN", "Deltas", "PostProcessor", "Stack", "Standardize"]


class PostProcessor(AliasedFactory):
    """A transform applied to a feature tensor (reference post.py:38-63)"""

    @abc.abstractmethod
    def apply(self, features: np.ndarray, axis: int = -1, in_place: bool = False) -> np.ndarray:
        pass


def _is_gpu_tensor(x) -> bool:
    return bool(getattr(x, "is_cuda", False))


def _to_device(array: np.ndarray):
    torch = _native.require_device()
    return torch.from_numpy(np.ascontiguousarray(array)).to("cuda")


def _stream(torch, tensor):
    return torch.cuda.current_stream(tensor.device).cuda_stream


_ROWS_META = {}  # device copies of the ragged-row descriptions used most recently (insertion-ordered)
_ROWS_META_ENTRIES = 64


def _rows_meta(row_offsets, device):
    """Device ``int64[2][B]`` (first row, row count) of a packed ragged batch + host row counts

    Repeated calls with the same `row_offsets` (a pipeline processing batch after batch of a few
    geometries) reuse the device copy instead of a host-to-device transfer per launch.
    """
    torch = _native.require_device()
    rows = np.ascontiguousarray(row_offsets, dtype=np.int64)
    key = (rows.tobytes(), str(device))
    hit = _ROWS_META.pop(key, None)
    if hit is None:
        nrows = np.diff(rows)
        hit = (torch.from_numpy(np.stack([rows[:-1], nrows])).to(device), nrows)
        while len(_ROWS_META) >= _ROWS_META_ENTRIES:
            _ROWS_META.pop(next(iter(_ROWS_META)))  # the least recently used
    _ROWS_META[key] = hit  # (re-)inserted as the most recent
    return hit


# ------------------------------------------------------------------ Standardize ------


class Standardize(PostProcessor):
    """Zero mean (and unit variance) per coefficient; always returns float64

    `axis` of :func:`apply` / :func:`accumulate` is the *coefficient* axis; statistics run
    over every other axis.  Without accumulated statistics the tensor's own are used
    (reference post.py:66-305).  Sums and the normalisation run on the GPU in float64.
    """

    aliases = {"standardize", "normalize", "unit", "cmvn"}

    def __init__(self, rfilename: Optional[str] = None, norm_var: bool = True, **kwargs):
        self._stats = None  # float64 [2, C + 1]: sums | count, sums of squares | unused
        self._norm_var = bool(norm_var)
        if rfilename is None:
            if kwargs:
                raise TypeError("Invalid keyword arguments: {}".format(tuple(kwargs)))
            return
        # Statistics written by save() or by Kaldi's compute-cmvn-stats (reference
        # post.py:104-122): read with util.read_signal; keyword arguments go to the reader.
        from .util import read_signal

        if "dtype" in kwargs:
            self._stats = read_signal(rfilename, **kwargs)
            return
        for guess in (np.float64, np.float32, "dm", "fm"):
            try:
                self._stats = read_signal(rfilename, dtype=guess, **kwargs)
                break
            except (IOError, ValueError, ImportError, TypeError):
                continue
        if self._stats is None:
            raise IOError("Unable to load stats from {}".format(rfilename))
        if self._stats.ndim == 1:
            self._stats = self._two_rows(self._stats)

    @staticmethod
    def _two_rows(flat):
        """Raw binary statistics -> ``[2, C + 1]`` float64 (reference post.py:124-152)

        A headerless file does not say whether it holds float64 or float32 words.  A valid
        table has an integral count in ``[0, -1]`` and no negative entry; if the first
        interpretation fails that test the same bytes are tried as the other float width.
        """

        def plausible(table):
            return bool(np.isclose(np.round(table[0, -1]), table[0, -1]) and np.all(table >= 0))

        if flat.dtype not in (np.float32, np.float64):
            raise ValueError(
                "Statistics were loaded with a weird data type ({}) and are invalid. Make sure "
                "the arguments you passed to the init are correct".format(flat.dtype)
            )
        other = np.float32 if flat.dtype == np.float64 else np.float64
        for words in (flat, np.frombuffer(flat.tobytes(), dtype=other).astype(np.float64)):
            if words.size and words.size % 2 == 0:
                table = words.reshape(2, -1)
                if plausible(table):
                    return table
        raise IOError(
            "Could not properly load statistics. Try specifying additional parameters in init "
            "(see docstring)"
        )

    def save(self, wfilename: str, key: Optional[str] = None, compress: bool = False,
             overwrite: bool = True) -> None:
        """Write the accumulated statistics (reference post.py:307-361)

        ``.npy``: :func:`numpy.save`.  ``.npz``: an archive entry named `key` (default: the first
        free ``arr_<n>``), next to the archive's existing entries when `overwrite` is true (the
        reference's flag reads that way round, post.py:349-353), compressed on request.  Any
        other name: raw float64 words (:meth:`numpy.ndarray.tofile`).
        """
        if not self.have_stats:
            raise ValueError("No stats have been accumulated to save")
        if wfilename.endswith(".npy"):
            np.save(wfilename, self._stats)
        elif wfilename.endswith(".npz"):
            entries = {}
            if overwrite:
                try:
                    with np.load(wfilename) as archive:
                        entries = {name: archive[name] for name in archive.files}
                except IOError:
                    pass
            if key is None:
                n = 0
                while "arr_{}".format(n) in entries:
                    n += 1
                key = "arr_{}".format(n)
            entries[key] = self._stats
            (np.savez_compressed if compress else np.savez)(wfilename, **entries)
        else:
            self._stats.tofile(wfilename)

    @property
    def have_stats(self) -> bool:
        return self._stats is not None and self._stats[0, -1]

    # -- device helpers -----------------------------------------------------------------

    @staticmethod
    def _as_device_3d(features, axis):
        # view as [outer, coeff, inner] on the GPU in float32 or float64
        torch = _native.require_device()
        if _is_gpu_tensor(features):
            t = features
            if t.dtype not in (torch.float32, torch.float64):
                t = t.to(torch.float64)
        else:
            arr = np.asarray(features)
            if arr.dtype not in (np.float32, np.float64):
                arr = arr.astype(np.float64)
            t = _to_device(arr)
        t = t.contiguous()
        shape = tuple(t.shape)
        axis = axis % len(shape)
        outer = int(np.prod(shape[:axis], dtype=np.int64))
        inner = int(np.prod(shape[axis + 1 :], dtype=np.int64))
        return t, outer, shape[axis], inner

    @staticmethod
    def _device_sums(t, outer, C, inner):
        torch = _native.require_device()
        lib = _native.lib()
        stats = torch.empty(2 * C, dtype=torch.float64, device=t.device)
        scratch = torch.empty(
            int(lib.pds_cmvn_scratch_len(C, inner)), dtype=torch.float64, device=t.device
        )
        fn = lib.pds_cmvn_stats_f32 if t.dtype == torch.float32 else lib.pds_cmvn_stats_f64
        with torch.cuda.device(t.device):
            rc = fn(t.data_ptr(), outer, C, inner, stats.data_ptr(), scratch.data_ptr(), _stream(torch, t))
        _native.check(rc, "pds_cmvn_stats")
        return stats.cpu().numpy().reshape(2, C)

    def _check_width(self, num_coeffs):
        if self._stats is not None and self._stats.shape[1] != num_coeffs + 1:
            raise ValueError(
                "Expected feature vector of length {}; got {}".format(
                    self._stats.shape[1] - 1, num_coeffs
                )
            )

    # -- public -------------------------------------------------------------------------

    def accumulate(self, features, axis: int = -1) -> None:
        """Add `features` to the global statistics (reference post.py:193-212)"""
        shape = tuple(features.shape)
        if (shape and not np.prod(shape)) or not len(features):
            raise ValueError("Cannot accumulate from empty array")
        if len(shape) <= 1:
            features, axis = features.reshape(1, -1), 1
        t, outer, C, inner = self._as_device_3d(features, axis)
        self._check_width(C)
        sums = self._device_sums(t, outer, C, inner)
        if self._stats is None:
            self._stats = np.zeros((2, C + 1), dtype=np.float64)
        self._stats[0, -1] += outer * inner
        self._stats[0, :-1] += sums[0]
        self._stats[1, :-1] += sums[1]

    def apply(self, features, axis: int = -1, in_place: bool = False):
        shape = tuple(features.shape)
        if (shape and not np.prod(shape)) or not len(features):
            raise ValueError("Cannot apply to empty array")
        torch = _native.require_device()
        on_gpu = _is_gpu_tensor(features)
        vector = len(shape) <= 1
        if vector:
            view, ax = features.reshape(1, -1), 1
        else:
            view, ax = features, axis
        t, outer, C, inner = self._as_device_3d(view, ax)
        self._check_width(C)
        count = outer * inner
        if self.have_stats:
            count = self._stats[0, -1]
            means = self._stats[0, :-1] / count
            varss = self._stats[1, :-1] / count - means ** 2
        elif count == 1:
            # a lone vector has no variance of its own (reference post.py:240-247, 268-277)
            if self._norm_var:
                raise ValueError(
                    "Unable to standardize the variance of a vector with no global statistics"
                )
            warnings.warn("Standardizing a single vector to 0")
            zeros = torch.zeros(shape, dtype=torch.float64, device=t.device)
            return zeros if on_gpu else zeros.cpu().numpy()
        else:
            sums = self._device_sums(t, outer, C, inner)
            means = sums[0] / count
            varss = sums[1] / count - means ** 2
        if self._norm_var:
            close_zero = np.isclose(varss, 0)
            if np.any(close_zero):
                warnings.warn("0 variance encountered. Replacing with 1")
                varss = np.where(close_zero, 1.0, varss)
            scales = 1 / (varss ** 0.5)
        else:
            scales = np.ones(C, dtype=np.float64)
        lib = _native.lib()
        d_scale = _to_device(np.asarray(scales, dtype=np.float64))
        d_shift = _to_device(np.asarray(means * scales, dtype=np.float64))
        if in_place and on_gpu and features.dtype == torch.float64 and features.is_contiguous():
            out = t
        else:
            out = torch.empty(t.shape, dtype=torch.float64, device=t.device)
        fn = lib.pds_cmvn_apply_f32 if t.dtype == torch.float32 else lib.pds_cmvn_apply_f64
        with torch.cuda.device(t.device):
            rc = fn(
                t.data_ptr(), outer, C, inner, d_scale.data_ptr(), d_shift.data_ptr(),
                out.data_ptr(), _stream(torch, t),
            )
        _native.check(rc, "pds_cmvn_apply")
        out = out.reshape(shape)
        if on_gpu:
            return out
        res = out.cpu().numpy()
        if in_place and isinstance(features, np.ndarray) and features.dtype == np.float64:
            features[...] = res
            return features
        return res

    def apply_rows(self, feats, row_offsets, out_dtype=None):
        """Per-utterance (local) CMVN over a packed ragged batch, entirely on the GPU

        `feats` is the ``(total_rows, C)`` GPU tensor :func:`compute_packed` returns and
        `row_offsets` its ``B + 1`` row offsets.  Returns a float64 tensor (or float32
        with ``out_dtype=torch.float32``) of the same shape.
        """
        torch = _native.require_device()
        lib = _native.lib()
        if self.have_stats:
            raise ValueError("apply_rows standardises locally; global statistics are set")
        if not _is_gpu_tensor(feats) or feats.dtype != torch.float32 or feats.dim() != 2:
            raise ValueError("feats must be a 2-D float32 GPU tensor")
        if feats.stride(1) != 1:
            feats = feats.contiguous()
        B, C = len(row_offsets) - 1, feats.shape[1]
        out_dtype = torch.float64 if out_dtype is None else out_dtype
        out = torch.empty(feats.shape, dtype=out_dtype, device=feats.device)
        if B <= 0 or feats.shape[0] == 0:
            return out
        meta, _ = _rows_meta(row_offsets, feats.device)
        stats = torch.empty((B, 2, C), dtype=torch.float64, device=feats.device)
        zero_var = torch.zeros(1, dtype=torch.int32, device=feats.device)
        fn = lib.pds_cmvn_rows_f32 if out_dtype == torch.float64 else lib.pds_cmvn_rows_f32out
        with torch.cuda.device(feats.device):
            for lo in range(0, B, 65535):
                hi = min(B, lo + 65535)
                rc = fn(
                    feats.data_ptr(), feats.stride(0), meta[0, lo:].data_ptr(),
                    meta[1, lo:].data_ptr(), hi - lo, C, int(self._norm_var),
                    stats[lo:].data_ptr(), out.data_ptr(), out.stride(0),
                    zero_var.data_ptr(), _stream(torch, feats),
                )
                _native.check(rc, "pds_cmvn_rows")
        self._last_zero_var = zero_var  # inspect with .item() (synchronises)
        return out


CMVN = Standardize


# ------------------------------------------------------------------ Deltas -----------


class Deltas(PostProcessor):
    """Append (or stack) delta features: repeated correlation with a ramp filter

    `axis` of :func:`apply` is the *time* axis.  The order-k filter is the k-fold
    convolution of ``arange(-W, W + 1) / sum(j^2)``; slices are padded (default
    ``"edge"``), correlated in float64 and cast back to the input dtype (reference
    post.py:367-491).
    """

    aliases = {"deltas"}

    def __init__(
        self,
        num_deltas: int,
        target_axis: int = -1,
        concatenate: bool = True,
        context_window: int = 2,
        pad_mode: Union[str, Callable] = "edge",
        **kwargs,
    ):
        self._target_axis = target_axis
        self._pad_mode = pad_mode
        self._pad_kwargs = kwargs
        self.concatenate = bool(concatenate)
        self.num_deltas = num_deltas
        ramp = np.arange(1 + 2 * context_window, dtype=np.float64)
        ramp -= context_window
        ramp /= np.sum(ramp ** 2)
        self._filts = [np.ones(1, dtype=np.float64)]
        for idx in range(num_deltas):
            self._filts.append(np.convolve(self._filts[idx], ramp))
        self._device_filts = {}

    def _filters_on(self, device):
        key = str(device)
        if key not in self._device_filts:
            torch = _native.require_device()
            lens = [len(f) for f in self._filts[1:]]
            offs = np.zeros(len(lens) + 1, dtype=np.int32)
            np.cumsum(lens, out=offs[1:])
            flat = np.concatenate(self._filts[1:]) if lens else np.zeros(1)
            self._device_filts[key] = (
                torch.from_numpy(flat).to(device),
                torch.from_numpy(offs).to(device),
            )
        return self._device_filts[key]

    def _apply_per_order(self, features, host, axis, target, on_gpu, in_dtype):
        """:func:`apply` for pad modes whose samples are computed or depend on the pad width"""
        torch = _native.require_device()
        lib = _native.lib()
        host = np.asarray(host, dtype=np.float64)
        shape = host.shape
        ndim = host.ndim
        time = shape[axis]
        outer = int(np.prod(shape[:axis], dtype=np.int64))
        inner = int(np.prod(shape[axis + 1 :], dtype=np.int64))
        statics = features if on_gpu else _to_device(np.asarray(features))
        res_dtype = statics.dtype if on_gpu else None
        pieces = [statics]
        for filt in self._filts[1:]:
            reach = (len(filt) - 1) // 2
            widths = [(0, 0)] * ndim
            widths[axis] = (reach, reach)
            src = _to_device(np.pad(host, widths, self._pad_mode, **self._pad_kwargs))
            d_filt = torch.from_numpy(np.ascontiguousarray(filt)).to(src.device)
            d_offs = torch.tensor([0, len(filt)], dtype=torch.int32, device=src.device)
            out = torch.empty((2,) + shape, dtype=torch.float64, device=src.device)
            with torch.cuda.device(src.device):
                rc = lib.pds_deltas_f64(
                    src.data_ptr(), outer, time, inner, d_filt.data_ptr(), d_offs.data_ptr(), 1, 0, reach,
                    out.data_ptr(), outer * time * inner, time * inner, inner, 1, _stream(torch, src),
                )
            _native.check(rc, "pds_deltas")
            pieces.append(out[1])
        if on_gpu:
            pieces = [pieces[0]] + [p.to(res_dtype) for p in pieces[1:]]
            return torch.cat(pieces, target) if self.concatenate else torch.stack(pieces, target)
        # the reference casts every order back to the input dtype before concatenating
        cast = [np.asarray(features)] + [p.cpu().numpy().astype(in_dtype) for p in pieces[1:]]
        return np.concatenate(cast, target) if self.concatenate else np.stack(cast, target)

    def apply(self, features, axis: int = -1, in_place: bool = False):
        torch = _native.require_device()
        lib = _native.lib()
        on_gpu = _is_gpu_tensor(features)
        K = self.num_deltas
        if on_gpu:
            t = features
            in_dtype = None
            if t.dtype not in (torch.float32, torch.float64):
                raise TypeError("GPU features must be float32 or float64")
        else:
            arr = np.asarray(features)
            in_dtype = arr.dtype
            work = arr if arr.dtype in (np.float32, np.float64) else arr.astype(np.float64)
            t = None
        shape = tuple(features.shape)
        ndim = len(shape)
        axis = axis % ndim
        target = self._target_axis
        out_ndim = ndim if self.concatenate else ndim + 1
        if not -out_ndim <= target < out_ndim:
            raise np.exceptions.AxisError(target, out_ndim)
        target %= out_ndim
        time = shape[axis]
        outer = int(np.prod(shape[:axis], dtype=np.int64))
        inner = int(np.prod(shape[axis + 1 :], dtype=np.int64))
        stacked_shape = (K + 1,) + shape
        if outer * time * inner == 0 or K == 0:
            # nothing to correlate; only the layout changes
            pieces = [features] * (K + 1)
            if on_gpu:
                return torch.cat(pieces, target) if self.concatenate else torch.stack(pieces, target)
            return np.concatenate(pieces, target) if self.concatenate else np.stack(pieces, target)
        edge = self._pad_mode == "edge" and not self._pad_kwargs
        max_off = (len(self._filts[-1]) - 1) // 2
        # modes that COPY samples from positions fixed relative to the signal's ends: padding once
        # by the widest filter's reach gives every order the samples the reference's own, narrower
        # padding would (post.py:470-483)
        copies = (isinstance(self._pad_mode, str) and self._pad_mode in ("reflect", "symmetric", "wrap")
                  and not self._pad_kwargs)
        if edge:
            if t is None:
                t = _to_device(work)
            src = t.contiguous()
        elif copies:
            host = work if t is None else t.cpu().numpy()
            widths = [(0, 0)] * ndim
            widths[axis] = (max_off, max_off)
            src = _to_device(np.pad(host, widths, self._pad_mode, **self._pad_kwargs))
        else:
            # every other numpy.pad mode (linear_ramp, statistics, constants, odd reflection,
            # callables): the padded samples are computed, in float64, and may depend on the pad
            # width -- pad each order by its own reach on a float64 copy, one launch per order, as
            # the reference does
            return self._apply_per_order(features, work if t is None else t.cpu().numpy(), axis, target,
                                         on_gpu, in_dtype)
        d_filts, d_offs = self._filters_on(src.device)
        direct = self.concatenate and ndim == 2 and target != axis
        if direct:
            # (time, coeff) or (coeff, time) matrix with deltas appended to the other axis:
            # the kernel writes the concatenated layout itself
            out_shape = list(shape)
            out_shape[target] *= K + 1
            out = torch.empty(out_shape, dtype=src.dtype, device=src.device)
            if axis == 0:  # [1, time, inner=F] -> (time, (K+1) F)
                sk, so, st, si = shape[1], 0, shape[1] * (K + 1), 1
            else:  # [outer=F, time, 1] -> ((K+1) F, time)
                sk, so, st, si = shape[0] * time, time, 1, 0
        else:
            out = torch.empty(stacked_shape, dtype=src.dtype, device=src.device)
            sk, so, st, si = outer * time * inner, time * inner, inner, 1
        fn = lib.pds_deltas_f32 if src.dtype == torch.float32 else lib.pds_deltas_f64
        with torch.cuda.device(src.device):
            rc = fn(
                src.data_ptr(), outer, time, inner, d_filts.data_ptr(), d_offs.data_ptr(), K,
                int(edge), max_off, out.data_ptr(), sk, so, st, si, _stream(torch, src),
            )
        _native.check(rc, "pds_deltas")
        if not direct:
            pieces = list(out.unbind(0))
            out = torch.cat(pieces, target) if self.concatenate else torch.stack(pieces, target)
        if on_gpu:
            return out
        res = out.cpu().numpy()
        return res if res.dtype == in_dtype else res.astype(in_dtype)

    def apply_rows(self, feats, row_offsets, out=None):
        """Deltas of every utterance of a packed ragged batch, appended per row, on the GPU

        `feats`: ``(total_rows, F)`` float32 GPU tensor (rows may be strided), utterance
        b in rows ``row_offsets[b]:row_offsets[b+1]``.  Returns ``(total_rows, (K+1) F)``.
        Time is the row axis (Kaldi layout); padding is ``"edge"``.
        """
        torch = _native.require_device()
        lib = _native.lib()
        if self._pad_mode != "edge" or self._pad_kwargs:
            raise ValueError("apply_rows supports the default 'edge' padding only")
        if not _is_gpu_tensor(feats) or feats.dtype != torch.float32 or feats.dim() != 2:
            raise ValueError("feats must be a 2-D float32 GPU tensor")
        if feats.stride(1) != 1:
            feats = feats.contiguous()
        # `feats` may be a column slice of `out` (statics already in place): the kernel then only
        # adds the delta columns, in place
        B, F, K = len(row_offsets) - 1, feats.shape[1], self.num_deltas
        if out is None:
            out = torch.empty((feats.shape[0], (K + 1) * F), dtype=torch.float32, device=feats.device)
        if B <= 0 or feats.shape[0] == 0:
            return out
        meta, nrows = _rows_meta(row_offsets, feats.device)
        d_filts, d_offs = self._filters_on(feats.device)
        with torch.cuda.device(feats.device):
            for lo in range(0, B, 65535):
                hi = min(B, lo + 65535)
                rc = lib.pds_deltas_rows_f32(
                    feats.data_ptr(), feats.stride(0), meta[0, lo:].data_ptr(),
                    meta[1, lo:].data_ptr(), hi - lo, int(nrows[lo:hi].max()), F,
                    d_filts.data_ptr(), d_offs.data_ptr(), K, (len(self._filts[-1]) - 1) // 2,
                    out.data_ptr(), out.stride(0),
                    _stream(torch, feats),
                )
                _native.check(rc, "pds_deltas_rows")
        return out


# ------------------------------------------------------------------ Stack ------------


class Stack(PostProcessor):
    """Concatenate every `num_vectors` consecutive frames into one (reference post.py:494-563)

    `time_axis` is the axis frames are drawn from; `axis` of :func:`apply` the coefficient axis
    that grows `num_vectors`-fold.  Frames that do not fill a last group are dropped, or, with
    `pad_mode` (and keyword arguments of :func:`numpy.pad`), the time axis is first padded on
    the right up to a multiple of `num_vectors`.

    Stacking is data movement, not arithmetic: a host array is re-viewed / re-gathered with
    numpy and a GPU tensor with torch indexing, exactly as the reference does it, and a packed
    ragged batch goes through one HIP copy kernel (:func:`apply_rows`).
    """

    aliases = {"stack"}

    def __init__(self, num_vectors: int, time_axis: int = 0,
                 pad_mode: Optional[Union[str, Callable]] = None, **kwargs):
        if num_vectors < 1:
            raise ValueError(f"Expected num_vectors to be positive, got {num_vectors}")
        self.num_vectors = num_vectors
        self.time_axis = time_axis
        self._pad_mode = pad_mode
        self._pad_kwargs = kwargs

    def apply(self, features, axis: int = -1, in_place: bool = False):
        nd = features.ndim if not _is_gpu_tensor(features) else features.dim()
        axis, time_axis = axis % nd, self.time_axis % nd
        if axis == time_axis:
            raise RuntimeError(f"feature and time axes are the same ({axis})")
        nv = self.num_vectors
        on_gpu = _is_gpu_tensor(features)
        T = features.shape[time_axis]
        if self._pad_mode is not None and T % nv:
            extra = nv - T % nv
            if on_gpu:
                features = self._pad_tensor(features, time_axis, extra)
            else:
                widths = [(0, 0)] * nd
                widths[time_axis] = (0, extra)
                features = np.pad(features, widths, self._pad_mode, **self._pad_kwargs)
            in_place = True  # already a private copy
            T += extra
        groups = T // nv
        if nd == 2:
            # the (T, F) matrix re-read as (T / nv, nv F): a reshape of the time-major layout
            if not in_place:
                features = features.clone() if on_gpu else features.copy()
            mat = features.T if time_axis else features
            mat = mat[: groups * nv].reshape(groups, mat.shape[1] * nv)
            return mat.T if time_axis else mat
        # general rank: frame v of every group, for v = 0 .. nv - 1, side by side on `axis`
        index = [slice(None)] * nd
        parts = []
        for v in range(nv):
            index[time_axis] = slice(v, groups * nv, nv)
            parts.append(features[tuple(index)])
        if on_gpu:
            return _native.require_device().cat(parts, axis)
        return np.concatenate(parts, axis)

    def _pad_tensor(self, t, time_axis, extra):
        torch = _native.require_device()
        if self._pad_mode == "edge" and not self._pad_kwargs:
            last = t.narrow(time_axis, t.shape[time_axis] - 1, 1)
            reps = [1] * t.dim()
            reps[time_axis] = extra
            return torch.cat([t, last.repeat(*reps)], time_axis)
        if self._pad_mode == "constant" and set(self._pad_kwargs) <= {"constant_values"}:
            shape = list(t.shape)
            shape[time_axis] = extra
            fill = self._pad_kwargs.get("constant_values", 0)
            return torch.cat([t, torch.full(shape, fill, dtype=t.dtype, device=t.device)], time_axis)
        raise ValueError("GPU tensors support pad_mode 'edge' and 'constant' only")

    def apply_rows(self, feats, row_offsets):
        """Stack every utterance of a packed ragged batch on the GPU

        `feats`: ``(total_rows, F)`` float32 GPU tensor, utterance b in rows
        ``row_offsets[b]:row_offsets[b+1]`` (time is the row axis).  Returns
        ``(stacked, new_row_offsets)`` with ``stacked`` of shape ``(new_total, num_vectors F)``.
        """
        torch = _native.require_device()
        lib = _native.lib()
        if not _is_gpu_tensor(feats) or feats.dtype != torch.float32 or feats.dim() != 2:
            raise ValueError("feats must be a 2-D float32 GPU tensor")
        if self._pad_mode is None:
            pad = 0
        elif self._pad_mode == "constant" and not self._pad_kwargs:
            pad = 1
        elif self._pad_mode == "edge" and not self._pad_kwargs:
            pad = 2
        else:
            raise ValueError("apply_rows supports pad_mode None, 'constant' (zeros) and 'edge'")
        if feats.stride(1) != 1:
            feats = feats.contiguous()
        nv, F = self.num_vectors, feats.shape[1]
        rows = np.ascontiguousarray(row_offsets, dtype=np.int64)
        nrows = np.diff(rows)
        out_rows = (nrows + nv - 1) // nv if pad else nrows // nv
        new_offsets = np.concatenate([[0], np.cumsum(out_rows)]).astype(np.int64)
        out = torch.empty((int(new_offsets[-1]), nv * F), dtype=torch.float32, device=feats.device)
        B = len(nrows)
        if B == 0 or out.shape[0] == 0:
            return out, new_offsets
        meta = torch.from_numpy(np.stack([rows[:-1], nrows, new_offsets[:-1]])).to(feats.device)
        with torch.cuda.device(feats.device):
            for lo in range(0, B, 65535):
                hi = min(B, lo + 65535)
                rc = lib.pds_stack_rows_f32(
                    feats.data_ptr(), feats.stride(0), meta[0, lo:].data_ptr(), meta[1, lo:].data_ptr(),
                    meta[2, lo:].data_ptr(), hi - lo, int(out_rows[lo:hi].max()), F, nv, pad,
                    out.data_ptr(), out.stride(0), _stream(torch, feats),
                )
                _native.check(rc, "pds_stack_rows")
        return out, new_offsets
