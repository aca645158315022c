"""ctypes binding of ``libpds_amd.so`` (C ABI: ``include/pds_amd.h``).

There is deliberately no fallback: if the shared library is missing or a compute call
is made without a HIP device, an exception is raised.  Nothing in this package computes
features on the CPU.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_int32, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# PDS_AMD_LIB selects another build of the same library (kernel tuning experiments)
LIB_PATH = os.environ.get("PDS_AMD_LIB") or os.path.join(_HERE, "csrc", "libpds_amd.so")

PDS_OK = 0


class NativeError(RuntimeError):
    """A call into libpds_amd.so failed"""


class StftDesc(Structure):
    # mirrors pds_stft_desc (include/pds_amd.h)
    _fields_ = [
        ("frame_length", c_int32),
        ("frame_shift", c_int32),
        ("dft_size", c_int32),
        ("pad_left", c_int32),
        ("num_filts", c_int32),
        ("nnz", c_int32),
        ("use_power", c_int32),
        ("use_log", c_int32),
        ("include_energy", c_int32),
        ("reserved", c_int32),
        ("log_floor", c_double),
    ]


class SiDesc(Structure):
    # mirrors pds_si_desc (include/pds_amd.h)
    _fields_ = [
        ("frame_shift", c_int32),
        ("max_support", c_int32),
        ("num_coeffs", c_int32),
        ("taps_complex", c_int32),
        ("use_power", c_int32),
        ("use_log", c_int32),
        ("reserved", c_int32),
        ("reserved2", c_int32),
        ("log_floor", c_double),
    ]


_SI_BATCH_ARGS = [
    c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_int64,
    c_void_p, c_int64, c_void_p,
]

_BATCH_ARGS = [
    c_void_p,  # plan
    c_void_p,  # d_signal
    c_void_p,  # d_offsets
    c_void_p,  # d_lengths
    c_void_p,  # d_nframes
    c_void_p,  # d_row_off
    c_int32,  # B
    c_int64,  # max_frames
    c_int32,  # pad_left
    c_double,  # preemph
    c_void_p,  # d_out
    c_int64,  # out_stride
    c_void_p,  # stream
]

_DELTAS_ARGS = [
    c_void_p, c_int64, c_int64, c_int64,  # d_in, outer, time, inner
    c_void_p, c_void_p, c_int32,  # d_filts, d_filt_off, K
    c_int32, c_int32,  # edge_clamp, max_off
    c_void_p, c_int64, c_int64, c_int64, c_int64,  # d_out, sk, so, st, si
    c_void_p,  # stream
]

_CMVN_ROWS_ARGS = [
    c_void_p, c_int64, c_void_p, c_void_p, c_int32, c_int32, c_int32,
    c_void_p, c_void_p, c_int64, c_void_p, c_void_p,
]

# name -> (restype, argtypes); every symbol include/pds_amd.h declares
SIGNATURES = {
    "pds_version": (c_int32, []),
    "pds_build_experiments": (c_int32, []),
    "pds_last_error": (c_char_p, []),
    "pds_device_count": (c_int32, []),
    "pds_stft_plan_create": (
        c_int32,
        [POINTER(StftDesc), c_void_p, c_void_p, c_void_p, c_void_p, POINTER(c_void_p)],
    ),
    "pds_stft_plan_destroy": (None, [c_void_p]),
    "pds_stft_num_coeffs": (c_int32, [c_void_p]),
    "pds_stft_num_frames": (c_int64, [c_void_p, c_int64]),
    "pds_stft_plan_kernel_kind": (c_int32, [c_void_p]),
    "pds_stft_batch_f32": (c_int32, _BATCH_ARGS),
    "pds_stft_batch_f64": (c_int32, _BATCH_ARGS),
    "pds_stft_batch_f32_generic": (c_int32, _BATCH_ARGS),
    "pds_stft_plan_has_f64in": (c_int32, [c_void_p]),
    "pds_stft_plan_has_fused_deltas": (c_int32, [c_void_p]),
    "pds_stft_plan_has_fused_cmvn": (c_int32, [c_void_p]),
    "pds_stft_cmvn_partials_len": (c_int64, [c_void_p, c_int32]),
    "pds_stft_prepare_chunk_prefix": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p]),
    "pds_stft_cmvn_batch_f32": (
        c_int32,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_int32, c_int32, c_void_p,
         c_int32, c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int32, c_int64, c_void_p, c_void_p],
    ),
    "pds_stft_deltas_batch": (
        c_int32,
        [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_int32, c_double,
         c_int32, c_int32, c_void_p, c_void_p, c_int32, c_void_p, c_int64, c_void_p],
    ),
    "pds_stft_deltas_batch_f32": (
        c_int32,
        [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_int32,
         c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_int64, c_void_p],
    ),
    "pds_stft_batch_f64in": (c_int32, _BATCH_ARGS[:11] + [c_int32] + _BATCH_ARGS[11:]),
    "pds_stft_plan_has_i16in": (c_int32, [c_void_p]),
    "pds_stft_batch_i16in": (c_int32, _BATCH_ARGS),
    "pds_stft_batch_ragged_f32": (c_int32, _BATCH_ARGS[:10] + [c_void_p] + _BATCH_ARGS[10:]),
    "pds_stft_batch_ragged_i16in": (c_int32, _BATCH_ARGS[:10] + [c_void_p] + _BATCH_ARGS[10:]),
    "pds_preemphasize_f32": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_double, c_void_p, c_void_p]),
    "pds_preemphasize_f64": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_double, c_void_p, c_void_p]),
    "pds_dither_f32": (c_int32, [c_void_p, c_int64, c_double, ctypes.c_uint64, c_void_p, c_void_p]),
    "pds_dither_f64": (c_int32, [c_void_p, c_int64, c_double, ctypes.c_uint64, c_void_p, c_void_p]),
    "pds_deltas_f32": (c_int32, _DELTAS_ARGS),
    "pds_deltas_f64": (c_int32, _DELTAS_ARGS),
    "pds_deltas_rows_f32": (
        c_int32,
        [c_void_p, c_int64, c_void_p, c_void_p, c_int32, c_int64, c_int32, c_void_p,
         c_void_p, c_int32, c_int32, c_void_p, c_int64, c_void_p],
    ),
    "pds_stack_rows_f32": (
        c_int32,
        [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_int32, c_int32,
         c_int32, c_void_p, c_int64, c_void_p],
    ),
    "pds_si_plan_create": (c_int32, [POINTER(SiDesc), c_void_p, c_void_p, POINTER(c_void_p)]),
    "pds_si_plan_destroy": (None, [c_void_p]),
    "pds_si_scratch_len": (c_int64, [c_void_p, c_int32, c_int64]),
    "pds_si_plan_fft_size": (c_int32, [c_void_p]),
    "pds_si_batch_f32": (c_int32, _SI_BATCH_ARGS[:9] + [c_void_p] + _SI_BATCH_ARGS[9:]),
    "pds_si_batch_f64": (c_int32, _SI_BATCH_ARGS),
    "pds_cmvn_scratch_len": (c_int64, [c_int64, c_int64]),
    "pds_cmvn_stats_f32": (c_int32, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p]),
    "pds_cmvn_stats_f64": (c_int32, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p]),
    "pds_cmvn_apply_f32": (c_int32, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pds_cmvn_apply_f64": (c_int32, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pds_cmvn_rows_f32": (c_int32, _CMVN_ROWS_ARGS),
    "pds_cmvn_rows_f32out": (c_int32, _CMVN_ROWS_ARGS),
    # host feed (pinned staging ring: upload / kernel / download of consecutive batches overlap)
    "pds_feed_create": (c_int32, [c_void_p, c_int32, c_int64, c_int32, c_int32, c_int32, POINTER(c_void_p)]),
    "pds_feed_destroy": (None, [c_void_p]),
    "pds_feed_slot_rows": (c_int64, [c_void_p]),
    "pds_feed_set_direct": (c_int32, [c_void_p, c_int32]),
    "pds_feed_acquire": (c_int32, [c_void_p, POINTER(c_int32), POINTER(c_void_p)]),
    "pds_feed_pack": (c_int32, [c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_int32]),
    "pds_feed_submit": (c_int32, [c_void_p, c_int32, c_void_p, c_int32, c_double, c_int32]),
    "pds_feed_submit_frames": (c_int32, [c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_int32, c_double, c_int32]),
    "pds_feed_device_view": (c_int32, [c_void_p, c_int32, POINTER(c_void_p), POINTER(c_int64), POINTER(c_void_p),
                                       POINTER(c_void_p)]),
    "pds_feed_download": (c_int32, [c_void_p, c_int32, c_void_p, c_int64]),
    "pds_feed_collect": (c_int32, [c_void_p, c_int32, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_int64)]),
    "pds_feed_unpack": (c_int32, [c_void_p, c_int32, c_void_p, c_int64, c_int32]),
    "pds_feed_release": (c_int32, [c_void_p, c_int32]),
    # multi-GPU gather over RCCL
    "pds_comm_unique_id": (c_int32, [c_void_p]),
    "pds_comm_init_rank": (c_int32, [c_void_p, c_int32, c_int32, POINTER(c_void_p)]),
    "pds_comm_init_all": (c_int32, [c_int32, c_void_p, POINTER(c_void_p)]),
    "pds_comm_world": (c_int32, [c_void_p]),
    "pds_comm_rank": (c_int32, [c_void_p]),
    "pds_comm_destroy": (None, [c_void_p]),
    "pds_comm_group_start": (c_int32, []),
    "pds_comm_group_end": (c_int32, []),
    "pds_gather_rows": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "pds_allreduce_sum_f64": (c_int32, [c_void_p, c_void_p, c_int64, c_void_p]),
}

_lib = None


def lib() -> ctypes.CDLL:
    """The loaded library; raises :class:`NativeError` if it has not been built"""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` or `make -C pydrobert-speech_amd/csrc` (there is no CPU fallback)"
            )
        # torch's HIP runtime first where torch is installed: the library's own dependency on libamdhip64
        # then resolves to the copy already in the process (loaded the other way round, the two runtimes
        # disagree about the devices)
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        loaded = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(loaded, name)  # AttributeError if the .so is stale
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = loaded
    return _lib


def last_error() -> str:
    return lib().pds_last_error().decode("utf-8", "replace")


def check(rc: int, what: str) -> None:
    if rc != PDS_OK:
        msg = last_error()
        if rc == -1:
            raise ValueError(f"{what}: {msg}")
        if rc == -3:
            raise MemoryError(f"{what}: {msg}")
        raise NativeError(f"{what}: {msg} (code {rc})")


def require_device():
    """torch with a visible HIP device, or an exception -- never a CPU path"""
    import torch

    if not torch.cuda.is_available():
        raise NativeError(
            "no HIP device is visible: this package computes features on an MI355X only "
            "(no CPU fallback); the CPU restatement in oracle/ is test infrastructure"
        )
    return torch
