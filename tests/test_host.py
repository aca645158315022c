"""Host logic of the package on CPU: the alias/config surface, the init-time tables
against reference-derived goldens, and the C-ABI library's exports.  No compute calls."""
import ctypes
import json
import os
import re

import numpy as np
import pytest

import pydrobert_speech_amd as ps
from pydrobert_speech_amd import _native
from pydrobert_speech_amd.alias import AliasedFactory, alias_factory_subclass_from_arg
from pydrobert_speech_amd.compute import (
    FrameComputer, STFTFrameComputer, bin_weight_table, fold_spectrum_index,
)
from pydrobert_speech_amd.filters import LinearFilterBank, WindowFunction
from pydrobert_speech_amd.post import CMVN, Deltas, PostProcessor, Standardize
from pydrobert_speech_amd.scales import ScalingFunction
from oracle import stft_oracle as orc
from tests.conftest import ROOT, config_names, oracle_params

CONFIGS = config_names()


def build(cfg):
    return alias_factory_subclass_from_arg(FrameComputer, json.loads(json.dumps(cfg)))


# ---- alias registry (reference alias.py:34-100) ------------------------------------------


def test_alias_lookup_and_config_forms():
    assert type(alias_factory_subclass_from_arg(ScalingFunction, "mel")).__name__ == "MelScaling"
    lin = alias_factory_subclass_from_arg(ScalingFunction, {"alias": "uniform", "low_hz": 5.0})
    assert lin.low_hz == 5.0 and lin.slope_hz == 1.0
    lin2 = alias_factory_subclass_from_arg(ScalingFunction, {"name": "linear", "low_hz": 1, "slope_hz": 2})
    assert lin2.scale_to_hertz(lin2.hertz_to_scale(100.0)) == pytest.approx(100.0)
    assert alias_factory_subclass_from_arg(ScalingFunction, lin) is lin
    # 'tri' is a bank under LinearFilterBank and a window under WindowFunction
    assert type(alias_factory_subclass_from_arg(WindowFunction, "tri")).__name__ == "BartlettWindow"
    bank = alias_factory_subclass_from_arg(LinearFilterBank, {"name": "tri", "scaling_function": "mel"})
    assert type(bank).__name__ == "TriangularOverlappingFilterBank"
    assert isinstance(alias_factory_subclass_from_arg(PostProcessor, "cmvn"), Standardize)
    assert isinstance(alias_factory_subclass_from_arg(PostProcessor, {"name": "deltas", "num_deltas": 2}), Deltas)
    assert CMVN is Standardize
    with pytest.raises(ValueError, match="Cannot find subclass with alias 'nope'"):
        ScalingFunction.from_alias("nope")
    with pytest.raises(KeyError):
        alias_factory_subclass_from_arg(ScalingFunction, {"low_hz": 3})


def test_alias_last_defined_subclass_wins():
    class Root(AliasedFactory):
        pass

    class A(Root):
        aliases = {"x"}

    class ChildOfA(A):
        aliases = {"x", "child"}

    class B(Root):
        aliases = {"x"}

    # most recently defined direct child first, descendants before the class itself
    assert type(Root.from_alias("x")) is B
    del B
    import gc

    gc.collect()
    assert type(Root.from_alias("x")) is ChildOfA
    assert type(Root.from_alias("child")) is ChildOfA
    assert type(A.from_alias("x")) is ChildOfA


def test_stft_alias_builds_our_computer():
    comp = build({"name": "stft", "bank": "fbank"})
    assert isinstance(comp, STFTFrameComputer)
    assert comp.frame_style == "centered" and comp.frame_shift == 160 and comp.num_coeffs == 40
    assert comp.frame_shift_ms == 10 and not comp.started and not comp.includes_energy
    with pytest.raises(ValueError, match="Invalid frame style"):
        STFTFrameComputer("fbank", frame_style="sideways")
    with pytest.raises(ValueError, match="Invalid frequency range"):
        build({"name": "stft", "bank": {"name": "fbank", "high_hz": 9000}})


# ---- init-time tables against the reference ----------------------------------------------


@pytest.mark.parametrize("name", CONFIGS)
def test_tables_match_reference(name, golden_meta, golden_tables):
    comp = build(golden_meta["configs"][name])
    dims = golden_tables[f"{name}/dims"]
    got = [comp.frame_length, comp.frame_shift, comp.dft_size, comp.num_coeffs,
           int(comp.bank.is_real), int(comp.frame_style == "centered"), int(comp.kaldi_shift),
           int(comp.includes_energy), int(bool(comp._power)), int(bool(comp._log))]
    assert got == [int(v) for v in dims]
    assert np.allclose(comp._window, golden_tables[f"{name}/window"], rtol=1e-13, atol=1e-300)
    assert [int(s) for s in comp._filt_start_idxs] == golden_tables[f"{name}/starts"].tolist()
    offs = golden_tables[f"{name}/tap_offsets"]
    taps = golden_tables[f"{name}/taps"]
    assert [len(t) for t in comp._truncated_filts] == np.diff(offs).tolist()
    for f, mine in enumerate(comp._truncated_filts):
        ref = taps[offs[f] : offs[f + 1]]
        assert mine.dtype == ref.dtype
        assert np.allclose(mine, ref, rtol=1e-12, atol=1e-15), (name, f)
    assert np.allclose(np.asarray(comp.bank.supports, float), golden_tables[f"{name}/supports"])
    assert np.allclose(np.asarray(comp.bank.supports_hz, float), golden_tables[f"{name}/supports_hz"], rtol=1e-12)
    p = oracle_params(golden_tables, name)
    assert comp.pad_left == p.pad_left
    for n in golden_meta["lengths"][name]:
        assert comp.num_frames(n) == p.num_frames(n)


@pytest.mark.parametrize("name", CONFIGS)
def test_bin_weight_table_equals_reference_walk(name, golden_meta, golden_tables):
    # closed-form fold (product) vs the reference's segment walk (oracle)
    comp = build(golden_meta["configs"][name])
    p = oracle_params(golden_tables, name)
    W = orc.weights_dense(p)
    row_ptr, col, val = comp.bin_weights
    dense = np.zeros_like(W)
    for f in range(len(row_ptr) - 1):
        dense[f, col[row_ptr[f] : row_ptr[f + 1]]] = val[row_ptr[f] : row_ptr[f + 1]]
    assert np.allclose(dense, W, rtol=1e-12, atol=1e-300)
    assert (np.diff(row_ptr) >= 0).all() and col.min() >= 0 and col.max() <= comp.dft_size // 2


@pytest.mark.parametrize("N", [1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 400, 512, 513])
def test_fold_matches_walk_for_any_size(N):
    for start in sorted({0, 1, N // 3, N // 2, max(0, N - 1)}):
        taps = 3 * N + 2
        assert fold_spectrum_index(start + np.arange(taps), N).tolist() == orc.walk_bins(start, taps, N).tolist()


def test_kaldi_filters_known_answer(golden_kaldi):
    # the reference's tests/test_filters.py:211-223 on OUR bank
    with open(os.path.join(ROOT, "tests", "golden", "configs.json")) as fh:
        cfg = json.load(fh)["configs"]["c1_kaldi_fbank"]["bank"]
    bank = alias_factory_subclass_from_arg(LinearFilterBank, cfg)
    offs = np.concatenate([[0], np.cumsum(golden_kaldi["filt_lens"])])
    for f in range(40):
        off, filt = bank.get_truncated_response(f, 2 ** 9)
        filt = filt ** 2
        kaldi = golden_kaldi["filt_vals"][offs[f] : offs[f + 1]]
        assert off == golden_kaldi["filt_offsets"][f]
        assert np.allclose(filt[: len(kaldi)], kaldi, atol=1e-5)
        assert np.allclose(filt[len(kaldi) :], 0.0)


BANKS = {
    "triangular_analytic": lambda n: ps.filters.TriangularOverlappingFilterBank("mel", low_hz=5, num_filts=n, sampling_rate=8000, analytic=True),
    "triangular": lambda n: ps.filters.TriangularOverlappingFilterBank("mel", low_hz=0, num_filts=n, sampling_rate=8000),
    "fbank_analytic": lambda n: ps.filters.Fbank(low_hz=0, num_filts=n, sampling_rate=8000, analytic=True),
    "fbank": lambda n: ps.filters.Fbank(low_hz=0, num_filts=n, sampling_rate=8000),
    "gabor_erb": lambda n: ps.filters.GaborFilterBank("mel", low_hz=0, num_filts=n, sampling_rate=8000, erb=True),
    "gabor": lambda n: ps.filters.GaborFilterBank("mel", low_hz=0, num_filts=n, sampling_rate=8000),
    "gammatone_erb": lambda n: ps.filters.ComplexGammatoneFilterBank("mel", low_hz=0, num_filts=n, sampling_rate=8000, max_centered=True, erb=True),
    "gammatone": lambda n: ps.filters.ComplexGammatoneFilterBank("mel", low_hz=0, num_filts=n, sampling_rate=8000, max_centered=True),
}


@pytest.mark.parametrize("num_filts", [1, 11])
@pytest.mark.parametrize("kind", sorted(BANKS))
def test_truncated_response_rebuilds_full_response(kind, num_filts):
    # property pinned by the reference's tests/test_filters.py:73-111
    bank = BANKS[kind](num_filts)
    eps = ps.config.EFFECTIVE_SUPPORT_THRESHOLD
    for f in range(bank.num_filts):
        lo_hz, hi_hz = bank.supports_hz[f]
        lo, hi = bank.supports[f]
        N = int(max(hi - lo, 2 * bank.sampling_rate / (hi_hz - lo_hz), 1))
        full = bank.get_frequency_response(f, N)
        start, trunc = bank.get_truncated_response(f, N)
        rebuilt = np.zeros(N, dtype=trunc.dtype)
        wrap = min(start + len(trunc), N) - start
        rebuilt[start : start + wrap] = trunc[:wrap]
        rebuilt[: len(trunc) - wrap] = trunc[wrap:]
        if bank.is_real:
            rebuilt[N - start - len(trunc) + 1 : N - start + 1] = trunc[: None if start else 0 : -1].conj()
        assert np.allclose(full, rebuilt, atol=eps), (kind, f)
        half = bank.get_frequency_response(f, N, half=True)
        assert np.allclose(full[: len(half)], half)


@pytest.mark.parametrize("kind", sorted(BANKS))
def test_frequency_matches_impulse(kind):
    # reference tests/test_filters.py:114-137
    bank = BANKS[kind](11)
    for f in range(bank.num_filts):
        lo_hz, hi_hz = bank.supports_hz[f]
        lo, hi = bank.supports[f]
        need_f, need_t = 2 * bank.sampling_rate / (hi_hz - lo_hz), hi - lo
        if need_t < 5 or need_f < 5:
            continue
        N = int(max(need_t, need_f))
        if N > 3000:
            continue  # keep the CPU suite quick; the Python loops here are O(N^2)
        X = bank.get_frequency_response(f, N)
        x = bank.get_impulse_response(f, N)
        assert np.allclose(np.fft.ifft(X), x, atol=1e-3), (kind, f)


def _outside(support_lo, support_hi, count, unit):
    """mask over `count` grid points (spacing `unit`, periodic with period count * unit): True where
    no periodic image of the point lies inside [support_lo, support_hi]; None if the support spans
    more than two periods (nothing to check)"""
    period = count * unit
    first, last = int(np.floor(support_lo / period)), int(np.ceil(support_hi / period))
    if last - first > 2:
        return None, 0
    mask = np.ones(count, dtype=bool)
    for image in range(first, last + 1):
        at = np.arange(count) * unit + image * period
        mask &= (at < support_lo) | (at > support_hi)
    return mask, last - first


@pytest.mark.parametrize("kind", sorted(BANKS))
def test_responses_vanish_outside_the_declared_supports(kind):
    # reference tests/test_filters.py:148-196: `supports_hz` bounds the frequency response and
    # `supports` the impulse response, each up to the effective-support threshold per period
    bank = BANKS[kind](11)
    eps = ps.config.EFFECTIVE_SUPPORT_THRESHOLD
    for f in range(bank.num_filts):
        lo_hz, hi_hz = bank.supports_hz[f]
        N = int(max(1, 2 * bank.sampling_rate / (hi_hz - lo_hz)))
        mask, periods = _outside(lo_hz, hi_hz, N, bank.sampling_rate / N)
        if mask is not None:
            if bank.is_real:  # a real filter's response is Hermitian: the mirror image counts too
                mask[1:] &= mask[-1:0:-1]
            if mask.any():
                X = bank.get_frequency_response(f, N)
                assert np.allclose(X[mask], 0, atol=periods * eps), (kind, f, "frequency")
        lo, hi = bank.supports[f]
        width = int(max(1, hi - lo))
        mask, periods = _outside(lo, hi, width, 1)
        if mask is not None and mask.any() and width <= 3000:
            x = bank.get_impulse_response(f, width)
            assert np.allclose(x[mask], 0, atol=periods * eps), (kind, f, "time")


@pytest.mark.parametrize("window_size", [10, 100, 1000])
@pytest.mark.parametrize("peak_ratio", [0.5, 0.75, 0.9])
@pytest.mark.parametrize("order", [2, 4])
def test_gamma_window_peak(window_size, peak_ratio, order):
    w = ps.filters.GammaWindow(order=order, peak=peak_ratio).get_impulse_response(window_size)
    k = int(np.argmax(w))
    assert int(window_size * peak_ratio) in (k, k + 1)


def test_scales_invertible():
    # reference tests/test_scales.py:20-26
    for scale in (ps.scales.LinearScaling(10.0, 3.0), ps.scales.OctaveScaling(19.0),
                  ps.scales.MelScaling(), ps.scales.BarkScaling()):
        for hz in np.linspace(20, 7999, 200):
            assert scale.scale_to_hertz(scale.hertz_to_scale(hz)) == pytest.approx(hz)


# ---- the C ABI ------------------------------------------------------------------------------


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "pds_amd.h")).read()
    declared = set(re.findall(r"\b(pds_[a-z0-9_]+)\s*\(", header))
    declared -= {"pds_stft_desc", "pds_stft_plan"}
    assert declared == set(_native.SIGNATURES), declared ^ set(_native.SIGNATURES)
    lib = _native.lib()
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.pds_version() >= 100
    assert ctypes.sizeof(_native.StftDesc) == 48


def test_no_cpu_fallback_without_a_device():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    comp = build({"name": "stft", "bank": "fbank"})
    with pytest.raises(_native.NativeError, match="no HIP device"):
        comp.compute_full(np.zeros(1000, dtype=np.float32))
    with pytest.raises(_native.NativeError, match="no HIP device"):
        Deltas(2).apply(np.zeros((10, 3), dtype=np.float32), axis=0)
    with pytest.raises(_native.NativeError, match="no HIP device"):
        Standardize().apply(np.ones((10, 3)))


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "pydrobert-speech_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f


def test_new_entry_points_validate_arguments_without_a_device():
    # argument checks come before any HIP call: they can be exercised on a machine without a GPU
    lib = _native.lib()
    assert ctypes.sizeof(_native.SiDesc) == 40
    null = None
    rc = lib.pds_stack_rows_f32(null, 4, null, null, null, 1, 1, 4, 0, 0, null, 4, null)  # num_vectors 0
    assert rc < 0 and b"stack_rows" in lib.pds_last_error()
    rc = lib.pds_stack_rows_f32(null, 4, null, null, null, 1, 1, 4, 2, 0, null, 8, null)  # null pointers
    assert rc < 0 and b"null pointer" in lib.pds_last_error()
    assert lib.pds_stack_rows_f32(null, 4, null, null, null, 0, 0, 4, 2, 0, null, 8, null) == 0  # empty batch
    rc = lib.pds_si_batch_f32(null, null, null, null, null, null, 1, 1, 0, null, null, 1, null)
    assert rc < 0 and b"null plan" in lib.pds_last_error()
    assert lib.pds_si_scratch_len(null, 4, 100) == 0 and lib.pds_si_plan_fft_size(null) == 0
    desc = _native.SiDesc(frame_shift=0, max_support=10, num_coeffs=1, taps_complex=0, use_power=1, use_log=1,
                          reserved=0, reserved2=0, log_floor=1e-5)
    handle = ctypes.c_void_p()
    taps = np.zeros(10)
    rc = lib.pds_si_plan_create(ctypes.byref(desc), taps.ctypes.data, taps.ctypes.data, ctypes.byref(handle))
    assert rc < 0 and b"frame_shift" in lib.pds_last_error()
    rc = lib.pds_deltas_rows_f32(null, 4, null, null, 1, 5, 4, null, null, 2, 4, null, 4, null)
    assert rc < 0 and b"deltas_rows" in lib.pds_last_error()


def test_widened_classes_have_no_cpu_path_either():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    from pydrobert_speech_amd.pre import Dither, Preemphasize

    si = build({"name": "si", "bank": {"name": "gabor", "scaling_function": "mel", "num_filts": 3}})
    for call in (lambda: si.compute_full(np.zeros(1000, dtype=np.float32)),
                 lambda: si.compute_chunk(np.zeros(1000, dtype=np.float32)),
                 lambda: Preemphasize().apply(np.zeros(10, dtype=np.float32)),
                 lambda: Dither().apply(np.zeros(10, dtype=np.float32))):
        with pytest.raises(_native.NativeError, match="no HIP device"):
            call()


def test_parity_helper_sees_nan():
    """A NaN in a kernel's output must fail the comparison (and a NaN the reference has must be there)"""
    from tests.conftest import assert_features_close

    assert_features_close([np.nan, 1.0], [np.nan, 1.0], 1e-4, 1e-5)
    with pytest.raises(AssertionError):
        assert_features_close([np.nan, 1.0], [0.5, 1.0], 1e-4, 1e-5)
    with pytest.raises(AssertionError):
        assert_features_close([0.5, 1.0], [np.nan, 1.0], 1e-4, 1e-5)
    with pytest.raises(AssertionError):
        assert_features_close([0.5, 1.1], [0.5, 1.0], 1e-4, 1e-5)


def test_comm_entry_points_validate_arguments_without_a_device():
    # argument checks of the gather exports come before RCCL is even loaded
    lib = _native.lib()
    assert lib.pds_comm_world(None) == 0 and lib.pds_comm_rank(None) == -1
    lib.pds_comm_destroy(None)
    assert lib.pds_comm_unique_id(None) == -1 and b"null" in lib.pds_last_error()
    handle = ctypes.c_void_p()
    ident = ctypes.create_string_buffer(128)
    assert lib.pds_comm_init_rank(ident, 2, 2, ctypes.byref(handle)) == -1 and b"rank" in lib.pds_last_error()
    assert lib.pds_gather_rows(None, None, None, 4, None, None) == -1
    assert lib.pds_allreduce_sum_f64(None, None, 4, None) == -1
