"""ShortIntegrationFrameComputer on the GPU (csrc/si.hip through the C ABI) against outputs of the
reference (tests/golden/si.npz) and, for sizes without fixtures, the pinned oracle."""
import json
import os

import numpy as np
import pytest

from oracle import si_oracle as so
from pydrobert_speech_amd.alias import alias_factory_subclass_from_arg
from pydrobert_speech_amd.compute import FrameComputer, frame_by_frame_calculation
from tests.conftest import GOLDEN

pytestmark = pytest.mark.gpu

with open(os.path.join(GOLDEN, "si_configs.json")) as _fh:
    META = json.load(_fh)
NAMES = sorted(META["configs"])
# float32: filters of up to ~250 taps accumulated in float32, features are logs (or raw sums)
F32 = dict(rtol=2e-4, atol=2e-5)


@pytest.fixture(scope="module")
def gsi():
    with np.load(os.path.join(GOLDEN, "si.npz")) as z:
        return {k: z[k] for k in z.files}


def build(name):
    return alias_factory_subclass_from_arg(FrameComputer, json.loads(json.dumps(META["configs"][name])))


def close(got, want, rtol, atol):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, (got.shape, want.shape)
    if want.size:
        scale = np.maximum(np.abs(want), np.abs(want).max() * 1e-3)  # raw sums span decades
        assert (np.abs(got - want) <= atol + rtol * scale).all(), float(np.abs(got - want).max())


@pytest.mark.parametrize("name", NAMES)
def test_compute_full_matches_reference(gsi, name):
    comp = build(name)
    for n in META["lengths"][name]:
        x4 = gsi["master"][:n].astype("f4")
        got = comp.compute_full(x4)
        assert got.dtype == np.float32
        close(got, gsi[f"{name}/full/{n}/f4"], **F32)
        got8 = comp.compute_full(gsi["master"][:n].astype("f8"))
        assert got8.dtype == np.float64
        close(got8, gsi[f"{name}/full/{n}/f8"], rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("name", NAMES)
def test_batch_equals_single_and_read_only_input(gsi, name):
    comp = build(name)
    sigs = [gsi["master"][:n].astype("f4") for n in META["lengths"][name]]
    for s in sigs:
        s.flags.writeable = False
    got = comp.compute_full_batch(sigs)
    for n, y in zip(META["lengths"][name], got):
        close(y, gsi[f"{name}/full/{n}/f4"], **F32)
    assert comp.compute_full_batch([]) == []


@pytest.mark.parametrize("tag", ["c1024", "ragged"])
def test_streaming_emits_the_references_frames(gsi, tag):
    comp = build("s1_gabor_mel")
    x = gsi["master"][:4001].astype("f4")
    pieces = np.split(x, gsi[f"s1_gabor_mel/stream/{tag}/cuts"])
    outs = [comp.compute_chunk(p) for p in pieces]
    assert comp.started
    outs.append(comp.finalize())
    assert not comp.started
    assert [len(o) for o in outs] == gsi[f"s1_gabor_mel/stream/{tag}/counts"].tolist()
    close(np.concatenate(outs), gsi[f"s1_gabor_mel/stream/{tag}/feats"], **F32)
    # a second stream on the same object starts clean; the generic driver agrees with compute_full
    close(frame_by_frame_calculation(comp, x, 333), gsi["s1_gabor_mel/full/4001/f4"], **F32)


def test_stream_and_dtype_errors(gsi):
    comp = build("s2_gammatone_power")
    with pytest.raises(ValueError, match="float type"):
        comp.compute_chunk(np.arange(10))
    with pytest.raises(ValueError, match="float type"):
        comp.compute_full(np.arange(10))
    comp.compute_chunk(np.zeros(10, "f4"))
    with pytest.raises(ValueError, match="share a type"):
        comp.compute_chunk(np.zeros(10, "f8"))
    with pytest.raises(ValueError, match="Already started"):
        comp.compute_full(np.zeros(10, "f4"))
    assert comp.finalize().shape == (0, comp.num_coeffs)
    assert comp.finalize().shape == (0, comp.num_coeffs)  # idle finalize: empty float64


def test_full_bank_long_signal_against_oracle():
    # 40 complex Gabor filters (supports up to ~380 taps), several workgroup tiles, ragged batch
    comp = alias_factory_subclass_from_arg(
        FrameComputer, {"name": "si", "bank": {"name": "gabor", "scaling_function": "mel", "num_filts": 40},
                        "include_energy": True, "use_power": True})
    p = so.SiParams(comp.frame_shift, comp._max_support, comp._translation, comp.dft_size, comp.taps,
                    comp._window.reshape(-1), comp.frame_style == "centered", True, True)
    rng = np.random.default_rng(3)
    sigs = [(3000 * rng.standard_normal(n)).astype("f4") for n in (16000, 2239, 2240, 2241, 5000)]
    got = comp.compute_full_batch(sigs)
    for x, y in zip(sigs, got):
        close(y, so.compute_full(x, p), **F32)


def test_fft_and_direct_forms_agree_and_cover_the_fixtures(gsi):
    # float32 takes the overlap-save FFT form by default; the direct form must give the same
    # features (both within tolerance of the reference), for every fixture configuration
    import torch

    for name in NAMES:
        comp = build(name)
        n = META["lengths"][name][-1]
        x = torch.from_numpy(gsi["master"][:n].astype("f4")).cuda()
        fft, rows = comp.compute_packed(x, [0], [n])
        direct, _ = comp.compute_packed(x, [0], [n], direct=True)
        want = gsi[f"{name}/full/{n}/f4"]
        assert rows.tolist() == [0, want.shape[0]]
        close(fft.cpu().numpy(), want, **F32)
        close(direct.cpu().numpy(), want, **F32)
        close(fft.cpu().numpy(), direct.cpu().numpy(), **F32)


def test_transform_size_follows_the_filter_supports():
    sizes = {name: build(name).fft_size for name in NAMES}
    assert sizes["s1_gabor_mel"] == 1024 and sizes["s4_gabor_8k_short"] == 1024
    assert sizes["s6_fbank_long"] == 2048  # 1037 taps: beyond 1024 - S
    big = alias_factory_subclass_from_arg(FrameComputer, {"name": "si", "bank": {"name": "fbank", "num_filts": 40}})
    assert big._max_support > 2048 and big.fft_size == 0  # ~7000 taps: direct form only


def test_real_bank_with_long_supports_against_oracle():
    # triangular filters are real (half the multiplies) and long (support ~750 samples)
    comp = alias_factory_subclass_from_arg(
        FrameComputer, {"name": "si", "bank": {"name": "tri", "scaling_function": "mel", "num_filts": 10},
                        "frame_shift_ms": 25, "use_log": False})
    p = so.SiParams(comp.frame_shift, comp._max_support, comp._translation, comp.dft_size, comp.taps,
                    comp._window.reshape(-1), comp.frame_style == "centered", False, False)
    x = (np.random.default_rng(4).standard_normal(9000) * 100).astype("f8")
    close(comp.compute_full(x), so.compute_full(x, p), rtol=1e-9, atol=1e-9)


def test_long_frame_shift_takes_several_passes():
    # 50 ms at 48 kHz = 2400 samples per block: beyond one pass of the direct kernel's thread block
    # and beyond the FFT form's 1024-point transforms (which then declines: scratch length 0)
    comp = alias_factory_subclass_from_arg(
        FrameComputer, {"name": "si", "bank": {"name": "gabor", "scaling_function": "mel", "num_filts": 4,
                                               "sampling_rate": 48000}, "frame_shift_ms": 50, "use_power": True})
    assert comp.frame_shift == 2400
    p = so.SiParams(comp.frame_shift, comp._max_support, comp._translation, comp.dft_size, comp.taps,
                    comp._window.reshape(-1), comp.frame_style == "centered", True, True)
    rng = np.random.default_rng(11)
    for n in (30000, 2399, 7201):
        x = (rng.standard_normal(n) * 1000).astype("f4")
        close(comp.compute_full(x), so.compute_full(x, p), **F32)
    x8 = (rng.standard_normal(12000) * 1000).astype("f8")
    close(comp.compute_full(x8), so.compute_full(x8, p), rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("name", NAMES)
def test_streaming_random_chunkings_call_by_call(name):
    # tests/golden/make_golden_stream.py: six lengths x random cut points per configuration
    with np.load(os.path.join(GOLDEN, "si_stream_random.npz")) as z:
        g = {k: z[k] for k in z.files if k.startswith(name + "/")}
    with np.load(os.path.join(GOLDEN, "si.npz")) as z:
        master = z["master"]
    comp = build(name)
    for case in range(6):
        n = int(g[f"{name}/{case}/n"])
        x = master[7 : 7 + n].astype("f4")
        outs = [comp.compute_chunk(p) for p in np.split(x, g[f"{name}/{case}/cuts"])] + [comp.finalize()]
        assert [len(o) for o in outs] == g[f"{name}/{case}/counts"].tolist(), (name, case)
        close(np.concatenate(outs), g[f"{name}/{case}/feats"], **F32)


@pytest.mark.parametrize("rate,shift_ms,bank,num_filts,use_power,size", [
    (8000, 5, "gabor", 5, True, 1024),         # S = 40: <= 3 window factors per lane and half
    (16000, 10, "gabor", 5, False, 1024),      # S = 160: 5 factors; magnitudes (the square-root branch)
    (16000, 15, "gabor", 5, True, 1024),       # S = 240: 8 factors
    (16000, 25, "gabor", 5, True, 1024),       # S = 400: 16 factors
    (48000, 2.5, "gammatone", 40, True, 2048),  # S = 120, 754 taps: 2048-point form (64 lanes), 3 factors
    (48000, 5, "gammatone", 5, True, 2048),    # S = 240: 5 factors
    (48000, 10, "gabor", 5, False, 2048),      # S = 480: 8 factors
    (48000, 16, "gabor", 5, True, 2048),       # S = 768: 16 factors
])
def test_fft_form_window_factor_buckets_against_oracle(rate, shift_ms, bank, num_filts, use_power, size):
    """csrc/si_fft.hip keeps ceil(S / lanes) window factors per lane in registers, built for 3 / 5 / 8 / 16 of them,
    for 32 (1024-point transforms) or 64 lanes (2048-point): every instantiation against the oracle, the direct form
    beside it, with lengths that leave partial transforms and partial blocks"""
    import torch

    comp = alias_factory_subclass_from_arg(
        FrameComputer, {"name": "si", "bank": {"name": bank, "scaling_function": "mel", "num_filts": num_filts,
                                               "sampling_rate": rate},
                        "frame_shift_ms": shift_ms, "use_power": use_power, "use_log": use_power})
    assert comp.fft_size == size, (comp.fft_size, comp._max_support, comp.frame_shift)
    p = so.SiParams(comp.frame_shift, comp._max_support, comp._translation, comp.dft_size, comp.taps,
                    comp._window.reshape(-1), comp.frame_style == "centered", use_power, use_power)
    rng = np.random.default_rng(int(rate + 10 * shift_ms))
    S = comp.frame_shift
    lens = [0, 1, S - 1, 3 * S + 7, 41 * S + S // 2, 9000]
    sigs = [(1000 * rng.standard_normal(n)).astype("f4") for n in lens]
    got = comp.compute_full_batch(sigs)
    for x, y in zip(sigs, got):
        close(y, so.compute_full(x, p), **F32)
    x = torch.from_numpy(np.concatenate(sigs)).cuda()
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]])
    fft, rows = comp.compute_packed(x, offs, lens)
    direct, _ = comp.compute_packed(x, offs, lens, direct=True)
    close(fft.cpu().numpy(), direct.cpu().numpy(), **F32)
