"""Parity of the HIP STFT path (through the C ABI) with the reference: golden fixtures
written by the reference, and the pinned oracle on fresh seeded inputs.

Tolerances.  float32 signals: the north star's bar is 1e-4 relative; features are logs
that may sit near 0, so the check is |got - want| <= 1e-5 + 1e-4 |want| (SURVEY.md
section 0), i.e. numpy.allclose(rtol=1e-4, atol=1e-5).  float64 signals run in float64 on
the GPU and must agree to 1e-9.
"""
import json

import numpy as np
import pytest

from pydrobert_speech_amd.alias import alias_factory_subclass_from_arg
from pydrobert_speech_amd.compute import FrameComputer, frame_by_frame_calculation
from oracle import stft_oracle as orc
import os

from tests.conftest import GOLDEN, assert_features_close, config_names, oracle_params

pytestmark = pytest.mark.gpu
CONFIGS = config_names()
F32 = dict(rtol=1e-4, atol=1e-5)
F64 = dict(rtol=1e-9, atol=1e-9)


def build(cfg):
    return alias_factory_subclass_from_arg(FrameComputer, json.loads(json.dumps(cfg)))


@pytest.fixture(scope="module")
def computers(golden_meta):
    return {name: build(golden_meta["configs"][name]) for name in CONFIGS}


@pytest.mark.parametrize("name", CONFIGS)
def test_compute_full_matches_reference_outputs(name, computers, golden_meta, golden_stft, master_signal):
    comp = computers[name]
    for n in golden_meta["lengths"][name]:
        for dt, tol in (("f4", F32), ("f8", F64)):
            x = master_signal[:n].astype(dt)
            x.flags.writeable = False  # the reference's tests pass read-only buffers
            got = comp.compute_full(x)
            want = golden_stft[f"{name}/{n}/{dt}"]
            assert got.dtype == want.dtype
            assert_features_close(got, want, what=(name, n, dt), **tol)


@pytest.mark.parametrize("name", CONFIGS)
def test_generic_kernel_matches_reference_outputs(name, computers, golden_meta, golden_stft, master_signal):
    # float32 through the direct-DFT kernel whatever the plan would pick
    import torch

    comp = computers[name]
    n = golden_meta["lengths"][name][-1]
    x = torch.from_numpy(master_signal[:n].astype("f4")).cuda()
    feats, rows = comp.compute_packed(x, [0], [n], generic=True)
    assert rows.tolist() == [0, comp.num_frames(n)]
    assert_features_close(feats.cpu().numpy(), golden_stft[f"{name}/{n}/f4"], what=name, **F32)


@pytest.mark.parametrize("name", CONFIGS)
def test_batch_ragged_matches_oracle(name, computers, golden_tables):
    # a ragged batch (including empty and too-short utterances) in one launch
    comp = computers[name]
    p = oracle_params(golden_tables, name)
    L, S = comp.frame_length, comp.frame_shift
    rng = np.random.default_rng(4321)
    lens = [0, 1, L // 2, L // 2 + 1, L, L + 1, 3 * L + 5, 40 * S + 3, 17 * S, 64 * S, 65 * S + S // 2, 2]
    sigs = [(3000 * rng.standard_normal(n)).astype("f4") for n in lens]
    got = comp.compute_full_batch(sigs)
    assert len(got) == len(sigs)
    for x, y in zip(sigs, got):
        assert y.dtype == np.float32 and y.shape == (comp.num_frames(len(x)), comp.num_coeffs)
        assert_features_close(y, orc.compute_full(x, p), what=(name, len(x)), **F32)


def test_known_answer_kaldi(computers, golden_kaldi):
    # the reference's tests/test_compute.py:190-208 on the GPU path
    comp = computers["c1_kaldi_fbank"]
    feats = comp.compute_full(golden_kaldi["noise"]).astype(np.float64)
    feats += 2 * np.log(0.5 * (comp.frame_length - 1))
    feats -= np.log(2)
    assert feats.shape == golden_kaldi["kaldi_feats"].shape
    assert np.allclose(feats, golden_kaldi["kaldi_feats"], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("name", CONFIGS)
def test_streaming_matches_reference_outputs(name, computers, golden_stream, master_signal):
    comp = computers[name]
    n = 5 * comp.frame_length
    x = master_signal[100 : 100 + n].astype("f4")
    for tag, chunks in (("c7", [7] * (n // 7 + 1)), ("c1024", [1024] * (n // 1024 + 1)),
                        ("mixed", [1, 0, 3 * comp.frame_length, 50, 1, 10 ** 6])):
        if tag == "c7" and n > 3000:
            chunks = None  # hundreds of tiny launches; covered by the shorter configs
        if chunks is None:
            continue
        pieces, pos = [], 0
        for c in chunks:
            if pos >= n:
                break
            pieces.append(comp.compute_chunk(x[pos : pos + c]))
            pos += c
        assert comp.started or n == 0
        pieces.append(comp.finalize())
        assert not comp.started
        got = np.concatenate(pieces)
        assert_features_close(got, golden_stream[f"{name}/{tag}"], what=(name, tag), **F32)
    # frame_by_frame_calculation feeds 1024-sample chunks: same as the "c1024" chunking.  (The
    # reference's chunked and full paths agree only while finalize's right padding fits in
    # its one-frame buffer -- compute.py:558-561 reflects the buffer, not the signal -- so
    # chunked output is pinned against the reference's chunked output.)
    assert_features_close(frame_by_frame_calculation(comp, x), golden_stream[f"{name}/c1024"], what=name, **F32)
    assert_features_close(comp.compute_full(x), golden_stream[f"{name}/full"], what=name, **F32)


def test_streaming_short_signal_quirk(computers, golden_stream, master_signal):
    # below L // 2 + 1 samples the reference's chunked path still emits a frame while
    # compute_full returns none (compute.py:552-561 vs 580-581); both are reproduced
    comp = computers["c1_kaldi_fbank"]
    x = master_signal[:150].astype("f4")
    got = np.concatenate([comp.compute_chunk(x), comp.finalize()])
    assert_features_close(got, golden_stream["short150/chunked"], **F32)
    assert comp.compute_full(x).shape == golden_stream["short150/full"].shape == (0, 40)


def test_contract_empty_started_and_repeatability(computers):
    # reference tests/test_compute.py:83-110
    comp = computers["c4_gabor64"]
    for dt in (np.float32, np.float64):
        assert comp.compute_full(np.empty(0, dt)).shape == (0, comp.num_coeffs)
        assert comp.compute_chunk(np.empty(0, dt)).shape == (0, comp.num_coeffs)
        assert comp.finalize().shape == (0, comp.num_coeffs)
    buff = np.random.default_rng(0).random(comp.frame_length * 2)
    buff.flags.writeable = False
    coeffs = np.concatenate([comp.compute_chunk(buff), comp.finalize()])
    assert coeffs.shape[0] >= 1 and coeffs.dtype == np.float64
    assert comp.finalize().shape == (0, comp.num_coeffs)
    assert not comp.started
    comp.compute_chunk(np.empty(1))
    assert comp.started
    with pytest.raises(ValueError, match="Already started computing frames"):
        comp.compute_full(buff)
    with pytest.raises(ValueError, match="Already started computing frames"):
        frame_by_frame_calculation(comp, buff)
    comp.finalize()
    assert not comp.started
    a, b = comp.compute_full(buff), comp.compute_full(buff)
    assert np.array_equal(a, b)


def test_gpu_tensors_stay_on_gpu(computers, golden_tables):
    import torch

    comp = computers["c3_fbank80_energy"]
    p = oracle_params(golden_tables, "c3_fbank80_energy")
    x = (3000 * np.random.default_rng(5).standard_normal(5000)).astype("f4")
    y = comp.compute_full(torch.from_numpy(x).cuda())
    assert y.is_cuda and y.dtype == torch.float32
    assert_features_close(y.cpu().numpy(), orc.compute_full(x, p), **F32)


def test_full_size_linearity_property(computers, golden_tables):
    # BASELINE.json configs[1] at its full size (1024 utterances x 160 000 samples, one launch): scaling the
    # signal by a shifts every log-power feature by 2 log a (size-independent property), frames do not
    # depend on the batch position, and three utterances of the very launch match the oracle
    import torch

    comp = computers["c2_tri_mel40"]
    B, n = 1024, 160000
    g = torch.Generator(device="cuda").manual_seed(7)
    x = 3000 * torch.randn(B * n, generator=g, device="cuda", dtype=torch.float32)
    offs, lens = np.arange(B) * n, np.full(B, n)
    y1, rows = comp.compute_packed(x, offs, lens)
    assert rows[-1] == B * 1000 and y1.shape == (B * 1000, 40)
    assert torch.isfinite(y1).all()
    x.mul_(4.0)
    y2, _ = comp.compute_packed(x, offs, lens)
    assert torch.allclose(y2 - y1, torch.full_like(y1, 2 * np.log(4.0)), atol=2e-4)
    # frames are independent of batch position: utterances alone give the same rows
    for b in (3, B - 1):
        y3, _ = comp.compute_packed(x[b * n : (b + 1) * n].clone(), [0], [n])
        assert torch.equal(y3, y2[b * 1000 : (b + 1) * 1000])
    p = oracle_params(golden_tables, "c2_tri_mel40")
    for b in (0, 517, B - 1):
        want = orc.compute_full(x[b * n : (b + 1) * n].cpu().numpy(), p)
        assert_features_close(y2[b * 1000 : (b + 1) * 1000].cpu().numpy(), want, **F32)


@pytest.mark.parametrize("fused", [True, False], ids=["one-launch", "two-launches"])
def test_full_size_statics_plus_deltas_chain(computers, fused):
    # BASELINE.json configs[2] per GPU at the benchmark batch: 80 mel + energy written with row stride 243,
    # Deltas(2) beside them -- by the one launch that forms the deltas from the coefficients in its registers
    # (pds_stft_deltas_batch_f32, what bench.py times) and by the STFT launch followed by the deltas launch.
    # Scaling the signal shifts the statics by 2 log a and leaves the deltas alone (the filters sum to
    # zero); an utterance alone gives the same rows.
    import torch

    from pydrobert_speech_amd.post import Deltas

    comp = computers["c3_fbank80_energy"]
    C = comp.num_coeffs
    B, n = 1024, 160000
    g = torch.Generator(device="cuda").manual_seed(11)
    x = 3000 * torch.randn(B * n, generator=g, device="cuda", dtype=torch.float32)
    offs, lens = np.arange(B) * n, np.full(B, n)
    layout = comp.prepare_layout(offs, lens, device=x.device)
    deltas = Deltas(2)

    def chain(sig, lay):
        out = torch.full((lay.total_rows, 3 * C), float("nan"), dtype=torch.float32, device=sig.device)
        if fused:
            assert comp._native_plan(sig.device).has_fused_deltas
        comp.launch_with_deltas(sig, lay, deltas, out=out, fused=fused)
        return out

    y1 = chain(x, layout)
    assert y1.shape == (B * 1000, 243) and torch.isfinite(y1).all()
    x.mul_(4.0)
    y2 = chain(x, layout)
    diff = y2 - y1
    assert torch.allclose(diff[:, :C], torch.full_like(diff[:, :C], 2 * np.log(4.0)), atol=2e-4)
    assert diff[:, C:].abs().max().item() < 2e-4
    b = 700
    one = comp.prepare_layout([0], [n], device=x.device)
    y3 = chain(x[b * n : (b + 1) * n].clone(), one)
    assert torch.equal(y3, y2[b * 1000 : (b + 1) * 1000])
    # the deltas of the launch against the oracle's, on some utterances' statics (the one launch accumulates
    # in float32: a few ulps of the statics, which are ~20 here)
    for b in (0, 333, B - 1):
        stat = y2[b * 1000 : (b + 1) * 1000, :C].cpu().numpy()
        want = orc.deltas(stat, axis=0, num_deltas=2, target_axis=-1)
        tol = 1e-5 if fused else 1e-6
        assert np.allclose(y2[b * 1000 : (b + 1) * 1000].cpu().numpy(), want, rtol=tol, atol=tol)


def test_full_size_gammatone_cmvn_chain(computers):
    # BASELINE.json configs[4] per GPU at the benchmark batch: 64 gammatone filters at 48 kHz (N = 1024), then
    # per-utterance CMVN (float64 out).  Scaling the signal moves every log feature by the same constant,
    # which the mean removal takes out again; an utterance alone gives the same rows.
    import torch

    from pydrobert_speech_amd.post import CMVN

    comp = computers["c5_gammatone64_48k"]
    B, n = 256, 480000
    g = torch.Generator(device="cuda").manual_seed(13)
    x = 3000 * torch.randn(B * n, generator=g, device="cuda", dtype=torch.float32)
    offs, lens = np.arange(B) * n, np.full(B, n)
    cmvn = CMVN()
    f1, rows = comp.compute_packed(x, offs, lens)
    assert rows[-1] == B * 1000 and f1.shape == (B * 1000, 64) and torch.isfinite(f1).all()
    c1 = cmvn.apply_rows(f1, rows)
    assert c1.dtype == torch.float64 and torch.isfinite(c1).all()
    x.mul_(4.0)
    f2, _ = comp.compute_packed(x, offs, lens)
    c2 = cmvn.apply_rows(f2, rows)
    assert torch.allclose(f2 - f1, torch.full_like(f1, 2 * np.log(4.0)), atol=2e-4)
    # (standardised features: float32 round-off of a log, ~2e-4, over a spread of 0.05 - 1 per coefficient)
    assert (c2 - c1).abs().max().item() < 1e-2
    b = 200
    f3, r3 = comp.compute_packed(x[b * n : (b + 1) * n].clone(), [0], [n])
    assert torch.equal(f3, f2[b * 1000 : (b + 1) * 1000])
    assert torch.equal(cmvn.apply_rows(f3, r3), c2[b * 1000 : (b + 1) * 1000])
    want = orc.cmvn_local(f2[:1000].cpu().numpy(), axis=-1)
    assert np.allclose(c2[:1000].cpu().numpy(), want, rtol=1e-8, atol=1e-8)


def _params_from_computer(comp):
    return orc.StftParams(
        frame_length=comp.frame_length, frame_shift=comp.frame_shift, dft_size=comp.dft_size,
        window=np.asarray(comp._window), starts=list(comp._filt_start_idxs),
        taps=[np.asarray(t) for t in comp._truncated_filts], is_real=comp.bank.is_real,
        centered=comp.frame_style == "centered", kaldi_shift=comp.kaldi_shift,
        include_energy=comp.includes_energy, use_power=bool(comp._power), use_log=bool(comp._log),
    )


EXTRA_GEOMETRIES = {
    # N = 2048 (64 x 32 lanes): 25 ms at 48 kHz
    "n2048_fbank_48k": {"name": "stft", "bank": {"name": "fbank", "num_filts": 60, "sampling_rate": 48000},
                        "frame_length_ms": 25, "use_power": True, "include_energy": True},
    # N = 2048, magnitude spectrum, complex bank, causal frames
    "n2048_gammatone_44k": {"name": "stft", "bank": {"name": "gammatone", "scaling_function": "bark",
                            "num_filts": 30, "sampling_rate": 44100}, "frame_length_ms": 30, "use_power": False},
    # N = 4096 (64 x 64: one frame per wavefront): 50 ms at 48 kHz, and 80 ms at 44.1 kHz with energy,
    # a complex bank and the magnitude spectrum
    "n4096_fbank_48k": {"name": "stft", "bank": {"name": "fbank", "num_filts": 80, "sampling_rate": 48000},
                        "frame_length_ms": 50, "frame_shift_ms": 12.5, "use_power": True},
    "n4096_gabor_44k": {"name": "stft", "bank": {"name": "gabor", "scaling_function": "mel", "num_filts": 48,
                        "sampling_rate": 44100}, "frame_length_ms": 80, "frame_shift_ms": 20,
                        "use_power": False, "include_energy": True, "frame_style": "causal"},
    # N = 1024 with every row in use (L = 1024)
    "n1024_full_rows": {"name": "stft", "bank": {"name": "tri", "scaling_function": "mel", "num_filts": 50,
                        "sampling_rate": 32000}, "frame_length_ms": 32, "frame_shift_ms": 8, "use_power": True},
    # N = 128 (16 x 8 lanes): 12.5 ms at 8 kHz
    "n128_fbank_8k": {"name": "stft", "bank": {"name": "fbank", "num_filts": 12, "sampling_rate": 8000},
                      "frame_length_ms": 12.5, "frame_shift_ms": 5, "use_power": True, "kaldi_shift": True},
    # N = 512 with a frame length that is not a multiple of the 16-sample row (L = 330)
    "n512_partial_row": {"name": "stft", "bank": {"name": "gabor", "scaling_function": "mel", "num_filts": 33},
                         "frame_length_ms": 20.625, "use_power": True, "include_energy": True},
    # N = 256 with L = 200 (25 rows of 8)
    "n256_tri_8k": {"name": "stft", "bank": {"name": "tri", "scaling_function": "linear", "num_filts": 17,
                    "sampling_rate": 8000, "scaling_function": {"name": "linear", "low_hz": 0.0}},
                    "frame_length_ms": 25, "use_log": False},
    # ---- no zero padding: N = L, mixed-radix geometries N1 x N2 --------
    # 400 = 25 x 16: the headline bank; 13 of the 16 lanes of a frame own a column
    "nopad400_tri_mel40": {"name": "stft", "bank": {"name": "tri", "scaling_function": "mel", "num_filts": 40},
                           "frame_length_ms": 25, "use_power": True, "pad_to_nearest_power_of_two": False},
    # 320 = 20 x 16 (even in-lane size: Nyquist column too), energy, kaldi shift
    "nopad320_fbank": {"name": "stft", "bank": {"name": "fbank", "num_filts": 23}, "frame_length_ms": 20,
                       "include_energy": True, "use_power": True, "kaldi_shift": True,
                       "pad_to_nearest_power_of_two": False},
    # 480 = 30 x 16, complex bank with wrap-around filters, magnitude
    "nopad480_gabor": {"name": "stft", "bank": {"name": "gabor", "scaling_function": "mel", "num_filts": 30},
                       "frame_length_ms": 30, "use_power": False, "pad_to_nearest_power_of_two": False},
    # 200 = 25 x 8 and 160 = 20 x 8 and 240 = 30 x 8 at 8 kHz (8 frames per wave)
    "nopad200_tri_8k": {"name": "stft", "bank": {"name": "tri", "scaling_function": "bark", "num_filts": 15,
                        "sampling_rate": 8000}, "frame_length_ms": 25, "use_power": True,
                        "pad_to_nearest_power_of_two": False},
    "nopad160_fbank_8k": {"name": "stft", "bank": {"name": "fbank", "num_filts": 10, "sampling_rate": 8000},
                          "frame_length_ms": 20, "frame_shift_ms": 5, "frame_style": "causal", "use_power": True,
                          "pad_to_nearest_power_of_two": False},
    "nopad240_fbank_8k": {"name": "stft", "bank": {"name": "fbank", "num_filts": 12, "sampling_rate": 8000},
                          "frame_length_ms": 30, "use_power": True, "include_energy": True, "use_log": False,
                          "pad_to_nearest_power_of_two": False},
    # 640 = 20 x 32, 800 = 25 x 32, 960 = 30 x 32 (two frames per wave)
    "nopad640_tri_32k": {"name": "stft", "bank": {"name": "tri", "scaling_function": "mel", "num_filts": 45,
                         "sampling_rate": 32000}, "frame_length_ms": 20, "use_power": True,
                         "pad_to_nearest_power_of_two": False},
    "nopad800_fbank_32k": {"name": "stft", "bank": {"name": "fbank", "num_filts": 64, "sampling_rate": 32000},
                           "frame_length_ms": 25, "include_energy": True, "use_power": True,
                           "pad_to_nearest_power_of_two": False},
    "nopad960_gammatone_48k": {"name": "stft", "bank": {"name": "gammatone", "scaling_function": "mel",
                               "num_filts": 24, "sampling_rate": 48000}, "frame_length_ms": 20, "use_power": True,
                               "pad_to_nearest_power_of_two": False},
}


@pytest.mark.parametrize("name", sorted(EXTRA_GEOMETRIES))
def test_other_kernel_geometries_match_oracle(name):
    # configurations without reference fixtures: the fused kernel (and the generic one) against
    # the pinned oracle driven by OUR tables (which test_host.py pins for the fixture configs)
    import torch

    comp = build(EXTRA_GEOMETRIES[name])
    assert comp.kernel_kind == comp.dft_size, (comp.kernel_kind, comp.dft_size)
    p = _params_from_computer(comp)
    L, S = comp.frame_length, comp.frame_shift
    rng = np.random.default_rng(99)
    lens = [0, L // 2, L // 2 + 1, L + 3, 37 * S + 11, 5 * L, 3]
    sigs = [(3000 * rng.standard_normal(n)).astype("f4") for n in lens]
    got = comp.compute_full_batch(sigs)
    for x, y in zip(sigs, got):
        assert_features_close(y, orc.compute_full(x, p), what=(name, len(x)), **F32)
    x = torch.from_numpy(sigs[4]).cuda()
    y_gen, _ = comp.compute_packed(x, [0], [len(sigs[4])], generic=True)
    assert_features_close(y_gen.cpu().numpy(), got[4], what=(name, "generic"), **F32)


def test_nonfinite_samples_do_not_leak_into_other_frames(computers):
    # a NaN poisons exactly the frames that contain it (like the reference), no neighbours
    comp = computers["c2_tri_mel40"]
    x = (3000 * np.random.default_rng(1).standard_normal(16000)).astype("f4")
    x[8000] = np.nan
    y = comp.compute_full(x)
    L, S, pad = comp.frame_length, comp.frame_shift, comp.pad_left
    t = np.arange(y.shape[0])
    contains = (t * S - pad <= 8000) & (8000 < t * S - pad + L)
    assert np.isnan(y[contains]).all()
    assert np.isfinite(y[~contains]).all()


def test_float64_signals_can_opt_into_float32_arithmetic(computers, golden_stft, golden_meta, master_signal):
    # config.FLOAT64_ARITHMETIC = "float32": float64 in, float64 out, fused float32 kernel inside
    from pydrobert_speech_amd import config

    comp = computers["c2_tri_mel40"]
    n = max(golden_meta["lengths"]["c2_tri_mel40"])
    x = master_signal[:n].astype("f8")
    want = golden_stft[f"c2_tri_mel40/{n}/f8"]
    exact = comp.compute_full(x)
    assert exact.dtype == np.float64
    assert_features_close(exact, want, rtol=1e-9, atol=1e-9)
    config.FLOAT64_ARITHMETIC = "float32"
    try:
        fast = comp.compute_full(x)
    finally:
        config.FLOAT64_ARITHMETIC = "float64"
    assert fast.dtype == np.float64 and fast.shape == want.shape
    assert_features_close(fast, want, **F32)
    assert np.abs(fast - want).max() > 1e-9  # it really was float32 arithmetic


@pytest.mark.parametrize("name", ["c2_tri_mel40", "c3_fbank80_energy", "c4_gabor64", "c5_gammatone64_48k"])
def test_fused_float64_input_kernel(name, computers, golden_tables, master_signal, monkeypatch):
    """pds_stft_batch_f64in: float64 samples straight into the fused kernel (rounded at the frame load),
    float32 or float64 features -- no conversion pass over the signal (the launch must not call .to)"""
    import torch

    from pydrobert_speech_amd import config

    comp = computers[name]
    assert comp._native_plan().has_f64in
    p = oracle_params(golden_tables, name)
    lens = [0, 1, 150, comp.frame_length // 2 + 1, 2000, 16000, 4801]
    sigs = [master_signal[100 : 100 + n].astype("f8") for n in lens]
    want = [orc.compute_full(x, p) for x in sigs]
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]])
    packed = torch.from_numpy(np.concatenate(sigs)).cuda()
    layout = comp.prepare_layout(offs, lens, device=packed.device)
    rows = layout.row_offsets
    out32 = torch.empty((int(rows[-1]), comp.num_coeffs), dtype=torch.float32, device="cuda")
    monkeypatch.setattr(config, "FLOAT64_ARITHMETIC", "float32")
    monkeypatch.setattr(torch.Tensor, "to", lambda *a, **k: pytest.fail("the launch converted a tensor"))
    feats64 = comp.launch(packed, layout)
    feats32 = comp.launch(packed, layout, out=out32)
    monkeypatch.undo()
    assert feats64.dtype == torch.float64 and feats32.dtype == torch.float32
    assert torch.equal(feats64, feats32.double())  # one kernel, two store widths
    got = feats32.cpu().numpy()
    for b, w in enumerate(want):
        assert_features_close(got[rows[b] : rows[b + 1]], w, **F32)


def test_fused_float64_input_with_preemphasis(computers, golden_tables, master_signal, monkeypatch):
    # float64 samples are pre-emphasised in float64 before the rounding, like the reference's own pass
    import torch

    from pydrobert_speech_amd import config
    from pydrobert_speech_amd.pre import Preemphasize

    comp = computers["c2_tri_mel40"]
    p = oracle_params(golden_tables, "c2_tri_mel40")
    lens = [3000, 401, 16000]
    sigs = [master_signal[7 : 7 + n].astype("f8") for n in lens]
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]])
    packed = torch.from_numpy(np.concatenate(sigs)).cuda()
    monkeypatch.setattr(config, "FLOAT64_ARITHMETIC", "float32")
    out32 = torch.empty((sum(comp.num_frames(n) for n in lens), comp.num_coeffs), dtype=torch.float32, device="cuda")
    feats, rows = comp.compute_packed(packed, offs, lens, preemphasis=0.97, out=out32)
    wide, _ = comp.compute_packed(packed, offs, lens, preemphasis=0.97)
    monkeypatch.undo()
    assert wide.dtype == torch.float64 and torch.equal(wide, feats.double())
    got = feats.cpu().numpy()
    for b, x in enumerate(sigs):
        y = x.copy()
        y[1:] -= 0.97 * x[:-1]
        assert_features_close(got[rows[b] : rows[b + 1]], orc.compute_full(y, p), rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("preemph", [0.0, 0.97], ids=["plain", "preemph"])
@pytest.mark.parametrize("name", ["c1_kaldi_fbank", "c2_tri_mel40", "c3_fbank80_energy", "c4_gabor64", "c5_gammatone64_48k"])
def test_fused_int16_input_kernel(name, preemph, computers, golden_tables, monkeypatch):
    """pds_stft_batch_i16in: int16 PCM straight into the fused kernel (converted at the frame load; the reference's
    readers hold such samples before their .astype, util.py:207-235): against the oracle on the converted signal,
    and equal to the float32 launch on it bit for bit where both take the same filter walk -- with no conversion
    pass (the launch must not call .to)"""
    import torch

    comp = computers[name]
    if not comp._native_plan().has_i16in:
        pytest.skip("no fused int16-input kernel for this transform size")
    p = oracle_params(golden_tables, name)
    rng = np.random.default_rng(5)
    lens = [0, 1, 150, comp.frame_length // 2 + 1, 2000, 16001, 4801]
    sigs = [rng.integers(-32768, 32768, size=n).astype(np.int16) for n in lens]
    sigs[4][:] = np.where(rng.random(lens[4]) < 0.5, -32768, 32767)  # (full scale: the largest magnitudes the format holds)
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]])
    packed = torch.from_numpy(np.concatenate(sigs)).cuda()
    as_float = packed.to(torch.float32)
    layout = comp.prepare_layout(offs, lens, device=packed.device)
    rows = layout.row_offsets
    monkeypatch.setattr(torch.Tensor, "to", lambda *a, **k: pytest.fail("the launch converted a tensor"))
    feats = comp.launch(packed, layout, preemphasis=preemph)
    monkeypatch.undo()
    assert feats.dtype == torch.float32 and feats.shape == (int(rows[-1]), comp.num_coeffs)
    same = comp.launch(as_float, layout, preemphasis=preemph)
    # (dense banks take another filter walk for float32 samples: tolerance instead of equality there)
    if not torch.equal(feats, same):
        assert_features_close(feats.cpu().numpy(), same.cpu().numpy(), rtol=2e-5, atol=2e-6)
    got = feats.cpu().numpy()
    for b, x in enumerate(sigs):
        y = x.astype(np.float64)
        if preemph:
            y[1:] -= preemph * x[:-1].astype(np.float64)
        tol = dict(rtol=2e-4, atol=2e-5) if preemph else F32
        assert_features_close(got[rows[b] : rows[b + 1]], orc.compute_full(y, p), what=(name, b), **tol)


def test_int16_input_without_a_fused_kernel_converts_first(golden_tables):
    # a transform size the int16 instantiations do not cover (N = L = 400 without padding): same features
    import torch

    cfg = {"name": "stft", "bank": {"name": "fbank", "num_filts": 40}, "frame_length_ms": 25, "frame_shift_ms": 10,
           "pad_to_nearest_power_of_two": False}
    comp = build(cfg)
    rng = np.random.default_rng(6)
    x = torch.from_numpy(rng.integers(-3000, 3000, size=16000).astype(np.int16)).cuda()
    got, _ = comp.compute_packed(x, [0], [16000])
    want, _ = comp.compute_packed(x.to(torch.float32), [0], [16000])
    assert got.dtype == torch.float32 and torch.equal(got, want)


@pytest.mark.parametrize("name", ["c1_kaldi_fbank", "c2_tri_mel40", "v_tri_analytic_nolog", "v_gabor_nopad_mag"])
def test_streaming_random_chunkings_match_reference_call_by_call(name, computers, master_signal):
    # tests/golden/make_golden_stream.py: eight lengths (one sample .. 4000) x random cut points;
    # every compute_chunk / finalize call must return as many frames as the reference's did
    with np.load(os.path.join(GOLDEN, "stream_random.npz")) as z:
        g = {k: z[k] for k in z.files if k.startswith(name + "/")}
    comp = computers[name]
    for case in range(8):
        n = int(g[f"{name}/{case}/n"])
        x = master_signal[50 : 50 + n].astype("f4")
        pieces = np.split(x, g[f"{name}/{case}/cuts"])
        outs = [comp.compute_chunk(p) for p in pieces] + [comp.finalize()]
        assert [len(o) for o in outs] == g[f"{name}/{case}/counts"].tolist(), (name, case)
        assert_features_close(np.concatenate(outs), g[f"{name}/{case}/feats"], what=(name, case), **F32)


def test_plans_are_per_device(computers):
    # a computer keeps one native plan per GPU; the C ABI refuses a plan on another device
    import torch

    comp = computers["c2_tri_mel40"]
    plan0 = comp._native_plan(torch.device("cuda", 0))
    assert comp._native_plan(0) is plan0 and comp._native_plan("cuda:0") is plan0
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU: the cross-device half needs two")
    x = torch.randn(16000, device="cuda:1")
    y1, _ = comp.compute_packed(x, [0], [16000])
    y0, _ = comp.compute_packed(x.to("cuda:0"), [0], [16000])
    assert y1.device.index == 1 and torch.equal(y1.cpu(), y0.cpu())
    from pydrobert_speech_amd import _native

    lib = _native.lib()
    layout = comp.prepare_layout([0], [16000], device="cuda:1")
    out = torch.empty((100, comp.num_coeffs), device="cuda:1")
    meta = layout.d_meta
    with torch.cuda.device(1):
        rc = lib.pds_stft_batch_f32(plan0.handle, x.data_ptr(), meta[0].data_ptr(), meta[1].data_ptr(),
                                    meta[2].data_ptr(), meta[3].data_ptr(), 1, 100, -1, 0.0, out.data_ptr(),
                                    out.stride(0), None)
    assert rc != 0 and b"created on device 0" in lib.pds_last_error()


@pytest.mark.parametrize("dtype", ["f4", "f8"])
def test_frames_beyond_the_fused_sizes_take_the_lds_fft(dtype):
    # 100 ms frames at 48 kHz: L = 4800 -> N = 8192, no fused geometry; the generic path runs its
    # radix-2 FFT in LDS (one frame per workgroup at this size) in the signal's precision
    comp = build({"name": "stft", "bank": {"name": "fbank", "num_filts": 64, "sampling_rate": 48000},
                  "frame_length_ms": 100, "frame_shift_ms": 25, "use_power": True, "include_energy": True})
    assert comp.dft_size == 8192 and comp.kernel_kind == 0
    p = _params_from_computer(comp)
    rng = np.random.default_rng(5)
    sigs = [(3000 * rng.standard_normal(n)).astype(dtype) for n in (0, 2399, 4800, 30011)]
    tol = F32 if dtype == "f4" else dict(rtol=1e-9, atol=1e-9)
    for x, y in zip(sigs, comp.compute_full_batch(sigs)):
        assert y.dtype == x.dtype
        assert_features_close(y, orc.compute_full(x, p), what=(dtype, len(x)), **tol)


@pytest.mark.parametrize("bank", ["gammatone64_48k", "gabor64", "mel40"])
def test_segmented_and_ell_filter_walks_agree(bank, monkeypatch):
    # the fused kernel has two filter phases (ELL rows per frame; equal-length segments shared by a
    # wave's four frames, chosen for dense banks): both forced in turn, against the oracle and
    # against each other, on ragged batches with partial last chunks
    cfg = {
        "gammatone64_48k": {"name": "stft", "bank": {"name": "gammatone", "scaling_function": "mel", "num_filts": 64,
                                                     "sampling_rate": 48000}, "frame_length_ms": 20, "use_power": True},
        "gabor64": {"name": "stft", "bank": {"name": "gabor", "scaling_function": "mel", "num_filts": 64},
                    "frame_length_ms": 25, "use_power": True, "include_energy": True},
        "mel40": {"name": "stft", "bank": {"name": "tri", "scaling_function": "mel", "num_filts": 40},
                  "frame_length_ms": 25, "use_power": False, "use_log": False},
    }[bank]
    rng = np.random.default_rng(11)
    outs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("PDS_STFT_SEGMENTED", mode)  # read when the plan is created
        comp = build(cfg)
        S = comp.frame_shift
        sigs = [(3000 * rng.standard_normal(n)).astype("f4") for n in (41 * S + 7, 3 * S, 0, 18 * S + 1, 2 * S - 1)]
        if mode == "0":
            keep = sigs
        outs[mode] = comp.compute_full_batch(keep)
        p = _params_from_computer(comp)
        for x, y in zip(keep, outs[mode]):
            assert_features_close(y, orc.compute_full(x, p), what=(bank, mode, len(x)), **F32)
    for a, b in zip(outs["0"], outs["1"]):
        assert a.shape == b.shape
        np.testing.assert_allclose(a, b, rtol=2e-5, atol=2e-5 * (np.abs(a).max() if a.size else 1))


@pytest.mark.parametrize("switch,value", [("PDS_STFT_FRONT", "mfma"), ("PDS_STFT_WALK", "ell"), ("PDS_STFT_WALK", "seg"),
                                          ("PDS_STFT_WALK", "rseg"), ("PDS_STFT_WALK", "mseg"),
                                          ("PDS_N1024_GEOM", "32x32")])
@pytest.mark.parametrize("name", ["c1_readme_fbank", "c2_tri_mel40", "c3_fbank80_energy", "c4_gabor64",
                                  "c5_gammatone64_48k"])
def test_alternative_kernel_forms_match_the_oracle(switch, value, name, golden_meta, golden_tables, master_signal,
                                                   monkeypatch):
    """The forms a plan does not pick by default -- matrix-pipe front end, each of the four filter walks --
    through the same ragged batch (switches are read when the plan is
    created, so the computer is built after the switch is set)"""
    import pydrobert_speech_amd as ps

    if switch in ("PDS_STFT_FRONT", "PDS_N1024_GEOM") and not ps._native.lib().pds_build_experiments():
        pytest.skip("measured-and-rejected form: only in libraries built with make EXTRA=-DPDS_EXPERIMENTS=1")
    monkeypatch.setenv(switch, value)
    comp = alias_factory_subclass_from_arg(FrameComputer, json.loads(json.dumps(golden_meta["configs"][name])))
    p = oracle_params(golden_tables, name)
    lens = [0, 150, comp.frame_length // 2 + 1, 401, 2000, 16000, 4801, 7]
    sigs = [master_signal[33 : 33 + n].astype("f4") for n in lens]
    feats = comp.compute_full_batch(sigs)
    for x, y in zip(sigs, feats):
        assert_features_close(y, orc.compute_full(x, p), **F32)
    # with fused pre-emphasis too (its own instantiations)
    import torch

    offs = np.concatenate([[0], np.cumsum(lens)[:-1]])
    got, rows = comp.compute_packed(torch.from_numpy(np.concatenate(sigs)).cuda(), offs, lens, preemphasis=0.97)
    got = got.cpu().numpy()
    import importlib.util

    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(GOLDEN, "..", "..", "tools", "fuzz_parity.py"))
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    close = fuzz.close  # (knows the float32 floor: pre-emphasised noise empties the lowest bins)

    for b, x in enumerate(sigs):
        y = x.astype("f8")
        y[1:] -= 0.97 * x[:-1].astype("f8")
        ok, msg = close(got[rows[b] : rows[b + 1]], orc.compute_full(y.astype("f4"), p), 2e-4, 2e-5, is_log=True)
        assert ok, (name, switch, value, b, msg)


@pytest.mark.parametrize("flow", ["preemph", "i16", "i16+preemph"])
@pytest.mark.parametrize("name", ["c2_tri_mel40", "c3_fbank80_energy", "c4_gabor64"])
def test_ragged_scheduling_with_preemphasis_and_int16_samples(name, flow, golden_meta, monkeypatch):
    """... for the flows real ragged batches come in: fused pre-emphasis and 16-bit PCM (the row-segment kernels of
    N = 512 / 1024 have the stretch schedule for them; other banks run the round-robin order through the same entry
    points) -- same rows as the plain launch, bit for bit"""
    import torch

    import pydrobert_speech_amd as ps

    comp = alias_factory_subclass_from_arg(FrameComputer, json.loads(json.dumps(golden_meta["configs"][name])))
    rng = np.random.default_rng(6)
    S = comp.frame_shift
    lens = [0, 1, S, 7 * S + 3, 400 * S, 3 * S, 0, 55 * S, 2 * S, 900 * S + 11] + list(rng.integers(0, 60 * S, size=300))
    host = np.clip(np.rint(3000 * rng.standard_normal(int(np.sum(lens)))), -32768, 32767)
    x = torch.from_numpy(host.astype("i2" if flow.startswith("i16") else "f4")).cuda()
    coeff = 0.97 if "preemph" in flow else 0.0
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]])
    layout = comp.prepare_layout(offs, lens, device=x.device)
    assert layout.fill < 0.9
    monkeypatch.setattr(ps.config, "RAGGED_SCHEDULING", True)
    a = torch.full((layout.total_rows, comp.num_coeffs), float("nan"), device="cuda")
    comp.launch(x, layout, out=a, preemphasis=coeff)
    monkeypatch.setattr(ps.config, "RAGGED_SCHEDULING", False)
    b = torch.full_like(a, float("nan"))
    comp.launch(x, layout, out=b, preemphasis=coeff)
    assert bool(torch.isfinite(a).all()) and torch.equal(a, b)


@pytest.mark.parametrize("name", ["c2_tri_mel40", "c4_gabor64", "c5_gammatone64_48k", "c3_fbank80_energy"])
def test_ragged_scheduling_is_bit_identical(name, golden_meta, monkeypatch):
    """pds_stft_batch_ragged_f32 (every wave one contiguous stretch of the chunks that exist) against the plain
    launch (waves dealt (utterance, chunk) pairs, skipping those short utterances do not have): same rows, bit for
    bit, on a batch with empty, one-frame and long utterances"""
    import torch

    import pydrobert_speech_amd as ps

    comp = alias_factory_subclass_from_arg(FrameComputer, json.loads(json.dumps(golden_meta["configs"][name])))
    rng = np.random.default_rng(5)
    S = comp.frame_shift
    lens = [0, 1, S, 7 * S + 3, 400 * S, 3 * S, 0, 55 * S, 2 * S, 900 * S + 11] + list(rng.integers(0, 60 * S, size=300))
    x = torch.from_numpy((3000 * rng.standard_normal(int(np.sum(lens)))).astype("f4")).cuda()
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]])
    layout = comp.prepare_layout(offs, lens, device=x.device)
    assert layout.fill < 0.9
    monkeypatch.setattr(ps.config, "RAGGED_SCHEDULING", True)
    a = torch.full((layout.total_rows, comp.num_coeffs), float("nan"), device="cuda")
    comp.launch(x, layout, out=a)
    assert layout.fill < 0.9  # (the ragged launch ran: its workspace is allocated per launch)
    monkeypatch.setattr(ps.config, "RAGGED_SCHEDULING", False)
    b = torch.full_like(a, float("nan"))
    comp.launch(x, layout, out=b)
    assert bool(torch.isfinite(a).all()) and torch.equal(a, b)
