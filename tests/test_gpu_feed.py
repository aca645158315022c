"""The host feed (pds_feed_*, csrc/feed.hip, pydrobert_speech_amd.feed.HostFeed): batches of host signals through
the staging ring give the features of the packed launch on the same samples -- for every sample format, ragged
batches, more batches than slots, a post-processor on the slot's stream -- and of the oracle."""
import json

import numpy as np
import pytest

from oracle import stft_oracle as orc
from pydrobert_speech_amd.alias import alias_factory_subclass_from_arg
from pydrobert_speech_amd.compute import FrameComputer
from pydrobert_speech_amd.feed import HostFeed
from tests.conftest import assert_features_close, oracle_params

pytestmark = pytest.mark.gpu
F32 = dict(rtol=1e-4, atol=1e-5)


def build(cfg):
    return alias_factory_subclass_from_arg(FrameComputer, json.loads(json.dumps(cfg)))


def batches_of(rng, dtype, count, utts, top):
    out = []
    for _ in range(count):
        lens = rng.integers(0, top, size=utts)
        if np.dtype(dtype) == np.int16:
            out.append([rng.integers(-20000, 20000, size=n).astype(np.int16) for n in lens])
        else:
            out.append([(3000 * rng.standard_normal(n)).astype(dtype) for n in lens])
    return out


@pytest.mark.parametrize("direct", [True, False], ids=["direct", "staged"])
@pytest.mark.parametrize("dtype", [np.float32, np.float64, np.int16], ids=["f32", "f64", "i16"])
def test_feed_matches_the_packed_launch_and_the_oracle(dtype, direct, golden_meta, golden_tables, monkeypatch):
    import torch

    from pydrobert_speech_amd import config

    name = "c2_tri_mel40"
    comp = build(golden_meta["configs"][name])
    p = oracle_params(golden_tables, name)
    rng = np.random.default_rng(11)
    batches = batches_of(rng, dtype, 7, 9, 6000)  # more batches than slots: the ring wraps twice
    batches[3] = []  # (an empty batch passes through)
    batches[4] = [b[:0] for b in batches[4]]  # (... and one of empty utterances)
    monkeypatch.setattr(config, "FLOAT64_ARITHMETIC", "float32")
    with HostFeed(comp, dtype, slot_samples=60000, slot_utts=16, slots=3, copy_threads=3, direct=direct) as feed:
        got = list(feed.run(batches))
    assert len(got) == len(batches)
    for batch, feats in zip(batches, got):
        assert len(feats) == len(batch)
        if not batch:
            continue
        lens = [len(x) for x in batch]
        offs = np.concatenate([[0], np.cumsum(lens)[:-1]])
        packed = torch.from_numpy(np.concatenate(batch)).cuda()
        out32 = torch.empty((sum(comp.num_frames(n) for n in lens), comp.num_coeffs), dtype=torch.float32, device="cuda")
        want, rows = comp.compute_packed(packed, offs, lens, out=out32 if np.dtype(dtype) == np.float64 else None)
        want = want.cpu().numpy()
        for b, x in enumerate(batch):
            assert feats[b].dtype == np.float32
            np.testing.assert_array_equal(feats[b], want[rows[b] : rows[b + 1]])
            assert_features_close(feats[b], orc.compute_full(x.astype(np.float64), p), what=(b, len(x)), **F32)


@pytest.mark.parametrize("direct", [True, False], ids=["direct", "staged"])
def test_feed_with_preemphasis_and_a_post_processor_on_the_slot_stream(direct, golden_meta, golden_tables):
    import torch

    from pydrobert_speech_amd.post import Deltas

    name = "c3_fbank80_energy"
    comp = build(golden_meta["configs"][name])
    deltas = Deltas(2)
    rng = np.random.default_rng(12)
    batches = batches_of(rng, np.float32, 4, 5, 9000)
    C = comp.num_coeffs

    def post(feats, row_offsets):
        out = torch.empty((feats.shape[0], 3 * C), dtype=torch.float32, device=feats.device)
        out[:, :C] = feats
        return deltas.apply_rows(out[:, :C], row_offsets, out=out)

    with HostFeed(comp, np.float32, slot_samples=50000, slot_utts=8, slots=2, feature_cols=3 * C, direct=direct) as feed:
        got = list(feed.run(batches, preemphasis=0.97, post=post))
    for batch, feats in zip(batches, got):
        lens = [len(x) for x in batch]
        offs = np.concatenate([[0], np.cumsum(lens)[:-1]])
        packed = torch.from_numpy(np.concatenate(batch)).cuda()
        layout = comp.prepare_layout(offs, lens, device=packed.device)
        want = comp.launch_with_deltas(packed, layout, deltas, fused=False, preemphasis=0.97).cpu().numpy()
        rows = layout.row_offsets
        for b in range(len(batch)):
            assert feats[b].shape == (rows[b + 1] - rows[b], 3 * C)
            np.testing.assert_array_equal(feats[b], want[rows[b] : rows[b + 1]])


def test_feed_refuses_what_does_not_fit_and_misuse(golden_meta):
    comp = build(golden_meta["configs"]["c2_tri_mel40"])
    with HostFeed(comp, np.float32, slot_samples=1000, slot_utts=2, slots=2) as feed:
        with pytest.raises(ValueError):
            feed.submit([np.zeros(600, np.float32), np.zeros(600, np.float32)])  # too many samples
    with HostFeed(comp, np.float32, slot_samples=1000, slot_utts=2, slots=2) as feed:
        with pytest.raises(ValueError):
            feed.submit([np.zeros(10, np.float32)] * 3)  # too many utterances
    with pytest.raises(TypeError):
        HostFeed(comp, np.int32)


@pytest.mark.parametrize("direct", [True, False], ids=["direct", "staged"])
def test_feed_widens_on_the_device_where_no_kernel_takes_the_format(direct):
    """A transform size without a fused int16-input kernel (N = L = 400), and a dense bank whose table does not
    fit in LDS for the int16 instantiations: the feed widens the samples on the device and takes the float32 launch
    -- the rows of compute_packed on the same samples"""
    import torch

    nopad = {"name": "stft", "bank": {"name": "fbank", "num_filts": 40}, "frame_length_ms": 25,
             "frame_shift_ms": 10, "pad_to_nearest_power_of_two": False}
    dense = {"name": "stft", "bank": {"name": "gammatone", "sampling_rate": 44100, "num_filts": 27, "scaling_function": "bark"},
             "frame_length_ms": 25.0, "frame_shift_ms": 8.0, "use_power": False, "use_log": False, "kaldi_shift": True}
    rng = np.random.default_rng(14)
    for cfg in (nopad, dense):
        comp = build(cfg)
        batch = [rng.integers(-20000, 20000, size=n).astype(np.int16) for n in (5000, 1, 12000, 0, 7000)]
        with HostFeed(comp, np.int16, slot_samples=30000, slot_utts=8, slots=2, direct=direct) as feed:
            got = list(feed.run([batch, batch]))
        lens = [len(x) for x in batch]
        offs = np.concatenate([[0], np.cumsum(lens)[:-1]])
        want, rows = comp.compute_packed(torch.from_numpy(np.concatenate(batch)).cuda(), offs, lens)
        want = want.cpu().numpy()
        for feats in got:
            for b in range(len(batch)):
                np.testing.assert_array_equal(feats[b], want[rows[b] : rows[b + 1]])


def test_compute_full_batch_of_host_signals_goes_through_the_feed(golden_meta, monkeypatch):
    """compute_full_batch of numpy signals: above a few seconds of audio the batch streams through the staging
    ring in slot-sized pieces -- same values as the plain path (config.HOST_FEED = False), bit for bit"""
    from pydrobert_speech_amd import compute, config

    comp = build(golden_meta["configs"]["c2_tri_mel40"])
    rng = np.random.default_rng(13)
    lens = rng.integers(1000, 60000, size=40)
    sigs = [(3000 * rng.standard_normal(n)).astype(np.float32) for n in lens]
    monkeypatch.setattr(compute, "_FEED_MIN_SAMPLES", 100000)
    monkeypatch.setattr(compute, "_FEED_SLOT_SAMPLES", 150000)  # (several pieces: the ring wraps)
    monkeypatch.setattr(compute, "_FEED_SLOT_UTTS", 7)
    got = comp.compute_full_batch(sigs, preemphasis=0.97)
    assert comp._feeds, "the batch did not go through the feed"
    monkeypatch.setattr(config, "HOST_FEED", False)
    want = comp.compute_full_batch(sigs, preemphasis=0.97)
    assert len(got) == len(want) == len(sigs)
    for g, w in zip(got, want):
        assert g.dtype == w.dtype == np.float32
        np.testing.assert_array_equal(g, w)


def test_compute_full_of_one_host_signal_goes_through_a_small_direct_feed(golden_meta, monkeypatch):
    """compute_full(numpy signal): the kernel reads the pinned copy of the signal and writes the pinned features
    itself (no upload / download calls) -- the plain path's values bit for bit, for signals that outgrow the slot too"""
    from pydrobert_speech_amd import config

    comp = build(golden_meta["configs"]["c1_kaldi_fbank"])
    rng = np.random.default_rng(15)
    sigs = [(3000 * rng.standard_normal(n)).astype(np.float32) for n in (16000, 1, 0, 400, 1 << 21, 77777)]
    got = [comp.compute_full(x) for x in sigs]
    assert any(len(k) == 3 for k in comp._feeds), "compute_full did not go through the feed"
    monkeypatch.setattr(config, "HOST_FEED", False)
    for x, g in zip(sigs, got):
        w = comp.compute_full(x)
        assert g.dtype == w.dtype == np.float32 and g.shape == w.shape
        np.testing.assert_array_equal(g, w)


def test_feed_views_of_the_pinned_buffers_and_an_early_stop(golden_meta):
    """run(copy=False): the arrays are views of the slot's pinned buffer, valid until the generator is advanced; a
    consumer that stops early leaves the ring drained and usable"""
    import torch

    comp = build(golden_meta["configs"]["c2_tri_mel40"])
    rng = np.random.default_rng(16)
    batches = batches_of(rng, np.float32, 6, 4, 5000)
    with HostFeed(comp, np.float32, slot_samples=30000, slot_utts=8, slots=3) as feed:
        seen = []
        for k, feats in enumerate(feed.run(batches, copy=False)):
            seen.append([f.copy() for f in feats])  # (the views die with the next iteration)
            if k == 2:
                break  # three batches are still in flight / unsent
        again = list(feed.run(batches[:2]))  # the ring is free again
    for batch, feats in list(zip(batches, seen)) + list(zip(batches[:2], again)):
        lens = [len(x) for x in batch]
        offs = np.concatenate([[0], np.cumsum(lens)[:-1]])
        want, rows = comp.compute_packed(torch.from_numpy(np.concatenate(batch)).cuda(), offs, lens)
        want = want.cpu().numpy()
        for b in range(len(batch)):
            np.testing.assert_array_equal(feats[b], want[rows[b] : rows[b + 1]])


def test_compute_full_from_several_threads_at_once(golden_meta):
    """compute_full touches no state of the computer (reference compute.py:574-607): threads may call it at once -- one
    of them is in the staging ring, the others take the plain path; every result is the single-threaded one"""
    import concurrent.futures

    comp = build(golden_meta["configs"]["c2_tri_mel40"])
    rng = np.random.default_rng(17)
    sigs = [(3000 * rng.standard_normal(n)).astype(np.float32) for n in rng.integers(2000, 40000, size=24)]
    want = [comp.compute_full(x) for x in sigs]
    with concurrent.futures.ThreadPoolExecutor(6) as pool:
        got = list(pool.map(comp.compute_full, sigs * 3))
    for k, g in enumerate(got):
        np.testing.assert_array_equal(g, want[k % len(sigs)])


def test_float64_batches_and_short_integration_batches_go_through_pinned_staging(golden_meta, golden_tables, monkeypatch):
    """_staging.PinnedStaging: the batches the native ring does not serve -- float64 signals in float64 arithmetic,
    the short-integration computer -- are packed into a pinned buffer by a few threads and come back through one.
    Same values as one compute_full per signal (itself checked against the reference's fixtures elsewhere) and as
    the oracle; a batch beyond the buffers' limit is split into pieces; a busy staging falls back to the plain path"""
    from pydrobert_speech_amd import _staging, compute

    name = "c2_tri_mel40"
    comp = build(golden_meta["configs"][name])
    rng = np.random.default_rng(31)
    lens = [0, 1, 5000, 160000, 33333, 48000, 7]
    sigs = [3000 * rng.standard_normal(n) for n in lens]  # float64
    assert sum(lens) * 8 > _staging._MIN_BYTES
    used = []
    real_upload = _staging.PinnedStaging.upload
    monkeypatch.setattr(_staging.PinnedStaging, "upload", lambda self, *a, **k: (used.append(1), real_upload(self, *a, **k))[1])
    got = comp.compute_full_batch(sigs)
    assert used, "the batch did not go through the pinned buffers"
    p = oracle_params(golden_tables, name)
    for x, y in zip(sigs, got):
        assert y.dtype == np.float64
        assert_features_close(y, orc.compute_full(x, p), rtol=1e-9, atol=1e-9, what="float64 batch through pinned staging")
    # (caller-owned results: the pinned buffer is reused by the next batch)
    keep = [y.copy() for y in got]
    comp.compute_full_batch([s[::-1].copy() for s in sigs])
    for y, k in zip(got, keep):
        np.testing.assert_array_equal(y, k)
    # beyond the limit: pieces
    del used[:]
    monkeypatch.setattr(compute, "_STAGING_MAX_BYTES", 1 << 20)  # (the batch holds 1.97 MB: halves, quarters ...)
    pieces = comp.compute_full_batch(sigs)
    assert len(pieces) == len(sigs) and len(used) >= 1
    for y, k in zip(pieces, keep):
        np.testing.assert_array_equal(y, k)
    # a staging in use by another thread: the plain path, same values
    del used[:]
    assert comp._staging.try_acquire(1 << 22)
    try:
        plain = comp.compute_full_batch(sigs[:4])
    finally:
        comp._staging.release()
    assert not used
    for y, k in zip(plain, keep):
        np.testing.assert_array_equal(y, k)
    # the short-integration computer
    si = alias_factory_subclass_from_arg(
        FrameComputer, {"name": "si", "bank": {"name": "gabor", "scaling_function": "mel", "num_filts": 6}})
    sigs4 = [(1000 * rng.standard_normal(n)).astype(np.float32) for n in (90000, 0, 12345, 160000, 3)]
    del used[:]
    got = si.compute_full_batch(sigs4)
    assert used
    for x, y in zip(sigs4, got):
        np.testing.assert_array_equal(y, si.compute_full(x))
