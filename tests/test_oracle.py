"""The oracle (oracle/stft_oracle.py) against fixtures produced by the reference itself.

CPU only.  This is what "parity pinned" rests on: every golden file was written by
tests/golden/make_golden.py importing the reference in the authoring container.
"""
import numpy as np
import pytest

from oracle import stft_oracle as orc
from tests.conftest import assert_features_close, config_names, oracle_params

CONFIGS = config_names()


@pytest.mark.parametrize("name", CONFIGS)
def test_vectorised_oracle_matches_reference_outputs(name, golden_meta, golden_tables, golden_stft, master_signal):
    p = oracle_params(golden_tables, name)
    for n in golden_meta["lengths"][name]:
        for dt, rtol, atol in (("f8", 1e-10, 1e-10), ("f4", 2e-6, 2e-6)):
            x = master_signal[:n].astype(dt)
            want = golden_stft[f"{name}/{n}/{dt}"]
            got = orc.compute_full(x, p)
            assert got.dtype == want.dtype
            assert_features_close(got, want, rtol, atol, (name, n, dt))


@pytest.mark.parametrize("name", CONFIGS)
def test_literal_walk_matches_reference_outputs(name, golden_meta, golden_tables, golden_stft, master_signal):
    # the slow, control-flow-faithful form on the shorter signals
    p = oracle_params(golden_tables, name)
    for n in golden_meta["lengths"][name][:6]:
        x = master_signal[:n].astype("f8")
        got = orc.compute_full_walk(x, p)
        assert_features_close(got, golden_stft[f"{name}/{n}/f8"], 1e-10, 1e-10, (name, n))


def test_known_answer_kaldi_features(golden_tables, golden_kaldi):
    # the reference's own golden vector (tests/test_compute.py:190-208): fbank.json on
    # noise.pkl equals Kaldi's features after undoing the window normalisation and the x2
    p = oracle_params(golden_tables, "c1_kaldi_fbank")
    feats = orc.compute_full(golden_kaldi["noise"], p).astype(np.float64)
    feats += 2 * np.log(0.5 * (p.frame_length - 1))
    feats -= np.log(2)
    assert feats.shape == golden_kaldi["kaldi_feats"].shape == (13, 40)
    assert np.allclose(feats, golden_kaldi["kaldi_feats"])


def test_known_answer_kaldi_filters(golden_tables, golden_kaldi):
    # tests/test_filters.py:211-223 of the reference: squared Fbank taps == Kaldi mel bins
    p = oracle_params(golden_tables, "c1_kaldi_fbank")
    offs = np.concatenate([[0], np.cumsum(golden_kaldi["filt_lens"])])
    for f in range(40):
        kaldi = golden_kaldi["filt_vals"][offs[f] : offs[f + 1]]
        assert p.starts[f] == golden_kaldi["filt_offsets"][f]
        mine = p.taps[f] ** 2
        assert np.allclose(mine[: len(kaldi)], kaldi, atol=1e-5)
        assert np.allclose(mine[len(kaldi) :], 0.0)


def test_walk_bins_examples():
    # N = 512: half = 257, the downward segment starts on the Nyquist bin again
    b = orc.walk_bins(250, 12, 512)
    assert b.tolist() == [250, 251, 252, 253, 254, 255, 256, 256, 255, 254, 253, 252]
    # N = 400 (half = 201, odd): same quirk; N = 6 (half = 4, even): proper conjugate walk
    assert orc.walk_bins(2, 5, 6).tolist() == [2, 3, 2, 1, 0]
    # a filter that starts beyond the Nyquist bin
    assert orc.walk_bins(300, 3, 512).tolist() == [256 - 43, 256 - 44, 256 - 45]
    # wraps all the way round to bin 0 again: period is half + (half - 1) = 513
    assert orc.walk_bins(510, 5, 512).tolist() == [3, 2, 1, 0, 1]


def test_reflect_indices_is_numpy_symmetric_pad():
    x = np.arange(7.0)
    for left, right in ((0, 0), (3, 2), (7, 7), (20, 16)):
        want = np.pad(x, (left, right), "symmetric")
        got = x[orc.reflect_indices(np.arange(-left, len(x) + right), len(x))]
        assert np.array_equal(got, want)


def test_deltas_oracle(golden_post):
    for T in (1, 3, 50):
        x = golden_post[f"deltas/in/T{T}"]
        for nd in (1, 2):
            for W in (2, 3):
                want = golden_post[f"deltas/out/T{T}/n{nd}/w{W}"]
                got = orc.deltas(x, axis=0, num_deltas=nd, context_window=W, target_axis=1)
                assert got.dtype == want.dtype and np.allclose(got, want, rtol=1e-6, atol=1e-6)
    x = golden_post["deltas/in/nd3"]
    got = orc.deltas(x, axis=1, num_deltas=2, target_axis=0, concatenate=False)
    assert np.allclose(got, golden_post["deltas/out/nd3/axis1_stack0"], rtol=1e-12, atol=1e-12)
    got = orc.deltas(x, axis=2, num_deltas=1, target_axis=1)
    assert np.allclose(got, golden_post["deltas/out/nd3/axis2_cat1"], rtol=1e-12, atol=1e-12)


def test_cmvn_oracle(golden_post):
    x = golden_post["cmvn/in"]
    assert np.allclose(orc.cmvn_local(x, -1), golden_post["cmvn/out/local"], rtol=1e-12, atol=1e-12)
    assert np.allclose(orc.cmvn_local(x, 1, norm_var=False), golden_post["cmvn/out/local_novar"], rtol=1e-12, atol=1e-12)
    assert np.allclose(orc.cmvn_local(x.T, 0), golden_post["cmvn/out/local_axis0"], rtol=1e-12, atol=1e-12)
    stats = orc.accumulate_stats([x[:20], x[20:45], x[45]])
    assert np.allclose(stats, golden_post["cmvn/stats"], rtol=1e-13)
    assert np.allclose(orc.cmvn_local(x, -1, stats=stats), golden_post["cmvn/out/global"], rtol=1e-12, atol=1e-12)
    x3 = golden_post["cmvn/in3"]
    assert np.allclose(orc.cmvn_local(x3, 1), golden_post["cmvn/out/in3_axis1"], rtol=1e-12, atol=1e-12)


def test_preemphasis_oracle(golden_pre, golden_tables, master_signal):
    for dt in ("f4", "f8"):
        x = master_signal[500:4500].astype(dt)
        pe = orc.preemphasize(x, 0.97)
        assert pe.dtype == x.dtype and np.array_equal(pe, golden_pre[f"preemph/out/{dt}"])
        p = oracle_params(golden_tables, "c1_readme_fbank")
        tol = dict(rtol=1e-10, atol=1e-10) if dt == "f8" else dict(rtol=2e-6, atol=2e-6)
        assert_features_close(orc.compute_full(pe, p), golden_pre[f"preemph/stft/{dt}"], **tol)
    x2 = master_signal[:600].reshape(3, 200)
    assert np.array_equal(orc.preemphasize(x2, 0.5), golden_pre["preemph/out/2d_last"])
    assert np.array_equal(orc.preemphasize(x2.T, 0.5).T, golden_pre["preemph/out/2d_axis0"])
