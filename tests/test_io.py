"""Rows either side of the hot path: signal readers, CMVN statistics files and
Stack on host arrays, against fixtures the reference produced (tests/golden/make_golden_io.py)."""
import io
import os

import numpy as np
import pytest

from pydrobert_speech_amd.post import PostProcessor, Stack, Standardize
from pydrobert_speech_amd.alias import alias_factory_subclass_from_arg
from pydrobert_speech_amd.util import read_signal
from tests.conftest import GOLDEN


@pytest.fixture(scope="module")
def gio():
    with np.load(os.path.join(GOLDEN, "io.npz")) as z:
        return {k: z[k] for k in z.files}


def path(name):
    return os.path.join(GOLDEN, name)


@pytest.mark.parametrize("key,name,kwargs", [
    ("read/mono", "sig_mono.wav", {}),
    ("read/stereo_f8", "sig_stereo.wav", {"dtype": np.float64}),
    ("read/npy", "sig.npy", {}),
    ("read/npz_default", "sig.npz", {}),
    ("read/npz_other_f4", "sig.npz", {"key": "other", "dtype": "f4"}),
    ("read/pt", "sig.pt", {}),
    ("read/raw_f4", "sig.raw", {"dtype": "f4", "force_as": "file"}),
])
def test_read_signal_matches_reference(gio, key, name, kwargs):
    got = read_signal(path(name), **kwargs)
    want = gio[key]
    assert got.dtype == want.dtype and got.shape == want.shape and np.array_equal(got, want)


def test_read_signal_streams_and_errors(gio, tmp_path):
    with open(path("sig.npy"), "rb") as fh:
        assert np.array_equal(read_signal(fh, force_as="npy"), gio["read/npy"])
    with open(path("sig_mono.wav"), "rb") as fh:
        assert np.array_equal(read_signal(fh, force_as="wav"), gio["read/mono"])
    with pytest.raises(ValueError, match="Set force_as"):
        read_signal(io.BytesIO(b""))
    with pytest.raises(ValueError, match="kaldi"):
        read_signal(io.BytesIO(b""), force_as="table")
    with pytest.raises(IOError):
        read_signal(str(tmp_path / "mystery.bin"))  # no catch-all (reference since v0.2.0)
    with pytest.raises(ValueError, match="is not one of"):
        read_signal(path("sig.npy"), force_as="mp9")
    for name in ("ark:foo.ark", "scp,p:foo.scp", "x.hdf5", "x.sph", "gunzip -c x.gz |"):
        with pytest.raises(ImportError):
            read_signal(name)


def test_wave_module_path_without_scipy(gio, monkeypatch):
    # the standard-library reader the reference falls back to (util.py:216-235)
    import builtins

    real = builtins.__import__

    def no_scipy(name, *a, **k):
        if name.startswith("scipy"):
            raise ImportError(name)
        return real(name, *a, **k)

    monkeypatch.setattr(builtins, "__import__", no_scipy)
    assert np.array_equal(read_signal(path("sig_mono.wav")), gio["read/mono"])
    got = read_signal(path("sig_stereo.wav"), dtype=np.float64)
    assert got.shape == (600, 2) and np.array_equal(got, gio["read/stereo_f8"])


# ---- Stack on host arrays --------------------------------------------------------------


@pytest.mark.parametrize("key,src,kwargs,axis", [
    ("stack/out2/nv3", "stack/in2", dict(num_vectors=3), 1),
    ("stack/out2/nv3_edge", "stack/in2", dict(num_vectors=3, pad_mode="edge"), 1),
    ("stack/out2/nv4_const", "stack/in2", dict(num_vectors=4, pad_mode="constant"), -1),
    ("stack/out2/nv1", "stack/in2", dict(num_vectors=1), 1),
    ("stack/out2/nv12", "stack/in2", dict(num_vectors=12), 1),
    ("stack/out3/nv3_t1_a2", "stack/in3", dict(num_vectors=3, time_axis=1), 2),
    ("stack/out3/nv4_t1_a0_reflect", "stack/in3", dict(num_vectors=4, time_axis=1, pad_mode="reflect"), 0),
    ("stack/out3/nv2_tm1_a1", "stack/in3", dict(num_vectors=2, time_axis=-1), 1),
])
def test_stack_matches_reference(gio, key, src, kwargs, axis):
    x = gio[src]
    keep = x.copy()
    got = alias_factory_subclass_from_arg(PostProcessor, dict(name="stack", **kwargs)).apply(x, axis=axis)
    want = gio[key]
    assert got.shape == want.shape and got.dtype == want.dtype and np.array_equal(got, want)
    assert np.array_equal(x, keep)  # input untouched


def test_stack_transposed_time_axis_and_errors(gio):
    x = gio["stack/in2"]
    got = Stack(2, time_axis=1).apply(x.T.copy(), axis=0)
    assert np.array_equal(got, gio["stack/out2/nv2_t1"])
    with pytest.raises(ValueError, match="positive"):
        Stack(0)
    with pytest.raises(RuntimeError, match="same"):
        Stack(2).apply(x, axis=0)


# ---- CMVN statistics files ----------------------------------------------------------------


@pytest.mark.parametrize("name,kwargs", [
    ("cmvn_stats.npy", {}),
    ("cmvn_stats.f64", {"force_as": "file"}),
    ("cmvn_stats.f32", {"force_as": "file"}),  # float width detected from the contents
])
def test_standardize_reads_the_references_statistics_files(gio, name, kwargs):
    st = Standardize(path(name), **kwargs)
    assert st.have_stats
    assert st._stats.shape == gio["cmvn_file/stats"].shape
    tol = 1e-6 if name.endswith("f32") else 0
    assert np.allclose(st._stats, gio["cmvn_file/stats"], rtol=tol, atol=0)


def test_standardize_save_round_trips(gio, tmp_path):
    st = Standardize(path("cmvn_stats.npy"))
    for name, kwargs in (("a.npy", {}), ("b.raw", {"force_as": "file"})):
        st.save(str(tmp_path / name))
        assert np.array_equal(Standardize(str(tmp_path / name), **kwargs)._stats, st._stats)
    arch = str(tmp_path / "c.npz")
    st.save(arch)
    st.save(arch, key="second", compress=True)
    with np.load(arch) as z:
        assert sorted(z.files) == ["arr_0", "second"] and np.array_equal(z["second"], st._stats)
    assert np.array_equal(Standardize(arch, key="second")._stats, st._stats)
    with pytest.raises(ValueError, match="No stats"):
        Standardize().save(str(tmp_path / "d.npy"))
    with pytest.raises(IOError):
        Standardize(str(tmp_path / "missing.npy"))
    with pytest.raises(TypeError):
        Standardize(norm_var=True, force_as="file")
    junk = tmp_path / "junk.raw"
    np.asarray([-1.0, 2.5, 3.0]).tofile(str(junk))
    with pytest.raises(IOError):
        Standardize(str(junk), force_as="file")


# ---- the two numeric helpers util exports beside the readers (reference tests/test_util.py:11-53) ----


@pytest.mark.parametrize("shift", [0, 1, -3, 100])
@pytest.mark.parametrize("dft_size", [1, 2, 51, 256])
@pytest.mark.parametrize("start_idx", [0, 1, -1])
@pytest.mark.parametrize("copy", [True, False])
def test_circshift_fourier_is_a_circular_shift_in_time(shift, dft_size, start_idx, copy):
    from pydrobert_speech_amd.util import circshift_fourier

    rng = np.random.default_rng(dft_size * 7 + start_idx)
    start_idx %= dft_size
    zeros = int(rng.integers(dft_size))
    X = 10 * rng.random(dft_size - zeros) + 10j * rng.random(dft_size - zeros)
    given = X.copy()
    Xs = circshift_fourier(given, shift, start_idx=start_idx, dft_size=dft_size, copy=copy)
    assert Xs.dtype == np.complex128 and (Xs is given) == (not copy)
    if copy:
        assert np.array_equal(given, X)
    full = np.roll(np.pad(X, (0, zeros)), start_idx)
    full_s = np.roll(np.pad(Xs, (0, zeros)), start_idx)
    assert np.allclose(np.roll(np.fft.ifft(full), shift), np.fft.ifft(full_s))
    # default transform size: the response ends at the last bin
    assert np.allclose(circshift_fourier(X, shift, start_idx=start_idx),
                       circshift_fourier(X, shift, start_idx=start_idx, dft_size=len(X) + start_idx))
    # real input is promoted rather than modified
    real = np.ones(dft_size)
    assert circshift_fourier(real, shift, dft_size=dft_size, copy=False).dtype == np.complex128
    assert np.array_equal(real, np.ones(dft_size))


@pytest.mark.parametrize("mu", [0, -1, 100])
@pytest.mark.parametrize("std", [0.1, 1, 10])
@pytest.mark.parametrize("with_scipy", [True, False])
def test_gauss_quant_inverts_the_gaussian_cdf(mu, std, with_scipy, monkeypatch):
    import math
    import sys

    from pydrobert_speech_amd import util

    if not with_scipy:
        monkeypatch.setitem(sys.modules, "scipy.special", None)  # the import inside fails
    xs = np.linspace(-4.5, 4.5, 181) * std + mu
    for x in xs:
        p = 0.5 * math.erfc(-(x - mu) / std / math.sqrt(2))
        assert np.isclose(util.gauss_quant(p, mu=mu, std=std), x, rtol=0, atol=1e-6 * std + 1e-9 * abs(mu))
    assert util.gauss_quant(0.5, mu=mu, std=std) == pytest.approx(mu)
    assert util.gauss_quant(0.0) == -np.inf and util.gauss_quant(1.0) == np.inf
