import json
import os
import sys
import warnings

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

# like the reference's suite (tests/conftest.py:16-22): warnings are errors
warnings.simplefilter("error")
warnings.filterwarnings("ignore", category=DeprecationWarning)
warnings.filterwarnings("ignore", category=ImportWarning)
warnings.filterwarnings("ignore", category=ResourceWarning)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a HIP device (run on the MI355X box)")
    config.addinivalue_line("markers", "slow: larger sizes")


def _load(name):
    with np.load(os.path.join(GOLDEN, name)) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden_meta():
    with open(os.path.join(GOLDEN, "configs.json")) as fh:
        return json.load(fh)


@pytest.fixture(scope="session")
def golden_tables():
    return _load("tables.npz")


@pytest.fixture(scope="session")
def golden_stft():
    return _load("stft.npz")


@pytest.fixture(scope="session")
def golden_stream():
    return _load("stream.npz")


@pytest.fixture(scope="session")
def golden_post():
    return _load("post.npz")


@pytest.fixture(scope="session")
def golden_kaldi():
    return _load("kaldi.npz")


@pytest.fixture(scope="session")
def master_signal():
    return _load("signals.npz")["master"]


def oracle_params(tables, name):
    """StftParams of configuration `name` from the golden tables (reference-derived)"""
    from oracle.stft_oracle import StftParams

    dims = tables[f"{name}/dims"]
    L, S, N, _ncoef, is_real, centered, kaldi, energy, power, log = (int(v) for v in dims)
    offs = tables[f"{name}/tap_offsets"]
    taps = tables[f"{name}/taps"]
    return StftParams(
        frame_length=L, frame_shift=S, dft_size=N, window=tables[f"{name}/window"],
        starts=[int(s) for s in tables[f"{name}/starts"]],
        taps=[taps[offs[i] : offs[i + 1]] for i in range(len(offs) - 1)],
        is_real=bool(is_real), centered=bool(centered), kaldi_shift=bool(kaldi),
        include_energy=bool(energy), use_power=bool(power), use_log=bool(log),
    )


def config_names():
    with open(os.path.join(GOLDEN, "configs.json")) as fh:
        return sorted(json.load(fh)["configs"])


def assert_features_close(got, want, rtol, atol, what=""):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    if got.size == 0:
        return
    nan_got, nan_want = np.isnan(got), np.isnan(want)
    assert (nan_got == nan_want).all(), (what, "NaN positions differ", np.argwhere(nan_got != nan_want)[:5].tolist())
    err = np.abs(got - want)
    tol = atol + rtol * np.abs(want)
    bad = ~(err <= tol) & ~nan_want  # (a NaN compares false both ways: it must not pass as "close")
    assert not bad.any(), (
        what, int(bad.sum()), float(err.max()), np.argwhere(bad)[:5].tolist(),
        got[bad][:5].tolist(), want[bad][:5].tolist(),
    )


@pytest.fixture(scope="session")
def golden_pre():
    return _load("pre.npz")
