"""Host build of the in-lane FFT templates checked against a naive float64 DFT (CPU)."""
import os
import shutil
import subprocess

import pytest

from tests.conftest import ROOT


def test_inlane_fft_templates(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    exe = str(tmp_path / "test_fft_inlane")
    src = os.path.join(ROOT, "tests", "csrc", "test_fft_inlane.cpp")
    subprocess.run([hipcc, "-O2", "-std=c++17", "--offload-arch=gfx950", "-x", "hip", src, "-o", exe],
                   check=True, capture_output=True)
    res = subprocess.run([exe], capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "worst normalised error" in res.stdout
