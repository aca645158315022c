"""Host replay of the matrix-pipe front end's tables and data flow against a float64 DFT (CPU)."""
import os
import shutil
import subprocess

import pytest

from tests.conftest import ROOT


def test_mfma_front_tables(tmp_path):
    cxx = shutil.which("g++")
    if cxx is None:
        pytest.skip("g++ not available")
    exe = str(tmp_path / "test_mfma_front")
    src = os.path.join(ROOT, "tests", "csrc", "test_mfma_front.cpp")
    subprocess.run([cxx, "-O2", "-std=c++17", src, "-o", exe, "-lm"], check=True, capture_output=True)
    res = subprocess.run([exe], capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "worst normalised error" in res.stdout
