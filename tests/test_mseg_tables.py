"""Host replay of the matrix-pipe segment walk's tables against the CSR product (CPU)."""
import os
import shutil
import subprocess

import pytest

from tests.conftest import ROOT


def test_mseg_tables(tmp_path):
    cxx = shutil.which("g++")
    if cxx is None:
        pytest.skip("g++ not available")
    exe = str(tmp_path / "test_mseg_tables")
    src = os.path.join(ROOT, "tests", "csrc", "test_mseg_tables.cpp")
    subprocess.run([cxx, "-O2", "-std=c++17", src, "-o", exe, "-lm"], check=True, capture_output=True)
    res = subprocess.run([exe], capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "worst normalised error" in res.stdout
