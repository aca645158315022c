"""Host replay of the twiddle regeneration of the prefetch instantiations against float64 twiddles (CPU)."""
import os
import shutil
import subprocess

import pytest

from tests.conftest import ROOT


def test_twiddle_chain(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    exe = str(tmp_path / "test_twiddle_chain")
    src = os.path.join(ROOT, "tests", "csrc", "test_twiddle_chain.cpp")
    # (-ffp-contract=off: the host build must not fuse what the device code rounds separately)
    subprocess.run([hipcc, "-O2", "-std=c++17", "-ffp-contract=off", "--offload-arch=gfx950", "-x", "hip", src, "-o", exe],
                   check=True, capture_output=True)
    res = subprocess.run([exe], capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "worst twiddle error" in res.stdout
