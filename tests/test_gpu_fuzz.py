"""A short run of the randomised differential test (tools/fuzz_parity.py): random frame computers,
ragged batches, fused and direct-DFT kernels and the short-integration kernels against the oracles."""
import subprocess
import sys
import os

import pytest

from tests.conftest import ROOT

pytestmark = pytest.mark.gpu


def test_random_configurations_match_the_oracles():
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "120", "7"],
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-2000:]
    assert "failures: 0" in res.stdout
