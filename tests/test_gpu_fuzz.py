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
    # the float32-floor exemption of tools/fuzz_parity.py::close, counted: a handful of near-empty coefficients of
    # pre-emphasised noise per run -- never more than one element in 10 000, and none further
    # than 50 x the strict tolerance (an element 60 dB below its frame's peak with the frame's float32 round-off)
    import re

    m = re.search(r"float32-floor rule: (\d+) of (\d+) compared elements .* worst error ([0-9.e+-]+) x", res.stdout)
    assert m, res.stdout[-500:]
    exempted, compared, worst = int(m.group(1)), int(m.group(2)), float(m.group(3))
    assert compared > 100000 and exempted * 10000 <= compared and worst <= 50.0, (exempted, compared, worst)
