"""Register budgets of the built library, read from its gfx950 code objects (tools/resource_table.py; CPU only).

A spilled kernel still computes the right values, so nothing else in the suite notices one; round 2 shipped float64
-sample instantiations with up to 79 spilled registers.  The kernels the BASELINE.json workloads and the reference
drivers' flows launch must have no scratch at all; the rest of the matrix is bounded."""
import importlib.util
import os

import pytest

from tests.conftest import ROOT

LIB = os.path.join(ROOT, "pydrobert-speech_amd", "csrc", "libpds_amd.so")


def _table():
    spec = importlib.util.spec_from_file_location("resource_table", os.path.join(ROOT, "tools", "resource_table.py"))
    rt = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rt)
    if not os.path.exists(LIB) or not os.path.exists(os.path.join(rt.LLVM, "llvm-readelf")):
        pytest.skip("library or llvm-readelf not available")
    rows, text = [], 0
    for elf in rt.code_objects(LIB):
        ks, t = rt.kernels_of(elf)
        rows += ks
        text += t
    return {rt.short(n): k for n, k in rows}, text


def _key(n1, n2, rows, maxw, minw, pre="F", seg=0, rsg="T", tin="f", tout="f", dlt=0, str_="F"):
    return (f"stft_wave<N1={n1} N2={n2} ROWS={rows} MAXW={maxw} MINW={minw} ELL_LDS=T PRE={pre} SEG={seg} MF=0 RSG={rsg} "
            f"TIN={tin} TOUT={tout} DLT={dlt} STR={str_} PF=F>")


# the kernels of BASELINE.json configs[1..4], of their float64-sample / pre-emphasis / ragged flows and of the
# one-launch statics + deltas: headline geometry 32 x 16 x 25 rows, N = 1024 geometry 64 x 16 x 60 rows
DEFAULT_PATH = [
    _key(32, 16, 25, 16, 4),                                  # configs[1]: 40 mel filters, row-segment walk
    _key(32, 16, 25, 16, 4, pre="T"),                         # ... with fused pre-emphasis
    _key(32, 16, 25, 16, 4, str_="T"),                        # ... ragged batches
    _key(32, 16, 25, 16, 4, str_="T", pre="T"),               # ... with fused pre-emphasis / of 16-bit PCM
    _key(32, 16, 25, 16, 4, str_="T", tin="s"),
    _key(32, 16, 25, 16, 4, str_="T", pre="T", tin="s"),
    _key(32, 16, 25, 16, 4, tin="d"),                         # ... float64 samples
    _key(32, 16, 25, 16, 4, tin="d", tout="d"),
    _key(32, 16, 25, 16, 4, pre="T", tin="d"),                # the reference drivers' flow: float64 audio, pre-emphasis
    _key(32, 16, 25, 16, 4, tin="s"),                         # ... int16 PCM
    _key(32, 16, 25, 16, 4, pre="T", tin="s"),
    _key(32, 16, 25, 12, 3, dlt=2),                           # configs[2]: statics + deltas in one launch
    _key(32, 16, 25, 12, 3, dlt=2, pre="T"),
    _key(32, 16, 25, 12, 3, dlt=2, tin="d"),
    _key(32, 16, 25, 12, 3, dlt=2, pre="T", tin="d"),
    _key(32, 16, 25, 12, 3, dlt=2, tin="s"),                  # ... 16-bit PCM -> (pre-emphasis) -> statics + deltas
    _key(32, 16, 25, 12, 3, dlt=2, pre="T", tin="s"),
    _key(32, 16, 25, 16, 4, seg=1, rsg="F"),                  # configs[3]: Gabor-64, segmented walk
    _key(64, 16, 60, 12, 3, seg=2, rsg="F"),                  # configs[4]: gammatone-64 at 48 kHz, matrix-pipe segments, 3 waves per SIMD
    _key(64, 16, 60, 12, 3),                                  # 20 ms frames at 48 kHz, mel bank
    _key(64, 16, 60, 8, 2, tin="d"),
    _key(64, 16, 60, 8, 2, pre="T", tin="d"),
    _key(64, 16, 60, 12, 3, tin="s"),
    _key(64, 32, 38, 12, 3, rsg="F"),                         # 25 ms frames at 48 kHz (N = 2048, lean form)
]


def test_default_path_kernels_have_no_scratch():
    table, _ = _table()
    missing = [k for k in DEFAULT_PATH if k not in table]
    assert not missing, missing
    spilled = {k: table[k].get("private_segment_fixed_size", 0) for k in DEFAULT_PATH
               if table[k].get("private_segment_fixed_size", 0) or table[k].get("vgpr_spill_count", 0)}
    assert not spilled, spilled


def test_instantiation_matrix_stays_bounded():
    """ product build: fewer than 290 instantiations of the fused kernel (228 without the 58
    int16-sample ones that came after that item), device code below 4.3 MB (3.5 MB without them), and no kernel
    anywhere in the library with more than 40 spilled registers"""
    table, text = _table()
    stft = [k for k in table if k.startswith("stft_wave<")]
    assert 0 < len(stft) < 290, len(stft)
    assert text < 4.3e6, text
    int16 = [k for k in stft if "TIN=s" in k]
    assert 0 < len(int16) <= 58 and len(stft) - len(int16) < 232, (len(int16), len(stft))
    worst = max(table.values(), key=lambda k: k.get("vgpr_spill_count", 0))
    assert worst.get("vgpr_spill_count", 0) <= 40, worst


def test_short_integration_fft_kernels_fit_two_waves_per_simd_without_scratch():
    """csrc/si_fft.hip runs two waves per SIMD (256 registers each); its rework of round 3 started from instantiations
    that spilled 70-110 registers through hoisted per-row addresses -- none may come back"""
    table, _ = _table()
    si = {n: k for n, k in table.items() if "si_fft_kernel" in n}
    assert len(si) == 8, sorted(si)  # {1024-, 2048-point form} x {3, 5, 8, 16 window factors per lane and half}
    for name, k in si.items():
        assert k.get("private_segment_fixed_size", 0) == 0 and k.get("vgpr_spill_count", 0) == 0, (name, k)
        assert k.get("vgpr_count", 0) + k.get("agpr_count", 0) <= 256, (name, k)
