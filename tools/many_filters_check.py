"""GPU-box check: banks far larger than the usual (hundreds of filters: many ELL slots, tables beyond
LDS) through the fused kernel, against the oracle."""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np

import pydrobert_speech_amd as ps
from oracle import stft_oracle as orc
from tools.fuzz_parity import close  # tolerance + the float32 round-off floor clause for weak bins
from pydrobert_speech_amd.alias import alias_factory_subclass_from_arg

rng = np.random.default_rng(0)
bad = 0
for bank, F, rate, ms in [("fbank", 200, 16000, 25), ("tri", 500, 16000, 25), ("gabor", 256, 16000, 25),
                          ("gammatone", 128, 48000, 20), ("fbank", 1000, 48000, 25), ("tri", 65, 8000, 25)]:
    cfg = {"name": "stft", "bank": {"name": bank, "num_filts": F, "sampling_rate": rate}, "frame_length_ms": ms,
           "use_power": True, "include_energy": bool(F % 2)}
    if bank != "fbank":
        cfg["bank"]["scaling_function"] = "mel"
    comp = alias_factory_subclass_from_arg(ps.compute.FrameComputer, cfg)
    p = orc.StftParams(
        frame_length=comp.frame_length, frame_shift=comp.frame_shift, dft_size=comp.dft_size,
        window=np.asarray(comp._window), starts=list(comp._filt_start_idxs),
        taps=[np.asarray(t) for t in comp._truncated_filts], is_real=comp.bank.is_real,
        centered=comp.frame_style == "centered", kaldi_shift=comp.kaldi_shift,
        include_energy=comp.includes_energy, use_power=bool(comp._power), use_log=bool(comp._log))
    sigs = [(3000 * rng.standard_normal(n)).astype("f4") for n in (0, 300, 16000, 5001)]
    got = comp.compute_full_batch(sigs)
    worst, ok = 0.0, True
    for x, y in zip(sigs, got):
        want = orc.compute_full(x, p)
        assert y.shape == want.shape, (y.shape, want.shape)
        if want.size:
            worst = max(worst, float(np.max(np.abs(y - want) / (1e-5 + 1e-4 * np.abs(want)))))
        fine, msg = close(y, want, 1e-4, 1e-5, is_log=True)
        ok &= fine
    bad += not ok
    print("%-10s F=%4d N=%4d kernel=%s nnz=%6d  worst error / tolerance %.3f  %s" % (
        bank, F, comp.dft_size, comp.kernel_kind, len(comp._col), worst, "ok" if ok else "FAIL")
          + ("" if worst <= 1 else "  (beyond 1e-4 only in bins at the float32 round-off floor of their frame)"))
sys.exit(1 if bad else 0)
