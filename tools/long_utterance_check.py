import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import pydrobert_speech_amd as ps
from pydrobert_speech_amd.alias import alias_factory_subclass_from_arg
cfg = {"name": "stft", "bank": {"name": "tri", "scaling_function": "mel", "num_filts": 40}, "frame_length_ms": 25, "frame_shift_ms": 10, "window_function": "hanning", "use_power": True}
comp = alias_factory_subclass_from_arg(ps.compute.FrameComputer, cfg)
n = 16000 * 3600
x = torch.randn(n, device="cuda") * 3000
t0 = time.perf_counter(); y = comp.compute_full(x); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("one hour of audio:", tuple(y.shape), "%.1f ms" % (dt * 1e3), "finite", bool(torch.isfinite(y).all()))
S, L, pad = comp.frame_shift, comp.frame_length, comp.pad_left
for t in (0, 1, 123456, y.shape[0] - 2, y.shape[0] - 1, 200000):
    # frame t of the long signal = frame k of a slice that starts at frame t - k
    k = min(t, 5)
    lo = (t - k) * S
    seg = x[max(lo - 0, 0): lo + (k + 8) * S].contiguous() if t - k > 0 else x[: (k + 8) * S].contiguous()
    if t - k > 0:
        # slices that do not start at the signal start get reflected on their left end, so compare a
        # frame far enough from the slice's ends: start the slice 10 frames earlier
        lo2 = (t - k - 10) * S
        seg = x[lo2: lo2 + (k + 30) * S].contiguous()
        ys = comp.compute_full(seg)
        ref = ys[10 + k]
    else:
        ref = comp.compute_full(seg)[t]
    if t < y.shape[0] - 2:
        err = float((y[t] - ref).abs().max())
        print("frame", t, "max |diff| vs slice", err)
        assert err < 1e-4
print("ok")
