import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import pydrobert_speech_amd as ps
from pydrobert_speech_amd.alias import alias_factory_subclass_from_arg
cfg = {"name": "stft", "bank": {"name": "tri", "scaling_function": "mel", "num_filts": 40}, "frame_length_ms": 25, "frame_shift_ms": 10, "window_function": "hanning", "use_power": True}
comp = alias_factory_subclass_from_arg(ps.compute.FrameComputer, cfg)
B, n = 1024, 160000
x = torch.randn(B * n, device="cuda") * 3000
layout = comp.prepare_layout(np.arange(B) * n, np.full(B, n))
out = comp.launch(x, layout)
torch.cuda.synchronize()
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(400)]
for a, b in evs:
    a.record(); comp.launch(x, layout, out=out); b.record()
torch.cuda.synchronize()
ms = np.array([a.elapsed_time(b) for a, b in evs])
for lo in range(0, 400, 40):
    print("steps %3d-%3d mean %.4f ms" % (lo, lo + 39, ms[lo:lo + 40].mean()))
