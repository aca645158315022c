import sys, os, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import pydrobert_speech_amd as ps
from pydrobert_speech_amd.alias import alias_factory_subclass_from_arg
cfg = {"name": "stft", "bank": {"name": "tri", "scaling_function": "mel", "num_filts": 40}, "frame_length_ms": 25, "frame_shift_ms": 10, "window_function": "hanning", "use_power": True}
comp = alias_factory_subclass_from_arg(ps.compute.FrameComputer, cfg)
B, n = 256, 160000
x = torch.randn(B * n, device="cuda", dtype=torch.float64) * 3000
layout = comp.prepare_layout(np.arange(B) * n, np.full(B, n))
out = comp.launch(x, layout)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): comp.launch(x, layout, out=out)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 20
print("float64 kernel (LDS FFT for this power-of-two size): %.2f ms per %d frames -> %.1f M frames/s" % (dt * 1e3, layout.total_rows, layout.total_rows / dt / 1e6))
xf = x.to(torch.float32)
outf = comp.launch(xf, layout, generic=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): comp.launch(xf, layout, out=outf, generic=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 20
print("float32 generic kernel (same): %.2f ms -> %.1f M frames/s" % (dt * 1e3, layout.total_rows / dt / 1e6))
