import sys, time, json
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pydrobert_speech_amd as ps
from pydrobert_speech_amd.alias import alias_factory_subclass_from_arg
cfg = {"name": "stft", "bank": {"name": "tri", "scaling_function": "mel", "num_filts": 40}, "frame_length_ms": 25, "frame_shift_ms": 10, "window_function": "hanning", "use_power": True}
comp = alias_factory_subclass_from_arg(ps.compute.FrameComputer, cfg)
B, n = 1024, 160000
rng = np.random.default_rng(0)
host = (3000 * rng.standard_normal(B * n)).astype(np.float32)
sigs = [host[i * n:(i + 1) * n] for i in range(B)]
comp.compute_full_batch(sigs[:4])
for trial in range(3):
    t0 = time.perf_counter(); feats = comp.compute_full_batch(sigs); t1 = time.perf_counter()
    print("list-of-numpy API: %.1f ms -> %.1f M frames/s" % (1e3 * (t1 - t0), B * 1000 / (t1 - t0) / 1e6))
pinned = torch.from_numpy(host).pin_memory()
layout = comp.prepare_layout(np.arange(B) * n, np.full(B, n))
out_host = torch.empty((layout.total_rows, comp.num_coeffs), dtype=torch.float32).pin_memory()
for trial in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    d = pinned.to("cuda", non_blocking=True)
    out = comp.launch(d, layout)
    out_host.copy_(out, non_blocking=True)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print("pinned H2D + kernel + D2H: %.1f ms -> %.1f M frames/s" % (1e3 * (t1 - t0), B * 1000 / (t1 - t0) / 1e6))
