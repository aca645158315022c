#!/usr/bin/env python3
"""Does the headline kernel's time depend on the DATA?  Same launch on Gaussian noise, on a constant and on zeros,
interleaved; the instruction stream is identical, so a difference is the clock the chip holds under the load's power
(MI355X_MICROARCH.md, DVFS give-back).  python tools/data_dependence.py [workload]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
import pydrobert_speech_amd as ps
from pydrobert_speech_amd.alias import alias_factory_subclass_from_arg

wl = sys.argv[1] if len(sys.argv) > 1 else bench.DEFAULT_WORKLOAD
cfg, n, B, post = bench.WORKLOADS[wl]
comp = alias_factory_subclass_from_arg(ps.compute.FrameComputer, cfg)
dev = torch.device("cuda", 0)
lengths = np.full(B, n, dtype=np.int64)
offsets = np.concatenate([[0], np.cumsum(lengths)[:-1]]).astype(np.int64)
total = int(lengths.sum())
sigs = {
    "gaussian x 3000": torch.randn(total, device=dev).mul_(3000.0),
    "constant 1000": torch.full((total,), 1000.0, device=dev),
    "zeros": torch.zeros(total, device=dev),
    "gaussian x 1e-3": torch.randn(total, device=dev).mul_(1e-3),
}
layout = comp.prepare_layout(offsets, lengths, device=dev)
out = torch.empty((layout.total_rows, comp.num_coeffs), dtype=torch.float32, device=dev)
for _ in range(400):
    comp.launch(sigs["gaussian x 3000"], layout, out=out)
torch.cuda.synchronize()
for rep in range(3):
    for name, x in sigs.items():
        for _ in range(100):
            comp.launch(x, layout, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            comp.launch(x, layout, out=out)
        e1.record()
        torch.cuda.synchronize()
        print(f"{wl} {name:18s} {e0.elapsed_time(e1) / 200:.4f} ms per launch")
