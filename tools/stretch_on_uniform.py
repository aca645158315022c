#!/usr/bin/env python3
"""Stretch scheduling (pds_stft_batch_ragged_f32: every wave walks one contiguous run of chunks) against the round-robin
launch (pds_stft_batch_f32) on the UNIFORM headline batch: is there a reason to keep both schedules?"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
import pydrobert_speech_amd as ps
from pydrobert_speech_amd import _native
from pydrobert_speech_amd.alias import alias_factory_subclass_from_arg

wl = sys.argv[1] if len(sys.argv) > 1 else bench.DEFAULT_WORKLOAD
cfg, n, B, _ = bench.WORKLOADS[wl]
comp = alias_factory_subclass_from_arg(ps.compute.FrameComputer, cfg)
lib = _native.lib()
plan = comp._native_plan()
C = comp.num_coeffs
layout = comp.prepare_layout(np.arange(B) * n, np.full(B, n))
meta = layout.d_meta
x = torch.randn(B * n, device="cuda").mul_(3000.0)
out = torch.empty((layout.total_rows, C), dtype=torch.float32, device="cuda")
out2 = torch.empty_like(out)
work = torch.empty(B + 1, dtype=torch.int64, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
mx = int(layout.nframes.max())


def plain():
    rc = lib.pds_stft_batch_f32(plan.handle, x.data_ptr(), meta[0].data_ptr(), meta[1].data_ptr(), meta[2].data_ptr(),
                                meta[3].data_ptr(), B, mx, -1, 0.0, out.data_ptr(), C, stream)
    assert rc == 0


def stretch():
    rc = lib.pds_stft_batch_ragged_f32(plan.handle, x.data_ptr(), meta[0].data_ptr(), meta[1].data_ptr(), meta[2].data_ptr(),
                                       meta[3].data_ptr(), B, mx, -1, 0.0, work.data_ptr(), out2.data_ptr(), C, stream)
    assert rc == 0


for _ in range(400):
    plain()
torch.cuda.synchronize()
for rep in range(3):
    for name, fn in (("round-robin", plain), ("stretch", stretch)):
        for _ in range(100):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(300):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print(f"{wl} {name:12s} {e0.elapsed_time(e1) / 300:.4f} ms per launch")
print("equal:", bool(torch.equal(out, out2)))
