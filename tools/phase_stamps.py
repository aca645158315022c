#!/usr/bin/env python3
"""Where a wave's time goes, per item phase: run with a library built with -DPDS_STAMPS=1
(tools/build_variant.sh stamps -DPDS_STAMPS=1; PDS_AMD_LIB=variants/lib_stamps.so python tools/phase_stamps.py
[workload]).  Prints shader-clock cycles per item and phase, averaged over all waves."""
import ctypes
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
lib = ps._native.lib()
fn = lib.pds_debug_set_stamp_buffer
fn.argtypes = [ctypes.c_void_p]
fn.restype = None
lengths = np.full(B, n, dtype=np.int64)
offsets = np.concatenate([[0], np.cumsum(lengths)[:-1]]).astype(np.int64)
signal = torch.randn(int(lengths.sum()), device=dev).mul_(3000.0)
layout = comp.prepare_layout(offsets, lengths, device=dev)
out = torch.empty((layout.total_rows, comp.num_coeffs), dtype=torch.float32, device=dev)
buf = torch.zeros(65536 * 8, dtype=torch.int64, device=dev)
fn(buf.data_ptr())
for _ in range(300):  # clocks up
    comp.launch(signal, layout, out=out)
torch.cuda.synchronize()
buf.zero_()
comp.launch(signal, layout, out=out)
torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(-1, 8)
t = t[t[:, 6] > 0]
items = t[:, 6].astype(np.float64)
names = ["issue frame loads", "wait for frame loads", "window + N1 transform + exchange stores",
         "exchange reads + N2 transform + power", "P stores (+ energy)", "filter walk + stores", None,
         "bookkeeping / previous item's tail"]
print(f"{wl}: {len(t)} waves, {items.mean():.1f} items per wave")
tot = 0.0
for i, nm in enumerate(names):
    if nm is None:
        continue
    per = (t[:, i] / items).mean()
    tot += per
    print(f"  {nm:44s} {per:9.0f} cycles per item")
print(f"  {'sum':44s} {tot:9.0f} cycles per item and wave")
