#!/usr/bin/env python3
"""How evenly the waves of the fused STFT kernel finish: run with a library built with -DPDS_STAMPS=2
(entry / loop start / loop end of every wave, nothing inside the loop, so the build runs like the product):
PDS_AMD_LIB=variants/lib_spread.so python tools/wave_spread.py [workload].  A launch lasts as long as its slowest
wave; the mean over the waves is what an even distribution of the items would take."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
import pydrobert_speech_amd as ps
from pydrobert_speech_amd.alias import alias_factory_subclass_from_arg

args = [a for a in sys.argv[1:] if not a.startswith("--")]
wl = args[0] if args else bench.DEFAULT_WORKLOAD
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
buf = torch.zeros(65536 * 12, dtype=torch.int64, device=dev)
fn(buf.data_ptr())
for _ in range(400):  # clocks up
    comp.launch(signal, layout, out=out)
torch.cuda.synchronize()
for rep in range(3):
    buf.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    comp.launch(signal, layout, out=out)
    e1.record()
    torch.cuda.synchronize()
    raw = buf.cpu().numpy().reshape(-1, 12)
    live = raw[:, 6] > 0
    ab = raw[live][:, 8:11].astype(np.float64)
    items = raw[live][:, 6]
    print(f"launch {rep}: {e0.elapsed_time(e1):.4f} ms, {int(live.sum())} waves, items per wave {items.min()}..{items.max()}")
    # every XCD counts from its own origin: cluster the entry times (a launch lasts < 1e6 ticks)
    order = np.argsort(ab[:, 0])
    keys = np.zeros(len(ab))
    keys[order] = np.cumsum(np.concatenate([[0], np.diff(ab[order, 0]) > 2e6]))
    spans, means = [], []
    for k in np.unique(keys):
        m = ab[keys == k]
        t0 = m[:, 0].min()
        busy = m[:, 2] - m[:, 1]
        span = m[:, 2].max() - t0
        spans.append(span)
        means.append(busy.mean())
        print("  XCD@%.0f: %4d waves | entry p50 %6.0f max %6.0f | loop start p50 %6.0f | loop end min %7.0f p50 %7.0f max %7.0f | "
              "busy mean %7.0f | cycles per item %5.0f" % (k, len(m), np.median(m[:, 0]) - t0, m[:, 0].max() - t0,
              np.median(m[:, 1]) - t0, m[:, 2].min() - t0, np.median(m[:, 2]) - t0, span, busy.mean(),
              (busy / items[keys == k]).mean()))
    print("  slowest XCD span %.0f ticks; mean busy %.0f; span / mean busy %.3f" % (max(spans), np.mean(means), max(spans) / np.mean(means)))
