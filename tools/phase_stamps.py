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

args = [a for a in sys.argv[1:] if not a.startswith("--")]
fused = "--fused-deltas" in sys.argv  # the one-launch statics + deltas kernel of a workload with Deltas
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
out = torch.empty((layout.total_rows, comp.num_coeffs * (3 if fused else 1)), dtype=torch.float32, device=dev)
deltas = ps.post.Deltas(2)


def launch():
    if fused:
        comp.launch_with_deltas(signal, layout, deltas, out=out, fused=True)
    else:
        comp.launch(signal, layout, out=out)


buf = torch.zeros(65536 * 12, dtype=torch.int64, device=dev)
fn(buf.data_ptr())
for _ in range(300):  # clocks up
    launch()
torch.cuda.synchronize()
buf.zero_()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
launch()
e1.record()
torch.cuda.synchronize()
print("timed launch: %.3f ms" % e0.elapsed_time(e1))
raw = buf.cpu().numpy().reshape(-1, 12)
t = raw[raw[:, 6] > 0][:, :8]
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
# spread over the waves: a launch lasts as long as its slowest wave
all_t = raw[:, :8]
live = all_t[:, 6] > 0
tot_w = np.delete(all_t, 6, axis=1).sum(axis=1)[live]
q = np.percentile(tot_w, [0, 5, 50, 95, 100])
print("  per-wave total cycles: min %.0f  p5 %.0f  median %.0f  p95 %.0f  max %.0f" % tuple(q))
idx = np.nonzero(live)[0]
for name, key in (("wave index % 8 (XCD of the workgroup when waves/wg divides)", idx % 8), ("first/second half of the grid", (idx * 2) // max(1, idx.max() + 1))):
    print("  mean total by " + name + ": " + " ".join("%.0f" % tot_w[key == k].mean() for k in np.unique(key)))
ab = raw[live][:, 8:11].astype(np.float64)
# (every XCD counts from its own origin: group the waves by it -- the origins lie ~1e11 ticks apart)
keys = np.round(ab[:, 0] / 1e9)
for k in np.unique(keys):
    m = ab[keys == k]
    t0 = m[:, 0].min()
    print("  XCD@%.0f: %4d waves | entry p50 %6.0f max %6.0f | loop start p50 %6.0f max %6.0f | loop end p5 %7.0f p50 %7.0f max %7.0f ticks"
          % (k, len(m), np.median(m[:, 0]) - t0, m[:, 0].max() - t0, np.median(m[:, 1]) - t0, m[:, 1].max() - t0,
             np.percentile(m[:, 2], 5) - t0, np.median(m[:, 2]) - t0, m[:, 2].max() - t0))
