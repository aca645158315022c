#!/usr/bin/env python3
"""Can the fused kernel read its samples straight from pinned HOST memory (and write its features there), so that
upload and download run concurrently, driven by the kernel's own loads and stores, instead of two DMA copies in
turn (tools/pcie_overlap.py: they do not overlap on this box)?  Times the headline batch with
  a) DMA upload -> kernel -> DMA download (one stream),
  b) kernel reading pinned host samples, features to device memory,
  c) kernel reading pinned host samples and writing pinned host features."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
import pydrobert_speech_amd as ps
from pydrobert_speech_amd import _native
from pydrobert_speech_amd.alias import alias_factory_subclass_from_arg

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
cfg, n, _, _ = bench.WORKLOADS[bench.DEFAULT_WORKLOAD]
comp = alias_factory_subclass_from_arg(ps.compute.FrameComputer, cfg)
lib = _native.lib()
plan = comp._native_plan()
rng = np.random.default_rng(0)
C = comp.num_coeffs
layout = comp.prepare_layout(np.arange(B) * n, np.full(B, n))
rows = layout.total_rows
meta = layout.d_meta
stream = torch.cuda.current_stream().cuda_stream


def launch(fn, sig_ptr, out_ptr):
    rc = fn(plan.handle, sig_ptr, meta[0].data_ptr(), meta[1].data_ptr(), meta[2].data_ptr(), meta[3].data_ptr(), B,
            int(layout.nframes.max()), -1, 0.0, out_ptr, C, stream)
    assert rc == 0, _native.last_error()


for name, dtype, fn in (("float32", np.float32, lib.pds_stft_batch_f32), ("int16", np.int16, lib.pds_stft_batch_i16in)):
    host = torch.from_numpy(np.clip(3000 * rng.standard_normal(B * n), -32768, 32767).astype(dtype)).pin_memory()
    out_host = torch.empty((rows, C), dtype=torch.float32).pin_memory()
    out_dev = torch.empty((rows, C), dtype=torch.float32, device="cuda")
    dev = host.cuda()
    launch(fn, dev.data_ptr(), out_dev.data_ptr())
    torch.cuda.synchronize()
    ref = out_dev.cpu()

    def timed(body, reps=5):
        body()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            body()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    def a():
        d = host.to("cuda", non_blocking=True)
        launch(fn, d.data_ptr(), out_dev.data_ptr())
        out_host.copy_(out_dev, non_blocking=True)

    def b():
        launch(fn, host.data_ptr(), out_dev.data_ptr())

    def c():
        launch(fn, host.data_ptr(), out_host.data_ptr())

    ta = timed(a)
    tb = timed(b)
    ok_b = torch.equal(out_dev.cpu(), ref)
    tc = timed(c)
    ok_c = torch.equal(out_host, ref)
    fr = B * comp.num_frames(n)
    print(f"{name}: DMA up + kernel + DMA down {ta * 1e3:7.2f} ms ({fr / ta / 1e6:6.1f} M frames/s) | kernel reads host "
          f"{tb * 1e3:7.2f} ms (equal: {ok_b}) | kernel reads and writes host {tc * 1e3:7.2f} ms ({fr / tc / 1e6:6.1f} M frames/s, equal: {ok_c})",
          flush=True)
