#!/usr/bin/env python3
"""Host-to-host rate of the headline workload: signals in HOST memory in, features in host memory out.

    python tools/feed_rate.py [batches] [utterances per batch]

Legs, each over the same batches of 10 s utterances at 16 kHz (BASELINE.json configs[1] per batch):
  * list-of-numpy API (compute_full_batch: concatenate, pageable upload, launch, download) -- what a caller of
    the reference's compute_full in a loop would switch to first;
  * one pinned buffer, upload + launch + download back to back on one stream (the number DESIGN.md quoted so far);
  * the host feed (pds_feed_*: pinned staging ring, three batches in flight), samples as float32, as the
    reference drivers' float64, and as int16 PCM -- with the copy into the staging buffer (signals start in
    ordinary pageable numpy arrays) and without it (the caller's readers write into the staging buffer).
Prints frames/s and the PCIe bytes per frame of each leg."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
import pydrobert_speech_amd as ps
from pydrobert_speech_amd.alias import alias_factory_subclass_from_arg
from pydrobert_speech_amd.feed import HostFeed, Ticket

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 8
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
cfg, n, _, _ = bench.WORKLOADS[bench.DEFAULT_WORKLOAD]
comp = alias_factory_subclass_from_arg(ps.compute.FrameComputer, cfg)
ps.config.FLOAT64_ARITHMETIC = "float32"
rng = np.random.default_rng(0)
base = (3000 * rng.standard_normal(B * n)).astype(np.float32)
frames_per_batch = B * comp.num_frames(n)
C = comp.num_coeffs


def report(name, seconds, batches, in_bytes):
    fr = batches * frames_per_batch
    print(f"{name:62s} {fr / seconds / 1e6:8.1f} M frames/s   {1e3 * seconds / batches:7.2f} ms per batch   "
          f"{in_bytes * comp.frame_shift + 4 * C:5d} B/frame over PCIe", flush=True)


sigs32 = [base[i * n : (i + 1) * n] for i in range(B)]
for host_feed in (False, True):
    ps.config.HOST_FEED = host_feed
    comp.compute_full_batch(sigs32)  # (first call: plan tables; with the feed, its pinned slots)
    t0 = time.perf_counter()
    for _ in range(max(2, nb // 2)):
        feats = comp.compute_full_batch(sigs32)
    report("list-of-numpy API, float32 (compute_full_batch), " + ("through the staging ring" if host_feed else "plain path"),
           time.perf_counter() - t0, max(2, nb // 2), 4)

pinned = torch.from_numpy(base).pin_memory()
layout = comp.prepare_layout(np.arange(B) * n, np.full(B, n))
out_host = torch.empty((layout.total_rows, C), dtype=torch.float32).pin_memory()
for trial in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(nb):
        d = pinned.to("cuda", non_blocking=True)
        out = comp.launch(d, layout)
        out_host.copy_(out, non_blocking=True)
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
report("one pinned buffer, upload + launch + download in turn, float32", dt, nb, 4)

for dtype, in_bytes, direct in ((np.float32, 4, False), (np.float32, 4, True), (np.float64, 8, False), (np.float64, 8, True),
                                (np.int16, 2, False), (np.int16, 2, True)):
    data = base.astype(dtype) if dtype != np.int16 else np.clip(base, -32768, 32767).astype(np.int16)
    sigs = [data[i * n : (i + 1) * n] for i in range(B)]
    mode = "direct" if direct else "staged"
    with HostFeed(comp, dtype, slot_samples=B * n, slot_utts=B, slots=3, copy_threads=16, direct=direct) as feed:
        for _ in feed.run([sigs] * 3, copy=False):
            pass
        t0 = time.perf_counter()
        for feats in feed.run([sigs] * nb, copy=False):
            pass
        report(f"host feed ({mode}), {np.dtype(dtype).name} samples, pageable numpy in (16 copy threads)", time.perf_counter() - t0, nb, in_bytes)
        # the caller's readers write into the staging buffer themselves: no copy in front of the upload
        lengths = np.full(B, n, dtype=np.int64)
        lib, handle = feed._lib, feed._handle
        pending = []
        for phase in range(2):
            t0 = time.perf_counter()
            for k in range(nb):
                if len(pending) >= 2:
                    feed.collect(pending.pop(0), copy=False)
                slot, view = feed.acquire()
                if phase == 0 and k < 3:
                    view[: B * n] = data  # (fill every slot once; later rounds re-send what is there)
                rc = lib.pds_feed_submit(handle, slot, lengths.ctypes.data, B, 0.0, 1)
                assert rc == 0
                pending.append(Ticket(slot, B))
            while pending:
                got, rows = feed.collect(pending.pop(0), copy=False)
            dt = time.perf_counter() - t0
        report(f"host feed ({mode}), {np.dtype(dtype).name} samples, staging buffer filled by the caller", dt, nb, in_bytes)
        ref = comp.compute_packed(torch.from_numpy(data).cuda(), np.arange(B) * n, np.full(B, n))[0].float().cpu().numpy()
        assert np.array_equal(got, ref), "feed result differs from the packed launch"
