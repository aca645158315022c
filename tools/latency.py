#!/usr/bin/env python3
"""Per-call latency of compute_full(signal) for one host signal (the reference's API, called per utterance in a loop):
the plain path (pageable upload, launch, download) against a small host feed in direct mode (the kernel reads the
pinned copy of the signal and writes the pinned features itself).   python tools/latency.py [seconds of audio]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
import pydrobert_speech_amd as ps
from pydrobert_speech_amd.alias import alias_factory_subclass_from_arg
from pydrobert_speech_amd.feed import HostFeed

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
cfg, _, _, _ = bench.WORKLOADS[bench.DEFAULT_WORKLOAD]
comp = alias_factory_subclass_from_arg(ps.compute.FrameComputer, cfg)
n = int(secs * 16000)
rng = np.random.default_rng(0)
x = (3000 * rng.standard_normal(n)).astype(np.float32)
ref = comp.compute_full(x)


def timed(fn, reps=300):
    for _ in range(20):
        fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    return (time.perf_counter() - t0) / reps * 1e6, out


ps.config.HOST_FEED = False
us, out = timed(lambda: comp.compute_full(x))
print(f"compute_full, {secs:g} s of audio ({out.shape[0]} frames): plain path {us:8.1f} us per call")
ps.config.HOST_FEED = True
us, out = timed(lambda: comp.compute_full(x))
assert np.array_equal(out, ref)
print(f"   compute_full through its own small direct feed (the default): {us:8.1f} us per call")
for direct in (True, False):
    with HostFeed(comp, np.float32, slot_samples=max(n, 1 << 16), slot_utts=4, slots=2, copy_threads=1, direct=direct) as feed:
        us, out = timed(lambda: feed.collect(feed.submit([x]))[0])
        assert np.array_equal(out, ref)
        print(f"   host feed, {'direct' if direct else 'staged'}: {us:8.1f} us per call")
