#!/usr/bin/env python3
"""compute_full_batch of float64 numpy signals (float64 arithmetic: the exact-parity path) host to host.
python tools/f64_batch_rate.py [utterances]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import pydrobert_speech_amd as ps
from pydrobert_speech_amd.alias import alias_factory_subclass_from_arg

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
comp = alias_factory_subclass_from_arg(ps.compute.FrameComputer, {"name": "stft", "bank": "fbank", "frame_length_ms": 25})
rng = np.random.default_rng(0)
sigs = [3000 * rng.standard_normal(160000) for _ in range(B)]
for _ in range(2):
    out = comp.compute_full_batch(sigs)
t0 = time.perf_counter()
for _ in range(3):
    out = comp.compute_full_batch(sigs)
dt = (time.perf_counter() - t0) / 3
frames = sum(o.shape[0] for o in out)
print("float64 compute_full_batch (%d x 10 s) host->host: %.1f ms, %.1f M frames/s, dtype %s" % (B, dt * 1e3, frames / dt / 1e6, out[0].dtype))
