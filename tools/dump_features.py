#!/usr/bin/env python3
"""Features of one seeded ragged batch to an .npy file -- for bit-identity checks between library builds
(PDS_AMD_LIB=variants/lib_x.so python tools/dump_features.py out.npy [rate] [preemph] [f32|i16|f64])."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import pydrobert_speech_amd as ps
from pydrobert_speech_amd.alias import alias_factory_subclass_from_arg

out = sys.argv[1]
rate = int(sys.argv[2]) if len(sys.argv) > 2 else 8000
pre = float(sys.argv[3]) if len(sys.argv) > 3 else 0.97
fmt = sys.argv[4] if len(sys.argv) > 4 else "f32"
comp = alias_factory_subclass_from_arg(ps.compute.FrameComputer, {
    "name": "stft", "bank": {"name": "fbank", "sampling_rate": rate, "num_filts": 24}, "frame_length_ms": 25, "include_energy": True})
rng = np.random.default_rng(5)
lens = [0, 1, 99, 100, 101, 2000, 16001, 333, 80000]
sig = 3000 * rng.standard_normal(sum(lens))
sig = {"f32": sig.astype("f4"), "f64": sig, "i16": np.rint(sig).astype("i2")}[fmt]
offs = np.concatenate([[0], np.cumsum(lens)[:-1]])
x = torch.from_numpy(sig).cuda()
layout = comp.prepare_layout(offs, lens, device=x.device)
feats = comp.launch(x, layout, preemphasis=pre).cpu().numpy()
np.save(out, feats)
print(out, feats.shape, comp.kernel_kind, float(np.nansum(feats)))
