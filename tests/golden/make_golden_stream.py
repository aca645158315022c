#!/usr/bin/env python3
"""More streaming fixtures: random chunkings of signals of assorted lengths through the
reference's compute_chunk / finalize (authoring container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_stream.py

Writes stream_random.npz (and si_stream_random.npz, the same for the short-integration computers): for four configurations x eight (length, chunking) draws, the cut
points, the number of frames each call returned and the concatenated features (float32 input).
The signal is the `master` array of signals.npz; configurations come from configs.json.
"""
import json
import os
import sys

import numpy as np

REF = "/root/reference"
sys.path.insert(0, os.path.join(REF, "src"))
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))

from pydrobert.speech import compute as rcompute  # noqa: E402
from pydrobert.speech.alias import alias_factory_subclass_from_arg  # noqa: E402

NAMES = ["c1_kaldi_fbank", "c2_tri_mel40", "v_tri_analytic_nolog", "v_gabor_nopad_mag"]


def main():
    with open(os.path.join(HERE, "configs.json")) as fh:
        configs = json.load(fh)["configs"]
    master = np.load(os.path.join(HERE, "signals.npz"))["master"]
    rng = np.random.default_rng(31337)
    out = {}
    for name in NAMES:
        comp = alias_factory_subclass_from_arg(rcompute.FrameComputer, json.loads(json.dumps(configs[name])))
        L, S = comp.frame_length, comp.frame_shift
        for case in range(8):
            n = int([1, L // 2, L // 2 + 1, L - 1, L + S, 3 * L + 7, 2000, 4000][case])
            x = master[50 : 50 + n].astype("f4")
            num_cuts = int(rng.integers(0, 9))
            cuts = np.sort(rng.integers(0, n + 1, size=num_cuts))
            pieces = np.split(x, cuts)
            outs = [comp.compute_chunk(p) for p in pieces] + [comp.finalize()]
            out[f"{name}/{case}/n"] = np.asarray(n)
            out[f"{name}/{case}/cuts"] = cuts.astype(np.int64)
            out[f"{name}/{case}/counts"] = np.asarray([len(o) for o in outs], dtype=np.int64)
            out[f"{name}/{case}/feats"] = np.concatenate(outs)
    np.savez_compressed(os.path.join(HERE, "stream_random.npz"), **out)
    print("stream_random.npz:", os.path.getsize(os.path.join(HERE, "stream_random.npz")), "bytes")
    # the same for the short-integration computers of si_configs.json (master signal of si.npz)
    with open(os.path.join(HERE, "si_configs.json")) as fh:
        si_configs = json.load(fh)["configs"]
    si_master = np.load(os.path.join(HERE, "si.npz"))["master"]
    out = {}
    for name, cfg in sorted(si_configs.items()):
        comp = alias_factory_subclass_from_arg(rcompute.FrameComputer, json.loads(json.dumps(cfg)))
        S = comp.frame_shift
        for case in range(6):
            n = int([1, S - 1, 2 * S + 1, 5 * S + 3, 1500, 4000][case])
            x = si_master[7 : 7 + n].astype("f4")
            cuts = np.sort(rng.integers(0, n + 1, size=int(rng.integers(0, 7))))
            outs = [comp.compute_chunk(p) for p in np.split(x, cuts)] + [comp.finalize()]
            out[f"{name}/{case}/n"] = np.asarray(n)
            out[f"{name}/{case}/cuts"] = cuts.astype(np.int64)
            out[f"{name}/{case}/counts"] = np.asarray([len(o) for o in outs], dtype=np.int64)
            out[f"{name}/{case}/feats"] = np.concatenate(outs)
    np.savez_compressed(os.path.join(HERE, "si_stream_random.npz"), **out)
    print("si_stream_random.npz:", os.path.getsize(os.path.join(HERE, "si_stream_random.npz")), "bytes")


if __name__ == "__main__":
    main()
