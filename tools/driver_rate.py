#!/usr/bin/env python3
"""End-to-end rate of the batched signals-to-torch-feat-dir driver on a synthetic corpus.

    python tools/driver_rate.py [num_utts] [seconds]      (GPU box; writes under /tmp)

Writes `num_utts` 16-bit wave files, then times map -> features -> one .pt file per utterance
(40-mel fbank, pre-emphasis fused, per-utterance CMVN) and prints frames/s and utterances/s.
"""
import json
import os
import sys
import tempfile
import time
import wave

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from pydrobert_speech_amd.command_line import signals_to_torch_feat_dir  # noqa: E402


def main():
    n_utts = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    secs = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
    rng = np.random.default_rng(0)
    with tempfile.TemporaryDirectory(dir="/tmp") as root:
        lines = []
        for i in range(n_utts):
            x = (rng.standard_normal(int(16000 * secs)) * 3000).astype("<i2")
            path = os.path.join(root, f"u{i:05d}.wav")
            with wave.open(path, "wb") as fh:
                fh.setnchannels(1), fh.setsampwidth(2), fh.setframerate(16000)
                fh.writeframes(x.tobytes())
            lines.append(f"u{i:05d} {path}")
        with open(os.path.join(root, "map.txt"), "w") as fh:
            fh.write("\n".join(lines) + "\n")
        cfg = {"name": "stft", "bank": {"name": "fbank", "num_filts": 40}, "frame_length_ms": 25, "use_power": True}
        for workers in (0, 8):
            out = os.path.join(root, f"feats{workers}")
            t0 = time.perf_counter()
            rc = signals_to_torch_feat_dir([
                os.path.join(root, "map.txt"), json.dumps(cfg), out, "--preprocess", json.dumps({"name": "preemph"}),
                "--postprocess", json.dumps({"name": "cmvn"}), "--num-workers", str(workers), "--batch-utts", "256"])
            dt = time.perf_counter() - t0
            assert rc == 0 and len(os.listdir(out)) == n_utts
            frames = n_utts * int(secs * 100)
            print(json.dumps({"num_workers": workers, "utterances": n_utts, "seconds_each": secs, "wall_s": round(dt, 3),
                              "frames_per_s": round(frames / dt), "utterances_per_s": round(n_utts / dt, 1)}))


if __name__ == "__main__":
    main()
