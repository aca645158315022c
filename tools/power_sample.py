#!/usr/bin/env python3
"""What clock and power does the chip hold under the headline kernel?  Runs the launch back to back for a few seconds
per data set (Gaussian noise, zeros) while a thread samples the SMI's power / clock readings (rocm-smi / amd-smi, or the
hwmon files), and prints launch time next to the readings.  python tools/power_sample.py [workload] [seconds]"""
import glob
import json
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
import pydrobert_speech_amd as ps
from pydrobert_speech_amd.alias import alias_factory_subclass_from_arg

wl = sys.argv[1] if len(sys.argv) > 1 else bench.DEFAULT_WORKLOAD
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
cfg, n, B, post = bench.WORKLOADS[wl]
comp = alias_factory_subclass_from_arg(ps.compute.FrameComputer, cfg)
dev = torch.device("cuda", 0)
lengths = np.full(B, n, dtype=np.int64)
offsets = np.concatenate([[0], np.cumsum(lengths)[:-1]]).astype(np.int64)
total = int(lengths.sum())
layout = comp.prepare_layout(offsets, lengths, device=dev)
C = comp.num_coeffs
deltas = ps.post.Deltas(2) if post == "deltas2" else None
out = torch.empty((layout.total_rows, 3 * C if deltas is not None else C), dtype=torch.float32, device=dev)
_launch = comp.launch
if deltas is not None:  # (the one-launch statics + deltas kernel where the plan has it)
    comp.launch = lambda x, layout, out=None: comp.launch_with_deltas(x, layout, deltas, out=out, fused=True)


def read_hwmon():
    vals = {}
    for card in glob.glob("/sys/class/drm/card*/device"):
        for name in ("pp_dpm_sclk", "pp_dpm_mclk"):
            try:
                txt = open(os.path.join(card, name)).read()
                cur = [l for l in txt.splitlines() if l.strip().endswith("*")]
                if cur:
                    vals[os.path.basename(os.path.dirname(card)) + ":" + name] = cur[0].strip()
            except OSError:
                pass
        for hw in glob.glob(os.path.join(card, "hwmon/hwmon*")):
            for name in ("power1_average", "power1_input", "power1_cap", "freq1_input", "temp1_input"):
                try:
                    vals[os.path.basename(os.path.dirname(card)) + ":" + name] = open(os.path.join(hw, name)).read().strip()
                except OSError:
                    pass
    return vals


def read_smi():
    for cmd in (["rocm-smi", "--showpower", "--showclocks", "--showmaxpower", "--json"],
                ["amd-smi", "metric", "--power", "--clock", "--json"]):
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=20)
            if r.returncode == 0 and r.stdout.strip():
                return cmd[0], r.stdout.strip()
        except (OSError, subprocess.TimeoutExpired):
            pass
    return None, None


samples = []
stop = threading.Event()


def sampler():
    while not stop.is_set():
        t = time.time()
        samples.append((t, "hwmon", read_hwmon()))
        which, txt = read_smi()
        if which:
            samples.append((time.time(), which, txt))
        time.sleep(0.2)


print("idle:", json.dumps(read_hwmon()))
which, txt = read_smi()
print("idle smi:", which, (txt or "")[:1500])
sigs = [("gaussian x 3000", torch.randn(total, device=dev).mul_(3000.0)), ("zeros", torch.zeros(total, device=dev)),
        ("gaussian x 3000 (again)", None)]
sigs[2] = (sigs[2][0], sigs[0][1])
for name, x in sigs:
    for _ in range(400):
        comp.launch(x, layout, out=out)
    torch.cuda.synchronize()
    samples.clear()
    stop.clear()
    th = threading.Thread(target=sampler)
    th.start()
    t0 = time.time()
    times = []
    while time.time() - t0 < secs:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(500):
            comp.launch(x, layout, out=out)
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) / 500)
    stop.set()
    th.join()
    print(f"== {wl} {name}: {np.mean(times):.4f} ms per launch over {len(times)} x 500 launches")
    hw = [s for s in samples if s[1] == "hwmon" and s[2]]
    if hw:
        keys = sorted(hw[0][2])
        for k in keys:
            vs = [s[2].get(k) for s in hw]
            print("   ", k, vs[len(vs) // 2], "(mid sample)", "distinct:", sorted(set(vs))[:6])
    smi = [s for s in samples if s[1] != "hwmon"]
    if smi:
        print("    smi mid sample:", smi[len(smi) // 2][2][:1500])
