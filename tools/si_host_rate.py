import sys, time, json
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import pydrobert_speech_amd as ps
from pydrobert_speech_amd.alias import alias_factory_subclass_from_arg
cfg = {'name': 'si', 'bank': {'name': 'gabor', 'scaling_function': 'mel', 'num_filts': 40}, 'include_energy': True, 'use_power': True}
comp = alias_factory_subclass_from_arg(ps.compute.FrameComputer, cfg)
rng = np.random.default_rng(0)
sigs = [(3000 * rng.standard_normal(160000)).astype('f4') for _ in range(64)]
for _ in range(2): out = comp.compute_full_batch(sigs)
t0 = time.perf_counter(); n = 5
for _ in range(n): out = comp.compute_full_batch(sigs)
dt = (time.perf_counter() - t0) / n
frames = sum(o.shape[0] for o in out)
print('SI compute_full_batch host->host: %.2f ms per 64 x 10 s, %.1f M frames/s' % (dt * 1e3, frames / dt / 1e6))
x = torch.from_numpy(np.concatenate(sigs)).cuda()
lens = [160000] * 64; offs = np.arange(64) * 160000
for _ in range(3): f, rows = comp.compute_packed(x, offs, lens)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): f, rows = comp.compute_packed(x, offs, lens)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print('SI compute_packed device: %.2f ms, %.1f M frames/s' % (dt * 1e3, frames / dt / 1e6))
