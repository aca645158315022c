import sys, time, cProfile, pstats
sys.path.insert(0, '/root/repo')
import numpy as np
import pydrobert_speech_amd as ps
from pydrobert_speech_amd.alias import alias_factory_subclass_from_arg
comp = alias_factory_subclass_from_arg(ps.compute.FrameComputer, {"name": "stft", "bank": "fbank", "frame_length_ms": 25})
rng = np.random.default_rng(0)
base = (3000 * rng.standard_normal(160000)).astype('f4')
sigs = [base.copy() for _ in range(1024)]
for _ in range(2): out = comp.compute_full_batch(sigs)
t0 = time.perf_counter()
for _ in range(3): out = comp.compute_full_batch(sigs)
dt = (time.perf_counter() - t0) / 3
print('compute_full_batch 1024 x 10 s f32: %.1f ms, %.1f M frames/s' % (dt * 1e3, 1024000 / dt / 1e6))
pr = cProfile.Profile(); pr.enable()
for _ in range(3): out = comp.compute_full_batch(sigs)
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
