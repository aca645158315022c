import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
import pydrobert_speech_amd as ps
from pydrobert_speech_amd.alias import alias_factory_subclass_from_arg
rng = np.random.default_rng(0)
def lat(f, n=200):
    for _ in range(20): f()
    t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e6
stft = alias_factory_subclass_from_arg(ps.compute.FrameComputer, {"name": "stft", "bank": "fbank", "frame_length_ms": 25})
si = alias_factory_subclass_from_arg(ps.compute.FrameComputer, {'name': 'si', 'bank': {'name': 'gabor', 'scaling_function': 'mel', 'num_filts': 40}})
x4 = (3000 * rng.standard_normal(160000)).astype('f4'); x8 = x4.astype('f8')
print('stft compute_full f32 10 s: %.0f us' % lat(lambda: stft.compute_full(x4)))
print('stft compute_full f64 10 s: %.0f us' % lat(lambda: stft.compute_full(x8)))
print('si   compute_full f32 10 s: %.0f us' % lat(lambda: si.compute_full(x4)))
print('si   compute_full f64 10 s: %.0f us' % lat(lambda: si.compute_full(x8), 50))
