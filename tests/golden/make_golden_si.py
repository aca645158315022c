#!/usr/bin/env python3
"""Golden fixtures for ShortIntegrationFrameComputer.

Authoring container only (imports the reference from /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_si.py

Writes si.npz: per configuration the reference's derived sizes, its FIR taps (inverse DFT of
the stored filter spectra, as its own test does: tests/test_compute.py:129-141), its window, and
compute_full outputs for several lengths and dtypes; chunked outputs for one configuration.
Configurations and lengths go to si_configs.json.
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

CONFIGS = {
    # zero-phase complex bank, centered frames, magnitude (the reference's defaults)
    "s1_gabor_mel": {"name": "si", "bank": {"name": "gabor", "scaling_function": "mel", "num_filts": 12}},
    # causal complex bank (GammaWindow integration), power
    "s2_gammatone_power": {"name": "si", "bank": {"name": "gammatone", "scaling_function": "mel", "num_filts": 9},
                           "use_power": True},
    # real bank with energy, short shift, no log, 8 kHz
    "s3_tri_energy_nolog": {"name": "si", "bank": {"name": "tri", "scaling_function": "mel", "num_filts": 6,
                            "sampling_rate": 8000}, "include_energy": True, "frame_shift_ms": 5, "use_log": False},
    # translation shorter than the shift (virtual leading zeros), no power-of-two padding, other window
    "s4_gabor_8k_short": {"name": "si", "bank": {"name": "gabor", "scaling_function": "bark", "num_filts": 8,
                          "sampling_rate": 8000}, "pad_to_nearest_power_of_two": False,
                          "window_function": "hamming", "use_power": True},
    # real bank sampled in frequency (impulse responses by inverse DFT), supports of ~7000 taps:
    # beyond the FFT form's 1024-point transforms, so float32 takes the direct kernel too
    "s6_fbank_long": {"name": "si", "bank": {"name": "fbank", "num_filts": 3}, "use_power": True},
    # causal style forced on a zero-phase bank
    "s5_gabor_causal": {"name": "si", "bank": {"name": "gabor", "scaling_function": "mel", "num_filts": 5},
                        "frame_style": "causal", "frame_shift_ms": 12.5},
}


def main():
    rng = np.random.default_rng(4242)
    master = rng.standard_normal(6000) * 3000
    out = {"master": master}
    meta = {"configs": CONFIGS, "lengths": {}}
    for name, cfg in CONFIGS.items():
        comp = alias_factory_subclass_from_arg(rcompute.FrameComputer, json.loads(json.dumps(cfg)))
        S, M = comp._frame_shift, comp._max_support
        out[f"{name}/dims"] = np.asarray([S, M, comp._translation, comp._dft_size, comp._frame_length,
                                          comp.num_coeffs, int(comp._real), int(comp._frame_style == "centered"),
                                          int(comp._power), int(comp._log), int(comp.includes_energy)])
        taps = np.stack([comp._compute_idft(f.copy())[:M] for f in comp._filts])
        tail = max(np.abs(comp._compute_idft(f.copy())[M + 1:]).max() for f in comp._filts)
        assert tail < 1e-9, (name, tail)
        out[f"{name}/taps"] = taps
        out[f"{name}/window"] = comp._window.reshape(-1).copy()
        lens = [0, 1, S - 1, S, 2 * S + 3, 1000, 4001]
        meta["lengths"][name] = lens
        for n in lens:
            for dt in ("f4", "f8"):
                y = comp.compute_full(master[:n].astype(dt))
                assert y.dtype == np.dtype(dt)
                out[f"{name}/full/{n}/{dt}"] = y
    # streaming: chunk lists -> per-chunk outputs, then finalize
    comp = alias_factory_subclass_from_arg(rcompute.FrameComputer, json.loads(json.dumps(CONFIGS["s1_gabor_mel"])))
    for tag, cuts in (("c1024", list(range(1024, 4001, 1024))), ("ragged", [7, 1007, 1008, 2500, 3999])):
        x = master[:4001].astype("f4")
        pieces = np.split(x, cuts)
        outs = [comp.compute_chunk(p) for p in pieces] + [comp.finalize()]
        out[f"s1_gabor_mel/stream/{tag}/cuts"] = np.asarray(cuts)
        out[f"s1_gabor_mel/stream/{tag}/counts"] = np.asarray([len(o) for o in outs])
        out[f"s1_gabor_mel/stream/{tag}/feats"] = np.concatenate(outs)
    np.savez_compressed(os.path.join(HERE, "si.npz"), **out)
    with open(os.path.join(HERE, "si_configs.json"), "w") as fh:
        json.dump(meta, fh, indent=1, sort_keys=True)
    print("si fixtures:", os.path.getsize(os.path.join(HERE, "si.npz")), "bytes")
    for name in CONFIGS:
        print(name, out[f"{name}/dims"].tolist(), [out[f"{name}/full/{n}/f4"].shape[0] for n in meta["lengths"][name]])


if __name__ == "__main__":
    main()
