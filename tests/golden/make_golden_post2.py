#!/usr/bin/env python3
"""Deltas fixtures for numpy.pad modes whose padded samples depend on the pad width or are
computed rather than copied (the reference pads every order by its own reach, in float64:
post.py:470-483).  Written by running the reference in the authoring container:

    PYTHONPATH=/root/reference/src python tests/golden/make_golden_post2.py
"""
import os

import numpy as np

from pydrobert.speech import post as rpost

HERE = os.path.dirname(os.path.abspath(__file__))


def ramp_to_first(vector, pad_width, iaxis, kwargs):
    """a callable pad mode whose values depend on the pad width"""
    lo, hi = pad_width
    if lo:
        vector[:lo] = vector[lo] * np.arange(lo, 0, -1) / (lo + 1)
    if hi:
        vector[-hi:] = vector[-hi - 1] + np.arange(1, hi + 1) * 0.25


CASES = {
    "linear_ramp": dict(pad_mode="linear_ramp", end_values=(1.5, -2.0)),
    "mean_stat2": dict(pad_mode="mean", stat_length=2),
    "maximum": dict(pad_mode="maximum"),
    "constant_tenth": dict(pad_mode="constant", constant_values=0.1),
    "reflect_odd": dict(pad_mode="reflect", reflect_type="odd"),
    "symmetric": dict(pad_mode="symmetric"),
    "wrap": dict(pad_mode="wrap"),
    "callable": dict(pad_mode=ramp_to_first),
}


def main():
    rng = np.random.default_rng(4242)
    out = {}
    for dt in ("f4", "f8"):
        x = (rng.standard_normal((23, 5)) * 3 + 1).astype(dt)
        out[f"in/{dt}"] = x
        for name, kwargs in CASES.items():
            y = rpost.Deltas(2, context_window=2, target_axis=-1, **kwargs).apply(x, axis=0)
            assert y.dtype == x.dtype and y.shape == (23, 15)
            out[f"out/{name}/{dt}"] = y
    np.savez_compressed(os.path.join(HERE, "post2.npz"), **out)
    print("wrote post2.npz:", len(out), "arrays")


if __name__ == "__main__":
    main()
