#!/usr/bin/env python3
"""Golden fixtures for the rows either side of the hot path.

Authoring container only (the reference is imported from /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_io.py

Writes
  io.npz            Stack inputs / reference outputs; CMVN outputs with statistics from a file
  cmvn_stats.npy    statistics written by the reference's Standardize.save (numpy format)
  cmvn_stats.f64    the same through numpy.ndarray.tofile (raw float64 words)
  cmvn_stats.f32    raw float32 words (exercises the reference's float-width detection)
  sig_mono.wav, sig_stereo.wav, sig.npy, sig.npz, sig.pt, sig.raw
                    one short signal in every container read_signal handles here, written
                    with the standard library / numpy / torch (data files, no code)
"""
import os
import sys
import wave

import numpy as np

REF = "/root/reference"
sys.path.insert(0, os.path.join(REF, "src"))
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))

from pydrobert.speech import post as rpost  # noqa: E402
from pydrobert.speech import util as rutil  # noqa: E402


def main():
    rng = np.random.default_rng(2024)
    out = {}
    # ---- Stack ----------------------------------------------------------------------------
    x = (rng.standard_normal((11, 5)) * 4).astype("f4")
    out["stack/in2"] = x
    out["stack/out2/nv3"] = rpost.Stack(3).apply(x, axis=1)
    out["stack/out2/nv3_edge"] = rpost.Stack(3, pad_mode="edge").apply(x, axis=1)
    out["stack/out2/nv4_const"] = rpost.Stack(4, pad_mode="constant").apply(x, axis=-1)
    out["stack/out2/nv2_t1"] = rpost.Stack(2, time_axis=1).apply(x.T.copy(), axis=0)
    out["stack/out2/nv1"] = rpost.Stack(1).apply(x, axis=1)
    out["stack/out2/nv12"] = rpost.Stack(12).apply(x, axis=1)
    x3 = rng.standard_normal((3, 10, 4))
    out["stack/in3"] = x3
    out["stack/out3/nv3_t1_a2"] = rpost.Stack(3, time_axis=1).apply(x3, axis=2)
    out["stack/out3/nv4_t1_a0_reflect"] = rpost.Stack(4, time_axis=1, pad_mode="reflect").apply(x3, axis=0)
    out["stack/out3/nv2_tm1_a1"] = rpost.Stack(2, time_axis=-1).apply(x3, axis=1)

    # ---- CMVN statistics files ---------------------------------------------------------------
    feats = (rng.standard_normal((40, 6)) * np.arange(1, 7) + np.arange(6)).astype("f4")
    st = rpost.Standardize()
    st.accumulate(feats[:25])
    st.accumulate(feats[25:])
    st.save(os.path.join(HERE, "cmvn_stats.npy"))
    st.save(os.path.join(HERE, "cmvn_stats.f64"))
    st._stats.astype(np.float32).tofile(os.path.join(HERE, "cmvn_stats.f32"))
    probe = (rng.standard_normal((9, 6)) * 3).astype("f4")
    out["cmvn_file/stats"] = st._stats.copy()
    out["cmvn_file/in"] = probe
    for name in ("cmvn_stats.npy",):
        out["cmvn_file/out_npy"] = rpost.Standardize(os.path.join(HERE, name)).apply(probe)
    out["cmvn_file/out_f64"] = rpost.Standardize(os.path.join(HERE, "cmvn_stats.f64"), force_as="file").apply(probe)
    out["cmvn_file/out_f32"] = rpost.Standardize(os.path.join(HERE, "cmvn_stats.f32"), force_as="file").apply(probe)
    out["cmvn_file/out_novar"] = rpost.Standardize(os.path.join(HERE, "cmvn_stats.npy"), norm_var=False).apply(probe)

    # ---- signal containers ------------------------------------------------------------------
    import torch

    sig = (rng.standard_normal(1200) * 3000).astype("<i2")
    with wave.open(os.path.join(HERE, "sig_mono.wav"), "wb") as fh:
        fh.setnchannels(1), fh.setsampwidth(2), fh.setframerate(16000)
        fh.writeframes(sig.tobytes())
    with wave.open(os.path.join(HERE, "sig_stereo.wav"), "wb") as fh:
        fh.setnchannels(2), fh.setsampwidth(2), fh.setframerate(16000)
        fh.writeframes(sig.tobytes())
    np.save(os.path.join(HERE, "sig.npy"), sig.astype("f4"))
    np.savez(os.path.join(HERE, "sig.npz"), sig.astype("f8"), other=sig[::-1].astype("f8"))
    torch.save(torch.from_numpy(sig.astype("f4")), os.path.join(HERE, "sig.pt"))
    sig.astype("f4").tofile(os.path.join(HERE, "sig.raw"))
    # what the reference reads back from each
    out["read/mono"] = rutil.read_signal(os.path.join(HERE, "sig_mono.wav"))
    out["read/stereo_f8"] = rutil.read_signal(os.path.join(HERE, "sig_stereo.wav"), dtype=np.float64)
    out["read/npy"] = rutil.read_signal(os.path.join(HERE, "sig.npy"))
    out["read/npz_default"] = rutil.read_signal(os.path.join(HERE, "sig.npz"))
    out["read/npz_other_f4"] = rutil.read_signal(os.path.join(HERE, "sig.npz"), key="other", dtype="f4")
    out["read/pt"] = rutil.read_signal(os.path.join(HERE, "sig.pt"))
    out["read/raw_f4"] = rutil.read_signal(os.path.join(HERE, "sig.raw"), dtype="f4", force_as="file")
    np.savez_compressed(os.path.join(HERE, "io.npz"), **out)
    print("io fixtures written:", sorted(out))


if __name__ == "__main__":
    main()
