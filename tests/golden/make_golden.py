#!/usr/bin/env python3
"""Generate the golden fixtures in this directory from the reference implementation.

Runs ONLY in the authoring container, where the reference checkout is mounted at
/root/reference (it never travels to the GPU box; the fixtures do).  Usage:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Writes
  configs.json          the configurations (alias-factory dicts) and signal lengths used
  signals.npz           the seeded input signals (float64 master copies)
  tables.npz            per config: window, filter start bins, truncated responses, sizes
  stft.npz              per config x length x dtype: reference compute_full output
  stream.npz            per config: reference compute_chunk/finalize outputs for fixed chunkings
  post.npz              Deltas / Standardize inputs and reference outputs
  pre.npz               Preemphasize outputs and compute_full of pre-emphasised signals
  kaldi.npz             the reference's own known-answer fixtures (tests/data/*.pkl),
                        decoded WITHOUT unpickling (opcode walk, nothing executed)
"""
import json
import os
import pickletools
import sys
import warnings

import numpy as np

REF = "/root/reference"
sys.path.insert(0, os.path.join(REF, "src"))
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))

from pydrobert.speech import compute as rcompute  # noqa: E402
from pydrobert.speech import post as rpost  # noqa: E402
from pydrobert.speech import pre as rpre  # noqa: E402
from pydrobert.speech.alias import alias_factory_subclass_from_arg  # noqa: E402

# ---------------------------------------------------------------------------------------
# configurations: BASELINE.json configs C1-C5 + edge variants
# ---------------------------------------------------------------------------------------
CONFIGS = {
    # README.md:11-21 of the reference (C1 / C2 bank with energy)
    "c1_readme_fbank": {
        "name": "stft", "bank": "fbank", "frame_length_ms": 25, "include_energy": True,
        "window_function": "hanning", "use_power": True,
    },
    # tests/data/fbank.json of the reference (C1 variant, kaldi_shift)
    "c1_kaldi_fbank": {
        "name": "stft",
        "bank": {"name": "fbank", "num_filts": 40, "low_hz": 20, "high_hz": 8000,
                 "sampling_rate": 16000, "analytic": False},
        "frame_length_ms": 25, "frame_shift_ms": 10, "frame_style": "centered",
        "include_energy": False, "pad_to_nearest_power_of_two": True,
        "window_function": "hanning", "use_log": True, "use_power": True, "kaldi_shift": True,
    },
    # BASELINE.json configs[0]/[1]: 40 TriangularOverlapping mel, Hann, 25/10 ms
    "c2_tri_mel40": {
        "name": "stft", "bank": {"name": "tri", "scaling_function": "mel", "num_filts": 40},
        "frame_length_ms": 25, "frame_shift_ms": 10, "window_function": "hanning",
        "use_power": True,
    },
    # C3: 80-mel fbank + energy
    "c3_fbank80_energy": {
        "name": "stft", "bank": {"name": "fbank", "num_filts": 80}, "frame_length_ms": 25,
        "include_energy": True, "use_power": True,
    },
    # C4: complex Gabor bank, 64 filters
    "c4_gabor64": {
        "name": "stft", "bank": {"name": "gabor", "scaling_function": "mel", "num_filts": 64},
        "frame_length_ms": 25, "use_power": True,
    },
    # C5: Gammatone 64 @ 48 kHz, 20 ms -> N = 1024, causal frames, Gamma window
    "c5_gammatone64_48k": {
        "name": "stft",
        "bank": {"name": "gammatone", "scaling_function": "mel", "num_filts": 64,
                 "sampling_rate": 48000},
        "frame_length_ms": 20, "use_power": True,
    },
    # variants exercising the remaining switches
    "v_gabor_nopad_mag": {  # N = L = 400 (not a power of two), magnitude spectrum, energy
        "name": "stft", "bank": {"name": "gabor", "scaling_function": "mel"},
        "frame_length_ms": 25, "pad_to_nearest_power_of_two": False, "use_power": False,
        "include_energy": True,
    },
    "v_tri_analytic_nolog": {  # complex (analytic) triangular bank, no log, causal + hamming
        "name": "stft",
        "bank": {"name": "tri", "scaling_function": "bark", "num_filts": 20, "analytic": True},
        "frame_length_ms": 20, "frame_shift_ms": 8, "frame_style": "causal",
        "window_function": "hamming", "use_log": False, "use_power": True,
    },
    "v_gammatone_16k_centered": {  # complex taps + centered frames + blackman, N = 256
        "name": "stft",
        "bank": {"name": "tonebank", "scaling_function": {"name": "linear", "low_hz": 0.0},
                 "num_filts": 12, "max_centered": True, "erb": True},
        "frame_length_ms": 16, "frame_shift_ms": 5, "frame_style": "centered",
        "window_function": "blackman", "use_power": False, "include_energy": True,
    },
    "v_gabor_default_length": {  # frame_length_ms=None -> derived from the bank's supports
        "name": "stft",
        "bank": {"name": "gabor", "scaling_function": {"name": "octave", "low_hz": 100.0},
                 "num_filts": 8,
                 "low_hz": 200.0, "scale_l2_norm": True, "erb": True}, "use_power": True,
    },
    "v_fbank_8k_bartlett": {
        "name": "stft", "bank": {"name": "fbank", "num_filts": 23, "sampling_rate": 8000},
        "frame_length_ms": 32, "frame_shift_ms": 16, "window_function": "bartlett",
        "use_power": True, "kaldi_shift": True,
    },
}

SEED = 1234


def lengths_for(computer):
    L, S = computer.frame_length, computer.frame_shift
    base = [0, max(1, L // 2 - 50), L // 2, L // 2 + 1, L // 2 + S // 4, L + 3 * S + 7,
            5 * L, 100 * S]
    return sorted(set(int(x) for x in base))


def build(cfg):
    return alias_factory_subclass_from_arg(rcompute.FrameComputer, json.loads(json.dumps(cfg)))


def master_signal(n):
    rng = np.random.default_rng(SEED)
    return 3000.0 * rng.standard_normal(n)


# ---------------------------------------------------------------------------------------
# pickle decoding without unpickling: walk the opcodes, pull out (shape, dtype, bytes)
# ---------------------------------------------------------------------------------------
class _Sym:
    """Symbolic stand-in for a global named in the pickle; never imported, never called"""

    def __init__(self, name):
        self.name = name


def load_numpy_pickle(path):
    """Decode a protocol-2 pickle of numpy arrays / ints / tuples by interpreting opcodes.

    Nothing named by the file is imported or called: the four globals these fixtures use
    (numpy.core.multiarray._reconstruct, numpy.ndarray, numpy.dtype, _codecs.encode) are
    handled symbolically by this function and anything else raises.  Array state is
    (version, shape, dtype, is_fortran, raw-bytes-as-latin1-text).
    """
    allowed = {
        "numpy.core.multiarray _reconstruct", "numpy ndarray", "numpy dtype", "_codecs encode",
    }
    stack, marks, memo = [], [], {}

    def pop_to_mark():
        k = marks.pop()
        items = stack[k:]
        del stack[k:]
        return items

    with open(path, "rb") as fh:
        blob = fh.read()
    for op, arg, _ in pickletools.genops(blob):
        n = op.name
        if n in ("PROTO", "STOP"):
            continue
        elif n == "GLOBAL":
            if arg not in allowed:
                raise ValueError(f"unexpected global in fixture pickle: {arg}")
            stack.append(_Sym(arg))
        elif n in ("BININT", "BININT1", "BININT2", "BINUNICODE", "SHORT_BINUNICODE",
                   "BINBYTES", "SHORT_BINBYTES"):
            stack.append(arg)
        elif n == "NONE":
            stack.append(None)
        elif n == "NEWFALSE":
            stack.append(False)
        elif n == "NEWTRUE":
            stack.append(True)
        elif n == "MARK":
            marks.append(len(stack))
        elif n == "TUPLE":
            stack.append(tuple(pop_to_mark()))
        elif n in ("TUPLE1", "TUPLE2", "TUPLE3"):
            k = int(n[-1])
            items = tuple(stack[-k:])
            del stack[-k:]
            stack.append(items)
        elif n in ("BINPUT", "LONG_BINPUT"):
            memo[arg] = stack[-1]
        elif n in ("BINGET", "LONG_BINGET"):
            stack.append(memo[arg])
        elif n == "REDUCE":
            args = stack.pop()
            fn = stack.pop()
            if not isinstance(fn, _Sym):
                raise ValueError("REDUCE of a non-global")
            if fn.name == "_codecs encode":
                text, codec = args
                assert codec == "latin1"
                stack.append(text.encode("latin1"))
            elif fn.name == "numpy dtype":
                stack.append({"dtype": args[0], "order": "="})
            elif fn.name == "numpy.core.multiarray _reconstruct":
                stack.append({"ndarray": None})
            else:
                raise ValueError(f"unexpected call in fixture pickle: {fn.name}")
        elif n == "BUILD":
            state = stack.pop()
            obj = stack[-1]
            if "dtype" in obj:
                obj["order"] = state[1]
            elif "ndarray" in obj:
                _version, shape, dt, fortran, raw = state
                assert not fortran
                np_dt = np.dtype(dt["dtype"]).newbyteorder(dt["order"])
                obj["ndarray"] = np.frombuffer(raw, dtype=np_dt).reshape(shape).copy()
            else:
                raise ValueError("BUILD of an unexpected object")
        else:
            raise ValueError(f"unexpected opcode in fixture pickle: {n}")

    def realise(x):
        if isinstance(x, dict) and "ndarray" in x:
            return x["ndarray"]
        if isinstance(x, tuple):
            return tuple(realise(v) for v in x)
        return x

    assert len(stack) == 1
    return realise(stack[0])


def main():
    warnings.simplefilter("error")
    out = {}
    tables, stft, stream = {}, {}, {}
    meta = {"configs": CONFIGS, "lengths": {}, "seed": SEED}
    sig = master_signal(60000)
    for name, cfg in CONFIGS.items():
        comp = build(cfg)
        lens = lengths_for(comp)
        meta["lengths"][name] = lens
        taps = comp._truncated_filts
        tables[f"{name}/window"] = np.asarray(comp._window, dtype=np.float64)
        tables[f"{name}/starts"] = np.asarray(comp._filt_start_idxs, dtype=np.int64)
        tables[f"{name}/tap_offsets"] = np.cumsum([0] + [len(t) for t in taps]).astype(np.int64)
        tables[f"{name}/taps"] = np.concatenate([np.asarray(t) for t in taps])
        tables[f"{name}/dims"] = np.asarray(
            [comp.frame_length, comp.frame_shift, comp._dft_size, comp.num_coeffs,
             int(comp.bank.is_real), int(comp.frame_style == "centered"),
             int(comp.kaldi_shift), int(comp.includes_energy), int(bool(comp._power)),
             int(bool(comp._log))], dtype=np.int64)
        tables[f"{name}/supports"] = np.asarray(comp.bank.supports, dtype=np.float64)
        tables[f"{name}/supports_hz"] = np.asarray(comp.bank.supports_hz, dtype=np.float64)
        for n in lens:
            for dt in ("f4", "f8"):
                x = sig[:n].astype(dt)
                y = comp.compute_full(x)
                assert y.dtype == x.dtype
                stft[f"{name}/{n}/{dt}"] = y
        # streaming: fixed chunkings of a 5L signal
        n = 5 * comp.frame_length
        x = sig[100 : 100 + n].astype("f4")
        for tag, chunks in (("c7", [7] * (n // 7 + 1)), ("c1024", [1024] * (n // 1024 + 1)),
                            ("mixed", [1, 0, 3 * comp.frame_length, 50, 1, 10 ** 6])):
            pieces, pos = [], 0
            for c in chunks:
                if pos >= n:
                    break
                pieces.append(comp.compute_chunk(x[pos : pos + c]))
                pos += c
            pieces.append(comp.finalize())
            stream[f"{name}/{tag}"] = np.concatenate(pieces)
        stream[f"{name}/full"] = comp.compute_full(x)
    # image of a short stream (the reference's chunked and full paths disagree below L//2+1)
    comp = build(CONFIGS["c1_kaldi_fbank"])
    x = sig[:150].astype("f4")
    stream["short150/chunked"] = np.concatenate([comp.compute_chunk(x), comp.finalize()])
    stream["short150/full"] = comp.compute_full(x)

    np.savez_compressed(os.path.join(HERE, "signals.npz"), master=sig)
    np.savez_compressed(os.path.join(HERE, "tables.npz"), **tables)
    np.savez_compressed(os.path.join(HERE, "stft.npz"), **stft)
    np.savez_compressed(os.path.join(HERE, "stream.npz"), **stream)

    # ---- post-processors ---------------------------------------------------------------
    post = {}
    rng = np.random.default_rng(77)
    for T in (1, 3, 50):
        x = (rng.standard_normal((T, 7)) * 5 + 2).astype("f4")
        post[f"deltas/in/T{T}"] = x
        for nd in (1, 2):
            for W in (2, 3):
                y = rpost.Deltas(nd, context_window=W, target_axis=1).apply(x, axis=0)
                post[f"deltas/out/T{T}/n{nd}/w{W}"] = y
    x = rng.standard_normal((4, 9, 5))
    post["deltas/in/nd3"] = x
    post["deltas/out/nd3/axis1_stack0"] = rpost.Deltas(2, concatenate=False, target_axis=0).apply(x, axis=1)
    post["deltas/out/nd3/axis2_cat1"] = rpost.Deltas(1, target_axis=1).apply(x, axis=2)
    post["deltas/out/nd3/axis0_reflect"] = rpost.Deltas(2, target_axis=-1, pad_mode="reflect").apply(x[:, :, :1].repeat(3, 2)[:4], axis=1)
    x = (rng.standard_normal((50, 6)) * np.arange(1, 7) + np.arange(6)).astype("f4")
    post["cmvn/in"] = x
    post["cmvn/out/local"] = rpost.Standardize().apply(x, axis=-1)
    post["cmvn/out/local_novar"] = rpost.Standardize(norm_var=False).apply(x, axis=1)
    post["cmvn/out/local_axis0"] = rpost.Standardize().apply(x.T.copy(), axis=0)
    st = rpost.Standardize()
    st.accumulate(x[:20])
    st.accumulate(x[20:45])
    st.accumulate(x[45])
    post["cmvn/stats"] = st._stats.copy()
    post["cmvn/out/global"] = st.apply(x)
    post["cmvn/out/global_vec"] = st.apply(x[3])
    x3 = rng.standard_normal((3, 4, 5))
    post["cmvn/in3"] = x3
    post["cmvn/out/in3_axis1"] = rpost.Standardize().apply(x3, axis=1)
    np.savez_compressed(os.path.join(HERE, "post.npz"), **post)

    # ---- pre-processors -------------------------------------------------------------------
    pre = {}
    comp = build(CONFIGS["c1_readme_fbank"])
    for dt in ("f4", "f8"):
        x = sig[500:4500].astype(dt)
        pe = rpre.Preemphasize(0.97).apply(x)
        assert pe.dtype == x.dtype
        pre[f"preemph/out/{dt}"] = pe
        pre[f"preemph/stft/{dt}"] = comp.compute_full(pe)
    x2 = sig[:600].reshape(3, 200)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", DeprecationWarning)
        pre["preemph/out/2d_axis0"] = rpre.Preemphasize(0.5).apply(x2.copy(), axis=0)
    pre["preemph/out/2d_last"] = rpre.Preemphasize(0.5).apply(x2.copy())
    comp5 = build(CONFIGS["c5_gammatone64_48k"])
    x = sig[:9000].astype("f4")
    pre["preemph/stft/c5"] = comp5.compute_full(rpre.Preemphasize(0.9).apply(x))
    compk = build(CONFIGS["v_fbank_8k_bartlett"])
    pre["preemph/stft/n256"] = compk.compute_full(rpre.Preemphasize(0.97).apply(sig[:3000].astype("f4")))
    np.savez_compressed(os.path.join(HERE, "pre.npz"), **pre)

    # ---- the reference's own known-answer fixtures -------------------------------------
    data = os.path.join(REF, "tests", "data")
    noise = load_numpy_pickle(os.path.join(data, "noise.pkl"))
    feats = load_numpy_pickle(os.path.join(data, "kaldi_feats.pkl"))
    pairs = load_numpy_pickle(os.path.join(data, "kaldi_filts.pkl"))
    assert noise.shape == (2000,) and noise.dtype == np.float32
    assert feats.shape == (13, 40) and len(pairs) == 40
    offsets = np.asarray([p[0] for p in pairs], dtype=np.int64)
    filts = [p[1] for p in pairs]
    np.savez_compressed(
        os.path.join(HERE, "kaldi.npz"), noise=noise, kaldi_feats=feats,
        filt_offsets=offsets, filt_lens=np.asarray([len(f) for f in filts], dtype=np.int64),
        filt_vals=np.concatenate(filts),
    )
    with open(os.path.join(HERE, "configs.json"), "w") as fh:
        json.dump(meta, fh, indent=1, sort_keys=True)
    total = sum(os.path.getsize(os.path.join(HERE, f)) for f in os.listdir(HERE))
    print("golden fixtures written, total bytes:", total)


if __name__ == "__main__":
    main()
