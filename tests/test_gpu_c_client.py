"""The C ABI driven by a plain C program (tests/csrc/c_abi_client.c) -- no Python, no torch on the
calling side: the binding a cgo / JNI / ctypes integration would make -- against the reference's
features for the same signals."""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

from tests.conftest import ROOT, assert_features_close

pytestmark = pytest.mark.gpu


def test_plain_c_client_reproduces_reference_features(tmp_path, golden_meta, golden_stft, master_signal):
    import json

    from pydrobert_speech_amd._native import LIB_PATH
    from pydrobert_speech_amd.alias import alias_factory_subclass_from_arg
    from pydrobert_speech_amd.compute import FrameComputer

    gcc = shutil.which("gcc")
    if gcc is None or not os.path.isdir("/opt/rocm/include"):
        pytest.skip("needs gcc and the ROCm headers")
    exe = str(tmp_path / "c_abi_client")
    libdir = os.path.dirname(LIB_PATH)
    subprocess.run(
        [gcc, "-O1", "-std=c11", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"),
         os.path.join(ROOT, "tests", "csrc", "c_abi_client.c"), "-o", exe, "-L" + libdir, "-lpds_amd",
         "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"],
        check=True, capture_output=True)
    name = "c1_readme_fbank"  # energy + 40 fbank filters
    comp = alias_factory_subclass_from_arg(FrameComputer, json.loads(json.dumps(golden_meta["configs"][name])))
    rp, col, val = comp.bin_weights
    lens = [n for n in golden_meta["lengths"][name] if n]
    sigs = [master_signal[:n].astype("f4") for n in lens]
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
    header = struct.pack(
        "<11i", comp.frame_length, comp.frame_shift, comp.dft_size, comp.pad_left, len(rp) - 1, len(col),
        int(comp._power), int(comp._log), int(comp.includes_energy), len(lens), int(sum(lens)))
    with open(tmp_path / "in.bin", "wb") as fh:
        fh.write(header)
        for arr, dt in ((comp._window, "<f8"), (rp, "<i4"), (col, "<i4"), (val, "<f8"), (offs, "<i8"),
                        (np.asarray(lens), "<i8"), (np.concatenate(sigs), "<f4")):
            fh.write(np.ascontiguousarray(arr, dtype=dt).tobytes())
    res = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "kernel kind 512" in res.stdout
    raw = open(tmp_path / "out.bin", "rb").read()
    rows, C = struct.unpack("<2q", raw[:16])
    feats = np.frombuffer(raw[16:], dtype="<f4").reshape(rows, C)
    want = np.concatenate([golden_stft[f"{name}/{n}/f4"] for n in lens])
    assert feats.shape == want.shape
    assert_features_close(feats, want, rtol=1e-4, atol=1e-5, what="c client")


def test_plain_c_client_of_the_host_feed(tmp_path, golden_meta, golden_tables, master_signal):
    """tests/csrc/c_feed_client.c: 16-bit PCM in host memory through pds_feed_* and back, from a C program that makes no
    HIP call and links nothing but libpds_amd.so -- against the oracle on the same samples"""
    import json

    from oracle import stft_oracle as orc
    from pydrobert_speech_amd._native import LIB_PATH
    from pydrobert_speech_amd.alias import alias_factory_subclass_from_arg
    from pydrobert_speech_amd.compute import FrameComputer
    from tests.conftest import oracle_params

    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("needs gcc")
    exe = str(tmp_path / "c_feed_client")
    libdir = os.path.dirname(LIB_PATH)
    subprocess.run(
        [gcc, "-O1", "-std=c11", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "csrc", "c_feed_client.c"),
         "-o", exe, "-L" + libdir, "-lpds_amd", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"],
        check=True, capture_output=True)
    name = "c2_tri_mel40"
    comp = alias_factory_subclass_from_arg(FrameComputer, json.loads(json.dumps(golden_meta["configs"][name])))
    rp, col, val = comp.bin_weights
    rng = np.random.default_rng(21)
    lens = [16000, 1, 4801, 0, 12000, 333, 8000]
    sigs = [rng.integers(-20000, 20000, size=n).astype("<i2") for n in lens]
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
    header = struct.pack(
        "<11i", comp.frame_length, comp.frame_shift, comp.dft_size, comp.pad_left, len(rp) - 1, len(col),
        int(comp._power), int(comp._log), int(comp.includes_energy), len(lens), int(sum(lens)))
    with open(tmp_path / "in.bin", "wb") as fh:
        fh.write(header)
        for arr, dt in ((comp._window, "<f8"), (rp, "<i4"), (col, "<i4"), (val, "<f8"), (offs, "<i8"),
                        (np.asarray(lens), "<i8"), (np.concatenate(sigs), "<i2")):
            fh.write(np.ascontiguousarray(arr, dtype=dt).tobytes())
    res = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr
    raw = open(tmp_path / "out.bin", "rb").read()
    rows, C = struct.unpack("<2q", raw[:16])
    feats = np.frombuffer(raw[16:], dtype="<f4").reshape(rows, C)
    p = oracle_params(golden_tables, name)
    want = np.concatenate([orc.compute_full(x.astype(np.float64), p) for x in sigs])
    assert feats.shape == want.shape
    assert_features_close(feats, want, rtol=1e-4, atol=1e-5, what="c feed client")
