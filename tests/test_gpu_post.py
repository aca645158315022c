"""Parity of the Deltas / Standardize kernels with reference outputs and the oracle"""
import numpy as np
import pytest

from pydrobert_speech_amd.post import Deltas, Standardize
from oracle import stft_oracle as orc

pytestmark = pytest.mark.gpu


def test_deltas_match_reference_outputs(golden_post):
    for T in (1, 3, 50):
        x = golden_post[f"deltas/in/T{T}"]
        for nd in (1, 2):
            for W in (2, 3):
                got = Deltas(nd, context_window=W, target_axis=1).apply(x, axis=0)
                want = golden_post[f"deltas/out/T{T}/n{nd}/w{W}"]
                assert got.dtype == want.dtype and got.shape == want.shape
                assert np.allclose(got, want, rtol=1e-6, atol=1e-6)
    x = golden_post["deltas/in/nd3"]
    got = Deltas(2, concatenate=False, target_axis=0).apply(x, axis=1)
    assert got.dtype == np.float64
    assert np.allclose(got, golden_post["deltas/out/nd3/axis1_stack0"], rtol=1e-12, atol=1e-12)
    got = Deltas(1, target_axis=1).apply(x, axis=2)
    assert np.allclose(got, golden_post["deltas/out/nd3/axis2_cat1"], rtol=1e-12, atol=1e-12)
    xr = x[:, :, :1].repeat(3, 2)[:4]
    got = Deltas(2, target_axis=-1, pad_mode="reflect").apply(xr, axis=1)
    assert np.allclose(got, golden_post["deltas/out/nd3/axis0_reflect"], rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("concatenate", [True, False])
@pytest.mark.parametrize("num_deltas", range(4))
def test_delta_shapes(concatenate, num_deltas):
    # reference tests/test_post.py:111-133
    rng = np.random.default_rng(0)
    for buff in (rng.random(10), rng.random((2, 5)), rng.random((3, 6, 4)), rng.random((5, 4, 0, 0, 1))):
        for target_axis in range(buff.ndim + 1 - int(concatenate)):
            d = Deltas(num_deltas, concatenate=concatenate, target_axis=target_axis)
            for axis in range(buff.ndim):
                shape = list(buff.shape)
                if concatenate:
                    shape[target_axis] *= num_deltas + 1
                else:
                    shape.insert(target_axis, num_deltas + 1)
                assert d.apply(buff, axis=axis).shape == tuple(shape)


@pytest.mark.parametrize("dtype", [np.float64, np.float32, np.int32, np.int16])
@pytest.mark.parametrize("window", range(1, 6))
@pytest.mark.parametrize("num_deltas", range(5))
def test_deltas_vs_oracle(dtype, window, num_deltas):
    # reference tests/test_post.py:179-193 (Kaldi comparison), oracle in Kaldi's role
    rng = np.random.default_rng(window * 10 + num_deltas)
    for shape in ((1, 3), (3, 1), (20, 50)):
        buff = (rng.random(shape) * 100).astype(dtype)
        got = Deltas(num_deltas, context_window=window, target_axis=1).apply(buff, axis=0)
        want = orc.deltas(buff, axis=0, num_deltas=num_deltas, context_window=window, target_axis=1)
        assert got.dtype == want.dtype and np.allclose(got, want)


def test_cmvn_matches_reference_outputs(golden_post):
    x = golden_post["cmvn/in"]
    for got, key in (
        (Standardize().apply(x, axis=-1), "local"),
        (Standardize(norm_var=False).apply(x, axis=1), "local_novar"),
        (Standardize().apply(x.T.copy(), axis=0), "local_axis0"),
    ):
        assert got.dtype == np.float64
        assert np.allclose(got, golden_post[f"cmvn/out/{key}"], rtol=1e-10, atol=1e-10), key
    st = Standardize()
    st.accumulate(x[:20])
    st.accumulate(x[20:45])
    st.accumulate(x[45])
    assert np.allclose(st._stats, golden_post["cmvn/stats"], rtol=1e-12)
    assert np.allclose(st.apply(x), golden_post["cmvn/out/global"], rtol=1e-10, atol=1e-10)
    assert np.allclose(st.apply(x[3]), golden_post["cmvn/out/global_vec"], rtol=1e-10, atol=1e-10)
    x3 = golden_post["cmvn/in3"]
    assert np.allclose(Standardize().apply(x3, axis=1), golden_post["cmvn/out/in3_axis1"], rtol=1e-10, atol=1e-10)


@pytest.mark.parametrize("dtype", [np.float64, np.float32, np.int32, np.int16])
@pytest.mark.parametrize("norm_var", [True, False])
def test_standardize_local_properties(norm_var, dtype):
    # reference tests/test_post.py:16-52
    rng = np.random.default_rng(3)
    for shape in ((100, 1), (5, 5), (10, 4, 3)):
        buff = (rng.random(shape) * 37 + 4).astype(dtype)
        stand = Standardize(norm_var=norm_var)
        for axis in range(buff.ndim):
            other = tuple(i for i in range(buff.ndim) if i != axis)
            if sum(buff.shape[i] for i in other) == len(other):
                continue
            b2 = buff.copy()
            s1 = [0] * buff.ndim
            s2 = [-1] * buff.ndim
            s1[axis] = s2[axis] = slice(None)
            b2[tuple(s1)] = b2[tuple(s2)] - 1
            out = stand.apply(b2, axis=axis)
            assert out.dtype == np.float64
            assert np.allclose(out.mean(axis=other), 0)
            if norm_var:
                assert np.allclose(out.var(axis=other), 1)


def test_standardize_errors_and_warnings():
    with pytest.raises(ValueError, match="Cannot apply to empty array"):
        Standardize().apply(np.zeros((0, 3)))
    with pytest.raises(ValueError, match="Cannot accumulate from empty array"):
        Standardize().accumulate(np.zeros((0, 3)))
    with pytest.raises(ValueError, match="Unable to standardize the variance"):
        Standardize().apply(np.ones(4))
    with pytest.warns(UserWarning, match="Standardizing a single vector to 0"):
        assert not Standardize(norm_var=False).apply(np.ones(4)).any()
    with pytest.warns(UserWarning, match="0 variance encountered"):
        out = Standardize().apply(np.ones((5, 2)))
    assert not out.any()
    st = Standardize()
    st.accumulate(np.ones((4, 3)))
    with pytest.raises(ValueError, match="Expected feature vector of length 3; got 2"):
        st.apply(np.ones((4, 2)))


def test_rows_variants_match_per_utterance_apply():
    import torch

    rng = np.random.default_rng(9)
    lens = [1, 7, 250, 3, 1000]
    rows = np.concatenate([[0], np.cumsum(lens)])
    feats = (rng.standard_normal((rows[-1], 24)) * 3 + 1).astype("f4")
    d_feats = torch.from_numpy(feats).cuda()
    got = Deltas(2).apply_rows(d_feats, rows).cpu().numpy()
    for b in range(len(lens)):
        want = orc.deltas(feats[rows[b] : rows[b + 1]], axis=0, num_deltas=2, target_axis=1)
        assert np.allclose(got[rows[b] : rows[b + 1]], want, rtol=1e-6, atol=1e-6)
    lens2 = [2, 7, 250, 3, 1000]
    rows2 = np.concatenate([[0], np.cumsum(lens2)])
    feats2 = (rng.standard_normal((rows2[-1], 70)) * 3 + 1).astype("f4")
    st = Standardize()
    got = st.apply_rows(torch.from_numpy(feats2).cuda(), rows2).cpu().numpy()
    assert got.dtype == np.float64
    for b in range(len(lens2)):
        want = orc.cmvn_local(feats2[rows2[b] : rows2[b + 1]], axis=-1)
        assert np.allclose(got[rows2[b] : rows2[b + 1]], want, rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("K,W", [(1, 1), (1, 2), (2, 1), (2, 2), (2, 3), (3, 2), (1, 4), (3, 3), (4, 2)])
def test_rows_deltas_every_filter_family_member_is_exact(K, W):
    # (3, 3) and (4, 2) have no register-window instantiation and take the LDS-tiled kernel;
    # float64 accumulation in the oracle's order => identical float32 results
    import torch

    rng = np.random.default_rng(100 * K + W)
    lens = [1, 2, 8, 9, 2 * K * W + 1, 8 * 13, 8 * 13 + 1, 333, 7]
    rows = np.concatenate([[0], np.cumsum(lens)])
    F = 81
    feats = (rng.standard_normal((rows[-1], F)) * 3 + 1).astype("f4")
    d = Deltas(K, context_window=W)
    got = d.apply_rows(torch.from_numpy(feats).cuda(), rows).cpu().numpy()
    assert got.shape == (rows[-1], (K + 1) * F)
    for b in range(len(lens)):
        want = orc.deltas(feats[rows[b] : rows[b + 1]], axis=0, num_deltas=K, context_window=W, target_axis=1)
        assert np.array_equal(got[rows[b] : rows[b + 1]], want), (K, W, lens[b])


def test_rows_deltas_c_abi_with_filters_outside_the_family():
    # the C entry point takes any odd-length filters: lengths that do not match 2 k W + 1 run
    # the kernel's generic loop (K = 2, halo 4 would be W = 2 with lengths 5 and 9)
    import ctypes

    import torch

    from pydrobert_speech_amd import _native

    lib = _native.lib()
    rng = np.random.default_rng(77)
    filts = [rng.standard_normal(3), rng.standard_normal(9)]
    lens = [5, 1, 40, 17]
    rows = np.concatenate([[0], np.cumsum(lens)])
    F = 13
    feats = rng.standard_normal((rows[-1], F)).astype("f4")
    d_in = torch.from_numpy(feats).cuda()
    d_out = torch.full((rows[-1], 3 * F), float("nan"), dtype=torch.float32, device="cuda")
    d_filts = torch.from_numpy(np.concatenate(filts)).cuda()
    d_offs = torch.tensor([0, 3, 12], dtype=torch.int32, device="cuda")
    d_rows = torch.from_numpy(rows[:-1].astype(np.int64)).cuda()
    d_n = torch.tensor(lens, dtype=torch.int64, device="cuda")
    rc = lib.pds_deltas_rows_f32(d_in.data_ptr(), F, d_rows.data_ptr(), d_n.data_ptr(), len(lens), max(lens), F,
                                 d_filts.data_ptr(), d_offs.data_ptr(), 2, 4, d_out.data_ptr(), 3 * F,
                                 ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    _native.check(rc, "pds_deltas_rows")
    got = d_out.cpu().numpy()
    for b, T in enumerate(lens):
        x = feats[rows[b] : rows[b + 1]].astype(np.float64)
        assert np.array_equal(got[rows[b] : rows[b + 1], :F], feats[rows[b] : rows[b + 1]])
        for k, f in enumerate(filts, 1):
            M = (len(f) - 1) // 2
            acc = np.zeros_like(x)
            for j, w in enumerate(f):
                acc += w * x[np.clip(np.arange(T) + j - M, 0, T - 1)]
            assert np.array_equal(got[rows[b] : rows[b + 1], k * F : (k + 1) * F], acc.astype("f4")), (b, k)


def test_statics_and_deltas_share_one_buffer():
    # the pipeline layout of BASELINE.json configs[2]: the STFT kernel writes the statics with a
    # row stride that leaves room for the deltas, which are then added in place
    import json

    import torch

    from pydrobert_speech_amd.alias import alias_factory_subclass_from_arg
    from pydrobert_speech_amd.compute import FrameComputer

    comp = alias_factory_subclass_from_arg(FrameComputer, json.loads(json.dumps(
        {"name": "stft", "bank": {"name": "fbank", "num_filts": 80}, "frame_length_ms": 25,
         "include_energy": True, "use_power": True})))
    rng = np.random.default_rng(12)
    lens = [16000, 4000, 250, 9000]
    sigs = [(3000 * rng.standard_normal(n)).astype("f4") for n in lens]
    offs = np.concatenate([[0], np.cumsum(lens)])
    x = torch.from_numpy(np.concatenate(sigs)).cuda()
    layout = comp.prepare_layout(offs[:-1], lens)
    C = comp.num_coeffs
    buf = torch.full((layout.total_rows, 3 * C), float("nan"), dtype=torch.float32, device="cuda")
    comp.launch(x, layout, out=buf)
    d = Deltas(2)
    res = d.apply_rows(buf[:, :C], layout.row_offsets, out=buf)
    assert res.data_ptr() == buf.data_ptr() and torch.isfinite(buf).all()
    got = buf.cpu().numpy()
    want_static = comp.compute_full_batch(sigs)
    for b in range(len(lens)):
        lo, hi = layout.row_offsets[b], layout.row_offsets[b + 1]
        assert np.array_equal(got[lo:hi, :C], want_static[b])
        want = orc.deltas(want_static[b], axis=0, num_deltas=2, target_axis=1)
        assert np.allclose(got[lo:hi], want, rtol=1e-6, atol=1e-6)
    # second call with the same geometry reuses the cached row description
    d.apply_rows(buf[:, :C], layout.row_offsets, out=buf)


def test_hip_graph_capture_of_the_batch_launch():
    # launch() allocates nothing and never synchronises: it can be captured into a HIP graph
    import json

    import torch

    from pydrobert_speech_amd.alias import alias_factory_subclass_from_arg
    from pydrobert_speech_amd.compute import FrameComputer

    comp = alias_factory_subclass_from_arg(FrameComputer, {"name": "stft", "bank": "fbank", "use_power": True})
    n, B = 16000, 8
    x = 3000 * torch.randn(B * n, device="cuda")
    layout = comp.prepare_layout(np.arange(B) * n, np.full(B, n))
    out = torch.empty((layout.total_rows, comp.num_coeffs), device="cuda")
    comp.launch(x, layout, out=out)  # warm up (plan creation, function attributes)
    eager = out.clone()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        comp.launch(x, layout, out=out)
    x.mul_(2.0)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.allclose(out, eager + 2 * np.log(2.0), atol=2e-4)


# ---- section 8(f): statistics files and Stack ---------------------------------------------


@pytest.fixture(scope="module")
def gio():
    import os

    from tests.conftest import GOLDEN

    with np.load(os.path.join(GOLDEN, "io.npz")) as z:
        return {k: z[k] for k in z.files}


@pytest.mark.parametrize("key,name,kwargs", [
    ("cmvn_file/out_npy", "cmvn_stats.npy", {}),
    ("cmvn_file/out_f64", "cmvn_stats.f64", {"force_as": "file"}),
    ("cmvn_file/out_f32", "cmvn_stats.f32", {"force_as": "file"}),
    ("cmvn_file/out_novar", "cmvn_stats.npy", {"norm_var": False}),
])
def test_cmvn_with_statistics_from_a_file(gio, key, name, kwargs):
    import os

    from tests.conftest import GOLDEN

    st = Standardize(os.path.join(GOLDEN, name), **kwargs)
    got = st.apply(gio["cmvn_file/in"])
    assert got.dtype == np.float64
    assert np.allclose(got, gio[key], rtol=1e-9, atol=1e-9)


def test_stack_on_gpu_tensors_matches_reference(gio):
    import torch

    from pydrobert_speech_amd.post import Stack

    x2, x3 = torch.from_numpy(gio["stack/in2"]).cuda(), torch.from_numpy(gio["stack/in3"]).cuda()
    for key, x, kwargs, axis in [
        ("stack/out2/nv3", x2, dict(num_vectors=3), 1),
        ("stack/out2/nv3_edge", x2, dict(num_vectors=3, pad_mode="edge"), 1),
        ("stack/out2/nv4_const", x2, dict(num_vectors=4, pad_mode="constant"), -1),
        ("stack/out3/nv3_t1_a2", x3, dict(num_vectors=3, time_axis=1), 2),
        ("stack/out3/nv2_tm1_a1", x3, dict(num_vectors=2, time_axis=-1), 1),
    ]:
        got = Stack(**kwargs).apply(x, axis=axis)
        assert got.is_cuda and np.array_equal(got.cpu().numpy(), gio[key]), key
    with pytest.raises(ValueError):
        Stack(4, time_axis=1, pad_mode="reflect").apply(x3, axis=0)


@pytest.mark.parametrize("pad_mode", [None, "constant", "edge"])
@pytest.mark.parametrize("nv", [1, 3, 4])
def test_stack_rows_matches_per_utterance_apply(nv, pad_mode):
    import torch

    from pydrobert_speech_amd.post import Stack

    rng = np.random.default_rng(nv)
    lens = [0, 1, nv, nv + 1, 250, 7, 1000]
    rows = np.concatenate([[0], np.cumsum(lens)])
    F = 41
    feats = rng.standard_normal((rows[-1], F)).astype("f4")
    st = Stack(nv, pad_mode=pad_mode)
    got, new_rows = st.apply_rows(torch.from_numpy(feats).cuda(), rows)
    got = got.cpu().numpy()
    assert got.shape[1] == nv * F and new_rows[-1] == got.shape[0]
    for b in range(len(lens)):
        want = st.apply(feats[rows[b] : rows[b + 1]], axis=1)
        assert np.array_equal(got[new_rows[b] : new_rows[b + 1]], want), (b, lens[b])
    # strided input rows (a column slice of a wider buffer)
    wide = torch.from_numpy(np.concatenate([feats, feats], 1)).cuda()
    got2, _ = st.apply_rows(wide[:, :F], rows)
    assert np.array_equal(got2.cpu().numpy(), got)


def _ramp_to_first(vector, pad_width, iaxis, kwargs):
    """the callable pad mode of tests/golden/make_golden_post2.py (values depend on the pad width)"""
    lo, hi = pad_width
    if lo:
        vector[:lo] = vector[lo] * np.arange(lo, 0, -1) / (lo + 1)
    if hi:
        vector[-hi:] = vector[-hi - 1] + np.arange(1, hi + 1) * 0.25


@pytest.mark.parametrize("name,kwargs", [
    ("linear_ramp", dict(pad_mode="linear_ramp", end_values=(1.5, -2.0))),
    ("mean_stat2", dict(pad_mode="mean", stat_length=2)),
    ("maximum", dict(pad_mode="maximum")),
    ("constant_tenth", dict(pad_mode="constant", constant_values=0.1)),
    ("reflect_odd", dict(pad_mode="reflect", reflect_type="odd")),
    ("symmetric", dict(pad_mode="symmetric")),
    ("wrap", dict(pad_mode="wrap")),
    ("callable", dict(pad_mode=_ramp_to_first)),
])
def test_deltas_pad_modes_match_reference_outputs(name, kwargs):
    """Every order padded by its own reach, in float64 (reference post.py:470-483)"""
    import os

    import torch

    from tests.conftest import GOLDEN

    with np.load(os.path.join(GOLDEN, "post2.npz")) as z:
        for dt, tol in (("f4", 1e-6), ("f8", 1e-12)):
            x, want = z[f"in/{dt}"], z[f"out/{name}/{dt}"]
            got = Deltas(2, context_window=2, target_axis=-1, **kwargs).apply(x, axis=0)
            assert got.dtype == want.dtype and got.shape == want.shape
            assert np.allclose(got, want, rtol=tol, atol=tol), (name, dt, np.abs(got - want).max())
            on_gpu = Deltas(2, context_window=2, target_axis=-1, **kwargs).apply(torch.from_numpy(x).cuda(), axis=0)
            assert on_gpu.is_cuda and np.allclose(on_gpu.cpu().numpy(), want, rtol=tol, atol=tol)


@pytest.mark.parametrize("K", [1, 2])
@pytest.mark.parametrize("bank", ["fbank80_energy", "mel40", "mel64_1024_energy", "mel96_energy", "mel26_energy"])
def test_fused_statics_and_deltas_launch(K, bank):
    """pds_stft_deltas_batch_f32 (one launch: every wave walks a stretch of frames and differentiates them
    from the coefficients in its registers) against the two launches, on a ragged batch with empty,
    one-frame and long utterances: statics bit for bit, deltas (float32 accumulation against float64)
    within a few float32 ulps of the statics' magnitude"""
    import torch

    from pydrobert_speech_amd.alias import alias_factory_subclass_from_arg
    from pydrobert_speech_amd.compute import FrameComputer

    cfg = {
        "fbank80_energy": {"name": "stft", "bank": {"name": "fbank", "num_filts": 80}, "frame_length_ms": 25,
                           "include_energy": True, "use_power": True},
        "mel40": {"name": "stft", "bank": {"name": "tri", "scaling_function": "mel", "num_filts": 40},
                  "frame_length_ms": 25, "frame_shift_ms": 10, "use_power": True},
        "mel64_1024_energy": {"name": "stft", "bank": {"name": "tri", "scaling_function": "mel", "num_filts": 64,
                                                       "sampling_rate": 48000},
                              "frame_length_ms": 20, "include_energy": True, "use_power": True},
        # rows wider than 64 lanes x 16 bytes (97 coefficients: 291 columns), and a row width of 4 n + 1 (27: 81)
        "mel96_energy": {"name": "stft", "bank": {"name": "fbank", "num_filts": 96}, "frame_length_ms": 25,
                         "include_energy": True, "use_power": True},
        "mel26_energy": {"name": "stft", "bank": {"name": "fbank", "num_filts": 26}, "frame_length_ms": 25,
                         "include_energy": True, "use_power": True},
    }[bank]
    comp = alias_factory_subclass_from_arg(FrameComputer, cfg)
    assert comp._native_plan().has_fused_deltas
    rng = np.random.default_rng(K)
    S = comp.frame_shift
    C = comp.num_coeffs
    deltas = Deltas(K)

    def check(lens, what):
        x = torch.from_numpy((3000 * rng.standard_normal(int(np.sum(lens)))).astype("f4")).cuda()
        offs = np.concatenate([[0], np.cumsum(lens)[:-1]])
        layout = comp.prepare_layout(offs, lens, device=x.device)
        fused = torch.full((layout.total_rows, (K + 1) * C), float("nan"), device="cuda")
        got = comp.launch_with_deltas(x, layout, deltas, out=fused, fused=True)
        assert got.shape == (layout.total_rows, (K + 1) * C)
        two = torch.empty_like(fused)
        comp.launch(x, layout, out=two)
        deltas.apply_rows(two[:, :C], layout.row_offsets, out=two)
        scale = float(two[:, :C].abs().max()) if layout.total_rows else 1.0
        # (the same kernel code computes the statics: bit for bit when the plain launch takes the
        # row-segment walk with the same segment length too, else the two walks' summation orders apart)
        same_walk = bank in ("fbank80_energy", "mel40")
        assert torch.equal(fused[:, :C], two[:, :C]) or (
            not same_walk and float((fused[:, :C] - two[:, :C]).abs().max()) <= 2e-6 * scale), what
        err = (fused[:, C:] - two[:, C:]).abs()
        assert bool(torch.isfinite(fused).all()) and float(err.max() if err.numel() else 0.0) <= 4e-6 * max(scale, 1.0), (
            what, float(err.max()), scale)
        return x, layout, two

    lens = [0, S, 5 * S, 9 * S + 3, 160000, 33 * S, 1, 4 * S, 57000, 12 * S, 100 * S + 7, 8 * S, 2 * S]
    x, layout, two = check(lens, "ragged")
    # into a wider buffer: nothing beyond the (K + 1) C columns is touched
    wide = torch.full((layout.total_rows, (K + 1) * C + 5), -7.0, device="cuda")
    got = comp.launch_with_deltas(x, layout, deltas, out=wide, fused=True)
    assert torch.allclose(got[:, :C], two[:, :C], rtol=1e-5, atol=1e-5) and bool((wide[:, (K + 1) * C :] == -7.0).all())
    # many short utterances (pieces of one or two chunks, empty utterances in between), few long ones
    # (a stretch per wave inside one utterance), and a batch of equal lengths
    check(rng.integers(0, 40 * S, size=1500), "many short")
    check([700 * S + 5, 0, 333 * S, 1000 * S], "few long")
    check([50 * S] * 64, "equal")


def test_launch_with_deltas_falls_back_to_two_launches():
    # a context window the fused kernel does not have, and float64 samples
    import torch

    from pydrobert_speech_amd.alias import alias_factory_subclass_from_arg
    from pydrobert_speech_amd.compute import FrameComputer

    comp = alias_factory_subclass_from_arg(FrameComputer, {"name": "stft", "bank": "fbank", "frame_length_ms": 25})
    rng = np.random.default_rng(3)
    lens = [4000, 801, 16000]
    x = torch.from_numpy((3000 * rng.standard_normal(sum(lens))).astype("f4")).cuda()
    layout = comp.prepare_layout(np.concatenate([[0], np.cumsum(lens)[:-1]]), lens, device=x.device)
    C = comp.num_coeffs
    d3 = Deltas(2, context_window=3)
    got = comp.launch_with_deltas(x, layout, d3, fused=True)
    want = torch.empty_like(got)
    comp.launch(x, layout, out=want)
    d3.apply_rows(want[:, :C], layout.row_offsets, out=want)
    assert torch.equal(got, want)


@pytest.mark.parametrize("K", [1, 2])
@pytest.mark.parametrize("bank", ["fbank80_energy", "mel64_1024_energy"])
@pytest.mark.parametrize("flow", ["f32+preemph", "f64", "f64+preemph", "i16", "i16+preemph"])
def test_fused_statics_and_deltas_launch_with_preemphasis_and_float64_samples(K, bank, flow, monkeypatch):
    """pds_stft_deltas_batch: the reference drivers' chain float64 audio -> Preemphasize -> compute_full -> Deltas
    (command_line.py:345-350) as ONE launch, against the separate launches of the same kernels' plain forms (statics
    within the feature tolerance of each other: the one-launch kernel regenerates its twiddles for float64 samples, the
    deltas within a few float32 ulps of the statics) and against the oracle on the pre-emphasised signal"""
    import torch

    from oracle import stft_oracle as orc
    from pydrobert_speech_amd import config
    from pydrobert_speech_amd.alias import alias_factory_subclass_from_arg
    from pydrobert_speech_amd.compute import FrameComputer
    from tests.test_gpu_stft import _params_from_computer

    cfg = {
        "fbank80_energy": {"name": "stft", "bank": {"name": "fbank", "num_filts": 80}, "frame_length_ms": 25,
                           "include_energy": True, "use_power": True},
        "mel64_1024_energy": {"name": "stft", "bank": {"name": "tri", "scaling_function": "mel", "num_filts": 64,
                                                       "sampling_rate": 48000},
                              "frame_length_ms": 20, "include_energy": True, "use_power": True},
    }[bank]
    monkeypatch.setattr(config, "FLOAT64_ARITHMETIC", "float32")
    comp = alias_factory_subclass_from_arg(FrameComputer, cfg)
    plan = comp._native_plan()
    assert plan.has_fused_deltas and plan.has_f64in
    coeff = 0.97 if "preemph" in flow else 0.0
    dt = "f8" if flow.startswith("f64") else "i2" if flow.startswith("i16") else "f4"
    rng = np.random.default_rng(7 + K)
    S, C = comp.frame_shift, comp.num_coeffs
    deltas = Deltas(K)
    lens = [0, S, 5 * S, 9 * S + 3, 60000, 33 * S, 1, 4 * S, 12 * S, 100 * S + 7, 2 * S]
    host = np.clip(np.rint(3000 * rng.standard_normal(int(np.sum(lens)))), -32768, 32767).astype(dt) if dt == "i2" else (
        3000 * rng.standard_normal(int(np.sum(lens)))).astype(dt)
    x = torch.from_numpy(host).cuda()
    offs = np.concatenate([[0], np.cumsum(lens)[:-1]])
    layout = comp.prepare_layout(offs, lens, device=x.device)
    # poison the paths a silent fallback would take: the separate pre-emphasis pass and dtype conversions
    one = torch.full((layout.total_rows, (K + 1) * C), float("nan"), device="cuda")
    with monkeypatch.context() as m:
        m.setattr(torch.Tensor, "to", lambda *a, **k: (_ for _ in ()).throw(AssertionError("conversion pass")))
        got = comp.launch_with_deltas(x, layout, deltas, out=one, fused=True, preemphasis=coeff)
    assert got.shape == (layout.total_rows, (K + 1) * C) and bool(torch.isfinite(one).all())
    two = torch.empty_like(one)
    comp.launch(x, layout, out=two, preemphasis=coeff)
    deltas.apply_rows(two[:, :C], layout.row_offsets, out=two)
    scale = float(two[:, :C].abs().max())
    assert float((one[:, :C] - two[:, :C]).abs().max()) <= 1e-5 + 1e-4 * scale
    # deltas of the launch's OWN statics, float64-accumulated
    own = one.cpu().numpy()
    p = _params_from_computer(comp)
    for b, n in enumerate(lens):
        rows = own[layout.row_offsets[b] : layout.row_offsets[b + 1]]
        if not len(rows):
            continue
        ref = orc.deltas(rows[:, :C], axis=0, num_deltas=K, target_axis=-1)
        assert np.abs(rows[:, C:] - ref[:, C:]).max() <= 4e-6 * max(scale, 1.0), (b, n)
        sig = host[offs[b] : offs[b] + n].astype("f8")
        if coeff:
            sig = orc.preemphasize(sig, coeff)
        want = orc.compute_full(sig.astype("f4"), p)
        err = np.abs(rows[:, :C] - want)
        # (pre-emphasised noise empties the lowest filters: the float32 floor of DESIGN.md section 2)
        lin = np.abs(np.exp(rows[:, :C].astype("f8")) - np.exp(want.astype("f8")))
        ok = (err <= 2e-5 + 2e-4 * np.abs(want)) | (lin <= 1e-6 * np.exp(want.astype("f8")).max(axis=1, keepdims=True))
        assert ok.all(), (b, n, float(err.max()))


def test_launch_with_deltas_float64_arithmetic_falls_back_to_float64_statics():
    """float64 samples under the default config.FLOAT64_ARITHMETIC = "float64": the statics come from the float64
    kernels and are rounded into the float32 rows (round 2 raised ValueError here)"""
    import torch

    from pydrobert_speech_amd.alias import alias_factory_subclass_from_arg
    from pydrobert_speech_amd.compute import FrameComputer

    comp = alias_factory_subclass_from_arg(FrameComputer, {"name": "stft", "bank": "fbank", "frame_length_ms": 25})
    rng = np.random.default_rng(5)
    lens = [4000, 801, 16000]
    x = torch.from_numpy(3000 * rng.standard_normal(sum(lens))).cuda()
    assert x.dtype == torch.float64
    layout = comp.prepare_layout(np.concatenate([[0], np.cumsum(lens)[:-1]]), lens, device=x.device)
    C = comp.num_coeffs
    d = Deltas(2)
    got = comp.launch_with_deltas(x, layout, d, preemphasis=0.97)
    assert got.dtype == torch.float32 and got.shape == (layout.total_rows, 3 * C)
    want = comp.launch(x, layout, preemphasis=0.97)
    assert want.dtype == torch.float64
    assert torch.equal(got[:, :C], want.to(torch.float32))
    ref = torch.empty_like(got)
    ref[:, :C] = got[:, :C]
    d.apply_rows(ref[:, :C], layout.row_offsets, out=ref)
    assert torch.equal(got, ref)


@pytest.mark.parametrize("bank", ["gammatone64_48k", "gabor64", "fbank80_energy", "mel40"])
@pytest.mark.parametrize("out_dtype", ["f8", "f4"])
def test_cmvn_sums_fused_with_the_stft_launch(bank, out_dtype):
    """pds_stft_cmvn_batch_f32 (the STFT launch adds what it stores to per-wave float64 sums; the normalising kernel
    adds an utterance's pieces in wave order and reads the features once) against launch + CMVN.apply_rows and the
    oracle, on ragged batches with empty, one-frame and long utterances; features bit for bit, sums to 1e-12
    relative, standardised features to the CMVN tolerance; the same call twice gives the same bits"""
    import torch

    from oracle import stft_oracle as orc
    from pydrobert_speech_amd.alias import alias_factory_subclass_from_arg
    from pydrobert_speech_amd.compute import FrameComputer
    from pydrobert_speech_amd.post import CMVN

    cfg = {
        "gammatone64_48k": {"name": "stft", "bank": {"name": "gammatone", "scaling_function": "mel", "num_filts": 64,
                                                     "sampling_rate": 48000}, "frame_length_ms": 20, "use_power": True},
        "gabor64": {"name": "stft", "bank": {"name": "gabor", "scaling_function": "mel", "num_filts": 64},
                    "frame_length_ms": 25, "use_power": True},
        "fbank80_energy": {"name": "stft", "bank": {"name": "fbank", "num_filts": 80}, "frame_length_ms": 25,
                           "include_energy": True, "use_power": True},
        "mel40": {"name": "stft", "bank": {"name": "tri", "scaling_function": "mel", "num_filts": 40},
                  "frame_length_ms": 25, "frame_shift_ms": 10, "use_power": True},
    }[bank]
    comp = alias_factory_subclass_from_arg(FrameComputer, cfg)
    plan = comp._native_plan()
    # (the dense banks' segment walks carry the sums; the 128-register row-segment kernel of the mel banks at N = 512
    # does not: launch_with_cmvn then makes the two calls, and the same checks hold)
    assert plan.has_fused_cmvn == (bank in ("gammatone64_48k", "gabor64"))
    tdt = torch.float64 if out_dtype == "f8" else torch.float32
    rng = np.random.default_rng(11)
    S, C = comp.frame_shift, comp.num_coeffs

    def check(lens, what):
        x = torch.from_numpy((3000 * rng.standard_normal(int(np.sum(lens)))).astype("f4")).cuda()
        offs = np.concatenate([[0], np.cumsum(lens)[:-1]])
        layout = comp.prepare_layout(offs, lens, device=x.device)
        feats = torch.full((layout.total_rows, C), float("nan"), device="cuda")
        fused = comp.launch_with_cmvn(x, layout, CMVN(), feats_out=feats, out_dtype=tdt, fused=True)
        again = comp.launch_with_cmvn(x, layout, CMVN(), out_dtype=tdt, fused=True)
        plain = comp.launch(x, layout)
        # (standardised from the fused launch's OWN features: the plain launch of this plan may be another
        # instantiation -- regenerated twiddles at N = 1024 -- a few float32 ulps apart, which the division by a
        # small standard deviation would magnify past the 1e-9 the sums themselves are held to)
        two = CMVN().apply_rows(feats, layout.row_offsets, out_dtype=tdt)
        assert fused.dtype == tdt and fused.shape == two.shape == (layout.total_rows, C), what
        assert bool(torch.isfinite(fused).all()), what
        assert torch.equal(fused, again), what  # deterministic: pieces are added in a fixed order
        # the features themselves: the stretch-scheduled launch of the same kernel
        assert torch.equal(feats, plain) or float((feats - plain).abs().max()) <= 4e-6 * float(plain.abs().max()), what
        tol = 1e-9 if out_dtype == "f8" else 2e-6
        err = (fused.double() - two.double()).abs().max().item() if fused.numel() else 0.0
        assert err <= tol * max(1.0, float(two.abs().max()) if two.numel() else 1.0), (what, err)
        return layout, feats, fused

    lens = [0, S, 5 * S, 9 * S + 3, 100000, 33 * S, 1, 4 * S, 57000, 12 * S, 100 * S + 7, 8 * S, 2 * S]
    layout, feats, fused = check(lens, "ragged")
    host, got = feats.cpu().numpy(), fused.cpu().numpy()
    for b in (2, 4, 10):
        rows = slice(layout.row_offsets[b], layout.row_offsets[b + 1])
        want = orc.cmvn_local(host[rows], axis=-1)
        assert np.allclose(got[rows], want, rtol=1e-8 if out_dtype == "f8" else 1e-5, atol=1e-8 if out_dtype == "f8" else 2e-5)
    check(rng.integers(0, 40 * S, size=700), "many short")
    check([700 * S + 5, 0, 333 * S, 1000 * S], "few long")
    check([50 * S] * 64, "equal")
