#!/usr/bin/env python3
"""Benchmark of the STFT filter-bank hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = one pass of the batched compute_full kernel over one batch of synthetic
utterances already resident in HBM.  Workload at N = 1: BASELINE.json configs[1]
(1024 x 10 s, 16 kHz float32, 40 triangular mel filters, Hann, 25/10 ms, log power).
For N > 1 the driver launches one rank per GPU with torch.distributed.run; utterances
shard across ranks (each rank a full 1024-utterance batch: weak scaling), no data-path
collective in the timed region.  A second, separately reported region adds the RCCL
gather of the feature matrices.

Prints ONE JSON line (rank 0) with the fields the driver expects plus `roofline` (HBM
bound, algorithmic bytes / kernel time measured with events on the launch stream) and
`cpu_baseline` (the oracle's vectorised numpy restatement timed on host cores, a bounded
sample; rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
VALU_PEAK_TFLOPS = 157.3  # fp32 vector peak, same table (v_fma_f32 at 2 cycles per wave64 instruction)
# HBM bytes per launch measured with rocprofv3 PMC passes (profiles/run_profile.sh), per workload:
# FETCH_SIZE x 2 (gfx950 counts a wide coalesced read stream at half its bytes) + WRITE_SIZE
PMC_TRAFFIC_FILE = os.path.join(ROOT, "profiles", "pmc_traffic.json")

WORKLOADS = {
    # name: (config, samples per utterance, utterances per GPU, post-processing in the step)
    # BASELINE.json configs[1] -- the headline workload (default)
    "fbank40_16k_25_10_b1024x10s": (
        {"name": "stft", "bank": {"name": "tri", "scaling_function": "mel", "num_filts": 40},
         "frame_length_ms": 25, "frame_shift_ms": 10, "window_function": "hanning",
         "use_power": True},
        160000, 1024, None,
    ),
    # configs[2] per GPU at reduced batch: 80 mel + energy, then Deltas(2) beside the statics
    "fbank80_energy_deltas2_b1024x10s": (
        {"name": "stft", "bank": {"name": "fbank", "num_filts": 80}, "frame_length_ms": 25,
         "include_energy": True, "use_power": True},
        160000, 1024, "deltas2",
    ),
    # configs[3]: complex Gabor bank, 64 filters
    "gabor64_b1024x10s": (
        {"name": "stft", "bank": {"name": "gabor", "scaling_function": "mel", "num_filts": 64},
         "frame_length_ms": 25, "use_power": True},
        160000, 1024, None,
    ),
    # configs[4] per GPU: Gammatone 64 @ 48 kHz, 20 ms frames (N = 1024), per-utterance CMVN
    "gammatone64_48k_cmvn_b256x10s": (
        {"name": "stft", "bank": {"name": "gammatone", "scaling_function": "mel", "num_filts": 64,
                                  "sampling_rate": 48000}, "frame_length_ms": 20, "use_power": True},
        480000, 256, "cmvn",
    ),
    # the headline bank without zero padding (N = L = 400 = 16 x 25)
    "fbank40_nopad400_b1024x10s": (
        {"name": "stft", "bank": {"name": "tri", "scaling_function": "mel", "num_filts": 40},
         "frame_length_ms": 25, "frame_shift_ms": 10, "window_function": "hanning",
         "use_power": True, "pad_to_nearest_power_of_two": False},
        160000, 1024, None,
    ),
    # ... 20 ms frames without zero padding (N = L = 320 = 16 x 20)
    "fbank40_nopad320_b1024x10s": (
        {"name": "stft", "bank": {"name": "tri", "scaling_function": "mel", "num_filts": 40},
         "frame_length_ms": 20, "frame_shift_ms": 10, "window_function": "hanning",
         "use_power": True, "pad_to_nearest_power_of_two": False},
        160000, 1024, None,
    ),
    # ... 30 ms frames without zero padding (N = L = 480 = 16 x 30)
    "fbank40_nopad480_b1024x10s": (
        {"name": "stft", "bank": {"name": "tri", "scaling_function": "mel", "num_filts": 40},
         "frame_length_ms": 30, "frame_shift_ms": 10, "window_function": "hanning",
         "use_power": True, "pad_to_nearest_power_of_two": False},
        160000, 1024, None,
    ),
    # 32 ms frames at 16 kHz (L = 512 = N: every one of the 32 rows of the N = 512 geometry in use)
    "fbank40_16k_32_10_b1024x10s": (
        {"name": "stft", "bank": {"name": "tri", "scaling_function": "mel", "num_filts": 40},
         "frame_length_ms": 32, "frame_shift_ms": 10, "window_function": "hanning", "use_power": True},
        160000, 1024, None,
    ),
    # 8 kHz telephone speech, the usual 25 ms / 10 ms framing: L = 200 -> N = 256 (32 x 8, eight frames per wave)
    "fbank40_8k_25_10_b1024x10s": (
        {"name": "stft", "bank": {"name": "fbank", "num_filts": 40, "sampling_rate": 8000},
         "frame_length_ms": 25, "frame_shift_ms": 10, "use_power": True},
        80000, 1024, None,
    ),
    # 48 kHz audio, 20 ms / 10 ms framing with a mel bank: L = 960 -> N = 1024 (64 x 16, four frames per wave,
    # row-segment walk; PDS_N1024_GEOM=32x32: two frames per wave)
    "fbank80_48k_20_10_b256x10s": (
        {"name": "stft", "bank": {"name": "fbank", "num_filts": 80, "sampling_rate": 48000},
         "frame_length_ms": 20, "frame_shift_ms": 10, "use_power": True},
        480000, 256, None,
    ),
    # 48 kHz audio with the usual 25 ms / 10 ms framing: L = 1200 -> N = 2048 (64 x 32, two frames per wave)
    "fbank80_48k_25_10_b256x10s": (
        {"name": "stft", "bank": {"name": "fbank", "num_filts": 80, "sampling_rate": 48000},
         "frame_length_ms": 25, "frame_shift_ms": 10, "use_power": True},
        480000, 256, None,
    ),
    # long analysis frames (music / audio tagging): 80 mel, 48 kHz, 50 ms frames -> N = 4096, one frame
    # per wavefront
    "fbank80_48k_50_12.5_b256x10s": (
        {"name": "stft", "bank": {"name": "fbank", "num_filts": 80, "sampling_rate": 48000},
         "frame_length_ms": 50, "frame_shift_ms": 12.5, "use_power": True},
        480000, 256, None,
    ),
    # ... 80 ms frames (3840 samples: every one of the 64 rows of the N = 4096 geometry in use)
    "fbank80_48k_80_20_b256x10s": (
        {"name": "stft", "bank": {"name": "fbank", "num_filts": 80, "sampling_rate": 48000},
         "frame_length_ms": 80, "frame_shift_ms": 20, "use_power": True},
        480000, 256, None,
    ),
    # short-integration features, 40 complex Gabor filters (supports up
    # to 380 taps) + energy; compute bound (direct time-domain filtering), so a smaller batch
    "si_gabor40_b64x10s": (
        {"name": "si", "bank": {"name": "gabor", "scaling_function": "mel", "num_filts": 40},
         "include_energy": True, "use_power": True},
        160000, 64, None,
    ),
    # longer supports (48 kHz gammatone bank, ~1300 taps): the 2048-point form of the same kernel
    "si_gammatone40_48k_b32x10s": (
        {"name": "si", "bank": {"name": "gammatone", "scaling_function": "mel", "num_filts": 40,
                                "sampling_rate": 48000}, "use_power": True},
        480000, 32, None,
    ),
}
DEFAULT_WORKLOAD = "fbank40_16k_25_10_b1024x10s"


def _cpu_worker(args):
    """Times the oracle on `count` utterances; runs in a spawned process (numpy only)"""
    cfg_tables, n, count, seed = args
    import numpy as np

    from oracle import stft_oracle as orc

    p = orc.StftParams(**cfg_tables)
    rng = np.random.default_rng(seed)
    x = (3000.0 * rng.standard_normal(n)).astype(np.float32)
    orc.compute_full(x[: n // 8], p)  # warm caches / imports
    t0 = time.perf_counter()
    frames = 0
    for _ in range(count):
        frames += orc.compute_full(x, p).shape[0]
    return frames, time.perf_counter() - t0


def cpu_baseline(comp, n, budget_s=12.0):
    """Oracle throughput on the host: one process per core, bounded sample"""
    import multiprocessing as mp

    import numpy as np

    os.environ.setdefault("OMP_NUM_THREADS", "1")
    os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
    tables = dict(
        frame_length=comp.frame_length, frame_shift=comp.frame_shift, dft_size=comp.dft_size,
        window=np.asarray(comp._window), starts=list(comp._filt_start_idxs),
        taps=[np.asarray(t) for t in comp._truncated_filts], is_real=comp.bank.is_real,
        centered=comp.frame_style == "centered", kaldi_shift=comp.kaldi_shift,
        include_energy=comp.includes_energy, use_power=bool(comp._power), use_log=bool(comp._log),
    )
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 32))
    # ~130 k frames/s/core measured for this oracle => utterances per worker for ~budget_s
    per_worker = max(2, int(budget_s * 100000 / comp.num_frames(n)))
    ctx = mp.get_context("spawn")
    t0 = time.perf_counter()
    with ctx.Pool(cores) as pool:
        res = pool.map(_cpu_worker, [(tables, n, per_worker, 1000 + i) for i in range(cores)])
    wall = time.perf_counter() - t0
    frames = sum(r[0] for r in res)
    slowest = max(r[1] for r in res)
    return {
        "value": frames / slowest,
        "unit": "frames/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{cores} procs x {per_worker} utterances x {n} samples "
                  f"(oracle/stft_oracle.compute_full, vectorised numpy float64; {wall:.1f} s wall)",
        "per_core": frames / slowest / cores,
        "reference_python_loop_frames_per_s_per_core": 3650.0,  # BASELINE.md section 2 (authoring container)
    }


def parity_spot_check(comp, signal, offsets, lengths, layout, out, picks=3, num_deltas=0, preemph=0.0):
    """Rows of the buffer the timed region wrote, against the oracle on the same samples

    First, middle and last utterance of the batch: their signal slices go to the host, the oracle
    (float64 restatement of the reference, oracle/stft_oracle.py) computes their features, and the
    statics columns of `out` must match within the north star's tolerance (1e-5 + 1e-4 |ref|).
    """
    import numpy as np

    from oracle import stft_oracle as orc

    p = orc.StftParams(
        frame_length=comp.frame_length, frame_shift=comp.frame_shift, dft_size=comp.dft_size,
        window=np.asarray(comp._window), starts=list(comp._filt_start_idxs),
        taps=[np.asarray(t) for t in comp._truncated_filts], is_real=comp.bank.is_real,
        centered=comp.frame_style == "centered", kaldi_shift=comp.kaldi_shift,
        include_energy=comp.includes_energy, use_power=bool(comp._power), use_log=bool(comp._log),
    )
    B, C = len(lengths), comp.num_coeffs
    worst_abs = worst_tol = worst_delta = 0.0
    rows = floor_elems = 0
    utts = sorted({0, B // 2, B - 1})[:picks]
    for b in utts:
        x = signal[int(offsets[b]) : int(offsets[b] + lengths[b])].cpu().numpy()
        if x.dtype.kind == "i":  # (int16 PCM: the reference's readers cast it before anything else, util.py:211-234)
            x = x.astype(np.float64)
        if preemph:
            x = orc.preemphasize(x, preemph)
        want = orc.compute_full(x, p)
        r0 = int(layout.row_offsets[b])
        got = out[r0 : r0 + want.shape[0], :C].cpu().numpy().astype(np.float64)
        if got.shape != want.shape or not want.size:
            return {"pass": False, "error": f"utterance {b}: shape {got.shape} against {want.shape}"}
        err = np.abs(got - want)
        if np.isnan(err).any():
            return {"pass": False, "error": f"utterance {b}: NaN"}
        over = err / (1e-5 + 1e-4 * np.abs(want))
        # float32 arithmetic: a coefficient 40 dB and more below its frame's largest one carries the
        # transform's round-off at full size (tools/fuzz_parity.py); such an element
        # counts as inside when its error is below 1e-6 of the frame's largest coefficient, linear domain
        if comp._log:
            lin_got, lin_want = np.exp(got), np.exp(want)
        else:
            lin_got, lin_want = got, want
        floor = (over > 1.0) & (np.abs(lin_got - lin_want) <= 1e-6 * np.abs(lin_want).max(axis=1, keepdims=True))
        floor_elems += int(floor.sum())
        over = np.where(floor, 0.0, over)
        worst_abs = max(worst_abs, float(err.max()))
        worst_tol = max(worst_tol, float(over.max()))
        rows += want.shape[0]
        if num_deltas:
            # the delta columns of the same rows against the oracle's Deltas of the buffer's OWN statics
            # (reference post.py:462-491, float64 accumulation): differences of logs sit near zero, so
            # the tolerance is taken on the statics' scale
            mine = out[r0 : r0 + want.shape[0], : (num_deltas + 1) * C].cpu().numpy()
            ref = orc.deltas(mine[:, :C], axis=0, num_deltas=num_deltas, target_axis=-1)
            derr = np.abs(mine[:, C:].astype(np.float64) - ref[:, C:])
            if np.isnan(derr).any():
                return {"pass": False, "error": f"utterance {b}: NaN in the deltas"}
            # (bounded as tests/test_gpu_post.py bounds them: 4e-6 of the statics' scale -- float32 accumulation
            # against float64 -- not by the feature tolerance, which at statics of ~20 would be 500 x looser)
            worst_delta = max(worst_delta, float((derr / (4e-6 * max(float(np.abs(want).max()), 1.0))).max()))
    res_d = {"deltas_max_err_over_tolerance": worst_delta, "deltas_columns": num_deltas * C} if num_deltas else {}
    # (a line whose statics pass only through the float32-floor rule in more than 0.1 % of the elements fails;
    # pre-emphasised white noise leaves ~0.02 % of them there: the lowest filters of a few frames)
    return {"pass": worst_tol <= 1.0 and worst_delta <= 1.0 and floor_elems <= max(4, rows * C // 1000),
            "utterances": utts, "rows": rows, "coeffs": C,
            "max_abs_err": worst_abs, "max_err_over_tolerance": worst_tol, "tolerance": "1e-5 + 1e-4 |ref|",
            "elements_at_the_float32_floor": floor_elems, "deltas_tolerance": "4e-6 max(|statics|, 1)", **res_d,
            "against": "oracle/stft_oracle.compute_full (float64) on the timed buffer's own input"}


def _hwmon_of_device(dev_index=0):
    """hwmon directory (socket power / shader clock sensors) of the HIP device, matched by its PCI address;
    None when the sysfs files are not there"""
    import glob

    import torch

    try:
        pr = torch.cuda.get_device_properties(dev_index)
        want = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
    except Exception:
        want = None
    found = []
    for card in glob.glob("/sys/class/drm/card*/device"):
        for hw in glob.glob(os.path.join(card, "hwmon", "hwmon*")):
            if os.path.exists(os.path.join(hw, "power1_input")):
                found.append((os.path.basename(os.path.realpath(card)), hw))
    for addr, hw in found:
        if want is not None and addr.lower() == want:
            return hw, True
    return (found[0][1], False) if len(found) == 1 else (None, False)


def power_probe(step, sync, seconds=1.2):
    """The socket power and shader clock the chip holds under back-to-back steps (tools/power_sample.py: the headline
    kernel runs AT the 1400 W cap, at ~1.94 GHz instead of 2.4): a thread samples the device's hwmon files while the
    steps run; the second half of the samples is averaged (the power reading is a moving average)."""
    import threading

    import numpy as np

    hw, matched = _hwmon_of_device()
    if hw is None:
        return None

    def rd(name):
        try:
            with open(os.path.join(hw, name)) as fh:
                return float(fh.read().strip())
        except (OSError, ValueError):
            return None

    samples, stop = [], threading.Event()

    def sampler():
        while not stop.is_set():
            samples.append((rd("power1_input"), rd("freq1_input")))
            time.sleep(0.02)

    th = threading.Thread(target=sampler, daemon=True)
    th.start()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(50):
            step()
        sync()
        n += 50
    elapsed = time.perf_counter() - t0
    stop.set()
    th.join()
    tail = [s for s in samples[len(samples) // 2:] if s[0] is not None]
    if not tail:
        return None
    cap = rd("power1_cap")
    freqs = [s[1] for s in tail if s[1] is not None]
    return {"socket_w": float(np.mean([s[0] for s in tail])) * 1e-6, "cap_w": cap * 1e-6 if cap else None,
            "sclk_mhz": float(np.mean(freqs)) * 1e-6 if freqs else None, "sclk_max_mhz": 2400.0,
            "ms_per_step_during_probe": 1e3 * elapsed / n, "samples": len(tail), "sensor_matched_by_pci_address": matched,
            "source": "hwmon power1_input / freq1_input of the device, sampled every 20 ms over %.1f s of back-to-back steps" % seconds}


def _stdout_to_stderr():
    """Send file descriptor 1 to stderr until the JSON line: libraries (RCCL prints a version
    banner on stdout when its communicator is created) must not add lines to the one the driver
    reads.  Returns the saved descriptor."""
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    return saved


def _emit(saved_fd, line):
    sys.stdout.flush()
    os.dup2(saved_fd, 1)
    os.close(saved_fd)
    print(line, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-power-probe", action="store_true",
                    help="skip the 1.2 s of extra steps under which socket power and shader clock are sampled")
    ap.add_argument("--preroll-ms", type=float, default=60.0,
                    help="untimed launches before the warm-up, for at least this many milliseconds and until "
                         "the step time has settled: the GPU leaves its idle clock state only after ~40 ms "
                         "of continuous work (tools/clock_ramp.py: 0.38 ms per step at first, 0.32 ms from "
                         "step ~100 on), longer after the CPU-baseline leg; 0 disables")
    ap.add_argument("--workload", default=DEFAULT_WORKLOAD, choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=None, help="utterances per GPU (default: workload's)")
    ap.add_argument("--ragged", action="store_true",
                    help="utterance lengths uniform in [1 s, 15 s] at the workload's rate (seed 99; "
                         "SURVEY.md section 8(d)) instead of the workload's fixed length")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64in", "i16in"],
                    help="f64in: float64 samples in HBM, rounded to float32 by the fused kernel as it loads the "
                         "frames (pds_stft_batch_f64in), float32 features -- the reference drivers' dtype flow; "
                         "i16in: int16 PCM in HBM, converted at the frame load (pds_stft_batch_i16in)")
    ap.add_argument("--fused-deltas", action="store_true",
                    help="deltas2 workloads: statics and deltas by one launch (the default where the plan has one)")
    ap.add_argument("--two-launch-deltas", action="store_true",
                    help="deltas2 workloads: the STFT launch followed by the deltas launch (A/B against the fused one)")
    ap.add_argument("--fused-cmvn", action="store_true",
                    help="CMVN workloads: the sums taken by the STFT launch (pds_stft_cmvn_batch_f32) instead of the "
                         "separate statistics + normalisation kernel")
    ap.add_argument("--preemph", type=float, default=0.0,
                    help="pre-emphasis coefficient fused into the frame loads (reference pre.py:140-149 in front of "
                         "compute_full); 0: none")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--generic", action="store_true",
                    help="force the direct-DFT kernel (si workloads: direct time-domain filtering)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL, one GPU per rank) or gloo (rehearsal: every rank on GPU 0, no gather)")
    args = ap.parse_args()

    saved_stdout = _stdout_to_stderr()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            # started without a launcher: become one (a child process -- this process has not touched
            # the GPU -- whose exit code is passed on)
            import socket
            import subprocess

            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0))
                port = sock.getsockname()[1]
            os.dup2(saved_stdout, 1)  # the rank-0 child prints the JSON line on the real stdout
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
                   "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
            sys.exit(subprocess.run(cmd).returncode)
        args.gpus = world

    import numpy as np

    import pydrobert_speech_amd as ps
    from pydrobert_speech_amd.alias import alias_factory_subclass_from_arg

    cfg, n, B, post = WORKLOADS[args.workload]
    if args.batch:
        B = args.batch
    comp = alias_factory_subclass_from_arg(ps.compute.FrameComputer, cfg)

    is_si = isinstance(comp, ps.si.ShortIntegrationFrameComputer)
    cpu = None  # (the CPU-baseline leg runs AFTER the timed region, see below)

    import torch
    import torch.distributed as dist

    if args.backend == "gloo":
        local_rank = 0  # rehearsal of the multi-rank control flow on a one-GPU box
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # PDS_BENCH_FORCE_DIST=1: take the multi-rank code path (process group, barrier, reductions,
    # gather) with a single rank too -- the RCCL rehearsal a one-GPU box allows
    use_dist = world > 1 or os.environ.get("PDS_BENCH_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            try:
                dist.init_process_group("nccl", device_id=dev)
            except Exception as exc:  # RCCL unusable on this node: the hot path has no collective, so time it anyway
                # (barrier and the reductions of the timings over gloo; the optional gather leg is skipped)
                print("bench.py: RCCL process group failed (%r): falling back to gloo for the barriers" % (exc,),
                      file=sys.stderr, flush=True)
                args.backend = "gloo"
        if args.backend != "nccl":
            dist.init_process_group("gloo")
            args.no_gather = True
    red_dev = dev if args.backend == "nccl" else torch.device("cpu")

    # synthetic batch: x = 3000 N(0,1), float32, seeded per rank; generated on the device
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    if args.ragged:
        rate = int(comp.sampling_rate)
        lengths = np.random.default_rng(99 + rank).integers(rate, 15 * rate + 1, size=B).astype(np.int64)
    else:
        lengths = np.full(B, n, dtype=np.int64)
    offsets = np.concatenate([[0], np.cumsum(lengths)[:-1]]).astype(np.int64)
    signal = torch.randn(int(lengths.sum()), generator=g, device=dev, dtype=torch.float32).mul_(3000.0)
    if args.dtype == "f64in":
        if is_si:
            raise SystemExit("--dtype f64in: STFT workloads only")
        signal = signal.double()
        ps.config.FLOAT64_ARITHMETIC = "float32"
    if args.dtype == "i16in":
        if is_si or post not in (None, "deltas2"):
            raise SystemExit("--dtype i16in: STFT workloads, alone or with the one-launch deltas")
        signal = signal.clamp_(-32768, 32767).to(torch.int16)
    if is_si:
        layout = None
        frames = int(sum(comp.num_frames(int(v)) for v in lengths))
    else:
        layout = comp.prepare_layout(offsets, lengths, device=dev)
        frames = layout.total_rows
    C = comp.num_coeffs
    out_cols = 3 * C if post == "deltas2" else C
    out = torch.empty((frames, out_cols), dtype=torch.float32, device=dev)
    deltas = ps.post.Deltas(2) if post == "deltas2" else None
    # (experiment: rows padded to a multiple of 16 bytes for the fused statics + deltas launch)
    pad = int(os.environ.get("PDS_BENCH_OUT_PAD", "0"))
    out_wide = torch.empty((frames, out_cols + pad), dtype=torch.float32, device=dev) if pad else None
    cmvn = ps.post.CMVN() if post == "cmvn" else None
    cmvn_out = None

    fused_deltas_used = bool(
        deltas is not None and not args.two_launch_deltas and not args.generic
        and not is_si and comp._native_plan(dev).has_fused_deltas)

    def step():
        nonlocal cmvn_out
        if is_si:
            comp.compute_packed(signal, offsets, lengths, out=out, direct=args.generic)
            return
        if deltas is not None and not args.two_launch_deltas and not args.generic:
            # statics and deltas by one launch where the plan has it (pds_stft_deltas_batch: float32 or float64
            # samples, fused pre-emphasis), else the two
            comp.launch_with_deltas(signal, layout, deltas, out=out_wide if out_wide is not None else out, fused=True,
                                    preemphasis=args.preemph)
            return
        if cmvn is not None and args.fused_cmvn and not args.generic and not args.preemph and args.dtype == "f32":
            # features + per-utterance CMVN with the sums taken by the STFT launch (pds_stft_cmvn_batch_f32): less
            # HBM traffic, measured 1 % slower than the two calls, hence opt-in
            cmvn_out = comp.launch_with_cmvn(signal, layout, cmvn, out=cmvn_out, feats_out=out, fused=True)
            return
        comp.launch(signal, layout, out=out, generic=args.generic, preemphasis=args.preemph)
        if deltas is not None:  # statics were written with row stride 3C; deltas go beside them
            deltas.apply_rows(out[:, :C], layout.row_offsets, out=out)
        if cmvn is not None:
            cmvn_out = cmvn.apply_rows(out, layout.row_offsets)

    flag = torch.zeros(1, device=dev) if (use_dist and args.backend == "nccl") else None

    def barrier():
        # NCCL: a one-element all-reduce queued behind the rank's own work is the barrier (it
        # completes on a rank only when every rank has reached it on its stream), then the device
        # synchronisation; torch's dist.barrier() costs ~2 ms here, several steps' worth
        if use_dist:
            if flag is not None:
                dist.all_reduce(flag)
            else:
                dist.barrier()
        torch.cuda.synchronize(dev)

    # pre-roll (not part of W or K): bring the clocks up so that the timed region measures the
    # steady state whatever K is.  At least --preroll-ms of work, then on until eight
    # consecutive blocks of 20 steps have not beaten the fastest block so far by 0.5 % (how long the
    # clocks take depends on what the box did before: after the CPU-baseline leg the ramp takes
    # ~150 ms, not ~40), at most 2 s.
    preroll = 0
    cold_ms = None  # mean step time of the first 20 launches (idle clocks), next to the steady state
    if args.preroll_ms > 0:
        step()  # (first launch: plan tables, kernel attributes, module load)
        torch.cuda.synchronize(dev)
        t_start = time.perf_counter()
        best, settled = float("inf"), 0
        while True:
            t_block = time.perf_counter()
            for _ in range(20):
                step()
            torch.cuda.synchronize(dev)
            now = time.perf_counter()
            preroll += 20
            block = now - t_block
            if cold_ms is None:
                cold_ms = 1e3 * block / 20
            settled = 0 if block < 0.995 * best else settled + 1  # (a ramp gains 1-3 % per block)
            best = min(best, block)
            if (now - t_start >= 1e-3 * args.preroll_ms and settled >= 8) or now - t_start > 2.0:
                break
    for _ in range(args.warmup):
        step()
    barrier()
    # timed region: exactly K steps, events on the launch stream give the kernel time
    # (one event per step boundary: step i runs between marks i and i + 1)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    host_t = [t0]
    for i in range(args.steps):
        step()
        marks[i + 1].record()
        host_t.append(time.perf_counter())
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
    if os.environ.get("PDS_BENCH_DUMP_STEPS"):  # diagnosis: the per-step series
        with open(os.environ["PDS_BENCH_DUMP_STEPS"], "w") as fh:
            json.dump({"kernel_ms": kernel_ms,
                       "host_enqueue_ms": [1e3 * (host_t[i + 1] - host_t[i]) for i in range(args.steps)]}, fh)
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    finite = bool(torch.isfinite(out).all().item())
    power = None
    if rank == 0 and world == 1 and not args.no_power_probe:
        try:
            power = power_probe(step, lambda: torch.cuda.synchronize(dev))
        except Exception as exc:  # an optional leg
            power = {"error": repr(exc)[:200]}
    # The CPU-baseline leg (one spawned worker per host core for ~15 s) comes after the timed region: run in
    # front of it, the GPU steps that followed were paced by a host still busy winding the workers down --
    # 0.308 ms per step against 0.269 ms for the same kernel (rocprofv3 kernel trace, profiles/r2d_*), i.e.
    # the line under-reported the GPU by 13 %.  Workers are spawned, not forked, so HIP being initialised in
    # this process does not matter to them.
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not is_si:
        torch.cuda.synchronize(dev)
        cpu = cpu_baseline(comp, n)
    spot = None
    if rank == 0 and not is_si:
        spot = parity_spot_check(comp, signal, offsets, lengths, layout, out, num_deltas=2 if deltas is not None else 0,
                                 preemph=args.preemph)

    def timed_gather():
        gathered = torch.empty((world * frames, out_cols), dtype=torch.float32, device=dev)
        for _ in range(2):
            step()
            dist.all_gather_into_tensor(gathered, out)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
            dist.all_gather_into_tensor(gathered, out)
        barrier()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
        gather = {"value": world * frames * args.steps / el, "unit": "frames/s",
                  "ms_per_step": 1e3 * el / args.steps,
                  "collective": "all_gather_into_tensor (RCCL)",
                  "bytes_per_rank": frames * out_cols * 4}
        return gather

    gather = None
    if use_dist and not args.no_gather:
        try:
            gather = timed_gather()
        except Exception as exc:  # the headline number must survive a failing optional leg
            gather = {"error": repr(exc)[:200]}

    if rank == 0:
        # every sample read once, every output coefficient written once
        bytes_per_frame = {"f64in": 8, "i16in": 2}.get(args.dtype, 4) * comp.frame_shift + 4 * out_cols
        if cmvn is not None:
            bytes_per_frame += 4 * C + 8 * C  # second read of the features + float64 result
        k_avg_s = 1e-3 * float(np.mean(kernel_ms))
        achieved = frames * bytes_per_frame / k_avg_s / 1e9
        value = world * frames * args.steps / elapsed
        traffic = traffic_bytes = None
        rec = None
        try:
            with open(PMC_TRAFFIC_FILE) as fh:
                rec = json.load(fh).get(args.workload)
            if rec and rec["frames_per_launch"] == frames and not args.generic and not args.ragged and args.dtype == "f32":
                traffic_bytes = rec["hbm_bytes_per_launch"]
                traffic = traffic_bytes / k_avg_s / 1e9
            else:
                rec = None
        except (OSError, ValueError, KeyError):
            rec = None
        # the ceilings the kernel meets before the HBM one.  Algorithmic flops
        # per frame: rFFT-N (~2.5 N log2 N) + window + |X|^2 + 2 flops per filter tap.
        secondary = None
        if not is_si:
            N = comp.dft_size
            taps = int(sum(len(t) for t in comp._truncated_filts))
            flop_per_frame = 2.5 * N * np.log2(N) + comp.frame_length + 3 * (N // 2 + 1) + 2 * taps
            secondary = {
                "fp32_valu": {"achieved_tflops": frames / k_avg_s * flop_per_frame / 1e12, "peak_tflops": VALU_PEAK_TFLOPS,
                              "frac": frames / k_avg_s * flop_per_frame / 1e12 / VALU_PEAK_TFLOPS,
                              "algorithmic_flop_per_frame": float(flop_per_frame)},
                "lds": ({"busy_frac_pmc": rec.get("lds_busy_frac"), "bank_conflict_share_pmc": rec.get("lds_bank_conflict_share"),
                         "valu_instr_per_frame_pmc": rec.get("valu_instr_per_frame"), "source": rec.get("sq_source")}
                        if rec else None),
            }
        line = {
            "metric": "frames/s (whole node) + HBM-roofline %, 40-mel fbank 16kHz 25/10ms"
            if args.workload == DEFAULT_WORKLOAD else f"frames/s ({args.workload})",
            "value": value,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "preroll_steps": preroll,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": {"f64in": "f64in->f32", "i16in": "i16in->f32"}.get(args.dtype, "f32"),  # (samples in HBM -> arithmetic and features)
            "data": "synthetic",
            "config": {
                "workload": args.workload + ("+ragged_1to15s" if args.ragged else "")
                            + ("+float64_samples" if args.dtype == "f64in" else "")
                            + ("+int16_samples" if args.dtype == "i16in" else "")
                            + (f"+preemph{args.preemph:g}" if args.preemph else ""), "utterances_per_gpu": B,
                "samples_per_utterance": int(lengths.mean()),
                "frames_per_gpu_per_step": frames, "num_coeffs": comp.num_coeffs, "post": post,
                "frame_length": comp.frame_length, "frame_shift": comp.frame_shift,
                "dft_size": comp.dft_size, "parallelism": f"utterance-sharded x{world}",
                "kernel": ("si-direct-fir" if (args.generic or not comp.fft_size) else
                           f"si-overlap-save-fft{comp.fft_size}") if is_si else
                          "generic (lds-fft for 2^k sizes, else direct-dft)" if (args.generic or not comp.kernel_kind) else "fused-fft",
                **({"deltas": "same launch (pds_stft_deltas_batch)" if fused_deltas_used else "second launch (pds_deltas_rows_f32)"}
                   if deltas is not None else {}),
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "traffic_bytes_per_launch_pmc": traffic_bytes,
                "kernel_ms_avg": 1e3 * k_avg_s, "kernel_ms_min": float(np.min(kernel_ms)),
                "algorithmic_bytes_per_frame": bytes_per_frame,
                "frames_per_s_per_gpu_kernel_only": frames / k_avg_s,
                "secondary": secondary,
                # the limit this kernel actually runs into on MI355X: the socket's power cap --
                # at the cap a step's joules ARE its time (modelled per instruction kind: profiles/r3s_energy_microbench.txt)
                "power": ({**power, "joules_per_step": power["socket_w"] * 1e-3 * power["ms_per_step_during_probe"],
                           "nj_per_frame": power["socket_w"] * 1e-3 * power["ms_per_step_during_probe"] / frames * 1e9}
                          if power and "socket_w" in power else power),
            },
            "cold_ms_per_step": cold_ms,
            "outputs_finite": finite,
            "parity_spot_check": spot,
        }
        if is_si:
            # compute bound: fused multiply-adds of the FIR bank per launch against the vector peak
            # (157 TFLOP/s counts packed pairs; one FMA per lane and issue is half of that)
            taps = comp.taps
            fma = frames * comp.frame_shift * taps.shape[0] * taps.shape[1] * (1 if comp._real else 2)
            line["compute"] = {"direct_form_equivalent_tflops": 2 * fma / k_avg_s / 1e12,
                               "peak_fp32_valu_tflops": 78.6, "direct_form_fma_per_launch": fma}
        if cpu is not None:
            line["cpu_baseline"] = cpu
        if gather is not None:
            line["with_gather"] = gather
        _emit(saved_stdout, json.dumps(line))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
