"""bench.py's output contract: one JSON line with the driver's fields, `roofline` and `cpu_baseline`."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

DRIVER_FIELDS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                 "scaling", "vs_baseline", "dtype", "data", "config", "roofline"}
ROOFLINE_FIELDS = {"bound", "achieved", "peak", "unit", "frac", "traffic"}
CPU_FIELDS = {"value", "unit", "cores", "kind", "sample"}


def test_cpu_baseline_leg_reports_the_oracle_on_host_cores():
    # the CPU leg runs before the GPU is touched and needs no device: the oracle in worker processes
    import bench
    import pydrobert_speech_amd as ps
    from pydrobert_speech_amd.alias import alias_factory_subclass_from_arg

    cfg, n, _, _ = bench.WORKLOADS[bench.DEFAULT_WORKLOAD]
    comp = alias_factory_subclass_from_arg(ps.compute.FrameComputer, cfg)
    os.environ["OMP_NUM_THREADS"] = "1"
    res = bench.cpu_baseline(comp, 16000, budget_s=0.002)  # two one-second utterances per worker
    assert CPU_FIELDS <= set(res)
    assert res["kind"] == "port" and res["unit"] == "frames/s" and res["value"] > 0 and 1 <= res["cores"] <= 32
    json.dumps(res)


def test_workloads_name_the_baseline_configurations():
    import bench

    with open(os.path.join(ROOT, "BASELINE.json")) as fh:
        baseline = json.load(fh)
    assert bench.DEFAULT_WORKLOAD in bench.WORKLOADS
    cfg, n, batch, post = bench.WORKLOADS[bench.DEFAULT_WORKLOAD]
    # configs[1]: 1024 utterances of 10 s at 16 kHz, 40 mel filters, 25 / 10 ms frames
    assert (n, batch, post) == (160000, 1024, None)
    assert cfg["bank"]["num_filts"] == 40 and cfg["frame_length_ms"] == 25 and cfg["frame_shift_ms"] == 10
    assert "40-mel" in baseline["metric"] and "frames/s" in baseline["metric"]


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_the_contract_fields():
    res = subprocess.run(
        [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--batch", "16",
         "--no-cpu-baseline", "--preroll-ms", "5"],
        capture_output=True, text=True, timeout=600,
    )
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, res.stdout[-2000:]
    line = json.loads(lines[0])
    assert DRIVER_FIELDS <= set(line), sorted(DRIVER_FIELDS - set(line))
    assert ROOFLINE_FIELDS <= set(line["roofline"])
    assert line["n_gpus"] == 1 and line["steps"] == 3 and line["warmup"] == 1 and line["unit"] == "frames/s"
    assert line["scaling"] == "weak" and line["higher_is_better"] is True and line["vs_baseline"] is None
    assert line["dtype"] == "f32" and line["data"] == "synthetic" and "workload" in line["config"]
    assert line["roofline"]["bound"] == "hbm" and line["roofline"]["peak"] == 8000.0
    assert 0 < line["roofline"]["frac"] < 1 and line["value"] > 0 and line["outputs_finite"] is True
    # self-proving line: rows of the timed buffer against the oracle, the secondary ceilings, the cold step time
    spot = line["parity_spot_check"]
    assert spot["pass"] is True and spot["rows"] > 0 and spot["max_err_over_tolerance"] <= 1.0
    sec = line["roofline"]["secondary"]
    assert sec["fp32_valu"]["peak_tflops"] == 157.3 and 0 < sec["fp32_valu"]["frac"] < 1
    assert 10e3 < sec["fp32_valu"]["algorithmic_flop_per_frame"] < 20e3  # ~14 kflop per frame
    assert "lds" in sec
    assert line["cold_ms_per_step"] > 0


@pytest.mark.gpu
def test_bench_two_ranks_rehearsal_on_one_gpu():
    # the N > 1 path (launcher environment, barrier, max-over-ranks timing, rank 0 prints) with two
    # ranks sharing GPU 0 over gloo: RCCL needs one GPU per rank, which the driver's scaling run has
    import socket

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
         "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo",
         "--steps", "3", "--warmup", "1", "--batch", "16", "--preroll-ms", "5"],
        capture_output=True, text=True, timeout=900, env=env,
    )
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0
    # whole-job value: both ranks' frames over the slowest rank's time
    assert line["config"]["frames_per_gpu_per_step"] * 2 * line["steps"] / (line["ms_per_step"] * 1e-3 * line["steps"]) == pytest.approx(line["value"], rel=1e-6)
