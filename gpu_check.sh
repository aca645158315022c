#!/bin/bash
# GPU-box helper: parity tests, then a short bench; writes under gpurun_out/
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout -k 10 600 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -4 gpurun_out/pytest_gpu.log | cut -c1-300
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" > gpurun_out/bench_fast.json 2> gpurun_out/bench_fast.err
tail -3 gpurun_out/bench_fast.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_fast.json"))
print("frames/s %.4g  ms/step %.4f  kernel_ms %.4f  hbm_frac %.4f" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms_avg"], d["roofline"]["frac"]))
PY
