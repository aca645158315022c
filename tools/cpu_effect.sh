#!/bin/bash
# GPU-box helper: does the CPU-baseline leg of bench.py change the GPU timing that follows it?
cd ${GRAFT_REPO_ROOT:-/root/repo}
show() { python - "$1" "$2" <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
for key in ("kernel_ms", "host_enqueue_ms"):
    v = d[key]
    n = len(v)
    q = [sum(v[i * n // 10:(i + 1) * n // 10]) / (n // 10) for i in range(10)]
    print("%-9s %-16s" % (sys.argv[1], key), "deciles(ms):", " ".join("%.3f" % x for x in q), " max %.3f" % max(v))
PY
}
for i in 1 2; do
  PDS_BENCH_DUMP_STEPS=gpurun_out/steps_cpu.json python bench.py --steps 300 > /dev/null 2>&1; show with-cpu gpurun_out/steps_cpu.json
  PDS_BENCH_DUMP_STEPS=gpurun_out/steps_nocpu.json python bench.py --steps 300 --no-cpu-baseline > /dev/null 2>&1; show no-cpu gpurun_out/steps_nocpu.json
done
rocm-smi --showpower --showclocks 2>/dev/null | head -20
