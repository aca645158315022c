#!/bin/bash
# GPU-box helper: A/B of library builds under variants/lib_*.so on one box (same clocks, same HBM):
#   bash tools/ab_libs.sh <workload> [<workload> ...]
cd ${GRAFT_REPO_ROOT:-/root/repo}
for wl in "$@"; do
  for rep in 1 2 3; do
    for lib in variants/lib_*.so; do
      PDS_AMD_LIB=$PWD/$lib timeout -k 5 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --workload $wl 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$wl $lib', 'frames/s %.4g kernel_ms %.4f' % (d['value'], d['roofline']['kernel_ms_avg']))"
    done
  done
done
