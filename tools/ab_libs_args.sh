#!/bin/bash
# GPU-box helper: A/B of library builds under variants/lib_*.so with extra bench.py arguments:
#   bash tools/ab_libs_args.sh "<bench args>" [<more bench args> ...]     e.g. "--preemph 0.97" "--dtype i16in --preemph 0.97"
cd ${GRAFT_REPO_ROOT:-/root/repo}
for args in "$@"; do
  for rep in $(seq 1 ${AB_REPS:-3}); do
    for lib in variants/lib_*.so; do
      PDS_AMD_LIB=$PWD/$lib timeout -k 5 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-power-probe $args 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$args | $lib', 'frames/s %.4g kernel_ms %.4f spot %s' % (d['value'], d['roofline']['kernel_ms_avg'], (d.get('parity_spot_check') or {}).get('pass')))"
    done
  done
done
