#!/bin/bash
# GPU-box helper: the sample-format / pre-emphasis flows of one workload.  bash tools/bench_flows.sh <workload> [...]
cd ${GRAFT_REPO_ROOT:-/root/repo}
for wl in "$@"; do
  for extra in "" "--preemph 0.97" "--dtype f64in" "--dtype f64in --preemph 0.97" "--dtype i16in" "--dtype i16in --preemph 0.97"; do
    timeout -k 5 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-power-probe --workload $wl $extra 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$wl $extra', 'frames/s %.4g step_ms %.4f frac %.3f spot %s' % (d['value'], d['roofline']['kernel_ms_avg'], d['roofline']['frac'], (d.get('parity_spot_check') or {}).get('pass')))" || echo "$wl $extra FAILED"
  done
done
