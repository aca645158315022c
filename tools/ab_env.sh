#!/bin/bash
# GPU-box helper: A/B of one environment switch of the library on one box, interleaved, three rounds:
#   bash tools/ab_env.sh <VAR> <value> [<value> ...] -- <workload> [<workload> ...]
# e.g. PDS_STFT_WALK ell rseg, PDS_STFT_FRONT valu mfma (both read at plan creation).
cd ${GRAFT_REPO_ROOT:-/root/repo}
var=$1; shift
vals=()
while [ "$1" != "--" ]; do vals+=("$1"); shift; done
shift
for wl in "$@"; do
  for rep in 1 2 3; do
    for v in "${vals[@]}"; do
      env $var=$v timeout -k 5 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --workload $wl 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$wl $var=$v', 'frames/s %.4g kernel_ms %.4f' % (d['value'], d['roofline']['kernel_ms_avg']))"
    done
  done
done
