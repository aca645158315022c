#!/bin/bash
# GPU-box helper: A/B of the fused kernel's front ends on one box, interleaved:
#   bash tools/ab_front.sh <workload> [<workload> ...]
# PDS_STFT_FRONT=mfma selects the matrix-pipe front end (libraries built with make EXTRA=-DPDS_EXPERIMENTS=1 only);
# the default, and the only form of the product build, is the in-lane N1-point transform.
cd ${GRAFT_REPO_ROOT:-/root/repo}
for wl in "$@"; do
  for rep in 1 2 3; do
    for front in valu mfma; do
      PDS_STFT_FRONT=$front timeout -k 5 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --workload $wl 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$wl front=$front', 'frames/s %.4g kernel_ms %.4f' % (d['value'], d['roofline']['kernel_ms_avg']))"
    done
  done
done
