#!/bin/bash
# GPU-box helper: one workload under each forced filter walk (PDS_STFT_WALK).  bash tools/walk_sweep.sh <workload> [walks...]
cd ${GRAFT_REPO_ROOT:-/root/repo}
wl=${1:-gabor64_b1024x10s}; shift
for w in ${@:-rseg seg mseg ell}; do
  for r in 1 2; do
    PDS_STFT_WALK=$w timeout -k 5 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-power-probe --workload $wl 2>gpurun_out/walk_err.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$wl', '$w', 'frames/s %.4g kernel_ms %.4f spot %s' % (d['value'], d['roofline']['kernel_ms_avg'], d['parity_spot_check']['pass']))" || tail -2 gpurun_out/walk_err.txt
  done
done
