#!/bin/bash
# GPU-box helper: kernel time of the headline configuration against the batch size
cd ${GRAFT_REPO_ROOT:-/root/repo}
for b in 1 4 16 64 128 256 512 1024 4096; do
  python bench.py --batch $b --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('B = %5d' % $b, 'kernel_ms %.4f min %.4f  frames/s %.4g  (%.0f%% of the 1024-utterance rate per frame)' % (r['kernel_ms_avg'], r['kernel_ms_min'], d['value'], 100 * (0.2952 / 1024 * $b) / r['kernel_ms_avg']))"
done
