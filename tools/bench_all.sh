#!/bin/bash
# GPU-box helper: one bench line per workload (no CPU baseline), summarised
cd ${GRAFT_REPO_ROOT:-/root/repo}
for wl in fbank40_16k_25_10_b1024x10s fbank80_energy_deltas2_b1024x10s gabor64_b1024x10s gammatone64_48k_cmvn_b256x10s fbank40_nopad320_b1024x10s fbank40_nopad400_b1024x10s fbank40_nopad480_b1024x10s fbank40_16k_32_10_b1024x10s fbank40_8k_25_10_b1024x10s fbank80_48k_20_10_b256x10s fbank80_48k_25_10_b256x10s fbank80_48k_50_12.5_b256x10s si_gabor40_b64x10s si_gammatone40_48k_b32x10s; do
  timeout -k 5 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --workload $wl "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); pw=d['roofline'].get('power') or {}; print('$wl', 'frames/s %.4g step_ms %.4f min %.4f frac %.3f' % (d['value'], d['roofline']['kernel_ms_avg'], d['roofline']['kernel_ms_min'], d['roofline']['frac']), 'power', pw.get('socket_w'), 'W of', pw.get('cap_w'), 'sclk', pw.get('sclk_mhz'))"
done
