#!/bin/bash
# GPU-box helper: A/B of the prefetch variants under variants/lib_pf*.so against the same libraries with the
# (libraries: tools/build_variant.sh pfX -DPDS_DEV_ONLY512 -DPDS_EXPERIMENTS=1 [-DPDS_PF_WIN=1 -DPDS_PF_PLACE=1 -DPDS_PF_TW=0 -DPDS_DEV_MINW=3 ...])
# prefetch instantiation switched off (PDS_STFT_PF=0), same box, interleaved repetitions; then the phase stamps
# of the variants/lib_st*.so builds.   bash tools/ab_pf.sh [workload] > gpurun_out/ab_pf.txt
cd ${GRAFT_REPO_ROOT:-/root/repo}
wl=${1:-fbank40_16k_25_10_b1024x10s}
one() {  # label, lib, PF env
  PDS_STFT_PF=$3 PDS_AMD_LIB=$PWD/$2 timeout -k 5 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-gather --workload $wl 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d.get('parity_spot_check',{}); print('$1', 'frames/s %.4g kernel_ms %.4f min %.4f frac %.3f parity %s %.3g' % (d['value'], d['roofline']['kernel_ms_avg'], d['roofline']['kernel_ms_min'], d['roofline']['frac'], c.get('pass'), c.get('max_err_over_tolerance', -1)))"
}
for rep in 1 2 3; do
  for lib in variants/lib_pf*.so; do
    one "$lib PF=1" $lib 1
    [ -n "$AB_BASE_ALL" -o $lib = "$(ls variants/lib_pf*.so | head -1)" -o -n "$(echo $lib | grep 3w0)" ] && one "$lib PF=0" $lib 0
  done
done
for lib in $(ls variants/lib_st*.so 2>/dev/null); do
  echo "== stamps $lib PF=1"; PDS_STFT_PF=1 PDS_AMD_LIB=$PWD/$lib timeout -k 5 200 python tools/phase_stamps.py $wl 2>&1 | head -12
done
