#!/bin/bash
# GPU-box helper: static round-robin against the workgroup ticket counter (PDS_STFT_DYN=0|1), same library, same box
# (libraries: tools/build_variant.sh dynX -DPDS_DEV_ONLY512 -DPDS_DYN=1 [-DPDS_ONE_WG]; spread*: ... -DPDS_STAMPS=2)
cd ${GRAFT_REPO_ROOT:-/root/repo}
wl=${1:-fbank40_16k_25_10_b1024x10s}
one() {  # label, lib, DYN env
  PDS_STFT_PF=0 PDS_STFT_DYN=$3 PDS_AMD_LIB=$PWD/$2 timeout -k 5 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-gather --workload $wl 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d.get('parity_spot_check',{}); print('$1', 'frames/s %.4g kernel_ms %.4f min %.4f frac %.3f parity %s %.3g' % (d['value'], d['roofline']['kernel_ms_avg'], d['roofline']['kernel_ms_min'], d['roofline']['frac'], c.get('pass'), c.get('max_err_over_tolerance', -1)))"
}
for rep in 1 2 3; do
  for lib in variants/lib_dyn*.so; do
    one "$lib DYN=0" $lib 0
    one "$lib DYN=1" $lib 1
  done
done
for lib in variants/lib_spread*.so; do
  for dyn in 0 1; do
    echo "== $lib DYN=$dyn"; PDS_STFT_PF=0 PDS_STFT_DYN=$dyn PDS_AMD_LIB=$PWD/$lib timeout -k 5 200 python tools/wave_spread.py $wl 2>&1 | grep -v "XCD@" | tail -4
  done
done
