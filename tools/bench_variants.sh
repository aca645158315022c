#!/bin/bash
# run on the GPU box: time every library in variants/ with the standard bench, interleaved rounds
cd ${GRAFT_REPO_ROOT:-/root/repo}
for round in 1 2; do
for lib in variants/lib_*.so; do
  PDS_AMD_LIB=$PWD/$lib timeout -k 5 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib round $round', 'frames/s %.4g kernel_ms %.4f min %.4f' % (d['value'], d['roofline']['kernel_ms_avg'], d['roofline']['kernel_ms_min']))"
done
done
