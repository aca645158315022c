#!/bin/bash
# GPU-box helper: SQ / cache counters of two bench.py invocations side by side, one counter group per pass:
#   bash tools/pmc_ab.sh <tag> "<label A>|<bench args A>" "<label B>|<bench args B>" ...
# Results: gpurun_out/pmc_ab_<tag>/<label>_g<n>/..., summary by tools/pmc_front_summary.py (means per launch of
# the kernels whose name contains "stft_")
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_ab_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
G1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT"
G2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD"
G3="SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"
G4="TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"
G5="FETCH_SIZE"
G6="WRITE_SIZE"
for spec in "$@"; do
  label=${spec%%|*}; args=${spec#*|}
  n=1
  for grp in "$G1" "$G2" "$G3" "$G4" "$G5" "$G6"; do
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/${label}_g$n -o pmc -- \
      python3 $ROOT/bench.py --no-cpu-baseline --steps 10 --warmup 2 --preroll-ms 0 $args > $OUT/${label}_g$n.log 2>&1
    echo "$label group $n exit $?"
    n=$((n+1))
  done
done
python3 $ROOT/tools/pmc_front_summary.py $OUT
