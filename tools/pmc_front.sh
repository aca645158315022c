#!/bin/bash
# GPU-box helper: SQ counters of the fused STFT kernel with both front ends (PDS_STFT_FRONT=valu | mfma),
# separate passes, one counter group per pass:  bash tools/pmc_front.sh <tag> [workload]
# Results: gpurun_out/pmc_front_<tag>/<front>_<group>/..., summary printed by tools/pmc_front_summary.py
TAG=${1:-x}
WL=${2:-fbank40_16k_25_10_b1024x10s}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_front_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
G1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT"
G2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD"
G3="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM GRBM_GUI_ACTIVE"
for front in ${FRONTS:-valu mfma}; do
  n=1
  for grp in "$G1" "$G2" "$G3"; do
    PDS_STFT_FRONT=$front timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/${front}_g$n -o pmc -- \
      python3 $ROOT/bench.py --no-cpu-baseline --steps 10 --warmup 2 --preroll-ms 0 --workload $WL > $OUT/${front}_g$n.log 2>&1
    echo "$front group $n exit $?"
    n=$((n+1))
  done
done
python3 $ROOT/tools/pmc_front_summary.py $OUT
