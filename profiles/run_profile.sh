#!/bin/bash
# Profiling recipe used for the summaries in this directory (run on the GPU box via gpurun):
#   [BENCH_ARGS="--workload <name>"] bash profiles/run_profile.sh <tag>
# Pass 1: kernel trace + stats.  Passes 2..: PMC counters, one group per pass, never
# combined with other trace domains.
set -e
TAG=${1:-r1}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --no-cpu-baseline ${BENCH_ARGS}"   # defaults: clock pre-roll, 10 warm-up, 100 timed steps; BENCH_ARGS="--workload ..." for another workload
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- $BENCH > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- $BENCH > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- $BENCH > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq1 -o pmc -- $BENCH > $OUT/pmc_sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc_sq2 -o pmc -- $BENCH > $OUT/pmc_sq2.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_misc -o pmc -- $BENCH > $OUT/pmc_misc.log 2>&1 || true
find $OUT -name "*.csv" | head -40
