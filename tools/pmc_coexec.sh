cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_front_b
mkdir -p $OUT
for front in valu mfma; do
  PDS_STFT_FRONT=$front PDS_STFT_WALK=ell timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $OUT/${front}_g1 -o pmc -- python3 $ROOT/bench.py --no-cpu-baseline --steps 10 --warmup 2 --preroll-ms 0 > $OUT/${front}_g1.log 2>&1
done
python3 $ROOT/tools/pmc_front_summary.py $OUT
