set -e
timeout -k 10 600 python -m pytest tests/test_gpu_post.py tests/test_gpu_stft.py -x -q -m gpu -k "fused_statics or falls_back or statics_plus_deltas" > gpurun_out/t_fused.log 2>&1 || { tail -40 gpurun_out/t_fused.log; exit 1; }
tail -1 gpurun_out/t_fused.log
for dbg in 0 16 0 16; do
PDS_DL_DEBUG=$dbg timeout -k 10 300 python bench.py --workload fbank80_energy_deltas2_b1024x10s --no-cpu-baseline > gpurun_out/b_tmp.json 2> gpurun_out/b_tmp.err || true
python - <<PY
import json
d=json.loads(open("gpurun_out/b_tmp.json").read().strip().splitlines()[-1])
print("dbg $dbg", d["value"], d["ms_per_step"], d["parity_spot_check"]["pass"])
PY
done
