set -e
PDS_DEBUG_PLAN=1 timeout -k 10 600 python -m pytest tests/test_gpu_stft.py -x -q -m gpu > gpurun_out/t_stft.log 2>&1 || { tail -40 gpurun_out/t_stft.log; exit 1; }
tail -1 gpurun_out/t_stft.log
for w in seg; do
for wl in fbank80_48k_25_10_b256x10s fbank80_48k_50_12.5_b256x10s gabor64_b1024x10s; do
PDS_DEBUG_PLAN=1 PDS_STFT_WALK=$w timeout -k 10 300 python bench.py --no-cpu-baseline --workload $wl > gpurun_out/b_tmp.json 2> gpurun_out/b_tmp.err || true
grep "pds plan" gpurun_out/b_tmp.err | head -1
python - <<PY
import json
d=json.loads(open("gpurun_out/b_tmp.json").read().strip().splitlines()[-1])
print("$w $wl", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["parity_spot_check"]["pass"], d["parity_spot_check"]["max_err_over_tolerance"])
PY
done
done
