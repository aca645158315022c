set -e
timeout -k 10 600 python -m pytest tests/test_gpu_stft.py -x -q -m gpu -k "gammatone or alternative" > gpurun_out/t_ms.log 2>&1 || { tail -40 gpurun_out/t_ms.log; exit 1; }
tail -1 gpurun_out/t_ms.log
for wl in gammatone64_48k_cmvn_b256x10s; do
timeout -k 10 300 python bench.py --no-cpu-baseline --workload $wl > gpurun_out/b_tmp.json 2> gpurun_out/b_tmp.err || true
python - <<PY
import json
d=json.loads(open("gpurun_out/b_tmp.json").read().strip().splitlines()[-1])
print("$wl", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["parity_spot_check"]["pass"])
PY
done
