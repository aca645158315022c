set -e
timeout -k 10 600 python -m pytest tests/test_gpu_stft.py -x -q -m gpu -k "4096 or geometr or long or beyond" > gpurun_out/t_4096.log 2>&1 || { tail -40 gpurun_out/t_4096.log; exit 1; }
tail -1 gpurun_out/t_4096.log
timeout -k 10 300 python bench.py --no-cpu-baseline --workload fbank80_48k_50_12.5_b256x10s > gpurun_out/b_tmp.json 2> gpurun_out/b_tmp.err || true
python - <<PY
import json
d=json.loads(open("gpurun_out/b_tmp.json").read().strip().splitlines()[-1])
print("N4096", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["parity_spot_check"])
PY
