cd ${GRAFT_REPO_ROOT:-/root/repo}
python bench.py > gpurun_out/bench_r3f_line.json 2> gpurun_out/bench_r3f.err; tail -c 600 gpurun_out/bench_r3f_line.json
bash profiles/run_profile.sh r3f > gpurun_out/prof_r3f.log 2>&1
BENCH_ARGS="--workload fbank80_energy_deltas2_b1024x10s" bash profiles/run_profile.sh r3f_c3 > gpurun_out/prof_r3f_c3.log 2>&1
BENCH_ARGS="--workload gammatone64_48k_cmvn_b256x10s" bash profiles/run_profile.sh r3f_c5 > gpurun_out/prof_r3f_c5.log 2>&1
bash tools/bench_all.sh > gpurun_out/bench_all_r3f.txt 2>&1
for args in "--dtype f64in" "--dtype f64in --preemph 0.97" "--preemph 0.97" "--ragged"; do
  timeout -k 5 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-gather $args 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('fbank40_16k_25_10_b1024x10s $args', 'frames/s %.4g step_ms %.4f frac %.3f' % (d['value'], d['roofline']['kernel_ms_avg'], d['roofline']['frac']))"
done >> gpurun_out/bench_all_r3f.txt 2>&1
cat gpurun_out/bench_all_r3f.txt
