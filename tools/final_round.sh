#!/bin/bash
# GPU-box helper: the round's closing measurements, all from one box -- default bench line (with CPU baseline and
# power probe), rocprofv3 stats + PMC passes of the headline, kernel stats of configs[2] and configs[4], one line per
# workload and flow.   bash tools/final_round.sh <tag>
TAG=${1:-r3z}
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
timeout -k 10 300 python bench.py > $O/${TAG}_bench_line.json 2> $O/${TAG}_bench.err; echo "bench done rc=$?"
bash profiles/run_profile.sh $TAG > $O/${TAG}_profile.log 2>&1; echo "headline profile done rc=$?"
python profiles/summarize.py $O/prof_$TAG > $O/${TAG}_wave_kernel_summary.txt 2>&1; cp $O/prof_$TAG/stats/stats_kernel_stats.csv $O/${TAG}_kernel_stats.csv
for w in fbank80_energy_deltas2_b1024x10s:c3 gammatone64_48k_cmvn_b256x10s:c5; do
  wl=${w%%:*}; short=${w##*:}
  (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/prof_${TAG}_$short -o stats -- python3 $OLDPWD/bench.py --no-cpu-baseline --no-power-probe --workload $wl > $OLDPWD/$O/${TAG}_${short}_bench.json 2>/dev/null)
  cp $O/prof_${TAG}_$short/stats_kernel_stats.csv $O/${TAG}_${short}_kernel_stats.csv 2>/dev/null || find $O/prof_${TAG}_$short -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_${short}_kernel_stats.csv \;
  echo "$short stats done"
done
{
  bash tools/bench_all.sh
  for extra in "--preemph 0.97" "--ragged" "--dtype i16in" "--dtype i16in --preemph 0.97" "--dtype f64in" "--dtype f64in --preemph 0.97"; do
    timeout -k 5 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-power-probe $extra 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('headline $extra', 'frames/s %.4g step_ms %.4f min %.4f frac %.3f' % (d['value'], d['roofline']['kernel_ms_avg'], d['roofline']['kernel_ms_min'], d['roofline']['frac']))"
  done
  for extra in "--preemph 0.97" "--dtype f64in" "--two-launch-deltas"; do
    timeout -k 5 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-power-probe --workload fbank80_energy_deltas2_b1024x10s $extra 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('fbank80_energy_deltas2 $extra', 'frames/s %.4g step_ms %.4f min %.4f frac %.3f' % (d['value'], d['roofline']['kernel_ms_avg'], d['roofline']['kernel_ms_min'], d['roofline']['frac']))"
  done
} > $O/${TAG}_bench_all_workloads.txt 2>&1
cat $O/${TAG}_bench_all_workloads.txt
python - <<PY
import json
d = json.load(open("$O/${TAG}_bench_line.json"))
r = d["roofline"]
print("LINE frames/s %.4g ms/step %.4f kernel_ms %.4f frac %.4f power %s spot %s cpu %s" % (d["value"], d["ms_per_step"], r["kernel_ms_avg"], r["frac"], r.get("power"), d["parity_spot_check"]["pass"], d.get("cpu_baseline", {}).get("value")))
PY
