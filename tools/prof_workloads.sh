ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for wl in gammatone64_48k_cmvn_b256x10s fbank80_energy_deltas2_b1024x10s; do
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_$wl -o p -- python3 $ROOT/bench.py --workload $wl --steps 30 --warmup 5 --no-cpu-baseline > $ROOT/gpurun_out/prof_$wl.json 2>/dev/null
python3 - <<PY
import csv,glob
f=glob.glob("$ROOT/gpurun_out/prof_$wl/**/p_kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:6]:
    print("$wl", r["Name"][:70], r["Calls"], r["AverageNs"], r["Percentage"])
PY
done
