ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for lib in a_noxcd b_xcd; do
  export PDS_AMD_LIB=$ROOT/variants/lib_$lib.so
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $ROOT/gpurun_out/fetch_$lib -o pmc -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
  python3 - <<PY
import csv
v=[float(r["Counter_Value"]) for r in csv.DictReader(open("$ROOT/gpurun_out/fetch_$lib/pmc_counter_collection.csv")) if "stft_wave" in r["Kernel_Name"] and r["Counter_Name"]=="FETCH_SIZE"]
print("$lib FETCH_SIZE KiB mean", sum(v)/len(v))
PY
done
