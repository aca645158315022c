#!/bin/bash
# GPU-box helper: HBM traffic per kernel of one bench workload (FETCH_SIZE / WRITE_SIZE in
# separate passes, as /opt/skills/guides/MI355X_MICROARCH.md prescribes; KiB units, FETCH_SIZE x2
# on gfx950):   bash tools/pmc_workload.sh <workload> [bench args]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
WL=$1; shift
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $ROOT/gpurun_out/pmc_${WL}_$ctr -o pmc -- python3 $ROOT/bench.py --workload $WL --steps 10 --warmup 2 --no-cpu-baseline "$@" > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(list))
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("$ROOT/gpurun_out/pmc_${WL}_%s/**/pmc_counter_collection.csv" % ctr, recursive=True)[0]
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == ctr and "pds::" in r["Kernel_Name"]:
            tot[r["Kernel_Name"].split("(")[0][:60]][ctr].append(float(r["Counter_Value"]))
for k, v in tot.items():
    fetch = 2 * 1024 * sum(v["FETCH_SIZE"]) / max(1, len(v["FETCH_SIZE"]))
    write = 1024 * sum(v["WRITE_SIZE"]) / max(1, len(v["WRITE_SIZE"]))
    print("$WL %-60s launches %d  fetch %.1f MB  write %.1f MB" % (k, len(v["FETCH_SIZE"]), fetch / 1e6, write / 1e6))
PY
