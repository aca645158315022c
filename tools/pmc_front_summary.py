#!/usr/bin/env python3
"""Means per launch of every counter tools/pmc_front.sh collected, side by side per front end"""
import collections
import csv
import glob
import os
import sys

out = sys.argv[1]
table = collections.defaultdict(dict)
for path in sorted(glob.glob(os.path.join(out, "*_g*", "**", "*counter_collection.csv"), recursive=True)):
    front = os.path.relpath(path, out).split(os.sep)[0].split("_g")[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if "stft_" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        table[k][front] = sum(v) / len(v)
fronts = sorted({f for v in table.values() for f in v})
print("%-32s" % "counter (mean per launch)" + "".join("%16s" % f for f in fronts))
for k in sorted(table):
    print("%-32s" % k + "".join("%16.6g" % table[k].get(f, float("nan")) for f in fronts))
