#!/usr/bin/env python3
"""Condense a profiles/run_profile.sh output directory into one text summary.

    python profiles/summarize.py gpurun_out/prof_<tag> [frames_per_launch [timed_steps [kernel-name substring]]] > profiles/<tag>_summary.txt
"""
import collections
import csv
import os
import sys

d = sys.argv[1]
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 1024000
tot = {}
lines = []
stats = list(csv.DictReader(open(os.path.join(d, "stats", "stats_kernel_stats.csv"))))
lines.append("rocprofv3 --kernel-trace --stats (top kernels)")
for r in stats[:4]:
    lines.append("  %-70s calls %s avg_ns %s min_ns %s max_ns %s pct %s" % (
        r["Name"][:70], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"], r["Percentage"]))
# bench.py times its LAST `steps` launches (after the clock pre-roll and the warm-up): the same
# window from the kernel trace is the duration its roofline line must agree with
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 100
KERN = sys.argv[4] if len(sys.argv) > 4 else "stft_"  # the kernel the summary is about
kern = next(r for r in stats if KERN in r["Name"])
avg_s = float(kern["AverageNs"]) * 1e-9
trace_path = os.path.join(d, "stats", "stats_kernel_trace.csv")
if os.path.exists(trace_path):
    rows = [r for r in csv.DictReader(open(trace_path)) if KERN in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    last = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[-steps:]]
    first = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[:40]]
    lines.append("  kernel trace: %d launches; first 40 mean %.1f us (clocks ramping), last %d (the timed region) "
                 "mean %.1f us min %.1f us" % (len(rows), sum(first) / len(first) / 1e3, len(last),
                                              sum(last) / len(last) / 1e3, min(last) / 1e3))
    avg_s = sum(last) / len(last) * 1e-9
for sub in ("pmc_fetch", "pmc_write", "pmc_sq1", "pmc_sq2", "pmc_misc"):
    path = os.path.join(d, sub, "pmc_counter_collection.csv")
    if not os.path.exists(path):
        continue
    agg = collections.defaultdict(list)
    meta = None
    for r in csv.DictReader(open(path)):
        if KERN in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = r
    for k, v in agg.items():
        tot[k] = sum(v) / len(v)
if meta:
    lines.append("dispatch: grid %s wg %s VGPR %s SGPR %s LDS_static %s" % (
        meta["Grid_Size"], meta["Workgroup_Size"], meta["VGPR_Count"], meta["SGPR_Count"], meta["LDS_Block_Size"]))
lines.append("PMC means per launch of the %s kernel (separate passes):" % KERN)
for k in sorted(tot):
    lines.append("  %-24s %.6g" % (k, tot[k]))
it = frames / 4.0
g = lambda k: tot.get(k, float("nan"))
lines.append("derived (frames per launch %d, kernel avg %.4f ms):" % (frames, avg_s * 1e3))
lines.append("  per wave-iteration (4 frames): VALU %.0f  SALU %.0f  LDS %.0f  VMEM_RD %.1f" % (
    g("SQ_INSTS_VALU") / it, g("SQ_INSTS_SALU") / it, g("SQ_INSTS_LDS") / it, g("SQ_INSTS_VMEM_RD") / it))
wc = g("SQ_WAVE_CYCLES")
lines.append("  wave-time split: VALU %.1f%%  LDS-issue %.1f%%  wait_any %.1f%%  wait_inst_any %.1f%% (of which LDS %.1f%%)" % (
    100 * g("SQ_ACTIVE_INST_VALU") / wc, 100 * g("SQ_ACTIVE_INST_LDS") / wc, 100 * g("SQ_WAIT_ANY") / wc,
    100 * g("SQ_WAIT_INST_ANY") / wc, 100 * g("SQ_WAIT_INST_LDS") / wc))
clk = g("GRBM_GUI_ACTIVE") / 8 / avg_s
lines.append("  effective clock %.2f GHz; LDS busy %.1f%% of kernel cycles per CU, bank-conflict share %.1f%%" % (
    clk / 1e9, 100 * g("SQ_LDS_IDX_ACTIVE") / 256 / (clk * avg_s), 100 * g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")))
# FETCH_SIZE / WRITE_SIZE are in KiB; gfx950 FETCH_SIZE reads half of a wide coalesced stream
# (MI355X_MICROARCH.md, HBM section): report raw and corrected
fetch_raw, write_raw = g("FETCH_SIZE") * 1024, g("WRITE_SIZE") * 1024
lines.append("  HBM traffic per launch: FETCH_SIZE raw %.1f MB (x2 gfx950 correction: %.1f MB), WRITE_SIZE %.1f MB" % (
    fetch_raw / 1e6, 2 * fetch_raw / 1e6, write_raw / 1e6))
lines.append("  => corrected traffic %.1f MB, %.2f TB/s; algorithmic %d B/frame -> %.1f MB" % (
    (2 * fetch_raw + write_raw) / 1e6, (2 * fetch_raw + write_raw) / avg_s / 1e12, 800, frames * 800 / 1e6))
print("\n".join(lines))
