#!/usr/bin/env python3
"""Summarise rocprofv3 csv output (kernel stats + counter_collection) per kernel name."""
import csv, glob, os, sys, collections
root = sys.argv[1]
for f in sorted(glob.glob(os.path.join(root, "**", "*kernel_stats.csv"), recursive=True)):
    print("==", os.path.relpath(f, root))
    for row in csv.DictReader(open(f)):
        print("  %-70s calls %6s  avg %12s ns  total %14s ns  %s%%" % (row.get("Name", "")[:70], row.get("Calls"), row.get("AverageNs"), row.get("TotalDurationNs"), row.get("Percentage")))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"][:60]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[k][row["Counter_Name"]] += 1
for k in agg:
    print("==", k)
    for c in sorted(agg[k]):
        print("  %-28s total %16.0f  per-dispatch %16.1f  (%d dispatches)" % (c, agg[k][c], agg[k][c] / cnt[k][c], cnt[k][c]))
