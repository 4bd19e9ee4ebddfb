#!/usr/bin/env python3
"""Average per-dispatch counter values per kernel from rocprofv3 --pmc csv output directories."""
import csv, glob, os, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "?")
            k = k.split("(")[0].replace("void adlhip::", "").replace("adlhip::", "")
            c = row.get("Counter_Name"); v = float(row.get("Counter_Value", 0) or 0)
            a = acc[k][c]; a[0] += v; a[1] += 1
for k in sorted(acc):
    if not any(s in k for s in ("onesweep", "scatter", "count", "hist")): continue
    print(k)
    for c in sorted(acc[k]):
        s, n = acc[k][c]
        print("   %-28s avg/dispatch %16.1f  (n=%d)" % (c, s / n, n))
