#!/usr/bin/env python3
"""Average per-dispatch counter values per kernel from rocprofv3 --pmc csv output directories.
Launches of one kernel with different grids are listed apart (the two MSD passes of the large sort are the same kernel:
pass 1 = n / tile workgroups, pass 2 = 256 x tiles-per-bucket)."""
import csv, glob, os, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "?")
            k = k.split("(")[0].replace("void adlhip::", "").replace("adlhip::", "")
            g = row.get("Grid_Size") or row.get("Grid_Size_X") or "?"
            k = "%s  grid=%s" % (k, g)
            c = row.get("Counter_Name"); v = float(row.get("Counter_Value", 0) or 0)
            a = acc[k][c]; a[0] += v; a[1] += 1
for k in sorted(acc):
    if not any(s in k for s in ("onesweep", "scatter", "count", "hist", "segment", "finish")): continue
    print(k)
    for c in sorted(acc[k]):
        s, n = acc[k][c]
        print("   %-34s avg/dispatch %16.1f  (n=%d)" % (c, s / n, n))
