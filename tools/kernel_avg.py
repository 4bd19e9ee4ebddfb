#!/usr/bin/env python3
"""Average duration per kernel name from a rocprofv3 --kernel-trace output directory (optionally only names containing argv[2])."""
import csv, glob, os, sys, collections
root = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
d = collections.defaultdict(list)
for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void adlhip::", "")
        if pat in name:
            d[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for name, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print("%-70s calls %5d  avg %8.2f us  min %8.2f  median %8.2f" % (name[:70], len(v), sum(v) / len(v), v[0], v[len(v) // 2]))
