#!/usr/bin/env python3
"""Summarise tools/hist_position.py runs: usage: hist_position_summary.py <dir with rocprofv3 csv output>"""
import csv, glob, os, sys, collections
root = sys.argv[1]
def rows(pattern):
    for f in glob.glob(os.path.join(root, "**", pattern), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                yield r
# kernel trace: durations of the hist kernel in dispatch order; dispatches alternate C, W after the warm-up sort
tr = [r for r in rows("*kernel_trace.csv") if "onesweep_hist_kernel" in r.get("Kernel_Name", "")]
tr.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in tr]
if dur:
    body = dur[1:]                      # drop the warm-up sort's
    c, w = body[0::2], body[1::2]
    print("onesweep_hist_kernel, 64Mi u32 keys: clean caches %.1f us (n=%d: %s)   behind a 256-MiB writer %.1f us (n=%d: %s)"
          % (sum(c) / len(c), len(c), " ".join("%.0f" % x for x in c), sum(w) / len(w), len(w), " ".join("%.0f" % x for x in w)))
cc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows("*counter_collection.csv"):
    if "onesweep_hist_kernel" in r.get("Kernel_Name", ""):
        cc[r["Counter_Name"]][int(r["Dispatch_Id"])].append(float(r["Counter_Value"]))
for name in sorted(cc):
    ids = sorted(cc[name])
    vals = [sum(cc[name][i]) for i in ids][1:]
    c, w = vals[0::2], vals[1::2]
    if c and w:
        print("  %-26s clean %14.0f   behind writer %14.0f" % (name, sum(c) / len(c), sum(w) / len(w)))
