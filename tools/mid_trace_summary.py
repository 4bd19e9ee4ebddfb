#!/usr/bin/env python3
import csv, glob, os, sys
root = sys.argv[1]
rows = []
for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
rows = [r for r in rows if "generate_keys" not in r["Kernel_Name"] and "selftest" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-32:]          # the last 8 sorts (4 kernels each)
prev_end = None
for r in rows[:12]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void adlhip::", "")[:60]
    print("%-62s dur %6.2f us   gap before %6.2f us   grid %s wg %s" % (name, (e - s) / 1e3, (s - prev_end) / 1e3 if prev_end else 0, r.get("Grid_Size_X", "?"), r.get("Workgroup_Size_X", "?")))
    prev_end = e
span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3 / (len(rows) / 4)
print("per sort (start of first kernel to end of last, averaged over %d sorts): %.1f us" % (len(rows) // 4, span))
