"""diagnostic: per-call durations (us) of the kernels whose name contains a pattern, in launch order, out of a rocprofv3 --kernel-trace directory
   python tools/kernel_seq.py <dir> <pattern> [<pattern> ...]"""
import csv, glob, sys
d = sys.argv[1]
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    for pat in sys.argv[2:]:
        v = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0 for r in rows if pat in r["Kernel_Name"]]
        print(pat, len(v), " ".join("%.0f" % x for x in v))
