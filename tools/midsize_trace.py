#!/usr/bin/env python3
"""Mid-size sorts (launch-bound regime): run `reps` back-to-back sorts of n keys; with --analyze DIR, read a
rocprofv3 --kernel-trace CSV of that run and print GPU-busy time vs wall span per sort.
  rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/midsize_trace.py --n 524288 --algo 1
  python3 tools/midsize_trace.py --analyze DIR"""
import argparse, csv, glob, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1 << 19)
ap.add_argument("--algo", type=int, default=-1)
ap.add_argument("--reps", type=int, default=200)
ap.add_argument("--analyze", default="")
args = ap.parse_args()
if args.analyze:
    rows = []
    for f in glob.glob(os.path.join(args.analyze, "**", "*kernel_trace.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    rows = [r for r in rows if "generate" not in r[2] and "selftest" not in r[2]]
    # steady state: skip the first quarter
    rows = rows[len(rows) // 4:]
    busy = sum(e - s for s, e, _ in rows)
    span = rows[-1][1] - rows[0][0]
    gaps = [rows[i + 1][0] - rows[i][1] for i in range(len(rows) - 1)]
    names = {}
    for s, e, k in rows:
        k = k.split("(")[0][-60:]
        a = names.setdefault(k, [0, 0]); a[0] += 1; a[1] += e - s
    print("kernels %d  span %.1f us  busy %.1f us (%.0f%%)  mean gap %.2f us  median gap %.2f us" %
          (len(rows), span / 1e3, busy / 1e3, 100.0 * busy / span, np.mean(gaps) / 1e3, np.median(gaps) / 1e3))
    for k, (c, t) in sorted(names.items(), key=lambda kv: -kv[1][1]):
        print("  %-62s calls %5d  avg %.2f us" % (k, c, t / c / 1e3))
    sys.exit(0)
from oclradixsort_amd import Buffer, DeviceUtils, Pprims, Stopwatch
d = DeviceUtils.allocate(); p = Pprims()
d.setParam("sort.algo", args.algo)
bufs = [Buffer(d, args.n, np.uint32) for _ in range(8)]
for i, b in enumerate(bufs): b.generate(args.n, seed=i)
DeviceUtils.waitForCompletion(d)
for trial in range(2):
    sw = Stopwatch(d); sw.start()
    for r in range(args.reps): p.radixSort(d, bufs[r % 8], args.n)
    sw.stop(); DeviceUtils.waitForCompletion(d)
    print("n=%d algo=%d: %.1f us/sort" % (args.n, args.algo, sw.getMs() / args.reps * 1e3))
