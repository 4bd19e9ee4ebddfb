#!/usr/bin/env python3
"""Throughput-vs-n curve (the shape of Fig. 5 of the reference's paper): u32 keys, n = 2^10 .. 2^30."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oclradixsort_amd import Buffer, DeviceUtils, Pprims, Stopwatch
d = DeviceUtils.allocate(); p = Pprims()
print("%12s %10s %10s %10s" % ("n", "us/sort", "Gkeys/s", "algo"))
for lg in range(10, 31):
    n = 1 << lg
    # every timed sort gets FRESH random keys: a buffer is sorted once per trial (re-sorting a sorted buffer favours the
    # three-kernel pass and had put the automatic choice's threshold too high)
    reps = max(3 if lg < 29 else 2, min(32, (1 << 28) // n))
    bufs = [Buffer(d, n, np.uint32) for _ in range(reps)]
    for algo in (-1, 0, 1):
        d.setParam("sort.algo", algo)
        best = 1e9
        for trial in range(5 if n < (1 << 22) else 3):
            for i, b in enumerate(bufs): b.generate(n, seed=trial * 100 + i)
            DeviceUtils.waitForCompletion(d)
            sw = Stopwatch(d); sw.start()
            for r in range(reps): p.radixSort(d, bufs[r], n)
            sw.stop()
            best = min(best, sw.getMs() / reps)
        print("%12d %10.1f %10.2f %10s" % (n, best * 1e3, n / best / 1e6, {-1: "auto", 0: "onesweep", 1: "3-kernel"}[algo]), flush=True)
    for b in bufs: b.release()
p.close(); DeviceUtils.deallocate(d)
