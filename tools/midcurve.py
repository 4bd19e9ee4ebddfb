#!/usr/bin/env python3
"""Launch-bound sizes: us/sort for n = 2^10 .. 2^22 u32 keys (and pairs), mid-size path on/off, fresh random keys per sort."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oclradixsort_amd import Buffer, DeviceUtils, Pprims, Stopwatch
d = DeviceUtils.allocate(); p = Pprims()
kinds = sys.argv[1:] or ["u32", "kv"]
for kind in kinds:
    dtype, gen = (np.uint32, 0) if kind == "u32" else (np.uint64, 1)
    print("%-4s %10s %12s %12s %12s" % (kind, "n", "mid=1 us", "mid=3 us", "mid=0 us"))
    for n in [1 << lg for lg in range(10, 23)] + [20000, 100000, 300007, 1000003]:
        reps = 16
        bufs = [Buffer(d, n, dtype) for _ in range(reps)]
        res = []
        for mid in (1, 3, 0):
            d.setParam("sort.mid", mid)
            best = 1e9
            for trial in range(5):
                for i, b in enumerate(bufs): b.generate(n, seed=trial * 100 + i, kind=gen)
                DeviceUtils.waitForCompletion(d)
                sw = Stopwatch(d); sw.start()
                for r in range(reps): p.radixSort(d, bufs[r], n)
                sw.stop()
                best = min(best, sw.getMs() / reps)
            res.append(best * 1e3)
        print("%-4s %10d %12.1f %12.1f %12.1f" % ("", n, res[0], res[1], res[2]), flush=True)
        for b in bufs: b.release()
p.close(); DeviceUtils.deallocate(d)
