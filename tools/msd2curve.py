#!/usr/bin/env python3
"""Large keys-only sort: ms/sort for u32 keys with "sort.msd2" forced on (2) and off (0), fresh random keys per sort."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oclradixsort_amd import Buffer, DeviceUtils, Pprims, Stopwatch
d = DeviceUtils.allocate(); p = Pprims()
sizes = [int(float(a)) for a in sys.argv[1:]] or [1 << 22, 6 << 20, 1 << 23, 12 << 20, 1 << 24, 24 << 20, 1 << 25, 48 << 20, 1 << 26]
print("%12s %12s %12s %10s %10s" % ("n", "msd2=2 ms", "msd2=0 ms", "Gkeys/s", "Gkeys/s"))
for n in sizes:
    reps = 4
    bufs = [Buffer(d, n, np.uint32) for _ in range(reps)]
    res = []
    for mode in (2, 0):
        d.setParam("sort.msd2", mode)
        best = 1e9
        for trial in range(5):
            for i, b in enumerate(bufs): b.generate(n, seed=trial * 100 + i, kind=0)
            DeviceUtils.waitForCompletion(d)
            sw = Stopwatch(d); sw.start()
            for r in range(reps): p.radixSort(d, bufs[r], n)
            sw.stop()
            best = min(best, sw.getMs() / reps)
        res.append(best)
    print("%12d %12.4f %12.4f %10.2f %10.2f" % (n, res[0], res[1], n / res[0] / 1e6, n / res[1] / 1e6), flush=True)
    for b in bufs: b.release()
d.checkFault()
p.close(); DeviceUtils.deallocate(d)
