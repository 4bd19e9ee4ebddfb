#!/usr/bin/env python3
"""Small and launch-bound sizes: us per sort (32 back-to-back sorts of fresh random keys, best of 6) and which kernels ran."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oclradixsort_amd import Buffer, DeviceUtils, Pprims, Stopwatch
d = DeviceUtils.allocate(); p = Pprims()
sizes = [int(x) for x in sys.argv[1:]] or [1024, 4096, 8192, 16384, 16385, 32768, 65536, 131072, 262144, 524288, 524289, 1048576, 2097152]
for n in sizes:
    bufs = [Buffer(d, n, np.uint32) for _ in range(32)]
    best = 1e9
    for t in range(6):   # fresh random keys for every timed sort (a sorted buffer sorts faster on some paths)
        for i, b in enumerate(bufs): b.generate(n, seed=n + 100 * t + i)
        DeviceUtils.waitForCompletion(d)
        sw = Stopwatch(d); sw.start()
        for b in bufs: p.radixSort(d, b, n)
        sw.stop(); best = min(best, sw.getMs() / len(bufs))
    d.toggleProfiling(True); d.profile(reset=True)
    p.radixSort(d, bufs[0], n)
    prof = d.profile(reset=True); d.toggleProfiling(False)
    print("%9d %7.1f us  %s" % (n, best * 1e3, {k: v[0] for k, v in prof.items()}))
    for b in bufs: b.release()
