#!/usr/bin/env python3
"""Per-kernel times of the mid-size sort (hipEvent pair around every launch)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oclradixsort_amd import Buffer, DeviceUtils, Pprims
d = DeviceUtils.allocate(); p = Pprims()
for n in (32768, 65536, 262144, 1 << 20, 1 << 21):
    bufs = [Buffer(d, n, np.uint32) for _ in range(8)]
    for mid in (1, 0):
        d.setParam("sort.mid", mid)
        for rep in range(2):
            for i, b in enumerate(bufs): b.generate(n, seed=rep * 10 + i)
            DeviceUtils.waitForCompletion(d)
            d.toggleProfiling(rep == 1); d.profile(reset=True)
            for b in bufs: p.radixSort(d, b, n)
            prof = d.profile(reset=True)
        d.toggleProfiling(False)
        print(n, "mid=%d" % mid, " ".join("%s=%.1fus(x%d)" % (k, v[1] / v[0] * 1e3, v[0] // 8) for k, v in prof.items()), flush=True)
    for b in bufs: b.release()
p.close(); DeviceUtils.deallocate(d)
