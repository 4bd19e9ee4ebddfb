#!/usr/bin/env python3
"""Per-kernel durations of the large keys-only sort (hipEvent pair around every launch): n and key shift from argv."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oclradixsort_amd import Buffer, DeviceUtils, Pprims
import oracle
d = DeviceUtils.allocate(); p = Pprims()
d.setParam("sort.msd2", 2)
for arg in sys.argv[1:] or ["67108864:0"]:
    n, shift = (int(x) for x in arg.split(":"))
    bufs = [Buffer(d, n, np.uint32) for _ in range(6)]
    for i, b in enumerate(bufs):
        if shift: b.write(oracle.keys_u32(n, seed=i) >> np.uint32(shift), n)
        else: b.generate(n, seed=i, kind=0)
    p.radixSort(d, bufs[0], n)
    DeviceUtils.waitForCompletion(d)
    d.toggleProfiling(True); d.profile(reset=True)
    for b in bufs[1:]: p.radixSort(d, b, n)
    prof = d.profile(reset=True); d.toggleProfiling(False)
    print("n = %d, keys >> %d: " % (n, shift) + "  ".join("%s %.1f us" % (k, v[1] / v[0] * 1e3) for k, v in prof.items()), flush=True)
    for b in bufs: b.release()
p.close(); DeviceUtils.deallocate(d)
