#!/usr/bin/env python3
"""Per-kernel times of the large sort on sorted and on uniform keys (64 Mi u32)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oclradixsort_amd import Buffer, DeviceUtils, Pprims
import oracle
d = DeviceUtils.allocate(); p = Pprims()
n = 1 << 26
base = Buffer(d, n, np.uint32); base.generate(n, seed=5, kind=0); p.radixSort(d, base, n); DeviceUtils.waitForCompletion(d)
w = Buffer(d, n, np.uint32)
for label in ("sorted", "uniform"):
    for rep in range(2):
        if label == "sorted": w.write(base, n)
        else: w.generate(n, seed=9 + rep, kind=0)
        DeviceUtils.waitForCompletion(d)
        d.toggleProfiling(True); d.profile(reset=True)
        p.radixSort(d, w, n)
        prof = d.profile(reset=True); d.toggleProfiling(False)
        print(label, " ".join("%s %.1f" % (k, v[1] / v[0] * 1e3) for k, v in prof.items()), flush=True)
