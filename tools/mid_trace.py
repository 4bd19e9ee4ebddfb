#!/usr/bin/env python3
"""Run a few mid-size sorts back to back (for rocprofv3 --kernel-trace): n from argv."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oclradixsort_amd import Buffer, DeviceUtils, Pprims
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
d = DeviceUtils.allocate(); p = Pprims()
bufs = [Buffer(d, n, np.uint32) for _ in range(12)]
for i, b in enumerate(bufs): b.generate(n, seed=i)
DeviceUtils.waitForCompletion(d)
for b in bufs: p.radixSort(d, b, n)
DeviceUtils.waitForCompletion(d)
for b in bufs: b.release()
p.close(); DeviceUtils.deallocate(d)
