#!/usr/bin/env python3
"""Pprims::scan throughput (exclusive prefix sum of u32): algorithmic bytes = 8 B/element (read + write)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oclradixsort_amd import Buffer, DeviceUtils, Pprims, Stopwatch
d = DeviceUtils.allocate(); p = Pprims()
d.toggleProfiling(False)
print("%12s %10s %12s %10s" % ("n", "us/scan", "Gelem/s", "GB/s(8B)"))
for lg in (10, 14, 17, 20, 22, 24, 26, 28):
    n = 1 << lg
    src = Buffer(d, n, np.uint32); dst = Buffer(d, n, np.uint32)
    src.generate(n, seed=3)
    reps = max(5, min(200, (1 << 28) // n))
    p.scan(d, dst, src, n); DeviceUtils.waitForCompletion(d)
    sw = Stopwatch(d); sw.start()
    for _ in range(reps): p.scan(d, dst, src, n)
    sw.stop()
    ms = sw.getMs() / reps
    print("%12d %10.1f %12.2f %10.1f" % (n, ms * 1e3, n / ms / 1e6, 8.0 * n / ms / 1e6), flush=True)
    src.release(); dst.release()
d.toggleProfiling(True); d.profile(reset=True)
n = 1 << 26
src = Buffer(d, n, np.uint32); dst = Buffer(d, n, np.uint32); src.generate(n, seed=3)
for _ in range(10): p.scan(d, dst, src, n)
print({k: round(v[1] / v[0], 4) for k, v in d.profile(reset=True).items()})
src.release(); dst.release(); p.close(); DeviceUtils.deallocate(d)
