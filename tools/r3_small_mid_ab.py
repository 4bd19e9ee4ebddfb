#!/usr/bin/env python3
"""u32 keys just below the one-workgroup sort's limit (16 Ki): one workgroup | two-launch mid-size sort (ADLHIP_MID_MIN).
   ADLHIP_MID_MIN=4096 python tools/r3_small_mid_ab.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from oclradixsort_amd import Buffer, DeviceUtils, Pprims, Stopwatch
d = DeviceUtils.allocate(); p = Pprims()
d.setParam("sort.algo", -1)
print("ADLHIP_MID_MIN =", os.environ.get("ADLHIP_MID_MIN", "(default)"))
for n in (4097, 6000, 8192, 8193, 10000, 12288, 16384, 16385, 20000, 32768):
    reps = 32
    bufs = [Buffer(d, n, np.uint32) for _ in range(reps)]
    best = 1e9
    for trial in range(6):
        for i, b in enumerate(bufs): b.generate(n, seed=trial * 100 + i)
        DeviceUtils.waitForCompletion(d)
        sw = Stopwatch(d); sw.start()
        for r in range(reps): p.radixSort(d, bufs[r], n)
        sw.stop()
        best = min(best, sw.getMs() / reps)
    k = oracle.keys_u32(n, seed=5); bufs[0].write(k); p.radixSort(d, bufs[0], n)
    ok = np.array_equal(bufs[0].toHost(), oracle.sort_u32(k))
    print("%8d %8.1f us  %s" % (n, best * 1e3, "OK" if ok else "MISMATCH"), flush=True)
    for b in bufs: b.release()
p.close(); DeviceUtils.deallocate(d)
