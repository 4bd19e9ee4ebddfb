#!/usr/bin/env python3
"""Diagnostic: where the large sort's net spends its time on u32 keys of D values (the larger dictionary, dict_big_kernels.hpp):
phase boundaries as workgroup 0 sees them ("debug.net_stamp<k>", 10-ns ticks): 0 net start, 1 before the build, 2 built, 3 behind the
barrier, 4 counted, 5 behind the barrier, 6 filled."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oclradixsort_amd import Buffer, DeviceUtils, Pprims, Stopwatch
d = DeviceUtils.allocate(); p = Pprims()
rng = np.random.RandomState(1)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 26
def rv(k): return np.unique(rng.randint(0, 2**32, 2 * k, dtype=np.uint64).astype(np.uint32))[:k]
for D in (300, 1000, 4096):
    a = rv(D)[rng.randint(0, D, n)]
    b = Buffer(d, n, np.uint32)
    for t in range(2):
        b.write(a); DeviceUtils.waitForCompletion(d)
        sw = Stopwatch(d); sw.start(); p.radixSort(d, b, n); sw.stop()
    st = [d.getParam("debug.net_stamp%d" % k) for k in range(7)]
    us = [((st[k + 1] - st[k]) & 0x7fffffff) / 100.0 for k in range(6)]
    ex = [d.getParam("debug.net_stamp%d" % k) for k in (7, 8, 9)]
    print("   inside the count (workgroup 0): tables into LDS %.0f us | keys %.0f | waiting for the other waves %.0f | counters out %.0f" %
          (((ex[0] - st[3]) & 0x7fffffff) / 100.0, ((ex[1] - ex[0]) & 0x7fffffff) / 100.0, ((ex[2] - ex[1]) & 0x7fffffff) / 100.0,
           ((st[4] - ex[2]) & 0x7fffffff) / 100.0))
    print("D=%d n=%d sort %.3f ms; us: small-dictionary attempt %.0f | build %.0f | barrier %.0f | count %.0f | barrier %.0f | fill %.0f" % ((D, n, sw.getMs()) + tuple(us)), flush=True)
    b.release()
p.close(); DeviceUtils.deallocate(d)
