#!/usr/bin/env python3
"""Where does the up-front histogram kernel's time go: its own LDS work, or the previous kernel's dirty lines?
Runs the SAME sort (64Mi u32 keys, one-sweep) in two positions, several times each:
  phase W ("behind a writer"): the sort follows a sort of another buffer at once -- its histogram kernel reads cold keys
                               while the caches still hold the 256 MiB the previous pass wrote;
  phase C ("clean caches")   : before the sort, a read-only kernel streams 1 GiB of other data (every dirty line of L2 and
                               of the Infinity Cache has been written back by the time it ends), then the device idles.
Under `rocprofv3 --kernel-trace` the per-dispatch durations of onesweep_hist_kernel fall into the two groups by order; under
`--pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE ...` so do the counters.  tools/hist_position_summary.py prints both.
Dispatch order per round: [probe_read x4] sort(A)  sort(B)   -> the hist of A is "C", the hist of B is "W"."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oclradixsort_amd import Buffer, DeviceUtils, Pprims, _lib
lib = _lib.load()
d = DeviceUtils.allocate(); p = Pprims()
d.setParam("sort.algo", 0)
n = 1 << 26
A, B = Buffer(d, n, np.uint32), Buffer(d, n, np.uint32)
big = Buffer(d, 1 << 28, np.uint32)          # 1 GiB of other data
sink = Buffer(d, 2, np.uint64); sink.clear()
big.generate(1 << 28, seed=9)
A.generate(n, seed=1); p.radixSort(d, A, n)  # warm-up: scratch allocation
DeviceUtils.waitForCompletion(d)
for rnd in range(6):
    A.generate(n, seed=10 + rnd); B.generate(n, seed=50 + rnd)
    DeviceUtils.waitForCompletion(d)
    lib.adlhip_probe_read(d._h, big.ptr(), (1 << 28) * 4, sink.ptr())
    DeviceUtils.waitForCompletion(d)
    time.sleep(0.01)
    p.radixSort(d, A, n)      # hist: clean caches ("C")
    p.radixSort(d, B, n)      # hist: right behind A's last pass ("W")
    DeviceUtils.waitForCompletion(d)
for x in (A, B, big, sink): x.release()
p.close(); DeviceUtils.deallocate(d)
