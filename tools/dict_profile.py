#!/usr/bin/env python3
"""Per-kernel durations of the counting sort (few distinct values) at 64 Mi u32 keys: all equal / 16 values / 256 values."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oclradixsort_amd import Buffer, DeviceUtils, Pprims
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 26
idx = (np.arange(n, dtype=np.uint32) * np.uint32(2654435761))
rest = np.full(n, 0x12345678, dtype=np.uint32); rest[::64] = 0x0badf00d
for name, a in (("all_equal", np.full(n, 0x12345678, dtype=np.uint32)), ("all_equal_0", np.zeros(n, dtype=np.uint32)),
                ("2_values", (idx >> np.uint32(31)) * np.uint32(0x71717171)), ("one_in_64_differs", rest),
                ("16_values", (idx >> np.uint32(28)) * np.uint32(0x11111111)),
                ("256_values", (idx >> np.uint32(24)) * np.uint32(0x01010101))):
    d = DeviceUtils.allocate(); p = Pprims()
    b = Buffer(d, n, np.uint32)
    for rep in range(3):
        b.write(a); DeviceUtils.waitForCompletion(d)
        d.toggleProfiling(True); d.profile(reset=True)
        p.radixSort(d, b, n); DeviceUtils.waitForCompletion(d)
        prof = d.profile(reset=True); d.toggleProfiling(False)
    print("%-12s " % name + "  ".join("%s %.1f us" % (k, v[1] / v[0] * 1e3) for k, v in prof.items()), flush=True)
    b.release(); p.close(); DeviceUtils.deallocate(d)
