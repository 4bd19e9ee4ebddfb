#!/usr/bin/env python3
"""Per-kernel times of the large sort on keys whose low bits are constant (the LDS finish then sees one digit per pass)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oclradixsort_amd import Buffer, DeviceUtils, Pprims
import oracle
d = DeviceUtils.allocate(); p = Pprims()
n = 1 << 26
w = Buffer(d, n, np.uint32)
u = oracle.keys_u32(n, seed=3)
mild = np.where(np.arange(n) % 5 == 0, u >> np.uint32(1), u).astype(np.uint32)   # lower half of the range 1.2 x as dense
mild14 = np.where(np.arange(n) % 5 < 2, u >> np.uint32(1), u).astype(np.uint32)   # lower half 1.4 x as dense
for label, keys in (("low 16 bits zero", u & np.uint32(0xffff0000)), ("low 8 bits zero", u & np.uint32(0xffffff00)),
                    ("uniform", u), ("mild skew (1.2x)", mild), ("mild skew (1.4x)", mild14)):
    for rep in range(2):
        w.write(keys, n); DeviceUtils.waitForCompletion(d)
        d.toggleProfiling(True); d.profile(reset=True)
        p.radixSort(d, w, n)
        prof = d.profile(reset=True); d.toggleProfiling(False)
    ok = np.array_equal(w.toHost(), np.sort(keys))
    print("%-18s %s  %s" % (label, " ".join("%s %.1f" % (k, v[1] / v[0] * 1e3) for k, v in prof.items()), "OK" if ok else "MISMATCH"), flush=True)
