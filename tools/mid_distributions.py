#!/usr/bin/env python3
"""Mid-size sort on friendly and skewed inputs: us/sort averaged over 48 sorts on a FRESH device handle per case (the
handle's hints start neutral), with the mid-size path on ("sort.mid" = 1: two-launch keys form, three-launch form and
per-digit passes chosen by the hints) and off (0)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oclradixsort_amd import Buffer, DeviceUtils, Pprims, Stopwatch
rng = np.random.RandomState(1)
for n in (1 << 18, 1 << 20):
    u = rng.randint(0, 2**32, n, dtype=np.uint64)
    cases = {
        "uniform": u,
        "below 2^24 (small indices)": u >> 8,
        "below 2^12": u >> 20,
        "two clusters (int32 +-small)": np.where(u & 1, u >> 16, 0xffff0000 | (u >> 16)),
        "two values": (u & 1) * 0xffffffff,
        "all equal": np.full(n, 12345, dtype=np.uint64),
        "sorted": np.sort(u),
        "top byte only": (u >> 24) << 24,
        "exponential": (rng.exponential(2.0**24, n)).astype(np.uint64) & 0xffffffff,
        "90% in one top byte": np.where(rng.rand(n) < 0.9, u >> 8, u),
    }
    print("n = %d" % n)
    for nm, k in cases.items():
        k = k.astype(np.uint32)
        res = []
        mids = (1, 0, 2, 3) if "--forced" in sys.argv else (1, 0)   # 2 / 3: every sort through the two- / three-launch form (and its net)
        for mid in mids:
            d = DeviceUtils.allocate(); p = Pprims()
            d.setParam("sort.mid", mid)
            bufs = [Buffer(d, n, np.uint32) for _ in range(8)]
            total = 0.0
            for trial in range(6):
                for b in bufs: b.write(k)
                DeviceUtils.waitForCompletion(d)
                sw = Stopwatch(d); sw.start()
                for b in bufs: p.radixSort(d, b, n)
                sw.stop()
                total += sw.getMs()
            ok = np.array_equal(bufs[0].toHost(), np.sort(k))
            res.append((total / 48 * 1e3, ok))
            for b in bufs: b.release()
            p.close(); DeviceUtils.deallocate(d)
        print("  %-30s " % nm + "   ".join("mid=%d %8.1f us %s" % (m, r[0], "OK" if r[1] else "WRONG") for m, r in zip(mids, res)), flush=True)
